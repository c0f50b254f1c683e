"""``DiscreteTransport``: entropic optimal transport between two fitted codebooks (reference
ot/transport/discrete_transport.py:28-99).

    compute()     fit both ``CodebookModel`` s; cost[i, j] = source_model.energy(target atoms) -- the reference's choice
                  (discrete_transport.py:58), i.e. 1 / (distance + 1e-8), kept as is; plan = HIP log-domain Sinkhorn
                  (``otvae_sinkhorn_log``) between the two atom-weight vectors; returns <cost, plan> per operator.
    transport(x)  soft-assign x to the source atoms in inference mode (``otvae_codebook_probs``), push the assignment
                  through the plan, then read target atoms:
                      'mean'    the plan-weighted average of the target atoms          (one GEMM)
                      'argmax'  the single most-coupled target atom                   (row gather, no one-hot GEMM)
                      'sample'  one target atom drawn from the pushed assignment      (row gather)
"""
from contextlib import contextmanager

import torch
from torch import Tensor
from torch.distributions import Categorical

from ..distribution_models.codebook_model import CodebookModel
from ..matrix_utils import mm
from ..w2_utils import sinkhorn_log
from .base import TransportOperator

__all__ = ["DiscreteTransport"]

_TRANSPORT_TYPES = ("sample", "argmax", "mean")


def _rows(table: Tensor, index: Tensor) -> Tensor:
    """table [*, K, d], index [*, B] -> table rows [*, B, d]"""
    return torch.gather(table, -2, index.unsqueeze(-1).expand(*index.shape, table.size(-1)))


class DiscreteTransport(TransportOperator):
    def __init__(self, *size: int, source_cfg={}, target_cfg={}, transport_type: str, sinkhorn_reg: float = 1e-5,
                 sinkhorn_max_iter: int = 1000, sinkhorn_threshold: float = 1e-6, **kwargs):
        if transport_type not in _TRANSPORT_TYPES:
            raise NotImplementedError(f"`transport_type` must be one of {_TRANSPORT_TYPES}, got {transport_type!r}")
        codebooks = dict(source_model=CodebookModel(*size, **source_cfg), target_model=CodebookModel(*size, **target_cfg))
        super().__init__(*size, **codebooks, **kwargs)
        self.transport_type = transport_type
        self.sinkhorn_reg, self.sinkhorn_max_iter, self.sinkhorn_threshold = sinkhorn_reg, sinkhorn_max_iter, sinkhorn_threshold
        self.transport_matrix = None            # [*, K_source, K_target] after compute()

    def _sinkhorn_cfg(self) -> dict:
        return dict(reg=self.sinkhorn_reg, max_iter=self.sinkhorn_max_iter, threshold=self.sinkhorn_threshold)

    def reset(self) -> None:
        super().reset()
        self.transport_matrix = None

    def compute(self) -> Tensor:
        self.fit_models()
        cost = self.source_model.energy(self.target_model.codebook)
        weights = self.source_distribution.probs, self.target_distribution.probs
        self.transport_matrix = plan = sinkhorn_log(*weights, cost, **self._sinkhorn_cfg())
        return (cost * plan).sum(dim=(-2, -1))

    @contextmanager
    def _inference_mode(self):
        was_training = self.training
        self.eval()
        try:
            yield
        finally:
            self.train(was_training)

    def transport(self, inputs: Tensor) -> Tensor:
        plan = self.transport_matrix
        if plan is None:
            raise RuntimeError("call `compute()` before `transport()`")
        with self._inference_mode():
            assignment = self.source_model.assign(inputs)[0]                    # [*, B, K_source]
        pushed = mm(assignment.type_as(plan), plan)                                # [*, B, K_target]
        atoms = self.target_model.codebook
        if self.transport_type == "mean":
            moved = mm(pushed.type_as(atoms), atoms)
        else:
            chosen = pushed.argmax(-1) if self.transport_type == "argmax" else Categorical(pushed).sample()
            moved = _rows(atoms, chosen)
        return moved.type_as(inputs)

    def extra_repr(self) -> str:
        cfg = ", ".join(f"sinkhorn_{k}={v}" for k, v in self._sinkhorn_cfg().items())
        return f"{super().extra_repr()}, transport_type={self.transport_type}, {cfg}"
