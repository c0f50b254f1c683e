"""``DiscreteTransport``: entropic optimal transport between two fitted codebooks (reference
ot/transport/discrete_transport.py:28-99).  ``compute`` fits both ``CodebookModel``s, takes the atom-to-atom cost the
reference takes (``source_model.energy(target codebook)``, i.e. 1 / (distance + 1e-8), discrete_transport.py:59) and
solves for the K x K plan with the HIP Sinkhorn solver; ``transport`` assigns every input to source atoms (HIP
assignment kernels, inference mode), pushes the assignment through the plan and reads the target atoms."""
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.distributions import Categorical

from ..distribution_models.codebook_model import CodebookModel
from ..w2_utils import sinkhorn_log
from .base import TransportOperator

__all__ = ["DiscreteTransport"]


class DiscreteTransport(TransportOperator):
    def __init__(self, *size: int, source_cfg={}, target_cfg={}, transport_type: str, sinkhorn_reg: float = 1e-5,
                 sinkhorn_max_iter: int = 1000, sinkhorn_threshold: float = 1e-6, **kwargs):
        if transport_type not in ("sample", "argmax", "mean"):
            raise NotImplementedError(f"`transport_type` must be 'sample', 'argmax' or 'mean', got {transport_type!r}")
        super().__init__(*size, source_model=CodebookModel(*size, **source_cfg),
                         target_model=CodebookModel(*size, **target_cfg), **kwargs)
        self.transport_type = transport_type
        self.sinkhorn_reg, self.sinkhorn_max_iter, self.sinkhorn_threshold = sinkhorn_reg, sinkhorn_max_iter, sinkhorn_threshold
        self.transport_matrix = None

    def reset(self) -> None:
        super().reset()
        self.transport_matrix = None

    def compute(self) -> Tensor:
        self.fit_models()
        cost = self.source_model.energy(self.target_model.codebook)                       # [*, K_s, K_t]
        self.transport_matrix = sinkhorn_log(self.source_distribution.probs, self.target_distribution.probs, cost,
                                             reg=self.sinkhorn_reg, max_iter=self.sinkhorn_max_iter,
                                             threshold=self.sinkhorn_threshold)
        return torch.sum(cost * self.transport_matrix, dim=(-2, -1))

    def transport(self, inputs: Tensor) -> Tensor:
        if self.transport_matrix is None:
            raise RuntimeError("call `compute()` before `transport()`")
        training = self.training
        self.eval()                                                                       # inference-mode assignment
        try:
            assignments, _, _ = self.source_model.assign(inputs)                          # [*, B, K_s]
        finally:
            self.train(training)
        moved = assignments.type_as(self.transport_matrix) @ self.transport_matrix        # [*, B, K_t]
        if self.transport_type == "argmax":       # every input goes to the target atom it is most coupled with
            moved = F.one_hot(moved.argmax(-1), moved.size(-1)).type_as(moved)
        elif self.transport_type == "sample":     # ... or to one drawn from its row of the plan
            moved = F.one_hot(Categorical(moved).sample(), moved.size(-1)).type_as(moved)
        return (moved.type_as(self.target_model.codebook) @ self.target_model.codebook).type_as(inputs)

    def extra_repr(self) -> str:
        return super().extra_repr() + (f", sinkhorn_reg={self.sinkhorn_reg}, sinkhorn_max_iter={self.sinkhorn_max_iter}, "
                                       f"sinkhorn_threshold={self.sinkhorn_threshold}")
