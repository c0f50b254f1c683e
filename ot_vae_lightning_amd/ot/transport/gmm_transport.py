"""``GMMTransport``: transport between two fitted Gaussian mixtures (reference ot/transport/gmm_transport.py:28-124,
after Chen, Georgiou & Tannenbaum): ``compute`` couples the components with the entropic plan of ``batch_ot_gmm``;
``transport`` assigns every input to source components by likelihood, moves the assignment through the plan, picks
the target component ('argmax' / 'sample') and applies the closed-form Gaussian map between the two selected
components to each input; 'barycenter' moves every input to the W2 barycentre of ALL target components weighted by its
coupled assignment (``gaussian_barycenter``).  Diagonal or full covariances (``transport_cfg['diag']``)."""
import torch

from ..matrix_utils import mm
import torch.nn.functional as F
from torch import Tensor
from torch.distributions import Categorical

from ..distribution_models.gaussian_mixture_model import GaussianMixtureModel
from ..w2_utils import W2Mixin, batch_ot_gmm
from .base import TransportOperator

__all__ = ["GMMTransport"]


class GMMTransport(TransportOperator, W2Mixin):
    def __init__(self, *size, transport_type: str, source_cfg={}, target_cfg={}, transport_cfg={}, **kwargs):
        if transport_type not in ("sample", "argmax", "barycenter"):
            raise NotImplementedError(f"`transport_type` must be 'sample', 'argmax' or 'barycenter', got {transport_type!r}")
        W2Mixin.__init__(self, **dict(transport_cfg))
        TransportOperator.__init__(self, *size,
                                   source_model=GaussianMixtureModel(*size, w2_cfg=dict(transport_cfg), **source_cfg),
                                   target_model=GaussianMixtureModel(*size, w2_cfg=dict(transport_cfg), **target_cfg),
                                   **kwargs)
        self.transport_type = transport_type
        self.transport_matrix = None

    def reset(self) -> None:
        super().reset()
        self.transport_matrix = None

    def compute(self) -> Tensor:
        self.fit_models()
        total, coupling = batch_ot_gmm(self.source_model.mean, self.target_model.mean, self.source_model.variances,
                                       self.target_model.variances, diag=self.diag, weight_source=self.source_model.weights,
                                       weight_target=self.target_model.weights, verbose=self.verbose, dtype=self.dtype,
                                       max_iter=100)
        self.transport_matrix = coupling.type_as(self.source_model.mean)
        return total

    @torch.no_grad()
    def transport(self, inputs: Tensor) -> Tensor:
        if self.transport_matrix is None:
            raise RuntimeError("call `compute()` before `transport()`")
        assignments, _, _ = self.source_model.assign(inputs.to(self.dtype))                 # [*, B, K_s]
        source_means, source_vars = self.source_model.predict_mean_var(assignments)
        moved = mm(assignments.type_as(self.transport_matrix).contiguous(), self.transport_matrix)   # [*, B, K_t], otvae_gemm_*
        if self.transport_type == "barycenter":
            # a smooth interpolation of all the target components, weighted by each input's coupled assignment
            # (gmm_transport.py:103-110); `barycenter_init` fixes the start of the full-covariance fixed point (parity tests)
            target_means, target_vars = self.gaussian_barycenter(
                self.target_model.mean.unsqueeze(-3), self.target_model.batched_variances,
                moved / moved.sum(-1, keepdim=True), n_iter=100, init_index=getattr(self, "barycenter_init", None))
        else:
            if self.transport_type == "argmax":
                idx = moved.argmax(-1)
            else:
                idx = Categorical(moved / moved.sum(-1, keepdim=True)).sample()
            target_means, target_vars = self.target_model.predict_mean_var(F.one_hot(idx, moved.size(-1)).type_as(moved))
        T, Cw = self.compute_transport_operators(source_vars, target_vars)
        return self.apply_transport(inputs, source_means, target_means, T, Cw).type_as(inputs)

    def extra_repr(self) -> str:
        return super().extra_repr() + W2Mixin.__repr__(self) + f", transport_type={self.transport_type}"
