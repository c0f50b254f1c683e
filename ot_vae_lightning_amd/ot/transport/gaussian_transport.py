"""``GaussianTransport``: closed-form W2 transport between two fitted Gaussians (reference
ot/transport/gaussian_transport.py): ``compute`` = fit + squared W2 + eq. 17 operator, ``transport`` = affine map."""
from torch import Tensor

from ..distribution_models.gaussian_model import GaussianModel
from ..w2_utils import W2Mixin
from .base import TransportOperator

__all__ = ["GaussianTransport"]


class GaussianTransport(TransportOperator, W2Mixin):
    def __init__(self, *size, source_cfg={}, target_cfg={}, transport_cfg={}, **kwargs):
        W2Mixin.__init__(self, **dict(transport_cfg))
        TransportOperator.__init__(
            self, *size,
            source_model=GaussianModel(*size, w2_cfg=dict(transport_cfg), **source_cfg),
            target_model=GaussianModel(*size, w2_cfg=dict(transport_cfg), **target_cfg),
            **kwargs)
        self.transport_operator = None
        self.cov_stochastic_noise = None

    def reset(self) -> None:
        super().reset()
        self.transport_operator = None
        self.cov_stochastic_noise = None

    def compute(self) -> Tensor:
        self.fit_models()
        cs, ct = self.source_model.cov, self.target_model.cov
        w2 = self.w2_gaussian(self.source_model.mean, self.target_model.mean, cs, ct)
        self.transport_operator, self.cov_stochastic_noise = self.compute_transport_operators(cs, ct)
        return w2

    def transport(self, inputs: Tensor) -> Tensor:
        if inputs.size(-1) != self.dim:
            raise ValueError("`inputs` dimensionality must match the model dimensionality")
        lead = tuple(self.leading_shape)
        if tuple(inputs.shape[:-2]) != lead and tuple(inputs.shape[:-1]) != lead:
            raise ValueError("`inputs` leading dims must match the model batch_shape with optional trailing batch dimensions")
        if self.transport_operator is None:
            raise RuntimeError("call `compute()` before `transport()`")
        batched = inputs.dim() == len(lead) + 2
        out = self.apply_transport(inputs, self.source_model.mean, self.target_model.mean, self.transport_operator,
                                   self.cov_stochastic_noise, batch_dim=-2 if batched else None)
        return out.type_as(inputs)

    def extra_repr(self) -> str:
        return super().extra_repr() + W2Mixin.__repr__(self)
