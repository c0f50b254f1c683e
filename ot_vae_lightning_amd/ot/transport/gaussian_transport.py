"""``GaussianTransport``: W2-optimal affine map between two fitted Gaussians (reference ot/transport/gaussian_transport.py:23-98).

Device work, all behind ``W2Mixin`` / ``GaussianModel`` (see ``ot/w2_utils.py`` of this package for the kernel map):

    update     otvae_gauss_stats        running sum / outer-product sum per side (fp64 accumulators)
    compute    otvae_mean_cov           -> (mean, cov) per side
               otvae_eigh_fn (one decomposition per covariance, eigenvector output), otvae_gemm_f64, otvae_w2_tail
                                        -> squared W2 distance, the eq. 17 operator T and the noise covariance
    transport  otvae_apply_transport    (x - mean_s) @ T^T + mean_t (+ noise)

``compute()`` leaves its result in ``transport_operator`` / ``cov_stochastic_noise`` (the reference's attribute names);
both are ``None`` until then and after ``reset()``.
"""
from typing import Optional, Tuple

from torch import Tensor

from ..distribution_models.gaussian_model import GaussianModel
from ..w2_utils import W2Mixin, w2_and_transport_operator
from .base import TransportOperator

__all__ = ["GaussianTransport"]


class GaussianTransport(TransportOperator, W2Mixin):
    def __init__(self, *size, source_cfg={}, target_cfg={}, transport_cfg={}, **kwargs):
        w2_cfg = dict(transport_cfg)
        W2Mixin.__init__(self, **w2_cfg)
        sides = {f"{side}_model": GaussianModel(*size, w2_cfg=dict(w2_cfg), **cfg)
                 for side, cfg in (("source", source_cfg), ("target", target_cfg))}
        TransportOperator.__init__(self, *size, **sides, **kwargs)
        self._set_operators(None, None)

    def _set_operators(self, operator: Optional[Tensor], noise_cov: Optional[Tensor]) -> None:
        self.transport_operator, self.cov_stochastic_noise = operator, noise_cov

    def _moments(self) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        s, t = self.source_model, self.target_model
        return s.mean, t.mean, s.cov, t.cov

    def reset(self) -> None:
        TransportOperator.reset(self)
        self._set_operators(None, None)

    def compute(self) -> Tensor:
        self.fit_models()
        if not (self.diag or self.stochastic):
            # full matrices: distance and operator share the eigendecompositions of the two covariances (4 Jacobi runs
            # instead of the 10 that `.cov` + w2_gaussian + compute_transport_operators make between them)
            s, t = self.source_model, self.target_model
            spec_s, spec_t = type(s).cov_spectra(s, t)   # both covariances decomposed by one batched launch
            distance, operator, noise_cov = w2_and_transport_operator(
                s.mean, t.mean, spec_s, spec_t, pg_star=self.pg_star, make_pd=self.make_pd, dtype=self.dtype)
            self._set_operators(operator, noise_cov)
            return distance
        mean_s, mean_t, cov_s, cov_t = self._moments()
        distance = self.w2_gaussian(mean_s, mean_t, cov_s, cov_t)
        self._set_operators(*self.compute_transport_operators(cov_s, cov_t))
        return distance

    def _sample_axis(self, inputs: Tensor) -> Optional[int]:
        """-2 when ``inputs`` is [*leading_shape, B, dim], None when it is one sample per operator [*leading_shape, dim]"""
        if inputs.size(-1) != self.dim:
            raise ValueError("`inputs` dimensionality must match the model dimensionality")
        lead = tuple(self.leading_shape)
        for extra, axis in ((2, -2), (1, None)):
            if inputs.dim() == len(lead) + extra and tuple(inputs.shape[:-extra]) == lead:
                return axis
        raise ValueError("`inputs` leading dims must match the model batch_shape with optional trailing batch dimensions")

    def transport(self, inputs: Tensor) -> Tensor:
        axis = self._sample_axis(inputs)
        if self.transport_operator is None:
            raise RuntimeError("call `compute()` before `transport()`")
        # the means only: reading a model's `cov` evaluates its positive-definite parametrisation (an eigendecomposition)
        moved = self.apply_transport(inputs, self.source_model.mean, self.target_model.mean, self.transport_operator,
                                     self.cov_stochastic_noise, batch_dim=axis)
        return moved.type_as(inputs)

    def extra_repr(self) -> str:
        return super().extra_repr() + W2Mixin.__repr__(self)
