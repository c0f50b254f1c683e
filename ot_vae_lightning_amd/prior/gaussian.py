"""``GaussianPrior``: q(z|x) = N(mu(x), exp(log_var(x)/2)), p = N(0, I), closed-form KL(q||p)
(reference prior/gaussian.py:24-102) on the fused MI355X kernel (re-parametrisation + KL + loss scaling in one
pass, explicit backward)."""
from typing import Optional

import torch
from torch import Tensor
from torch.distributions import Normal

from .. import functional as HF
from .base import Prior

__all__ = ["GaussianPrior"]


class _LazyNormal:
    """``Normal(loc, scale)`` built on first use.  The reference hands torch Distributions back as artifacts
    (prior/gaussian.py:95); constructing them eagerly costs extra launches and an argument check that synchronises
    the stream, so the training step only pays for them if somebody looks."""

    def __init__(self, loc_fn, scale_fn):
        self._fns, self._d = (loc_fn, scale_fn), None

    def _get(self) -> Normal:
        if self._d is None:
            self._d = Normal(self._fns[0](), self._fns[1](), validate_args=False)
        return self._d

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __repr__(self):
        return repr(self._get())


class GaussianPrior(Prior):
    def __init__(self, loss_coeff: float = 1., empirical_kl: bool = False, reparam_dim: int = 1,
                 annealing_steps: int = 0, fixed_var: bool = False):
        super().__init__(loss_coeff, annealing_steps)
        self.empirical_kl, self.reparam_dim, self.fixed_var = empirical_kl, reparam_dim, fixed_var

    def out_size(self, size):
        if self.fixed_var:   # prior/gaussian.py:84: no log-variance half
            return torch.Size(size)
        out = list(size)
        out[self.reparam_dim - 1 if self.reparam_dim > 0 else self.reparam_dim] //= 2
        return torch.Size(out)

    def reparametrization(self, z: Tensor, temperature: Optional[Tensor] = None):
        if self.fixed_var:   # prior/gaussian.py:74-77: unit scale, or the temperature (+ 1e-8) broadcast over the sample
            def scale():
                s = torch.ones_like(z)
                if temperature is not None:
                    s = s * temperature.reshape(-1, *([1] * (z.dim() - 1))) + 1e-8
                return s
            return _LazyNormal(lambda: z, scale)
        mu, log_var = torch.chunk(z, 2, self.reparam_dim)
        return _LazyNormal(lambda: mu, lambda: (log_var / 2).exp())

    def _encode(self, x: Tensor, coeff: float, eps: Optional[Tensor] = None, time: Optional[Tensor] = None):
        shape = list(x.shape)
        if not self.fixed_var:
            shape[self.reparam_dim] //= 2
        if eps is None:
            eps = torch.randn(shape, device=x.device, dtype=x.dtype)
        if self.fixed_var or self.empirical_kl or self.reparam_dim not in (1, 1 - x.dim()):
            z, loss = HF.gaussian_prior_ex(x, eps, coeff, self.empirical_kl, self.fixed_var, time, self.reparam_dim)
        else:
            z, loss = HF.gaussian_prior(x, eps, coeff)
        zd = z.detach()  # the lazily built distributions must not keep this step's autograd graph alive
        artifacts = {"prior": _LazyNormal(lambda: torch.zeros_like(zd), lambda: torch.ones_like(zd)),
                     "distribution": self.reparametrization(x.detach(), temperature=time)}
        return z, loss, artifacts

    def encode(self, x: Tensor, time: Optional[Tensor] = None, eps: Optional[Tensor] = None) -> Prior.EncodingResults:
        if time is not None and not self.fixed_var:
            raise NotImplementedError("temperature (`time`) is only meaningful with fixed_var=True")
        return self._encode(x, 1.0, eps, time)

    def sample(self, shape, device) -> Tensor:
        return torch.randn(*shape, device=device)

    def forward(self, x: Tensor, step: int, time: Optional[Tensor] = None, eps: Optional[Tensor] = None):
        # loss_coeff * annealing is folded into the kernel (one multiply per sample instead of a separate launch)
        if time is not None and not self.fixed_var:
            raise NotImplementedError("temperature (`time`) is only meaningful with fixed_var=True")
        return self._encode(x, float(self.loss_coeff * self.annealing(step)), eps, time)
