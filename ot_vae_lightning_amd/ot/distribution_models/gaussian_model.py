"""``GaussianModel``: multivariate Gaussian fitted from streaming fp64 sufficient statistics
(n, sum x, sum x x^T), reference ot/distribution_models/gaussian_model.py.  The statistics kernel accumulates the
batch straight into the running buffers (or hands raw batch statistics to the all-reduce when running
data-parallel); ``fit`` and the ``cov`` parametrisations (symmetrise, make strictly positive definite) run on the
fp64 device kernels.  ``update_with_autograd=True`` (reference gaussian_model.py:52-55,76-93: mean and the Cholesky factor of the
covariance as trainable nn.Parameters, fitted through the negative log-likelihood) evaluates the log-density and its gradient
on the kernels of csrc/mvn.hip + the library GEMM."""
from typing import Optional

import torch
import torch.distributions as D
import torch.nn as nn
import torch.nn.utils.parametrize as P
from torch import Tensor

from ... import _lib
from ..._lib import check, ptr, stream
from ..matrix_utils import eigh_vectors, eye_like, make_psd, mm, psd_shift
from ..w2_utils import W2Mixin
from .base import DistributionModel

__all__ = ["GaussianModel"]


class GaussianModel(DistributionModel, W2Mixin):
    Distribution = D.Distribution

    def __init__(self, *size: int, w2_cfg={}, **kwargs):
        DistributionModel.__init__(self, *size, **kwargs)
        W2Mixin.__init__(self, **dict(w2_cfg))
        self.batch_dim = -2
        self.register_buffer("cov_init", torch.ones_like(self.vec_init) if self.diag else
                             eye_like(self.mat_init).clone())
        self.mean = nn.Parameter(self.vec_init.clone(), requires_grad=self.update_with_autograd)
        self.cov = nn.Parameter(self.cov_init.clone(), requires_grad=self.update_with_autograd)
        if self.update_with_autograd:
            # `cov` holds the Cholesky factor (diag: the variances) through the exp + tril re-parametrisation, which keeps it a
            # valid scale while an optimizer moves the raw parameter (reference gaussian_model.py:52-55,186-201)
            P.register_parametrization(self, "cov", ExpScaleTril(diag=self.diag))
            return
        self.register_buffer("_running_sum", torch.zeros_like(self.mean.data))
        self.register_buffer("_running_sum_cov", torch.zeros_like(self.cov.data))
        self.register_buffer("_n_obs", torch.zeros(self.vec_shape[:-1], dtype=self.vec_init.dtype))
        P.register_parametrization(self, "cov", Symmetric(diag=self.diag))
        P.register_parametrization(self, "cov", MakePositiveDefinite(diag=self.diag, strict=True))

    # -- state -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset(self) -> None:
        self.mean.copy_(self.vec_init)
        self.cov = self.cov_init
        if self.update_with_autograd:
            return
        self._running_sum.zero_()
        self._running_sum_cov.zero_()
        self._n_obs.zero_()

    @property
    def distribution(self):
        if self.update_with_autograd:  # cov is the scale (gaussian_model.py:76-83)
            return self.instantiate_normal(self.mean, scale=self.cov ** 0.5, scale_tril=self.cov)
        return self.instantiate_normal(self.mean, scale=self.cov ** 0.5, covariance_matrix=self.cov)

    @property
    def variances(self) -> Tensor:
        if self.update_with_autograd:  # get_var_normal(distribution): sigma^2, resp. L L^T
            return self.cov if self.diag else mm(self.cov, self.cov.transpose(-1, -2).contiguous())
        return self.cov

    # -- statistics --------------------------------------------------------------------------------------------
    def _batch_stats(self, samples: Tensor, accumulate: bool):
        """Launches the statistics kernel.  accumulate=True: EMA/add straight into the running buffers (single
        process); False: returns the raw (n, sum, sum_cov) of this batch."""
        lib = _lib.load()
        _lib.require_cuda(samples, "samples")
        d = self.dim
        lead = self.leading_shape
        bsz = samples.shape[-2]
        x = samples.detach()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.float()
        x = x.expand(*lead, bsz, d).reshape(-1, bsz, d).contiguous()
        nb = x.shape[0]
        if self._running_sum.dtype != torch.float64:
            raise TypeError("GaussianModel on the MI355X path keeps its statistics in float64 (pass dtype=torch.double)")
        ws = torch.empty(max(8, lib.otvae_gauss_stats_ws(nb, bsz, d, int(self.diag))), device=x.device, dtype=torch.uint8)
        if accumulate:
            n, sx, sxx = self._n_obs, self._running_sum, self._running_sum_cov
        else:
            n = torch.empty(nb, device=x.device, dtype=torch.float64)
            sx = torch.empty((nb, d), device=x.device, dtype=torch.float64)
            sxx = torch.empty((nb, d) if self.diag else (nb, d, d), device=x.device, dtype=torch.float64)
        decay = -1.0 if self.decay is None else float(self.decay)
        check(lib.otvae_gauss_stats(0 if x.dtype == torch.float32 else 1, ptr(x), nb, bsz, d, int(self.diag),
                                    int(accumulate), decay, ptr(ws), ptr(n), ptr(sx), ptr(sxx), stream()),
              "otvae_gauss_stats")
        if accumulate:
            return None
        return n.reshape(lead), sx.reshape(*lead, d), sxx.reshape(*lead, *( (d,) if self.diag else (d, d)))

    def _will_reduce(self) -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    @torch.no_grad()
    def update(self, samples: Tensor) -> None:
        self._validate_samples(samples)
        if self.update_with_autograd:
            raise RuntimeError("`update_with_autograd` is True: the parameters are trained with autograd; the running statistics "
                               "`update` feeds were not created (reference base.py:85-88 warns, then fails on the missing buffers)")
        if self.reduce_on_update and self._will_reduce():
            n, sx, sxx = self._batch_stats(samples, accumulate=False)
            self._accumulate(self.reduce(n), self.reduce(sx), self.reduce(sxx))
        else:
            self._batch_stats(samples, accumulate=True)

    @torch.no_grad()
    def _accumulate(self, n: Tensor, sx: Tensor, sxx: Tensor) -> None:
        """running <- ema(running, batch) IN PLACE (reference gaussian_model.py:106-108 rebinds the attributes; here the
        buffers keep their addresses: a captured training step has them baked into its kernel arguments)."""
        self._n_obs.copy_(self.ema_update(self._n_obs, n.reshape(self._n_obs.shape)))
        self._running_sum.copy_(self.ema_update(self._running_sum, sx))
        self._running_sum_cov.copy_(self.ema_update(self._running_sum_cov, sxx))

    @torch.no_grad()
    def _reduce_running(self) -> None:
        """all-reduce(SUM) of the running statistics over the ranks, written back IN PLACE (reference
        gaussian_model.py:123 with ``_stats(None, reduce=True)``): same values, stable addresses."""
        for buf in (self._n_obs, self._running_sum, self._running_sum_cov):
            red = self.reduce(buf)
            if red is not buf:
                buf.copy_(red)

    @torch.no_grad()
    def fit(self, samples: Optional[Tensor] = None) -> None:
        if self.update_with_autograd:
            # gaussian_model.py:111-116: overrides the trained parameters with the moments of `samples`, assigned through the
            # parametrisation's right inverse exactly as the reference does; nothing to do without samples
            if samples is None:
                return
            n, sx, sxx = self._batch_stats(samples, accumulate=False)
            mean, cov = self.mean_cov(self.reduce(sx), self.reduce(sxx), self.reduce(n))
            self.mean.copy_(mean.type_as(self.mean))
            self.cov = cov.type_as(self.mean)
            return
        if samples is not None:
            self.update(samples)
        self._reduce_running()
        n = self._n_obs
        if bool((n == 0).all()):
            return
        seen = n > 1e-8
        if bool(seen.all()):
            mean, cov = self.mean_cov(self._running_sum, self._running_sum_cov, n)
            self.mean.copy_(mean.type_as(self.mean))
            self.cov = cov.type_as(self.mean)
        else:
            mean, cov = self.mean_cov(self._running_sum[seen], self._running_sum_cov[seen], n[seen])
            self.mean.data[seen] = mean.type_as(self.mean)
            tmp = self.cov
            tmp[seen] = cov.type_as(tmp)
            self.cov = tmp

    def predict(self, samples: Tensor) -> Tensor:
        self._validate_samples(samples)
        if self.update_with_autograd:
            return mvn_log_prob(samples.type_as(self.mean), self.mean, self.cov ** 0.5 if self.diag else self.cov, self.diag)
        dist = self.instantiate_normal(self.mean.unsqueeze(-2), scale=self.cov.unsqueeze(-2) ** 0.5,
                                       covariance_matrix=self.cov.unsqueeze(-3) if not self.diag else None)
        return dist.log_prob(samples.type_as(self.mean))

    def cov_spectrum(self):
        """(cov, eigvals, Vt) of the full-matrix model with ONE eigendecomposition: ``cov`` is what the attribute of that
        name returns (the stored matrix mirrored from its upper triangle, then shifted to be strictly positive definite:
        the two parametrisations registered in ``__init__``), and (eigvals, Vt) is its spectrum, which the attribute
        computes too -- to find the shift -- and throws away."""
        if self.diag:
            raise ValueError("cov_spectrum is for full covariance matrices")
        raw = self.parametrizations.cov.original
        sym = (raw.triu() + raw.triu(1).transpose(-1, -2)).double()
        lam, vt = eigh_vectors(sym)
        shift = psd_shift(lam, strict=True, only_if_needed=False)
        return sym + shift[..., None, None] * eye_like(sym), lam + shift[..., None], vt

    @staticmethod
    def cov_spectra(*models):
        """``cov_spectrum`` of several models of one shape from ONE batched eigendecomposition (the matrices are independent: the
        solver runs a workgroup per matrix side by side)."""
        if any(m.diag for m in models):
            raise ValueError("cov_spectra is for full covariance matrices")
        syms = []
        for m in models:
            raw = m.parametrizations.cov.original
            syms.append((raw.triu() + raw.triu(1).transpose(-1, -2)).double())
        lam, vt = eigh_vectors(torch.stack(syms))
        out = []
        for i, sym in enumerate(syms):
            shift = psd_shift(lam[i], strict=True, only_if_needed=False)
            out.append((sym + shift[..., None, None] * eye_like(sym), lam[i] + shift[..., None], vt[i]))
        return out

    def w2(self, other) -> Tensor:
        return self.w2_gaussian(self.mean, other.mean, self.variances, self.get_var_normal(other))

    def extra_repr(self) -> str:
        return super().extra_repr() + W2Mixin.__repr__(self)


class _MvnLogProbFn(torch.autograd.Function):
    """log N(x; mean, L L^T) (diag: N(mean, diag(scale^2))) on csrc/mvn.hip: x [nb, B, D], mean [nb, D], scale [nb, D, D] | [nb, D]"""

    @staticmethod
    def forward(ctx, x, mean, scale, diag):
        lib = _lib.load()
        nb, b, d = x.shape
        y, lp = torch.empty_like(x), torch.empty((nb, b), device=x.device, dtype=torch.float64)
        check(lib.otvae_mvn_logprob_fwd(ptr(x), ptr(mean), ptr(scale), nb, b, d, int(diag), ptr(y), ptr(lp), stream()),
              "otvae_mvn_logprob_fwd")
        ctx.save_for_backward(y, scale)
        ctx.diag = diag
        return lp

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y, scale = ctx.saved_tensors
        nb, b, d = y.shape
        g = g.contiguous()
        qg = torch.empty_like(y)
        check(lib.otvae_mvn_logprob_bwd(ptr(g), ptr(y), ptr(scale), nb, b, d, int(ctx.diag), ptr(qg), stream()),
              "otvae_mvn_logprob_bwd")
        gsum = g.sum(-1)
        if ctx.diag:      # d sigma = sum_b qg_b * y_b - (sum_b g_b) / sigma
            gscale = (qg * y).sum(-2) - gsum[:, None] / scale
        else:             # d L = tril(sum_b qg_b y_b^T) - (sum_b g_b) diag(1 / L_ii)
            gscale = mm(qg.transpose(-1, -2).contiguous(), y).tril() - torch.diag_embed(gsum[:, None] / scale.diagonal(dim1=-1, dim2=-2))
        return -qg, qg.sum(-2), gscale, None


def mvn_log_prob(samples: Tensor, mean: Tensor, scale: Tensor, diag: bool) -> Tensor:
    """log-density of samples [*, B, D] under the Gaussians (mean [*, D], Cholesky factor [*, D, D] or standard deviations
    [*, D]); differentiable in all three.  fp64."""
    _lib.require_cuda(samples, "samples")
    lead, b, d = mean.shape[:-1], samples.shape[-2], samples.shape[-1]
    x = samples.double().expand(*lead, b, d).reshape(-1, b, d).contiguous()
    m = mean.double().reshape(-1, d).contiguous()
    sc = scale.double().reshape(-1, *( (d,) if diag else (d, d))).contiguous()
    return _MvnLogProbFn.apply(x, m, sc, diag).reshape(*lead, b).type_as(mean)


class ExpScaleTril(nn.Module):
    """cov parametrisation of the autograd-trained model: strictly lower triangle + exp(diagonal) (diag: exp), so that the
    value is always a valid Cholesky factor / variance vector (reference gaussian_model.py:186-201)."""

    def __init__(self, diag):
        super().__init__()
        self.diag = diag

    def forward(self, x: Tensor) -> Tensor:
        if self.diag:
            return x.exp()
        return x.tril(-1) + torch.diag_embed(x.diagonal(dim1=-1, dim2=-2).exp())

    def right_inverse(self, x: Tensor) -> Tensor:
        return x if self.diag else x.tril()


class MakePositiveDefinite(nn.Module):
    """cov parametrisation: always add |min(lambda_min,0)| + 1e-8 to the diagonal (reference gaussian_model.py:204-214)."""

    def __init__(self, diag, strict):
        super().__init__()
        self.diag, self.strict = diag, strict

    def forward(self, x):
        if not x.is_cuda:  # parametrize evaluates forward at registration time, before the model is moved
            return x
        return make_psd(x, strict=self.strict, return_correction=False, diag=self.diag)

    def right_inverse(self, x):
        return x


class Symmetric(nn.Module):
    """cov parametrisation: mirror the upper triangle (reference gaussian_model.py:217-226)."""

    def __init__(self, diag):
        super().__init__()
        self.diag = diag

    def forward(self, X):
        return X if self.diag else X.triu() + X.triu(1).transpose(-1, -2)

    def right_inverse(self, X):
        return X
