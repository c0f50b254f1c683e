"""``GaussianMixtureModel``, diagonal and full covariances (reference ot/distribution_models/gassian_mixture_model.py:28-177,
mixture behaviour from base.py:165-262, fitting loop from codebook_model.py:121-143): K Gaussians per leading index
fitted by (EMA) k-means style updates on streaming batches -- every sample is assigned to components by its
log-likelihood + log-weight, the assignment weights accumulate (count, sum x, sum x^2) per component, and mean /
variance / mixture weight of the OBSERVED components follow from the Laplace-smoothed counts.

MI355X path: the O(B K d) energies come from ``otvae_gmm_diag_energy`` (diagonal) or one batched eigendecomposition of the
component covariances + fp64 products (full); the assignment soft-max (``otvae_softmax_rows``) and the weighted sums
``weights^T @ x`` (``matrix_utils.mm``: ``otvae_gemm_f32 / _f64``) run on the library's own kernels in the reference's order;
``w2`` composes ``batch_ot_gmm`` (HIP pairwise Gaussian W2 costs + Sinkhorn kernels)."""
import math
from functools import partial
from typing import Optional, Tuple

import torch
import torch.distributions as D
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.utils.parametrize as P
from torch import Tensor

from ... import _lib, utils
from ..._lib import check, ptr, stream
from ..w2_utils import W2Mixin, batch_ot_gmm
from .base import MIXTURE_MODES, DistributionModel, gumbel_weights
from .gaussian_model import ExpScaleTril, MakePositiveDefinite, Symmetric, mvn_log_prob
from ..matrix_utils import eigh_vectors, eye_like, lse_rows, matmul64, mm, softmax_rows

__all__ = ["GaussianMixtureModel"]



class NormSum(nn.Module):
    """weights parametrisation: read back normalised to sum ``val`` (gassian_mixture_model.py:180-189)"""

    def __init__(self, val: float = 1.):
        super().__init__()
        self.val = val

    def forward(self, X):
        return self.val * X / X.sum(-1, keepdim=True)

    def right_inverse(self, X):
        return X


class GaussianMixtureModel(DistributionModel, W2Mixin):
    Distribution = D.MixtureSameFamily

    def __init__(self, *size: int, mixture_cfg={}, w2_cfg={}, **kwargs) -> None:
        cfg = dict(n_components=None, metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax",
                   inference_mode="argmax", kmeans_iter=100, laplace_eps=1e-5)
        unknown = set(mixture_cfg) - set(cfg)
        if unknown:
            raise TypeError(f"unexpected mixture_cfg keys: {sorted(unknown)}")
        cfg.update(mixture_cfg)
        if cfg["n_components"] is None:
            raise TypeError("mixture_cfg must give `n_components`")
        for m in (cfg["training_mode"], cfg["inference_mode"]):
            if m not in MIXTURE_MODES:
                raise NotImplementedError(f"assignment mode {m!r}: expected one of {MIXTURE_MODES}")
        self.n_components = int(cfg["n_components"])
        self.topk = cfg["topk"]     # (metric and p do not enter a Gaussian mixture's energies, gassian_mixture_model.py:91-99)
        self.temperature = float(cfg["temperature"])
        self.training_mode, self.inference_mode = cfg["training_mode"], cfg["inference_mode"]
        self.kmeans_iter = int(cfg["kmeans_iter"])
        self.laplace_smoothing = partial(utils.laplace_smoothing, n_categories=self.n_components, eps=cfg["laplace_eps"])
        DistributionModel.__init__(self, *size, **kwargs)
        W2Mixin.__init__(self, **dict(w2_cfg))
        self.batch_dim = -3
        self.register_buffer("cov_init", torch.ones_like(self.vec_init) if self.diag else eye_like(self.mat_init).clone())
        w = torch.ones(*self.leading_shape, self.n_components)
        self.register_buffer("weight_init", (w / w.sum(-1, keepdim=True)).type_as(self.vec_init))
        auto = self.update_with_autograd
        self.mean = nn.Parameter(self.vec_init.clone(), requires_grad=auto)
        self.cov = nn.Parameter(self.cov_init.clone(), requires_grad=auto)
        self._weights = nn.Parameter(self.weight_init.clone(), requires_grad=auto)
        if auto:
            # trained through the mixture log-likelihood (`forward` = `predict`): `cov` holds each component's Cholesky factor
            # (diag: its variances) behind the exp + tril re-parametrisation, the weights sit behind a soft-max; the k-means
            # buffers are not created (reference gaussian_model.py:52-72, gassian_mixture_model.py:53-58)
            P.register_parametrization(self, "cov", ExpScaleTril(diag=self.diag))
            P.register_parametrization(self, "_weights", nn.Softmax(-1))
            return
        self.register_buffer("_running_sum", torch.zeros_like(self.mean.data))
        self.register_buffer("_running_sum_cov", torch.zeros_like(self.cov.data))
        self.register_buffer("_n_obs", torch.zeros(self.vec_shape[:-1], dtype=self.vec_init.dtype))
        if not self.diag:  # as GaussianModel: mirror the upper triangle, then shift to strictly positive definite
            P.register_parametrization(self, "cov", Symmetric(diag=False))
        P.register_parametrization(self, "cov", MakePositiveDefinite(diag=self.diag, strict=True))
        P.register_parametrization(self, "_weights", NormSum(1.))

    # ---- shapes / distributions
    @property
    def vec_shape(self):
        return (*self.leading_shape, self.n_components, self.dim)

    @property
    def weights(self) -> Tensor:
        return self._weights

    @property
    def variances(self) -> Tensor:
        if self.update_with_autograd and not self.diag:  # get_var_normal(components): L L^T
            return mm(self.cov, self.cov.transpose(-1, -2).contiguous())
        return self.cov

    @property
    def batched_variances(self) -> Tensor:
        return self.variances.unsqueeze(-3 if self.diag else -4)

    @property
    def mode(self) -> str:
        return self.training_mode if self.training else self.inference_mode

    def _components(self, batched: bool):
        cov = self.cov
        if self.update_with_autograd and not self.diag:
            mean, tril = (self.mean.unsqueeze(-3), cov.unsqueeze(-4)) if batched else (self.mean, cov)
            return D.MultivariateNormal(mean, scale_tril=tril)
        if self.diag:
            mean, cov = (self.mean.unsqueeze(-3), cov.unsqueeze(-3)) if batched else (self.mean, cov)
            return D.Independent(D.Normal(mean, cov ** 0.5), 1)
        mean, cov = (self.mean.unsqueeze(-3), cov.unsqueeze(-4)) if batched else (self.mean, cov)
        return D.MultivariateNormal(mean, covariance_matrix=cov)

    @property
    def distribution(self) -> D.MixtureSameFamily:
        return D.MixtureSameFamily(D.Categorical(self.weights), self._components(False))

    @property
    def batched_distribution(self) -> D.MixtureSameFamily:
        return D.MixtureSameFamily(D.Categorical(self.weights.unsqueeze(-2)), self._components(True))

    @torch.no_grad()
    def reset(self) -> None:
        self.mean.copy_(self.vec_init)
        self.cov = self.cov_init
        self._weights = self.weight_init
        if self.update_with_autograd:
            return
        self._running_sum.zero_()
        self._running_sum_cov.zero_()
        self._n_obs.zero_()

    # ---- assignment
    def energy(self, samples: Tensor) -> Tensor:
        """log N(x; mean_k, var_k) + log w_k, [*, B, K] (gassian_mixture_model.py:91-99)"""
        self._validate_samples(samples)
        lib = _lib.load()
        _lib.require_cuda(samples, "samples")
        dt = self.mean.dtype
        if dt not in (torch.float32, torch.float64):
            raise TypeError("GaussianMixtureModel parameters must be float32 or float64")
        lead = torch.broadcast_shapes(samples.shape[:-2], self.leading_shape)
        bsz, K, d = samples.shape[-2], self.n_components, self.dim
        x3 = samples.to(dt).expand(*lead, bsz, d).reshape(-1, bsz, d).contiguous()
        nb = x3.shape[0]
        flat = lambda t, tail: t.detach().to(dt).expand(*lead, *tail).reshape(nb, *tail).contiguous()  # noqa: E731
        logw = flat(torch.log_softmax(torch.log(self.weights), dim=-1), (K,))
        if not self.diag:
            return self._energy_full(x3, flat(self.mean, (K, d)), flat(self.variances, (K, d, d)), logw).reshape(*lead, bsz, K) \
                .type_as(samples if samples.is_floating_point() else x3)
        mean, var = flat(self.mean, (K, d)), flat(self.cov, (K, d))
        out = torch.empty((nb, bsz, K), device=x3.device, dtype=dt)
        check(lib.otvae_gmm_diag_energy(0 if dt == torch.float32 else 1, ptr(x3), ptr(mean), ptr(var), ptr(logw), nb, bsz, K, d,
                                        ptr(out), stream()), "otvae_gmm_diag_energy")
        return out.reshape(*lead, bsz, K).type_as(samples if samples.is_floating_point() else out)

    def _energy_full(self, x3: Tensor, mean: Tensor, cov: Tensor, logw: Tensor) -> Tensor:
        """log N(x; mean_k, C_k) + log w_k for full covariances, [nb, B, K]: from ONE batched eigendecomposition of the nb*K
        covariances (C = V L V^T): Mahalanobis = |L^-1/2 V^T (x - mean)|^2 (one fp64 product per component), log det = sum log L.
        The reference goes through MultivariateNormal's Cholesky factor (gassian_mixture_model.py:91-99); same value."""
        nb, bsz, d = x3.shape
        K = mean.shape[1]
        lam, vt = eigh_vectors(cov.double().reshape(nb * K, d, d))                  # vt[k] rows = eigenvectors
        whiten = lam.rsqrt().unsqueeze(-1) * vt                                     # L^-1/2 V^T
        centred = (x3.double().unsqueeze(1) - mean.double().unsqueeze(2)).reshape(nb * K, bsz, d)   # [nb*K, B, d]
        y = matmul64(centred, whiten, trans_b=True)                                 # [nb*K, B, d]
        maha = y.square().sum(-1).reshape(nb, K, bsz).transpose(1, 2)               # [nb, B, K]
        logdet = lam.log().sum(-1).reshape(nb, 1, K)
        return (-0.5 * (maha + logdet + d * math.log(2 * math.pi)) + logw.double().unsqueeze(1)).to(x3.dtype)

    def assign(self, samples: Tensor):
        """(assignment weights [*, B, K], sampled indices [*, B], Categorical(softmax weights)) -- base.py:206-239"""
        from .codebook_model import topk_energy
        energy = topk_energy(self.energy(samples), self.topk)     # base.py:217-220
        weights = softmax_rows(energy, 1.0 / self.temperature)
        distribution = D.Categorical(weights)
        indices = distribution.sample()
        mode = self.mode
        if mode == "mean" or self.topk == 1:
            pass
        elif mode == "sample":
            weights = F.one_hot(indices, self.n_components).type_as(weights)
        elif mode == "argmax":
            weights = F.one_hot(weights.argmax(-1), self.n_components).type_as(weights)
        elif "gumbel" in mode:
            noise, self.gumbel_noise = getattr(self, "gumbel_noise", None), None   # injected draws are used once
            weights = gumbel_weights(energy, self.temperature, "hard" in mode, noise)
        return weights, indices, distribution

    def predict_mean_var(self, assignments: Tensor) -> Tuple[Tensor, Tensor]:
        """per-sample mean and variance of the assigned component(s): assignments [*, B, K] -> [*, B, d] each"""
        mean = mm(assignments.type_as(self.mean), self.mean)
        cov = self.cov
        if self.diag:
            var = mm(assignments.type_as(cov), cov)
        else:  # [*, B, K] x [*, K, d*d] -> [*, B, d, d]
            var = mm(assignments.type_as(cov), cov.flatten(-2)).unflatten(-1, (self.dim, self.dim))
        return mean.type_as(assignments), var.type_as(assignments)

    def predict(self, samples: Tensor) -> Tensor:
        """log-density of the mixture at the samples, [*, B] -- what `model(samples)` returns.  (The reference class inherits
        from GaussianModel first: its `predict` is GaussianModel.predict on the MixtureSameFamily `batched_distribution`,
        gaussian_model.py:129-132, not CodebookModel's (encodings, indices, distribution) triple.)  With
        ``update_with_autograd`` this is the training objective: differentiable in mean, Cholesky factor / variances and weights
        through ``mvn_log_prob`` (csrc/mvn.hip)."""
        self._validate_samples(samples)
        if not self.update_with_autograd:
            return lse_rows(self.energy(samples))
        x = samples.type_as(self.mean).unsqueeze(-3)                                  # [*, 1, B, d] against K components
        scale = self.cov ** 0.5 if self.diag else self.cov
        comp = mvn_log_prob(x, self.mean, scale, self.diag).transpose(-1, -2)         # [*, K, B] -> [*, B, K]
        logw = torch.log_softmax(torch.log(self.weights), dim=-1).unsqueeze(-2)
        return lse_rows(comp + logw)

    def encode(self, samples: Tensor):
        """(expected component mean per sample, sampled indices, assignment distribution): the triple CodebookModel.predict returns
        for a codebook (round 1-2 exposed it under the name `predict`, which the reference class resolves differently)"""
        weights, indices, distribution = self.assign(samples)
        return mm(weights.type_as(self.mean), self.mean), indices, distribution

    # ---- fitting (codebook_model.py:121-143 driving gassian_mixture_model.py:108-170)
    def kmean_iteration(self, samples: Optional[Tensor]):
        if samples is None:
            return self._n_obs, self._running_sum, self._running_sum_cov
        weights, _, _ = self.assign(samples)                               # [*, B, K]
        wt = weights.transpose(-1, -2).type_as(samples)
        if self.diag:
            return weights.sum(-2).type_as(samples), mm(wt, samples), mm(wt, samples ** 2)
        outer = (samples.unsqueeze(-1) * samples.unsqueeze(-2)).flatten(-2)   # [*, B, d*d]: the outer products x x^T
        return weights.sum(-2).type_as(samples), mm(wt, samples), mm(wt, outer).unflatten(-1, (self.dim, self.dim))

    def _init_parameters(self, samples: Tensor) -> None:
        if torch.allclose(self.mean, self.vec_init):
            rand_indices = torch.randperm(samples.size(-2))[:self.n_components]  # host generator, as the reference
            self.mean.copy_(samples[..., rand_indices.to(samples.device), :].type_as(self.mean))
            self._n_obs += 1

    def _update_buffers(self, weights_sum: Tensor, samples_sum: Tensor, samples_cov_sum: Tensor, decay: bool = False):
        hit = weights_sum > 1e-8
        if decay:
            self._n_obs[hit] = self.ema_update(self._n_obs[hit], weights_sum[hit])
            self._running_sum[hit] = self.ema_update(self._running_sum[hit], samples_sum[hit])
            self._running_sum_cov[hit] = self.ema_update(self._running_sum_cov[hit], samples_cov_sum[hit])
        else:
            self._n_obs[hit] = weights_sum[hit]
            self._running_sum[hit] = samples_sum[hit]
            self._running_sum_cov[hit] = samples_cov_sum[hit]
        return self._n_obs, self._running_sum, self._running_sum_cov

    def _update_parameters(self, weights_sum: Tensor, samples_sum: Tensor, samples_cov_sum: Tensor) -> None:
        n_obs = self.laplace_smoothing(weights_sum)
        if bool((n_obs == 0).all()):
            return
        seen = n_obs > 1e-8
        mean, cov = self.mean_cov(samples_sum[seen], samples_cov_sum[seen], n_obs[seen])
        self.mean.data[seen] = mean.type_as(self.mean)
        tmp = self.cov
        tmp[seen] = cov.type_as(tmp)
        self.cov = tmp
        tmp = self._weights                        # normalised read-back ...
        tmp[seen] = weights_sum[seen].type_as(tmp)  # ... with the raw counts of the observed components, as the reference
        self._weights = tmp

    def _no_buffers(self, what: str):
        raise RuntimeError(f"`update_with_autograd` is True: the parameters are trained with autograd; the k-means buffers `{what}` "
                           "feeds were not created (the reference warns, base.py:82-90, then fails on the missing buffers)")

    @torch.no_grad()
    def update(self, samples: Tensor) -> None:
        self._validate_samples(samples)
        if self.update_with_autograd:
            self._no_buffers("update")
        samples = samples.detach().type_as(self._running_sum)
        self._init_parameters(samples)
        res = self.kmean_iteration(samples)
        if self.reduce_on_update:
            res = [self.reduce(r) for r in res]
        self._update_parameters(*self._update_buffers(*res, decay=True))

    @torch.no_grad()
    def fit(self, samples: Optional[Tensor] = None) -> None:
        if self.update_with_autograd:
            self._no_buffers("fit")
        if samples is not None:
            self._validate_samples(samples)
            samples = samples.detach().type_as(self._running_sum)
            self._init_parameters(samples)
        res = None
        for _ in range(self.kmeans_iter):
            res = self.kmean_iteration(samples)
            res_r = [self.reduce(r) for r in res]
            self._update_parameters(*res_r)
            # From the stored buffers every iteration recomputes the same values for the observed components; only the
            # unobserved ones drift (their variance is read back + 1e-8, their weight re-normalised), so the loop runs on
            # for them alone -- as the reference's does.
            if samples is None and bool((self.laplace_smoothing(res_r[0]) > 1e-8).all()):
                break
        if self.kmeans_iter > 0:
            self._update_buffers(*res, decay=False)

    def w2(self, other: D.MixtureSameFamily) -> Tensor:
        total, _ = batch_ot_gmm(self.mean, other.component_distribution.mean, self.variances,
                                self.get_var_normal(other.component_distribution), diag=self.diag, weight_source=self.weights,
                                weight_target=other.mixture_distribution.probs, dtype=self.dtype, max_iter=100)
        return total

    def extra_repr(self) -> str:
        return (DistributionModel.extra_repr(self) + ", " + W2Mixin.__repr__(self) + f", num_components={self.n_components}, "
                f"temperature={self.temperature}, training_mode={self.training_mode}, inference_mode={self.inference_mode}")
