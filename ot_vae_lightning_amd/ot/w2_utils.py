"""Wasserstein-2 utilities with the reference's names (ot/w2_utils.py) on the MI355X kernels:
``sinkhorn_log`` (:276-319), ``w2_gaussian`` (:40-80), ``compute_transport_operators`` (:391-458, eq. 17),
``apply_transport`` (:464-527), ``batch_w2_dissimilarity_gaussian_diag`` (:86-134), ``W2Mixin`` (:533-600).
Out of scope for this path: GMM transport (``batch_ot_gmm``), barycenters, the stochastic
(eq. 19) operators -- they raise ``NotImplementedError``."""
import math
import warnings
from functools import partial
from typing import Optional, Tuple, Union

import torch
import torch.distributions as D
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream
from .matrix_utils import *  # noqa: F401,F403
from .matrix_utils import (STABILITY_CONST, cholesky, pinv_sym, spectral_fn, eigh_vectors, eigvals_and_fn, eye_like, is_symmetric, matmul64, mean_cov, psd_shift,
                           spectral_fn)

__all__ = ["w2_gaussian", "batch_w2_dissimilarity_gaussian_diag", "batch_w2_dissimilarity_gaussian", "gaussian_barycenter", "batch_ot_gmm", "sinkhorn_log", "sinkhorn_log_potentials",
           "sq_euclidean_cost", "ot_cost", "compute_transport_operators", "apply_transport", "W2Mixin"]

_DT = {torch.float32: 0, torch.float64: 1}


def _dt(t: Tensor) -> int:
    if t.dtype not in _DT:
        raise TypeError(f"expected float32 or float64, got {t.dtype}")
    return _DT[t.dtype]


# ------------------------------------------------------------------------------------------------ Sinkhorn
class SinkhornSolverStarved(RuntimeError):
    """the single-launch Sinkhorn solver gave up waiting for its other workgroups; its outputs are NaN"""


def raise_if_solver_starved(iters: Tensor) -> None:
    """``iters`` is the device int32 a solve left (-1: a bounded wait of the persistent kernel ran out and the outputs were
    poisoned with NaN).  Looked at only when that costs no capture: while a stream is capturing the host cannot read device
    memory, and the NaN loss is then the signal."""
    if torch.cuda.is_current_stream_capturing():
        return
    if int(iters.item()) < 0:
        raise SinkhornSolverStarved(
            "sinkhorn_log: the persistent solver's workgroups were not co-resident within its wait budget "
            "(OTVAE_SK_SPIN_LIMIT); outputs are NaN.  OTVAE_SK_MULTILAUNCH=1 selects the one-launch-per-half-iteration path.")


def sinkhorn_log_potentials(a: Tensor, b: Tensor, C: Tensor, reg: float = 1e-5, max_iter: int = 1000,
                            threshold: float = STABILITY_CONST, check_starved: bool = True):
    """Returns (pi, u, v, iters_done[device int32]).  See ``sinkhorn_log``."""
    lib = _lib.load()
    _lib.require_cuda(C, "C")
    if C.dim() < 2 or a.shape[-1] != C.shape[-2] or b.shape[-1] != C.shape[-1]:
        raise ValueError(f"sinkhorn_log: a {tuple(a.shape)}, b {tuple(b.shape)} do not match C {tuple(C.shape)}")
    lead = C.shape[:-2]
    n, m = C.shape[-2:]
    dt = _dt(C)
    a2 = a.to(C.dtype).expand(*lead, n).reshape(-1, n).contiguous()
    b2 = b.to(C.dtype).expand(*lead, m).reshape(-1, m).contiguous()
    c3 = C.reshape(-1, n, m).contiguous()
    nb = c3.shape[0]
    ws = torch.empty(lib.otvae_sinkhorn_ws(dt, nb, n, m), device=C.device, dtype=torch.uint8)
    pi = torch.empty_like(c3)
    u = torch.empty_like(a2)
    v = torch.empty_like(b2)
    iters = torch.zeros(1, device=C.device, dtype=torch.int32)
    check(lib.otvae_sinkhorn_log(dt, ptr(a2), ptr(b2), ptr(c3), nb, n, m, float(reg), int(max_iter), float(threshold),
                                 ptr(ws), ptr(pi), ptr(u), ptr(v), ptr(iters), stream()), "otvae_sinkhorn_log")
    if check_starved:
        raise_if_solver_starved(iters)
    return pi.reshape(*lead, n, m), u.reshape(*lead, n), v.reshape(*lead, m), iters


class _SinkhornLogFn(torch.autograd.Function):
    """``sinkhorn_log`` with the reference's autograd semantics: the reference function is plain torch arithmetic, so a gradient
    reaches a, b and C THROUGH every iteration (ot/w2_utils.py:301-319).  Forward = ``otvae_sinkhorn_log_tape`` (the same solve, one
    launch per half-iteration, every potential kept), backward = ``otvae_sinkhorn_log_bwd`` (the reverse sweep: 2 row-reduction
    passes per iteration + one pass that forms the cost gradient; csrc/sinkhorn_diff.hip)."""

    @staticmethod
    def forward(ctx, a2, b2, c3, reg, max_iter, threshold):
        lib = _lib.load()
        nb, n, m = c3.shape
        dt = _dt(c3)
        tape = torch.empty(lib.otvae_sinkhorn_tape_bytes(dt, nb, n, m, int(max_iter)), device=c3.device, dtype=torch.uint8)
        pi = torch.empty_like(c3)
        iters = torch.zeros(1, device=c3.device, dtype=torch.int32)
        check(lib.otvae_sinkhorn_log_tape(dt, ptr(a2), ptr(b2), ptr(c3), nb, n, m, float(reg), int(max_iter), float(threshold),
                                          ptr(tape), ptr(pi), None, None, ptr(iters), stream()), "otvae_sinkhorn_log_tape")
        ctx.save_for_backward(a2, b2, pi, tape)
        ctx.cfg = (dt, nb, n, m, float(reg), int(max_iter))
        return pi

    @staticmethod
    def backward(ctx, gpi):
        a2, b2, pi, tape = ctx.saved_tensors
        dt, nb, n, m, reg, max_iter = ctx.cfg
        lib = _lib.load()
        gpi = gpi.to(pi.dtype).contiguous()
        need_a, need_b, need_c = ctx.needs_input_grad[:3]
        ws = torch.empty(lib.otvae_sinkhorn_bwd_ws(dt, nb, n, m, max_iter), device=pi.device, dtype=torch.uint8)
        gc = torch.empty_like(pi) if need_c else None
        ga = torch.empty_like(a2) if need_a else None
        gb = torch.empty_like(b2) if need_b else None
        check(lib.otvae_sinkhorn_log_bwd(dt, ptr(gpi), ptr(pi), ptr(a2), ptr(b2), nb, n, m, reg, max_iter, ptr(tape), ptr(ws),
                                         ptr(gc), ptr(ga), ptr(gb), stream()), "otvae_sinkhorn_log_bwd")
        return ga, gb, gc, None, None, None


def sinkhorn_log(a: Tensor, b: Tensor, C: Tensor, reg: float = 1e-5, max_iter: int = 1000,
                 threshold: float = STABILITY_CONST) -> Tensor:
    """Entropic OT plan by log-domain Sinkhorn iterations: a [*, N], b [*, M], C [*, N, M] -> pi [*, N, M].
    Same arithmetic and stopping rule as the reference (stop every problem of the batch at the first iteration
    where the smallest per-problem L1 change of (u, v) is below ``threshold``); the test runs on the device, so
    unlike the reference there is no host synchronisation per iteration.

    Autograd: as in the reference (plain torch arithmetic, ot/w2_utils.py:301-319) the plan is differentiable with respect to
    ``a``, ``b`` and ``C`` through all iterations: when any of them requires a gradient the call takes the tape-recording solver
    and carries ``_SinkhornLogFn``'s backward; otherwise the single-launch solver runs and the result has no ``grad_fn``."""
    if torch.is_grad_enabled() and any(isinstance(t, Tensor) and t.requires_grad for t in (a, b, C)):
        _lib.require_cuda(C, "C")
        if C.dim() < 2 or a.shape[-1] != C.shape[-2] or b.shape[-1] != C.shape[-1]:
            raise ValueError(f"sinkhorn_log: a {tuple(a.shape)}, b {tuple(b.shape)} do not match C {tuple(C.shape)}")
        _dt(C)
        lead = C.shape[:-2]
        n, m = C.shape[-2:]
        a2 = a.to(C.dtype).expand(*lead, n).reshape(-1, n).contiguous()
        b2 = b.to(C.dtype).expand(*lead, m).reshape(-1, m).contiguous()
        pi = _SinkhornLogFn.apply(a2, b2, C.reshape(-1, n, m).contiguous(), reg, max_iter, threshold)
        return pi.reshape(*lead, n, m)
    return sinkhorn_log_potentials(a, b, C, reg, max_iter, threshold)[0]


class _SqEuclideanCostFn(torch.autograd.Function):
    """C_ij = |x_i - y_j|^2 with its gradient: gx_i = 2 sum_j G_ij (x_i - y_j), gy_j = 2 sum_i G_ij (y_j - x_i) -- the plan-times-samples
    kernel of the minibatch-OT prior (``otvae_ot_cost_grad``) with the upstream gradient in the plan's place."""

    @staticmethod
    def forward(ctx, x3, y3):
        lib = _lib.load()
        nb, n, d = x3.shape
        m = y3.shape[1]
        out = torch.empty((nb, n, m), device=x3.device, dtype=x3.dtype)
        check(lib.otvae_sqdist(_dt(x3), ptr(x3), ptr(y3), nb, n, m, d, ptr(out), stream()), "otvae_sqdist")
        ctx.save_for_backward(x3, y3)
        return out

    @staticmethod
    def backward(ctx, g):
        x3, y3 = ctx.saved_tensors
        lib = _lib.load()
        nb, n, d = x3.shape
        m = y3.shape[1]
        g = g.contiguous()
        one = torch.ones(1, device=g.device, dtype=g.dtype)
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x3)
            for k in range(nb):
                check(lib.otvae_ot_cost_grad(_dt(x3), ptr(x3[k]), ptr(y3[k]), ptr(g[k]), ptr(one), 1, 1.0, None, n, m, d, ptr(gx[k]),
                                             stream()), "otvae_ot_cost_grad")
        if ctx.needs_input_grad[1]:
            gy = torch.empty_like(y3)
            gt = g.transpose(1, 2).contiguous()
            for k in range(nb):
                check(lib.otvae_ot_cost_grad(_dt(x3), ptr(y3[k]), ptr(x3[k]), ptr(gt[k]), ptr(one), 1, 1.0, None, m, n, d, ptr(gy[k]),
                                             stream()), "otvae_ot_cost_grad")
        return gx, gy


def sq_euclidean_cost(x: Tensor, y: Tensor) -> Tensor:
    """C[*, i, j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j for x [*, N, D], y [*, M, D]; differentiable with respect to both."""
    lib = _lib.load()
    _lib.require_cuda(x, "x")
    lead, n, d = x.shape[:-2], x.shape[-2], x.shape[-1]
    m = y.shape[-2]
    x3, y3 = x.reshape(-1, n, d).contiguous(), y.to(x.dtype).reshape(-1, m, d).contiguous()
    if torch.is_grad_enabled() and (x3.requires_grad or y3.requires_grad):
        return _SqEuclideanCostFn.apply(x3, y3).reshape(*lead, n, m)
    out = torch.empty((x3.shape[0], n, m), device=x.device, dtype=x.dtype)
    check(lib.otvae_sqdist(_dt(x), ptr(x3), ptr(y3), x3.shape[0], n, m, d, ptr(out), stream()), "otvae_sqdist")
    return out.reshape(*lead, n, m)


def ot_cost(C: Tensor, pi: Tensor) -> Tensor:
    """sum_ij C_ij pi_ij over the last two dims (fp64 accumulation, fixed order)."""
    lib = _lib.load()
    lead, n, m = C.shape[:-2], C.shape[-2], C.shape[-1]
    c3, p3 = C.reshape(-1, n, m).contiguous(), pi.to(C.dtype).reshape(-1, n, m).contiguous()
    nb = c3.shape[0]
    ws = torch.empty(nb * 256, device=C.device, dtype=torch.float64)
    out = torch.empty(nb, device=C.device, dtype=C.dtype)
    check(lib.otvae_ot_cost(_dt(C), ptr(c3), ptr(p3), nb, n, m, ptr(ws), ptr(out), stream()), "otvae_ot_cost")
    return out.reshape(lead)


# ------------------------------------------------------------------------------------------------ Gaussian W2
def _check_vec_mat(mean_s, mean_t, cov_s, cov_t):
    for name, t in (("mean_source", mean_s), ("mean_target", mean_t), ("cov_source", cov_s), ("cov_target", cov_t)):
        if not isinstance(t, Tensor):
            raise ValueError(f"`{name}` is expected to be a torch.Tensor, got `{type(t)}` instead.")
    if mean_s.dim() < 1 or mean_t.dim() < 1:
        raise ValueError("means should be 1-dim vectors (+ optional leading batch dimensions)")
    if cov_s.dim() < 2 or cov_t.dim() < 2:
        raise ValueError("covariances should be 2-dim matrices (+ optional leading batch dimensions)")
    dims = {mean_s.size(-1), mean_t.size(-1), cov_s.size(-1), cov_s.size(-2), cov_t.size(-1), cov_t.size(-2)}
    if len(dims) != 1:
        raise ValueError(f"All the inputs dimensionalities should match, got {sorted(dims)}")


def _require_spd(cov: Tensor, name: str, make_pd: bool, strict: bool, verbose: bool) -> Tensor:
    """The 'spd'/'spsd' argument validation of the reference (w2_utils.py:661-679)."""
    if not bool(is_symmetric(cov).all()):
        raise ValueError(f"`{name}` should be symmetric.")
    if make_pd:
        return make_psd(cov, strict=strict, only_if_needed=True)  # noqa: F405
    ev, _ = eigvals_and_fn(cov, 0)
    lo = ev.min(-1)[0]
    if not bool(((lo > 0) if strict else (lo >= 0)).all()):
        raise ValueError(f"`{name}` should be symmetric and positive {'' if strict else 'semi '}definite. "
                         "Use `make_pd=True` to automatically add a small value to the matrix diagonals.")
    return cov


def _spd_and_roots(cov: Tensor, name: str, make_pd: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """the strict 'spd' validation of ``_require_spd`` plus cov^1/2 and (cov + 1e-8 I)^-1/2, all from ONE eigendecomposition
    ([nb, D, D] -> validated cov, sqrt, inverse sqrt): a diagonal shift moves the eigenvalues and leaves the vectors"""
    if not bool(is_symmetric(cov).all()):
        raise ValueError(f"`{name}` should be symmetric.")
    lam, vt = eigh_vectors(cov)
    if make_pd:
        shift = psd_shift(lam, strict=True, only_if_needed=True)
        cov, lam = cov + shift[..., None, None] * eye_like(cov), lam + shift[..., None]
    elif not bool((lam.min(-1)[0] > 0).all()):
        raise ValueError(f"`{name}` should be symmetric and positive definite. "
                         "Use `make_pd=True` to automatically add a small value to the matrix diagonals.")
    return cov, spectral_fn(lam.sqrt(), vt), spectral_fn((lam + STABILITY_CONST).rsqrt(), vt)


def w2_gaussian(mean_source: Tensor, mean_target: Tensor, cov_source: Tensor, cov_target: Tensor,
                make_pd: bool = False, verbose: bool = False, dtype=torch.double) -> Tensor:
    """Squared Gelbrich distance |ms-mt|^2 + tr(Cs + Ct - 2 (Ct^1/2 Cs Ct^1/2)^1/2), fp64, batched over leading dims."""
    lib = _lib.load()
    _check_vec_mat(mean_source, mean_target, cov_source, cov_target)
    lead = torch.broadcast_shapes(mean_source.shape[:-1], mean_target.shape[:-1], cov_source.shape[:-2],
                                  cov_target.shape[:-2])
    d = mean_source.shape[-1]
    ms = mean_source.double().expand(*lead, d).reshape(-1, d).contiguous()
    mt = mean_target.double().expand(*lead, d).reshape(-1, d).contiguous()
    cs = _require_spd(cov_source.double().expand(*lead, d, d).reshape(-1, d, d).contiguous(), "cov_source", make_pd,
                      True, verbose)
    ct, rt = _spd_and_roots(cov_target.double().expand(*lead, d, d).reshape(-1, d, d).contiguous(), "cov_target", make_pd)[:2]
    mix = matmul64(matmul64(rt, cs), rt)
    if not bool(is_symmetric(mix).all()):
        raise ValueError("`cov_target_sqrt @ cov_source @ cov_target_sqrt` should be symmetric.")
    sq = eigvals_and_fn(mix, 1)[1]
    nb = ms.shape[0]
    out = torch.empty(nb, device=ms.device, dtype=torch.float64)
    check(lib.otvae_w2_tail(ptr(ms), ptr(mt), ptr(cs.contiguous()), ptr(ct.contiguous()), ptr(sq), nb, d, ptr(out),
                            stream()), "otvae_w2_tail")
    return out.reshape(lead)


def batch_w2_dissimilarity_gaussian_diag(mean_source: Tensor, mean_target: Tensor, var_source: Tensor,
                                         var_target: Tensor, dtype=torch.double) -> Tensor:
    """D[*, i, j] = W2^2(N(ms_i, diag vs_i), N(mt_j, diag vt_j)) = |ms_i - mt_j|^2 + |sqrt vs_i - sqrt vt_j|^2."""
    if (var_source < 0).any() or (var_target < 0).any():
        raise ValueError("variances are expected to have positive entries.")
    return sq_euclidean_cost(mean_source.to(dtype), mean_target.to(dtype)) + \
        sq_euclidean_cost(var_source.to(dtype).sqrt(), var_target.to(dtype).sqrt())


def batch_w2_dissimilarity_gaussian(mean_source: Tensor, mean_target: Tensor, cov_source: Tensor, cov_target: Tensor,
                                    make_pd: bool = False, verbose: bool = False, dtype=torch.double) -> Tensor:
    """D[*, i, j] = W2^2(N(ms_i, Cs_i), N(mt_j, Ct_j)) for full covariances: means [*, N, D] / [*, M, D], covariances
    [*, N, D, D] / [*, M, D, D] (reference ot/w2_utils.py:138-189).  The N x M pairs go through ONE batched ``w2_gaussian``
    (the eigensolver runs a workgroup per matrix); as in the reference each side must be symmetric positive definite."""
    for cov, name in ((cov_source, "cov_source"), (cov_target, "cov_target")):
        _require_spd(cov.double().reshape(-1, cov.shape[-1], cov.shape[-1]).contiguous(), name, False, True, verbose)
    n, m = mean_source.size(-2), mean_target.size(-2)
    ones = [1] * (mean_source.dim() - 2)
    dis = w2_gaussian(mean_source.repeat_interleave(m, -2), mean_target.repeat(*ones, n, 1),
                      cov_source.repeat_interleave(m, -3), cov_target.repeat(*ones, n, 1, 1),
                      make_pd=make_pd, verbose=verbose, dtype=dtype)
    return dis.view(*mean_source.shape[:-2], n, m)


def gaussian_barycenter(mean: Tensor, cov: Tensor, weights: Tensor, diag: bool, n_iter: int = 100, dtype=torch.double,
                        init_index: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """W2 barycentre of the Gaussians N(mean_i, cov_i) with weights w_i (reference ot/w2_utils.py:325-385, the fixed point of
    Alvarez-Esteban et al.): mean [*, N, D], cov [*, N, D, D] ([*, N, D] if diag), weights [*, N] -> ([*, D], [*, D, D] | [*, D]).
    diag: mean_b = sum w mu, var_b = (sum w sigma)^2.  Full: S <- sum_i w_i (S^1/2 C_i S^1/2)^1/2 iterated ``n_iter`` times from
    one of the C_i -- the reference draws its index with torch.randint; ``init_index`` fixes it (parity tests)."""
    mean, cov, weights = mean.to(dtype), cov.to(dtype), weights.to(dtype)
    if mean.dim() < 2 or weights.dim() < 1 or mean.size(-2) != weights.size(-1) or cov.size(-2 if diag else -3) != mean.size(-2):
        raise ValueError("All the inputs component dimension should match")
    total = weights.sum(-1)
    if bool((weights < -1e-5).any()) or bool((total < 1 - 1e-5).any()) or bool((total > 1 + 1e-5).any()):
        raise ValueError("`weights` is expected to be a valid probability vector with positive entries that sum up to 1.")
    w_row = weights.unsqueeze(-2)                                            # [*, 1, N]
    mean_b = mm(w_row, mean).squeeze(-2)
    if diag:
        if bool((cov < 0).any()):
            raise ValueError("`cov` is expected to be a valid variance vector with positive entries.")
        return mean_b, (mm(w_row, torch.sqrt(cov)) ** 2).squeeze(-2)
    _require_spd(cov.reshape(-1, cov.shape[-1], cov.shape[-1]).contiguous(), "cov", False, True, False)
    n, d = cov.size(-3), cov.size(-1)
    lead = torch.broadcast_shapes(cov.shape[:-3], weights.shape[:-1])
    cflat = cov.expand(*lead, n, d, d).reshape(-1, d, d).contiguous()          # [L*N, D, D]
    w4 = weights.expand(*lead, n).reshape(-1, n, 1, 1)
    if init_index is None:
        init_index = int(torch.randint(size=(1,), high=n).item())
    cov_b = cflat.reshape(-1, n, d, d)[:, init_index].contiguous()            # [L, D, D]
    for _ in range(n_iter):
        root = sqrtm(cov_b).unsqueeze(1).expand(-1, n, d, d).reshape(-1, d, d)
        mix = matmul64(matmul64(root, cflat), root)
        cov_b = (w4 * sqrtm(mix).reshape(-1, n, d, d)).sum(1)
    return mean_b, cov_b.reshape(*lead, d, d)


def _check_mixture(mean: Tensor, var: Tensor, weight: Tensor, side: str, tol: float = 1e-5):
    """the argument checks of the reference's ``_validate_args`` for ('vec', 'var', 'prob') triples (w2_utils.py:605-708)"""
    for t, name in ((mean, f"mean_{side}"), (var, f"cov_{side}"), (weight, f"weight_{side}")):
        if not isinstance(t, Tensor):
            raise ValueError(f"`{name}` is expected to be a torch.Tensor, got `{type(t)}` instead.")
    if mean.dim() < 2 or var.dim() < 2:
        raise ValueError(f"`mean_{side}` / `cov_{side}` should be 1-dim vectors with a leading component dimension")
    if bool((var < 0).any()):
        raise ValueError(f"`cov_{side}` is expected to be a valid variance vector with positive entries.")
    total = weight.sum(-1)
    if bool((weight < -tol).any()) or bool((total < 1 - tol).any()) or bool((total > 1 + tol).any()):
        raise ValueError(f"`weight_{side}` is expected to be a valid probability vector with positive entries that sum up to 1.")
    if mean.size(-1) != var.size(-1):
        raise ValueError(f"All the inputs dimensionalities should match, got {[mean.size(-1), var.size(-1)]}")
    if not (mean.size(-2) == var.size(-2) == weight.size(-1)):
        raise ValueError(f"All the inputs component dimension should match, got {[mean.size(-2), var.size(-2), weight.size(-1)]}")


def batch_ot_gmm(mean_source: Tensor, mean_target: Tensor, cov_source: Tensor, cov_target: Tensor, diag: bool,
                 weight_source: Optional[Tensor] = None, weight_target: Optional[Tensor] = None, verbose: bool = False,
                 dtype=torch.double, **sinkhorn_kwargs) -> Tuple[Tensor, Tensor]:
    """Entropy-regularised W2^2 upper bound between two Gaussian mixtures (reference ot/w2_utils.py:197-270): the
    component-to-component Gaussian W2^2 as ground cost, the coupling of the mixture weights from ``sinkhorn_log`` on the
    cost scaled by its maximum, total = <cost, coupling>.  means [*, N, D] / [*, M, D]; diag=True: variances alike;
    diag=False: covariances [*, N, D, D] / [*, M, D, D] and the ground cost of ``batch_w2_dissimilarity_gaussian(make_pd=True)``."""
    if weight_source is None:
        weight_source = torch.ones_like(mean_source.select(dim=-1, index=0)) / mean_source.size(-2)
    if weight_target is None:
        weight_target = torch.ones_like(mean_target.select(dim=-1, index=0)) / mean_target.size(-2)
    if mean_source.size(-1) != mean_target.size(-1):
        raise ValueError("All the inputs dimensionalities should match")
    if diag:
        _check_mixture(mean_source, cov_source, weight_source, "source")
        _check_mixture(mean_target, cov_target, weight_target, "target")
        cost = batch_w2_dissimilarity_gaussian_diag(mean_source, mean_target, cov_source, cov_target, dtype=dtype)
    else:
        for mu, cv, wt in ((mean_source, cov_source, weight_source), (mean_target, cov_target, weight_target)):
            tot = wt.sum(-1)
            if bool((wt < -1e-5).any()) or bool((tot < 1 - 1e-5).any()) or bool((tot > 1 + 1e-5).any()):
                raise ValueError("mixture weights are expected to be valid probability vectors with positive entries that sum up to 1.")
            if not (mu.size(-2) == cv.size(-3) == wt.size(-1)) or cv.size(-1) != mu.size(-1) or cv.size(-2) != mu.size(-1):
                raise ValueError("All the inputs component / dimensionality sizes should match")
        cost = batch_w2_dissimilarity_gaussian(mean_source, mean_target, cov_source, cov_target, make_pd=True, verbose=verbose,
                                               dtype=dtype)
    max_per_mat = cost.max(-2, keepdim=True)[0].max(-1, keepdim=True)[0]
    coupling = sinkhorn_log(weight_source.to(dtype), weight_target.to(dtype), cost / max_per_mat, **sinkhorn_kwargs)
    return torch.sum(cost * coupling, dim=(-2, -1)), coupling


def _stochastic_operators_diag(cs: Tensor, ct: Tensor, pg_star: float) -> Tuple[Tensor, Tensor]:
    """eq. 19 for diagonal covariances (reference ot/w2_utils.py:732-751)"""
    t_star = torch.sqrt(cs / ct + STABILITY_CONST)
    big = cs > STABILITY_CONST
    pinv = torch.where(big, 1.0 / torch.where(big, cs, torch.ones_like(cs)), torch.zeros_like(cs))
    T = (1 - pg_star) * torch.sqrt(ct * cs) * pinv + pg_star
    return T, math.sqrt(1 - pg_star) * ct * (1 - ct * pinv * t_star ** 2)


def _stochastic_operators_full(cs: Tensor, ct: Tensor, pg_star: float) -> Tuple[Tensor, Tensor]:
    """eq. 19 for full matrices [nb, D, D] (reference ot/w2_utils.py:771-786): the source may be rank deficient (pseudo-inverse),
    T = (1-pg) Ct^1/2 (Ct^1/2 Cs Ct^1/2)^1/2 Ct^-1/2 Cs^+ + pg I;  Cw = sqrt(1-pg) Ct^1/2 (I - Ct^1/2 T* Cs^+ T* Ct^1/2) Ct^1/2 with
    T* the eq. 17 operator from the target to the source.  One decomposition of Ct serves its three functions."""
    eye = eye_like(cs)
    pinv_s = pinv_sym(cs)
    lam_t, vt_t = eigh_vectors(ct)
    rt = spectral_fn(lam_t.sqrt(), vt_t)
    irt = spectral_fn((lam_t + STABILITY_CONST).rsqrt(), vt_t)
    inner = eigvals_and_fn(matmul64(matmul64(rt, cs), rt), 1)[1]
    t_star = matmul64(matmul64(irt, inner), irt)
    T = (1 - pg_star) * matmul64(matmul64(matmul64(rt, inner), irt), pinv_s) + pg_star * eye
    core = eye - matmul64(matmul64(matmul64(matmul64(rt, t_star), pinv_s), t_star), rt)
    return T, math.sqrt(1 - pg_star) * matmul64(matmul64(rt, core), rt)


def compute_transport_operators(cov_source: Tensor, cov_target: Tensor, stochastic: bool, diag: bool,
                                pg_star: float = 0, make_pd: bool = False, verbose: bool = False,
                                dtype=torch.double) -> Tuple[Tensor, Tensor]:
    """Eq. 17 / eq. 19 of Freirich et al. (reference ot/w2_utils.py:391-458): (T, Cw).  Non-stochastic:
    T = (1-pg) Cs^-1/2 (Cs^1/2 Ct Cs^1/2)^1/2 Cs^-1/2 + pg I, Cw = 0.  Stochastic: the operator that also handles a degenerate
    source, with the noise covariance Cw; as in the reference, a noise covariance that is not positive definite falls back to
    eq. 17 only when ``verbose`` is set (the fallback sits behind the same condition as its warning)."""
    if stochastic:
        if diag:
            cs, ct = cov_source.to(dtype).clone(), cov_target.to(dtype)
            cs[cs < STABILITY_CONST] = 0
            if (cs < 0).any() or (ct < 0).any():
                raise ValueError("variances are expected to have positive entries.")
            T, Cw = _stochastic_operators_diag(cs, ct, pg_star)
            if bool((Cw <= 0).any()) and verbose:
                warnings.warn("The noise covariance matrix is not positive definite. Falling back to the non-stochastic implementation")
                return compute_transport_operators(cs, ct, False, True, pg_star, make_pd, verbose, dtype)
            return T, Cw
        d = cov_source.shape[-1]
        lead = torch.broadcast_shapes(cov_source.shape[:-2], cov_target.shape[:-2])
        flat = lambda m: m.double().expand(*lead, d, d).reshape(-1, d, d).contiguous()  # noqa: E731
        cs = _require_spd(flat(cov_source), "cov_source", make_pd, False, verbose)           # 'spsd'
        ct = _require_spd(flat(cov_target), "cov_target", make_pd, True, verbose)            # 'spd'
        T, Cw = _stochastic_operators_full(cs, ct, pg_star)
        if verbose and not bool(torch.logical_and(is_symmetric(Cw), eigvals_and_fn(Cw, 0)[0].min(-1)[0] > 0).all()):
            warnings.warn("The noise covariance matrix is not positive definite. Falling back to the non-stochastic implementation")
            return compute_transport_operators(cs.reshape(*lead, d, d), ct.reshape(*lead, d, d), False, False, pg_star, make_pd,
                                               verbose, dtype)
        return T.reshape(*lead, d, d).to(dtype), Cw.reshape(*lead, d, d).to(dtype)
    if diag:
        cs, ct = cov_source.to(dtype), cov_target.to(dtype)
        if (cs < 0).any() or (ct < 0).any():
            raise ValueError("variances are expected to have positive entries.")
        T = (1 - pg_star) * torch.sqrt(ct / cs + STABILITY_CONST) + pg_star
        return T, torch.zeros_like(T)
    d = cov_source.shape[-1]
    lead = torch.broadcast_shapes(cov_source.shape[:-2], cov_target.shape[:-2])
    cs, rs, irs = _spd_and_roots(cov_source.double().expand(*lead, d, d).reshape(-1, d, d).contiguous(), "cov_source", make_pd)
    ct = cov_target.double().expand(*lead, d, d).reshape(-1, d, d).contiguous()
    if not bool(is_symmetric(ct).all()):
        raise ValueError("`cov_target` should be symmetric.")
    inner = eigvals_and_fn(matmul64(matmul64(rs, ct), rs), 1)[1]
    T = (1 - pg_star) * matmul64(matmul64(irs, inner), irs) + pg_star * eye_like(cs)
    T = T.reshape(*lead, d, d).to(dtype)
    return T, torch.zeros_like(T)


def w2_and_transport_operator(mean_source: Tensor, mean_target: Tensor, spec_source, spec_target, pg_star: float = 0,
                              make_pd: bool = False, dtype=torch.double) -> Tuple[Tensor, Tensor, Tensor]:
    """``w2_gaussian`` and the non-stochastic full-matrix ``compute_transport_operators`` of the same pair of Gaussians in
    one go.  Separately they run ten Jacobi eigendecompositions (definiteness tests, square roots and inverse square
    roots of the same two covariances, twice over); here each covariance is decomposed once -- ``spec_*`` =
    (cov, eigvals, Vt) as ``GaussianModel.cov_spectrum`` returns them -- and every function of it is V f(lambda) V^T
    from that spectrum, which leaves four decompositions (the two covariances and the two inner products).  The
    arithmetic per step is that of the two functions above (reference ot/w2_utils.py:40-80, 756-769).
    Returns (W2^2 [*], T [*, D, D], Cw = 0)."""
    lib = _lib.load()
    (cs, lam_s, vt_s), (ct, lam_t, vt_t) = spec_source, spec_target
    lead, d = cs.shape[:-2], cs.shape[-1]
    f64c = lambda t, *shape: t.double().reshape(*shape).contiguous()  # noqa: E731
    cs3, ct3 = f64c(cs, -1, d, d), f64c(ct, -1, d, d)
    nb = cs3.shape[0]
    ms = mean_source.double().expand(*lead, d).reshape(-1, d).contiguous()
    mt = mean_target.double().expand(*lead, d).reshape(-1, d).contiguous()
    dev = cs3.device
    ws = torch.empty(lib.otvae_w2_transport_ws(nb, d), device=dev, dtype=torch.uint8)
    ews = torch.empty(lib.otvae_eigh_ws(2 * nb, d), device=dev, dtype=torch.uint8)
    w2 = torch.empty(nb, device=dev, dtype=torch.float64)
    T = torch.empty((nb, d, d), device=dev, dtype=torch.float64)
    flags = torch.empty(3, device=dev, dtype=torch.int32)
    # one native call: shifts, V f(lambda) V^T x 3, the two inner products, their square roots (one batched decomposition), the
    # tail and the operator -- no host read in between (the step-by-step composition was ~90 launches and four synchronisations)
    check(lib.otvae_w2_transport(ptr(ms), ptr(mt), ptr(cs3), ptr(ct3), ptr(f64c(lam_s, -1, d)), ptr(f64c(vt_s, -1, d, d)),
                                 ptr(f64c(lam_t, -1, d)), ptr(f64c(vt_t, -1, d, d)), nb, d, float(pg_star), int(bool(make_pd)),
                                 ptr(ws), ptr(ews), ptr(w2), ptr(T), ptr(flags), stream()), "otvae_w2_transport")
    bad_s, bad_t, asym = (int(v) for v in flags.tolist())   # the one host read: the reference's argument errors
    if not make_pd:
        for bad, name in ((bad_s, "cov_source"), (bad_t, "cov_target")):
            if bad:
                raise ValueError(f"`{name}` should be symmetric and positive definite. Use `make_pd=True` to automatically add a "
                                 "small value to the matrix diagonals.")
    if asym:
        raise ValueError("`cov_target_sqrt @ cov_source @ cov_target_sqrt` should be symmetric.")
    T = T.reshape(*lead, d, d).to(dtype)
    return w2.reshape(lead), T, torch.zeros_like(T)


def _transport_noise(shape, Cw: Tensor, diag: bool, make_pd: bool, noise_eps: Optional[Tensor], dtype) -> Tensor:
    """W ~ N(0, Cw) as the reference draws it (ot/w2_utils.py:521-525): diagonal: ``Normal(0, Cw)`` -- Cw is handed over as the
    SCALE --, full: ``MultivariateNormal(0, Cw)`` = chol(Cw) eps.  ``noise_eps``: the standard-normal draws to use (parity
    tests); otherwise drawn on the device."""
    Cw = Cw.to(dtype)
    if diag:
        if bool((Cw < 0).any()):
            raise ValueError("`Cw` is expected to be a valid variance vector with positive entries.")
        eps = noise_eps.to(Cw) if noise_eps is not None else torch.randn(shape, device=Cw.device, dtype=dtype)
        return Cw * eps
    d = Cw.shape[-1]
    cw = _require_spd(Cw.double().reshape(-1, d, d).contiguous(), "Cw", make_pd, True, False).reshape(Cw.shape)
    L = cholesky(cw)
    eps = noise_eps.to(L) if noise_eps is not None else torch.randn(shape, device=Cw.device, dtype=torch.double)
    # L eps per sample = eps L^T per batch: one [B, D] x [D, D] product per operator instead of B matrix-vector products
    if L.dim() == eps.dim() + 1 and L.shape[-3] == 1:
        return mm(eps, L.squeeze(-3).transpose(-1, -2)).to(dtype)
    if L.dim() == eps.dim():
        return mm(eps, L.transpose(-1, -2)).to(dtype)
    return mm(L, eps.unsqueeze(-1)).squeeze(-1).to(dtype)


def apply_transport(input: Tensor, mean_source: Tensor, mean_target: Tensor, T: Tensor, Cw: Optional[Tensor] = None,
                    diag: bool = False, make_pd: bool = False, verbose: bool = False, dtype=torch.double,
                    noise_eps: Optional[Tensor] = None) -> Tensor:
    """T (x - mean_source) + mean_target (+ W, W ~ N(0, Cw), when a non-zero noise covariance comes with a stochastic operator),
    computed in fp64; returns ``dtype``.  input [*, B, D] against [*, D]/[*, D, D] operators (already unsqueezed by the caller
    like the reference) or matching shapes."""
    # the reference's test is torch.allclose(Cw, 0) (ot/w2_utils.py:507): entries up to 1e-8 count as "no noise" -- a stochastic
    # operator of a full-rank source carries a Cw of that size (rounding + the 1e-8 regularisation) and is applied without it
    if Cw is not None and bool((~(Cw.abs() <= 1e-8)).any()):
        moved = apply_transport(input, mean_source, mean_target, T, None, diag, make_pd, verbose, dtype)
        return moved + _transport_noise(moved.shape, Cw, diag, make_pd, noise_eps, dtype)
    if diag:
        return (T.to(dtype) * (input.to(dtype) - mean_source.to(dtype)) + mean_target.to(dtype))
    lib = _lib.load()
    _lib.require_cuda(input, "input")
    d = input.shape[-1]
    if T.shape[-1] != d or T.shape[-2] != d:
        raise ValueError("All the inputs dimensionalities should match")
    # normalise to x [nb, B, D], operators [nb, ...]
    x = input
    Tm, ms, mt = T, mean_source, mean_target
    if Tm.dim() == x.dim() + 1 and Tm.shape[-3] == 1:  # operators carry an explicit singleton batch dim
        Tm, ms, mt = Tm.squeeze(-3), ms.squeeze(-2), mt.squeeze(-2)
    if x.dim() == Tm.dim() - 1:  # one vector per operator: [*, D]
        x = x.unsqueeze(-2)
        squeeze = True
    else:
        squeeze = False
    lead = x.shape[:-2]
    bsz = x.shape[-2]
    x3 = x.reshape(-1, bsz, d)
    x3 = (x3 if x3.dtype in _DT else x3.double()).contiguous()
    nb = x3.shape[0]
    Tm = Tm.double().expand(*lead, d, d).reshape(nb, d, d).contiguous()
    ms = ms.double().expand(*lead, d).reshape(nb, d).contiguous()
    mt = mt.double().expand(*lead, d).reshape(nb, d).contiguous()
    y = torch.empty_like(x3)
    check(lib.otvae_apply_transport(_dt(x3), ptr(x3), ptr(ms), ptr(mt), ptr(Tm), nb, bsz, d, ptr(y), stream()),
          "otvae_apply_transport")
    y = y.reshape(*lead, bsz, d)
    if squeeze:
        y = y.squeeze(-2)
    return y.to(dtype)


class W2Mixin(object):
    """Binds the w2 configuration (diag / stochastic / pg_star / make_pd / verbose / dtype) like the reference
    (ot/w2_utils.py:533-600)."""

    def __init__(self, **kwargs):
        self._orig_kwargs = dict(kwargs)
        self.stochastic = kwargs.pop("stochastic", False)
        self.diag = kwargs.pop("diag", False)
        self.pg_star = kwargs.pop("pg_star", 0.)
        self.make_pd = kwargs.pop("make_pd", False)
        self.verbose = kwargs.pop("verbose", False)
        self.dtype = kwargs.pop("dtype", torch.double)
        self.mean_cov = partial(mean_cov, diag=self.diag)
        self.batch_w2_dissimilarity_gaussian_diag = partial(batch_w2_dissimilarity_gaussian_diag, dtype=self.dtype)
        self.batch_w2_dissimilarity_gaussian = partial(batch_w2_dissimilarity_gaussian, make_pd=self.make_pd,
                                                       verbose=self.verbose, dtype=self.dtype)
        self.gaussian_barycenter = partial(gaussian_barycenter, diag=self.diag, dtype=self.dtype)
        self.batch_ot_gmm = partial(batch_ot_gmm, diag=self.diag, verbose=self.verbose, dtype=self.dtype)
        self.compute_transport_operators = partial(compute_transport_operators, diag=self.diag,
                                                   stochastic=self.stochastic, pg_star=self.pg_star,
                                                   make_pd=self.make_pd, verbose=self.verbose, dtype=self.dtype)

    def get_var_normal(self, distribution: Union[D.Normal, D.MultivariateNormal]):
        return distribution.variance if self.diag else distribution.covariance_matrix

    def instantiate_normal(self, *args, **kwargs):
        if self.diag:
            for k in ("covariance_matrix", "precision_matrix", "scale_tril"):
                kwargs.pop(k, None)
            return D.Independent(D.Normal(*args, **kwargs), 1)
        kwargs.pop("scale", None)
        return D.MultivariateNormal(*args, **kwargs)

    def w2_gaussian(self, mean_source: Tensor, mean_target: Tensor, cov_source: Tensor, cov_target: Tensor) -> Tensor:
        return w2_gaussian(mean_source, mean_target,
                           torch.diag_embed(cov_source) if self.diag else cov_source,
                           torch.diag_embed(cov_target) if self.diag else cov_target,
                           make_pd=self.make_pd, verbose=self.verbose, dtype=self.dtype)

    def apply_transport(self, inputs: Tensor, mean_source: Tensor, mean_target: Tensor, T: Tensor, Cw: Tensor,
                        batch_dim: Optional[int] = None, noise_eps: Optional[Tensor] = None) -> Tensor:
        return apply_transport(
            inputs,
            mean_source.unsqueeze(batch_dim) if batch_dim is not None else mean_source,
            mean_target.unsqueeze(batch_dim) if batch_dim is not None else mean_target,
            T.unsqueeze(batch_dim - bool(not self.diag)) if batch_dim is not None else T,
            Cw.unsqueeze(batch_dim - bool(not self.diag)) if (batch_dim is not None and Cw is not None) else Cw,
            diag=self.diag, make_pd=self.make_pd, verbose=self.verbose, dtype=self.dtype, noise_eps=noise_eps)

    def __repr__(self):
        return ", ".join(f"{k}={v}" for k, v in self._orig_kwargs.items())
