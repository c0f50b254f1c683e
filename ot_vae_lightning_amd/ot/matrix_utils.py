"""Matrix utilities with the reference's names and semantics (ot/matrix_utils.py): eigh-based ``sqrtm``/``invsqrtm``,
``min_eig``/``is_pd``/``is_symmetric``/``is_spd``/``make_psd``, ``mean_cov``, ``eye_like``, ``STABILITY_CONST`` --
computed in fp64 on the GPU by the LDS-resident Jacobi eigensolver and small fp64 kernels of ``gaussian_ot.hip``."""
from typing import Tuple, Union

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["eye_like", "sqrtm", "invsqrtm", "is_spd", "is_pd", "is_symmetric", "min_eig", "make_psd", "mean_cov",
           "STABILITY_CONST", "eigvals_and_fn", "matmul64", "eigh_vectors", "spectral_fn", "psd_shift", "pinv_sym", "cholesky", "mm", "softmax_rows", "lse_rows"]

STABILITY_CONST = 1e-8


def _as_batch(m: Tensor) -> Tuple[Tensor, torch.Size]:
    _lib.require_cuda(m, "matrix")
    lead = m.shape[:-2]
    d = m.shape[-1]
    if m.shape[-2] != d:
        raise ValueError(f"expected square matrices, got {tuple(m.shape)}")
    return m.to(torch.float64).reshape(-1, d, d).contiguous(), lead


def eigvals_and_fn(matrices: Tensor, fn: int):
    """fn: 0 -> eigenvalues only, 1 -> sqrtm, 2 -> invsqrtm, 3 -> eigenvector rows.  Returns (eigvals [*, D], f(M) [*, D, D] | None), fp64.
    Reads the lower triangle like torch.linalg.eigh(UPLO='L') (reference matrix_utils.py:44)."""
    lib = _lib.load()
    a, lead = _as_batch(matrices)
    nb, d = a.shape[0], a.shape[-1]
    ws = torch.empty(lib.otvae_eigh_ws(nb, d), device=a.device, dtype=torch.uint8)
    ev = torch.empty((nb, d), device=a.device, dtype=torch.float64)
    out = torch.empty_like(a) if fn else None
    check(lib.otvae_eigh_fn(ptr(a), nb, d, fn, ptr(out), ptr(ev), ptr(ws), stream()), "otvae_eigh_fn")
    return ev.reshape(*lead, d), (out.reshape(*lead, d, d) if fn else None)


def eigh_vectors(matrices: Tensor) -> Tuple[Tensor, Tensor]:
    """(eigvals [*, D], Vt [*, D, D]) with Vt[..., k, :] the unit eigenvector of eigvals[..., k]: one decomposition from which
    several functions of the same matrix can be formed (``spectral_fn``) instead of one Jacobi run per function."""
    return eigvals_and_fn(matrices, 3)


def spectral_fn(f_of_eigvals: Tensor, vt: Tensor) -> Tensor:
    """V f(lambda) V^T from ``eigh_vectors`` output: [*, D], [*, D, D] -> [*, D, D] (fp64)"""
    lead, d = vt.shape[:-2], vt.shape[-1]
    v3 = vt.reshape(-1, d, d)
    out = matmul64(v3, f_of_eigvals.reshape(-1, d, 1) * v3, trans_a=True)
    return out.reshape(*lead, d, d)


def psd_shift(eigvals: Tensor, strict: bool, only_if_needed: bool) -> Tensor:
    """the diagonal shift ``make_psd`` adds, from eigenvalues already known ([*, D] -> [*]): |min(lambda_min, 0)| (+1e-8 if
    strict); with ``only_if_needed`` it is zero unless some matrix of the batch fails the definiteness test.  Device
    arithmetic only, the same as ``otvae_make_psd``."""
    lo = eigvals.min(-1)[0]
    shift = lo.clamp(max=0).abs()
    if strict:
        shift = shift + STABILITY_CONST
    if only_if_needed:
        bad = ~(lo > 0) if strict else ~(lo >= 0)
        shift = shift * bad.any().to(shift.dtype)
    return shift


def pinv_sym(matrices: Tensor) -> Tensor:
    """Moore-Penrose inverse of symmetric matrices from the eigendecomposition: V diag(1/lambda or 0) V^T with torch.linalg.pinv's
    default cut-off (|lambda| <= D eps max|lambda| -> 0), which the reference's stochastic operator applies to the source
    covariance (ot/w2_utils.py:779)."""
    lam, vt = eigh_vectors(matrices)
    d = matrices.shape[-1]
    cut = d * torch.finfo(torch.float64).eps * lam.abs().max(-1, keepdim=True)[0]
    inv = torch.where(lam.abs() > cut, 1.0 / lam, torch.zeros_like(lam))
    return spectral_fn(inv, vt).to(matrices.dtype)


def cholesky(matrices: Tensor) -> Tensor:
    """lower Cholesky factor [*, D, D] (fp64 on the device; NaN where a matrix is not positive definite)"""
    lib = _lib.load()
    a, lead = _as_batch(matrices)
    out = torch.empty_like(a)
    check(lib.otvae_cholesky(ptr(a), a.shape[0], a.shape[-1], ptr(out), None, stream()), "otvae_cholesky")
    return out.reshape(*lead, a.shape[-1], a.shape[-1])


def eye_like(matrices: Tensor) -> Tensor:
    return torch.eye(matrices.shape[-2], matrices.shape[-1], device=matrices.device,
                     dtype=matrices.dtype).expand_as(matrices)


def sqrtm(matrices: Tensor) -> Tensor:
    """V sqrt(lambda) V^T of a batch of SPSD matrices (NaN where an eigenvalue is negative, like the reference)."""
    return eigvals_and_fn(matrices, 1)[1].to(matrices.dtype)


def invsqrtm(matrices: Tensor) -> Tensor:
    return eigvals_and_fn(matrices, 2)[1].to(matrices.dtype)


def is_symmetric(matrices: Tensor) -> Tensor:
    if matrices.size(-1) != matrices.size(-2):
        return torch.zeros(matrices.shape[:-2], dtype=torch.bool, device=matrices.device)
    return torch.sum((matrices - matrices.transpose(-2, -1)) ** 2, dim=(-1, -2)) < STABILITY_CONST


def min_eig(matrices: Tensor) -> Tensor:
    return eigvals_and_fn(matrices, 0)[0].min(dim=-1)[0].to(matrices.dtype)


def is_pd(matrices: Tensor, strict=True) -> Tensor:
    return min_eig(matrices) > 0 if strict else min_eig(matrices) >= 0


def is_spd(matrices: Tensor, strict=True) -> Tensor:
    return torch.logical_and(is_symmetric(matrices), is_pd(matrices, strict=strict)).bool()


def make_psd(matrices: Tensor, strict: bool = False, return_correction: bool = False, diag: bool = False,
             only_if_needed: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
    """Adds |min(lambda_min, 0)| (+1e-8 if ``strict``) to every diagonal (reference matrix_utils.py:123-142).
    ``only_if_needed`` applies the shift only when some matrix of the batch fails the definiteness test, decided on
    the device (the reference's ``_validate_args(make_pd=True)`` path, w2_utils.py:667-669, without the host sync)."""
    if diag:
        smallest = matrices.min(-1)[0]
        corr = smallest.clamp(max=0).abs()
        if strict:
            corr = corr + STABILITY_CONST
        res = matrices + corr[..., None]
        return (res, corr) if return_correction else res
    lib = _lib.load()
    a, lead = _as_batch(matrices)
    a = a.clone() if a.data_ptr() == matrices.data_ptr() else a
    nb, d = a.shape[0], a.shape[-1]
    ev, _ = eigvals_and_fn(a, 0)
    check(lib.otvae_make_psd(ptr(a), ptr(ev), nb, d, int(strict), int(only_if_needed), stream()), "otvae_make_psd")
    res = a.reshape(*lead, d, d).to(matrices.dtype)
    if return_correction:
        corr = ev.min(-1)[0].clamp(max=0).abs().reshape(lead)
        if strict:
            corr = corr + STABILITY_CONST
        return res, corr.to(matrices.dtype)
    return res


def mean_cov(sum: Tensor, sum_corr: Tensor, num_obs: Union[Tensor, int, float], diag: bool = False):
    """mean = sum/n, cov = sum_corr/n - mean mean^T (reference matrix_utils.py:145-158), fp64 on the device."""
    lib = _lib.load()
    _lib.require_cuda(sum, "sum")
    d = sum.shape[-1]
    lead = sum.shape[:-1]
    sx = sum.to(torch.float64).reshape(-1, d).contiguous()
    sxx = sum_corr.to(torch.float64).reshape(-1, d if diag else d * d).contiguous()
    nb = sx.shape[0]
    n = torch.as_tensor(num_obs, dtype=torch.float64, device=sum.device).expand(lead).reshape(-1).contiguous()
    mean = torch.empty_like(sx)
    cov = torch.empty_like(sxx)
    check(lib.otvae_mean_cov(ptr(n), ptr(sx), ptr(sxx), nb, d, int(diag), ptr(mean), ptr(cov), stream()),
          "otvae_mean_cov")
    cov = cov.reshape(*lead, d) if diag else cov.reshape(*lead, d, d)
    return mean.reshape(*lead, d).to(sum.dtype), cov.to(sum_corr.dtype)


def matmul64(a: Tensor, b: Tensor, trans_a: bool = False, trans_b: bool = False) -> Tensor:
    """Batched fp64 product of [nb, m, k] x [nb, k, n] (operands may be 2-D = broadcast over the batch)."""
    lib = _lib.load()
    a_b, b_b = a.dim() == 2, b.dim() == 2
    a3 = a.contiguous() if not a_b else a.contiguous().unsqueeze(0)
    b3 = b.contiguous() if not b_b else b.contiguous().unsqueeze(0)
    nb = max(a3.shape[0], b3.shape[0])
    m, k = (a3.shape[2], a3.shape[1]) if trans_a else (a3.shape[1], a3.shape[2])
    n = b3.shape[1] if trans_b else b3.shape[2]
    out = torch.empty((nb, m, n), device=a.device, dtype=torch.float64)
    check(lib.otvae_gemm_f64(int(trans_a), int(trans_b), nb, m, n, k, 1.0, ptr(a3), int(a3.shape[0] == 1 and nb > 1),
                             ptr(b3), int(b3.shape[0] == 1 and nb > 1), 0.0, ptr(out), stream()), "otvae_gemm_f64")
    return out


def _mm_raw(a: Tensor, b: Tensor, trans_a: bool = False, trans_b: bool = False) -> Tensor:
    """[nb, m, k] x [nb, k, n] (a 2-D operand is broadcast over the batch) in the operands' dtype (fp32 / fp64) on our kernels"""
    if a.dtype == torch.float64:
        return matmul64(a, b, trans_a, trans_b)
    lib = _lib.load()
    a_b, b_b = a.dim() == 2, b.dim() == 2
    a3 = a.contiguous() if not a_b else a.contiguous().unsqueeze(0)
    b3 = b.contiguous() if not b_b else b.contiguous().unsqueeze(0)
    nb = max(a3.shape[0], b3.shape[0])
    m, k = (a3.shape[2], a3.shape[1]) if trans_a else (a3.shape[1], a3.shape[2])
    n = b3.shape[1] if trans_b else b3.shape[2]
    out = torch.empty((nb, m, n), device=a.device, dtype=torch.float32)
    check(lib.otvae_gemm_f32(int(trans_a), int(trans_b), nb, m, n, k, 1.0, ptr(a3), int(a3.shape[0] == 1 and nb > 1),
                             ptr(b3), int(b3.shape[0] == 1 and nb > 1), 0.0, ptr(out), stream()), "otvae_gemm_f32")
    return out


class _MatmulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _mm_raw(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _mm_raw(g, b, trans_b=True)                  # [nb, m, n] x [nb, k, n]^T
            if a.dim() == 2:
                ga = ga.sum(0)
        if ctx.needs_input_grad[1]:
            gb = _mm_raw(a, g, trans_a=True)                  # [nb, m, k]^T x [nb, m, n]
            if b.dim() == 2:
                gb = gb.sum(0)
        return ga, gb


def mm(a: Tensor, b: Tensor) -> Tensor:
    """``a @ b`` for [*, m, k] x [*, k, n] (equal leading shapes, or a 2-D operand shared by the batch) on the library's own GEMM
    kernels, differentiable; the dtype is ``a``'s (fp32 or fp64)."""
    _lib.require_cuda(a, "a")
    if a.dtype not in (torch.float32, torch.float64):
        raise TypeError("mm computes in fp32 or fp64")
    b = b.to(a.dtype)
    lead = a.shape[:-2] if a.dim() > 2 else b.shape[:-2]
    a3 = a if a.dim() == 2 else a.reshape(-1, *a.shape[-2:])
    b3 = b if b.dim() == 2 else b.reshape(-1, *b.shape[-2:])
    if a3.dim() == 3 and b3.dim() == 3 and a3.shape[0] != b3.shape[0]:
        if a.shape[:-2] != b.shape[:-2]:   # general broadcasting of the leading dimensions: materialise it
            lead = torch.broadcast_shapes(a.shape[:-2], b.shape[:-2])
            a3 = a.expand(*lead, *a.shape[-2:]).reshape(-1, *a.shape[-2:])
            b3 = b.expand(*lead, *b.shape[-2:]).reshape(-1, *b.shape[-2:])
    out = _MatmulFn.apply(a3, b3)
    if a.dim() == 2 and b.dim() == 2:
        return out[0]
    return out.reshape(*lead, out.shape[-2], out.shape[-1])


class _SoftmaxRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        lib = _lib.load()
        x2 = x.contiguous()
        y = torch.empty_like(x2)
        rows, k = x2.numel() // x2.shape[-1], x2.shape[-1]
        check(lib.otvae_softmax_rows(int(x2.dtype == torch.float64), ptr(x2), rows, k, float(scale), ptr(y), stream()), "otvae_softmax_rows")
        ctx.save_for_backward(y)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        rows, k = y.numel() // y.shape[-1], y.shape[-1]
        check(_lib.load().otvae_softmax_rows_bwd(int(y.dtype == torch.float64), ptr(y), ptr(gy), rows, k, ctx.scale, ptr(gx), stream()),
              "otvae_softmax_rows_bwd")
        return gx, None


def softmax_rows(x: Tensor, scale: float = 1.0) -> Tensor:
    """softmax(scale * x, dim=-1) on the library's kernel (fp32 / fp64), differentiable: the assignment distributions of the mixture models"""
    _lib.require_cuda(x, "x")
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("softmax_rows computes in fp32 or fp64")
    return _SoftmaxRowsFn.apply(x, scale)


class _LseRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x2 = x.contiguous()
        rows, k = x2.numel() // x2.shape[-1], x2.shape[-1]
        out = torch.empty(x2.shape[:-1], device=x2.device, dtype=x2.dtype)
        check(_lib.load().otvae_lse_rows(int(x2.dtype == torch.float64), ptr(x2), rows, k, ptr(out), stream()), "otvae_lse_rows")
        ctx.save_for_backward(x2, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, out = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x2)
        rows, k = x2.numel() // x2.shape[-1], x2.shape[-1]
        check(_lib.load().otvae_lse_rows_bwd(int(x2.dtype == torch.float64), ptr(x2), ptr(out), ptr(g), rows, k, ptr(gx), stream()),
              "otvae_lse_rows_bwd")
        return gx


def lse_rows(x: Tensor) -> Tensor:
    """torch.logsumexp(x, dim=-1) on the library's kernel (fp32 / fp64), differentiable: the mixture log-density read-out"""
    _lib.require_cuda(x, "x")
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("lse_rows computes in fp32 or fp64")
    if x.numel() == 0:
        return x.new_empty(x.shape[:-1])
    return _LseRowsFn.apply(x)
