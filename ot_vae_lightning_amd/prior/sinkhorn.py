"""Minibatch optimal-transport priors for BASELINE configs 3-4.  The reference has no such class: these
are new ``Prior`` subclasses on the reference's plug-in API whose arithmetic is the reference's own
``sinkhorn_log`` (ot/w2_utils.py:276-319) composed with a squared-euclidean cost normalised by its maximum and the
``sum(C * pi)`` read-out, exactly as ``batch_ot_gmm`` does (ot/w2_utils.py:265-269)."""
from typing import Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream
from ..ot import w2_utils as W
from .base import Prior

__all__ = ["SinkhornPrior"]


class _SinkhornLossFn(torch.autograd.Function):
    """loss = sum_ij C_ij pi_ij with C_ij = |z_i - y_j|^2 and pi the (detached) entropic plan: by the envelope
    argument the plan is treated as a constant, so d loss / d z_i = 2 sum_j pi_ij (z_i - y_j)."""

    @staticmethod
    def forward(ctx, z, y, reg, max_iter, threshold):
        C = W.sq_euclidean_cost(z, y)
        n, m = C.shape
        a = torch.full((n,), 1.0 / n, device=z.device, dtype=z.dtype)
        b = torch.full((m,), 1.0 / m, device=z.device, dtype=z.dtype)
        cmax = C.max()
        pi = W.sinkhorn_log(a, b, C / cmax, reg=reg, max_iter=max_iter, threshold=threshold)
        ctx.save_for_backward(z, y, pi)
        return W.ot_cost(C, pi)

    @staticmethod
    def backward(ctx, g):
        z, y, pi = (t.contiguous() for t in ctx.saved_tensors)
        n, d = z.shape
        if n * y.shape[0] * d >= (1 << 26):
            # a real GEMM ([1024 x 1024] x [1024 x 128] at the bench size): the library's tiles beat the small fused kernel
            gz = 2.0 * (pi.sum(1, keepdim=True) * z - pi @ y) * g
        else:
            # per-GPU batches of a few hundred: one fused launch instead of six (the library picks a 256 x 256 macro-tile
            # for the 256 x 256 x 256 product: 67 us)
            gz = torch.empty_like(z)
            check(_lib.load().otvae_ot_cost_grad(0 if z.dtype == torch.float32 else 1, ptr(z), ptr(y), ptr(pi), ptr(g.contiguous()),
                                                 n, y.shape[0], d, ptr(gz), stream()), "otvae_ot_cost_grad")
        return gz, None, None, None, None


class SinkhornPrior(Prior):
    """Deterministic encoder + entropic OT between the minibatch of latents and a minibatch of N(0, I) draws.
    ``forward`` returns (z, loss[B], artifacts) with loss[b] = OT cost (identical for every b so that the VAE's
    ``prior_loss.mean()`` equals it)."""

    def __init__(self, reg: float = 0.05, max_iter: int = 50, threshold: float = 0., loss_coeff: float = 1.,
                 annealing_steps: int = 0):
        super().__init__(loss_coeff, annealing_steps)
        self.reg, self.max_iter, self.threshold = reg, max_iter, threshold

    def out_size(self, size):
        return size

    def sample(self, shape, device) -> Tensor:
        return torch.randn(*shape, device=device)

    def forward(self, x: Tensor, step: int, prior_samples: Optional[Tensor] = None) -> Prior.EncodingResults:
        return super().forward(x, step, prior_samples=prior_samples)

    def encode(self, x: Tensor, prior_samples: Optional[Tensor] = None) -> Prior.EncodingResults:
        z = x
        zf = z.flatten(1)
        if prior_samples is None:
            prior_samples = torch.randn_like(zf)
        cost = _SinkhornLossFn.apply(zf, prior_samples.flatten(1), self.reg, self.max_iter, self.threshold)
        return z, cost.expand(z.shape[0]), {"prior_samples": prior_samples}
