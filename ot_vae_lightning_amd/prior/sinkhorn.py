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
    argument the plan is treated as a constant, so d loss / d z_i = 2 sum_j pi_ij (z_i - y_j).
    Forward = ``torch.ops.otvae.sinkhorn_prior`` -> ``otvae_sinkhorn_prior_fwd`` (cost tiles + tile maxima on the matrix cores, the solve on C / max C with uniform
    marginals, the read-out), backward = ``otvae_ot_cost_grad`` (plan x samples on the matrix cores): no library GEMM and no
    ATen kernel on either side."""

    @staticmethod
    def forward(ctx, z, y, reg, max_iter, threshold, scale):
        from ..functional import PriorLane
        ctx.lane = PriorLane.active(z.device)
        if ctx.lane:  # beside the decoder, on the prior lane of a training engine's step (functional.PriorLane)
            PriorLane.hold(z.device, z, y)
            with PriorLane.section(z.device):
                cost, pi, iters = torch.ops.otvae.sinkhorn_prior(z, y, float(reg), int(max_iter), float(threshold), float(scale))
        else:
            cost, pi, iters = torch.ops.otvae.sinkhorn_prior(z, y, float(reg), int(max_iter), float(threshold), float(scale))
        ctx.save_for_backward(z, y, pi)
        ctx.scale = float(scale)
        ctx.mark_non_differentiable(iters)
        ctx.set_materialize_grads(False)  # no zeros tensor for the iteration count's "gradient"
        # the latents leave through this node too (an alias of z): the gradient the decoder sends back and the OT term's
        # own gradient are then summed inside otvae_ot_cost_grad instead of by an autograd accumulation kernel
        return z.view_as(z), cost, iters

    @staticmethod
    def backward(ctx, gz_out, g, _giters):
        z, y, pi = ctx.saved_tensors
        if ctx.lane:
            from ..functional import PriorLane
            PriorLane.join(z.device)  # the plan (and the loss vector behind it) is complete from here on
        if g is None:  # only the latents were used downstream
            return gz_out, None, None, None, None, None
        return torch.ops.otvae.sinkhorn_prior_backward(g, gz_out, z, y, pi, ctx.scale), None, None, None, None, None


class SinkhornPrior(Prior):
    """Deterministic encoder + entropic OT between the minibatch of latents and a minibatch of N(0, I) draws.
    ``forward`` returns (z, loss[B], artifacts) with loss[b] = OT cost (identical for every b so that the VAE's
    ``prior_loss.mean()`` equals it).

    The loss VALUE is the reference's own composition (``sinkhorn_log`` on C / max C, read-out sum(C * pi): ot/w2_utils.py:265-269,
    276-319).  Its GRADIENT comes in two conventions, and they are NOT close:

    * ``differentiate_plan=False`` (default, the training configuration of BASELINE configs 2-3): the envelope form -- the plan is
      a constant, d loss / d z_i = 2 sum_j pi_ij (z_i - y_j).  This is what the composition gives when the solve runs under
      ``torch.no_grad()``; it is the gradient of the converged OT cost and is cheap (one plan-times-samples product).
    * ``differentiate_plan=True``: what autograd gives for the composition of the reference's functions as written -- the
      gradient also flows through the cost normalisation and through all ``max_iter`` iterations of ``sinkhorn_log``
      (``ot.sinkhorn_log``'s backward, csrc/sinkhorn_diff.hip).  With reg = 0.05 and 50 iterations on a max-normalised cost the plan
      is far from converged and the two gradients differ by 37 % (N = 7), 93 % (N = 64), 97 % (N = 256) of the gradient's
      largest entry on the recorded problems (tests/golden/sinkhorn_autograd.npz: ``gz_full`` vs ``gz_envelope``, both from the
      reference's function).  Costs 2 max_iter + 5 extra launches per step and runs on the launch stream."""

    def __init__(self, reg: float = 0.05, max_iter: int = 50, threshold: float = 0., loss_coeff: float = 1.,
                 annealing_steps: int = 0, seed: int = None, differentiate_plan: bool = False):
        super().__init__(loss_coeff, annealing_steps)
        self.reg, self.max_iter, self.threshold = reg, max_iter, threshold
        self.seed = seed
        self.differentiate_plan = bool(differentiate_plan)
        self.last_iters = None  # device int32: iterations of the last solve (-1: the solver was starved, loss is NaN)

    def out_size(self, size):
        return size

    def sample(self, shape, device) -> Tensor:
        return torch.randn(*shape, device=device)

    def raise_if_starved(self) -> None:
        """Host check of the last solve (one device read): raises ``SinkhornSolverStarved`` if the single-launch solver gave up
        waiting for its other workgroups.  The training step itself never synchronises for this: a starved solve makes the
        loss NaN (every entry of the plan is), which is what a captured step can show; ``HipTrainer.close`` and callers that
        log losses call this."""
        if self.last_iters is not None:
            W.raise_if_solver_starved(self.last_iters)

    def _draw(self, like: Tensor) -> Tensor:
        """prior samples N(0, I) from the device-side counter-based generator: inside a captured step every replay draws
        fresh samples (the call counter lives in device memory and advances in the kernel)"""
        from .. import functional as HF
        key = self.__dict__.get("_rng_key")
        if key is None or key.device != like.device:
            key = self.__dict__["_rng_key"] = HF.new_rng_key(like.device, self.seed)
        return HF.normal_like(like, key, stream_id=1)

    def forward(self, x: Tensor, step: int, prior_samples: Optional[Tensor] = None) -> Prior.EncodingResults:
        # loss_coeff x annealing is folded into the read-out kernel (and the backward's scale): no separate multiply
        return self.encode(x, prior_samples=prior_samples, _scale=float(self.loss_coeff * self.annealing(step)))

    def encode(self, x: Tensor, prior_samples: Optional[Tensor] = None, _scale: float = 1.0) -> Prior.EncodingResults:
        z = x
        zf = z.flatten(1)
        if prior_samples is None:
            prior_samples = self._draw(zf) if zf.dtype == torch.float32 else torch.randn_like(zf)
        if self.differentiate_plan and torch.is_grad_enabled() and zf.requires_grad:
            y = prior_samples.flatten(1).to(zf.dtype)
            n, m = zf.shape[0], y.shape[0]
            C = W.sq_euclidean_cost(zf, y)                     # differentiable (plan-times-samples kernel in its backward)
            Cn = C / C.amax()                                  # the maximum takes part in the gradient, as under autograd
            a = torch.full((n,), 1.0 / n, device=zf.device, dtype=zf.dtype)
            b = torch.full((m,), 1.0 / m, device=zf.device, dtype=zf.dtype)
            pi = W.sinkhorn_log(a, b, Cn, self.reg, self.max_iter, self.threshold)   # carries _SinkhornLogFn's backward
            loss = ((C * pi).sum() * _scale).expand(n)
            self.last_iters = None
            return z, loss, {"prior_samples": prior_samples}
        z_out, loss, self.last_iters = _SinkhornLossFn.apply(zf, prior_samples.flatten(1), self.reg, self.max_iter, self.threshold,
                                                             _scale)
        return z_out.view(z.shape), loss, {"prior_samples": prior_samples}
