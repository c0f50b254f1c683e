"""``GaussianW2Prior``: squared 2-Wasserstein (Gelbrich) distance between the minibatch's empirical Gaussian and a target
Gaussian as the prior term of the VAE loss -- "Gaussian W2 with empirical covariance" of BASELINE.json's north_star.  The
reference has no such class: this is a new ``Prior`` subclass on the reference's plug-in contract
(prior/base.py:42-78) whose arithmetic is the reference's own pieces composed under autograd,

    n, sum_x, sum_xx = GaussianModel._stats(z)          ot/distribution_models/gaussian_model.py:144-151   (fp64)
    mean, cov        = mean_cov(sum_x, sum_xx, n)        ot/matrix_utils.py:145-158
    loss             = w2_gaussian(mean, mean_t, cov, cov_t, make_pd=True)                 ot/w2_utils.py:40-80

with a hand-written backward (csrc/w2_prior.hip) instead of autograd through ``eigh``.  Golden vectors: the three reference
functions run under torch.autograd (tests/golden/w2_prior.npz)."""
from typing import Optional

import torch
from torch import Tensor

from .. import ops as _ops  # noqa: F401  (registers torch.ops.otvae.*)
from .base import Prior

__all__ = ["GaussianW2Prior"]


class _W2PriorFn(torch.autograd.Function):
    """``torch.ops.otvae.gaussian_w2_prior`` / ``..._backward`` (ops.py) plus one thing an operator cannot do: the latents leave
    through this node too (an alias of z, see prior/sinkhorn.py), so the decoder's gradient is added inside the backward kernel."""

    @staticmethod
    def forward(ctx, z, mut, covt, rt, v_init, warm, scale):
        from ..functional import PriorLane
        ctx.lane = PriorLane.active(z.device)
        if ctx.lane:  # beside the decoder, on the prior lane of a training engine's step (functional.PriorLane)
            PriorLane.hold(z.device, z, mut, covt, rt)
            with PriorLane.section(z.device):
                loss, mu, q, vt = torch.ops.otvae.gaussian_w2_prior(z, mut, covt, rt, v_init, warm, float(scale))
                if v_init is not None:  # the warm-start basis of the next step, behind the solve that produced it
                    v_init.copy_(vt)
                    warm.fill_(1)
        else:
            loss, mu, q, vt = torch.ops.otvae.gaussian_w2_prior(z, mut, covt, rt, v_init, warm, float(scale))
        ctx.save_for_backward(z, mu, q)
        ctx.target = (mut, rt)
        ctx.scale = float(scale)
        ctx.mark_non_differentiable(vt)
        return z.view_as(z), loss, vt

    @staticmethod
    def backward(ctx, gz_out, g, _gvt):
        z, mu, q = ctx.saved_tensors
        mut, rt = ctx.target
        if ctx.lane:
            from ..functional import PriorLane
            PriorLane.join(z.device)
        if g is None:
            return gz_out, None, None, None, None, None, None
        return (torch.ops.otvae.gaussian_w2_prior_backward(g, gz_out, z, mu, q, mut, rt, ctx.scale), None, None, None, None, None,
                None)


class GaussianW2Prior(Prior):
    """Deterministic encoder + W2^2( N(mean_B, cov_B), N(target_mean, target_cov) ) of the minibatch of latents (flattened to
    [B, D]); ``target_mean`` / ``target_cov`` default to the standard normal.  ``forward`` returns (z, loss[B], artifacts) with
    every loss entry equal to the distance, so the VAE's ``prior_loss.mean()`` is the distance.  The batch must hold more
    samples than latent dimensions for the empirical covariance to be positive definite (otherwise the reference's
    ``make_pd`` shift of 1e-8 applies and the gradient of the square root is ill-conditioned, exactly as under autograd)."""

    def __init__(self, loss_coeff: float = 1., annealing_steps: int = 0, target_mean: Optional[Tensor] = None,
                 target_cov: Optional[Tensor] = None):
        super().__init__(loss_coeff, annealing_steps)
        if target_cov is not None and (target_cov.dim() != 2 or target_cov.shape[0] != target_cov.shape[1]):
            raise ValueError("`target_cov` should be a 2-dim square matrix")
        if target_mean is not None and target_cov is not None and target_mean.shape[-1] != target_cov.shape[-1]:
            raise ValueError(f"All the inputs dimensionalities should match, got {[target_mean.shape[-1], target_cov.shape[-1]]}")
        self.register_buffer("target_mean", None if target_mean is None else target_mean.detach().double().clone())
        self.register_buffer("target_cov", None if target_cov is None else target_cov.detach().double().clone())
        self._root = None  # (validated target covariance, its square root), made on first use on the device
        # Warm start of the eigendecomposition: the eigenvectors of the previous training step's matrix are the start basis of this
        # step's solve (consecutive minibatch covariances are close: 2-4 Jacobi sweeps instead of ~9).  `_warm` says whether
        # `_v_prev` holds a basis yet; it is a BUFFER so that an engine capturing the step snapshots / restores it with the
        # other buffers and a captured step decides cold / warm per replay exactly as the eagerly issued one does.  The basis is
        # a product of exactly orthogonal rotations from step to step; its orthogonality error grows like sqrt(rotations) * 1e-16
        # (1e-12 after 1e8 of them), far below what the loss resolves.
        self.register_buffer("_warm", torch.zeros(1, dtype=torch.int32), persistent=False)
        self._v_prev = None

    def out_size(self, size):
        return size

    def _otvae_step_state(self, latent_shape, device):
        """(engine.HipTrainer's step guard) the warm-start basis is state a training step rewrites without being a registered buffer:
        allocated here, before the first step, so that a refused step can put the previous basis back together with `_warm`"""
        d = 1
        for n_ in latent_shape[1:]:
            d *= int(n_)
        if d > 128:
            return []
        if self._v_prev is None or self._v_prev.shape[-1] != d or self._v_prev.device != device:
            self._v_prev = torch.zeros((1, d, d), device=device, dtype=torch.float64)
            self._warm.zero_()
        return [self._v_prev]

    def sample(self, shape, device) -> Tensor:
        x = torch.randn(*shape, device=device)
        if self.target_cov is None and self.target_mean is None:
            return x
        flat = x.flatten(1).double()
        if self.target_cov is not None:
            from ..ot.matrix_utils import mm
            flat = mm(flat, self._target_root()[1][0].to(device))
        if self.target_mean is not None:
            flat = flat + self.target_mean.to(device)
        return flat.to(x.dtype).reshape(x.shape)

    def _target_root(self):
        if self._root is None or self._root[0].device != self.target_cov.device:
            from ..ot.w2_utils import _spd_and_roots
            cov, root, _ = _spd_and_roots(self.target_cov.reshape(1, *self.target_cov.shape).contiguous(), "target_cov", True)
            self._root = (cov.contiguous(), root.contiguous())
        return self._root

    def forward(self, x: Tensor, step: int) -> Prior.EncodingResults:
        # loss_coeff x annealing is folded into the tail kernel (and the backward's scale): no separate multiply
        return self.encode(x, _scale=float(self.loss_coeff * self.annealing(step)))

    def encode(self, x: Tensor, _scale: float = 1.0) -> Prior.EncodingResults:
        zf = x.flatten(1)
        if self.target_cov is not None and self.target_cov.shape[-1] != zf.shape[1]:
            raise ValueError(f"All the inputs dimensionalities should match, got {[zf.shape[1], self.target_cov.shape[-1]]}")
        covt, rt = self._target_root() if self.target_cov is not None else (None, None)
        d = zf.shape[1]
        warm_ok = self.training and zf.is_cuda and d <= 128
        if warm_ok and (self._v_prev is None or self._v_prev.shape[-1] != d or self._v_prev.device != zf.device):
            self._v_prev = torch.zeros((1, d, d), device=zf.device, dtype=torch.float64)
            self._warm.zero_()
        z_out, loss, vt = _W2PriorFn.apply(zf, self.target_mean, covt, rt, self._v_prev if warm_ok else None,
                                           self._warm if warm_ok else None, _scale)
        from ..functional import PriorLane
        if warm_ok and not PriorLane.active(zf.device):  # (on the prior lane the node itself refreshed the basis, behind its solve)
            with torch.no_grad():
                self._v_prev.copy_(vt)
                self._warm.fill_(1)
        return z_out.view(x.shape), loss, {}
