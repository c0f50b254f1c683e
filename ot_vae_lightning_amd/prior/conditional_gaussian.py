"""``ConditionalGaussianPrior``: Gaussian prior conditioned on a class label (reference prior/conditional_gaussian.py:
30-123): q(z | x) from the re-parametrised encoder output, p(z | y) = N(mu_y, exp(log_std_y)^2) from two class
embeddings that are either learned by gradient descent or tracked as exponential moving averages of the observed q's.
The re-parametrisation and the closed-form KL(q || p_y) (and its backward, including the gradients of the gathered
embedding rows) run in one HIP kernel each; the EMA statistics are one [classes x batch] membership product over the
packed (mean | log-std | 1) rows, followed by a single all-reduce."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor
from torch.distributions import Normal

from .. import functional as HF
from .. import utils
from .base import Prior
from .gaussian import GaussianPrior, _LazyNormal

__all__ = ["ConditionalGaussianPrior"]


class ConditionalGaussianPrior(GaussianPrior, utils.DDPMixin):
    def __init__(self, dim, num_classes: int, loss_coeff: float = 1., empirical_kl: bool = False, reparam_dim: int = 1,
                 annealing_steps: int = 0, fixed_var: bool = False, embedding_ema_decay: Optional[float] = None,
                 eps: float = 1e-5, **ddp_kwargs):
        GaussianPrior.__init__(self, loss_coeff, empirical_kl, reparam_dim, annealing_steps, fixed_var)
        utils.DDPMixin.__init__(self, **ddp_kwargs)
        self.dim = tuple(int(d) for d in dim)
        self.num_classes, self.decay, self.eps = num_classes, embedding_ema_decay, eps
        n = int(np.prod(self.dim))
        self._mu = torch.nn.Embedding(num_classes, n, _weight=-torch.rand(num_classes, n))
        self._log_std = torch.nn.Embedding(num_classes, n, _weight=-torch.rand(num_classes, n))
        if self.decay is not None and self.decay > 0:
            self.register_buffer("_size", torch.zeros(num_classes))
            self.register_buffer("_mu_avg", torch.zeros_like(self._mu.weight))
            self.register_buffer("_log_std_avg", torch.zeros_like(self._log_std.weight))
            self._mu.requires_grad_(False)
            self._log_std.requires_grad_(False)

    def p(self, labels: Tensor) -> Normal:
        return Normal(self._mu(labels).unflatten(1, self.dim), self._log_std(labels).unflatten(1, self.dim).exp())

    def _encode(self, x: Tensor, coeff: float, labels: Tensor, eps: Optional[Tensor] = None):
        shape = list(x.shape)
        if not self.fixed_var:
            shape[self.reparam_dim] //= 2
        if eps is None:
            eps = torch.randn(shape, device=x.device, dtype=x.dtype)
        # [B, prod(dim)] rows of the class embeddings (nn.Embedding's lookup with the library's deterministic backward)
        pm, pl = HF.embedding(self._mu.weight, labels), HF.embedding(self._log_std.weight, labels)
        if self.fixed_var or self.empirical_kl or self.reparam_dim not in (1, 1 - x.dim()):
            z, loss = HF.gaussian_prior_conditional_ex(x, eps, pm, pl, coeff, self.empirical_kl, self.fixed_var, self.reparam_dim)
        else:
            z, loss = HF.gaussian_prior_conditional(x, eps, pm, pl, coeff)
        q = self.reparametrization(x.detach())
        if self.decay is not None and self.decay > 0 and self.training:
            self.ema_update(q, labels)
        # built on first use: constructing a Normal validates its arguments with a host synchronisation, which a
        # hipGraph capture of the training step cannot contain
        prior = _LazyNormal(lambda: self._mu(labels).detach().unflatten(1, self.dim),
                            lambda: self._log_std(labels).detach().unflatten(1, self.dim).exp())
        return z, loss, {"prior": prior, "distribution": q}

    def encode(self, x: Tensor, labels: Tensor, eps: Optional[Tensor] = None) -> Prior.EncodingResults:  # noqa
        return self._encode(x, 1.0, labels, eps)

    def sample(self, shape, device, labels: Tensor) -> Tensor:  # noqa
        return self.p(labels).sample().to(device)

    @torch.no_grad()
    def _class_sums(self, q, labels: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """per class: how many samples of the batch carry it, and the sums of their q means / log-stds.  The three
        statistics ride in one [B, 2n+1] matrix so that a single [classes x B] membership product and a single
        all-reduce (one RCCL call per step instead of three) produce them."""
        n = q.mean[0].numel()
        rows = torch.cat([q.mean.flatten(1), q.stddev.log().flatten(1), q.mean.new_ones(labels.numel(), 1)], dim=1)
        member = (torch.arange(self.num_classes, device=labels.device).unsqueeze(1) == labels.unsqueeze(0)).type_as(rows)
        from ..ot.matrix_utils import mm
        sums = self.reduce(mm(member.contiguous(), rows.contiguous()))                # [classes, 2n+1], otvae_gemm_f32
        return sums[:, 2 * n], sums[:, :n], sums[:, n:2 * n]

    @torch.no_grad()
    def ema_update(self, q, labels: Tensor) -> None:
        """moving averages of the class statistics (decay ``embedding_ema_decay``), then embedding row = average sum /
        Laplace-smoothed average count (reference conditional_gaussian.py:106-120)"""
        tracked = (self._size, self._mu_avg, self._log_std_avg)
        for average, batch_sum in zip(tracked, self._class_sums(q, labels)):
            utils.ema_inplace(average, batch_sum, decay=self.decay)
        counts = utils.laplace_smoothing(self._size, self.num_classes, self.eps).unsqueeze(-1)
        for embedding, average in ((self._mu, self._mu_avg), (self._log_std, self._log_std_avg)):
            embedding.weight.copy_(average / counts)

    def forward(self, x: Tensor, step: int, labels: Tensor, eps: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, dict]:  # noqa
        # loss_coeff * annealing is folded into the kernel, as in GaussianPrior.forward
        return self._encode(x, float(self.loss_coeff * self.annealing(step)), labels, eps)
