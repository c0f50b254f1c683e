"""``ConditionalGaussianPrior``: Gaussian prior conditioned on a class label (reference prior/conditional_gaussian.py:
30-123): q(z | x) from the re-parametrised encoder output, p(z | y) = N(mu_y, exp(log_std_y)^2) from two class
embeddings that are either learned by gradient descent or tracked as exponential moving averages of the observed q's.
The re-parametrisation and the closed-form KL(q || p_y) (and its backward, including the gradients of the gathered
embedding rows) run in one HIP kernel each; the EMA update is a [classes x batch] one-hot product on tiny tensors."""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
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
        shape[1] //= 2
        if eps is None:
            eps = torch.randn(shape, device=x.device, dtype=x.dtype)
        pm, pl = self._mu(labels), self._log_std(labels)            # [B, prod(dim)] rows of the class embeddings
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
    def ema_update(self, q, labels: Tensor) -> None:
        one_hot = F.one_hot(labels, num_classes=self.num_classes).type(q.mean.dtype)      # [B, classes]
        sizes = one_hot.sum(dim=0)
        mu_sum = one_hot.transpose(-2, -1) @ q.mean.flatten(1)
        log_std_sum = one_hot.transpose(-2, -1) @ q.stddev.log().flatten(1)
        utils.ema_inplace(self._size, self.reduce(sizes), decay=self.decay)
        utils.ema_inplace(self._mu_avg, self.reduce(mu_sum), decay=self.decay)
        utils.ema_inplace(self._log_std_avg, self.reduce(log_std_sum), decay=self.decay)
        sizes = utils.laplace_smoothing(self._size, self.num_classes, self.eps)
        self._mu.weight.copy_(self._mu_avg.data / sizes.unsqueeze(-1))
        self._log_std.weight.copy_(self._log_std_avg.data / sizes.unsqueeze(-1))

    def forward(self, x: Tensor, step: int, labels: Tensor, eps: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, dict]:  # noqa
        # loss_coeff * annealing is folded into the kernel, as in GaussianPrior.forward
        return self._encode(x, float(self.loss_coeff * self.annealing(step)), labels, eps)
