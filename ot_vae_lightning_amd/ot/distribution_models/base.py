"""``DistributionModel`` abstract base with the reference's contract (ot/distribution_models/base.py:30-161):
``update`` / ``fit`` / ``predict`` / ``w2`` / ``reset`` over samples ``[*leading_shape, batch, dim]``, running
statistics optionally all-reduced through the injected DDP callables."""
from abc import ABC, abstractmethod
from functools import partial
from typing import Any, Optional

import torch
import torch.distributions as D
import torch.nn as nn
from torch import Tensor

from ... import utils
from ..matrix_utils import softmax_rows

__all__ = ["DistributionModel", "gumbel_weights", "MIXTURE_MODES"]

MIXTURE_MODES = ("argmax", "sample", "mean", "gumbel-softmax", "gumbel-hardmax")   # MixtureMixin.Mode, base.py:166


def gumbel_weights(energy: Tensor, temperature: float, hard: bool, noise: Optional[Tensor] = None) -> Tensor:
    """The 'gumbel-softmax' / 'gumbel-hardmax' assignment weights of ``MixtureMixin.assign`` (reference base.py:234-235:
    ``F.gumbel_softmax(energy, tau=temperature, hard='hard' in mode)``): softmax((energy + g) / T) with g ~ Gumbel(0, 1), and for
    the hard variant its one-hot argmax with the soft value's gradient (straight-through).  ``noise``: the Gumbel draws to use
    (parity tests inject the reference's); otherwise they are drawn on the device, -log(Exponential(1))."""
    g = noise.to(energy) if noise is not None else -torch.empty_like(energy).exponential_().log()
    soft = softmax_rows(energy + g, 1.0 / temperature)
    if not hard:
        return soft
    index = soft.argmax(-1, keepdim=True)
    one_hot = torch.zeros_like(soft).scatter_(-1, index, 1.0)
    return one_hot - soft.detach() + soft


class DistributionModel(nn.Module, utils.DDPMixin, ABC):
    Distribution = None

    def __init__(self, *size: int, reduce_on_update: bool = True, update_decay: Optional[float] = None,
                 update_with_autograd: bool = False, device=None, dtype=None, **ddp_kwargs):
        nn.Module.__init__(self)
        utils.DDPMixin.__init__(self, **ddp_kwargs)
        self.leading_shape = torch.Size(size[:-1])
        self.dim = size[-1]
        self.reduce_on_update = reduce_on_update
        self.decay = update_decay
        self.ema_update = partial(utils.ema, decay=update_decay)
        self.register_buffer("vec_init", torch.randn(*self.vec_shape, dtype=dtype, device=device))
        self.register_buffer("mat_init", torch.randn(*self.vec_shape, self.dim, dtype=dtype, device=device))
        self.update_with_autograd = update_with_autograd

    @property
    def vec_shape(self):
        return (*self.leading_shape, self.dim)

    def _validate_samples(self, samples: Tensor) -> None:
        # the leading dimensions must be broadcast-compatible with the model's: equal, absent, or -- a codebook shared by all
        # positions of a latent, CodebookPrior(embed_dims=(1,)) over a (1, dim) model -- larger where the model's are 1.  (The
        # reference builds this ValueError without raising it, base.py:76-79: whatever broadcasts, runs.)
        try:
            torch.broadcast_shapes(samples.shape[:-2], self.leading_shape)
        except RuntimeError:
            raise ValueError(f"`samples` leading dimensions {tuple(samples.shape[:-2])} do not broadcast with "
                             f"{tuple(self.leading_shape)}") from None
        if samples.size(-1) != self.dim:
            raise ValueError(f"`samples` are expected to have dimensionality {self.dim}")

    @abstractmethod
    def reset(self) -> None:
        """reset internal model states"""

    @property
    def distribution(self) -> D.Distribution:
        raise NotImplementedError()

    @property
    def variances(self) -> Tensor:
        raise NotImplementedError()

    def forward(self, samples: Tensor) -> Any:
        self._validate_samples(samples)
        if self.training and not self.update_with_autograd:
            self.update(samples)
        return self.predict(samples)

    @abstractmethod
    def update(self, samples: Tensor) -> None:
        """on-the-fly update of the running statistics"""

    @abstractmethod
    def fit(self, samples: Optional[Tensor] = None) -> None:
        """fit the parameters from the running statistics (and optional extra samples)"""

    @abstractmethod
    def predict(self, samples: Tensor) -> Any:
        """model dependent"""

    @abstractmethod
    def w2(self, other) -> Tensor:
        """squared W2 distance to another distribution"""

    def extra_repr(self) -> str:
        return (f"leading_dim={tuple(self.leading_shape)}, dim={self.dim}, decay={self.decay}, "
                f"update_with_autograd={self.update_with_autograd}")
