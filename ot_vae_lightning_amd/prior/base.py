"""``Prior``: what sits between the encoder output and the decoder input (reference contract: prior/base.py:26-78).

A prior turns the encoder's tensor ``x`` into the latent ``z`` and charges a per-sample regulariser for it:

    prior(x, step, **kw) -> (z, loss[B], artifacts)     ``encode`` result with the loss weighted (below)
    prior.sample(shape, device) -> z                    draw latents without an encoder
    prior.out_size(size)                                shape of ``z`` for an ``x`` of ``size`` (batch axis excluded)

Weighting: ``loss_coeff`` times a half-cosine ramp that rises from 0 at step 0 to 1 at ``annealing_steps`` and stays
there (``annealing_steps=0`` disables the ramp).

Rules the subclasses of this package follow so that a step can be captured into a hipGraph by ``engine.HipTrainer``:
``encode`` launches kernels only (no ``.item()``, no host-side branching on device values), returns ``loss`` as a
device tensor of shape [B], and puts anything expensive to build on the host (``torch.distributions`` objects) behind a
lazy proxy in ``artifacts``.  ``step`` is a host integer; under graph capture the weight is therefore frozen at capture
time, which is exact whenever ``annealing_steps == 0`` (every shipped config).
"""
import math
from abc import ABC, abstractmethod
from typing import Dict, Tuple, Union

import torch.nn as nn
from torch import Tensor
from torch.distributions import Distribution

__all__ = ["Prior"]


def _half_cosine(progress: float) -> float:
    """0 at progress=0, 1 at progress>=1, cosine-shaped in between"""
    return 1.0 if progress >= 1.0 else 0.5 - 0.5 * math.cos(math.pi * progress)


class Prior(nn.Module, ABC):
    EncodingResults = Tuple[Tensor, Tensor, Dict[str, Union[Tensor, Distribution]]]

    def __init__(self, loss_coeff: float = 1., annealing_steps: int = 0):
        super().__init__()
        self._loss_coeff, self.annealing_steps = loss_coeff, annealing_steps

    @property
    def loss_coeff(self):
        return self._loss_coeff

    def annealing(self, step: int) -> float:
        return _half_cosine(step / self.annealing_steps) if self.annealing_steps > step else 1

    def forward(self, x: Tensor, step: int, **kwargs) -> "Prior.EncodingResults":
        z, loss, artifacts = self.encode(x, **kwargs)
        return z, loss * (self.loss_coeff * self.annealing(step)), artifacts

    @staticmethod
    def empirical_reverse_kl(p: Distribution, q: Distribution, z: Tensor) -> Tensor:
        """single-sample estimate of KL(q || p) at z ~ q, summed over everything but the batch axis"""
        gap = q.log_prob(z) - p.log_prob(z)
        return gap.flatten(1).sum(1) if gap.dim() > 1 else gap

    # ---- to be provided
    @abstractmethod
    def encode(self, x: Tensor) -> "Prior.EncodingResults":
        ...

    @abstractmethod
    def sample(self, shape, device) -> Tensor:
        ...

    @abstractmethod
    def out_size(self, size):
        ...
