"""Abstract ``Prior`` with the reference's contract (prior/base.py:26-78):
``forward(x, step, **kw) -> (z, loss[B], artifacts)``, ``sample(shape, device)``, ``out_size(size)``;
the loss is scaled by ``loss_coeff`` times a cosine warm-up over ``annealing_steps``."""
from abc import ABC, abstractmethod
from math import cos, pi
from typing import Dict, Tuple, Union

import torch.nn as nn
from torch import Tensor
from torch.distributions import Distribution

__all__ = ["Prior"]


class Prior(nn.Module, ABC):
    EncodingResults = Tuple[Tensor, Tensor, Dict[str, Union[Tensor, Distribution]]]

    def __init__(self, loss_coeff: float = 1., annealing_steps: int = 0):
        nn.Module.__init__(self)
        self._loss_coeff = loss_coeff
        self.annealing_steps = annealing_steps

    @abstractmethod
    def encode(self, x: Tensor) -> "Prior.EncodingResults":
        """re-parametrisation / loss / re-sampling logic; called by ``forward``"""

    @abstractmethod
    def sample(self, shape, device) -> Tensor:
        """draw from the prior"""

    @abstractmethod
    def out_size(self, size):
        """size after ``encode`` of a tensor of ``size`` (without the batch dimension)"""

    @staticmethod
    def empirical_reverse_kl(p: Distribution, q: Distribution, z: Tensor) -> Tensor:
        return (q.log_prob(z) - p.log_prob(z)).sum(list(range(1, z.dim())))

    @property
    def loss_coeff(self):
        return self._loss_coeff

    def annealing(self, step: int) -> float:
        if self.annealing_steps > step:
            return 0.5 * cos(pi * (step / self.annealing_steps + 1)) + 0.5
        return 1

    def forward(self, x: Tensor, step: int, **kwargs) -> "Prior.EncodingResults":
        z, loss, artifacts = self.encode(x, **kwargs)
        loss = loss * (self.loss_coeff * self.annealing(step))
        return z, loss, artifacts
