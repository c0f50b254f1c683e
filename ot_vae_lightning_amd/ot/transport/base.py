"""``TransportOperator`` abstract base (reference ot/transport/base.py:28-173): owns a source and a target
``DistributionModel``; ``update(source_samples=, target_samples=)``, ``compute() -> distance``, ``transport(x)``,
``reset()``.  Sample storing (``store_source/target``) keeps raw samples for models that need them."""
from abc import ABC, abstractmethod
from typing import Optional

import torch
import torch.distributions as D
import torch.nn as nn
from torch import Tensor

from ... import utils
from ..distribution_models.base import DistributionModel

__all__ = ["TransportOperator"]


class TransportOperator(nn.Module, utils.DDPMixin, ABC):
    def __init__(self, *size: int, source_model: DistributionModel, target_model: DistributionModel,
                 reset_source: bool = True, reset_target: bool = True, store_source: bool = False,
                 store_target: bool = False, **ddp_kwargs):
        nn.Module.__init__(self)
        utils.DDPMixin.__init__(self, **ddp_kwargs)
        self.dim = size[-1]
        self.leading_shape = size[:-1]
        self.source_model, self.target_model = source_model, target_model
        self.reset_source, self.reset_target = reset_source, reset_target
        self.store_source, self.store_target = store_source, store_target
        if store_source:
            self.register_buffer("_source_samples", None)
        if store_target:
            self.register_buffer("_target_samples", None)
        if store_source or store_target:
            self.warn(f"The transport operator `{self.__class__.__name__}` will save all extracted features in "
                      "buffers. For large datasets this may lead to a large memory footprint.")

    def reset(self) -> None:
        if self.reset_source:
            if self.store_source:
                self._source_samples = None
            self.source_model.reset()
        if self.reset_target:
            if self.store_target:
                self._target_samples = None
            self.target_model.reset()

    @property
    def source_distribution(self) -> D.Distribution:
        return self.source_model.distribution

    @property
    def target_distribution(self) -> D.Distribution:
        return self.target_model.distribution

    @staticmethod
    def _append(store: Optional[Tensor], new: Tensor) -> Tensor:
        new = new.detach()
        return new if store is None else torch.cat([store, new.type_as(store)], dim=-2)

    def update(self, source_samples: Optional[Tensor] = None, target_samples: Optional[Tensor] = None) -> None:
        if source_samples is not None:
            self.source_model.update(source_samples)
            if self.store_source:
                self._source_samples = self._append(self._source_samples, source_samples)
        if target_samples is not None:
            self.target_model.update(target_samples)
            if self.store_target:
                self._target_samples = self._append(self._target_samples, target_samples)

    def fit_models(self):
        src = tgt = None
        if self.store_source:
            self._source_samples = torch.cat(self.gather(self._source_samples), dim=-2)
            src = self._source_samples
        if self.store_target:
            self._target_samples = torch.cat(self.gather(self._target_samples), dim=-2)
            tgt = self._target_samples
        self.source_model.fit(src)
        self.target_model.fit(tgt)

    @abstractmethod
    def compute(self) -> Tensor:
        """fit both models, build the transport operators, return the source-target distance"""

    @abstractmethod
    def transport(self, inputs: Tensor) -> Tensor:
        """[*leading_shape, (B,) dim] -> transported samples of the same shape"""

    def forward(self, inputs: Tensor) -> Tensor:
        return self.transport(inputs)

    def extra_repr(self) -> str:
        return (f"leading_dim={tuple(self.leading_shape)}, dim={self.dim}, reset_source={self.reset_source}, "
                f"reset_target={self.reset_target}, store_source={self.store_source}, store_target={self.store_target}")
