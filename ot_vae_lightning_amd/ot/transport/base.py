"""``TransportOperator``: the interface every latent transport implements (reference ot/transport/base.py:28-173).

An operator holds two ``DistributionModel`` s, one per *side* ("source", "target").  Both sides behave identically, so
the bookkeeping below is written once over ``_SIDES`` instead of twice: per side there is the model
(``<side>_model``), whether ``reset()`` clears it (``reset_<side>``), and whether raw samples are kept for models that
are fitted on samples rather than on running statistics (``store_<side>`` -> buffer ``_<side>_samples``, concatenated
along the sample axis -2 and all-gathered across ranks before ``fit``).

    update(source_samples=None, target_samples=None)   feed [*leading_shape, B, dim] samples to either side
    compute() -> distance [*leading_shape]              fit both models and build the operator   (abstract)
    transport(x) / __call__(x)                          push source samples onto the target        (abstract)
    reset()                                             forget what ``reset_<side>`` allows
"""
from abc import ABC, abstractmethod
from typing import Optional

import torch
import torch.distributions as D
import torch.nn as nn
from torch import Tensor

from ... import utils
from ..distribution_models.base import DistributionModel

__all__ = ["TransportOperator"]

_SIDES = ("source", "target")


class TransportOperator(nn.Module, utils.DDPMixin, ABC):
    def __init__(self, *size: int, source_model: DistributionModel, target_model: DistributionModel,
                 reset_source: bool = True, reset_target: bool = True, store_source: bool = False,
                 store_target: bool = False, **ddp_kwargs):
        nn.Module.__init__(self)
        utils.DDPMixin.__init__(self, **ddp_kwargs)
        *leading, self.dim = size
        self.leading_shape = tuple(leading)
        given = dict(source=(source_model, reset_source, store_source), target=(target_model, reset_target, store_target))
        for side, (model, resets, stores) in given.items():
            setattr(self, f"{side}_model", model)
            setattr(self, f"reset_{side}", resets)
            setattr(self, f"store_{side}", stores)
            if stores:
                self.register_buffer(f"_{side}_samples", None)
        if store_source or store_target:
            self.warn(f"`{type(self).__name__}` keeps every sample it is updated with in a buffer: memory grows with the "
                      f"dataset")

    # ---- per-side accessors
    def _model(self, side: str) -> DistributionModel:
        return getattr(self, f"{side}_model")

    def _stored(self, side: str) -> Optional[Tensor]:
        return getattr(self, f"_{side}_samples") if getattr(self, f"store_{side}") else None

    def _feed(self, side: str, samples: Optional[Tensor]) -> None:
        if samples is None:
            return
        self._model(side).update(samples)
        if getattr(self, f"store_{side}"):
            kept, fresh = self._stored(side), samples.detach()
            setattr(self, f"_{side}_samples", fresh if kept is None else torch.cat([kept, fresh.type_as(kept)], dim=-2))

    @property
    def source_distribution(self) -> D.Distribution:
        return self._model("source").distribution

    @property
    def target_distribution(self) -> D.Distribution:
        return self._model("target").distribution

    # ---- life cycle
    def update(self, source_samples: Optional[Tensor] = None, target_samples: Optional[Tensor] = None) -> None:
        self._feed("source", source_samples)
        self._feed("target", target_samples)

    def reset(self) -> None:
        for side in _SIDES:
            if not getattr(self, f"reset_{side}"):
                continue
            if getattr(self, f"store_{side}"):
                setattr(self, f"_{side}_samples", None)
            self._model(side).reset()

    def fit_models(self) -> None:
        """all-gather what was stored (sample axis), then ``fit`` each model on it (``None`` when nothing is stored)"""
        for side in _SIDES:
            if getattr(self, f"store_{side}"):
                setattr(self, f"_{side}_samples", torch.cat(self.gather(self._stored(side)), dim=-2))
            self._model(side).fit(self._stored(side))

    @abstractmethod
    def compute(self) -> Tensor:
        ...

    @abstractmethod
    def transport(self, inputs: Tensor) -> Tensor:
        ...

    def forward(self, inputs: Tensor) -> Tensor:
        return self.transport(inputs)

    def extra_repr(self) -> str:
        flags = ", ".join(f"{kind}_{side}={getattr(self, f'{kind}_{side}')}" for kind in ("reset", "store") for side in _SIDES)
        return f"leading_dim={self.leading_shape}, dim={self.dim}, {flags}"
