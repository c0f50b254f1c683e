"""Exponential moving average of the trainable parameters: the reference's ``ema_decay`` option (model/base.py:99,146-190).

The reference keeps the average through the third-party ``torch_ema`` package (requirements.txt:10, unpinned; not vendored under
/root/reference) inside Lightning's hooks: ``ExponentialMovingAverage(optim_parameters, decay)`` at ``on_fit_start``, ``update`` in
``on_before_zero_grad`` (after every optimizer step), ``store(); copy_to()`` at the start and ``restore()`` at the end of every
validation / test / predict epoch.  This class restates that package's published arithmetic (torch_ema 0.3,
``use_num_updates=True``):

    num_updates += 1;  d = min(decay, (1 + num_updates) / (10 + num_updates));  shadow -= (1 - d) * (shadow - param)

on the MI355X: with ``engine.HipTrainer`` the update is part of the optimizer kernel's own pass over the flat parameter buffer
(``otvae_adam_step_ema``: no launch of its own, one update per accepted step, so ``num_updates`` is the device step counter);
on the host-driven route (the reference's loop with a stock optimizer) ``update()`` is one launch per parameter tensor -- or one in
all when the parameters are views of one flat buffer (``GraphedNelbo``'s engine).  ``store / copy_to / restore`` are plain copies.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["ParamEMA"]


class ParamEMA:
    """``torch_ema.ExponentialMovingAverage`` for fp32 GPU parameters.  ``flat``: the buffer all ``params`` are views of (then the
    shadow is one flat tensor and every operation one launch / copy)."""

    def __init__(self, params, decay: float, flat: Optional[Tensor] = None, in_optimizer: bool = False,
                 step_tensor: Optional[Tensor] = None):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")   # torch_ema's own check and text
        self.decay = float(decay)
        self.params: List[Tensor] = list(params)
        self.flat = flat
        self.in_optimizer = in_optimizer   # HipTrainer: the optimizer kernel performs `update` itself
        self.num_updates = 0               # host-driven route only (the engine's count is its device step counter)
        self._step_tensor = step_tensor
        with torch.no_grad():
            self.shadow = flat.detach().clone() if flat is not None else [p.detach().clone() for p in self.params]
        self.collected = None

    # -- torch_ema's API -----------------------------------------------------------------------------------------
    def update(self, parameters=None) -> None:
        if self.in_optimizer:
            return  # (the engine's optimizer kernel has already done it; a loop that also calls the hook must not double it)
        lib = _lib.load()
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        if self.flat is not None:
            check(lib.otvae_ema_update(ptr(self.shadow), ptr(self.flat), self.flat.numel(), d, stream()), "otvae_ema_update")
            return
        for i, p in enumerate(self.params):
            if not _dense_like(self.shadow[i], p):
                # an engine moved the parameter into its flat buffer since (conv weights change to HWIO memory): same values, new layout
                with torch.no_grad():
                    self.shadow[i] = torch.empty_like(p.detach()).copy_(self.shadow[i])
        for s_, p in zip(self.shadow, self.params):
            if not p.is_cuda or p.dtype != torch.float32:
                raise TypeError("ParamEMA: fp32 GPU parameters only (there is no CPU path)")
            pd = p.detach()
            check(lib.otvae_ema_update(ptr(s_), ptr(pd), pd.numel(), d, stream()), "otvae_ema_update")

    def store(self, parameters=None) -> None:
        with torch.no_grad():
            self.collected = self.flat.detach().clone() if self.flat is not None else [p.detach().clone() for p in self.params]

    def copy_to(self, parameters=None) -> None:
        with torch.no_grad():
            if self.flat is not None:
                self.flat.copy_(self.shadow)
            else:
                for s_, p in zip(self.shadow, self.params):
                    p.data.copy_(s_)

    def restore(self, parameters=None) -> None:
        if self.collected is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")   # torch_ema's text
        with torch.no_grad():
            if self.flat is not None:
                self.flat.copy_(self.collected)
            else:
                for c_, p in zip(self.collected, self.params):
                    p.data.copy_(c_)
        self.collected = None

    def state_dict(self) -> dict:
        sh = self.shadow if self.flat is None else [self.shadow]
        n = int(self._step_tensor.item()) if (self.in_optimizer and self._step_tensor is not None) else self.num_updates
        return {"decay": self.decay, "num_updates": n, "shadow_params": [t.clone() for t in sh]}

    def load_state_dict(self, state: dict) -> None:
        self.decay, self.num_updates = float(state["decay"]), int(state["num_updates"])
        sh = self.shadow if self.flat is None else [self.shadow]
        with torch.no_grad():
            for dst, src in zip(sh, state["shadow_params"]):
                dst.copy_(src)


def _dense_like(a: Tensor, b: Tensor) -> bool:
    return a.shape == b.shape and a.stride() == b.stride()
