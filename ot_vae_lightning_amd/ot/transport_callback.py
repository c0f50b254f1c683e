"""``LatentTransport``: the glue that decides which latents reach the transport operator, and when (reference
ot/transport_callback.py:56-287).  The reference class is a Lightning ``Callback``; this one keeps its constructor, its
hook names and their argument order, but depends on nothing from Lightning: ``trainer`` may be ``None`` and ``pl_module``
is any object with ``encode(x, **kw)`` / ``decode(z, **kw)`` (``VAE`` has them), optionally ``log(name, value)``.  A host
loop (``engine.HipTrainer`` feeds the train-time statistics itself, inside the captured step) calls the hooks at the
same points Lightning would.  Everything numeric happens in the ``TransportOperator`` (HIP kernels behind
``GaussianModel.update`` / ``GaussianTransport.compute`` / ``apply_transport``); this file only routes tensors.

Routing rules (transport_callback.py:173-237), with ``unpaired`` = source and target must come from different batches:

    train batch end   target <- outputs[latents_key]           if target_latents_from_train and (paired or no source-from-train or even batch)
                      (no `latents_key` in the output: target <- encode(samples) only on batch 0 of a `verbose` callback, as the reference)
                      source <- encode(transform(samples))     if source_latents_from_train and (paired or no target-from-train or odd batch)
    val batch end     target <- outputs[latents_key] / encode  if not target_latents_from_train and (paired or source-from-train or even batch)
                      source <- encode(transform(samples))     if not source_latents_from_train and (paired or target-from-train or odd batch)
    val epoch start   operator.reset();    val epoch end   cost = operator.compute().mean()
"""
from __future__ import annotations

import re
import warnings
from typing import Any, Callable, Dict, Optional, Sequence, Tuple, Type

import numpy as np
import torch
from torch import Tensor

from .. import utils
from .transport.base import TransportOperator

__all__ = ["LatentTransport", "ConditionalLatentTransport"]


def _snake(name: str) -> str:
    return re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


class LatentTransport:
    def __init__(self, size: Sequence[int], transport_dims: Sequence[int], transport_operator: Type[TransportOperator],
                 transformations: Callable[[Tensor], Tensor], *, common_operator: bool = False, samples_key: str = "samples",
                 latents_key: str = "latents", logging_prefix: Optional[str] = None, target_latents_from_train: bool = False,
                 source_latents_from_train: bool = False, unpaired: bool = True, num_samples_to_log: int = 8,
                 verbose: bool = False, class_idx: Optional[int] = None, conditional_key: str = "y",
                 **transport_operator_kwargs) -> None:
        if (source_latents_from_train or target_latents_from_train) and verbose:
            warnings.warn("latents of TRAINING samples will feed the transport operator: source and target are then not "
                          "unseen data, which biases the transport experiment")
        # [B, C, H, W] -> [H*W, B, C] for per-position operators; [B*H*W, C] for one common operator
        self.layout = utils.VectorLayout(size, transport_dims, "transport_dims", batch_first=common_operator,
                                         flatten_batch=common_operator)
        self.size, self.transport_dims, self.batch_dims = self.layout.size, self.layout.vector_dims, self.layout.position_dims
        self.batch_shape, self.event_shape, self.dim = self.layout.batch_shape, self.layout.event_shape, self.layout.dim
        self.transformations, self.common_operator = transformations, common_operator
        op_size = (self.dim,) if common_operator else (*self.batch_shape, self.dim)
        self.transport_operator = transport_operator(*op_size, **transport_operator_kwargs)
        self.unpaired = unpaired
        self.source_latents_from_train, self.target_latents_from_train = source_latents_from_train, target_latents_from_train
        self.samples_key, self.latents_key = samples_key, latents_key
        kind = _snake(transport_operator.__name__)
        kind = kind[:-len("_transport")] if kind.endswith("_transport") else kind
        self.logging_prefix = f"transport/{kind}/{logging_prefix}/"
        self.num_samples_to_log, self.verbose = num_samples_to_log, verbose
        self.class_idx, self.conditional_key = class_idx, conditional_key
        self.test_metrics = None            # optional callable(pred, target) with .compute() / .reset()
        self.logged: Dict[str, Tensor] = {}  # what the hooks would have logged (also sent to pl_module.log if it exists)

    # ---- layout
    def _permute_and_flatten(self, latents: Tensor) -> Tensor:
        return self.layout.split(latents)

    def _unflatten_and_unpermute(self, flat: Tensor) -> Tensor:
        return self.layout.join(flat)

    # ---- what the hooks are made of
    def update_transport_operator(self, latents: Tensor, source: bool) -> None:
        flat = self._permute_and_flatten(latents)
        if source:
            self.transport_operator.update(source_samples=flat)
        else:
            self.transport_operator.update(target_samples=flat)

    def transport(self, latents: Tensor) -> Tensor:
        return self._unflatten_and_unpermute(self.transport_operator(self._permute_and_flatten(latents)))

    def sample(self, batch_size: int, from_dist: str = "source") -> Tensor:
        n = batch_size * int(np.prod(self.batch_shape)) if self.common_operator else batch_size
        if from_dist == "source":
            dist = self.transport_operator.source_distribution
        elif from_dist == "target":
            dist = self.transport_operator.target_distribution
        else:
            raise NotImplementedError(f"cannot sample from `{from_dist}`: 'source' or 'target'")
        return self._unflatten_and_unpermute(dist.sample((n,)))

    def _get_samples(self, pl_module, outputs) -> Tuple[Tensor, Dict[str, Any]]:
        if not isinstance(outputs, dict):
            raise ValueError(f"the step must return a dict, got {type(outputs)}")
        if self.samples_key not in outputs:
            raise ValueError(f"neither '{self.latents_key}' nor '{self.samples_key}' in the step's output dict: nothing to "
                             f"encode")
        device = getattr(pl_module, "device", None)
        samples = outputs[self.samples_key]
        if device is not None:
            samples = samples.to(device)
        kwargs = dict(outputs.get("kwargs") or {})
        if self.class_idx is None:
            return samples, kwargs
        if self.conditional_key in outputs:
            condition = outputs[self.conditional_key]
        elif self.conditional_key in kwargs:
            condition = kwargs[self.conditional_key]
        else:
            raise ValueError(f"`class_idx` given but no '{self.conditional_key}' in the step's outputs or its 'kwargs'")
        keep = condition.to(samples.device) == self.class_idx
        kwargs[self.conditional_key] = condition[keep.to(condition.device)]
        return samples[keep], kwargs

    @staticmethod
    def _encode(pl_module, image: Tensor, **kwargs) -> Tensor:
        if not hasattr(pl_module, "encode"):
            raise NotImplementedError("the module must implement `encode(image) -> latents`")
        return pl_module.encode(image, **kwargs)

    @staticmethod
    def _decode(pl_module, latents: Tensor, **kwargs) -> Tensor:
        if not hasattr(pl_module, "decode"):
            raise NotImplementedError("the module must implement `decode(latents) -> image`")
        return pl_module.decode(latents, **kwargs)

    def _eval_encode(self, pl_module, samples: Tensor, **kwargs) -> Tensor:
        was_training = getattr(pl_module, "training", False)
        if was_training:
            pl_module.eval()
        try:
            return self._encode(pl_module, samples, **kwargs).detach()
        finally:
            if was_training:
                pl_module.train()

    def _log(self, pl_module, name: str, value: Tensor) -> None:
        self.logged[name] = value.detach()
        if hasattr(pl_module, "log"):
            pl_module.log(name, value, sync_dist=True)

    # ---- hooks (same names and argument order as the Lightning callback)
    def on_fit_start(self, trainer, pl_module) -> None:
        metrics = getattr(pl_module, "test_metrics", None)
        self.test_metrics = metrics.clone(prefix=self.logging_prefix) if hasattr(metrics, "clone") else metrics
        device = getattr(pl_module, "device", None)
        if device is not None:
            self.transport_operator = self.transport_operator.to(device)

    @torch.no_grad()
    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int, unused: int = 0) -> None:
        src, tgt = self.source_latents_from_train, self.target_latents_from_train
        if not (src or tgt):
            return
        if tgt and (not self.unpaired or not src or batch_idx % 2 == 0):
            if self.latents_key in outputs:
                self.update_transport_operator(outputs[self.latents_key].detach(), source=False)
            elif self.verbose and batch_idx == 0:
                # The reference encodes the samples itself when the step output carries no latents -- but that fallback sits INSIDE
                # its `if self.verbose and batch_idx == 0:` warning block (transport_callback.py:189-203): without `verbose`, or on
                # any later batch, a step output without `latents_key` feeds nothing to the target.  Kept as is: the statistics a
                # user gets must be the reference's (VAE.training_step always returns the latents, so this path is a corner).
                samples, kwargs = self._get_samples(pl_module, outputs)
                self.update_transport_operator(self._eval_encode(pl_module, samples, **kwargs), source=False)
        if src and (not self.unpaired or not tgt or batch_idx % 2 == 1):
            samples, kwargs = self._get_samples(pl_module, outputs)
            self.update_transport_operator(self._eval_encode(pl_module, self.transformations(samples), **kwargs), source=True)

    @torch.no_grad()
    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int, dataloader_idx: int = 0) -> None:
        src, tgt = self.source_latents_from_train, self.target_latents_from_train
        if not tgt and (not self.unpaired or src or batch_idx % 2 == 0):
            if self.latents_key in outputs:
                self.update_transport_operator(outputs[self.latents_key].detach(), source=False)
            else:
                samples, kwargs = self._get_samples(pl_module, outputs)
                self.update_transport_operator(self._encode(pl_module, samples, **kwargs), source=False)
        if not src and (not self.unpaired or tgt or batch_idx % 2 == 1):
            samples, kwargs = self._get_samples(pl_module, outputs)
            self.update_transport_operator(self._encode(pl_module, self.transformations(samples), **kwargs), source=True)

    def on_validation_epoch_start(self, trainer, pl_module) -> None:
        self.transport_operator.reset()

    @torch.no_grad()
    def on_validation_epoch_end(self, trainer, pl_module) -> None:
        if getattr(trainer, "sanity_checking", False) and (self.source_latents_from_train or self.target_latents_from_train):
            return
        self._log(pl_module, self.logging_prefix + "avg_transport_cost", self.transport_operator.compute().mean())

    def on_test_epoch_start(self, trainer, pl_module) -> None:
        if self.test_metrics is not None:
            self.test_metrics.reset()

    @torch.no_grad()
    def on_test_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int, dataloader_idx: int = 0) -> None:
        if self.test_metrics is None:
            return
        samples, kwargs = self._get_samples(pl_module, outputs)
        latents = self._encode(pl_module, self.transformations(samples), **kwargs)
        self.test_metrics(self._decode(pl_module, self.transport(latents), **kwargs), samples)

    def on_test_epoch_end(self, trainer, pl_module) -> None:
        if self.test_metrics is not None:
            for k, v in dict(self.test_metrics.compute()).items():
                self._log(pl_module, k, torch.as_tensor(v))


class ConditionalLatentTransport:
    """One ``LatentTransport`` per class index, every hook fanned out (transport_callback.py:385-430)."""

    def __init__(self, num_classes: int, logging_prefix: str, num_samples_to_log: int = 10, *args, **kwargs):
        self.num_classes, self.logging_prefix, self.num_samples_to_log = num_classes, logging_prefix, num_samples_to_log
        self.transports = [LatentTransport(*args, **kwargs, logging_prefix=logging_prefix, class_idx=i,
                                           num_samples_to_log=max(1, num_samples_to_log // num_classes))
                           for i in range(num_classes)]

    def __getattr__(self, name: str):
        if name.startswith("on_"):
            def fan_out(*args, **kwargs):
                for t in self.transports:
                    getattr(t, name)(*args, **kwargs)
            return fan_out
        raise AttributeError(name)
