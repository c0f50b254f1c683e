"""``VisionModule``: the thin slice of the reference's LightningModule base (model/base.py:36-292) that the training
step touches.  Derives from ``pytorch_lightning.LightningModule`` when Lightning is importable (so the class drops
into a Lightning ``Trainer`` exactly like the reference); otherwise from a minimal ``nn.Module`` stand-in exposing
the attributes the step uses (``global_step``, ``device``, ``log_dict``, ``hparams``)."""
import functools
import types
from typing import Any, Callable, Dict, Optional

import torch
import torch.nn as nn

try:  # pragma: no cover - Lightning is not installed in the build/GPU images
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = types.SimpleNamespace()
            self.global_step = 0
            self.logged: Dict[str, Any] = {}

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log_dict(self, logs, **kwargs):
            self.logged = dict(logs)

        def save_hyperparameters(self, *args, ignore=(), **kwargs):
            pass

__all__ = ["VisionModule", "HAVE_LIGHTNING"]


class VisionModule(_Base):
    def __init__(self, metrics=None, monitor: str = "accuracy", mode: str = "min", checkpoints=None,
                 metric_on_train: bool = False, inference_preprocess: Optional[Callable] = None,
                 inference_postprocess: Optional[Callable] = None, ema_decay: Optional[float] = None):
        super().__init__()
        self.checkpoints = checkpoints
        self.loss = ...
        self.val_metrics = metrics.clone(prefix="val/metrics/") if metrics is not None else None
        self.test_metrics = metrics.clone(prefix="test/metrics/") if metrics is not None else None
        self.train_metrics = metrics.clone(prefix="train/metrics/") if metrics is not None and metric_on_train else None
        self.monitor = ("val/metrics/" if self.val_metrics is None else self.val_metrics.prefix) + monitor
        self.mode = mode
        self.inference_preprocess = inference_preprocess
        self.inference_postprocess = inference_postprocess
        self._inference_flag = False
        # the reference's parameter moving average (model/base.py:99,146-190): created at `on_fit_start` (or by engine.HipTrainer, which
        # folds the update into its optimizer kernel), swapped in around evaluation by the epoch hooks below (engine/ema.py)
        self.ema_decay = ema_decay
        self._ema = None

    def optim_parameters(self):
        return (p for p in self.parameters() if p.requires_grad)

    # ---- parameter moving average: the reference's hooks (model/base.py:146-190), same names, same order of operations
    def on_fit_start(self) -> None:
        if self.ema_decay is not None and self._ema is None:
            from ..engine.ema import ParamEMA
            self._ema = ParamEMA(self.optim_parameters(), decay=self.ema_decay)

    def on_before_zero_grad(self, optimizer=None) -> None:
        if self._ema is not None:
            self._ema.update(self.optim_parameters())

    def _ema_swap_in(self) -> None:
        if self.inference_preprocess is not None and self.inference_postprocess is not None:
            self.inference = True   # (the reference fills in default transforms at on_fit_start; without any, the flag stays off)
        if self._ema is not None:
            self._ema.store()
            self._ema.copy_to()

    def _ema_swap_out(self, *args) -> None:
        if self._ema is not None:
            self._ema.restore()

    def on_train_epoch_start(self) -> None:
        self.inference = False

    on_validation_epoch_start = on_test_epoch_start = on_predict_epoch_start = _ema_swap_in
    on_validation_epoch_end = on_test_epoch_end = on_predict_epoch_end = _ema_swap_out

    def training_step(self, batch, batch_idx, optimizer_idx=0):
        loss_fn = self.loss[optimizer_idx] if hasattr(self.loss, "__getitem__") else self.loss
        loss, logs, pbatch = loss_fn(self.batch_preprocess(batch), batch_idx)
        if self.train_metrics is not None:
            logs = {**logs, **self.train_metrics(pbatch["preds"], pbatch["target"])}
        self.log_dict(logs, rank_zero_only=True, prog_bar=True, logger=True, sync_dist=False)
        return {"loss": loss, **logs, **pbatch}

    # ---- checkpoints
    _CARRIED_IN_CHECKPOINT = ("inference_preprocess", "inference_postprocess")

    def setup(self, stage=None):
        """load every ``{attribute: PartialCheckpoint}`` given at construction (Lightning calls this before fit/test;
        a host loop calls it once itself)"""
        for attr, partial in (self.checkpoints or {}).items():
            partial.load_attribute(self, attr)

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        checkpoint.update({k: getattr(self, k) for k in self._CARRIED_IN_CHECKPOINT if getattr(self, k) is not None})

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        for k in self._CARRIED_IN_CHECKPOINT:
            if k in checkpoint:
                setattr(self, k, checkpoint[k])

    def export_checkpoint(self, path: str) -> None:
        """what a Lightning ``ModelCheckpoint`` would write for this module, reduced to what another process needs to
        restore it: ``{'state_dict', 'global_step'}`` plus the inference transforms.  The keys are the reference's, so
        the file loads into the reference's modules as well (and through ``PartialCheckpoint`` into parts of them)."""
        blob = {"state_dict": {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                "global_step": int(self.global_step)}
        self.on_save_checkpoint(blob)
        torch.save(blob, path)

    @property
    def inference(self):
        return self._inference_flag

    @inference.setter
    def inference(self, flag: bool):
        if flag:
            assert self.inference_preprocess is not None, "inference mode needs self.inference_preprocess"
            assert self.inference_postprocess is not None, "inference mode needs self.inference_postprocess"
        self._inference_flag = flag

    @staticmethod
    def preprocess(method):
        @functools.wraps(method)
        def wrapper(self, samples, *args, no_preprocess_override=False, **kwargs):
            if self.inference and not no_preprocess_override:
                samples = self.inference_preprocess(samples)
            return method(self, samples, *args, **kwargs)
        return wrapper

    @staticmethod
    def postprocess(method):
        @functools.wraps(method)
        def wrapper(self, *args, no_postprocess_override=False, **kwargs):
            out = method(self, *args, **kwargs)
            if self.inference and not no_postprocess_override:
                out = [self.inference_postprocess(o) for o in out] if isinstance(out, list) \
                    else self.inference_postprocess(out)
            return out
        return wrapper
