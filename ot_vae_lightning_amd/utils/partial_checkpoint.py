"""``PartialCheckpoint``: load the weights of ONE attribute of a model out of a checkpoint of a bigger one -- e.g. the
encoder of a VAE trained with the reference into ``model.encoder`` here (reference utils/partial_checkpoint.py:24-81,
used by ``VisionModule.setup``, model/base.py:192-195).

The state-dict keys, shapes and values of this package's modules are those of the reference (convolution weights keep
the OIHW *shape* on HWIO *memory*; ``load_state_dict`` copies element-wise), so a checkpoint written by the reference's
Lightning run loads unchanged.  Accepted files: a Lightning checkpoint (``{'state_dict': ...}``) or a bare state dict.

Key selection: with ``attr_name='encoder'`` only keys whose leading dotted components equal ``encoder`` are taken and
that prefix (with its dot) is replaced by ``replace_str`` (first occurrence); when no key of the file mentions
``attr_name`` at all, the file is taken to hold that attribute's state dict already and is used whole.

Files are read with ``torch.load(weights_only=True)``: a checkpoint that carries pickled objects (Lightning stores the
inference transforms, hyper-parameters, callbacks) needs ``trusted=True``, which un-pickles arbitrary code -- only for
files you wrote yourself.
"""
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch import Tensor, nn

__all__ = ["PartialCheckpoint", "human_format"]

_UNITS = ("", "K", "M", "B", "T")


def human_format(num: float) -> str:
    """1234567 -> '1.23M' (three significant digits, thousands units)"""
    value, unit = float(f"{num:.3g}"), 0
    while abs(value) >= 1000 and unit < len(_UNITS) - 1:
        value, unit = value / 1000.0, unit + 1
    return f"{value:f}".rstrip("0").rstrip(".") + _UNITS[unit]


def _resolve(module: nn.Module, dotted: str) -> nn.Module:
    for name in dotted.split("."):
        module = getattr(module, name)
    return module


class PartialCheckpoint:
    def __init__(self, checkpoint_path: str, attr_name: Optional[str] = None, replace_str: str = "", strict: bool = True,
                 freeze: bool = False, trusted: bool = False):
        if not os.path.exists(checkpoint_path):
            raise FileNotFoundError(f"checkpoint {checkpoint_path} not found")
        self.checkpoint_path, self.attr_name, self.replace_str = checkpoint_path, attr_name, replace_str
        self.strict, self.freeze, self.trusted = strict, freeze, trusted

    def _read(self) -> Dict[str, Tensor]:
        blob = torch.load(self.checkpoint_path, map_location="cpu", weights_only=not self.trusted)
        return blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob

    @property
    def state_dict(self) -> Dict[str, Tensor]:
        full = self._read()
        name = self.attr_name
        if name is None or not any(name in key for key in full):
            return full
        depth = name.count(".") + 1
        picked = OrderedDict()
        for key, value in full.items():
            if ".".join(key.split(".")[:depth]) == name:
                picked[key.replace(name + ".", self.replace_str, 1)] = value
        return picked

    def load_attribute(self, module: nn.Module, attr: str, verbose: bool = True) -> nn.Module:
        target = _resolve(module, attr)
        target.load_state_dict(self.state_dict, strict=self.strict)
        if self.freeze:
            target.requires_grad_(False)
            target.eval()
        if verbose:
            n_params = sum(p.numel() for p in target.parameters())
            n_bytes = sum(t.numel() * t.element_size() for t in list(target.parameters()) + list(target.buffers()))
            print(f"[info]: self.{attr} [{human_format(n_params)} parameters - {n_bytes / 2 ** 20:.0f}MiB] loaded"
                  f"{' and frozen' if self.freeze else ''}")
        return target
