"""
TEST INFRASTRUCTURE -- build-container only.  Never imported by the product path.

Makes the *real* reference (``/root/reference/ot_vae_lightning``) importable in the build
container so that ``oracle/gen_golden.py`` can record golden input/output vectors from it.
The reference itself never travels to the GPU box; only the numbers it produced do
(``tests/golden/*.npz``).

The reference's hot-path modules are plain torch, but the package ``__init__`` star-imports
datamodules / Lightning / torchvision / wandb, none of which are installed here (SURVEY.md
section 8c).  This file therefore
  1. pre-registers *empty namespace packages* for ``ot_vae_lightning``, ``.data``, ``.model``
     whose ``__path__`` points into /root/reference, so their ``__init__`` never executes, and
  2. registers inert stand-in modules for the absent third-party packages.  Only the handful
     of helpers the hot path really calls get a real (tiny) implementation.
It contains no reference code.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("OTVAE_REFERENCE_ROOT", "/root/reference")


class _Anything:
    """Inert class: any attribute is another inert class, calling returns an instance."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []  # behave like a package so sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _apply_to_collection(data, dtype, function, *args, **kwargs):
    if isinstance(data, dtype):
        return function(data, *args, **kwargs)
    if isinstance(data, dict):
        return {k: _apply_to_collection(v, dtype, function, *args, **kwargs) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return type(data)(_apply_to_collection(v, dtype, function, *args, **kwargs) for v in data)
    return data


class _LightningModule(nn.Module):
    """Just enough of pl.LightningModule for VAE.__init__/nelbo to run."""

    def __init__(self, *a, **k):
        super().__init__()
        self.hparams = types.SimpleNamespace()
        self.global_step = 0
        self.trainer = None

    def save_hyperparameters(self, *a, ignore=(), **k):
        import inspect
        frame = inspect.currentframe().f_back
        for key, val in frame.f_locals.items():
            if key in ("self", "__class__") or key in ignore or key.startswith("_"):
                continue
            if isinstance(val, (bool, int, float, str, type(None))):
                setattr(self.hparams, key, val)

    @property
    def device(self):
        return next(self.parameters()).device

    def log_dict(self, *a, **k):
        pass

    def print(self, *a, **k):
        pass


class _Callback:
    pass


class _MetricCollection(_Anything):
    prefix = ""

    def clone(self, prefix=""):
        c = _MetricCollection()
        c.prefix = prefix
        return c


def install():
    """Idempotent.  After this, ``import ot_vae_lightning.networks.cnn`` etc. work."""
    if "ot_vae_lightning" in sys.modules and getattr(sys.modules["ot_vae_lightning"], "_otvae_shim", False):
        return
    sys.dont_write_bytecode = True
    if not os.path.isdir(os.path.join(REF_ROOT, "ot_vae_lightning")):
        raise RuntimeError(f"reference not found under {REF_ROOT}; goldens can only be generated in the build container")

    # -- third-party stand-ins --------------------------------------------------------------
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms", ToTensor=type("ToTensor", (object,), {"__call__": lambda self, x: x}))
    tv.transforms = tvt
    _stub("torchvision.transforms.functional")
    _stub("torchvision.utils")
    _stub("torchvision.datasets")
    _stub("wandb")
    _stub("torch_ema")
    _stub("lovely_tensors")
    tm = _stub("torchmetrics", MetricCollection=_MetricCollection)
    _stub("torchmetrics.metric")
    _stub("torchmetrics.image")
    _stub("torchmetrics.image.fid")
    _stub("torchmetrics.image.psnr")
    _stub("torchmetrics.utilities")
    _stub("torchmetrics.utilities.data")
    _stub("jsonargparse")

    rank_zero = lambda *a, **k: None
    pl = _stub("pytorch_lightning", LightningModule=_LightningModule, Callback=_Callback,
               seed_everything=lambda s, **k: torch.manual_seed(s))
    plu = _stub("pytorch_lightning.utilities", rank_zero_info=rank_zero, rank_zero_warn=rank_zero,
                rank_zero_only=lambda f: f,
                move_data_to_device=lambda b, d: _apply_to_collection(b, torch.Tensor, lambda t: t.to(d)))
    pl.utilities = plu
    _stub("pytorch_lightning.utilities.apply_func", apply_to_collection=_apply_to_collection)
    _stub("pytorch_lightning.utilities.distributed",
          sync_ddp_if_available=lambda t, *a, **k: t,
          gather_all_tensors=lambda t, *a, **k: [t],
          distributed_available=lambda: False)
    _stub("pytorch_lightning.utilities.types")
    _stub("pytorch_lightning.utilities.memory")
    _stub("pytorch_lightning.utilities.cli")
    _stub("pytorch_lightning.loggers")
    _stub("pytorch_lightning.loggers.wandb")
    _stub("pytorch_lightning.callbacks", Callback=_Callback)
    _stub("pytorch_lightning.cli")

    # -- namespace packages whose star-importing __init__ must not run ------------------------
    for pkg in ("ot_vae_lightning", "ot_vae_lightning.data", "ot_vae_lightning.model",
                "ot_vae_lightning.metrics"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF_ROOT, *pkg.split("."))]
        m._otvae_shim = True
        sys.modules[pkg] = m
    # data.progressive_callback is imported by model/vae.py for a decorator: identity stand-in
    _stub("ot_vae_lightning.data.progressive_callback",
          transform_batch_tv=lambda *a, **k: (lambda f: f))
    sys.modules["ot_vae_lightning.data"].TorchvisionDatamodule = _Anything


def ref(module: str):
    """``ref('networks.cnn')`` -> the reference module object."""
    install()
    return importlib.import_module("ot_vae_lightning." + module)
