"""ot_vae_lightning_amd -- the training-step hot path of theoad/ot-vae-lightning, rebuilt for AMD Instinct MI355X
(gfx950): hand-written HIP kernels behind a C ABI (``include/otvae.h``, ``libotvae_hip.so``) and the reference's own
plug-in API on top (``networks.CNN``, ``prior.GaussianPrior``, ``model.VAE``, ``ot.sinkhorn_log``,
``ot.GaussianTransport`` ...).  There is no CPU execution path: see DESIGN.md."""
from . import _lib  # noqa: F401
from .networks import *  # noqa: F401,F403
from .prior import *  # noqa: F401,F403
from .model import *  # noqa: F401,F403
from .ot import *  # noqa: F401,F403
from .engine import *  # noqa: F401,F403

__version__ = "0.1.0"
