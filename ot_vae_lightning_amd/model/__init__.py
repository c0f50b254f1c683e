from .base import *  # noqa: F401,F403
from .vae import *  # noqa: F401,F403
