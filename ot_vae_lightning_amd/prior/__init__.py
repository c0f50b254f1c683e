from .base import *  # noqa: F401,F403
from .gaussian import *  # noqa: F401,F403
from .sinkhorn import *  # noqa: F401,F403
from .gaussian_w2 import *  # noqa: F401,F403
from .codebook import *  # noqa: F401,F403
from .conditional_gaussian import *  # noqa: F401,F403
