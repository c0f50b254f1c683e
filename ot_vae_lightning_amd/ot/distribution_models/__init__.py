from .base import *  # noqa: F401,F403
from .gaussian_model import *  # noqa: F401,F403
from .codebook_model import *  # noqa: F401,F403
from .gaussian_mixture_model import *  # noqa: F401,F403
