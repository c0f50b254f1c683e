from .base import *  # noqa: F401,F403
from .gaussian_transport import *  # noqa: F401,F403
from .discrete_transport import *  # noqa: F401,F403
from .gmm_transport import *  # noqa: F401,F403
