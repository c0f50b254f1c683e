from .cnn import *  # noqa: F401,F403
from .nets_utils import *  # noqa: F401,F403
from .vit import *  # noqa: F401,F403
