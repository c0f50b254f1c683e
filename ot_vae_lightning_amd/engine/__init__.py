from .trainer import *  # noqa: F401,F403
from .dp import *  # noqa: F401,F403
from .graphed import *  # noqa: F401,F403
