from .matrix_utils import *  # noqa: F401,F403
from .w2_utils import *  # noqa: F401,F403
from .codebook import *  # noqa: F401,F403
from .transport import *  # noqa: F401,F403
from .distribution_models import *  # noqa: F401,F403
from .transport_callback import *  # noqa: F401,F403
