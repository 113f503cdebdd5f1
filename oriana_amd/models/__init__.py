from .base import *     # noqa: F401,F403
from .gap import *      # noqa: F401,F403
from .zigap import *    # noqa: F401,F403
