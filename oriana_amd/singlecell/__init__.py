from .generation import *   # noqa: F401,F403
from .cmatrix import CountMatrix   # noqa: F401
