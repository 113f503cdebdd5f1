from .generation import *   # noqa: F401,F403
