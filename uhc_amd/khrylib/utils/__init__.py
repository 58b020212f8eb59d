from .zfilter import RunningStat, ZFilter  # noqa: F401
from .torch import *  # noqa: F401,F403
