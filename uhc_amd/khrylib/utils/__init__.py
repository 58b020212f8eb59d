from .logger import create_logger  # noqa: F401
from .memory import Memory  # noqa: F401
from .zfilter import RunningStat, ZFilter  # noqa: F401
from .torch import *  # noqa: F401,F403
