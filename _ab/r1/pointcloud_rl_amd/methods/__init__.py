from .builder import MFRL, build_agent  # noqa: F401
from .sac import SAC  # noqa: F401
from .drq import DrQ  # noqa: F401
