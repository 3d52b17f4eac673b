from drba_amd.models.rife import *  # noqa: F401,F403
from drba_amd.models.rife import RIFE  # noqa: F401
