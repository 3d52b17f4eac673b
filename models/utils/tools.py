from drba_amd.models.utils.tools import *  # noqa: F401,F403
