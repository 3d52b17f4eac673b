from drba_amd.models.drm import *  # noqa: F401,F403
from drba_amd.models.drm import calc_drm_gmfss, calc_drm_rife, calc_drm_rife_auxiliary, get_drm_t  # noqa: F401
