from drba_amd.models.rife_426_heavy.warplayer import warp  # noqa: F401
