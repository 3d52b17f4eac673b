from drba_amd.models.gmfss import GMFSS  # noqa: F401
