from drba_amd.models.gmfss_union import GMFSS_UNION  # noqa: F401
