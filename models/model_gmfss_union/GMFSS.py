from drba_amd.models.model_gmfss_union.GMFSS import Model  # noqa: F401
