from drba_amd.models.model_gmfss.GMFSS import Model  # noqa: F401
