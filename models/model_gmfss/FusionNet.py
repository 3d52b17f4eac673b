from drba_amd.models.model_gmfss.FusionNet import GridNet  # noqa: F401
