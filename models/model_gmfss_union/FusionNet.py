from drba_amd.models.model_gmfss_union.FusionNet import GridNet  # noqa: F401
