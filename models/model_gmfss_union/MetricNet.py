from drba_amd.models.model_gmfss_union.MetricNet import MetricNet  # noqa: F401
