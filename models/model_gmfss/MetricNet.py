from drba_amd.models.model_gmfss.MetricNet import MetricNet, backwarp  # noqa: F401
