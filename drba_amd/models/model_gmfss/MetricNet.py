"""MetricNet of the non-union GMFSS (reference models/model_gmfss/MetricNet.py:23-44: metric_out = PReLU, Conv2d(64, 2) --
no Tanh()*10 as model_gmfss_union/MetricNet.py:40-44 has); same kernels, other default."""
from drba_amd.models.model_gmfss_union.MetricNet import MetricNet as _MetricNet
from drba_amd.models.model_gmfss_union.MetricNet import backwarp  # noqa: F401


class MetricNet(_MetricNet):
    def __init__(self, sd, device, tanh10=False):
        super().__init__(sd, device, tanh10=tanh10)
