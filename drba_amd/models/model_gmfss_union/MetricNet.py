"""MetricNet on the HIP library (reference models/model_gmfss_union/MetricNet.py:23-65; model_gmfss/MetricNet.py
differs only by the missing Tanh()*10).  The 14-channel input (photometric error after a zeros-padded backwarp,
normalised flows, forward/backward occlusion masks) is one fused kernel; the residual adds, PReLUs and the final
tanh*10 live in the conv loader/epilogue."""
from drba_amd import ops as _ops


def backwarp(tenIn, tenflow):
    """MetricNet.py:10-20: grid_sample(bilinear, padding_mode='zeros', align_corners=True)."""
    return _ops.backwarp(tenIn, tenflow, "zeros")


class MetricNet:
    def __init__(self, sd, device, tanh10=True):
        self.conv_in = _ops.Conv3x3(sd["metric_in.weight"], sd["metric_in.bias"], act=None, device=device)
        self.mid = [_ops.Conv3x3(sd[f"metric_net{k}.1.weight"], sd[f"metric_net{k}.1.bias"], act=None, device=device,
                                 pre_slope=float(sd[f"metric_net{k}.0.weight"])) for k in (1, 2, 3)]
        self.conv_out = _ops.Conv3x3(sd["metric_out.1.weight"], sd["metric_out.1.bias"], act="tanh10" if tanh10 else None,
                                     device=device, pre_slope=float(sd["metric_out.0.weight"]))

    def __call__(self, img0, img1, flow01, flow10):
        feat = self.conv_in(_ops.metric_input(img0, img1, flow01, flow10))
        for c in self.mid:
            feat = c(feat, residual=feat)  # metric_netK(feat) + feat
        m = self.conv_out(feat)
        return m[:, :1], m[:, 1:2]

    forward = __call__
