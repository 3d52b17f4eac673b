"""FeatureNet on the HIP library (reference models/model_gmfss_union/FeatureNet.py:6-33, same state-dict keys).
Three (PReLU, conv3x3 s2, PReLU, conv3x3) stages -> 64 @ 1/2, 128 @ 1/4, 192 @ 1/8.  Each PReLU (one shared slope)
is fused into the loader of the convolution that follows it."""
from drba_amd import ops as _ops


class FeatureNet:
    def __init__(self, sd, device):
        self.stages = []
        for b in (1, 2, 3):
            p = f"block{b}."
            a = _ops.Conv3x3(sd[p + "1.weight"], sd[p + "1.bias"], stride=2, act=None, device=device,
                             pre_slope=float(sd[p + "0.weight"]))
            c = _ops.Conv3x3(sd[p + "3.weight"], sd[p + "3.bias"], stride=1, act=None, device=device,
                             pre_slope=float(sd[p + "2.weight"]))
            self.stages.append((a, c))

    def __call__(self, x):
        outs = []
        for a, c in self.stages:
            x = c(a(x))
            outs.append(x)
        return outs

    forward = __call__
