"""RIFE wrapper with the DRBA call surface (reference models/rife.py:15-109), HIP path.

    RIFE(weights, scale, device).inference_ts(I0, I1, ts) -> [frames]
    RIFE(...).inference_ts_drba(I0, I1, I2, ts, reuse=None, linear=False) -> ([frames], reuse)

Everything runs in fp32 (the reference's CPU autocast path is bf16 and deviates ~2e-3 from
its own fp32 evaluation; parity is against the fp32 evaluation).
"""
import os

import torch

from drba_amd import ops as _ops
from drba_amd.models.drm import calc_drm_rife
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
from drba_amd.models.utils.tools import convert


class RIFE:
    def __init__(self, weights="weights/train_log_rife_426_heavy", scale=1.0, device=None):
        device = _ops.default_device() if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("drba_amd RIFE runs on the MI355X HIP path only; there is no CPU fallback")
        if isinstance(weights, dict):  # an already-loaded state dict (no 'module.' prefix)
            sd = weights
        else:
            sd = convert(torch.load(os.path.join(weights, "flownet.pkl"), map_location="cpu"))
        self.device = device
        self.ifnet = IFNet().to(device).eval()
        self.ifnet.load_state_dict(sd, strict=False)
        self.scale = scale
        self.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
        self.pad_size = 64

    def encode(self, img):
        return self.ifnet.encode(img[:, :3])

    def inference_ts(self, I0, I1, ts):
        """t == 0 / t == 1 return the input tensor objects themselves (reference rife.py:30-33)."""
        output = []
        f0 = f1 = None
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            else:
                if f0 is None:  # encode once per call, not once per t (same values)
                    f0, f1 = self.ifnet.encode(I0), self.ifnet.encode(I1)
                output.append(self.ifnet.forward_pair(I0, I1, float(t), self.scale_list, f0, f1)[0])
        return output

    def calc_flow(self, a, b, f0=None, f1=None):
        """Coarse bidirectional flow from block0 at t=0.5, reversed to the frame's own time by a
        forward splat (reference rife.py:41-75); one fused kernel pair per direction."""
        _, _, H, W = a.shape
        f0 = self.ifnet.encode(a) if f0 is None else f0
        f1 = self.ifnet.encode(b) if f1 is None else f1
        s = self.scale_list[0]
        xin = _ops.ifblock_input(a, b, f0, f1, 0.5, None, None, 1.0, s)
        flow = _ops.ifblock_update(self.ifnet.block[0].core(xin), None, H, W, s)
        flow01 = _ops.flow_reverse(flow[:, :2])   # 2 * (-splat_avg(flow50)), holes -> 2*max(H, W)
        flow10 = _ops.flow_reverse(flow[:, 2:])
        return flow01, flow10, f0, f1

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
        flow10, flow01, f1, f0 = self.calc_flow(I1, I0) if not reuse else reuse
        if reuse is None:
            flow12, flow21, f1, f2 = self.calc_flow(I1, I2)
        else:
            flow12, flow21, f1, f2 = self.calc_flow(I1, I2, f0=reuse[2])
        output = []
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            elif t == 2:
                output.append(I2)
            elif 0 < t < 1:
                t = 1 - t
                # only the map this frame consumes is computed (the reference builds both, drm.py:89-96)
                drm = (_ops.drm_rife_linear(flow10, flow12, t, 1e-4) if linear
                       else calc_drm_rife(t, flow10, flow12, False)["drm_t1_t01"])
                output.append(self.ifnet.forward_pair(I1, I0, drm, self.scale_list, f1, f0)[0])
            elif 1 < t < 2:
                t = t - 1
                drm = (_ops.drm_rife_linear(flow12, flow10, t, 1e-4) if linear
                       else calc_drm_rife(t, flow10, flow12, False)["drm_t1_t12"])
                output.append(self.ifnet.forward_pair(I1, I2, drm, self.scale_list, f1, f2)[0])
        return output, (flow21, flow12, f2, f1)
