"""Oracle RIFE wrapper — restates models/rife.py in fp32 (decorators' autocast NOT imitated,
SURVEY.md 0.4).  (test infra)"""
import torch

from . import ifnet as _ifnet
from .drm import calc_drm_rife
from .ops import softsplat


class RifeOracle:
    """Same call surface as the reference RIFE class (models/rife.py:15-109)."""

    def __init__(self, state_dict, scale=1.0):
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.scale = scale
        self.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]  # rife.py:22
        self.pad_size = 64  # rife.py:23

    @torch.no_grad()
    def encode(self, img):
        return _ifnet.head(self.sd, img[:, :3])

    @torch.no_grad()
    def inference_ts(self, I0, I1, ts):
        """models/rife.py:25-39: t==0 / t==1 return the input tensor object itself."""
        out = []
        for t in ts:
            if t == 0:
                out.append(I0)
            elif t == 1:
                out.append(I1)
            else:
                out.append(_ifnet.ifnet(self.sd, torch.cat((I0, I1), 1), timestep=t, scale_list=self.scale_list)[0])
        return out

    @torch.no_grad()
    def calc_flow(self, a, b, f0=None, f1=None):
        """models/rife.py:41-75: block0-only bidirectional flow at t=0.5, reversed by an 'avg'
        forward splat, holes (ones-splat < 0.999) set to max(H, W), then x2."""
        tmap = (a[:, :1].clone() * 0 + 1) * 0.5
        f0 = _ifnet.head(self.sd, a[:, :3]) if f0 is None else f0
        f1 = _ifnet.head(self.sd, b[:, :3]) if f1 is None else f1
        xin = torch.cat((a[:, :3], b[:, :3], f0, f1, tmap), 1)
        flow, _, _ = _ifnet.ifblock(self.sd, "block0.", xin, None, self.scale_list[0])
        flow50, flow51 = flow[:, :2], flow[:, 2:]
        flow05 = -1 * softsplat(flow50, flow50, None, "avg")
        flow15 = -1 * softsplat(flow51, flow51, None, "avg")
        ones = flow05.clone() * 0 + 1
        gap05 = softsplat(ones, flow50, None, "avg") < 0.999
        gap15 = softsplat(ones, flow51, None, "avg") < 0.999
        fill = ones * max(flow05.shape[2], flow05.shape[3])
        flow05 = torch.where(gap05, fill, flow05)
        flow15 = torch.where(gap15, fill, flow15)
        return flow05 * 2, flow15 * 2, f0, f1

    @torch.no_grad()
    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False, trace=None):
        """models/rife.py:77-109."""
        flow10, flow01, f1, f0 = self.calc_flow(I1, I0) if not reuse else reuse
        if reuse is None:
            flow12, flow21, f1, f2 = self.calc_flow(I1, I2)
        else:
            flow12, flow21, f1, f2 = self.calc_flow(I1, I2, f0=reuse[2])
        if trace is not None:
            trace.update(flow10=flow10, flow12=flow12, flow21=flow21, f0=f0, f1=f1, f2=f2)
        out = []
        for k, t in enumerate(ts):
            if t == 0:
                out.append(I0)
            elif t == 1:
                out.append(I1)
            elif t == 2:
                out.append(I2)
            elif 0 < t < 1:
                drm = calc_drm_rife(1 - t, flow10, flow12, linear)
                if trace is not None:
                    trace[f"drm{k}"] = drm["drm_t1_t01"]
                out.append(_ifnet.ifnet(self.sd, torch.cat((I1, I0), 1), timestep=drm["drm_t1_t01"],
                                        scale_list=self.scale_list, f0=f1, f1=f0)[0])
            elif 1 < t < 2:
                drm = calc_drm_rife(t - 1, flow10, flow12, linear)
                if trace is not None:
                    trace[f"drm{k}"] = drm["drm_t1_t12"]
                out.append(_ifnet.ifnet(self.sd, torch.cat((I1, I2), 1), timestep=drm["drm_t1_t12"],
                                        scale_list=self.scale_list, f0=f1, f1=f2)[0])
        return out, (flow21, flow12, f2, f1)
