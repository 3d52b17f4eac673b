"""Oracle GMFSS / GMFSS_UNION (flow-guided softmax-splatting synthesis), functional over state dicts.  (test infra)

Restates models/model_gmfss_union/{GMFSS,MetricNet,FeatureNet,FusionNet}.py, models/gmfss_union.py and the
non-union variants models/model_gmfss/*, models/gmfss.py in fp32 (autocast not imitated).
"""
import torch
import torch.nn.functional as F

from . import gmflow as _gmflow
from . import ifnet as _ifnet
from .drm import calc_drm_gmfss, calc_drm_rife_auxiliary
from .ops import backwarp_zeros, resize, softsplat


def _prelu(x, a):
    return F.prelu(x, a)


# ----------------------------------------------------------------------------------------- FeatureNet
def featurenet(sd, x):
    """FeatureNet.forward (FeatureNet.py:6-33): three (PReLU, conv s2, PReLU, conv) stages -> 64@1/2, 128@1/4, 192@1/8."""
    outs = []
    for b in (1, 2, 3):
        p = f"block{b}."
        x = F.conv2d(_prelu(x, sd[p + "0.weight"]), sd[p + "1.weight"], sd[p + "1.bias"], stride=2, padding=1)
        x = F.conv2d(_prelu(x, sd[p + "2.weight"]), sd[p + "3.weight"], sd[p + "3.bias"], stride=1, padding=1)
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------------------- MetricNet
def fb_consistency(fwd, bwd, alpha=0.01, beta=0.5):
    """geometry.py:87-108: occlusion where |f + warp(b, f)| > alpha*(|f|+|b|) + beta."""
    mag = torch.norm(fwd, dim=1) + torch.norm(bwd, dim=1)
    d_f = torch.norm(fwd + _gmflow.flow_warp(bwd, fwd), dim=1)
    d_b = torch.norm(bwd + _gmflow.flow_warp(fwd, bwd), dim=1)
    thr = alpha * mag + beta
    return (d_f > thr).to(fwd), (d_b > thr).to(bwd)


def metricnet(sd, img0, img1, flow01, flow10, union=True):
    """MetricNet.forward (model_gmfss_union/MetricNet.py:45-65).  union=False: model_gmfss/MetricNet.py (no Tanh*10)."""
    m0 = F.l1_loss(img0, backwarp_zeros(img1, flow01), reduction="none").mean([1], True)
    m1 = F.l1_loss(img1, backwarp_zeros(img0, flow10), reduction="none").mean([1], True)
    occ_f, occ_b = fb_consistency(flow01, flow10)
    nf01 = torch.cat([flow01[:, 0:1] / ((flow01.shape[3] - 1.0) / 2.0), flow01[:, 1:2] / ((flow01.shape[2] - 1.0) / 2.0)], 1)
    nf10 = torch.cat([flow10[:, 0:1] / ((flow10.shape[3] - 1.0) / 2.0), flow10[:, 1:2] / ((flow10.shape[2] - 1.0) / 2.0)], 1)
    x = torch.cat((img0, img1, -m0, -m1, nf01, nf10, occ_f.unsqueeze(1), occ_b.unsqueeze(1)), 1)
    feat = F.conv2d(x, sd["metric_in.weight"], sd["metric_in.bias"], padding=1)
    for k in (1, 2, 3):
        p = f"metric_net{k}."
        feat = F.conv2d(_prelu(feat, sd[p + "0.weight"]), sd[p + "1.weight"], sd[p + "1.bias"], padding=1) + feat
    out = F.conv2d(_prelu(feat, sd["metric_out.0.weight"]), sd["metric_out.1.weight"], sd["metric_out.1.bias"], padding=1)
    if union:
        out = torch.tanh(out) * 10
    return out[:, :1], out[:, 1:2]


# ----------------------------------------------------------------------------------------- GridNet
def _two_conv(sd, p, x, stride1=1, transposed=False):
    """PReLU - conv(stride1) | deconv4x4 s2 - PReLU - conv (FusionNet.py:6-31)."""
    x = _prelu(x, sd[p + "0.weight"])
    if transposed:
        x = F.conv_transpose2d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride=2, padding=1)
    else:
        x = F.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride=stride1, padding=1)
    return F.conv2d(_prelu(x, sd[p + "2.weight"]), sd[p + "3.weight"], sd[p + "3.bias"], stride=1, padding=1)


def gridnet(sd, x, x1, x2, x3):
    """GridNet.forward (FusionNet.py:106-146): 3-row grid of residual / down / up blocks, PixelShuffle tail."""
    R = lambda n, t: _two_conv(sd, f"residual_model_{n}.", t)  # noqa: E731
    D = lambda n, t: _two_conv(sd, f"downsample_model_{n}.", t, stride1=2)  # noqa: E731
    U = lambda n, t: _two_conv(sd, f"upsample_model_{n}.", t, transposed=True)  # noqa: E731
    head0 = "head0" if "residual_model_head0.0.weight" in sd else "head"  # model_gmfss/FusionNet.py:59 names it "head"
    X00 = R(head0, x) + R("head1", x1)
    X01 = R("01", X00) + X00
    X10 = D("10", X00) + R("head2", x2)
    X20 = D("20", X10) + R("head3", x3)
    X11 = (R("11", X10) + X10) + D("11", X01)
    X21 = (R("21", X20) + X20) + D("21", X11)
    X24 = R("24", X21) + X21
    X25 = R("25", X24) + X24
    X14 = U("14", X24) + (R("14", X11) + X11)
    X04 = U("04", X14) + (R("04", X01) + X01)
    X15 = U("15", X25) + (R("15", X14) + X14)
    X05 = U("05", X15) + (R("05", X04) + X04)
    p = "residual_model_tail."
    t = _prelu(F.conv2d(X05, sd[p + "conv_before_upsample.0.weight"], sd[p + "conv_before_upsample.0.bias"], padding=1),
               sd[p + "conv_before_upsample.1.weight"])
    t = F.pixel_shuffle(F.conv2d(t, sd[p + "upsample.0.weight"], sd[p + "upsample.0.bias"], padding=1), 2)
    return F.conv2d(t, sd[p + "conv_last.weight"], sd[p + "conv_last.bias"], padding=1)


# ----------------------------------------------------------------------------------------- Model
class GmfssModel:
    """Model of model_gmfss_union/GMFSS.py (union=True) or model_gmfss/GMFSS.py (union=False)."""

    def __init__(self, flownet, metric, feat, fusion, union=True):
        f = lambda d: {k: v.detach().float().cpu() for k, v in d.items()}  # noqa: E731
        self.flownet, self.metric, self.feat, self.fusion, self.union = f(flownet), f(metric), f(feat), f(fusion), union

    def reuse(self, img0, img1, scale):
        """Model.reuse (GMFSS.py:55-78): features of both frames, bidirectional GMFlow at 1/2*scale res, metrics."""
        feat0, feat1 = featurenet(self.feat, img0), featurenet(self.feat, img1)
        img0 = F.interpolate(img0, scale_factor=0.5, mode="bilinear", align_corners=False)
        img1 = F.interpolate(img1, scale_factor=0.5, mode="bilinear", align_corners=False)
        if scale != 1.0:
            if0 = F.interpolate(img0, scale_factor=scale, mode="bilinear", align_corners=False)
            if1 = F.interpolate(img1, scale_factor=scale, mode="bilinear", align_corners=False)
        else:
            if0, if1 = img0, img1
        flow01 = _gmflow.gmflow(self.flownet, if0, if1)
        flow10 = _gmflow.gmflow(self.flownet, if1, if0)
        if scale != 1.0:
            flow01 = F.interpolate(flow01, scale_factor=1. / scale, mode="bilinear", align_corners=False) / scale
            flow10 = F.interpolate(flow10, scale_factor=1. / scale, mode="bilinear", align_corners=False) / scale
        m0, m1 = metricnet(self.metric, img0, img1, flow01, flow10, self.union)
        return flow01, flow10, m0, m1, feat0, feat1

    def inference(self, img0, img1, reuse, timestep0, timestep1, rife=None):
        """Model.inference (GMFSS.py:80-155): soft-splat the half-res frames and the 3-level feature pyramid to time
        t from both sides; with map timesteps (DRBA) also splat the timestep maps, fill holes with 1 and swap regions
        where one side's timestep is > 25x the other's; GridNet fusion; clamp."""
        out = gridnet(self.fusion, *self.fusion_inputs(img0, img1, reuse, timestep0, timestep1, rife))
        return torch.clamp(out, 0, 1)

    def fusion_inputs(self, img0, img1, reuse, timestep0, timestep1, rife=None):
        """GMFSS.py:80-152, everything of Model.inference before GridNet: -> (x [1,9,h,w], pyramid level 1 [1,128,h,w], level 2
        [1,256,h/2,w/2], level 3 [1,384,h/4,w/4]).  Split out so that the tests can check the splat stage and GridNet each on
        the other implementation's inputs (same operations in the same order as before the split)."""
        flow01, flow10, metric0, metric1, (f11, f12, f13), (f21, f22, f23) = reuse
        F1t, F2t = timestep0 * flow01, timestep1 * flow10
        Z1t, Z2t = timestep0 * metric0, timestep1 * metric1
        img0 = F.interpolate(img0, scale_factor=0.5, mode="bilinear", align_corners=False)
        img1 = F.interpolate(img1, scale_factor=0.5, mode="bilinear", align_corners=False)
        I1t = softsplat(img0, F1t, Z1t, "soft")
        I2t = softsplat(img1, F2t, Z2t, "soft")
        a1 = softsplat(f11, F1t, Z1t, "soft")
        b1 = softsplat(f21, F2t, Z2t, "soft")

        def down(flow, z, s):
            return (F.interpolate(flow, scale_factor=s, mode="bilinear", align_corners=False) * s,
                    F.interpolate(z, scale_factor=s, mode="bilinear", align_corners=False))

        a2 = softsplat(f12, *down(F1t, Z1t, 0.5), "soft")
        b2 = softsplat(f22, *down(F2t, Z2t, 0.5), "soft")
        a3 = softsplat(f13, *down(F1t, Z1t, 0.25), "soft")
        b3 = softsplat(f23, *down(F2t, Z2t, 0.25), "soft")
        if self.union and isinstance(timestep0, torch.Tensor):
            t0 = softsplat(timestep0, F1t, Z1t, "soft")
            t1 = softsplat(timestep1, F2t, Z2t, "soft")
            gaps0 = softsplat(t0.clone() * 0 + 1, F1t, Z1t, "soft") < 0.999
            gaps1 = softsplat(t1.clone() * 0 + 1, F2t, Z2t, "soft") < 0.999
            bad = torch.logical_or(gaps0, gaps1)
            t0 = torch.where(bad, torch.ones_like(t0), t0)
            t1 = torch.where(bad, torch.ones_like(t1), t1)

            def swap(x, y, c, s):
                u, v = t0, t1
                if s != 1.0:
                    u = F.interpolate(u, scale_factor=s, mode="bilinear", align_corners=False)
                    v = F.interpolate(v, scale_factor=s, mode="bilinear", align_corners=False)
                m0 = (u / v > 25).repeat(1, c, 1, 1)
                m1 = (v / u > 25).repeat(1, c, 1, 1)
                # x[m0], y[m1] = y[m0], x[m1]  (both right-hand sides gathered before either write)
                return torch.where(m0, y, x), torch.where(m1, x, y)

            I1t, I2t = swap(I1t, I2t, 3, 1.0)
            a1, b1 = swap(a1, b1, 64, 1.0)
            a2, b2 = swap(a2, b2, 128, 0.5)
            a3, b3 = swap(a3, b3, 192, 0.25)
        if self.union:
            x = torch.cat([I1t, rife, I2t], 1)
        else:
            x = torch.cat([img0, I1t, I2t, img1], 1)  # model_gmfss/GMFSS.py:162
        return x, torch.cat([a1, b1], 1), torch.cat([a2, b2], 1), torch.cat([a3, b3], 1)


class GmfssUnionOracle:
    """Call surface of models/gmfss_union.py:10-100."""

    def __init__(self, flownet, metric, feat, fusion, rife, scale=1.0):
        self.model = GmfssModel(flownet, metric, feat, fusion, union=True)
        self.rife = {k: v.detach().float().cpu() for k, v in rife.items()}
        self.scale = scale
        self.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
        self.pad_size = 128

    @torch.no_grad()
    def warm_reuse(self, Ia, Ib):
        """reuse entering the step after a DRBA step on (.., Ia, Ib): model.reuse(Ia, Ib), roles swapped."""
        r = self.model.reuse(Ia, Ib, self.scale)
        return [v for pair in zip(r[1::2], r[0::2]) for v in pair]

    @torch.no_grad()
    def inference_ts(self, I0, I1, ts):
        reuse = self.model.reuse(I0, I1, self.scale)
        out = []
        for t in ts:
            if t == 0:
                out.append(I0)
            elif t == 1:
                out.append(I1)
            else:
                I0s = F.interpolate(I0, scale_factor=0.5, mode="bilinear", align_corners=False)
                I1s = F.interpolate(I1, scale_factor=0.5, mode="bilinear", align_corners=False)
                rife = _ifnet.ifnet(self.rife, torch.cat((I0s, I1s), 1), timestep=t, scale_list=self.scale_list)[0]
                out.append(self.model.inference(I0, I1, reuse, timestep0=t, timestep1=1 - t, rife=rife))
        return out

    @torch.no_grad()
    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
        r10 = self.model.reuse(I1, I0, self.scale) if reuse is None else reuse
        r12 = self.model.reuse(I1, I2, self.scale)
        flow10, metric10 = r10[0], r10[2]
        flow12, metric12 = r12[0], r12[2]
        I0s, I1s, I2s = [F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False) for x in (I0, I1, I2)]
        out = []
        for t in ts:
            if t == 0:
                out.append(I0)
            elif t == 1:
                out.append(I1)
            elif t == 2:
                out.append(I2)
            elif 0 < t < 1 or 1 < t < 2:
                left = t < 1
                tt = 1 - t if left else t - 1
                dg = calc_drm_gmfss(tt, flow10, flow12, metric10, metric12, linear)
                dr = calc_drm_rife_auxiliary(tt, flow10, flow12, metric10, metric12, linear)
                dr = {k: resize(v, I0s.shape[2:]) for k, v in dr.items()}
                if left:
                    rife = _ifnet.ifnet(self.rife, torch.cat((I1s, I0s), 1), timestep=dr["drm_t1_t01"], scale_list=self.scale_list)[0]
                    out.append(self.model.inference(I1, I0, r10, dg["drm1t_t01"], dg["drm0t_t01"], rife))
                else:
                    rife = _ifnet.ifnet(self.rife, torch.cat((I1s, I2s), 1), timestep=dr["drm_t1_t12"], scale_list=self.scale_list)[0]
                    out.append(self.model.inference(I1, I2, r12, dg["drm1t_t12"], dg["drm2t_t12"], rife))
        new = [v for pair in zip(r12[1::2], r12[0::2]) for v in pair]  # (flow10, flow01, m1, m0, feat1, feat0)
        return out, new


class GmfssOracle:
    """Call surface of models/gmfss.py:7-73 (non-union: no auxiliary RIFE frame, no swap masks, pad 64)."""

    def __init__(self, flownet, metric, feat, fusion, scale=1.0):
        self.model = GmfssModel(flownet, metric, feat, fusion, union=False)
        self.scale = scale
        self.pad_size = 64

    @torch.no_grad()
    def warm_reuse(self, Ia, Ib):
        """reuse entering the step after a DRBA step on (.., Ia, Ib): model.reuse(Ia, Ib), roles swapped."""
        r = self.model.reuse(Ia, Ib, self.scale)
        return [v for pair in zip(r[1::2], r[0::2]) for v in pair]

    @torch.no_grad()
    def inference_ts(self, I0, I1, ts):
        reuse = self.model.reuse(I0, I1, self.scale)
        return [I0 if t == 0 else I1 if t == 1 else self.model.inference(I0, I1, reuse, t, 1 - t) for t in ts]

    @torch.no_grad()
    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
        r10 = self.model.reuse(I1, I0, self.scale) if reuse is None else reuse
        r12 = self.model.reuse(I1, I2, self.scale)
        flow10, metric10, flow12, metric12 = r10[0], r10[2], r12[0], r12[2]
        out = []
        for t in ts:
            if t == 0:
                out.append(I0)
            elif t == 1:
                out.append(I1)
            elif t == 2:
                out.append(I2)
            elif 0 < t < 1:
                d = calc_drm_gmfss(1 - t, flow10, flow12, metric10, metric12, linear)
                out.append(self.model.inference(I1, I0, r10, d["drm1t_t01"], d["drm0t_t01"]))
            elif 1 < t < 2:
                d = calc_drm_gmfss(t - 1, flow10, flow12, metric10, metric12, linear)
                out.append(self.model.inference(I1, I2, r12, d["drm1t_t12"], d["drm2t_t12"]))
        return out, [v for pair in zip(r12[1::2], r12[0::2]) for v in pair]
