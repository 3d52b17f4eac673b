"""GMFSS_UNION wrapper with the DRBA call surface (reference models/gmfss_union.py:10-100), HIP path, fp32."""
import os

import torch

from drba_amd import ops as _ops
from drba_amd.models.drm import calc_drm_gmfss, calc_drm_rife_auxiliary
from drba_amd.models.lookahead import Lookahead, split as split_lookahead
from drba_amd.models.model_gmfss_union.GMFSS import Model, _half
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import IFNet
from drba_amd.models.utils.tools import convert, load_weights, resize


class GMFSS_UNION:
    def __init__(self, weights="weights/train_log_gmfss_union", scale=1.0, device=None):
        device = _ops.default_device() if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("drba_amd GMFSS_UNION runs on the MI355X HIP path only; there is no CPU fallback")
        self.device = device
        _ops.status_init(device)  # the two-term fp16 kernels report an overflow from now on (checked once per call)
        self.model = Model(union=True)
        if isinstance(weights, dict):  # already-loaded state dicts: flownet, metric, feat, fusion, rife
            self.model.load_state_dicts(weights["flownet"], weights["metric"], weights["feat"], weights["fusion"], device)
            rife_sd = weights["rife"]
        else:
            self.model.load_model(weights, -1, device)
            rife_sd = convert(load_weights(os.path.join(weights, "rife.pkl")))
        self.ifnet = IFNet().to(device).eval()
        self.ifnet.load_state_dict(rife_sd, strict=False)
        self.scale = scale
        self.scale_list = [16 / scale, 8 / scale, 4 / scale, 2 / scale, 1 / scale]
        self.pad_size = 128

    def inference_ts(self, I0, I1, ts):
        _ops.check_overflow(getattr(self, "device", None))
        reuse = self.model.reuse(I0, I1, self.scale)
        output = []
        I0s = I1s = None
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            else:
                if I0s is None:
                    I0s, I1s = _half(I0), _half(I1)
                rife = self.ifnet.forward_pair(I0s, I1s, float(t), self.scale_list, want_flows=False)[0]
                output.append(self.model.inference(I0, I1, reuse, timestep0=float(t), timestep1=float(1 - t), rife=rife))
        return output

    supports_lookahead = True
    _look = None

    def warm_reuse(self, Ia, Ib):
        """The `reuse` a DRBA step ending on the pair (Ia, Ib) hands to the next step (gmfss_union.py:95-98):
        model.reuse(Ia, Ib) with the roles swapped.  Used by drba_amd.parallel to rebuild the state entering a shard."""
        r = self.model.reuse(Ia, Ib, self.scale)
        return [v for pair in zip(r[1::2], r[0::2]) for v in pair]

    def _half_of(self, frame):
        """F.interpolate(frame, 0.5) (gmfss_union.py:72-74), once per frame tensor."""
        return self.model._cached(frame, "_drba_half", None, lambda: _half(frame))

    def _aux_enc(self, half):
        """The auxiliary IFNet's context encoding of a half-resolution frame, once per tensor (pair-interleaved layout)."""
        return self.model._cached(half, "_drba_auxenc", None, lambda: self.ifnet.encode(half, planar=False))

    def _pair_state(self, a, b):
        """model.reuse(a, b), taken from a matching lookahead if there is one (models/lookahead.py)."""
        res = self._look.take(a, b) if self._look is not None else None
        return res if res is not None else self.model.reuse(a, b, self.scale)

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False, lookahead=None):
        """`lookahead` (not in the reference): the frame that will be I2 of the next call; its pair state
        model.reuse(I2, lookahead) -- FeatureNet, GMFlow in both directions, MetricNet: hundreds of small launches --
        is started on a side stream and overlaps this call's splats and GridNet."""
        _ops.check_overflow(getattr(self, "device", None))
        reuseI1I0 = self.model.reuse(I1, I0, self.scale) if reuse is None else reuse
        reuseI1I2 = self._pair_state(I1, I2)
        lookahead, _ = split_lookahead(lookahead)
        if lookahead is not None and I2.is_cuda:
            if self._look is None:
                self._look = Lookahead()
            self._look.start(I2, lookahead, lambda: self.model.reuse(I2, lookahead, self.scale))
        flow10, metric10 = reuseI1I0[0], reuseI1I0[2]
        flow12, metric12 = reuseI1I2[0], reuseI1I2[2]
        # the auxiliary RIFE frames of ALL timesteps of the step in one stacked IFNet pass (the reference runs the half-resolution
        # IFNet once per frame, gmfss_union.py:77-93: ~100 latency-bound launches each), every frame's half-resolution copy and
        # context encoding made once per frame tensor (a frame is I2, then I1, then I0 of consecutive steps)
        I0s, I1s, I2s = self._half_of(I0), self._half_of(I1), self._half_of(I2)
        output, jobs, items = [], [], []
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            elif t == 2:
                output.append(I2)
            elif 0 < t < 1 or 1 < t < 2:
                left = t < 1
                tt = 1 - t if left else t - 1
                dg = calc_drm_gmfss(tt, flow10, flow12, metric10, metric12, linear)
                dr = calc_drm_rife_auxiliary(tt, flow10, flow12, metric10, metric12, linear)
                dr = {k: resize(v, I0s.shape[2:]) for k, v in dr.items()}
                a, b = (I1s, I0s) if left else (I1s, I2s)
                items.append((a, b, dr["drm_t1_t01"] if left else dr["drm_t1_t12"], self._aux_enc(a), self._aux_enc(b)))
                jobs.append((len(output), left, dg))
                output.append(None)
        if items:
            rifes = self.ifnet.forward_pairs(items, self.scale_list)
            work = [((I1, I0, reuseI1I0, dg["drm1t_t01"], dg["drm0t_t01"], rife) if left
                     else (I1, I2, reuseI1I2, dg["drm1t_t12"], dg["drm2t_t12"], rife)) for (_, left, dg), rife in zip(jobs, rifes)]
            for (slot, _, _), frame in zip(jobs, self.model.inference_many(work)):  # one GridNet pass over the step's frames
                output[slot] = frame
        # next step's (I1, I0) state = this step's (I1, I2) state with the roles swapped (gmfss_union.py:95-98)
        new_reuse = [v for pair in zip(reuseI1I2[1::2], reuseI1I2[0::2]) for v in pair]
        return output, new_reuse
