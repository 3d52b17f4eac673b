"""GMFSS wrapper with the DRBA call surface (reference models/gmfss.py:7-73), HIP path, fp32.
Same networks as GMFSS_UNION without the auxiliary RIFE frame and the swap masks; MetricNet has no tanh*10."""
import torch

from drba_amd import ops as _ops
from drba_amd.models.drm import calc_drm_gmfss
from drba_amd.models.lookahead import Lookahead, split as split_lookahead
from drba_amd.models.model_gmfss_union.GMFSS import Model


class GMFSS:
    def __init__(self, weights="weights/train_log_gmfss", scale=1.0, device=None):
        device = _ops.default_device() if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("drba_amd GMFSS runs on the MI355X HIP path only; there is no CPU fallback")
        self.device = device
        _ops.status_init(device)  # the two-term fp16 kernels report an overflow from now on (checked once per call)
        self.model = Model(union=False)
        if isinstance(weights, dict):
            self.model.load_state_dicts(weights["flownet"], weights["metric"], weights["feat"], weights["fusion"], device)
        else:
            self.model.load_model(weights, -1, device)
        self.scale = scale
        self.pad_size = 64

    def inference_ts(self, I0, I1, ts):
        _ops.check_overflow(getattr(self, "device", None))
        reuse = self.model.reuse(I0, I1, self.scale)
        output = []
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            else:
                output.append(self.model.inference(I0, I1, reuse, timestep0=float(t), timestep1=float(1 - t)))
        return output

    supports_lookahead = True
    _look = None

    def warm_reuse(self, Ia, Ib):
        """The `reuse` a DRBA step ending on the pair (Ia, Ib) hands to the next step (gmfss.py:70-72): model.reuse(Ia, Ib)
        with the roles swapped.  Used by drba_amd.parallel to rebuild the state entering a shard."""
        r = self.model.reuse(Ia, Ib, self.scale)
        return [v for pair in zip(r[1::2], r[0::2]) for v in pair]

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
        output = []
        for t in ts:
            if t == 0:
                output.append(I0)
            elif t == 1:
                output.append(I1)
            elif t == 2:
                output.append(I2)
            elif 0 < t < 1:
                d = calc_drm_gmfss(1 - t, flow10, flow12, metric10, metric12, linear)
                output.append(self.model.inference(I1, I0, reuseI1I0, d["drm1t_t01"], d["drm0t_t01"]))
            elif 1 < t < 2:
                d = calc_drm_gmfss(t - 1, flow10, flow12, metric10, metric12, linear)
                output.append(self.model.inference(I1, I2, reuseI1I2, d["drm1t_t12"], d["drm2t_t12"]))
        new_reuse = [v for pair in zip(reuseI1I2[1::2], reuseI1I2[0::2]) for v in pair]
        return output, new_reuse
