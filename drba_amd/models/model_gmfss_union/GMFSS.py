"""GMFSS(_UNION) Model on the HIP path (reference models/model_gmfss_union/GMFSS.py:19-155; model_gmfss/GMFSS.py for
union=False): FeatureNet x2, GMFlow both directions, MetricNet (`reuse`), then softmax-splatting of the half-res
frames and the 3-level feature pyramid, timestep-map swap masks and GridNet fusion (`inference`)."""
import torch

from drba_amd import ops as _ops
from drba_amd.models.gmflow.gmflow import GMFlow
from drba_amd.models.lookahead import _tensors
from drba_amd.models.model_gmfss_union.FeatureNet import FeatureNet
from drba_amd.models.model_gmfss_union.FusionNet import GridNet
from drba_amd.models.model_gmfss_union.MetricNet import MetricNet
from drba_amd.models.softsplat.softsplat import softsplat as warp


def _half(x, s=0.5):
    """F.interpolate(x, scale_factor=s, bilinear, align_corners=False)."""
    _, _, h, w = x.shape
    return _ops.resize_bilinear_scale(x, (int(h * s), int(w * s)), 1.0 / s)


def _times(t, x):
    """timestep * x for a scalar or a [1,1,H,W] map (GMFSS.py:86-90)."""
    return _ops.mul_map(x, t) if torch.is_tensor(t) else _ops.affine(x, float(t), 0.0)


class Model:
    def __init__(self, union=True):
        self.union = union
        self.flownet = self.metricnet = self.feat_ext = self.fusionnet = None
        self._device = None
        self.version = 3.9

    def eval(self):
        return self

    def device(self, device=None):
        self._device = device

    def load_state_dicts(self, flownet, metric, feat, fusion, device):
        self._device = device
        self.flownet = GMFlow(flownet, device)
        self.metricnet = MetricNet(metric, device, tanh10=self.union)
        self.feat_ext = FeatureNet(feat, device)
        self.fusionnet = GridNet(fusion, device)

    def load_model(self, path, rank=-1, device=None):
        """flownet.pkl / metric.pkl / feat.pkl / fusionnet.pkl (GMFSS.py:42-53); CUDA-tagged pickles load via map_location."""
        device = device or self._device or _ops.default_device()
        from drba_amd.models.utils.tools import load_weights
        ld = lambda n: load_weights(f"{path}/{n}.pkl")  # noqa: E731
        self.load_state_dicts(ld("flownet"), ld("metric"), ld("feat"), ld("fusionnet"), device)

    def _cached(self, frame, attr, key, make):
        """Per-frame cache on the frame tensor, keyed by (this model, key).  The value may have been produced on the
        lookahead's side stream: the producing event travels with it and the consumer's stream waits on it (a scene cut
        drops the lookahead RESULT without waiting, but these caches survive), and the allocator is told about the
        second stream."""
        key = (id(self), key)
        c = getattr(frame, attr, None)
        if c is not None and c[0] == key:
            if c[2] is not None:
                cur = torch.cuda.current_stream(frame.device)
                cur.wait_event(c[2])
                for t in _tensors(c[1]):
                    t.record_stream(cur)
            return c[1]
        val = make()
        if frame.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(frame.device))
            setattr(frame, attr, (key, val, ev))
        return val

    def _features(self, img):
        """FeatureNet pyramid of a frame, computed once per frame tensor (the reference recomputes it for both pairs a
        frame belongs to, GMFSS.py:56-57; same values)."""
        return self._cached(img, "_drba_feat", None, lambda: self.feat_ext(img))

    def _encoded(self, frame, flow_input, scale):
        """GMFlow's CNN encoding of `flow_input` (the frame at flow resolution), cached on the frame tensor per scale."""
        return self._cached(frame, "_drba_gmf", scale, lambda: self.flownet.encode_frame(flow_input))

    def reuse(self, img0, img1, scale):
        feat0, feat1 = self._features(img0), self._features(img1)
        f0_src, f1_src = img0, img1
        img0, img1 = _half(img0), _half(img1)
        if scale != 1.0:
            if0, if1 = _half(img0, scale), _half(img1, scale)
        else:
            if0, if1 = img0, img1
        # == (flownet(if0, if1), flownet(if1, if0)); each frame's CNN encoding is kept on the frame tensor, like its
        # FeatureNet pyramid: a frame is the second of one pair and the first of the next
        flow01, flow10 = self.flownet.bidirectional(if0, if1, feats=(self._encoded(f0_src, if0, scale), self._encoded(f1_src, if1, scale)))
        if scale != 1.0:
            _, _, h, w = img0.shape
            up = lambda f: _ops.affine(_ops.resize_bilinear_scale(f, (h, w), scale), 1.0 / scale, 0.0)  # noqa: E731
            flow01, flow10 = up(flow01), up(flow10)
        metric0, metric1 = self.metricnet(img0, img1, flow01, flow10)
        return flow01, flow10, metric0, metric1, feat0, feat1

    def inference(self, img0, img1, reuse_things, timestep0, timestep1, rife=None):
        return _ops.clamp(self.fusionnet(*self.fusion_inputs(img0, img1, reuse_things, timestep0, timestep1, rife)), 0.0, 1.0)

    # One GridNet pass over the stacked frames of a step: measured SLOWER in the step (same box, tools/gmfss_bench.py: 42.3 / 42.4
    # against 43.2 / 42.9 frames/s per frame -- alone the N = 2 layers are 10 % faster per FLOP, beside the lookahead stream's GMFlow
    # the longer launches overlap worse), so the frames go through GridNet one by one; True is kept for A/B runs (--batch-fusion).
    BATCH_FUSION = False
    SWAP_IN_PLACE = True  # tools/gmfss_bench.py --swap-copies: the splats go to temporaries and swap_select writes every pixel (round 5's form)

    def inference_many(self, jobs):
        """inference(*job) for every job of a step -- [(img0, img1, reuse_things, timestep0, timestep1, rife), ...], frames of one
        size -- with ONE GridNet pass over the stacked fusion inputs (the samples are independent, FusionNet.py:106-146): the
        splat stage of each job writes its slice of the [B, ...] buffers.  -> list of [1,3,H,W] frames."""
        B = len(jobs)
        if B == 1 or not self.BATCH_FUSION:
            return [self.inference(*j) for j in jobs]
        bufs = None
        for k, job in enumerate(jobs):
            bufs = self.fusion_inputs(*job, bufs=bufs, k=k, B=B)
        out = _ops.clamp(self.fusionnet(*bufs), 0.0, 1.0)
        return [out[k:k + 1] for k in range(B)]

    def fusion_inputs(self, img0, img1, reuse_things, timestep0, timestep1, rife=None, bufs=None, k=0, B=1):
        """GMFSS.py:80-152: the splat stage -> GridNet's inputs (x [1,9,h,w], [1,128,h,w], [1,256,h/2,w/2], [1,384,h/4,w/4]).
        bufs / k / B: sample k of B stacked inputs (inference_many): the [B, ...] buffers are made by the first job and returned."""
        flow01, flow10, metric0, metric1, (f11, f12, f13), (f21, f22, f23) = reuse_things
        F1t, F2t = _times(timestep0, flow01), _times(timestep1, flow10)
        Z1t, Z2t = _times(timestep0, metric0), _times(timestep1, metric1)
        img0, img1 = _half(img0), _half(img1)
        _, _, h, w = img0.shape
        dev = img0.device
        maps = self.union and torch.is_tensor(timestep0)  # DRBA: timestep maps -> the swap masks (GMFSS.py:112-150)
        # GridNet's inputs are written where they are consumed: slices of the concatenation buffers (no torch.cat copies)
        if bufs is None:
            bufs = (torch.empty((B, 9 if self.union else 12, h, w), dtype=torch.float32, device=dev),
                    torch.empty((B, 2 * f11.shape[1], h, w), dtype=torch.float32, device=dev),
                    torch.empty((B, 2 * f12.shape[1], h // 2, w // 2), dtype=torch.float32, device=dev),
                    torch.empty((B, 2 * f13.shape[1], h // 4, w // 4), dtype=torch.float32, device=dev))
        x, p1, p2, p3 = (t[k:k + 1] for t in bufs)
        c1, c2, c3 = f11.shape[1], f12.shape[1], f13.shape[1]
        xa, xb = (x[:, 0:3], x[:, 6:9]) if self.union else (x[:, 3:6], x[:, 6:9])
        # the splats write the slices; with maps swap_select then exchanges the selected pixels in place (False: temporaries, A/B runs)
        dst = (lambda t: t) if self.SWAP_IN_PLACE or not maps else (lambda t: None)  # noqa: E731

        def down(flow, z, s):
            return _ops.affine(_half(flow, s), s, 0.0), _half(z, s)

        # one sorted index per (flow, metric): the frame, its 64-channel features, the timestep map and the ones-mask of a side
        # are four gathers through it (the reference: four independent softsplat calls, GMFSS.py:92-117)
        def side(img, f1, ts, F, Z, xd, pd):
            ins, outs = [img, f1], [dst(xd), dst(pd)]
            if maps:
                ins.append(ts)
                outs.append(None)
            r = _ops.softsplat_many(ins, F, Z, "soft", outs, keep_quad=True)  # (the pyramid levels are per-frame caches: _features)
            cov = None
            if maps:  # t.clone() * 0 + 1 splatted along the same flow (GMFSS.py:116-117)
                cov = _ops.softsplat_many([_ops.affine(r[2], 0.0, 1.0)], F, Z, "soft", reuse_index=True)[0]
            return r, cov
        (I1t, a1, *ta), cov0 = side(img0, f11, timestep0, F1t, Z1t, xa, p1[:, :c1])
        (I2t, b1, *tb), cov1 = side(img1, f21, timestep1, F2t, Z2t, xb, p1[:, c1:])
        a2 = warp(f12, *down(F1t, Z1t, 0.5), "soft", out=dst(p2[:, :c2]), keep_quad=True)
        b2 = warp(f22, *down(F2t, Z2t, 0.5), "soft", out=dst(p2[:, c2:]), keep_quad=True)
        a3 = warp(f13, *down(F1t, Z1t, 0.25), "soft", out=dst(p3[:, :c3]), keep_quad=True)
        b3 = warp(f23, *down(F2t, Z2t, 0.25), "soft", out=dst(p3[:, c3:]), keep_quad=True)
        if maps:
            t0, t1 = _ops.timestep_fix(ta[0], tb[0], cov0, cov1)
            _ops.swap_select(I1t, I2t, t0, t1, 25.0, out=(xa, xb))
            _ops.swap_select(a1, b1, t0, t1, 25.0, out=(p1[:, :c1], p1[:, c1:]))
            _ops.swap_select(a2, b2, _half(t0, 0.5), _half(t1, 0.5), 25.0, out=(p2[:, :c2], p2[:, c2:]))
            _ops.swap_select(a3, b3, _half(t0, 0.25), _half(t1, 0.25), 25.0, out=(p3[:, :c3], p3[:, c3:]))
        if self.union:
            x[:, 3:6].copy_(rife)
        else:  # model_gmfss/GMFSS.py:162: cat([img0, I1t, I2t, img1])
            x[:, 0:3].copy_(img0)
            x[:, 9:12].copy_(img1)
        return bufs
