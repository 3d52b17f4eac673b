"""Shared parity cases: one list, three executors.

Each case is (name, fn) where fn(backend) returns a tensor (or dict of tensors).  The SAME
list is evaluated by
  * tests/golden/make_golden.py with the imported reference  -> writes the fixtures,
  * tests/test_oracle_golden.py with the oracle               -> oracle vs fixtures (CPU),
  * tests/test_gpu_*.py with the HIP path                     -> HIP vs oracle and vs fixtures.

A backend exposes: dev, warp, softsplat, distance, resize, calc_drm_rife, calc_drm_gmfss,
calc_drm_rife_auxiliary, get_drm_t, ssim_matlab, check_scene, make_rife(sd, scale).
Inputs regenerate from seeds; nothing here reads /root/reference.
"""
import numpy as np
import torch

from drba_amd.utils import synth


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# ------------------------------------------------------------------------------------------ ops
def ops_inputs():
    H, W = 48, 80
    d = {
        "x3": torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1)),
        "x16": rnd((1, 16, H, W), 2),
        "flow": rnd((1, 2, H, W), 3, 4.0),
        "flow_big": rnd((1, 2, H, W), 4, 30.0),  # mostly out of bounds: border clamp / dropped corners
        "metric": rnd((1, 1, H, W), 5, 2.0),
    }
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    # zoom-out by 4 about the centre: ~25 sources land on every output pixel near the centre (list overflow path of
    # the gather splat); the outer columns exceed the 16-pixel tile halo (long-flow path)
    d["flow_converge"] = torch.stack([-0.75 * (xs - W / 2 + 0.3), -0.75 * (ys - H / 2 + 0.2)]).unsqueeze(0).contiguous()
    fn = d["flow"].clone()
    fn[0, 0, 5, 7] = float("nan")
    fn[0, 1, 9, 11] = float("inf")
    d["flow_nan"] = fn
    return d


def ops_cases():
    i = ops_inputs()

    def on(b, *names):
        return [None if n is None else i[n].to(b.dev) for n in names]

    def warp(xn, fn):
        return lambda b: b.warp(*on(b, xn, fn))

    def splat(xn, fn, mn, mode, sl=None, mabs=False):
        def run(b):
            x, f, m = on(b, xn, fn, mn)
            if sl is not None:
                x = x[:, sl].contiguous()
            if mabs:
                m = m.abs() + 0.1
            return b.softsplat(x, f, m, mode)
        return run

    cases = [
        ("warp3", warp("x3", "flow")), ("warp16", warp("x16", "flow")), ("warp3_big", warp("x3", "flow_big")),
        ("splat_sum", splat("x3", "flow", None, "sum")),
        ("splat_avg", splat("x3", "flow", None, "avg")),
        ("splat_avg_big", splat("x3", "flow_big", None, "avg")),
        ("splat_avg_nan", splat("x3", "flow_nan", None, "avg")),
        ("splat_avg1", splat("x3", "flow", None, "avg", sl=slice(0, 1))),
        ("splat_avg2", splat("flow", "flow", None, "avg")),
        ("splat_linear", splat("x3", "flow", "metric", "linear", mabs=True)),
        ("splat_soft", splat("x3", "flow", "metric", "soft")),
        ("splat_soft16", splat("x16", "flow", "metric", "soft")),
        ("splat_soft_zeroeps", splat("x3", "flow_big", "metric", "soft-zeroeps")),
        ("splat_soft_clipeps", splat("x3", "flow_big", "metric", "soft-clipeps")),
        ("splat_avg_addeps", splat("x3", "flow", None, "avg-addeps")),
        ("splat_soft16_big", splat("x16", "flow_big", "metric", "soft")),
        ("splat_soft16_nan", splat("x16", "flow_nan", "metric", "soft")),
        ("splat_soft16_converge", splat("x16", "flow_converge", "metric", "soft")),
        ("splat_sum16_converge", splat("x16", "flow_converge", None, "sum")),
        ("splat_linear16_zeroeps", splat("x16", "flow_big", "metric", "linear-zeroeps", mabs=True)),
        ("splat_avg3_converge", splat("x3", "flow_converge", None, "avg")),
        ("distance", lambda b: b.distance(*on(b, "flow"))),
        ("resize_up", lambda b: b.resize(on(b, "x3")[0], (64, 96))),
        ("resize_down", lambda b: b.resize(on(b, "x3")[0], (30, 50))),
    ]
    return cases


# ------------------------------------------------------------------------------------------ drm
def drm_inputs():
    H, W = 48, 80
    f10 = rnd((1, 2, H, W), 11, 5.0)
    f12 = rnd((1, 2, H, W), 12, 3.0)
    f10[0, :, 3, 4] = 0.0
    f12[0, :, 3, 4] = 0.0  # both-zero pixel: NaN in calc_drm_gmfss (no +1e-4), finite in the rife variants
    return {"f10": f10, "f12": f12, "m10": rnd((1, 1, H, W), 13, 2.0), "m12": rnd((1, 1, H, W), 14, 2.0),
            "d": torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(15))}


def drm_cases():
    i = drm_inputs()
    cases = []

    def rife(t, lin):
        return lambda b: b.calc_drm_rife(t, i["f10"].to(b.dev), i["f12"].to(b.dev), lin)

    def other(kind, t, lin, mm):
        def run(b):
            m10, m12 = (i["m10"].to(b.dev), i["m12"].to(b.dev)) if mm else (None, None)
            return getattr(b, kind)(t, i["f10"].to(b.dev), i["f12"].to(b.dev), m10, m12, lin)
        return run

    for t in (0.25, 0.4, 0.5):
        for lin in (True, False):
            cases.append((f"rife_t{t}_{'lin' if lin else 'nl'}", rife(t, lin)))
    for t, lin, mm in ((0.25, True, True), (0.4, False, True), (0.5, True, False)):
        tag = f"t{t}_{'lin' if lin else 'nl'}_{'soft' if mm else 'avg'}"
        cases.append((f"gmfss_{tag}", other("calc_drm_gmfss", t, lin, mm)))
        cases.append((f"aux_{tag}", other("calc_drm_rife_auxiliary", t, lin, mm)))
    for t in (0.2, 0.5, 0.8):
        cases.append((f"drm_to_t_{t}", (lambda tt: (lambda b: b.get_drm_t(i["d"].to(b.dev), tt)))(t)))
    return cases


# ------------------------------------------------------------------------------------------ scdet
SCDET_PAIRS = ((0, 1), (1, 2), (2, 3), (3, 4), (0, 0))


def scdet_frames():
    clip = synth.make_clip(6, 96, 160, seed=77, cut_at=3)
    return [torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float().div(255.0) for f in clip]


# ------------------------------------------------------------------------------------------ rife end to end
RIFE_CONFIGS = ((1.0, (128, 192)), (0.5, (256, 384)))
TS_PATTERNS = (("t2", np.array([0.75, 1.25])), ("f3", np.array([0.6, 1.0, 1.4])), ("f2", np.array([0.8, 1.2])))


def rife_frames(H, W):
    clip = synth.make_clip(4, H, W, seed=1234)
    return [torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float().div(255.0) for f in clip]


def planar(t):
    """A feature tensor in the reference's [1,16,H,W] layout: the HIP path carries the encoder features pair-interleaved
    ([8,H,W,2], tagged `_drba_is_pair`; drba_amd.ops.features_planar) wherever the caller does not ask for them."""
    if getattr(t, "_drba_is_pair", False):
        c2, h, w, _ = t.shape
        return t.permute(0, 3, 1, 2).reshape(1, 2 * c2, h, w).contiguous()
    return t


def rife_run(b, sd, scale, H, W):
    """All end-to-end RIFE outputs for one (scale, size) as an ordered {name: tensor} dict."""
    m = b.make_rife(sd, scale)
    I0, I1, I2, I3 = [f.to(b.dev) for f in rife_frames(H, W)]
    tag = f"s{scale}"
    out = {}
    if scale == 1.0:
        out["head"] = m.encode(I1)
    ts = np.array([0.0, 0.25, 0.5, 1.0])
    r = m.inference_ts(I0, I1, ts)
    assert r[0] is I0 and r[3] is I1, "t==0/1 must return the input tensor object itself (rife.py:30-33)"
    out[f"ts_{tag}_1"], out[f"ts_{tag}_2"] = r[1], r[2]
    r = m.calc_flow(I1, I0)
    for k, nm in enumerate(("flow01", "flow10", "f0", "f1")):
        out[f"calcflow_{tag}_{nm}"] = r[k]
    for ts_name, ts in TS_PATTERNS:
        r, reuse = m.inference_ts_drba(I0, I1, I2, ts, None, True)
        for k in range(len(ts)):
            if ts[k] == 1.0:
                assert r[k] is I1
                continue
            out[f"drba_cold_{tag}_{ts_name}_{k}"] = r[k]
        if ts_name == "t2":
            for k, nm in enumerate(("flow21", "flow12", "f2", "f1")):
                out[f"drba_cold_{tag}_reuse_{nm}"] = reuse[k]
            r2, _ = m.inference_ts_drba(I1, I2, I3, ts, reuse, True)
            for k in range(len(ts)):
                out[f"drba_warm_{tag}_{ts_name}_{k}"] = r2[k]
    if scale == 1.0:
        r, _ = m.inference_ts_drba(I0, I1, I2, np.array([0.75]), None, False)
        out["drba_nonlinear_0"] = r[0]
    return {k: planar(v) for k, v in out.items()}


# ------------------------------------------------------------------------------------------ gmfss_union end to end
GMFSS_CONFIGS = ((1.0, (128, 256)), (0.5, (256, 512)))


def gmfss_frames(H, W):
    clip = synth.make_clip(4, H, W, seed=4321)
    return [torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float().div(255.0) for f in clip]


def gmfss_union_run(b, sds, scale, H, W, ulp_noise=False):
    """End-to-end GMFSS_UNION outputs for one (scale, size): inference_ts, cold + warm inference_ts_drba, reuse parts.
    ulp_noise=True adds a seeded +-1e-7 (about one ulp of a pixel value) perturbation to the frames: the change it
    causes in the oracle's outputs is the conditioning floor of an fp32 evaluation of this path."""
    m = b.make_gmfss_union(sds, scale)
    frames = gmfss_frames(H, W)
    if ulp_noise:
        g = torch.Generator().manual_seed(99)
        frames = [f + (torch.rand(f.shape, generator=g) - 0.5) * 2e-7 for f in frames]
    I0, I1, I2, I3 = [f.to(b.dev) for f in frames]
    tag = f"s{scale}"
    out = {}
    r = m.inference_ts(I0, I1, np.array([0.0, 0.5, 1.0]))
    assert r[0] is I0 and r[2] is I1
    out[f"ts_{tag}"] = r[1]
    for ts_name, ts in (("t2", np.array([0.75, 1.25])), ("f3", np.array([0.6, 1.0, 1.4]))):
        r, reuse = m.inference_ts_drba(I0, I1, I2, ts, None, True)
        for k in range(len(ts)):
            if ts[k] != 1.0:
                out[f"drba_cold_{tag}_{ts_name}_{k}"] = r[k]
        if ts_name == "t2":
            for k, nm in enumerate(("flow21", "flow12", "metric2", "metric1")):
                out[f"reuse_{tag}_{nm}"] = reuse[k]
            out[f"reuse_{tag}_feat2_0"], out[f"reuse_{tag}_feat2_2"] = reuse[4][0], reuse[4][2]
            r2, _ = m.inference_ts_drba(I1, I2, I3, ts, reuse, True)
            for k in range(len(ts)):
                out[f"drba_warm_{tag}_{ts_name}_{k}"] = r2[k]
    if scale == 1.0:
        r, _ = m.inference_ts_drba(I0, I1, I2, np.array([1.3]), None, False)  # non-linear DRM
        out["drba_nonlinear"] = r[0]
    return out


def gmfss_state_dicts(seed=0):
    """Non-union GMFSS: same GMFlow / MetricNet (no tanh) / FeatureNet weights, GridNet with 12 input channels."""
    sds = synth.gmfss_union_state_dicts(seed)
    return {"flownet": sds["flownet"], "metric": sds["metric"], "feat": sds["feat"],
            "fusion": synth.seeded_state_dict(synth.gridnet_shapes(12, "head"), seed, "grid12.")}


def gmfss_run(b, sds, scale, H, W):
    """End-to-end GMFSS (models/gmfss.py) outputs: inference_ts, cold + warm inference_ts_drba."""
    m = b.make_gmfss(sds, scale)
    I0, I1, I2, I3 = [f.to(b.dev) for f in gmfss_frames(H, W)]
    out = {}
    r = m.inference_ts(I0, I1, np.array([0.0, 0.4, 1.0]))
    assert r[0] is I0 and r[2] is I1
    out["ts"] = r[1]
    ts = np.array([0.75, 1.25])
    r, reuse = m.inference_ts_drba(I0, I1, I2, ts, None, True)
    out["drba_cold_0"], out["drba_cold_1"] = r
    out["reuse_metric2"] = reuse[2]
    r2, _ = m.inference_ts_drba(I1, I2, I3, ts, reuse, False)
    out["drba_warm_nl_0"], out["drba_warm_nl_1"] = r2
    return out


# ------------------------------------------------------------------------------------------ trained weights
TRAINED_NPZ = "trained_union_weights.npz"  # FeatureNet / MetricNet of the reference's weights/train_log_gmfss_union (data)


def trained_state_dicts(golden_dir, seed=0):
    """GMFSS_UNION weight set whose FeatureNet and MetricNet are the TRAINED ones shipped in the reference mount
    (weights/train_log_gmfss_union/{feat,metric}.pkl, CUDA-tagged pickles re-saved as plain fp32 arrays by
    tests/golden/make_golden.py trained); GMFlow, GridNet and the auxiliary RIFE stay seeded (the mount has none)."""
    import os
    z = np.load(os.path.join(golden_dir, TRAINED_NPZ))
    sds = dict(synth.gmfss_union_state_dicts(seed))
    for net in ("feat", "metric"):
        sds[net] = {k[len(net) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(net + "/")}
    return sds


def trained_run(b, sds, H=128, W=256):
    """Trained FeatureNet / MetricNet in isolation (MetricNet on seeded smooth flows) and inside one cold + one warm
    GMFSS_UNION step.  Backends expose featurenet(sd, x) and metricnet(sd, h0, h1, f01, f10, union)."""
    import torch.nn.functional as F
    I0, I1, I2, I3 = [f.to(b.dev) for f in gmfss_frames(H, W)]
    h0 = F.interpolate(I0, scale_factor=0.5, mode="bilinear", align_corners=False)
    h1 = F.interpolate(I1, scale_factor=0.5, mode="bilinear", align_corners=False)
    f01 = ((synth._smooth_field(2, H // 2, W // 2, 71) - 0.5) * 12.0).to(b.dev)
    f10 = ((synth._smooth_field(2, H // 2, W // 2, 72) - 0.5) * 12.0).to(b.dev)
    out = {}
    for k, t in enumerate(b.featurenet(sds["feat"], I0)):
        out[f"featurenet_{k}"] = t
    for k, t in enumerate(b.metricnet(sds["metric"], h0, h1, f01, f10, True)):
        out[f"metricnet_{k}"] = t
    m = b.make_gmfss_union(sds, 1.0)
    ts = np.array([0.75, 1.25])
    r, reuse = m.inference_ts_drba(I0, I1, I2, ts, None, True)
    out["drba_cold_0"], out["drba_cold_1"] = r
    out["reuse_metric2"], out["reuse_feat2_0"] = reuse[2], reuse[4][0]
    r2, _ = m.inference_ts_drba(I1, I2, I3, ts, reuse, True)
    out["drba_warm_0"], out["drba_warm_1"] = r2
    return out


def undamped_gmflow_sd(seed=0):
    """GMFlow weights with the transformer's LayerNorm gains at ~1 instead of the damped 0.1 of the main fixtures: the
    random attention/FFN messages swamp the CNN features, the correlation softmax is diffuse and the flow is
    ill-conditioned -- the case that documents how far the 1e-3 bar holds."""
    return synth.seeded_state_dict(synth.gmflow_shapes(), seed, "gmflow.", damp_transformer=False)


def undamped_gmflow_run(b, H=128, W=256, ulp_noise=False):
    import torch.nn.functional as F
    fr = gmfss_frames(H, W)[:2]
    if ulp_noise:
        g = torch.Generator().manual_seed(99)
        fr = [f + (torch.rand(f.shape, generator=g) - 0.5) * 2e-7 for f in fr]
    h0, h1 = [F.interpolate(f, scale_factor=0.5, mode="bilinear", align_corners=False).to(b.dev) for f in fr]
    return {"flow01": b.gmflow(undamped_gmflow_sd(), h0, h1)}


# ------------------------------------------------------------------------------------------ fixture packing
MAX_FULL = 1 << 15


def pack(t):
    """Whole tensor if small, else a strided sample + float64 moments."""
    a = t.detach().float().cpu().contiguous().numpy()
    flat = a.reshape(-1)
    if flat.size <= MAX_FULL:
        return {"full": a}
    stride = max(1, flat.size // 6000)
    return {"sample": flat[::stride].copy(), "stride": np.int64(stride), "shape": np.array(a.shape, np.int64),
            "sum": np.float64(flat.astype(np.float64).sum()), "abssum": np.float64(np.abs(flat.astype(np.float64)).sum())}


def flatten(name, value):
    """A case result (tensor or dict of tensors) -> [(key, tensor)]."""
    if isinstance(value, dict):
        return [(f"{name}_{k}", v) for k, v in value.items()]
    return [(name, value)]


def compare_to_fixture(z, key, t, count_above=None):
    """Max abs difference between tensor t and fixture entry `key` in npz z (NaNs must coincide).
    count_above=tol -> (max, number of compared elements above tol, number compared)."""
    a = t.detach().float().cpu().contiguous().numpy()
    if f"{key}/full" in z.files:
        ref = z[f"{key}/full"]
        got = a
    else:
        assert tuple(z[f"{key}/shape"]) == a.shape, (key, a.shape, tuple(z[f"{key}/shape"]))
        ref = z[f"{key}/sample"]
        got = a.reshape(-1)[::int(z[f"{key}/stride"])]
    assert ref.shape == got.shape, (key, ref.shape, got.shape)
    nan_r, nan_g = np.isnan(ref), np.isnan(got)
    assert (nan_r == nan_g).all(), f"{key}: NaN pattern differs"
    d = np.abs(np.where(nan_r, 0, ref) - np.where(nan_g, 0, got))
    mx = float(d.max()) if d.size else 0.0
    if count_above is None:
        return mx
    return mx, int((d > count_above).sum()), int(d.size)
