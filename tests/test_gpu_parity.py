"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the reference fixtures.

Tolerances: per-operator checks are fp32-roundoff class (atomics and MFMA summation order differ from
the CPU); the end-to-end bar is BASELINE.json's 1e-3 max-abs on the synthesised frames.
"""
import os

import numpy as np
import pytest
import torch

from tests import cases, gpu_checks, report

pytestmark = pytest.mark.gpu


def _assert_rows(rows, fixture_too=False):
    """Every row within its tolerance against the oracle; fixture_too: the same bar against the reference's own stored
    outputs (`vs_fixture=` in the row's details)."""
    import inspect
    report.record(inspect.stack()[1].function, rows)
    bad = [(n, e, t, x) for n, e, t, x in rows if not e <= t]
    if fixture_too:
        bad += [(n, float(x.split("vs_fixture=")[1].split()[0]), t, "vs the reference fixture") for n, e, t, x in rows
                if "vs_fixture=" in x and not float(x.split("vs_fixture=")[1].split()[0]) <= t]
    assert not bad, "\n".join(f"{n}: err={e:.3e} tol={t:.1e} {x}" for n, e, t, x in bad)


def test_native_library_is_what_runs():
    """The product must be the HIP library: it is loaded from the tree and there is no fallback."""
    from drba_amd import _lib
    lib = _lib.load()
    assert lib.drba_abi_version() == _lib.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libdrba_hip.so" in f.read()


def test_ops_parity(hip_backend, oracle_backend, golden_dir):
    _assert_rows(gpu_checks.check_cases(cases.ops_cases(), hip_backend, oracle_backend, 2e-5,
                                        np.load(os.path.join(golden_dir, "ops.npz"))), fixture_too=True)


def test_drm_parity(hip_backend, oracle_backend, golden_dir):
    _assert_rows(gpu_checks.check_cases(cases.drm_cases(), hip_backend, oracle_backend, 2e-5,
                                        np.load(os.path.join(golden_dir, "drm.npz"))), fixture_too=True)


def test_conv_layers_parity(hip_backend):
    _assert_rows(gpu_checks.check_conv_layers(hip_backend.dev))


def test_ifnet_glue_parity(hip_backend):
    _assert_rows(gpu_checks.check_glue(hip_backend.dev))


def test_scene_detection_parity(hip_backend, golden_dir):
    _assert_rows(gpu_checks.check_scdet(hip_backend, np.load(os.path.join(golden_dir, "scdet.npz"))))


@pytest.mark.parametrize("scale,size", cases.RIFE_CONFIGS)
def test_rife_end_to_end_parity(hip_backend, oracle_backend, golden_dir, scale, size):
    rows = gpu_checks.check_rife(hip_backend, oracle_backend, np.load(os.path.join(golden_dir, "rife.npz")), scale, size)
    frames = [r for r in rows if r[0].startswith(("ts_", "drba_cold", "drba_warm", "drba_nonlinear")) and "reuse" not in r[0]]
    _assert_rows(frames, fixture_too=True)  # synthesised frames: 1e-3 max-abs vs the oracle AND vs the reference's stored frames
    report.record(f"test_rife_end_to_end_parity flows/features scale={scale} size={size}", [r for r in rows if r not in frames])
    # flows / features: same bar except isolated hole-fill flips, which must stay rare
    for name, err, tol, extra in rows:
        if (name, err, tol, extra) in frames:
            continue
        n_out, n = map(int, extra.split("outliers>")[1].split(":")[1].split()[0].split("/"))
        assert n_out <= max(2, n // 2000), f"{name}: {n_out}/{n} elements off by more than {tol} (max {err:.3e})"


def test_ops_reject_cpu_tensors():
    from drba_amd import _lib, ops
    with pytest.raises(_lib.DrbaHipError):
        ops.flow_distance(torch.zeros(1, 2, 4, 4))


def test_gmfss_subnets_parity(hip_backend):
    """FeatureNet, MetricNet, GridNet, GMFlow and its stages, each fed the oracle's inputs."""
    _assert_rows(gpu_checks.check_gmfss_parts(hip_backend.dev))


@pytest.mark.parametrize("scale,size", cases.GMFSS_CONFIGS)
def test_gmfss_union_end_to_end_parity(hip_backend, oracle_backend, golden_dir, scale, size):
    """GMFSS_UNION through the reference call surface: every output (frames, flows, metrics, features) within
    1e-3 max-abs -- flat -- of the oracle and of the reference's own outputs in the fixture; at most 0.02 % of an
    output's elements (discontinuous splat / mask decisions) above it, none above 5e-2 (gpu_checks.check_gmfss_union)."""
    rows = gpu_checks.check_gmfss_union(hip_backend, oracle_backend, np.load(os.path.join(golden_dir, "gmfss_union.npz")),
                                        scale, size)
    _assert_rows(rows)


def test_gmfss_plain_end_to_end_parity(hip_backend, oracle_backend, golden_dir):
    rows = gpu_checks.check_gmfss_plain(hip_backend, oracle_backend, np.load(os.path.join(golden_dir, "gmfss.npz")))
    _assert_rows(rows)
    for name, _, tol, extra in rows:
        assert float(extra.split("vs_fixture=")[1]) <= tol, f"{name}: {extra}"


def test_trained_weights_parity(hip_backend, oracle_backend, golden_dir):
    """FeatureNet / MetricNet with the reference's trained weights, and GMFlow with un-damped LayerNorm gains."""
    _assert_rows(gpu_checks.check_trained(hip_backend, oracle_backend, np.load(os.path.join(golden_dir, "trained_union.npz")),
                                          golden_dir))


@pytest.mark.parametrize("mode", ("frame", "frame+ts", "frame+ts f3", "frame+other ts"))
def test_lookahead_flow_matches_inline(hip_backend, mode):
    """inference_ts_drba(..., lookahead=...) computes the next step's coarse flow (and, when the next timesteps are
    given, its DRM maps and low-resolution IFNet stages) on a side stream; the frames and the reuse state must equal
    the inline computation (same kernels, other stream).  A lookahead announced with other timesteps than the next
    call uses must be ignored for the stages (the coarse flow is still taken)."""
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    H, W = 128, 192
    fr = [f.to(hip_backend.dev) for f in cases.rife_frames(H, W)]
    ts = np.array([0.6, 1.0, 1.4]) if mode.endswith("f3") else np.array([0.75, 1.25])

    def run(look):
        m = hip_backend.make_rife(sd, 1.0)
        outs = []
        reuse = None
        for k in range(2):  # (f0,f1,f2) then (f1,f2,f3)
            nxt = fr[k + 3] if (look and k + 3 < len(fr)) else None
            if nxt is not None and mode != "frame":
                nxt = (nxt, np.array([0.8, 1.2]) if mode == "frame+other ts" else ts)
            o, reuse = m.inference_ts_drba(fr[k], fr[k + 1], fr[k + 2], ts, reuse, True, lookahead=nxt)
            if look and k == 0:
                assert m._look.pending is not None and m._look.pending[0] is fr[2] and m._look.pending[1] is fr[3]
                staged = m._look.pending[2][1]
                assert (staged is not None) == (mode != "frame")
            outs += o
        torch.cuda.synchronize()
        return outs, reuse, m

    a, ra, ma = run(True)
    b, rb, _ = run(False)
    assert ma._look.pending is None  # the second step consumed the lookahead and had no further frame
    assert len(a) == len(b)
    for x, y in zip(a + list(ra), b + list(rb)):
        assert float((x - y).abs().max()) <= 2e-6


def test_prefetched_encoder_matches_inline(hip_backend):
    """RIFE.prefetch_frame starts a frame's context encoder on its own stream as soon as the driver has read it (two
    frames ahead of its use as I2); calc_flow must pick those features up and every frame and the reuse state must
    equal the run without prefetching (same kernels, other stream; the splats' sums are order-dependent: 2e-6)."""
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    fr = [f.to(hip_backend.dev) for f in cases.rife_frames(128, 192)]
    ts = np.array([0.75, 1.25])

    def run(prefetch):
        m = hip_backend.make_rife(sd, 1.0)
        m.GROUP = 1  # the one-step machinery (with groups of steps the coarse flows are made where a group is staged)
        frames = [f.clone() for f in fr]  # fresh tensor objects: the prefetch is keyed by frame identity
        outs, reuse = [], None
        if prefetch:
            m.prefetch_frame(frames[2])
            m.prefetch_frame(frames[3])
            m.prefetch_pair(frames[2], frames[3])  # the pair the first call's lookahead starts from
        for k in range(2):
            if prefetch and k + 4 < len(frames):
                m.prefetch_frame(frames[k + 4])
            nxt = (frames[k + 3], ts) if k + 3 < len(frames) else None
            o, reuse = m.inference_ts_drba(frames[k], frames[k + 1], frames[k + 2], ts, reuse, True, lookahead=nxt)
            outs += o
        torch.cuda.synchronize()
        if prefetch:
            assert getattr(frames[3], "_drba_enc", None) is not None and frames[3]._drba_enc[0] is reuse[2]  # f of the last I2
            assert frames[3]._drba_pairflow[1][1] is reuse[0]  # flow21 of the last call came from prefetch_pair
        return outs, reuse

    a, ra = run(True)
    b, rb = run(False)
    for x, y in zip(a + list(ra), b + list(rb)):
        assert float((x - y).abs().max()) <= 2e-6


def test_scene_checks_do_not_retain_frames(hip_backend):
    """tools.SceneChecks remembers DECISIONS, not frames: once the driver has dropped a tested pair nothing in the table may
    keep the two frames (and the encoder features hung on them: 270 MB per 1080p frame) alive; an id recycled by a new frame
    is a miss, not a stale hit."""
    import gc
    import weakref
    from drba_amd.models.utils import tools
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(3)
    sc = tools.SceneChecks(0.3)
    a, b = torch.rand(1, 3, 64, 96, generator=g).to(dev), torch.rand(1, 3, 64, 96, generator=g).to(dev)
    key = (id(a), id(b))
    sc.submit(key, a, b)
    want = tools.check_scene(a, b, 0.3)
    assert sc.cut(key, a, b) == want and sc.cut(key, a, b) == want  # second call: the remembered decision
    assert not sc.pending
    ra, rb = weakref.ref(a), weakref.ref(b)
    del a, b
    gc.collect()
    assert ra() is None and rb() is None, "SceneChecks keeps tested frames alive"
    c, d = torch.rand(1, 3, 64, 96, generator=g).to(dev), torch.rand(1, 3, 64, 96, generator=g).to(dev)
    sc.done[(id(c), id(d))] = sc.done.pop(key)  # the old decision under the new pair's key: as if the ids had been recycled
    assert not sc._known((id(c), id(d)), c, d)
    assert sc.cut((id(c), id(d)), c, d) == tools.check_scene(c, d, 0.3)


def test_two_model_instances_in_one_process(hip_backend):
    """A second RIFE in the process (the per-device kernel attributes and the per-(device, stream) work counters are looked
    up, not assumed from the first instance) gives the first one's results."""
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(9)
    fr = [torch.rand(1, 3, 128, 192, generator=g).to(dev) for _ in range(3)]
    ts = np.array([0.75, 1.25])
    m1, m2 = hip_backend.make_rife(sd, 1.0), hip_backend.make_rife(sd, 1.0)
    o1, _ = m1.inference_ts_drba(*fr, ts, None, True)
    o2, _ = m2.inference_ts_drba(*fr, ts, None, True)
    torch.cuda.synchronize()
    # (the fused splats sum a key's records in arrival order and the first instance's first call runs layer by layer while
    # the autotuner decides, the second one through the chains: equal to rounding, not bit for bit)
    err = max(float((x - y).abs().max()) for x, y in zip(o1, o2))
    assert err <= 2e-5, err


def test_two_term_and_three_term_conv_families_agree(hip_backend, monkeypatch):
    """The conv autotuner's default candidate set includes kernel family 4 (fp32 operands as two fp16 terms, 22 bits); with
    CONV_FAMILIES = {0, 1, 2, 3} every MFMA operand keeps 24 bits (three bf16 terms).  One DRBA step both ways: the frames
    agree to the rounding level of two instances of ONE setting (test_two_model_instances_in_one_process) -- the operand
    bits family 4 drops are below the fp32 accumulation's own error."""
    from drba_amd import ops
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(21)
    fr = [torch.rand(1, 3, 256, 448, generator=g).to(dev) for _ in range(3)]
    ts = np.array([0.75, 1.25])
    outs = {}
    for name, fams in (("three-term", {0, 1, 2, 3}), ("two-term", {0, 1, 2, 3, 4})):
        monkeypatch.setattr(ops, "CONV_FAMILIES", set(ops.CONV_FAMILIES))  # (restored after the test)
        ops.set_precision(fams)  # no reset of the tuner's cache: winners are kept per (shape, family set)
        m = hip_backend.make_rife(sd, 1.0)
        m.inference_ts_drba(*fr, ts, None, True)          # tunes layer by layer
        o, _ = m.inference_ts_drba(*fr, ts, None, True)   # the chains, with the winners
        torch.cuda.synchronize()
        outs[name] = [x.clone() for x in o]
        fams_used = {ops._lib.load().drba_conv3x3_cfg_family(c) for (k, f), c in ops._tuned.items() if k[0] == "conv3x3" and set(f) == fams}
        assert (4 in fams_used) == (name == "two-term"), (name, fams_used)
    err = max(float((a - b).abs().max()) for a, b in zip(outs["three-term"], outs["two-term"]))
    assert err <= 2e-5, err


def test_family4_overflow_is_reported_by_default_without_a_sync(hip_backend):
    """Kernel family 4 holds an operand as fp16(x / 16) + ...: an activation of 65504 * 16 ~ 1.05e6 or more (an attention Q / V
    of 65504 or more) overflows where families 0-3 keep fp32's range.  ALWAYS ON (ABI 8): every family-4 kernel sets its byte of
    the device's host-mapped status word when a value it stores is inf / NaN -- no extra kernel, no synchronisation -- and
    ops.check_overflow (called by the model wrappers once per call and by tools.to_out behind its copy) raises.  A weight
    beyond fp16's range never reaches a family-4 kernel: its pack refuses it and the tuner is not offered family 4 for that
    layer.  The debug range check (drba_set_range_check / DRBA_CHECK_RANGE=1: a scan + a stream sync per call) still turns
    the overflow into an error at the call itself."""
    from drba_amd import _lib, ops
    lib = _lib.load()
    dev = hip_backend.dev
    ops.status_init(dev)
    torch.cuda.synchronize()
    ops.overflow_groups(dev)  # whatever earlier tests left
    g = torch.Generator().manual_seed(3)
    f4 = [c for c in range(lib.drba_conv3x3_num_cfgs()) if lib.drba_conv3x3_cfg_family(c) == 4 and lib.drba_conv3x3_cfg_stride(c) == 1
          and lib.drba_conv3x3_packed_floats(64, 64, c) > 0]
    f1 = [c for c in range(lib.drba_conv3x3_num_cfgs()) if lib.drba_conv3x3_cfg_family(c) == 1 and lib.drba_conv3x3_packed_floats(64, 64, c) > 0]
    assert f4 and f1
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    x_ok = (torch.rand(1, 64, 12, 40, generator=g) * 2 - 1) * 1.0e6
    x_big = x_ok.clone()
    x_big[0, 5, 3, 7] = 2.0e6   # one activation past 65504 * 16
    ref = torch.nn.functional.conv2d(x_big.double(), wt.double(), None, padding=1)

    def flagged(run):
        """Did `run` set a status byte?  (the word is read after the stream has drained: no kernel of `run` is still in flight)"""
        y = run()
        torch.cuda.synchronize()
        return ops.overflow_groups(dev), y

    for cfg in f4:  # the default path: the overflow is reported, below the bound nothing is
        conv = ops.Conv3x3(wt, torch.zeros(64), 1, None, None, device=dev, cfg=cfg)
        groups, y = flagged(lambda: conv(x_ok.to(dev)))
        assert groups == [] and bool(torch.isfinite(y).all()), (cfg, groups)
        groups, y = flagged(lambda: conv(x_big.to(dev)))
        assert len(groups) == 1 and groups[0].startswith("conv_"), (cfg, groups)
        assert not bool(torch.isfinite(y).all())
        conv(x_big.to(dev))
        torch.cuda.synchronize()
        with pytest.raises(_lib.DrbaHipError, match="family 4"):
            ops.check_overflow(dev)
        ops.check_overflow(dev)  # cleared by the raise: sticky until read, not beyond
        wb = wt.clone()
        wb[3, 4, 1, 1] = 1.0e5  # a weight past fp16's range (no pre-scale on weights): refused at pack time ...
        with pytest.raises(_lib.DrbaHipError, match="unsupported"):
            ops.Conv3x3(wb, torch.zeros(64), 1, None, None, device=dev, cfg=cfg)(x_ok.to(dev) * 1e-6)
    # ... and a layer built without a pinned configuration is tuned among the 24-bit families only
    wb = wt.clone()
    wb[3, 4, 1, 1] = 1.0e5
    auto = ops.Conv3x3(wb, torch.zeros(64), 1, None, None, device=dev)
    xs = x_ok * 1e-6
    ya = auto(xs.to(dev))
    refb = torch.nn.functional.conv2d(xs.double(), wb.double(), None, padding=1)
    assert float((ya.double().cpu() - refb).abs().max()) <= 5e-6 * float(refb.abs().max())
    assert all(lib.drba_conv3x3_cfg_family(c) != 4 for c in auto._keep)
    got = ops.Conv3x3(wt, torch.zeros(64), 1, None, None, device=dev, cfg=f1[0])(x_big.to(dev))  # 24-bit family: fp32's range
    torch.cuda.synchronize()
    assert ops.overflow_groups(dev) == []
    assert float((got.double().cpu() - ref).abs().max()) <= 5e-6 * float(ref.abs().max())
    # linear layers, attention, the fused stage kernel and the fused encoder report into their own bytes
    lin = ops.LinearSplit(torch.randn(128, 128, generator=g) / 11.0, None, device=dev, terms=2)
    t = torch.randn(256, 128, generator=g)
    groups, y = flagged(lambda: lin(t.to(dev)))
    assert groups == [] and bool(torch.isfinite(y).all())
    t[17, 5] = 2.0e6
    groups, _ = flagged(lambda: lin(t.to(dev)))
    assert groups == ["linear_split"], groups
    groups, y = flagged(lambda: ops.LinearSplit(torch.randn(128, 128, generator=g) / 11.0, None, device=dev, terms=3)(t.to(dev)))
    assert groups == [] and bool(torch.isfinite(y).all())
    q, k, v = [torch.randn(1, 32 * 32, 128, generator=g) for _ in range(3)]
    groups, _ = flagged(lambda: ops.window_attention(q.to(dev), k.to(dev), v.to(dev), 32, 32, 2, False, 128 ** -0.5, terms=2))
    assert groups == [], groups
    vb = v.clone()
    vb[0, 100, 7] = 7.0e4  # an attention V past 65504
    groups, _ = flagged(lambda: ops.window_attention(q.to(dev), k.to(dev), vb.to(dev), 32, 32, 2, False, 128 ** -0.5, terms=2))
    assert groups == ["window_attention"], groups
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    head = Head(sd, "encode.", dev)
    img = torch.rand(1, 3, 64, 96, generator=g)
    groups, _ = flagged(lambda: head(img.to(dev), planar=False))
    assert groups == [], groups
    imgb = img.clone()
    imgb[0, 1, 20, 30] = 3.0e7  # (the encoder's first layer sees |x| * |w| summed: far beyond 65504 * 16)
    groups, _ = flagged(lambda: head(imgb.to(dev), planar=False))
    assert groups == ["head_fused16"], groups
    sd_big = {kk: vv.clone() for kk, vv in sd.items()}
    sd_big["encode.cnn1.weight"][2, 3, 1, 1] = 1.0e5  # an encoder weight beyond fp16: the exact-fp32 fused encoder takes the layer
    hb = Head(sd_big, "encode.", dev)
    groups, fb = flagged(lambda: hb(img.to(dev), planar=False))
    assert groups == [] and bool(torch.isfinite(fb).all())
    # the debug range check still raises at the call
    was = lib.drba_set_range_check(1)
    try:
        conv = ops.Conv3x3(wt, torch.zeros(64), 1, None, None, device=dev, cfg=f4[0])
        with pytest.raises(_lib.DrbaHipError, match="unsupported"):
            conv(x_big.to(dev))
    finally:
        lib.drba_set_range_check(was)
    torch.cuda.synchronize()
    ops.overflow_groups(dev)


def test_model_wrappers_raise_on_family4_overflow(hip_backend):
    """RIFE end to end: a frame far outside [0, 1] overflows the two-term fp16 kernels; the wrapper's per-call check (and
    tools.to_out behind its copy) raises instead of handing inf / NaN frames on -- and says which kernels; with the family
    switched off (ops.set_precision({0, 1, 2, 3})) the same input goes through in fp32's range."""
    from drba_amd import _lib, ops
    from drba_amd.models.rife import RIFE
    from drba_amd.models.utils import tools
    from drba_amd.utils import synth
    dev = hip_backend.dev
    torch.cuda.synchronize()
    ops.overflow_groups(dev)
    sd = synth.ifnet_state_dict(seed=0)
    m = RIFE(weights=sd, scale=1.0, device=dev)
    clip = synth.make_clip(3, 128, 192, seed=1234)
    fr = [torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float().div(255.0).to(dev) for f in clip]
    ts = np.array([0.75, 1.25])
    out, reuse = m.inference_ts_drba(fr[0], fr[1], fr[2], ts, None, True)
    assert tools.to_out(out[0], (128, 192)).shape == (128, 192, 3)  # in range: nothing raised
    bad = [f.clone() for f in fr]
    bad[1][0, 2, 40, 50] = 1.0e9
    out, _ = m.inference_ts_drba(bad[0], bad[1], bad[2], ts, None, True)
    with pytest.raises(_lib.DrbaHipError, match="family 4"):
        tools.to_out(out[0], (128, 192))
    m.inference_ts_drba(bad[0], bad[1], bad[2], ts, None, True)
    torch.cuda.synchronize()
    with pytest.raises(_lib.DrbaHipError, match="head_fused16|stage_conv16|conv_"):
        m.inference_ts_drba(fr[0], fr[1], fr[2], ts, None, True)  # the NEXT call sees what the previous one left
    out, _ = m.inference_ts_drba(fr[0], fr[1], fr[2], ts, None, True)  # (cleared by the raise)
    torch.cuda.synchronize()
    assert ops.overflow_groups(dev) == []


def test_fused_stage_kernels_against_fp64(hip_backend):
    """stage_conv16 (scale 1) and stage_conv16_s2 (scale 2): the stage input fused with conv0[0] in the two-term fp16 form, against
    an fp64 convolution of the unfused stage input (ops.stage_inputs) at 5e-6 max|y| -- ragged sizes, tiles cut by the border, the
    fold / the finished flow / the flow as terms, 16 and 32 output channels, smooth and rough flows, 1088x1920 -- and the folded
    flows bit-identical to ifblock_input_lds' (tools/stage_conv16_check.py holds the cases; it exits non-zero on any failure)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stage_conv16_check.py"), "--no-time"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FAIL" not in r.stdout and "all ok" in r.stdout and "scale 2 cout 32" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_drm_maps_of_a_group_in_one_launch_equal_the_single_calls(hip_backend):
    """drba_drm_rife_linear_batch (the DRM maps of a group of steps: one launch pair) against drba_drm_rife_linear per map:
    the same kernels on the same inputs, for smooth, long (beyond the tile halo) and non-finite flows, a ragged size, more jobs
    than one launch takes.  Equal to summation order: a tile ranks the sources of an output pixel by an LDS atomic, long flows
    go through global atomics -- two runs of ONE call differ in the last bits as well (2e-6 on maps of magnitude <= 1)."""
    from drba_amd import ops
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(17)
    for (h, w, amp, n) in ((70, 90, 3.0, 3), (128, 256, 40.0, 8), (64, 96, 2.0, 11)):
        jobs = []
        for k in range(n):
            a, b = torch.randn(1, 2, h, w, generator=g) * amp, torch.randn(1, 2, h, w, generator=g) * amp
            if k == 1:
                a[0, 0, 5, 7], b[0, 1, 9, 3] = float("nan"), float("inf")
            jobs.append((a.to(dev), b.to(dev), 0.1 + 0.07 * k))
        many = ops.drm_rife_linear_many(jobs, 1e-4)
        for (a, b, t), m in zip(jobs, many):
            one = ops.drm_rife_linear(a, b, t, 1e-4)
            assert torch.equal(torch.isnan(one), torch.isnan(m)), (h, w, amp, t)
            d = float((torch.nan_to_num(one, nan=0.0) - torch.nan_to_num(m, nan=0.0)).abs().max())
            assert d <= 2e-6 * max(1.0, float(torch.nan_to_num(one, nan=0.0).abs().max())), (h, w, amp, t, d)


def test_splat_index_reuse_is_validated(hip_backend):
    """softsplat_many(reuse_index=True) reuses the sorted index of the previous splat only while the workspace still holds it
    for the same (flow, metric, mode, geometry): another user of the stream's workspace in between, or another flow, makes the
    call rebuild the index instead of gathering through a stale one (it used to trust the caller)."""
    from drba_amd import ops
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(23)
    x, y = torch.rand(1, 3, 48, 80, generator=g).to(dev), torch.rand(1, 5, 48, 80, generator=g).to(dev)
    f1, f2 = (torch.randn(1, 2, 48, 80, generator=g) * 3).to(dev), (torch.randn(1, 2, 48, 80, generator=g) * 3).to(dev)
    def same(a, b):  # (the sort ranks a pixel's sources by an atomic: two runs of one call agree to summation order)
        return float((a - b).abs().max()) <= 2e-6

    ref_y1 = ops.softsplat(y, f1, None, "avg")
    ref_y2 = ops.softsplat(y, f2, None, "avg")
    ops.softsplat(x, f1, None, "avg")
    assert same(ops.softsplat_many([y], f1, None, "avg", reuse_index=True)[0], ref_y1)      # the honest reuse
    assert same(ops.softsplat_many([y], f2, None, "avg", reuse_index=True)[0], ref_y2)      # another flow: rebuilt
    ops.softsplat(x, f1, None, "avg")
    ops.instance_norm(torch.rand(1, 4, 48, 80, generator=g).to(dev))                                # another workspace user
    assert same(ops.softsplat_many([y], f1, None, "avg", reuse_index=True)[0], ref_y1)
    ops.softsplat(x, f1, None, "avg")
    f1.add_(0.25)                                                                                   # the flow changed in place
    assert same(ops.softsplat_many([y], f1, None, "avg", reuse_index=True)[0], ops.softsplat(y, f1, None, "avg"))


def test_cloned_reuse_features_keep_their_layout(hip_backend):
    """The carried encoder features are pair-interleaved [8,H,W,2] tensors (ops.head_fused(planar=False)); a caller that
    clones the `reuse` tuple loses the tag on them, and a planar [1,16,H,W] tensor (the reference's layout, e.g. an oracle's
    reuse) may come in as well: both must be read in the layout they are in."""
    from drba_amd import ops
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(13)
    fr = [torch.rand(1, 3, 128, 192, generator=g).to(dev) for _ in range(4)]
    ts = np.array([0.75, 1.25])
    m = hip_backend.make_rife(sd, 1.0)
    _, reuse = m.inference_ts_drba(fr[0], fr[1], fr[2], ts, None, True)
    assert ops.is_pair(reuse[2]) and tuple(reuse[2].shape) == (8, 128, 192, 2)
    want, _ = m.inference_ts_drba(fr[1], fr[2], fr[3], ts, reuse, True)
    cloned = tuple(t.clone() for t in reuse)                                             # tags gone, shapes tell
    planar = (reuse[0], reuse[1], ops.features_planar(reuse[2]), ops.features_planar(reuse[3]))  # the reference's layout
    for name, r in (("cloned", cloned), ("planar", planar)):
        got, _ = hip_backend.make_rife(sd, 1.0).inference_ts_drba(fr[1], fr[2], fr[3], ts, r, True)
        torch.cuda.synchronize()
        err = max(float((a - b).abs().max()) for a, b in zip(got, want))
        assert err <= 2e-5, (name, err)


def test_prefetch_does_not_retain_frames(hip_backend):
    """The prefetch caches hang on the frame tensors (encoder output on the frame, coarse flow on the pair's second frame);
    nothing may keep a frame -- with its 16-channel features -- alive once the driver has dropped it: memory is flat over a
    long run of the two-frames-ahead loop."""
    import gc
    from drba_amd.utils import synth
    m = hip_backend.make_rife(synth.ifnet_state_dict(seed=0), 1.0)
    m.GROUP = 1
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(5)
    base = [torch.rand(1, 3, 128, 192, generator=g).to(dev) for _ in range(4)]
    ts = np.array([0.75, 1.25])

    def frame(k):
        return torch.roll(base[k % 4], k // 4, dims=3).contiguous()  # a fresh tensor object per source frame

    win = [frame(k) for k in range(5)]  # I0, I1, I2, next, next2
    for x in win[2:]:
        m.prefetch_frame(x)
    m.prefetch_pair(win[2], win[3])
    m.prefetch_pair(win[3], win[4])
    reuse, marks = None, {}
    for k in range(60):
        _, reuse = m.inference_ts_drba(win[0], win[1], win[2], ts, reuse, True, lookahead=(win[3], ts))
        nxt2 = frame(k + 5)
        m.prefetch_frame(nxt2)
        m.prefetch_pair(win[4], nxt2)
        win = win[1:] + [nxt2]
        if k in (19, 59):
            torch.cuda.synchronize()
            gc.collect()
            marks[k] = torch.cuda.memory_allocated(dev)
    assert marks[59] <= marks[19] + (1 << 20), marks  # 40 more steps: not one frame's worth (0.3 MB + 3.1 MB features) each


def test_gmfss_union_lookahead_matches_inline(hip_backend):
    """Same for GMFSS_UNION: the pair state model.reuse(I2, next) prefetched on the side stream (and the per-frame
    FeatureNet cache) must give the frames of the inline computation."""
    from drba_amd.utils import synth
    sds = synth.gmfss_union_state_dicts(seed=0)
    fr = [f.to(hip_backend.dev) for f in cases.gmfss_frames(128, 256)]
    ts = np.array([0.75, 1.25])

    def run(look):
        frames = [f.clone() for f in fr]  # fresh tensors: no cached per-frame state from the other run
        m = hip_backend.make_gmfss_union(sds, 1.0)
        outs, reuse = [], None
        for k in range(2):
            nxt = frames[k + 3] if (look and k + 3 < len(frames)) else None
            o, reuse = m.inference_ts_drba(frames[k], frames[k + 1], frames[k + 2], ts, reuse, True, lookahead=nxt)
            outs += o
        torch.cuda.synchronize()
        return outs

    for x, y in zip(run(True), run(False)):
        assert float((x - y).abs().max()) <= 1e-5


def test_gmflow_bidirectional_equals_two_calls(hip_backend):
    """GMFlow.bidirectional shares the encoder and the coarsest transformer pass between the two directions; it must
    return what two separate forward calls return."""
    from drba_amd.models.gmflow.gmflow import GMFlow
    from drba_amd.utils import synth
    import torch.nn.functional as F
    net = GMFlow(synth.gmfss_union_state_dicts(seed=0)["flownet"], hip_backend.dev)
    I0, I1 = [F.interpolate(f, scale_factor=0.5, mode="bilinear", align_corners=False).to(hip_backend.dev)
              for f in cases.gmfss_frames(128, 256)[:2]]
    a, b = net.bidirectional(I0, I1)
    a2, b2 = net(I0, I1), net(I1, I0)
    assert float((a - a2).abs().max()) <= 1e-5 and float((b - b2).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_window_attention(hip_backend):
    _assert_rows(gpu_checks.check_window_attention(hip_backend.dev))


@pytest.mark.gpu
def test_global_expect2(hip_backend):
    _assert_rows(gpu_checks.check_global_expect2(hip_backend.dev))


@pytest.mark.gpu
def test_linear_split(hip_backend):
    _assert_rows(gpu_checks.check_linear_split(hip_backend.dev))


@pytest.mark.gpu
def test_feature_splat_quad_source(hip_backend):
    _assert_rows(gpu_checks.check_splat_quad(hip_backend.dev))


@pytest.mark.gpu
def test_swap_select_in_place_equals_the_reference_assignment(hip_backend):
    """GMFSS.py:137-150: x[m0], y[m1] = y[m0], x[m1] with m0 = t0 / t1 > 25, m1 = t1 / t0 > 25 -- bit-exact (a selection), out of
    place and in place (the form GMFSS uses: the splats wrote the destination slices, only selected pixels move), including pixels
    where both masks hold (negative ratios cannot, zeros can: x / 0 = inf > 25 on one side, 0 / x = 0 on the other)."""
    from drba_amd import ops
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(11)
    c, h, w = 7, 19, 33
    x, y = torch.randn(1, c, h, w, generator=g), torch.randn(1, c, h, w, generator=g)
    t0 = torch.rand(1, 1, h, w, generator=g) + 0.01
    t1 = torch.rand(1, 1, h, w, generator=g) + 0.01
    t0[0, 0, 3:6, 4:20] *= 100.0   # m0 region
    t1[0, 0, 10:15, 8:30] *= 100.0  # m1 region
    t1[0, 0, 0, 0:5] = 0.0          # t0 / 0 = inf
    m0, m1 = (t0 / t1 > 25).repeat(1, c, 1, 1), (t1 / t0 > 25).repeat(1, c, 1, 1)
    assert int(m0.sum()) > 0 and int(m1.sum()) > 0
    rx, ry = x.clone(), y.clone()
    rx[m0], ry[m1] = y[m0], x[m1]
    gx, gy = ops.swap_select(x.to(dev), y.to(dev), t0.to(dev), t1.to(dev), 25.0)
    assert torch.equal(gx.cpu(), rx) and torch.equal(gy.cpu(), ry)
    buf = torch.cat([x, y], dim=1).to(dev)  # in place, on channel slices of one buffer
    xs, ys = buf[:, :c], buf[:, c:]
    ox, oy = ops.swap_select(xs, ys, t0.to(dev), t1.to(dev), 25.0, out=(xs, ys))
    assert ox.data_ptr() == xs.data_ptr()
    assert torch.equal(buf[:, :c].cpu(), rx) and torch.equal(buf[:, c:].cpu(), ry)


@pytest.mark.gpu
def test_kept_quad_source_equals_the_rewritten_one(hip_backend):
    """softsplat(..., keep_quad=True) gathers a feature tensor from the interleaved copy kept on it (drba_softsplat_index +
    drba_softsplat_gather_quad): same kernels on the same values as drba_softsplat's own copy -- equal bit for bit up to the
    gather's segment order (the sort's slots are claimed by atomics), i.e. to rounding; a tensor rewritten in place gets a new copy."""
    from drba_amd import ops
    dev = hip_backend.dev
    g = torch.Generator().manual_seed(5)
    for (c, h, w) in ((64, 37, 52), (20, 16, 24)):
        x = torch.randn(1, c, h, w, generator=g).to(dev)
        flow = (torch.randn(1, 2, h, w, generator=g) * 3.0).to(dev)
        z = torch.randn(1, 1, h, w, generator=g).to(dev)
        ref = ops.softsplat(x, flow, z, "soft")
        got = ops.softsplat(x, flow, z, "soft", keep_quad=True)
        assert getattr(x, "_drba_quad", None) is not None
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        kept = x._drba_quad[0]
        assert ops.softsplat(x, flow, z, "soft", keep_quad=True) is not None and x._drba_quad[0] is kept  # reused, not remade
        x.mul_(2.0)  # a torch in-place write bumps the version: the copy is remade
        got2 = ops.softsplat(x, flow, z, "soft", keep_quad=True)
        assert x._drba_quad[0] is not kept
        assert float((got2 - 2.0 * ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    small = torch.randn(1, 3, 9, 12, generator=g).to(dev)  # not a feature tensor: the ordinary path, nothing kept
    fl = torch.zeros(1, 2, 9, 12, device=dev)
    ops.softsplat(small, fl, None, "avg", keep_quad=True)
    assert getattr(small, "_drba_quad", None) is None


@pytest.mark.parametrize("scale,n_ts", [(1.0, 7), (1.0, 10), (2.0, 2), (2.0, 5)])
def test_rife_many_timesteps_and_model_scale_above_one(hip_backend, oracle_backend, scale, n_ts):
    """A step with more frames to synthesise than one batched glue launch takes (DRBA_MAX_STAGE_ITEMS = 8: the 10-timestep case;
    `-t 8` has 7) runs as independent groups whose carried state -- the flow terms -- is re-joined, and with a model scale > 1 (scale_list ends at 0.5: the last stage runs ABOVE the frame
    resolution) the frame is finished by the plain flow update + warp_blend pair; both against the oracle at 1e-3, cold
    and warm step (reference rife.py:77-109 takes any number of timesteps and any scale)."""
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    fr = cases.rife_frames(128, 192)
    ts = np.linspace(0.5, 1.5, n_ts + 2)[1:-1]
    ts = ts[ts != 1.0] if n_ts % 2 == 0 else ts  # an odd count keeps t = 1 (pass-through frame)
    hip, ora = hip_backend.make_rife(sd, scale), oracle_backend.make_rife(sd, scale)
    dev = hip_backend.dev
    g = [f.to(dev) for f in fr]
    out, reuse = hip.inference_ts_drba(g[0], g[1], g[2], ts, None, True)
    out2, _ = hip.inference_ts_drba(g[1], g[2], g[3], ts, reuse, True)
    ref, rre = ora.inference_ts_drba(fr[0], fr[1], fr[2], ts, None, True)
    ref2, _ = ora.inference_ts_drba(fr[1], fr[2], fr[3], ts, rre, True)
    assert len(out + out2) == len(ref + ref2) == 2 * len(ts)
    errs = [float((a.cpu() - b).abs().max()) for a, b in zip(out + out2, ref + ref2)]
    assert max(errs) <= 1e-3, errs
    plain = hip.inference_ts(g[0], g[1], list(np.linspace(0, 1, n_ts + 2)))
    pref = ora.inference_ts(fr[0], fr[1], list(np.linspace(0, 1, n_ts + 2)))
    assert max(float((a.cpu() - b).abs().max()) for a, b in zip(plain, pref)) <= 1e-3


@pytest.mark.parametrize("ts_name,group", (("t2", 2), ("t2", 4), ("fps60", 4), ("fps60", 3), ("t2", 8)))
def test_step_groups_match_single_steps(hip_backend, ts_name, group):
    """With the driver announcing frames ahead, RIFE computes GROUP consecutive DRBA steps in one stacked IFNet pass (2 GROUP
    samples per launch), stages the low-resolution part of the NEXT group on the side stream, and the later calls of a
    group only collect their results.  Frames and the carried reuse state must equal the step-by-step computation (same
    kernels on other batch sizes: the autotuner may pick another tiling for N = 8, hence 1e-5 instead of bit equality).
    The clip ends inside a group and the driver's entries run out before it: partial groups, unstaged groups."""
    from drba_amd.utils import synth
    sd = synth.ifnet_state_dict(seed=0)
    H, W = 128, 192
    base = cases.rife_frames(H, W)
    g = torch.Generator().manual_seed(3)
    # (group 8 = 16 samples per pass, more than one glue launch takes: the chunked glue launches around whole-batch convolution
    # chains of IFNet._forward_pairs_lazy; a longer clip so that several such groups form)
    frames = [f.to(hip_backend.dev) for f in base] + [torch.rand(1, 3, H, W, generator=g).to(hip_backend.dev) for _ in range(11 if group <= 4 else 26)]
    ts_seq = [np.array([0.75, 1.25])] * 32 if ts_name == "t2" else [np.array([0.6, 1.0, 1.4]), np.array([0.8, 1.2])] * 16

    def run(grp):
        m = hip_backend.make_rife(sd, 1.0)
        m.GROUP = grp
        depth = max(3, 2 * grp - 1)
        fr = [f.clone() for f in frames]  # fresh tensor objects: the caches are keyed by frame identity
        nf = len(fr)
        for j in range(2, min(2 + depth + 1, nf)):
            m.prefetch_frame(fr[j])
            if j > 2:
                m.prefetch_pair(fr[j - 1], fr[j])
        outs, reuse, collected = [], None, 0
        for k in range(nf - 2):
            j = k + 3 + depth
            if j < nf:
                m.prefetch_frame(fr[j])
                m.prefetch_pair(fr[j - 1], fr[j])
            ahead = list(range(k + 3, min(k + 3 + depth, nf)))
            look = None
            if ahead:
                look = (fr[ahead[0]], ts_seq[k + 1])
                if len(ahead) >= 2:
                    look = tuple(v for i, a in enumerate(ahead) for v in (fr[a], ts_seq[k + 1 + i]))
            was_cached = len(m._group_out) > 0
            o, reuse = m.inference_ts_drba(fr[k], fr[k + 1], fr[k + 2], ts_seq[k], reuse, True, lookahead=look)
            collected += int(was_cached)
            outs += o
        torch.cuda.synchronize()
        return outs, reuse, collected

    a, ra, n_collected = run(group)
    b, rb, none = run(1)
    assert n_collected >= group and none == 0, (n_collected, none)  # groups were formed (the first call is cold: reuse is None)
    assert len(a) == len(b)
    for x, y in zip(a + list(ra), b + list(rb)):
        assert float((x - y).abs().max()) <= 1e-5
