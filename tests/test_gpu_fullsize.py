"""GPU (-m gpu): parity AT THE BENCHMARKED SIZES, driven the way bench.py drives the model.

The small-size end-to-end cases (tests/cases.py, 128x192 .. 256x512) never reach the code the headline number runs:
the persistent multi-tile loop of conv_split_mfma (more than 768 tiles), the XCD-banded gathers at 2 088 960 pixels, the
three side-stream stages of the lookahead at N=2, drba_conv_chain on the 1080p shapes.  These tests do:

  * RIFE 1088x1920 scale 1.0 (1 cold + 9 warm steps) and 2176x3840 scale 0.5 (1 + 8) driven by bench.AnnouncedLoop -- the
    object bench.py's timed region runs: 7 frames read ahead, encoders / coarse flows prefetched, the calls announced so
    that the model computes groups of 4 steps (8 samples per launch) and stages the following group's low-resolution
    stages on the side stream --, ts = [0.75, 1.25] (-t 2) and the [0.6, 1.0, 1.4] / [0.8, 1.2] alternation (-fps 60);
    every synthesised frame and the carried reuse state against RifeOracle on identical fp32 inputs, 1e-3 max-abs; the
    model's path counters prove that one group was computed in place and one came from the side stream;
  * the layers of that path at the batch it launches them with (N = 8): split-bf16 convolutions, stage_conv0, the lazy
    gathers and warp_blend_lazy with 8 items at 1088x1920 against fp64 (tests/gpu_checks.py);
  * GMFSS_UNION 1152x1920: one warm step against GmfssUnionOracle (bar of gpu_checks.check_gmfss_union);
  * every split-bf16 convolution configuration on shapes with more than 768 tiles against an fp64 convolution.

The oracle needs ~5 s per 1080p RIFE step and ~1 min per GMFSS_UNION step on the GPU box's host cores.
"""
import numpy as np
import pytest
import torch

from drba_amd.utils import synth
from tests import gpu_checks

pytestmark = pytest.mark.gpu

TS_T2 = np.array([0.75, 1.25])
TS_F3 = np.array([0.6, 1.0, 1.4])
TS_F2 = np.array([0.8, 1.2])


def _net_frames(n, H, W, net, seed=1234):
    """bench.py's synthetic uint8 clip -> fp32 frames at the network size, converted ONCE on the CPU (oracle resize) so
    that the HIP path and the oracle see bit-identical inputs; to_inp itself is checked separately below."""
    import bench
    import oracle
    out = []
    for f in bench.make_frames_u8(n, H, W, seed=seed):
        x = torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0
        out.append(oracle.ops.resize(x, net) if tuple(net) != (H, W) else x)
    return out


def _drive(model, frames, ts_seq, dev, announced):
    """announced=True: bench.AnnouncedLoop, THE loop of bench.py's timed region (7 frames read ahead, prefetch_frame /
    prefetch_pair, the following calls named so that the model computes groups of RIFE.GROUP steps and stages the next
    group on the side stream).  announced=False: the reference's call pattern (the oracle).  Step k =
    inference_ts_drba(f[k], f[k+1], f[k+2], ts_seq[k], reuse, linear=True); the first step is cold (reuse=None).
    -> (per-step synthesised frames, per-step reuse)."""
    fr = [f.to(dev) for f in frames]
    outs, reuses = [], []
    if announced:
        import bench
        loop = bench.AnnouncedLoop(model, lambda k: fr[k] if k < len(fr) else None, lambda k: ts_seq[k] if k < len(ts_seq) else None)
        for _ in ts_seq:
            o, ts = loop.step()
            outs.append([x for x, t in zip(o, ts) if t not in (0.0, 1.0, 2.0)])
            reuses.append(loop.reuse)
    else:
        reuse = None
        for k, ts in enumerate(ts_seq):
            o, reuse = model.inference_ts_drba(fr[k], fr[k + 1], fr[k + 2], ts, reuse, True)
            outs.append([x for x, t in zip(o, ts) if t not in (0.0, 1.0, 2.0)])
            reuses.append(reuse)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return outs, reuses


def _rife_fullsize(hip_backend, oracle_backend, src, net, scale, ts_seq, min_groups, min_staged):
    sd = synth.ifnet_state_dict(seed=0)
    frames = _net_frames(len(ts_seq) + 2, src[0], src[1], net)
    hip = hip_backend.make_rife(sd, scale)
    got, greuse = _drive(hip, frames, ts_seq, hip_backend.dev, announced=True)
    # what was checked is the grouped path the benchmark times: the first step is cold (one-step path), then one group
    # computed in place and -- for the longer sequences -- groups whose low-resolution stages ran on the side stream
    st = dict(hip.stats)
    assert st["groups_formed"] >= min_groups and st["staged_groups"] >= min_staged and st["groups_dropped"] == 0, st
    assert st["single_steps"] + st["group_collects"] + st["groups_formed"] == len(ts_seq), st
    want, oreuse = _drive(oracle_backend.make_rife(sd, scale), frames, ts_seq, torch.device("cpu"), announced=False)
    rows = []
    for k, (g, o) in enumerate(zip(got, want)):
        assert len(g) == len(o)
        for j, (a, b) in enumerate(zip(g, o)):
            rows.append((f"step{k} frame{j} ({'cold' if k == 0 else 'warm'})", gpu_checks._diff(a, b), 1e-3, ""))
    # reuse = (flow21, flow12, f2, f1) after the LAST step (every step's reuse feeds the next step's frames, which are
    # checked above): features to 1e-3; flows carry the hole-fill discontinuity -> outlier budget
    from tests.cases import planar
    for name, a, b in zip(("flow21", "flow12", "f2", "f1"), greuse[-1], oreuse[-1]):
        a = planar(a)  # (the HIP path carries the features pair-interleaved)
        n_out, n = gpu_checks._outliers(a, b, 1e-3)
        rows.append((f"reuse {name}", 0.0 if n_out <= max(2, n // 2000) else gpu_checks._diff(a, b), 1e-3, f"outliers {n_out}/{n}"))
    rows.append(("path: " + ", ".join(f"{k}={v}" for k, v in st.items() if v), 0.0, 0.0, ""))
    return rows


def _assert_rows(rows):
    import inspect
    from tests import report
    report.record(inspect.stack()[1].function, rows)
    for r in rows:
        print(f"{r[0]:<40} err={r[1]:.3e} tol={r[2]:.1e} {r[3]}")
    bad = [r for r in rows if not r[1] <= r[2]]
    assert not bad, "\n".join(f"{n}: err={e:.3e} tol={t:.1e} {x}" for n, e, t, x in bad)


@pytest.mark.parametrize("ts_name", ("t2", "fps60"))
def test_rife_1080p_bench_loop_parity(hip_backend, oracle_backend, ts_name):
    """BASELINE.json configs[1] (-t 2) and configs[2] (-fps 60 timesteps) at 1088x1920, scale 1.0."""
    # 1 cold + 9 warm steps: steps 1-4 one group computed in place (while it runs, steps 5-8 are staged on the side stream),
    # steps 5-8 the staged group, step 9 the clip's last step on the one-step path
    ts_seq = [TS_T2] * 10 if ts_name == "t2" else [TS_F3, TS_F2] * 5
    _assert_rows(_rife_fullsize(hip_backend, oracle_backend, (1080, 1920), (1088, 1920), 1.0, ts_seq, 2, 1))


def test_rife_4k_half_scale_bench_loop_parity(hip_backend, oracle_backend):
    """BASELINE.json configs[4]'s per-GPU work: 2176x3840, scale 0.5, -fps 60 timesteps; one cold + 8 warm steps (one group
    computed in place, one staged on the side stream)."""
    _assert_rows(_rife_fullsize(hip_backend, oracle_backend, (2160, 3840), (2176, 3840), 0.5, [TS_F3, TS_F2] * 4 + [TS_F3], 2, 1))


def test_to_inp_to_out_fullsize_bit_exact(hip_backend):
    """to_inp / to_out (one fused kernel each) at 1080p <-> 1088x1920, 4K <-> 2176x3840 and 1080p <-> 1152x1920
    (tools.py:33-38,59-72): every fp32 value and every uint8 byte equal to the oracle's (= ATen's CPU kernels)."""
    import bench
    import oracle
    from drba_amd.models.utils import tools
    for (src, net) in (((1080, 1920), (1088, 1920)), ((2160, 3840), (2176, 3840)), ((1080, 1920), (1152, 1920))):
        f = bench.make_frames_u8(1, src[0], src[1], seed=1234)[0]
        x = torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0
        want = oracle.ops.resize(x, net)
        got = tools.to_inp(f, net)
        assert torch.equal(got.cpu(), want), (src, net, float((got.cpu() - want).abs().max()))
        back = tools.to_out(got, src)
        ref = (oracle.ops.resize(want, src)[0].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
        assert np.array_equal(back, ref), (src, net, int(np.abs(back.astype(np.int32) - ref.astype(np.int32)).max()))
        assert np.array_equal(tools.to_out(got, src, rgb=True), ref[:, :, ::-1])


def test_gmfss_union_1080p_warm_step_parity(hip_backend, oracle_backend):
    """BASELINE.json configs[3] at 1152x1920 (pad 128), scale 1.0: the reuse entering the step is rebuilt with
    warm_reuse (what the previous DRBA step would have returned), then one warm inference_ts_drba, ts = [0.75, 1.25].
    Bar: 1e-3 max-abs, flat (the synthetic GMFlow weights are well conditioned since round 5: the oracle's own frame moves by
    6e-5 under a 1-ulp input change at this size, tools/exp/union_floor_probe.py); at most 0.02 % of a tensor's elements above
    it (splat / mask decisions), none above 5e-2.  Flows are compared in pixels: up to 20 px here."""
    sds = synth.gmfss_union_state_dicts(seed=0)
    frames = _net_frames(3, 1080, 1920, (1152, 1920), seed=4321)

    def run(b, fr):
        m = b.make_gmfss_union(sds, 1.0)
        fr = [f.to(b.dev) for f in fr]
        out, new = m.inference_ts_drba(fr[0], fr[1], fr[2], TS_T2, m.warm_reuse(fr[0], fr[1]), True)
        if b.dev.type == "cuda":
            torch.cuda.synchronize()
        return {"frame0": out[0], "frame1": out[1], "flow21": new[0], "flow12": new[1], "metric2": new[2], "metric1": new[3]}

    with torch.no_grad():
        g = run(hip_backend, frames)
        o = run(oracle_backend, frames)
    rows = []
    for k in o:
        d = gpu_checks._diff(g[k], o[k])
        tk = 1e-3
        n_out, n = gpu_checks._outliers(g[k], o[k], tk)
        ok = n_out <= n // 5000 and d <= 5e-2
        rows.append((k, gpu_checks.Budgeted(d, ok, n_out, n) if d > tk else d, tk, f"max={d:.2e} outliers>{tk:.2g}: {n_out}/{n} |ref|max={float(o[k].abs().max()):.3g}"))
    _assert_rows(rows)


def test_gmfss_union_1080p_teacher_forced_stages(hip_backend):
    """BASELINE.json configs[3] at 1152x1920, stage by stage on the oracle's intermediate tensors: a check that CAN fail at
    1e-4 * max|ref| (a tighter, per-stage bar beside the end-to-end 1e-3 row above)."""
    frames = _net_frames(3, 1080, 1920, (1152, 1920), seed=4321)
    _assert_rows(gpu_checks.check_gmfss_union_teacher_forced(hip_backend.dev, frames))


def test_split_conv_configs_on_many_tile_shapes(hip_backend):
    """conv_split_mfma's persistent workgroups walk more than one tile only when a launch has more tiles than resident
    workgroups (768): the 1080p layer shapes.  Every split configuration (conv cfg >= 14, deconv cfg >= 6) on those
    shapes against an fp64 convolution, 5e-6 * max|y|."""
    import torch.nn.functional as F
    from drba_amd import ops
    dev = hip_backend.dev
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(321)
    rows = []
    # nb = 8: the batch the benchmarked loop launches these layers with (groups of 4 steps x 2 frames; the autotuner picks a
    # configuration per batch size, the forced loop below runs all of them)
    for (nb, cin, cout, h, w, kind) in ((2, 32, 32, 272, 480, "res"), (2, 64, 64, 136, 240, "res"), (2, 96, 96, 68, 120, "res"),
                                        (1, 32, 16, 544, 960, "conv"), (1, 64, 64, 288, 960, "pre"),
                                        (8, 32, 32, 272, 480, "res"), (8, 64, 64, 136, 240, "res"), (8, 96, 96, 68, 120, "res"),
                                        (8, 128, 128, 34, 60, "res"), (8, 192, 192, 17, 30, "res")):
        x = torch.randn(nb, cin, h, w, generator=g) * 2.0
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        beta = torch.rand(1, cout, 1, 1, generator=g) + 0.5
        xd = x.double()
        if kind == "res":
            ref = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1) * beta.double() + xd, 0.2)
        elif kind == "pre":
            ref = F.conv2d(F.prelu(xd, torch.tensor([0.25], dtype=torch.float64)), wt.double(), b.double(), padding=1)
        else:
            ref = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1), 0.2)
        ref = ref.float()
        scale = float(ref.abs().max())
        xg = x.to(dev)
        for cfg in range(14, lib.drba_conv3x3_num_cfgs()):
            if lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
                continue
            if kind == "res":
                got = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)(xg, residual=xg)
            elif kind == "pre":
                got = ops.Conv3x3(wt, b, 1, None, None, device=dev, cfg=cfg, pre_slope=0.25)(xg)
            else:
                got = ops.Conv3x3(wt, b, 1, True, None, device=dev, cfg=cfg)(xg)
            rows.append((f"conv split cfg{cfg} {kind} [{nb}x{cin}->{cout} {h}x{w}]", gpu_checks._diff(got, ref),
                         5e-6 * max(1.0, scale), f"|ref|max={scale:.2f}"))
    # the stride-2 tiles of the two-term form on the step's conv0 layers at N = 8 (ragged Cin 52 / 39 / 16)
    for (nb, cin, cout, h, w) in ((8, 52, 32, 544, 960), (8, 52, 48, 272, 480), (8, 16, 32, 544, 960), (8, 39, 96, 68, 120)):
        x = torch.randn(nb, cin, h, w, generator=g) * 2.0
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.leaky_relu(F.conv2d(x[:2].double(), wt.double(), b.double(), stride=2, padding=1), 0.2).float()  # (items 0, 1)
        last = F.leaky_relu(F.conv2d(x[-1:].double(), wt.double(), b.double(), stride=2, padding=1), 0.2).float()
        scale = float(ref.abs().max())
        xg = x.to(dev)
        for cfg in range(14, lib.drba_conv3x3_num_cfgs()):
            if lib.drba_conv3x3_cfg_stride(cfg) != 2 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
                continue
            got = ops.Conv3x3(wt, b, 2, True, None, device=dev, cfg=cfg)(xg)
            rows.append((f"conv split s2 cfg{cfg} [{nb}x{cin}->{cout} {h}x{w}] items 0-1", gpu_checks._diff(got[:2], ref),
                         5e-6 * max(1.0, scale), f"|ref|max={scale:.2f}"))
            rows.append((f"conv split s2 cfg{cfg} [{nb}x{cin}->{cout} {h}x{w}] item 7", gpu_checks._diff(got[-1:], last),
                         5e-6 * max(1.0, scale), ""))
    for (nb, cin, cout, h, w, ps) in ((2, 32, 52, 272, 480, True), (2, 64, 52, 136, 240, True), (1, 96, 64, 192, 480, False),
                                      (8, 32, 20, 272, 480, True), (8, 64, 52, 136, 240, True)):
        x = torch.randn(nb, cin, h, w, generator=g) * 2.0
        wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
        ref = (F.pixel_shuffle(ref, 2) if ps else ref).float()
        scale = float(ref.abs().max())
        for cfg in range(6, lib.drba_deconv4x4_num_cfgs()):
            if lib.drba_deconv4x4_packed_floats(cin, cout, cfg) == 0:
                continue
            got = ops.Deconv4x4(wt, b, ps, device=dev, cfg=cfg)(x.to(dev))
            rows.append((f"deconv split cfg{cfg} [{nb}x{cin}->{cout} {h}x{w} ps={ps}]", gpu_checks._diff(got, ref),
                         5e-6 * max(1.0, scale), f"|ref|max={scale:.2f}"))
    assert len(rows) >= 12
    _assert_rows(rows)


def test_glue_launches_with_8_items_at_1080p(hip_backend):
    """stage_conv0, the lazy scale-2 gather and warp_blend_lazy with 8 items per launch at 1088x1920 (the launch geometry of
    the benchmarked loop) against the reference's arithmetic (oracle ops, fp64 convolution)."""
    _assert_rows(gpu_checks.check_glue_n8_fullsize(hip_backend.dev))


def test_config3_one_clip_1080p_fps60_scdet_and_its_sharding(hip_backend):
    """BASELINE.json configs[2] as ONE clip at the benchmarked size: 24 source frames 1080x1920 (net 1088x1920), 24 -> 60
    fps (fractional timesteps + DRM), scene detection on with a planted cut in the middle (24 source frames: the driver forms
    groups of steps on each side of it) -- what bench.py's `config3_*` leg times --
    through the real driver loop on the HIP path (to_inp / to_out / check_scene on the device, lookahead and prefetch
    active) against the same loop on the CPU oracle: same number of frames, same cut decisions (a different decision
    would put a copy where a synthesised frame belongs: far more than 1 LSB), every byte within 1 LSB.
    Then configs[4]'s logic at a benchmarked frame size: the clip frame-sharded over two ranks (run one after the other on
    the one GPU, each rebuilding its halo state) must reproduce the sequential HIP run, at scale 1.0 and at scale 0.5."""
    import oracle
    from drba_amd import infer as drv
    from drba_amd import parallel
    from tests.clip_common import ListIO, cpu_hooks
    from tests.test_gpu_parallel import _hooks
    dev = hip_backend.dev
    sd = synth.ifnet_state_dict(seed=0)
    frames = synth.make_clip(24, 1080, 1920, seed=1234, cut_at=12)  # 12 frames on each side of the cut: groups of 4 steps form on both
    to_inp, to_out = _hooks(dev)
    hip = hip_backend.make_rife(sd, 1.0)
    io = ListIO(frames, 24.0)
    n = drv.interpolate_stream(hip, io, 60.0, enable_scdet=True, to_inp=to_inp, to_out=to_out)
    torch.cuda.synchronize()
    st = dict(hip.stats)  # the driver loop took the grouped path on both sides of the cut (and staged a group on the side stream)
    assert st["groups_formed"] >= 3 and st["staged_groups"] >= 1, st
    cio = ListIO(frames, 24.0)
    c_inp, c_out, c_check = cpu_hooks()
    drv.interpolate_stream(oracle.rife.RifeOracle(sd, 1.0), cio, 60.0, enable_scdet=True, to_inp=c_inp, to_out=c_out,
                           check_scene=c_check)
    assert n == len(io.written) == len(cio.written)
    worst, differing, total = 0, 0, 0
    for a, b in zip(io.written, cio.written):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        worst, differing, total = max(worst, int(d.max())), differing + int((d > 0).sum()), total + d.size
    rows = [("config 3 clip: HIP driver vs oracle driver, uint8 LSB", float(worst), 1.0,
             f"{len(io.written)} frames, {differing}/{total} bytes differ")]
    # ---- the same clip sharded over 2 ranks (sequentially on this GPU) vs the sequential HIP run
    for scale in (1.0, 0.5):
        m = hip if scale == 1.0 else hip_backend.make_rife(sd, 0.5)
        seq = io.written
        if scale != 1.0:
            sio = ListIO(frames, 24.0)
            drv.interpolate_stream(m, sio, 60.0, enable_scdet=True, to_inp=to_inp, to_out=to_out)
            seq = sio.written
        parts = []
        for rank in range(2):
            parts += parallel.interpolate_shard(m, frames, 24.0, 60.0, rank, 2, enable_scdet=True, to_inp=to_inp, to_out=to_out)
        torch.cuda.synchronize()
        assert len(parts) == len(seq)
        w = max(int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()) for a, b in zip(parts, seq))
        rows.append((f"config 3 clip sharded over 2 ranks vs sequential HIP run, scale {scale}, uint8 LSB", float(w), 1.0, ""))
    _assert_rows(rows)
    assert differing / total < 2e-3
