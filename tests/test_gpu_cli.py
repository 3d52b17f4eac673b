"""GPU: the infer.py command line end to end on a synthetic .npz clip (the MI355X image has no cv2/ffmpeg),
with scene detection on and a planted cut; output frame count and pass-through frames must match the schedule."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from drba_amd.utils import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_infer_cli_rife_npz_roundtrip(tmp_path):
    frames = np.stack(synth.make_clip(8, 128, 192, seed=21, cut_at=5))
    wdir = tmp_path / "w"
    wdir.mkdir()
    torch.save({"module." + k: v for k, v in synth.ifnet_state_dict(0).items()}, str(wdir / "flownet.pkl"))
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, frames=frames, fps=np.float64(24.0))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import drba_amd.infer as I\n"
        "a = I.parse_args(['-m','rife','-i',%r,'-o',%r,'-t','2','-s'])\n"
        "from drba_amd.models.rife import RIFE\n"
        "m = RIFE(weights=%r, scale=a.scale)\n"
        "print('written', I.inference(m, a))\n" % (ROOT, inp, out, str(wdir)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert z["frames"].shape == (16, 128, 192, 3) and z["frames"].dtype == np.uint8  # -t 2: 2 per source frame
    assert float(z["fps"]) == 48.0
    assert "written 16" in r.stdout


def test_infer_cli_gmfss_union_npz_roundtrip(tmp_path):
    """`-m gmfss_union` through load_model's weight-directory convention (flownet/metric/feat/fusionnet/rife .pkl)."""
    frames = np.stack(synth.make_clip(5, 96, 160, seed=22))
    wdir = tmp_path / "w"
    wdir.mkdir()
    sds = synth.gmfss_union_state_dicts(0)
    for key, fn in (("flownet", "flownet"), ("metric", "metric"), ("feat", "feat"), ("fusion", "fusionnet")):
        torch.save(sds[key], str(wdir / (fn + ".pkl")))
    torch.save({"module." + k: v for k, v in sds["rife"].items()}, str(wdir / "rife.pkl"))
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, frames=frames, fps=np.float64(30.0))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import drba_amd.infer as I\n"
        "a = I.parse_args(['-m','gmfss_union','-i',%r,'-o',%r,'-t','2'])\n"
        "m = I.load_model(a.model_type, a.scale, weights=%r)\n"
        "assert m.pad_size == 128\n"
        "print('written', I.inference(m, a))\n" % (ROOT, inp, out, str(wdir)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert z["frames"].shape == (10, 96, 160, 3) and z["frames"].dtype == np.uint8
    assert float(z["fps"]) == 60.0


def _oracle_clip(frames, fps, dst_fps, times, scdet):
    """The whole clip through the driver on the CPU: oracle model + the reference's frame conversion (the checker)."""
    import oracle
    from drba_amd import infer as drv
    from tests.clip_common import ListIO, cpu_hooks
    io = ListIO(list(frames), fps)
    to_inp, to_out, check = cpu_hooks()
    drv.interpolate_stream(oracle.rife.RifeOracle(synth.ifnet_state_dict(0), 1.0), io, dst_fps, times=times, enable_scdet=scdet,
                           to_inp=to_inp, to_out=to_out, check_scene=check)
    return np.stack(io.written)


def _cli_clip(tmp_path, frames, fps, argv):
    wdir = tmp_path / "w"
    wdir.mkdir(exist_ok=True)
    torch.save({"module." + k: v for k, v in synth.ifnet_state_dict(0).items()}, str(wdir / "flownet.pkl"))
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, frames=np.stack(frames), fps=np.float64(fps))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import drba_amd.infer as I\n"
        "a = I.parse_args(['-m','rife','-i',%r,'-o',%r] + %r)\n"
        "m = I.load_model(a.model_type, a.scale, weights=%r)\n"
        "print('written', I.inference(m, a))\n" % (ROOT, inp, out, list(argv), str(wdir)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)["frames"]


def _assert_frames_close(got, want):
    """uint8 frames of the HIP CLI vs the CPU driver run: the fp32 frames agree to ~1e-6, so after *255 truncation a
    value sitting on an integer boundary may land one LSB apart; nothing may differ by more, and few may differ at all."""
    assert got.shape == want.shape and got.dtype == np.uint8
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"max diff {d.max()} LSB"
    assert (d > 0).mean() < 2e-3, f"{(d > 0).mean():.2e} of the bytes differ"


def test_cli_whole_clip_config1_matches_cpu_driver(tmp_path):
    """BASELINE.json configs[0]'s clip (16 frames, 854x480, -t 2, scdet off) through infer.py on the HIP path: all 32
    written frames within 1 LSB of the same clip run through the driver on the CPU oracle."""
    frames = synth.make_clip(16, 480, 854, seed=1234)
    got = _cli_clip(tmp_path, frames, 24.0, ["-t", "2"])
    assert got.shape == (32, 480, 854, 3)
    _assert_frames_close(got, _oracle_clip(frames, 24.0, 48.0, 2, False))


def test_cli_whole_clip_fps60_scdet_matches_cpu_driver(tmp_path):
    """-fps 60 -s on a 24 fps clip with a planted cut (configs[2] at 480p): fractional timesteps, DRM, scene-cut branches,
    reuse reset -- frame for frame against the CPU driver run (same cut decisions, same copies)."""
    frames = synth.make_clip(12, 480, 854, seed=77, cut_at=6)
    got = _cli_clip(tmp_path, frames, 24.0, ["-fps", "60", "-s"])
    want = _oracle_clip(frames, 24.0, 60.0, -1, True)
    _assert_frames_close(got, want)


def test_bench_sharded_legs_run_at_world_1():
    """bench.py's N > 1 legs (untimed sharded warm-up, then interpolate_shard + StreamedGather over the headline clip and
    over the 4K config-5 clip with its planted cut) cannot be launched on the one-GPU box; --selftest-sharded runs the
    same functions at world 1 (no process group): they must execute on the HIP model and count what the schedule says."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-sharded", "--steps", "4", "--config", "480p"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    h, c5 = d["headline_clip"], d["config5_clip"]
    # -t 2 over 6 source frames: head (1 synthesised, 1 copy) + 4 DRBA steps x 2 + tail (1): 12 written, 10 generated
    assert (h["frames_generated"], h["writer_frames"]) == (10, 12), h
    # the rank reports which path its shard took (groups formed / staged, ramp-in steps): 4 DRBA steps = 1 cold + a group or single steps
    rp = h["rank_path"]
    assert rp is not None and rp["single_steps"] + rp["group_collects"] + rp["groups_formed"] == 4 and rp["ramp_in_steps"] >= 1, rp
    # 24 -> 60 fps over 6 source frames with a cut: every written frame is accounted for, copies replace synthesised ones
    assert c5["writer_frames"] >= c5["frames_generated"] > 0, c5


def test_bench_two_rank_rehearsal_over_gloo():
    """`python bench.py --gpus 2` typed WITHOUT a launcher (the shape of the driver's N = 1 command): bench.py starts its
    own ranks under torch.distributed.run.  Two ranks share the one GPU and the collectives are carried by gloo
    (DRBA_BENCH_BACKEND=gloo, a rehearsal switch): the replica loop, the untimed sharded warm-up, the sharded headline
    clip with its streamed gather to rank 0 and the strong-scaling config-5 clip with its planted cut all execute with two
    processes -- matching barriers, gather rounds and reductions -- the writer receives every frame of both clips, and the
    line says what torch.distributed saw (backend, world size, one entry per rank)."""
    import json
    env = dict(os.environ, DRBA_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--config",
                        "480p"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["writer_frames"] == 20  # 2 ranks x 4 steps + 2 source frames, `-t 2`: 2 per source frame
    c5 = d["config5_sharded"]
    assert c5["clip_source_frames"] == 34 and c5["writer_frames"] == 86 and c5["frames_generated"] > 0 and c5["scaling"] == "strong"
    assert d["replica_loop"]["value"] > 0
    j = d["dist"]
    assert j["world_size"] == 2 and j["backend"].startswith("gloo") and [x["rank"] for x in j["devices"]] == [0, 1]
    assert len({x["pid"] for x in j["devices"]}) == 2 and all(x["rank_seconds"] > 0 for x in j["devices"])
    assert d["settle_steps"] > 0


def test_infer_cli_pipes_rgb_frames_into_ffmpeg(tmp_path):
    """`infer.py -m rife -i clip.npz -o out.mp4 -t 2` (and `-hw`) with a stub `ffmpeg` first on PATH: the encoder-pipe branch of
    VideoFI_IO runs (reference tools.py:174-204), the bytes on the pipe are the frames of the .npz run in RGB order -- flipped by
    the to_out kernel (rgb=True), not on the host -- and the command line is the reference's (tests/test_ffmpeg_sink.py compares it
    token by token)."""
    from tests import ffmpeg_stub
    frames = synth.make_clip(6, 128, 192, seed=23)
    want = _cli_clip(tmp_path, frames, 24.0, ["-t", "2"])  # BGR frames of the same run into a .npz sink
    bindir = tmp_path / "bin"
    bindir.mkdir()
    ffmpeg_stub.write_stub(bindir)
    env = dict(os.environ, PATH=str(bindir) + os.pathsep + os.environ.get("PATH", ""))
    inp = str(tmp_path / "in.npz")
    for extra, name in (([], "out.mp4"), (["-hw"], "out_hw.mp4")):
        out = str(tmp_path / name)
        code = (
            "import sys; sys.path.insert(0, %r)\n"
            "import drba_amd.infer as I\n"
            "a = I.parse_args(['-m','rife','-i',%r,'-o',%r,'-t','2'] + %r)\n"
            "m = I.load_model(a.model_type, a.scale, weights=%r)\n"
            "print('written', I.inference(m, a))\n" % (ROOT, inp, out, extra, str(tmp_path / "w")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "written 12" in r.stdout
        argv, data = ffmpeg_stub.recorded(out)
        got = np.frombuffer(data, dtype=np.uint8).reshape(12, 128, 192, 3)
        # the same frames, channel order flipped (two processes: the autotuner's picks and the splats' summation order may differ
        # in the last bits, i.e. by one LSB on a value sitting on an integer boundary)
        _assert_frames_close(got, np.ascontiguousarray(want[:, :, :, ::-1]))
        assert argv[-1] == out and argv[argv.index("-s") + 1] == "192x128" and argv[argv.index("-r") + 1] == "48.0"
        assert argv[argv.index("-c:v") + 1] == ("h264_vaapi" if extra else "libx264")
