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
