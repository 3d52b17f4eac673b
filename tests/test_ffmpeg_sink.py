"""CPU: the encoder-pipe sink of VideoFI_IO (reference models/utils/tools.py:174-204) is EXECUTED: a stub `ffmpeg`
on PATH records argv and the piped bytes.  The command line is compared with the reference's token by token (the documented
differences: no audio input for a .npz source, h264_vaapi instead of h264_nvenc under -hw), the bytes with the frames in RGB
order, the driver's rgb=True hand-over, a failing encoder and the container (cv2) source through a fake cv2 module."""
import os
import sys
import types

import numpy as np
import pytest

from drba_amd.models.utils import tools
from tests import ffmpeg_stub

# /root/reference/models/utils/tools.py:174-186 (generate_frame_renderer), as data: the tokens the reference passes to Popen
REF_CMD = ["ffmpeg", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-r", "{dst_fps}", "-s", "{width}x{height}", "-i", "pipe:0", "-i",
           "{input_path}", "-map", "0:v", "-map", "1:a?", "-c:v", "{encoder}", "-movflags", "+faststart", "-pix_fmt", "yuv420p", "-qp", "16",
           "-preset", "{preset}", "-c:a", "aac", "-b:a", "320k", "{output_path}"]


def ref_cmd(**kw):
    return [t.format(**kw) for t in REF_CMD]


@pytest.fixture()
def stub_on_path(tmp_path, monkeypatch):
    d = tmp_path / "bin"
    d.mkdir()
    ffmpeg_stub.write_stub(d)
    monkeypatch.setenv("PATH", str(d) + os.pathsep + os.environ.get("PATH", ""))
    assert tools._have_ffmpeg()
    return d


def _clip(n=5, h=36, w=48, seed=3):
    return np.random.default_rng(seed).integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


def _run_sink(io, frames):
    for f in frames:
        io.write_frame(f)
    while not io.finish_writing():
        pass
    io.close()


def test_npz_source_pipes_rgb_frames_into_the_reference_command(tmp_path, stub_on_path):
    frames = _clip()
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.mp4")
    np.savez(inp, frames=frames, fps=np.float64(24.0))
    io = tools.VideoFI_IO(inp, out, dst_fps=60, times=-1, hwaccel=False)
    assert io.wants_rgb and io.src_fps == 24.0 and io.dst_fps == 60 and (io.width, io.height) == (48, 36)
    got = [io.read_frame() for _ in range(len(frames) + 1)]
    assert got[-1] is None and all(np.array_equal(a, b) for a, b in zip(got, frames))
    _run_sink(io, list(frames))
    argv, data = ffmpeg_stub.recorded(out)
    # BGR frames in, RGB bytes on the pipe (tools.py:202: item[:, :, ::-1])
    assert data == np.ascontiguousarray(frames[:, :, :, ::-1]).tobytes()
    g, inputs, opts, maps, o = ffmpeg_stub.split_cmd(argv)
    rg, rin, ropts, rmaps, ro = ffmpeg_stub.split_cmd(ref_cmd(dst_fps=60, width=48, height=36, input_path=inp, encoder="libx264",
                                                              preset="medium", output_path=out))
    assert g == rg and o == ro == out
    assert inputs == [rin[0]]            # the rawvideo pipe exactly as the reference; a .npz has no audio to map: no second input
    assert maps == []
    assert opts == {k: v for k, v in ropts.items() if k not in ("-c:a", "-b:a")}  # same encoder options, no audio codec
    assert open(out, "rb").read() == b"stub-container"


def test_hwaccel_selects_the_amd_encoder(tmp_path, stub_on_path):
    frames = _clip(3)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out_hw.mp4")
    np.savez(inp, frames=frames, fps=np.float64(25.0))
    io = tools.VideoFI_IO(inp, out, dst_fps=60, times=2, hwaccel=True)
    assert io.dst_fps == 50.0  # times wins over dst_fps (tools.py:160-162)
    _run_sink(io, list(frames))
    argv, data = ffmpeg_stub.recorded(out)
    assert data == np.ascontiguousarray(frames[:, :, :, ::-1]).tobytes()
    _, inputs, opts, _, _ = ffmpeg_stub.split_cmd(argv)
    # the reference's -hw is h264_nvenc -preset p7 (tools.py:176-178): no AMD counterpart; VAAPI needs its device in front of the
    # input, the upload filter, and takes neither -preset nor a software pixel format
    assert inputs[0][:2] == ["-vaapi_device", "/dev/dri/renderD128"]
    assert inputs[0][2:] == ["-f", "rawvideo", "-pix_fmt", "rgb24", "-r", "50.0", "-s", "48x36", "-i", "pipe:0"]
    assert opts["-c:v"] == "h264_vaapi" and opts["-vf"] == "format=nv12,hwupload" and opts["-qp"] == "16"
    assert "-preset" not in opts and "-pix_fmt" not in opts and opts["-movflags"] == "+faststart"


def test_driver_hands_rgb_frames_over_and_the_pipe_does_not_flip_twice(tmp_path, stub_on_path, monkeypatch):
    """drba_amd.infer.inference with a sink that wants RGB asks to_out for rgb=True (the flip happens in the to_out kernel) and sets
    frames_are_rgb: the bytes on the pipe are the RGB frames once.  The model and the frame conversion are CPU stand-ins here
    (tests/test_gpu_cli.py runs the real ones)."""
    from drba_amd import infer as drv
    frames = _clip(4, 32, 48, seed=9)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "out.mkv")
    np.savez(inp, frames=frames, fps=np.float64(24.0))
    calls = []

    class Copy:  # "interpolates" by repeating the nearer frame: only the plumbing is under test
        scale, pad_size, supports_lookahead = 1.0, 32, False

        def inference_ts(self, I0, I1, ts):
            return [I0 if t < 0.5 else I1 for t in ts]

        def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
            return [I0 if t < 0.5 else (I1 if t < 1.5 else I2) for t in ts], None

    monkeypatch.setattr(tools, "to_inp", lambda fr, size, device=None: np.asarray(fr))
    def fake_to_out(x, size, rgb=False):
        calls.append(rgb)
        return np.ascontiguousarray(x[:, :, ::-1]) if rgb else x
    monkeypatch.setattr(tools, "to_out", fake_to_out)
    args = drv.parse_args(["-m", "rife", "-i", inp, "-o", out, "-t", "2"])
    n = drv.inference(Copy(), args)
    assert n == 8 and calls and all(calls)
    argv, data = ffmpeg_stub.recorded(out)
    assert argv[-1] == out and "48.0" in argv  # -t 2 on 24 fps
    want = [frames[0], frames[0], frames[1], frames[1], frames[2], frames[2], frames[3], frames[3]]
    assert data == np.stack([np.ascontiguousarray(f[:, :, ::-1]) for f in want]).tobytes()


def test_a_failing_encoder_is_reported_by_close(tmp_path, stub_on_path, monkeypatch):
    monkeypatch.setenv("DRBA_FFMPEG_STUB_RC", "3")
    frames = _clip(2)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "bad.mp4")
    np.savez(inp, frames=frames, fps=np.float64(24.0))
    io = tools.VideoFI_IO(inp, out)
    for f in frames:
        io.write_frame(f)
    with pytest.raises(RuntimeError, match="ffmpeg exited with code 3"):
        io.close()


def test_container_source_maps_the_audio_of_the_input_like_the_reference(tmp_path, stub_on_path, monkeypatch):
    """A real container goes through cv2.VideoCapture (absent from the image: a fake module with the four properties the
    reference reads, tools.py:158-165) and the command gains the reference's second input, stream maps and audio codec."""
    frames = _clip(3, 24, 40, seed=11)

    class Cap:
        def __init__(self, path):
            self.path, self.k = path, 0

        def get(self, prop):
            return {5: 23.976, 7: float(len(frames)), 3: 40.0, 4: 24.0}[prop]

        def read(self):
            if self.k >= len(frames):
                return False, None
            self.k += 1
            return True, frames[self.k - 1].copy()

    fake = types.ModuleType("cv2")
    fake.VideoCapture, fake.CAP_PROP_FPS, fake.CAP_PROP_FRAME_WIDTH, fake.CAP_PROP_FRAME_HEIGHT = Cap, 5, 3, 4
    monkeypatch.setitem(sys.modules, "cv2", fake)
    inp, out = str(tmp_path / "in.mp4"), str(tmp_path / "out.mp4")
    open(inp, "wb").write(b"x")
    io = tools.VideoFI_IO(inp, out, dst_fps=60)
    assert io.src_fps == 23.976 and io.total_frames_count == 3.0 and (io.width, io.height) == (40, 24)
    got = [io.read_frame() for _ in range(4)]
    assert got[-1] is None and all(np.array_equal(a, b) for a, b in zip(got, frames))
    _run_sink(io, list(frames))
    argv, data = ffmpeg_stub.recorded(out)
    assert data == np.ascontiguousarray(frames[:, :, :, ::-1]).tobytes()
    mine = ffmpeg_stub.split_cmd(argv)
    ref = ffmpeg_stub.split_cmd(ref_cmd(dst_fps=60, width=40, height=24, input_path=inp, encoder="libx264", preset="medium",
                                        output_path=out))
    assert mine == ref  # same global flags, both inputs, output options (as a set), stream maps and output path
