"""Host-side utilities of the DRBA driver, same names as the reference's models/utils/tools.py.

Pure host logic (sizes, timestep mapping, weight-key conversion, frame IO) lives here in
Python like the reference.  Anything that touches pixels on the hot path (resize,
distance, scene check) dispatches to the HIP library through drba_amd.ops; there is no
CPU fallback in this module.
"""
import json
import math
import weakref
import os
import subprocess
import threading
from queue import Queue

import numpy as np
import torch

from drba_amd import ops as _ops


def check_cupy_env():
    """The reference switches between a cupy/CUDA and a torch splat here (tools.py:14-24).
    This build has exactly one backend (the HIP library); there is no cupy path."""
    return False


# ----------------------------------------------------------------------------- sizes / conversion
def get_valid_net_inp_size(img, scale, div=64):
    """Network input size: each side rounded up so that side*scale is a multiple of `div`
    (frames are later *resized*, not padded).  Float arithmetic then int(), as in
    reference tools.py:41-56, so 1080 -> 1088 (div 64), 1152 (div 128); 2160@0.5 -> 2176."""
    src_h, src_w = int(img.shape[0]), int(img.shape[1])

    def round_up(v):
        if v * scale % div != 0:
            return int((v * scale // div + 1) * div / scale)
        return v

    return {"src_size": (src_h, src_w), "dst_size": (round_up(src_h), round_up(src_w))}


def convert(param):
    """Keep only keys carrying the DataParallel 'module.' prefix, with it stripped (tools.py:83-88)."""
    return {k.replace("module.", ""): v for k, v in param.items() if "module." in k}


def load_weights(path):
    """torch.load of a checkpoint file the way the reference reads it (map_location='cpu': its shipped pickles carry CUDA-tagged
    storages) but with weights_only=True: a state dict of tensors unpickles, arbitrary code in a downloaded .pkl does not run
    (the reference's plain torch.load executes whatever the file holds, rife.py:19 / GMFSS.py:50-53)."""
    return torch.load(path, map_location="cpu", weights_only=True)


def to_tensor(img, device=None):
    """uint8 HWC -> fp32 [1,3,H,W] in [0,1] on the GPU (tools.py:33-34).  Channel order is kept."""
    device = _ops.default_device() if device is None else device
    return _ops.u8hwc_to_f32nchw(torch.from_numpy(np.ascontiguousarray(img)).to(device, non_blocking=True))


def to_cv2(img):
    """fp32 [1,3,H,W] -> uint8 HWC with (x*255.) truncation, no clamp/round (tools.py:37-38)."""
    return _ops.f32nchw_to_u8hwc(img).cpu().numpy()


def resize(tensor, size):
    """Bilinear, align_corners=False, explicit output size (tools.py:71-72)."""
    return _ops.resize_bilinear(tensor, size)


def to_inp(npInp, dst_size, device=None):
    """resize(to_tensor(npInp), dst_size) (tools.py:59-60) as one kernel on the uploaded uint8 frame (the full-size
    fp32 frame of the reference is never materialised); bit-exact with the two-step form."""
    device = _ops.default_device() if device is None else device
    if torch.is_tensor(npInp):
        u8 = npInp.to(device, non_blocking=True)
    else:
        u8 = torch.from_numpy(np.ascontiguousarray(npInp)).to(device, non_blocking=True)
    return _ops.to_inp(u8, dst_size)


def to_out(tenInp, src_size, rgb=False):
    """to_cv2(resize(tenInp, src_size)) (tools.py:63-64) as one kernel + the D2H copy.  rgb=True returns the frame in RGB
    order (the flip the reference's writer thread does on the host, tools.py:202, done on the device instead)."""
    frame = _ops.to_out(tenInp, src_size, rgb=rgb).cpu().numpy()
    # the copy has waited for every kernel behind this frame: had one of the two-term fp16 kernels overflowed on the way, its
    # status byte is set by now -- raise here rather than hand the frame to the writer (ops.check_overflow: a host memory read)
    _ops.check_overflow(tenInp.device)
    return frame


def distance_calculator(_x):
    """sqrt(u^2 + v^2) per pixel in fp32 (tools.py:77-80)."""
    return _ops.flow_distance(_x)


def check_scene(x1, x2, scdet_threshold=0.3):
    """Scene cut test: SSIM(32x32 thumbnails, 3-D 11^3 gaussian) < threshold (tools.py:27-30).
    Returns a Python bool (the reference returns a 0-dim bool tensor used in `if`)."""
    return _ops.ssim_thumb32(x1, x2) < scdet_threshold


class SceneChecks:
    """check_scene for frame pairs announced ahead of their use (not in the reference, same decisions): submit(key, x1, x2)
    enqueues the test when the driver has both frames, cut(key, x1, x2) returns the bool -- waiting only for that test's own
    event -- and remembers it (the drivers ask for the same pair in several iterations).  `key` identifies the pair for
    the caller (frame indices, or object ids).  Only the decision is kept once it is known, with WEAK references to the
    two frames as the identity guard (an id can be recycled once a frame is gone: a dead or different referent is a miss):
    a remembered pair must not keep its frames -- and the encoder features hung on them, 270 MB per 1080p frame -- alive.
    A pending test holds its frames only until it is collected; the last 32 decisions are kept."""

    def __init__(self, scdet_threshold=0.3):
        self.thr, self.pending, self.done = scdet_threshold, {}, {}

    def submit(self, key, x1, x2):
        if key not in self.pending and not self._known(key, x1, x2):
            self.pending[key] = (_ops.ssim_thumb32_async(x1, x2), x1, x2)
            while len(self.pending) > 32:  # tests nobody asked for (a driver that stopped early): do not pin their frames
                self.pending.pop(next(iter(self.pending)))

    def _known(self, key, x1, x2):
        d = self.done.get(key)
        return d is not None and d[1]() is x1 and d[2]() is x2

    def cut(self, key, x1, x2):
        if self._known(key, x1, x2):
            return self.done[key][0]
        self.done.pop(key, None)
        p = self.pending.pop(key, None)
        if p is None or p[1] is not x1 or p[2] is not x2:
            p = (_ops.ssim_thumb32_async(x1, x2), x1, x2)
        (host, ev), _, _ = p
        ev.synchronize()
        res = float(host.item()) < self.thr
        self.done[key] = (res, weakref.ref(x1), weakref.ref(x2))
        while len(self.done) > 32:
            self.done.pop(next(iter(self.done)))
        return res


# ----------------------------------------------------------------------------- timestep mapping
class TMapper:
    """Maps a source-frame interval to the output timestamps that fall inside it (tools.py:120-134)."""

    def __init__(self, src=-1.0, dst=0.0, times=-1):
        self.times = dst / src if times == -1 else times
        self.now_step = -1

    def get_range_timestamps(self, _min, _max, lclose=True, rclose=False, normalize=True):
        first = math.ceil(_min * self.times)
        last = math.ceil(_max * self.times)
        if not lclose:
            first += 1
        if rclose:
            last += 1
        if first >= last:
            return []
        if normalize:
            return [((i / self.times) - _min) / (_max - _min) for i in range(first, last)]
        return [i / self.times for i in range(first, last)]


def calc_t(idx, times, t_mapper):
    """Timesteps (in [0.5, 1.5), relative to frame idx-... see driver) for one centre frame.

    Integer `times` (reference infer.py:76-87): odd -> [..., 1, ...] symmetric about 1,
    even -> (i + 0.5)/times mirrored about 1.  Otherwise (infer.py:89-91) the output
    timestamps inside [idx-0.5, idx+0.5) from TMapper, shifted by -idx, rounded to 4
    decimals, +1.  float64 throughout; must be bit-identical to the reference.
    """
    if times != -1:
        if times % 2:
            half = [(i + 1) / times for i in range((times - 1) // 2)]
            return np.array(list(reversed([1 - t for t in half])) + [1] + [t + 1 for t in half])
        half = [(i + 0.5) / times for i in range(times // 2)]
        return np.array(list(reversed([1 - t for t in half])) + [t + 1 for t in half])
    stamps = np.array(t_mapper.get_range_timestamps(idx - 0.5, idx + 0.5, lclose=True, rclose=False, normalize=False))
    return np.round(stamps - idx, 4) + 1


# ----------------------------------------------------------------------------- frame source / sink
class VideoFI_IO:
    """Frame source/sink with the reference's interface (tools.py:156-213): read_frame(),
    write_frame(), finish_writing(), .src_fps, .total_frames_count, .width, .height.

    Containers: when OpenCV and an ffmpeg binary exist the reference's path is used
    (cv2.VideoCapture decode, rawvideo rgb24 pipe into ffmpeg libx264 -qp 16, audio copied).
    Neither exists on the MI355X image, so the native formats are:
      input : .npz  {frames: uint8 [N,H,W,3] BGR, fps: float}  or .npy (+ optional .json {"fps"})
      output: .npz  {frames, fps}  or .npy;  anything else -> raw rgb24 bytes (ffmpeg's pipe format)
    `hwaccel` selects a hardware encoder when ffmpeg is present (h264_vaapi/h264_amf on AMD
    instead of the reference's h264_nvenc).
    """

    def __init__(self, input_path, output_path, dst_fps=60, times=-1, hwaccel=False, src_fps=None):
        self.input_path, self.output_path = input_path, output_path
        self._cv2 = None
        ext = os.path.splitext(input_path)[1].lower()
        if ext in (".npz", ".npy"):
            if ext == ".npz":
                z = np.load(input_path)
                self._frames = z["frames"]
                fps = float(z["fps"]) if "fps" in z.files else None
            else:
                self._frames = np.load(input_path, mmap_mode="r")
                side = os.path.splitext(input_path)[0] + ".json"
                fps = float(json.load(open(side))["fps"]) if os.path.exists(side) else None
            self.src_fps = float(src_fps if src_fps is not None else (fps if fps is not None else 24.0))
            self.total_frames_count = float(len(self._frames))
            self.height, self.width = int(self._frames.shape[1]), int(self._frames.shape[2])
        else:
            try:
                import cv2  # noqa: WPS433 (optional dependency, absent on the GPU image)
            except ImportError as e:
                raise RuntimeError(f"decoding {ext or 'this input'} needs OpenCV, which is not installed; "
                                   "use a .npz/.npy clip") from e
            self._cv2 = cv2.VideoCapture(input_path)
            self.src_fps = self._cv2.get(cv2.CAP_PROP_FPS)
            self.total_frames_count = self._cv2.get(7)
            self.width = int(self._cv2.get(cv2.CAP_PROP_FRAME_WIDTH))
            self.height = int(self._cv2.get(cv2.CAP_PROP_FRAME_HEIGHT))
        self.dst_fps = times * self.src_fps if times != -1 else dst_fps

        oext = os.path.splitext(output_path)[1].lower()
        self._sink_frames = [] if oext in (".npz", ".npy") else None
        self._ffmpeg = None
        self._raw = None
        if self._sink_frames is None:
            if _have_ffmpeg() and oext not in (".raw", ".rgb"):
                self._ffmpeg = self._spawn_ffmpeg(hwaccel)
            else:
                self._raw = open(output_path, "wb")
        # sinks that want RGB bytes (the ffmpeg pipe and the raw file, tools.py:202): the driver can hand over frames that
        # are already RGB (to_out(..., rgb=True) flips on the device) by setting frames_are_rgb
        self.wants_rgb = self._sink_frames is None
        self.frames_are_rgb = False
        self._sink_error = None
        self.read_buffer = Queue(maxsize=100)
        self.write_buffer = Queue(maxsize=-1)
        self._closed = threading.Event()
        threading.Thread(target=self._reader, daemon=True).start()
        self._writer_thread = threading.Thread(target=self._writer, daemon=True)
        self._writer_thread.start()

    def _ffmpeg_cmd(self, hwaccel):
        """The reference's pipe (tools.py:176-186): rawvideo rgb24 in, libx264 -qp 16 -preset medium, audio copied from the
        source container.  -hw selects h264_vaapi (the reference's h264_nvenc has no AMD counterpart), which needs the
        device + upload filter and takes neither -preset nor a software pixel format.  A .npz/.npy source has no audio
        stream to map, so the second input is only added for real containers."""
        container = self._cv2 is not None
        cmd = ["ffmpeg", "-y"]
        if hwaccel:
            cmd += ["-vaapi_device", "/dev/dri/renderD128"]
        cmd += ["-f", "rawvideo", "-pix_fmt", "rgb24", "-r", f"{self.dst_fps}", "-s", f"{self.width}x{self.height}", "-i", "pipe:0"]
        if container:
            cmd += ["-i", self.input_path, "-map", "0:v", "-map", "1:a?"]
        if hwaccel:
            cmd += ["-vf", "format=nv12,hwupload", "-c:v", "h264_vaapi", "-qp", "16"]
        else:
            cmd += ["-c:v", "libx264", "-pix_fmt", "yuv420p", "-qp", "16", "-preset", "medium"]
        cmd += ["-movflags", "+faststart"]
        if container:
            cmd += ["-c:a", "aac", "-b:a", "320k"]
        return cmd + [f"{self.output_path}"]

    def _spawn_ffmpeg(self, hwaccel):
        return subprocess.Popen(self._ffmpeg_cmd(hwaccel), stdin=subprocess.PIPE)

    def _reader(self):
        if self._cv2 is not None:
            ok, fr = self._cv2.read()
            while ok:
                self.read_buffer.put(fr)
                ok, fr = self._cv2.read()
        else:
            for k in range(len(self._frames)):
                self.read_buffer.put(np.ascontiguousarray(self._frames[k]))
        self.read_buffer.put(None)

    def _writer(self):
        while True:
            item = self.write_buffer.get()
            if item is None:
                break
            if self._sink_frames is not None:
                self._sink_frames.append(item)
            elif self._sink_error is None:
                # BGR -> RGB as the reference's pipe, unless the driver already flipped on the device
                rgb = np.ascontiguousarray(item if self.frames_are_rgb else item[:, :, ::-1])
                try:
                    if self._ffmpeg is not None and self._ffmpeg.poll() is not None:
                        raise BrokenPipeError(f"ffmpeg exited with code {self._ffmpeg.returncode}")
                    (self._ffmpeg.stdin if self._ffmpeg is not None else self._raw).write(rgb)
                except (BrokenPipeError, OSError) as e:  # keep draining the queue; close() re-raises
                    self._sink_error = e
        if self._sink_frames is not None:
            arr = np.stack(self._sink_frames) if self._sink_frames else np.zeros((0, self.height, self.width, 3), np.uint8)
            if self.output_path.lower().endswith(".npz"):
                np.savez(self.output_path, frames=arr, fps=np.float64(self.dst_fps))
            else:
                np.save(self.output_path, arr)
        elif self._ffmpeg is not None:
            try:
                self._ffmpeg.stdin.close()
            except OSError:
                pass
            if self._ffmpeg.wait() != 0 and self._sink_error is None:
                self._sink_error = RuntimeError(f"ffmpeg exited with code {self._ffmpeg.returncode}")
        else:
            self._raw.close()
        self._closed.set()

    def write_frame(self, x):
        self.write_buffer.put(x)

    def read_frame(self):
        return self.read_buffer.get()

    def finish_writing(self):
        return self.write_buffer.empty() or self._closed.is_set()

    def close(self):
        """Flush and close the sink (the reference never terminates its writer thread; this does)."""
        self.write_buffer.put(None)
        self._writer_thread.join()
        if self._sink_error is not None:
            raise RuntimeError(f"writing {self.output_path} failed: {self._sink_error}") from self._sink_error


def _have_ffmpeg():
    from shutil import which
    return which("ffmpeg") is not None
