"""Shared pieces of the whole-clip tests: list-backed frame IO, a call-counting model wrapper, CPU conversion hooks."""
import numpy as np
import torch


class ListIO:
    """VideoFI_IO's read/write surface over a list of uint8 frames."""

    def __init__(self, frames, fps):
        self.src_fps, self.total_frames_count = fps, len(frames)
        self._it = iter(list(frames) + [None])
        self.written = []

    def read_frame(self):
        return next(self._it)

    def write_frame(self, x):
        self.written.append(x)


class CountingModel:
    """Delegates to a model and counts calls / synthesised frames per entry point."""

    def __init__(self, m):
        self.m, self.scale, self.pad_size = m, m.scale, m.pad_size
        self.calls = {"inference_ts": 0, "inference_ts_drba": 0}
        self.generated = {"inference_ts": 0, "inference_ts_drba": 0}

    def inference_ts(self, I0, I1, ts):
        self.calls["inference_ts"] += 1
        self.generated["inference_ts"] += sum(1 for t in ts if t not in (0, 1))
        return self.m.inference_ts(I0, I1, ts)

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
        self.calls["inference_ts_drba"] += 1
        self.generated["inference_ts_drba"] += sum(1 for t in ts if t not in (0, 1, 2))
        return self.m.inference_ts_drba(I0, I1, I2, ts, reuse, linear)


def cpu_hooks():
    """to_inp / to_out / check_scene of the reference (tools.py:27-38,59-72) on the CPU, through the oracle."""
    import oracle

    def to_inp(fr, size):
        return oracle.ops.resize(torch.from_numpy(np.ascontiguousarray(fr).transpose(2, 0, 1)).unsqueeze(0).float() / 255.0, size)

    def to_out(x, size):
        return (oracle.ops.resize(x, size)[0].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)

    return to_inp, to_out, oracle.scdet.check_scene
