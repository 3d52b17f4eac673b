"""CPU, world_size 2 over gloo: the frame-sharded runner must reproduce the sequential driver exactly,
including scene-cut state and the warm `reuse` rebuilt from the halo frame (drba_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from drba_amd import infer as drv
from drba_amd import parallel
from drba_amd.utils import synth


def _cpu_hooks():
    import oracle

    def to_inp(fr, size):
        return oracle.ops.resize(torch.from_numpy(fr.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0, size)

    def to_out(x, size):
        return (oracle.ops.resize(x, size)[0].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)

    return to_inp, to_out, oracle.scdet.check_scene


class _IO:
    def __init__(self, frames, fps):
        self.src_fps, self.total_frames_count = fps, len(frames)
        self._it = iter(list(frames) + [None])
        self.written = []

    def read_frame(self):
        return next(self._it)

    def write_frame(self, x):
        self.written.append(x)


def _clip(case):
    if case == "cut":
        return synth.make_clip(9, 64, 128, seed=3, cut_at=4)
    return synth.make_clip(9, 64, 128, seed=3)


def _worker(rank, world, port, case, times, dst_fps, scdet, q):
    import oracle
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), 1.0)
    frames = _clip(case)
    to_inp, to_out, check = _cpu_hooks()
    mine = parallel.interpolate_shard(model, frames, 24.0, dst_fps, rank, world, times=times, enable_scdet=scdet,
                                      to_inp=to_inp, to_out=to_out, check_scene=check)
    allf = parallel.gather_frames(mine, rank, world)
    if rank == 0:
        io = _IO(frames, 24.0)
        drv.interpolate_stream(model, io, dst_fps, times=times, enable_scdet=scdet, to_inp=to_inp, to_out=to_out,
                               check_scene=check)
        ok = len(allf) == len(io.written) and all(np.array_equal(a.numpy(), b) for a, b in zip(allf, io.written))
        q.put((ok, len(allf), len(io.written)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case,times,dst_fps,scdet", [("plain", 2, 60, False), ("cut", -1, 60, True)])
def test_sharded_equals_sequential_world2(case, times, dst_fps, scdet):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, times, dst_fps, scdet, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    ok, n_sharded, n_seq = q.get(timeout=10)
    assert n_sharded == n_seq
    assert ok, "sharded output differs from the sequential driver"


def test_partition_covers_range():
    for n in (0, 1, 7, 16, 100):
        for w in (1, 2, 3, 8):
            parts = parallel.partition(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
