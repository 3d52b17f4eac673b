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
    if case.startswith("gmfss"):  # GMFlow needs >= 128 x 256 (window splits of the 1/8 and 1/4 resolution maps)
        return synth.make_clip(6, 128, 256, seed=3)
    return synth.make_clip(9, 64, 128, seed=3)


def _model(case):
    import oracle
    from tests import cases
    if case == "gmfss_union":
        s = synth.gmfss_union_state_dicts(seed=0)
        return oracle.gmfss.GmfssUnionOracle(s["flownet"], s["metric"], s["feat"], s["fusion"], s["rife"], 1.0)
    if case == "gmfss":
        s = cases.gmfss_state_dicts(seed=0)
        return oracle.gmfss.GmfssOracle(s["flownet"], s["metric"], s["feat"], s["fusion"], 1.0)
    return oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), 1.0)


def _worker(rank, world, port, case, times, dst_fps, scdet, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = _model(case)
    frames = _clip(case)
    to_inp, to_out, check = _cpu_hooks()
    if case in ("cut", "gmfss_union"):  # frames streamed to the writer in rounds of 2 emissions while the shard runs
        sg = parallel.StreamedGather(rank, world, parallel.emission_counts(len(frames), 24.0, dst_fps, times, world), chunk=2,
                                     frame_shape=None if case == "cut" else frames[0].shape)
        rest = parallel.interpolate_shard(model, frames, 24.0, dst_fps, rank, world, times=times, enable_scdet=scdet,
                                          to_inp=to_inp, to_out=to_out, check_scene=check, sink=sg.push)
        assert rest == []
        allf = sg.finish()
    else:
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


@pytest.mark.parametrize("case,times,dst_fps,scdet", [("plain", 2, 60, False), ("cut", -1, 60, True),
                                                      ("gmfss", 2, 60, False), ("gmfss_union", -1, 60, False)])
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


class _FakeModel:
    """Cheap stand-in with the models' call surface whose outputs depend on the carried `reuse` state the way the real
    ones do (cold and warm steps differ), so a wrong halo / cut state in a shard changes the frames."""
    scale, pad_size = 1.0, 16

    def calc_flow(self, a, b):
        return a - b, b - a * 0.5, a * 2.0, b * 3.0

    def inference_ts(self, I0, I1, ts):
        return [I0 if t == 0 else I1 if t == 1 else (1 - t) * I0 + t * I1 for t in ts]

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False):
        flow10, _, f1, _ = self.calc_flow(I1, I0) if not reuse else reuse
        flow12, flow21, f1b, f2 = self.calc_flow(I1, I2)
        out = []
        for t in ts:
            if t in (0, 1, 2):
                out.append((I0, I1, I2)[int(t)])
            else:
                out.append((I1 + 0.1 * flow10 * (1 - t) + 0.05 * flow12 * t + 0.01 * f1).clamp(0, 1))
        return out, (flow21, flow12, f2, f1b)


def _fake_worker(rank, world, port, scdet, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    frames = synth.make_clip(7, 32, 64, seed=5, cut_at=3 if scdet else None)
    to_inp, to_out, check = _cpu_hooks()
    model = _FakeModel()
    seen = []
    sg = parallel.StreamedGather(rank, world, parallel.emission_counts(len(frames), 24.0, 60, -1, world), chunk=1, in_flight=1,
                                 on_round=lambda j, rows: seen.append((j, [len(r) for r in rows])))
    parallel.interpolate_shard(model, frames, 24.0, 60, rank, world, enable_scdet=scdet, to_inp=to_inp, to_out=to_out,
                               check_scene=check, sink=sg.push)
    allf = sg.finish()
    # bounded buffering: rounds are retired (send / receive buffers dropped, the writer's frames handed on) as later ones
    # are issued, never more than in_flight + the one being issued alive at a time
    assert sg.n_rounds >= 3 and sg.peak_device_rounds <= 2 and all(b is None for b in sg.keep) and all(r is None for r in sg.recv)
    if rank == 0:
        assert [j for j, _ in seen] == list(range(sg.n_rounds))
        assert [c for _, c in seen] == [[sg.per_round[r][j] for r in range(world)] for j in range(sg.n_rounds)]
    if rank == 0:
        io = _IO(frames, 24.0)
        drv.interpolate_stream(model, io, 60, enable_scdet=scdet, to_inp=to_inp, to_out=to_out, check_scene=check)
        q.put((len(allf), len(io.written), all(np.array_equal(a.numpy(), b) for a, b in zip(allf, io.written))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scdet", (False, True))
def test_streamed_gather_uneven_world3(scdet):
    """7 frames -> 5 loop iterations over 3 ranks (2, 2, 1): ranks have 3 / 2 / 2 emissions, rounds of 2, so the last
    round is empty for two ranks; the streamed result must equal the sequential driver's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fake_worker, args=(r, 3, port, scdet, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    n_sharded, n_seq, same = q.get(timeout=10)
    assert n_sharded == n_seq and same


def _cli_worker(rank, world, port, tmp, q):
    """drba_amd.infer.inference_sharded (what `torchrun infer.py` runs per rank) with the oracle model and CPU hooks."""
    import argparse
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = _model("cut")
    to_inp, to_out, check = _cpu_hooks()
    args = argparse.Namespace(input=os.path.join(tmp, "in.npz"), output=os.path.join(tmp, "sharded.npz"), dst_fps=60.0, times=-1,
                              enable_scdet=True, scdet_threshold=0.3, hwaccel=False)
    n = drv.inference_sharded(model, args, rank, world, to_inp=to_inp, to_out=to_out, check_scene=check, chunk=2)
    if rank == 0:
        frames = _clip("cut")
        io = _IO(frames, 24.0)
        drv.interpolate_stream(model, io, 60.0, times=-1, enable_scdet=True, to_inp=to_inp, to_out=to_out, check_scene=check)
        z = np.load(args.output)
        ok = n == len(io.written) and z["frames"].shape[0] == n and all(np.array_equal(a, b) for a, b in zip(z["frames"], io.written))
        q.put((ok, n, len(io.written), float(z["fps"])))
    dist.barrier()
    dist.destroy_process_group()


def test_cli_sharded_run_writes_the_sequential_file(tmp_path):
    """The CLI's frame-sharded path (WORLD_SIZE > 1): rank 0's output file == the single-process driver's frames."""
    np.savez(os.path.join(tmp_path, "in.npz"), frames=np.stack(_clip("cut")), fps=np.float64(24.0))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, n, n_seq, fps = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert ok, (n, n_seq)
    assert fps == 60.0
