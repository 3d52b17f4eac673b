"""Frame-level data parallelism: shard the clip across ranks, gather finished frames to the writer.

The reference has no parallelism at all; this is the scheme BASELINE.json's
north_star prescribes.  Every (I0, I1, I2) step is independent *given* the driver state that
enters it, so the source-step range is cut into contiguous chunks, one per rank (one process
per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests).
Each rank rebuilds the state the sequential driver would carry into its first step from a
halo of one extra leading frame:

    cut_left = check_scene(frame[a-1... see below])      (only with scene detection on)
    reuse    = swap(model.calc_flow(frame[a], frame[a+1])) if the previous step was a DRBA step else None

which is exactly what step a-1 of the sequential run returns (reference models/rife.py:82-85,109),
so the sharded result equals the sequential one (up to fp32 atomic-order jitter of the splats).
The only collective in the data path is the gather of finished uint8 frames to rank 0; a 1080p
frame is 6.2 MB, i.e. ~0.2 GB/s per xGMI link at 8 x 30 fps -- no ring tuning needed.
"""
import numpy as np
import torch

from drba_amd.models.utils import tools as _tools


def partition(n_loop_steps, world):
    """Contiguous, near-equal split of loop iterations [0, n) -> list of (first, last_exclusive) per rank."""
    base, rem = divmod(n_loop_steps, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


def warm_reuse(model, Ia, Ib):
    """The `reuse` state a DRBA step on (.., Ia, Ib) hands to the next step (rife.py:82-85,109)."""
    if hasattr(model, "warm_reuse"):
        return model.warm_reuse(Ia, Ib)
    flow_ab, flow_ba, fa, fb = model.calc_flow(Ia, Ib)
    return (flow_ba, flow_ab, fb, fa)


def interpolate_shard(model, frames, src_fps, dst_fps, rank, world, times=-1, enable_scdet=False, scdet_threshold=0.3,
                      to_inp=None, to_out=None, check_scene=None):
    """Run this rank's share of the clip.  `frames` is a random-access sequence of uint8 HWC frames
    (every rank can index it; only its own range plus the halo is touched).
    Returns the list of output frames (whatever to_out returns) this rank is responsible for, in order.
    Concatenating the lists of ranks 0..world-1 gives exactly the sequential driver's output."""
    to_inp = to_inp or _tools.to_inp
    to_out = to_out or _tools.to_out
    check_scene = check_scene or _tools.check_scene
    if dst_fps <= src_fps:
        raise ValueError(f"dst fps should be greater than src fps, but got dst_fps={dst_fps} and src_fps={src_fps}")
    n = len(frames)
    n_loop = max(n - 2, 0)
    a, b = partition(n_loop, world)[rank]
    size = _tools.get_valid_net_inp_size(frames[0], model.scale, div=model.pad_size)
    src_size, dst_size = size["src_size"], size["dst_size"]
    mapper = _tools.TMapper(src_fps, dst_fps, times)
    cache = {}

    def inp(k):
        if k not in cache:
            cache[k] = to_inp(frames[k], dst_size)
        return cache[k]

    def cut(k):  # scene cut between frame k and k+1
        return bool(check_scene(inp(k), inp(k + 1), scdet_threshold)) if enable_scdet else False

    out = []

    def emit(xs):
        out.extend(to_out(x, src_size) for x in xs)

    # ---- head (rank 0 only): infer.py:93-110
    if rank == 0:
        ts = _tools.calc_t(0, times, mapper)
        if cut(0):
            emit([inp(0) for _ in ts])
        else:
            emit([inp(0) for _ in ts[ts < 1]] + list(model.inference_ts(inp(0), inp(1), ts[ts >= 1] - 1)))

    # ---- state entering loop iteration a, as the sequential driver would have it
    cut_left = cut(a) if (a < b or rank == world - 1) and n >= 2 else False
    reuse = None
    if a > 0 and a < b and not cut_left and not cut(a - 1):
        # iteration a-1 had (left, right) = (cut(a-1), cut(a)): a DRBA step iff both are False
        reuse = warm_reuse(model, inp(a), inp(a + 1))

    # ---- loop iterations [a, b): infer.py:112-156 with idx == k
    can_look = bool(getattr(model, "supports_lookahead", False))
    for k in range(a, b):
        I0, I1, I2 = inp(k), inp(k + 1), inp(k + 2)
        ts = _tools.calc_t(k, times, mapper)
        cut_right = cut(k + 1)
        if cut_left and cut_right:
            res, reuse = [I1 for _ in ts], None
        elif cut_left:
            reuse = None
            res = [I1 for _ in ts[ts < 1]] + list(model.inference_ts(I1, I2, ts[ts >= 1] - 1))
        elif cut_right:
            reuse = None
            res = list(model.inference_ts(I0, I1, ts[ts <= 1])) + [I1 for _ in ts[ts > 1] - 1]
        elif can_look and k + 1 < b and k + 3 < n:
            # one-frame lookahead inside the shard (drba_amd/models/lookahead.py): frame k+3 is I2 of iteration k+1
            res, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, linear=True, lookahead=(inp(k + 3), _tools.calc_t(k + 1, times, mapper)))
        else:
            res, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, linear=True)
        emit(res)
        cut_left = cut_right
        cache.pop(k, None)

    # ---- tail (last rank only): infer.py:158-169, idx == n-2
    if rank == world - 1 and n >= 2:
        ts = _tools.calc_t(n - 2, times, mapper)
        emit(list(model.inference_ts(inp(n - 2), inp(n - 1), ts[ts <= 1])) + [inp(n - 1) for _ in ts[ts > 1] - 1])
    return out


def gather_frames(local_frames, rank, world, device=None, group=None):
    """Gather every rank's finished uint8 frames on rank 0 (the writer), in rank order.

    One size exchange + one padded gather (torch.distributed, RCCL on GPUs).  Returns the full
    ordered list on rank 0, None elsewhere."""
    import torch.distributed as dist
    if world == 1:
        return list(local_frames)
    dev = device if device is not None else torch.device("cpu")
    as_t = [f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f)) for f in local_frames]
    shape = tuple(as_t[0].shape) if as_t else None
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(as_t)], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    shapes = [None] * world
    dist.all_gather_object(shapes, shape, group=group)
    shape = next(s for s in shapes if s is not None)
    nmax = max(counts)
    buf = torch.zeros((nmax,) + shape, dtype=torch.uint8, device=dev)
    if as_t:
        buf[:len(as_t)] = torch.stack([t.to(dev) for t in as_t])
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, recv, dst=0, group=group)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        out.extend(recv[r][k] for k in range(counts[r]))
    return out
