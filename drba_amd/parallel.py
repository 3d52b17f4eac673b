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


def emission_counts(n_frames, src_fps, dst_fps, times, world):
    """Output frames per emission (head / loop iteration / tail) of every rank: [[count, ...] per rank].
    Every branch of the driver loop emits len(calc_t(idx)) frames (copies at scene cuts replace synthesised frames one
    for one, infer.py:98-103,121-143,160-162), so the counts are known to all ranks without any communication."""
    mapper = _tools.TMapper(src_fps, dst_fps, times)
    n_loop = max(n_frames - 2, 0)
    out = []
    for r, (a, b) in enumerate(partition(n_loop, world)):
        c = [len(_tools.calc_t(0, times, mapper))] if r == 0 else []
        c += [len(_tools.calc_t(k, times, mapper)) for k in range(a, b)]
        if r == world - 1 and n_frames >= 2:
            c.append(len(_tools.calc_t(n_frames - 2, times, mapper)))
        out.append(c)
    return out


def interpolate_shard(model, frames, src_fps, dst_fps, rank, world, times=-1, enable_scdet=False, scdet_threshold=0.3,
                      to_inp=None, to_out=None, check_scene=None, sink=None):
    """Run this rank's share of the clip.  `frames` is a random-access sequence of uint8 HWC frames
    (every rank can index it; only its own range plus the halo is touched).
    Returns the list of output frames (whatever to_out returns) this rank is responsible for, in order.
    Concatenating the lists of ranks 0..world-1 gives exactly the sequential driver's output.
    `sink(list_of_frames)`, if given, receives each emission (head / one loop iteration / tail) as it is produced
    instead (StreamedGather.push: the frames travel to the writer rank while the next steps compute) and the
    function returns []."""
    to_inp = to_inp or _tools.to_inp
    to_out = to_out or _tools.to_out
    # the library's own scene test can be asked for ahead of its use (tools.SceneChecks); an injected one is called in place
    ahead_checks = _tools.SceneChecks(scdet_threshold) if (check_scene is None and enable_scdet) else None
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

    intake = [None]  # the model's intake stream (RIFE: its prefetch stream), set where the loop starts reading ahead

    def inp(k):
        if k not in cache:
            cache[k] = to_inp(frames[k], dst_size)
        return cache[k]

    def take_in(j):
        """Frame j read ahead: to_inp, the scene test of the pair (j - 1, j) and the model's prefetches on the intake stream, clear
        of the synthesis queue (drba_amd/infer.py `read`: the decision is needed before the frame can be announced as part of a
        group of steps; behind the caller's queue, waiting for it put host and GPU in lock step)."""
        s = intake[0]
        if s is None or j in cache:
            x = inp(j)
            cm = None
        else:
            main = torch.cuda.current_stream(s.device)
            cm = torch.cuda.stream(s)
            cm.__enter__()
            x = inp(j)
            if x.is_cuda:
                ev = torch.cuda.Event()
                ev.record(s)
                main.wait_event(ev)
                x.record_stream(main)
                x4 = getattr(x, "_drba_x4", None)
                if x4 is not None:
                    x4[0].record_stream(main)
        try:
            if ahead_checks is not None and j - 1 not in cuts:
                ahead_checks.submit(j - 1, inp(j - 1), x)  # the cut test the iterations before j - 1 will ask for
            prefetch(x)
            if prefetch_pair is not None:
                prefetch_pair(inp(j - 1), x)
        finally:
            if cm is not None:
                cm.__exit__(None, None, None)

    cuts = {}

    def cut(k):  # scene cut between frame k and k+1 (each pair is tested once: the loop looks one step ahead)
        if not enable_scdet:
            return False
        if k not in cuts:
            if ahead_checks is not None and inp(k).is_cuda:
                cuts[k] = ahead_checks.cut(k, inp(k), inp(k + 1))
            else:
                cuts[k] = bool(check_scene(inp(k), inp(k + 1), scdet_threshold))
        return cuts[k]

    out = []

    def emit(xs):
        fr = [to_out(x, src_size) for x in xs]
        if sink is not None:
            sink(fr)
        else:
            out.extend(fr)

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
    prefetch = getattr(model, "prefetch_frame", None) if can_look else None
    prefetch_pair = getattr(model, "prefetch_pair", None) if can_look else None
    prefetched = set()
    depth = max(3, 2 * int(getattr(model, "GROUP", 1)) - 1) if prefetch is not None else 1
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
            # lookahead inside the shard (drba_amd/models/lookahead.py): frame k+3 is I2 of iteration k+1.  A model that can
            # (RIFE) has the encoder and the coarse flow of every frame up to three ahead started on the prefetch stream, as
            # the sequential driver does, and is told the next iterations' frames and timesteps: it computes iterations k and
            # k+1 in one stacked pass when there is no cut on k+1's right (RIFE._drba_pair)
            look = (inp(k + 3), _tools.calc_t(k + 1, times, mapper))
            if prefetch is not None:
                far = min(k + 2 + depth, b + 1, n - 1)  # last frame this shard may name: iteration b - 1 reads frame b + 1
                if intake[0] is None and getattr(model, "intake_stream", None) is not None and I2.is_cuda:
                    intake[0] = model.intake_stream(I2.device)
                    if intake[0] is not None:
                        intake[0].wait_stream(torch.cuda.current_stream(I2.device))  # frames made on the caller's stream so far
                for j in range(k + 3, far + 1):
                    if j not in prefetched:
                        prefetched.add(j)
                        take_in(j)
                # the following iterations of this shard, as far as they are DRBA steps too (no cut up to the last frame named)
                entries = []
                for j in range(k + 3, far + 1):  # entry: iteration j - 2, whose I2 is frame j
                    if cut(j - 1):
                        break
                    entries += [inp(j), _tools.calc_t(j - 2, times, mapper)]
                if len(entries) >= 4:
                    look = tuple(entries)
            res, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, linear=True, lookahead=look)
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


def _default_device(group=None):
    """Where collective buffers must live: the current GPU for the nccl (= RCCL) backend, host memory for gloo."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class StreamedGather:
    """The RCCL gather of finished uint8 frames to the writer rank, overlapped with the computation.

    One gather at the end of a shard leaves the xGMI links idle while the GPUs compute and the GPUs idle while the
    frames travel (8 ranks x 40 1080p frames = 2 GB into rank 0).  Here the emissions (head / loop iteration / tail)
    of a rank are grouped into rounds of `chunk` emissions; after its j-th round every rank contributes a fixed-size
    padded buffer to an ASYNCHRONOUS dist.gather (RCCL runs it on its own stream, under the next steps' kernels).  All
    sizes follow from emission_counts(), which every rank computes locally, so ranks with fewer emissions (no head /
    tail) still take part in every round and the collectives are issued in the same order everywhere.
    Device memory stays bounded: at most `in_flight` rounds keep their send / receive buffers on the GPU; older rounds
    are retired as later ones are issued -- every rank drops its send buffer, the writer copies the round's received
    frames into pinned host memory on a drain stream of its own (a 10-minute 1080p60 clip is > 200 GB of uint8 frames:
    it cannot wait on the GPU for finish()).  `on_round(j, frames_by_rank)`, if given, receives the writer's host
    frames of round j as soon as they have landed (list over ranks of lists of HWC uint8 tensors).
    finish() returns the ordered frame list on rank 0 (host tensors; None elsewhere)."""

    def __init__(self, rank, world, counts, chunk=4, device=None, group=None, frame_shape=None, in_flight=2, on_round=None):
        self.rank, self.world, self.group, self.chunk = rank, world, group, max(1, int(chunk))
        self.dev = device if device is not None else (_default_device(group) if world > 1 else None)
        self.counts = counts
        self.n_rounds = max((len(c) + self.chunk - 1) // self.chunk for c in counts) if counts else 0
        # frames rank r contributes in round j
        self.per_round = [[sum(c[j * self.chunk:(j + 1) * self.chunk]) for j in range(self.n_rounds)] for c in counts]
        self.cap = [max(self.per_round[r][j] for r in range(world)) for j in range(self.n_rounds)]
        self.pending, self.emissions, self.round = [], 0, 0
        self.handles, self.recv, self.keep = [], [], []
        self.shape = None if frame_shape is None else tuple(frame_shape)  # (H, W, 3) of an output frame, if known
        self.local = []  # world == 1: plain accumulation
        self.in_flight, self.on_round = max(1, int(in_flight)), on_round
        self.retired = 0            # rounds [0, retired) no longer hold device buffers
        self.host = []              # writer: per retired round, per rank, the received frames in host memory
        self._drain, self._landed = None, []  # writer on a GPU: the D2H stream and one event per retired round
        self.peak_device_rounds = 0  # (tests) most rounds that held device buffers at any time

    def _tensor(self, f):
        t = f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f))
        return t.to(self.dev, non_blocking=True) if self.dev is not None else t

    def _issue(self):
        """The gather of round self.round (every rank calls this the same number of times, in the same order)."""
        import torch.distributed as dist
        j = self.round
        want = self.per_round[self.rank][j]
        assert len(self.pending) == want, (self.rank, j, len(self.pending), want)
        if self.shape is None:  # ranks whose first rounds are empty learn the frame shape from rank 0
            shp = [tuple(self.pending[0].shape) if self.pending else None]
            dist.broadcast_object_list(shp, src=0, group=self.group)
            self.shape = shp[0]
        buf = torch.empty((max(self.cap[j], 1),) + self.shape, dtype=torch.uint8, device=self.dev)
        if self.pending:
            torch.stack(self.pending, out=buf[:want])
        recv = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
        self.handles.append(dist.gather(buf, recv, dst=0, group=self.group, async_op=True))
        self.recv.append(recv)
        self.keep.append(buf)
        self.pending = []
        self.round += 1
        self.peak_device_rounds = max(self.peak_device_rounds, self.round - self.retired)
        while self.round - self.retired > self.in_flight:
            self._retire()

    def _retire(self):
        """Round self.retired leaves the device: wait for its gather, (writer) copy what it received to host memory."""
        j = self.retired
        h = self.handles[j]
        on_gpu = self.dev is not None and torch.device(self.dev).type == "cuda"
        if self.rank != 0:
            h.wait()  # (RCCL: orders the current stream behind the collective; gloo: blocks the host)
        elif not on_gpu:
            h.wait()
            self.host.append([[self.recv[j][r][k] for k in range(self.per_round[r][j])] for r in range(self.world)])
        else:
            if self._drain is None:
                self._drain = torch.cuda.Stream(device=self.dev)
            with torch.cuda.stream(self._drain):
                h.wait()  # the drain stream, not the compute stream, waits for the collective
                rows = []
                for r in range(self.world):
                    n = self.per_round[r][j]
                    src = self.recv[j][r]
                    src.record_stream(self._drain)
                    dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.uint8, pin_memory=True)
                    if n:
                        dst.copy_(src[:n], non_blocking=True)
                    rows.append([dst[k] for k in range(n)])
                ev = torch.cuda.Event()
                ev.record(self._drain)
            self.keep[j].record_stream(self._drain)
            self._landed.append(ev)
            self.host.append(rows)
        self.recv[j] = self.keep[j] = None  # device buffers back to the allocator
        self.retired += 1
        if self.rank == 0 and self.on_round is not None:
            if self._landed:
                self._landed[-1].synchronize()
            self.on_round(j, self.host[j])

    def push(self, frames):
        """One emission of this rank (a list of uint8 HWC frames)."""
        if self.world == 1:
            self.local.extend(frames)
            return
        self.pending.extend(self._tensor(f) for f in frames)
        self.emissions += 1
        if self.emissions % self.chunk == 0 or self.emissions == len(self.counts[self.rank]):
            self._issue()

    def finish(self):
        if self.world == 1:
            return list(self.local)
        assert self.emissions == len(self.counts[self.rank]), "every emission must be pushed before finish()"
        while self.round < self.n_rounds:  # ranks with fewer emissions: empty contributions to the remaining rounds
            self._issue()
        while self.retired < self.n_rounds:
            self._retire()
        if self.rank != 0:
            return None
        for ev in self._landed:
            ev.synchronize()
        out = []
        for r in range(self.world):
            for j in range(self.n_rounds):
                out.extend(self.host[j][r])
        return out


def gather_frames(local_frames, rank, world, device=None, group=None):
    """Gather every rank's finished uint8 frames on rank 0 (the writer), in rank order.

    One size exchange + one padded gather (torch.distributed, RCCL on GPUs).  Returns the full
    ordered list on rank 0, None elsewhere."""
    import torch.distributed as dist
    if world == 1:
        return list(local_frames)
    dev = device if device is not None else _default_device(group)
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
