"""One-frame lookahead on a side HIP stream (not in the reference).

The per-pair state a DRBA step needs for its NEW frame pair (RIFE: encoder + block0 coarse flow + flow reversal;
GMFSS: FeatureNet + GMFlow both ways + MetricNet) depends only on the two frames, so once the driver has read the next
frame it can be computed while the current step's frames are synthesised.  Those kernels are small and latency-bound
(hundreds of launches on 1/4..1/16-resolution maps for GMFlow), the synthesis kernels are large: overlapping them on
two streams fills the chip.  The result is handed to the next call by frame identity; a scene cut or any other call
pattern simply leaves it unused."""
import torch


_STREAMS = {}


ONE_STREAM = False  # measurement only (tools/step_timeline.py --one-stream): side / prefetch work on the caller's stream, so that a
#                     traced launch's duration is the kernel's own at the loop's launch geometry (8 samples per launch)
PRIORITY = {}  # role -> HIP stream priority (-1 = high); experiments only (tools/ab_bench.py --prefetch-priority)


# role -> (first CU, number of CUs) PER XCD (MI355X: 32 CUs on each of 8 XCDs): that role's stream is created with a CU mask
# (drba_stream_create_cu_mask) and its kernels run on those CUs of every XCD only.  Experiments: tools/ab_bench.py --cu-mask.
CU_MASK = {}
N_XCD, CUS_PER_XCD = 8, 32


def cu_mask_words(first, count):
    """The mask of CUs [first, first + count) on every XCD: mask bit i is CU i // 8 of XCD i % 8 (tools/exp/cu_mask/census.hip)."""
    import ctypes as C
    words = (C.c_uint32 * (N_XCD * CUS_PER_XCD // 32))()
    for cu in range(first, min(first + count, CUS_PER_XCD)):
        for x in range(N_XCD):
            b = cu * N_XCD + x
            words[b >> 5] |= 1 << (b & 31)
    return words


def masked_stream(device, first, count):
    """A torch stream object over a HIP stream restricted to CUs [first, first + count) of every XCD (never destroyed: the
    handful of streams a process makes live as long as it does)."""
    import ctypes as C
    from drba_amd import _lib
    words = cu_mask_words(first, count)
    ptr = C.c_void_p()
    with torch.cuda.device(device):
        _lib.check(_lib.load().drba_stream_create_cu_mask(words, len(words), C.byref(ptr)), "drba_stream_create_cu_mask")
    return torch.cuda.ExternalStream(ptr.value, device=device)


def shared_stream(device, role):
    """ONE extra HIP stream per (device, role) for every model instance of the process ("side": the lookahead's chain,
    "prefetch": encoders / coarse flows of frames read ahead).  HIP multiplexes streams onto a handful of hardware queues;
    a process that builds several models (bench.py's extra configs, a test session) would otherwise create streams until
    a side stream shares its queue with the main stream and the overlap is silently gone (measured: GMFSS_UNION 41.6 ->
    35.0 frames/s as the 7th stream of the process)."""
    if ONE_STREAM:
        return torch.cuda.current_stream(device)
    key = (device.index, role)
    s = _STREAMS.get(key)
    if s is None:
        if role in CU_MASK:
            s = _STREAMS[key] = masked_stream(device, *CU_MASK[role])
        else:
            s = _STREAMS[key] = torch.cuda.Stream(device=device, priority=PRIORITY.get(role, 0))
    return s


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from _tensors(y)
    elif isinstance(x, dict):
        for y in x.values():
            yield from _tensors(y)


def split(lookahead):
    """`lookahead` argument of inference_ts_drba: the next frame, or (next frame, timesteps of the next call)."""
    if isinstance(lookahead, (tuple, list)):
        return lookahead[0], lookahead[1]
    return lookahead, None


class Lookahead:
    def __init__(self):
        self.side = None
        self.pending = None  # (frame a, frame b, result, completion event on the side stream)

    def start(self, a, b, fn, inputs=()):
        """Run fn() on the side stream after everything enqueued so far on the caller's stream; keep its result for
        take(a, b).  `inputs`: further tensors fn reads (their memory must outlive the side-stream work)."""
        if not a.is_cuda:
            return
        main = torch.cuda.current_stream(a.device)
        if self.side is None:
            self.side = shared_stream(a.device, "side")  # (stream priorities measured: no effect on the step)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            for t in _tensors((a, b, inputs)):
                t.record_stream(self.side)
            res = fn()
            done = torch.cuda.Event()
            done.record(self.side)
        for t in _tensors(res):
            t.record_stream(main)  # consumed on the caller's stream by the next step
            fp = getattr(t, "_drba_pair", None)  # the pair-interleaved copy hung on a feature tensor (ops.pair_interleaved):
            if fp is not None:                   # allocated on the side / prefetch stream, read by the main stream's gathers
                fp.record_stream(main)
        self.pending = (a, b, res, done)

    def take(self, a, b):
        """The prefetched result for the pair (a, b), or None.  The caller's stream is made to wait for it."""
        pend, self.pending = self.pending, None
        if pend is not None and pend[0] is a and pend[1] is b:
            torch.cuda.current_stream(a.device).wait_event(pend[3])
            return pend[2]
        return None
