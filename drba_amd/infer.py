"""DRBA command line and per-frame driver loop (same CLI as the reference's infer.py).

    python infer.py -m rife -i in.npz -o out.npz [-fps 60 | -t 2] [-s] [-st 0.3] [-hw] [-scale 1.0]

The driver logic is host-side Python like the reference; the model calls it makes run on
the HIP library.  `interpolate_stream` is the loop of reference infer.py:58-174 with the
module globals turned into arguments, so tests can drive it with any model object.
"""
import argparse
import os
import time
import weakref

from drba_amd.models.utils import tools as _tools


def parse_args(argv=None):
    """Same flags, defaults and dest names as reference infer.py:18-36."""
    p = argparse.ArgumentParser(description="Interpolation a video with DRBA")
    p.add_argument("-m", "--model_type", dest="model_type", type=str, default="rife",
                   help="model network type, current support rife/gmfss/gmfss_union")
    p.add_argument("-i", "--input", dest="input", type=str, default="input.mp4", help="absolute path of input video")
    p.add_argument("-o", "--output", dest="output", type=str, default="output.mp4", help="absolute path of output video")
    p.add_argument("-fps", "--dst_fps", dest="dst_fps", type=float, default=60, help="interpolate to ? fps")
    p.add_argument("-t", "--times", dest="times", type=int, default=-1, help="interpolate to ?x fps")
    p.add_argument("-s", "--enable_scdet", dest="enable_scdet", action="store_true", default=False,
                   help="enable scene change detection")
    p.add_argument("-st", "--scdet_threshold", dest="scdet_threshold", type=float, default=0.3,
                   help="ssim scene detection threshold")
    p.add_argument("-hw", "--hwaccel", dest="hwaccel", action="store_true", default=False,
                   help="enable hardware acceleration encode")
    p.add_argument("-scale", "--scale", dest="scale", type=float, default=1.0,
                   help="flow scale, generally use 1.0 with 1080P and 0.5 with 4K resolution")
    return p.parse_args(argv)


def load_model(model_type, scale=1.0, device=None, weights=None):
    """Model factory (reference infer.py:39-55); unknown type -> ValueError."""
    kw = {} if device is None else {"device": device}
    if model_type == "rife":
        from drba_amd.models.rife import RIFE
        return RIFE(weights=weights or r"weights/train_log_rife_426_heavy", scale=scale, **kw)
    if model_type == "gmfss":
        from drba_amd.models.gmfss import GMFSS
        return GMFSS(weights=weights or r"weights/train_log_gmfss", scale=scale, **kw)
    if model_type == "gmfss_union":
        from drba_amd.models.gmfss_union import GMFSS_UNION
        return GMFSS_UNION(weights=weights or r"weights/train_log_gmfss_union", scale=scale, **kw)
    raise ValueError(f"model_type must in {model_type}")


def interpolate_stream(model, video_io, dst_fps, times=-1, enable_scdet=False, scdet_threshold=0.3,
                       to_inp=None, to_out=None, check_scene=None, on_step=None):
    """Run the whole clip.  Returns the number of frames written.

    Schedule quirks kept from the reference: calc_t is evaluated at an
    index one behind the centre frame in the loop and tail (infer.py:118,159); the
    left/right split uses `ts < 1` when the left pair is unusable and `ts <= 1` when the
    right pair is (infer.py:102-103,127-128 vs :135-136,160-161); after any scene cut the
    model's `reuse` state is dropped.
    """
    to_inp = to_inp or _tools.to_inp
    to_out = to_out or _tools.to_out
    # the library's own scene test can be asked for ahead of its use (tools.SceneChecks); an injected one is called in place
    ahead_checks = _tools.SceneChecks(scdet_threshold) if (check_scene is None and enable_scdet) else None
    check_scene = check_scene or _tools.check_scene
    src_fps = video_io.src_fps
    if dst_fps <= src_fps:
        raise ValueError(f"dst fps should be greater than src fps, but got dst_fps={dst_fps} and src_fps={src_fps}")

    written = 0

    def emit(frames, src_size):
        nonlocal written
        for x in frames:
            video_io.write_frame(to_out(x, src_size))
            written += 1

    i0, i1 = video_io.read_frame(), video_io.read_frame()
    size = _tools.get_valid_net_inp_size(i0, model.scale, div=model.pad_size)
    src_size, dst_size = size["src_size"], size["dst_size"]
    I0, I1 = to_inp(i0, dst_size), to_inp(i1, dst_size)
    mapper = _tools.TMapper(src_fps, dst_fps, times)
    idx = 0

    # ---- head: frames before/around the first source frame
    ts = _tools.calc_t(idx, times, mapper)
    cut_left = bool(check_scene(I0, I1, scdet_threshold)) if enable_scdet else False
    reuse = None
    if cut_left:
        out = [I0 for _ in ts]
    else:
        out = [I0 for _ in ts[ts < 1]]
        out.extend(model.inference_ts(I0, I1, ts[ts >= 1] - 1))
    emit(out, src_size)
    if on_step:
        on_step(idx)

    # ---- steady state: one (I0, I1, I2) triplet per source frame.  The loop reads one frame ahead of the reference's
    # (same frames, same order, same outputs): a model that supports it starts the next step's coarse flow on a side
    # stream while this step's frames are synthesised (RIFE.inference_ts_drba(..., lookahead=)).
    can_look = bool(getattr(model, "supports_lookahead", False))
    prefetch = getattr(model, "prefetch_frame", None) if can_look else None

    prefetch_pair = getattr(model, "prefetch_pair", None) if prefetch is not None else None
    eof, last = [False], [I1]
    # Frame intake on its own stream (a model that prefetches offers one: RIFE.intake_stream = its prefetch stream): to_inp, the
    # scene test of the pair the new frame closes and the frame's encoder depend on nothing but the frame, and the driver needs the
    # test's DECISION before it can announce the frame as part of a group of steps.  Enqueued on the caller's stream they sat behind
    # every synthesis kernel issued so far, and waiting for the decision drained that whole queue once per iteration: with groups
    # (a seven-frame window, the newest pair asked about at once) the host and the GPU ran in lock step -- 6.4 ms of host per
    # 1080p step.  On the intake stream the decision is ~100 us away whatever the main stream has queued.
    intake = getattr(model, "intake_stream", None) if prefetch is not None else None
    intake_s = None
    if intake is not None and getattr(I1, "is_cuda", False):
        import torch
        intake_s = intake(I1.device)
        if intake_s is not None:
            intake_s.wait_stream(torch.cuda.current_stream(I1.device))  # I1 (the first pair's older frame) was made on the caller's stream

    def take_in(raw, made=None):
        """to_inp + the pair's scene test + the model's prefetches for a newly read frame (made(x): called right behind to_inp)."""
        x = to_inp(raw, dst_size)
        if made is not None:
            made(x)
        if ahead_checks is not None and x.is_cuda:
            ahead_checks.submit((id(last[0]), id(x)), last[0], x)  # the cut test of this pair: asked for in this or a later iteration
        if prefetch is not None:
            prefetch(x)
            if prefetch_pair is not None:
                prefetch_pair(last[0], x)
        return x

    def read():  # -> (raw frame, network input); a model that can starts the new frame's encoder (and the coarse flow
        raw = None if eof[0] else video_io.read_frame()  # of the pair it forms with the frame before it) right away
        if raw is None:  # (the source is not asked again once it has ended)
            eof[0] = True
            return None, None
        if intake_s is not None:
            import torch
            main = torch.cuda.current_stream(I1.device)
            def made(x):
                # the frame is consumed on the caller's stream later (the model's kernels, to_out of a pass-through copy): that
                # stream waits for to_inp -- an event wait behind a queue that is far from reaching the frame -- and the
                # allocator is told about the second stream
                if x.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record(intake_s)
                    main.wait_event(ev)
                    x.record_stream(main)
                    x4 = getattr(x, "_drba_x4", None)
                    if x4 is not None:
                        x4[0].record_stream(main)
            with torch.cuda.stream(intake_s):
                x = take_in(raw, made)
        else:
            x = take_in(raw)
        last[0] = x
        return raw, x

    cuts = {}  # (id(a), id(b)) -> (decision, weakref(a), weakref(b)): every pair is tested ONCE, as in the reference loop

    def is_cut(a, b):
        if ahead_checks is not None and a.is_cuda:
            return ahead_checks.cut((id(a), id(b)), a, b)
        # an injected check_scene (or CPU frames): called in place, but a pair that moves through the look-ahead window is
        # still asked about in several iterations -- memoised here, so that a stateful / counting check sees each pair once
        k = (id(a), id(b))
        c = cuts.get(k)
        if c is not None and c[1]() is a and c[2]() is b:
            return c[0]
        res = bool(check_scene(a, b, scdet_threshold))
        try:
            cuts[k] = (res, weakref.ref(a), weakref.ref(b))
        except TypeError:  # frames that cannot be weakly referenced (plain ndarrays in a test double): not memoised
            return res
        while len(cuts) > 32:
            cuts.pop(next(iter(cuts)))
        return res

    # A model that can look ahead gets the loop reading THREE frames ahead (same frames, same order, same outputs): (i3, I3)
    # is the lookahead frame of this iteration; a model that can (RIFE: `prefetch_frame`) has the encoder and the coarse flow
    # of every frame started the moment it is read, and is told the frames and timesteps of the next iterations so that it
    # may compute several consecutive steps in one stacked pass (RIFE._drba_group): the next iterations then only collect.
    group = int(getattr(model, "GROUP", 1)) if prefetch is not None else 1
    depth = max(3, 2 * group - 1) if prefetch is not None else (1 if can_look else 0)
    i2, I2 = read()
    ahead = []  # [(raw, tensor)] of the frames after I2, oldest first
    while len(ahead) < depth and not eof[0] and i2 is not None:
        r, x = read()
        if r is None:
            break
        ahead.append((r, x))
    cut_next = None  # scene cut between I2 and the frame after it, when it was already evaluated
    while i2 is not None:
        I3 = ahead[0][1] if ahead else None
        ts = _tools.calc_t(idx, times, mapper)
        if cut_next is not None:
            cut_right = cut_next
        else:
            cut_right = is_cut(I1, I2) if enable_scdet else False
        cut_next = None
        if cut_left and cut_right:
            out, reuse = [I1 for _ in ts], None
        elif cut_left:
            reuse = None
            out = [I1 for _ in ts[ts < 1]]
            out.extend(model.inference_ts(I1, I2, ts[ts >= 1] - 1))
        elif cut_right:
            reuse = None
            out = model.inference_ts(I0, I1, ts[ts <= 1])
            out.extend([I1 for _ in ts[ts > 1] - 1])
        elif can_look and I3 is not None:
            look = (I3, _tools.calc_t(idx + 1, times, mapper))
            if prefetch is not None and group > 1:
                # the following iterations, as far as they are DRBA steps too (no cut up to the last frame named): the model may
                # take them in one stacked pass with this one and stage the group after them
                entries, prev = [], I2
                for j, (_, x) in enumerate(ahead):
                    c = is_cut(prev, x) if enable_scdet else False
                    if j == 0:
                        cut_next = c
                    if c:
                        break
                    entries += [x, _tools.calc_t(idx + 1 + j, times, mapper)]
                    prev = x
                if len(entries) >= 4:
                    look = tuple(entries)
            out, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, linear=True, lookahead=look)
        else:
            out, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, linear=True)
        emit(out, src_size)
        I0, I1 = I1, I2
        i2, I2 = ahead.pop(0) if ahead else (read() if depth == 0 else (None, None))
        if depth and not eof[0]:
            r, x = read()
            if r is not None:
                ahead.append((r, x))
        cut_left = cut_right
        idx += 1
        if on_step:
            on_step(idx)

    # ---- tail: the last pair
    ts = _tools.calc_t(idx, times, mapper)
    out = model.inference_ts(I0, I1, ts[ts <= 1])
    out.extend([I1 for _ in ts[ts > 1] - 1])
    emit(out, src_size)
    if on_step:
        on_step(idx + 1)
    return written


def inference(model, args):
    video_io = _tools.VideoFI_IO(args.input, args.output, dst_fps=args.dst_fps, times=args.times, hwaccel=args.hwaccel)
    try:
        from tqdm import tqdm
        bar = tqdm(total=video_io.total_frames_count)
        step = lambda _i: bar.update(1)  # noqa: E731
    except ImportError:
        bar, step = None, None
    to_out = None
    if getattr(video_io, "wants_rgb", False):  # encoder pipe / raw sink: BGR -> RGB inside the to_out kernel, not on the host
        video_io.frames_are_rgb = True
        to_out = lambda x, size: _tools.to_out(x, size, rgb=True)  # noqa: E731
    n = interpolate_stream(model, video_io, args.dst_fps, times=args.times, enable_scdet=args.enable_scdet,
                           scdet_threshold=args.scdet_threshold, to_out=to_out, on_step=step)
    while not video_io.finish_writing():
        time.sleep(0.01)
    video_io.close()
    if bar is not None:
        bar.close()
    return n


def inference_sharded(model, args, rank, world, to_inp=None, to_out=None, check_scene=None, device=None, chunk=4):
    """ONE clip frame-sharded over the ranks of a torch.distributed job (not in the reference, which has no parallelism;
    BASELINE.json configs[4]).  Every rank opens the clip (random access: .npz / .npy sources), runs its contiguous share
    of the driver loop with the one-frame halo of drba_amd.parallel.interpolate_shard, and the finished uint8 frames
    stream to rank 0 -- the writer -- through StreamedGather (the only collective: RCCL over xGMI on the GPUs).  The file
    rank 0 writes equals the single-process run's.  Returns the number of frames written (rank 0), 0 elsewhere."""
    import numpy as np

    from drba_amd import parallel
    ext = os.path.splitext(args.input)[1].lower()
    if ext == ".npz":
        z = np.load(args.input)
        frames, fps = z["frames"], (float(z["fps"]) if "fps" in z.files else 24.0)
    elif ext == ".npy":
        frames = np.load(args.input, mmap_mode="r")
        side = os.path.splitext(args.input)[0] + ".json"
        fps = 24.0
        if os.path.exists(side):
            import json
            fps = float(json.load(open(side))["fps"])
    else:
        raise RuntimeError("the frame-sharded run needs random access to the clip: use a .npz / .npy source")
    counts = parallel.emission_counts(len(frames), fps, args.dst_fps, args.times, world)
    sg = parallel.StreamedGather(rank, world, counts, chunk=chunk, device=device, frame_shape=tuple(frames[0].shape))
    if to_out is None and world > 1 and device is not None and device.type == "cuda":
        # finished frames stay on the device until the gather has moved them (tools.to_out would copy each one to the
        # host only for StreamedGather to copy it back): the uint8 frame is written straight into the send path
        from drba_amd import ops as _ops
        to_out = lambda x, size: _ops.to_out(x, size)  # noqa: E731
    parallel.interpolate_shard(model, frames, fps, args.dst_fps, rank, world, times=args.times, enable_scdet=args.enable_scdet,
                               scdet_threshold=args.scdet_threshold, to_inp=to_inp, to_out=to_out, check_scene=check_scene,
                               sink=sg.push)
    allf = sg.finish()
    if rank != 0:
        return 0
    video_io = _tools.VideoFI_IO(args.input, args.output, dst_fps=args.dst_fps, times=args.times, hwaccel=args.hwaccel)
    for f in allf:
        video_io.write_frame(f.cpu().numpy() if hasattr(f, "cpu") else f)
    while not video_io.finish_writing():
        time.sleep(0.01)
    video_io.close()
    return len(allf)


def main(argv=None):
    """`python infer.py ...` = the reference CLI.  Under torch.distributed.run (WORLD_SIZE > 1, one rank per GPU) the same
    command line shards the clip over the ranks: `python -m torch.distributed.run --nproc-per-node 8 infer.py -m rife ...`."""
    args = parse_args(argv)
    if not os.path.exists(args.input):
        raise FileNotFoundError(f"can't find the video file {args.input}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        model = load_model(args.model_type, scale=args.scale)
        return inference(model, args)
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl")  # RCCL over xGMI
    try:
        model = load_model(args.model_type, scale=args.scale, device=torch.device("cuda", local))
        return inference_sharded(model, args, rank, world, device=torch.device("cuda", local))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
