#!/usr/bin/env python3
"""DRBA hot-path benchmark: interpolated frames/s, `rife -t 2`, 1080p (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1080p|4k|480p] [--no-cpu-baseline]

A *step* = one source frame of the driver's steady state: the next uint8 frame (already resident
in HBM) -> to_inp (u8->fp32 + bilinear resize to the network size) -> warm
RIFE.inference_ts_drba(I0, I1, I2, ts=[0.75, 1.25], reuse, linear=True) -> to_out for the two
model-generated frames (resize back + *255 truncation, still on the device).  Decode/encode and
PCIe are outside the metric.  value = model-generated frames of ALL ranks / max-rank time.
As in drba_amd.infer.interpolate_stream the loop reads one frame ahead, so the next step's coarse flow runs on a
side stream under this step's interpolation (`--no-lookahead` disables it); every frame is converted and encoded
exactly once either way, and the K timed steps contain K coarse-flow computations.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): frame-level data parallelism, every
rank interpolates its own contiguous shard of the clip (weak scaling: per-GPU work fixed); the only
collective in the data path is the RCCL gather of the finished uint8 frames to rank 0 (the writer),
inside the timed region.

Extra objects on the JSON line:
  roofline     the kernel with the largest total time among those that dominate the rocprof trace
               (profiles/): algorithmic bytes (or FLOPs) per launch / average launch duration from HIP
               events attached to those launches' dispatch packets on the launch stream during the timed
               region (the kernel's own execution time, as rocprofv3's kernel trace reports it); peaks from
               MI355X_MICROARCH.md (HBM 8 TB/s; dense fp32 MFMA 157.3 TFLOP/s).  `others` lists the next ones.
  cpu_baseline the fp32 CPU oracle (a port validated against the reference) on the same workload,
               bounded sample, host cores stated.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from drba_amd.utils import synth  # noqa: E402

CONFIGS = {
    #            src (H, W)     scale  description
    "1080p": ((1080, 1920), 1.0, "rife -t 2, 1080p synthetic (net 1088x1920), scale 1.0"),
    "4k": ((2160, 3840), 0.5, "rife -t 2, 4K synthetic (net 2176x3840), scale 0.5"),
    "480p": ((480, 854), 1.0, "rife -t 2, 480p synthetic (net 512x896), scale 1.0"),
    "4k_s1": ((2160, 3840), 1.0, "rife -t 2, 4K synthetic (net 2176x3840), scale 1.0"),
}
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense; the split-bf16 convolution spends 6 bf16 MFMA products per fp32 multiply
N_FP32_CONV_CFGS = 14  # drba_conv3x3 cfg ids below this are the fp32 MFMA kernels, the rest the split-bf16 family
HBM_PEAK_GBS = 8000.0
TS = np.array([0.75, 1.25])  # what `-t 2` yields every step (infer.py:76-87)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="1080p", choices=sorted(CONFIGS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=1)
    p.add_argument("--cpu-threads", type=int, default=32, help="threads for the CPU baseline (capped by affinity)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-lookahead", action="store_true", help="do not overlap the next step's coarse flow (A/B runs)")
    return p.parse_args()


def make_frames_u8(n, h, w, seed):
    """n distinct uint8 HWC frames; built from a short seeded clip cycled with a roll so the content keeps moving."""
    base = synth.make_clip(min(n, 8), h, w, seed=seed)
    out = []
    for k in range(n):
        f = base[k % len(base)]
        out.append(np.roll(f, (k // len(base)) * 3, axis=1) if k >= len(base) else f)
    return out


def gpu_leg(args, rank, world):
    import torch.distributed as dist

    from drba_amd import ops
    from drba_amd.models.rife import RIFE
    from drba_amd.models.utils import tools

    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    (H, W), scale, desc = CONFIGS[args.config]
    model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
    size = tools.get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), model.scale, div=model.pad_size)
    src_size, dst_size = size["src_size"], size["dst_size"]

    n_steps = args.warmup + args.steps
    frames = [torch.from_numpy(f).to(dev) for f in make_frames_u8(min(n_steps + 2, 12), H, W, seed=1234 + rank)]
    nf = len(frames)

    def to_inp(k):
        return ops.resize_bilinear(ops.u8hwc_to_f32nchw(frames[k % nf]), dst_size)

    def to_out(x):
        return ops.f32nchw_to_u8hwc(ops.resize_bilinear(x, src_size))

    I0, I1 = to_inp(0), to_inp(1)
    state = {"I0": I0, "I1": I1, "reuse": None, "k": 2}
    sink = []

    lookahead = not args.no_lookahead

    def step():
        # the driver reads one frame ahead (as drba_amd.infer.interpolate_stream does): the next step's coarse flow
        # overlaps this step's interpolation on a side stream; every frame is still converted / encoded exactly once
        I2 = state.pop("next", None)
        if I2 is None:
            I2 = to_inp(state["k"])
        nxt = to_inp(state["k"] + 1) if lookahead else None
        out, state["reuse"] = model.inference_ts_drba(state["I0"], state["I1"], I2, TS, state["reuse"], linear=True,
                                                      lookahead=None if nxt is None else (nxt, TS))
        for x in out:
            sink.append(to_out(x))
        state["I0"], state["I1"] = state["I1"], I2
        state["next"] = nxt
        state["k"] += 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sink.clear()
    timing = None
    if not args.no_roofline:
        # time only the kernels that dominate the rocprof trace: the stage-input gather (HBM-bound) and the large
        # ResConv layers (MFMA-bound).  The event pair of a timed launch is attached to the launch's own dispatch
        # packet inside the library (no barrier packets around it), and only every `roof_every`-th step is timed
        def want(kind, key):
            if kind == "ifblock_input":
                return key[0] == 52
            return kind == "conv3x3" and key[1] == key[2] and key[5] == 1 and key[3] * key[4] >= 30000
        timing = {"want": want, "records": []}
    roof_every = max(1, min(10, args.steps // 2))
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ops.TIMING = timing if (timing is not None and k % roof_every == 0) else None
        step()
    ops.TIMING = None
    t_host = time.perf_counter() - t0  # all launches enqueued: the host side of a step (the GPU may still be working)
    if world > 1:  # the only data-path collective: finished frames -> the writer rank
        mine = torch.stack(sink)
        gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gathered, dst=0)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roof = None
    if timing and timing["records"]:
        agg = {}
        for kind, key, work, unit, slot in timing["records"]:
            a = agg.setdefault((kind, key), [0.0, 0, work, unit, []])
            d = ops.timing_ms(slot)
            a[0] += d
            a[1] += 1
            a[4].append(d)
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # HBM bytes per launch from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))

        def entry(k, v):
            ms, cnt, work, unit, durs = v
            # average launch duration over the launches within 1.5x of the median (guards against a preempted launch)
            med = sorted(durs)[len(durs) // 2]
            good = [d for d in durs if d <= 1.5 * med] or durs
            avg_s = sum(good) / len(good) / 1e3
            extra = {}
            if unit == "flop":
                split = k[0] == "conv3x3" and k[1][0] >= N_FP32_CONV_CFGS
                # algorithmic fp32 flops of the layer against the matrix-core peak of the kernel that ran: fp32 MFMA, or
                # for the split-bf16 family (six bf16 MFMA products per fp32 product) the dense bf16 peak / 6
                peak = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1) if split else FP32_MFMA_PEAK_TFLOPS
                ach, u, bound = work / avg_s / 1e12, "TFLOP/s", "mfma"
                kern = "conv_split_mfma" if split else "conv_mfma"
                name = f"{kern} {k[1][1]}->{k[1][2]}ch {k[1][3]}x{k[1][4]} s{k[1][5]} N{k[1][6]} (ResConv)"
                extra = {"peak_basis": "dense bf16 MFMA 2500 TFLOP/s / 6 products per fp32 multiply" if split
                         else "dense fp32 MFMA"}
            else:
                ach, peak, u, bound = work / avg_s / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
                name = f"ifblock_input_kernel<true> {k[1][0]}ch {k[1][1]}x{k[1][2]} -> {k[1][3]}x{k[1][4]}"
            return {"bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": u, "frac": round(ach / peak, 4),
                    "traffic": traffic.get(name), "kernel": name, "launches": cnt, "avg_us": round(avg_s * 1e6, 2),
                    "algorithmic_per_launch": work, "cfg": k[1][0] if unit == "flop" else None,
                    "ms_per_step": round(avg_s * 1e3 * cnt / n_instr, 3), **extra}

        n_instr = len(range(0, args.steps, roof_every))  # instrumented steps
        ranked = sorted(agg.items(), key=lambda kv: -kv[1][0])
        roof = entry(*ranked[0])
        roof["others"] = [entry(k, v) for k, v in ranked[1:4]]
    frames_per_step = len(TS)
    return {"dt": dt, "host_dt": t_host, "frames": frames_per_step * args.steps * world, "desc": desc, "dst_size": dst_size, "roofline": roof}


def cpu_leg(args):
    """The CPU baseline: the fp32 oracle (port of the reference, pinned to it by tests/golden) on the same workload.
    Bounded sample: one untimed warm-up (a calc_flow to build `reuse` + one IFNet pass so oneDNN primitives exist),
    then `--cpu-steps` timed warm steps including to_inp/to_out."""
    import oracle  # the checker, timed here as the reported CPU baseline (never the product path)
    (H, W), scale, _ = CONFIGS[args.config]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, args.cpu_threads))
    torch.set_num_threads(cores)
    ora = oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), scale)
    from drba_amd.models.utils.tools import get_valid_net_inp_size
    dst = get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), scale, div=64)["dst_size"]
    fr = make_frames_u8(3 + args.cpu_steps, H, W, seed=1234)

    def to_inp(f):
        return oracle.ops.resize(torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0, dst)

    I = [to_inp(f) for f in fr]
    with torch.no_grad():
        flow12, flow21, f1, f2 = ora.calc_flow(I[1], I[2])  # untimed: what the previous step would have left behind
        reuse = (flow21, flow12, f2, f1)
        oracle.ifnet.ifnet(ora.sd, torch.cat((I[1], I[2]), 1), 0.5, ora.scale_list, f0=f1, f1=f2)
        t0 = time.perf_counter()
        n = 0
        for k in range(args.cpu_steps):
            out, reuse = ora.inference_ts_drba(I[k + 1], I[k + 2], I[k + 3], TS, reuse, True)
            for x in out:
                (oracle.ops.resize(x, (H, W))[0].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
            n += len(out)
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} warm inference_ts_drba step(s) = {n} frames at {dst[0]}x{dst[1]} incl. to_inp/to_out, after an "
                      f"untimed warm-up; torch {torch.__version__} CPU fp32, {cores} threads of {avail} available"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group(backend="nccl")  # RCCL over xGMI
    r = gpu_leg(args, rank, world)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_leg(args)
    if rank == 0:
        line = {
            "metric": "interpolated frames/sec @1080p RIFE x2" if args.config == "1080p" else f"interpolated frames/sec @{args.config} RIFE x2",
            "value": round(r["frames"] / r["dt"], 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["dt"] / args.steps * 1e3, 3),
            "host_ms_per_step": round(r["host_dt"] / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["desc"] + "; warm inference_ts_drba ts=[0.75,1.25] + to_inp/to_out on device",
                       "net_size": list(r["dst_size"]), "frames_per_step": len(TS), "weights": "seeded random IFNet 4.26-heavy",
                       "parallelism": f"frame-sharded dp{world}"},
            "roofline": r["roofline"], "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
