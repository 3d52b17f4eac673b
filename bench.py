#!/usr/bin/env python3
"""DRBA hot-path benchmark: interpolated frames/s, `rife -t 2`, 1080p (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1080p|4k|480p|4k_s1] [--no-cpu-baseline] [--no-extra]

A *step* = one source frame of the driver's steady state: the next uint8 frame (already resident in HBM) -> to_inp
(u8 HWC -> fp32 NCHW + bilinear resize to the network size) -> warm RIFE.inference_ts_drba(I0, I1, I2, ts=[0.75, 1.25],
reuse, linear=True) -> to_out for the two model-generated frames (resize back + *255 truncation, on the device).
Decode/encode and PCIe are outside the metric; `pcie_inclusive` reports the same loop with the frames
starting and ending in pinned host memory.  value = model-generated frames of ALL ranks / max-rank time.
As in drba_amd.infer.interpolate_stream the loop reads two frames ahead: the next step's coarse flow and
low-resolution stages run on a side stream under this step's full-resolution stages, and the frame after that has its
context encoder and the coarse flow of the pair it forms with its predecessor started on a third stream (`tools/ab_bench.py --no-lookahead` disables both); every frame is converted and encoded exactly
once either way, one new frame per step, and the K timed steps contain K steps of work.

N = 1: the K-step loop above.
N > 1 (launched by torch.distributed.run, one rank per GPU): ONE clip of N*K + 2 source frames is sharded with
drba_amd.parallel.interpolate_shard (contiguous chunks, one-frame halo for the warm `reuse`, weak scaling: K steps per
GPU), the finished uint8 frames travel to rank 0 (the writer) through StreamedGather -- the only data-path collective,
RCCL over xGMI, inside the timed region.  `replica_loop` is the N = 1 loop run by every rank on its own clip (no halo,
no collective) and `config5_sharded` BASELINE.json configs[4]: one FIXED 4K clip (-fps 60, scale 0.5, one planted scene
cut) sharded over the N ranks (strong scaling).

Extra objects on the JSON line:
  roofline      the kernel SYMBOL with the largest total time over the instrumented steps: 4 consecutive steps of the same
                loop, run right after the timed region with the library's kernel trace on for every launch (each launch
                carries an event pair on its own dispatch packet, i.e. the kernel's own execution time as rocprofv3's
                kernel trace reports it; 3 traced steps before them settle the streams' interleaving).
                achieved = sum of the launches' algorithmic FLOPs (or bytes) / sum of their durations; avg_us and
                algorithmic_per_launch are plain means over the same launches; `by_geometry` splits the symbol by layer
                shape; `others` = the next symbols; `step_kernels_ms` = all kernel time per step, `traced_ms_per_step` the
                wall time of a traced step and `avg_concurrency` their ratio: how many kernels (three streams) share the
                chip on average -- a kernel's duration, hence `frac`, is that of the shared execution, about
                avg_concurrency times its stand-alone duration; `standalone` = the heaviest geometry of the symbol launched
                back to back on the idle GPU right after (stride-1 conv layers).  Peaks from
                MI355X_MICROARCH.md (HBM 8 TB/s; dense fp32 MFMA 157.3 TFLOP/s; dense bf16 2500 TFLOP/s).
                `traffic` = HBM bytes per launch from rocprofv3 --pmc passes (profiles/pmc_traffic.json, keyed by symbol
                and launch label; tools/pmc_targets.py + tools/pmc_traffic.py regenerate it), null where a geometry the
                symbol ran with was not measured.
  cpu_baseline  the fp32 CPU oracle (a port pinned bit-for-bit to the reference, tests/golden) on the same workload:
                bounded sample, all host cores (`value`) and one thread (`single_thread`).
  max_abs_vs_oracle  max |HIP frame - oracle frame| over the frames of the cpu_baseline sample, same uint8 inputs
                (fp32 at network size), plus the largest difference of the uint8 outputs in LSB.
  extra_configs BASELINE.json configs[2], [3], [4] at N = 1 through the real driver loop (interpolate_stream) with the
                clip resident in HBM, each with its own roofline entry, `path` (which of the model's paths the timed calls
                took), `host_ms_per_step` and `group`; 40 timed iterations, an untimed rehearsal clip in front of the
                scene-detection legs (clip_leg).
  The timed `-t 2` loop keeps at most kMaxStepsInFlight steps queued (step_loop): the frames stay on the device, so nothing
  else would stop the host from queueing the whole HBM full of pending steps at 4K.
"""
import argparse
import collections
import gc
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense; the split-bf16 kernels spend 6 bf16 MFMA products per fp32 multiply
HBM_PEAK_GBS = 8000.0
TS = np.array([0.75, 1.25])  # what `-t 2` yields every step (infer.py:76-87)
# The roofline object's kernel durations come from a block of consecutive fully traced steps run right AFTER the timed
# region (kTraceWarm to settle, kTraceKeep kept), so the timed steps run untraced.  What the records mean with three
# streams: a step's kernels sum to ~6.9 ms inside a ~3.0 ms step, i.e. 2.3 kernels share the chip on average and each
# reads about twice its stand-alone duration (32-channel ResConv: 68 us here, 35 us alone in tools/resconv_bench.py).
# rocprofv3's kernel trace of the same command runs the loop 13 % slower and less overlapped (4.35 ms of kernels in a
# 3.35 ms step) and reads 36 us for that layer; tracing only the main stream's launches does not change the 68 us
# (tools/exp/trace_settle.py).  `roofline.frac` is therefore the fraction under the step's real concurrency.
kTraceWarm, kTraceKeep = 3, 4
kSettleSeconds = 0.5  # untimed steps after the W warm-up steps (step_loop) / minimum warm-up iterations of the clip legs
kClipWarmup = 24
kMaxStepsInFlight = 16  # step_loop: the host waits for the step this many steps back (four groups of steps stay queued on the GPU; 32: the 4K runs stall again)
kClipSteps = 40  # timed iterations of the driver-loop legs (extra_configs): 5 groups of 4 steps on each side of the planted cut
SRC_FPS = 24.0
kMinTimedSeconds = 0.25  # the K-step block is repeated until the timed region is at least this long (K = 20 steps at 2 ms are 40 ms: box-to-box noise)
# SURVEY.md 8(d) / BASELINE.md 3: algorithmic work of ONE model-generated frame (warm), per config: (GFLOP, conv-boundary GB)
FRAME_WORK = {"1080p": (92.9, 1.59), "4k": (107.2, 2.13)}


# A/B switches between kernel variants / schedules of THIS library live in tools/ab_bench.py (it fills AB and calls main()):
# the benchmark's own command line carries the contract's flags only.
AB = {}

_T0 = time.perf_counter()
LAST_SHARD = {"rank_dt": None, "path": None}  # this rank's own wall time of the last sharded_leg (the line reports every rank's)
LAST_PATH = {}  # RIFE.stats over the timed region of the last step_loop: which path the K timed calls took (reported on the line)
LAST_BLOCKS = {"n": 1}  # K-step blocks the last step_loop timed back to back (kMinTimedSeconds)
LAST_SETTLE = {"steps": 0}  # untimed settling steps the last step_loop ran after its W warm-up steps (reported on the line)


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="1080p", choices=sorted(CONFIGS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=2)
    p.add_argument("--cpu-threads", type=int, default=16,
                   help="threads of the multi-core CPU leg (capped by the affinity mask; 0 = all).  16 threads are the fastest setting "
                        "on the 256-thread GPU box (0.42 frames/s; 8: 0.40, 32: 0.38, 1 thread 0.21, 128 threads 0.15, 256 threads "
                        "< 0.013: oneDNN/OpenMP oversubscription), profiles/r02_cpu_threads.txt (tools/cpu_threads.py)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip extra_configs / pcie_inclusive (configs 3-5 at N = 1)")
    p.add_argument("--selftest-sharded", action="store_true",
                   help="N = 1 only: run the N > 1 legs' code (sharded warm-up + sharded run of the headline clip and of the "
                        "config-5 clip) on one GPU without a process group and print their rates; not the metric")
    p.add_argument("--conv-families", default=None,
                   help="kernel families the conv autotuner may pick from (default 0,1,2,3,4); 0,1,2,3 keeps every MFMA operand "
                        "at 24 bits (three bf16 terms) instead of the two-term fp16 form -- the number of the previous rounds")
    args = p.parse_args()
    if args.conv_families is not None:
        from drba_amd import ops
        ops.CONV_FAMILIES = {int(x) for x in args.conv_families.split(",")}
    args.no_lookahead = bool(AB.get("no_lookahead", False))  # (the single-stream roofline block sets it on a copy of args)
    return args


def make_frames_u8(n, h, w, seed, cut_at=None):
    """n distinct uint8 HWC frames; built from a short seeded clip cycled with a roll so the content keeps moving.
    `cut_at`: frames from that index on come from an independently seeded scene (scene-cut configs)."""
    base = synth.make_clip(min(n, 8), h, w, seed=seed)
    other = synth.make_clip(min(n, 8), h, w, seed=seed + 7919) if cut_at is not None else None
    out = []
    for k in range(n):
        src = other if (cut_at is not None and k >= cut_at) else base
        f = src[k % len(src)]
        out.append(np.roll(f, (k // len(src)) * 3, axis=1) if k >= len(src) else f)
    return out


class DeviceClip:
    """make_frames_u8 as a random-access sequence of uint8 HWC tensors resident in HBM (only 8 (+8) base frames are
    stored; frame k is a view or a roll of one of them), so a rank touches only its own range of a long clip."""

    def __init__(self, n, h, w, seed, dev, cut_at=None, pingpong=False):
        # pingpong (the scene-detection legs): the 8 base frames are walked 0..7, 6..1, 0.. so that consecutive frames are ALWAYS
        # neighbours of the base clip.  The cyclic order jumps back by 13 motion steps from frame 8 j + 7 to 8 j + 8, and at 4K
        # check_scene reads that jump as a cut (SSIM of the 32 x 32 thumbnails < 0.3): round 5's "one planted cut" clip of config 5
        # held five detected cuts in 40 iterations -- a quarter of its steps were cold restarts
        self.n, self.cut_at, self.pingpong = n, cut_at, pingpong
        self.base = [torch.from_numpy(f).to(dev) for f in synth.make_clip(min(n, 8), h, w, seed=seed)]
        self.other = ([torch.from_numpy(f).to(dev) for f in synth.make_clip(min(n, 8), h, w, seed=seed + 7919)]
                      if cut_at is not None else None)
        self.shape = tuple(self.base[0].shape)

    def __len__(self):
        return self.n

    def __getitem__(self, k):
        if not 0 <= k < self.n:
            raise IndexError(k)
        if isinstance(self.cut_at, (list, tuple)):  # several cuts: the two scenes alternate
            src = self.other if sum(1 for c in self.cut_at if k >= c) % 2 else self.base
        else:
            src = self.other if (self.cut_at is not None and k >= self.cut_at) else self.base
        if self.pingpong and len(src) > 1:
            period = 2 * len(src) - 2
            j = k % period
            return src[j if j < len(src) else period - j]
        f = src[k % len(src)]
        return torch.roll(f, (k // len(src)) * 3, dims=1) if k >= len(src) else f


class _Counting:
    """Wraps a model: counts the frames it SYNTHESISES (pass-through copies at t in {0, 1, 2} are not model output)."""

    def __init__(self, m):
        self.m, self.generated = m, 0
        self.scale, self.pad_size = m.scale, m.pad_size
        self.supports_lookahead = bool(getattr(m, "supports_lookahead", False))

    def inference_ts(self, I0, I1, ts):
        self.generated += sum(1 for t in ts if t not in (0, 1))
        return self.m.inference_ts(I0, I1, ts)

    def inference_ts_drba(self, I0, I1, I2, ts, reuse=None, linear=False, **kw):
        self.generated += sum(1 for t in ts if t not in (0, 1, 2))
        return self.m.inference_ts_drba(I0, I1, I2, ts, reuse, linear, **kw)

    def warm_reuse(self, a, b):
        return self.m.warm_reuse(a, b)

    def __getattr__(self, name):  # optional driver hooks (prefetch_frame): present only if the model has them.  GROUP: what the
        # drivers size their read-ahead by -- rounds 3-5 did not pass it on, so every driver-loop leg (configs 3, 4, 5 and the sharded
        # legs) ran WITHOUT groups of steps (2-item launches, 83 kernels per step) while the shipped CLI, which hands the model itself
        # to interpolate_stream, forms them: the 876 / 449 frames/s of BENCH_r05's configs 3 / 5 against 1139 / 562 were this wrapper
        if name in ("prefetch_frame", "prefetch_pair", "GROUP", "stats", "intake_stream"):
            return getattr(self.m, name)
        raise AttributeError(name)


class _DevIO:
    """VideoFI_IO's read/write surface over a clip resident in HBM (decode/encode are outside the metric)."""

    def __init__(self, clip, fps):
        self.src_fps, self.total_frames_count = fps, len(clip)
        self.clip, self.k, self.written = clip, 0, 0
        self.last = None

    def read_frame(self):
        if self.k >= len(self.clip):
            return None
        self.k += 1
        return self.clip[self.k - 1]

    def write_frame(self, x):
        self.written += 1
        self.last = x


def _dev_hooks():
    from drba_amd import ops

    def to_inp(frame_u8, dst_size):
        return ops.to_inp(frame_u8, dst_size)

    def to_out(x, src_size):
        return ops.to_out(x, src_size)

    return to_inp, to_out


class AnnouncedLoop:
    """The read-ahead of the shipped driver (drba_amd.infer.interpolate_stream) over a random-access clip, without its
    head / tail / scene logic: step k = inference_ts_drba(F[k], F[k+1], F[k+2], ts(k), reuse, linear=True, lookahead=...).
    The frames of the next steps are fetched (`get_frame(k)`: the network input of source frame k, None past the end) as
    far ahead as the model's groups need -- 2 GROUP - 1 frames --, their context encoders and coarse flows are started on
    the prefetch stream (`prefetch_frame` / `prefetch_pair`) and the model is told the frames and timesteps of the
    following calls; it computes RIFE.GROUP consecutive steps in one stacked pass and stages the low-resolution part of
    the group after them on a side stream.  Every frame is fetched / encoded exactly once, one per step.
    THE loop of the timed region (step_loop), of `max_abs_vs_oracle` (cpu_leg) and of tests/test_gpu_fullsize.py: what is
    checked against the oracle is what is timed."""

    def __init__(self, model, get_frame, get_ts, lookahead=True, reuse=None, first=0):
        self.model, self.get_frame, self.get_ts = model, get_frame, get_ts
        self.lookahead = lookahead
        self.prefetch = getattr(model, "prefetch_frame", None) if lookahead else None
        self.prefetch_pair = getattr(model, "prefetch_pair", None) if lookahead else None
        self.I0, self.I1 = get_frame(first), get_frame(first + 1)
        self.k, self.reuse = first, reuse  # step k uses frames k, k + 1, k + 2
        self.next, self.ahead, self.eof = None, [], False

    def step(self):
        """-> (outputs of step k, its timesteps); advances to step k + 1."""
        model, k = self.model, self.k
        I2 = self.next if self.next is not None else self.get_frame(k + 2)
        ahead = self.ahead  # network inputs of frames k + 3, k + 4, ...
        if self.lookahead:
            depth = max(3, 2 * int(getattr(model, "GROUP", 1)) - 1) if self.prefetch is not None else 1
            while len(ahead) < depth and not self.eof:
                x = self.get_frame(k + 3 + len(ahead))
                if x is None:
                    self.eof = True
                    break
                if self.prefetch is not None:
                    self.prefetch(x)
                    if self.prefetch_pair is not None:
                        self.prefetch_pair(ahead[-1] if ahead else I2, x)
                ahead.append(x)
        ts = self.get_ts(k)
        look = None
        if ahead:
            look = (ahead[0], self.get_ts(k + 1))
            if len(ahead) >= 3:  # (frame, ts) of the following steps: all of them DRBA steps (no scene detection in this loop)
                look = tuple(v for j, x in enumerate(ahead) for v in (x, self.get_ts(k + 1 + j)))
        out, self.reuse = model.inference_ts_drba(self.I0, self.I1, I2, ts, self.reuse, linear=True, lookahead=look)
        self.I0, self.I1 = self.I1, I2
        self.next = ahead.pop(0) if ahead else None
        self.k += 1
        return out, ts


# ------------------------------------------------------------------------------------------------- roofline
def _peak(name, unit):
    if unit == "byte":
        return HBM_PEAK_GBS, "GB/s", "hbm", "HBM3E 8 TB/s"
    if "split" in name or "conv_dma" in name or "conv_ks" in name or "window_attention16" in name or "head_fused16" in name:
        if _two_term(name):  # fp32 operands as two fp16 terms (22 bits), three MFMA products
            return round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1), "TFLOP/s", "mfma", "dense f16 MFMA 2500 TFLOP/s / 3 products per fp32 multiply"
        # fp32 operands as three bf16 terms, six MFMA products
        return round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "TFLOP/s", "mfma", "dense bf16 MFMA 2500 TFLOP/s / 6 products per fp32 multiply"
    return FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", "mfma", "dense fp32 MFMA"


def _arithmetic_note():
    """What `dtype` "f32" stands for on this run: which kernel families the conv autotuner was allowed (drba_amd.ops.CONV_FAMILIES)."""
    from drba_amd import ops
    two = 4 in ops.CONV_FAMILIES
    return {"tensors": "fp32 in HBM, fp32 accumulation, fp32 outputs",
            "mfma_operands": ("stride-1 / transposed / most stride-2 convolutions, the encoder, GMFlow's linears and window attention: each "
                              "fp32 operand as two fp16 terms h + 2^-11 l (22 significand bits, 3 MFMA products, kernel family 4); "
                              "the fused stage kernel (stage_conv16) likewise; the largest stride-2 layers, global correlation: fp32 MFMA" if two else
                              "stride-1 / transposed convolutions and GMFlow linears: each fp32 operand as three bf16 terms (24 bits, 6 MFMA "
                              "products); the rest fp32 MFMA"),
            "conv_families": sorted(ops.CONV_FAMILIES)}


def _dtype_note():
    """The line's `dtype`: tensors, accumulation and outputs are fp32 either way; what the MFMA operands are depends on the
    kernel families allowed (the `arithmetic` object spells it out)."""
    from drba_amd import ops
    if 4 in ops.CONV_FAMILIES:
        return "f32 (2xfp16-term MFMA operands = 22 significand bits, fp32 accumulate)"
    return "f32 (24-bit MFMA operands: fp32 MFMA / 3xbf16 terms, fp32 accumulate)"


def _two_term(name):
    """True for the two-term fp16 instantiations of the split families (the last template argument, PL, is 2):
    conv_split_mfma<SplitCfg<M, RW, MW, NT, 2[, CS]>, ...>, conv_dma1<PRE, RL, 2>, conv_ks<KW, PRE, RL, DW, 2>, linear_split_kernel<LinCfg<.., 2>, ..>;
    window_attention16_kernel, head_fused16 and stage_conv16 exist in that form only.  A kernel that runs on the 16-bit matrix
    pipe is never priced against the fp32-MFMA peak."""
    import re
    # (SplitCfg<MODE, RW, MW, NT, PL[, CS]>: PL is the FIFTH argument -- since round 6 CS and BP follow it in every printed name)
    return bool(re.search(r"SplitCfg<-?\d+, \d+, \d+, \d+, 2(, \d+)*>", name) or re.search(r"conv_dma1<[^<>]*, 2>", name)
                or re.search(r"conv_ks<\d+, \w+, \w+, \w+, 2>", name) or re.search(r"LinCfg<[^<>]*, 2>", name)
                or "window_attention16" in name or "head_fused16" in name or "stage_conv16" in name)


def _symbol_totals(recs, n_steps):
    """{symbol: (ms per step, launches per step, mean us, algorithmic work per ms or None)} of a block of traced steps.
    The work rate is the block's OWN: a single-stream block launches 2 samples per kernel, the timed loop 8."""
    acc = {}
    for r in recs or []:
        a = acc.setdefault(r["name"], [0.0, 0, 0.0, 0])
        a[0] += r["ms"]
        a[1] += 1
        if r["work"] is not None:
            a[2] += r["work"]
            a[3] += 1
    return {k: (v[0] / n_steps, v[1] / n_steps, v[0] / v[1] * 1e3, (v[2] / v[0] if v[3] == v[1] and v[0] > 0 else None))
            for k, v in acc.items()}


def roofline_from_trace(recs, n_steps, traffic=None, serial=None, workload=None):
    """recs: ops.trace_end() records of `n_steps` instrumented steps of the loop as it is timed (three streams).
    serial: _symbol_totals of the same steps run on ONE stream (no lookahead, no prefetch): a kernel's duration there is its
    own execution -- in the three-stream loop a launch of a small side-stream kernel also spans the time its workgroups wait
    for a CU that the main stream's kernels hold (a 17 us layer reads 59 us) -- both durations are reported on every entry
    (serial_*); the ranking is the timed loop's."""
    if not recs or n_steps <= 0:
        return None
    agg = {}
    for r in recs:
        a = agg.setdefault(r["name"], {"ms": 0.0, "n": 0, "work": 0.0, "tagged": 0, "unit": None, "geo": {}})
        a["ms"] += r["ms"]
        a["n"] += 1
        if r["work"] is not None:
            a["work"] += r["work"]
            a["tagged"] += 1
            a["unit"] = r["unit"]
            g = a["geo"].setdefault(r["label"], [0.0, 0, 0.0])
            g[0] += r["ms"]
            g[1] += 1
            g[2] += r["work"]
    total_ms = sum(a["ms"] for a in agg.values())
    # ranked by kernel time inside the loop as it is timed: that is the symbol rocprofv3's table of this command puts on top.
    # (Round 3's first passes ranked by the single-stream block's time; since the loop launches 8 samples per kernel and the
    # single-stream block 2, the two run different configurations of the layers and only the in-loop ranking describes the
    # timed region.  The single-stream figures stay on every entry as serial_*.)
    # ... capped by the symbol's single-stream time where that is known: a side-stream kernel (the encoder on the low-priority
    # prefetch stream, the coarse stages) spends most of its in-step "duration" waiting for CUs the main stream's kernels hold
    # (round 5: head_fused16 read 526 us in the step, 83 us alone, and outranked every main-stream kernel)
    def rank_ms(kv):
        ms = kv[1]["ms"] / n_steps
        return min(ms, serial[kv[0]][0]) if serial and kv[0] in serial else ms
    ranked = sorted(agg.items(), key=lambda kv: -rank_ms(kv))

    def entry(name, a, with_geo):
        e = {"kernel": name, "launches_per_step": round(a["n"] / n_steps, 2), "avg_us": round(a["ms"] / a["n"] * 1e3, 2),
             "ms_per_step": round(a["ms"] / n_steps, 4), "share_of_kernel_time": round(a["ms"] / total_ms, 4)}
        if a["tagged"] == a["n"] and a["ms"] > 0:  # every launch of the symbol carries its algorithmic work
            peak, unit, bound, basis = _peak(name, a["unit"])
            ach = a["work"] / (a["ms"] * 1e-3) / (1e9 if a["unit"] == "byte" else 1e12)
            # HBM bytes per launch from the PMC passes: {label: bytes} for this symbol; the symbol's figure is the mean over
            # its launches when every geometry it ran was measured
            tr = (traffic or {}).get(name) or {}
            known = all(lab in tr for lab in a["geo"])
            e.update({"bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                      "algorithmic_per_launch": a["work"] / a["n"], "peak_basis": basis,
                      "traffic": (round(sum(tr[lab] * g[1] for lab, g in a["geo"].items()) / a["n"]) if known and tr else None)})
            if serial and name in serial:
                sm, sn, su, swork = serial[name]
                e["serial_avg_us"], e["serial_ms_per_step"] = round(su, 2), round(sm, 4)
                e["serial_launches_per_step"] = round(sn, 2)
                if swork is not None:  # the single-stream launches' own work over their own time
                    e["frac_serial"] = round(swork * 1e3 / (1e9 if a["unit"] == "byte" else 1e12) / peak, 4)
            if with_geo:
                e["by_geometry"] = [
                    {"launch": lab, "launches_per_step": round(g[1] / n_steps, 2), "avg_us": round(g[0] / g[1] * 1e3, 2),
                     "frac": round(g[2] / (g[0] * 1e-3) / (1e9 if a["unit"] == "byte" else 1e12) / peak, 4),
                     "algorithmic_per_launch": g[2] / g[1], "traffic": tr.get(lab)}
                    for lab, g in sorted(a["geo"].items(), key=lambda kv: -kv[1][0])[:6]]
        else:
            e.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None})
        return e

    roof = entry(*ranked[0], True)
    roof["others"] = [entry(n, a, False) for n, a in ranked[1:5]]
    # the same symbols under rocprofv3's kernel trace of this command (profiles/rocprof_frac.json, written by
    # tools/rocprof_frac.py from the steady-state table of the round's trace): the loop runs ~13 % slower and less
    # overlapped there, so a kernel reads shorter than in the step and longer than alone
    # -- ONLY when that table was taken on this workload (its "__workload__" entry): a 1080p duration under a 4K launch's
    # work is not evidence (round 3 printed fractions above 1 that way); otherwise null
    rp = _rocprof_table()
    meta = rp.get("__workload__") or {}
    # ... and only when it was taken on THESE kernel sources (round 4 quoted a table one kernel commit behind HEAD)
    same = workload is not None and meta.get("config") == workload and meta.get("csrc_sha") == csrc_sha()
    roof["rocprof_table"] = {"source": meta.get("source"), "config": meta.get("config"), "csrc_sha": meta.get("csrc_sha"),
                             "this_build_csrc_sha": csrc_sha(), "used": bool(same)}
    for e in [roof] + roof["others"]:
        r = rp.get(e["kernel"]) if same else None
        e["frac_rocprof"] = None
        if r and e.get("algorithmic_per_launch") and e.get("peak"):
            ach = e["algorithmic_per_launch"] / (r["avg_us"] * 1e-6) / (1e9 if e["unit"] == "GB/s" else 1e12)
            if ach <= e["peak"]:  # (a symbol whose launches differ in work between the two runs: no figure rather than a wrong one)
                e["frac_rocprof"] = round(ach / e["peak"], 4)
                e["rocprof_avg_us"] = r["avg_us"]
                e["rocprof_source"] = r.get("source")
    roof["step_kernels_ms"] = round(total_ms / n_steps, 3)
    if all("start_ms" in r for r in recs):  # how many kernels share the chip on average while these durations were taken
        span = max(r["start_ms"] + r["ms"] for r in recs) - min(r["start_ms"] for r in recs)
        if span > 0:
            roof["traced_ms_per_step"] = round(span / n_steps, 3)
            roof["avg_concurrency"] = round(total_ms / span, 2)
    roof["kernels_per_step"] = round(len(recs) / n_steps, 1)
    roof["instrumented_steps"] = n_steps
    roof["ranked_by"] = "kernel time per step: in-step, capped by the same symbol's single-stream time"
    if serial:
        roof["serial_step_kernels_ms"] = round(sum(v[0] for v in serial.values()), 3)
    return roof


def standalone_of(roof, dev, reps=30):
    """The dominant symbol's heaviest geometry launched back to back on an otherwise idle GPU (stride-1 conv layers only):
    the kernel's own quality, next to the in-step figure that is taken while ~2 other kernels share the chip."""
    import re
    geo = sorted((roof or {}).get("by_geometry") or [], key=lambda g: -g["algorithmic_per_launch"])  # most work first
    for g in geo:
        m = re.match(r"stage_conv(?:0|16)\+lazy \(52, 16, (\d+), (\d+), (\d+)\)$", g["launch"])
        if m:
            return _standalone_stage_conv0(roof, dev, g, *(int(v) for v in m.groups()), reps=reps)
    for g in geo:
        m = re.match(r"conv3x3 \((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\)$", g["launch"])
        if not m:
            continue
        cfg, cin, cout, ho, wo, stride, n = (int(v) for v in m.groups())
        if stride != 1:
            continue
        from drba_amd import ops
        gen = torch.Generator().manual_seed(0)
        x = torch.randn(n, cin, ho, wo, generator=gen).to(dev)
        res = cin == cout  # ResConv: lrelu(conv(x) * beta + x), the residual rebuilt from the staged input
        layer = ops.Conv3x3(torch.randn(cout, cin, 3, 3, generator=gen) / (cin * 9) ** 0.5, torch.zeros(cout), 1, True,
                            torch.ones(1, cout, 1, 1) if res else None, device=dev, cfg=cfg)
        out = torch.empty(n, cout, ho, wo, device=dev)
        run = (lambda: layer(x, residual=x, out=out)) if res else (lambda: layer(x, out=out))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        flop = 2.0 * cout * cin * 9 * ho * wo * n
        return {"launch": g["launch"], "avg_us": round(us, 2), "achieved": round(flop / us / 1e6, 2), "unit": "TFLOP/s",
                "frac": round(flop / us / 1e6 / roof["peak"], 4),
                "what": f"{reps} back-to-back launches of this layer alone (events around the run: includes the ~5 us between "
                        "dependent launches)"}
    return None


def _standalone_stage_conv0(roof, dev, g, H, W, n, reps=30):
    """drba_stage_conv16_batch / drba_stage_conv0_batch (whichever the family set selects) alone on the idle GPU: the launch geometry of the loop (n samples, the flow as the terms of
    the four earlier stages -- a coarse flow of a few pixels plus sub-pixel refinements, i.e. smooth, as a trained net's)."""
    from drba_amd import ops
    gen = torch.Generator().manual_seed(0)
    # the items of a group share their frames as the pipeline's do: step j = (I[j+1], I[j]) and (I[j+1], I[j+2])
    fr = [(torch.rand(1, 3, H, W, generator=gen).to(dev), torch.randn(1, 16, H, W, generator=gen).to(dev)) for _ in range(n // 2 + 2)]
    for im, _ in fr:
        ops.rgbx(im)  # the [H,W,4] copy a pipeline frame carries (ops.to_inp writes it with the frame)
    items = []
    for j in range(n // 2):
        (a, fa), (b, fb), (c, fc) = fr[j], fr[j + 1], fr[j + 2]
        items.append((b, a, torch.rand(1, 1, H, W, generator=gen).to(dev), fb, fa))
        items.append((b, c, torch.rand(1, 1, H, W, generator=gen).to(dev), fb, fc))
    items = items[:n] if len(items) >= n else items + items[:n - len(items)]

    def head(st, amp):  # smooth flow channels (low-resolution noise, bicubic): a coarse flow of a few pixels plus refinements
        hh, ww = H // st, W // st
        t = torch.randn(n, 13, hh, ww, generator=gen)
        lo = torch.randn(n, 4, max(hh // 8, 2), max(ww // 8, 2), generator=gen) * amp
        t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
        return t.to(dev)
    terms = [(head(16, 1.0), 16.0), (head(8, 0.3), 8.0), (head(4, 0.3), 4.0)]
    tprev = head(2, 0.3)
    conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=gen) * 0.05, torch.zeros(16), 2, True, None, device=dev)
    run = lambda: ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ops.trace_begin()
    for _ in range(reps):
        run()
    recs = [r for r in ops.trace_end() if "stage_conv" in r["name"]]
    us = sum(r["ms"] for r in recs) / max(len(recs), 1) * 1e3
    ach = g["algorithmic_per_launch"] / us / 1e3  # GB/s
    return {"launch": g["launch"], "avg_us": round(us, 2), "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / roof["peak"], 4),
            "what": f"{reps} launches of this geometry alone (items sharing six frames as in the loop, smooth synthetic flows), the kernel's own dispatch times"}


def csrc_sha():
    """Identity of the kernel sources a profile table was taken on: sha256 over drba_amd/csrc/*.{hip,hpp} and the header (the GPU
    box has no .git).  tools/rocprof_frac.py records it; a table of other sources is not evidence for this build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "drba_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "drba_amd", "csrc", "*.hpp"))
                    + [os.path.join(ROOT, "include", "drba_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _rocprof_table():
    path = os.path.join(ROOT, "profiles", "rocprof_frac.json")  # {kernel symbol: {"avg_us": mean duration under rocprofv3}}
    return json.load(open(path)) if os.path.exists(path) else {}


def _traffic_table():
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # HBM bytes per launch from rocprofv3 --pmc passes
    return json.load(open(tpath)) if os.path.exists(tpath) else {}


# ------------------------------------------------------------------------------------------------- GPU legs
# DRBA_BENCH_BACKEND=gloo is a REHEARSAL switch, not a mode of the benchmark: the one-GPU build box cannot run RCCL with
# two ranks, so `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with it lets both ranks share cuda:0 and
# carries the collectives over gloo on host tensors -- every barrier, gather round and reduction of the N > 1 legs
# executes (tests/test_gpu_cli.py); its rates mean nothing.
BACKEND = os.environ.get("DRBA_BENCH_BACKEND", "nccl")


def _local_device():
    local = int(os.environ.get("LOCAL_RANK", 0))
    if BACKEND != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    return local


def _coll_device(dev):
    """Where the collectives' tensors live: the GPU for RCCL, the host for the gloo rehearsal."""
    return dev if BACKEND == "nccl" else torch.device("cpu")


def _fence(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def _quiet_gc():
    """Before a timed region: collect now and move everything alive to the permanent generation, so that a full
    (generation 2) collection of the interpreter -- 40-80 ms with torch loaded, triggered once the trace bookkeeping has
    allocated enough containers (tools/exp/trace_stall.py) -- does not land inside it."""
    gc.collect()
    gc.freeze()


def step_loop(model, frames, n_steps_total, args, world, trace=True, pcie=False, settle=True, check_path=True):
    """The N = 1 loop of the module docstring over `frames` (sequence of uint8 HWC frames: device tensors, or -- with
    pcie=True -- pinned host tensors with the outputs copied back to pinned host buffers).
    -> (seconds for args.steps steps, host-side enqueue seconds, trace records, instrumented steps)."""
    from drba_amd import ops
    from drba_amd.models.utils import tools
    H, W = frames[0].shape[:2]
    size = tools.get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), model.scale, div=model.pad_size)
    src_size, dst_size = size["src_size"], size["dst_size"]
    nf = len(frames)
    dev = model.device
    host_out = [torch.empty((H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(4)] if pcie else None

    def to_inp(k):
        f = frames[k % nf]
        if pcie:
            f = f.to(dev, non_blocking=True)
        return ops.to_inp(f, dst_size)

    n_out = [0]

    def to_out(x):
        y = ops.to_out(x, src_size)
        if pcie:
            host_out[n_out[0] % 4].copy_(y, non_blocking=True)
            n_out[0] += 1
        return y

    lookahead = not args.no_lookahead
    loop = AnnouncedLoop(model, to_inp, lambda k: TS, lookahead=lookahead)

    # The host enqueues a step in a third of the time the GPU needs for it and nothing in this loop synchronises (the frames stay on
    # the device): unthrottled it runs hundreds of steps ahead, every one holding its tensors -- at 4K, 6 GB per group of steps --
    # until the caching allocator has the whole HBM in flight and falls back to freeing / synchronising on every allocation: a
    # third of the 4K runs of round 6 ended at 6.4 ms per step instead of 3.5 (per-step log: the host time per group climbs
    # from 1.2 to 25 ms mid-run).  A real driver is throttled by the copy of every written frame to the host; here the loop waits
    # for the step kMaxStepsInFlight steps back, which leaves the GPU four groups of work queued.  The waits are not host work:
    # `throttle_s` is subtracted from the host time of the region.
    inflight, throttle = collections.deque(), [0.0]
    max_inflight = int(os.environ.get("DRBA_BENCH_INFLIGHT", kMaxStepsInFlight))

    def step():
        out, _ = loop.step()
        res = [to_out(x) for x in out]
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > max_inflight:
            t_w = time.perf_counter()
            inflight.popleft().synchronize()
            throttle[0] += time.perf_counter() - t_w
        return res

    for _ in range(args.warmup):
        step()
    # W is the minimum.  With three streams (per-stream allocator pools, the lookahead state two frames deep) and on a box
    # whose first GPU process this is, three steps do not reach the steady state: the first ~30 steps enqueue at 3.5 ms
    # each instead of 1.0 ms (allocator growth, first touches) and a 20-step timed region started there reads 508 instead
    # of 670 frames/s (profiles/README.md).  So the untimed warm-up goes on for kSettleSeconds of steps.
    t_settle, settled = time.perf_counter() + (kSettleSeconds if settle else 0.0), 0
    while time.perf_counter() < t_settle:
        step()
        settled += 1
        if settled % 8 == 0:
            torch.cuda.synchronize()
    while getattr(model, "_group_out", ()):  # the timed region starts on a call that computes a group of steps, so that K calls
        step()                                # hold ceil(K / GROUP) group computations whatever the phase of the settling loop was
        settled += 1
    torch.cuda.synchronize()
    LAST_SETTLE["steps"] = settled
    _quiet_gc()
    _fence(world)
    stats0 = dict(getattr(model, "stats", {}))
    throttle[0] = 0.0
    t0 = time.perf_counter()
    per_step = []
    blocks = 0
    while True:  # EXACTLY K steps per block; blocks are repeated (whole blocks, one fence at the end) until the region is long enough to quote
        for k in range(args.steps):
            step()
            per_step.append(time.perf_counter())
        blocks += 1
        if world > 1 or not settle or not args.steps or time.perf_counter() - t0 >= kMinTimedSeconds or blocks >= 64:  # (N > 1: every rank times one block)
            break
    t_host = time.perf_counter() - t0 - throttle[0]  # all launches enqueued: the host side of a step (the GPU may still be working)
    _fence(world)
    dt = time.perf_counter() - t0
    LAST_BLOCKS["n"] = blocks
    LAST_PATH.clear()
    LAST_PATH.update({k: v - stats0.get(k, 0) for k, v in getattr(model, "stats", {}).items()})
    grp = int(getattr(model, "GROUP", 1))
    if check_path and settle and lookahead and grp > 1 and args.steps and hasattr(model, "stats"):
        # the timed calls must have taken the grouped, side-stream-staged path the number is quoted for: a silent fall-back to
        # the one-step path (frames cloned on the way, a stale lookahead) would read as a kernel regression
        want = -(-(args.steps * blocks) // grp)
        if LAST_PATH["groups_formed"] != want or LAST_PATH["staged_groups"] != want or LAST_PATH["single_steps"]:
            raise RuntimeError(f"timed region did not run the grouped path: {LAST_PATH} (expected {want} staged groups)")
    recs, traced = None, 0
    if trace:  # the roofline block: same loop, same state, every launch traced
        ops.trace_begin()
        for _ in range(kTraceWarm):
            step()
        first = ops._trace_pos()
        for _ in range(kTraceKeep):
            step()
        last = ops._trace_pos()
        torch.cuda.synchronize()
        recs, traced = ops.trace_end()[first:last], kTraceKeep
        step()  # the first untraced launches after traced ones pay a one-time mode switch: not in anybody's timed region
        torch.cuda.synchronize()
    if os.environ.get("DRBA_BENCH_STEPLOG"):  # host-side enqueue time of every step (diagnostic)
        log("host ms per step: " + " ".join(f"{(b - a) * 1e3:.2f}" for a, b in zip([t0] + per_step, per_step)))
    return dt, t_host, recs, traced, dst_size


def clip_leg(model, clip, dst_fps, times, scdet, args, label):
    """One extra config through the real driver loop: drba_amd.infer.interpolate_stream over a clip resident in HBM.
    The K loop iterations after W warm-up iterations are timed (on_step marks them); the iterations after them are the traced roofline block (see kTraceWarm)."""
    from drba_amd import infer as drv
    from drba_amd import ops
    to_inp, to_out = _dev_hooks()
    if scdet and getattr(clip, "cut_at", None) is not None:
        # Untimed rehearsal: a 66-frame clip of the same size with four cuts, 13 / 14 / 15 / 16 frames apart.  The planted cut of the
        # timed clip lies INSIDE the timed region, and what a cut runs are launch shapes the warm-up iterations in front of it
        # never see: inference_ts on one pair (batch 1 and 2), a cold calc_flow, and the SHORTER groups of steps on either side of
        # it (the driver announces frames only up to the cut: 2 or 3 steps = 4 or 6 samples per launch, depending on where the
        # cut falls in the group phase -- hence the four spacings).  Their first launches run the conv autotuner (a device
        # synchronisation per candidate) and pack this instance's weights: 0.2 s inside a 0.25 s region (profiled: 53 _tune calls),
        # i.e. 6 ms of host per step -- what a long clip pays once per process.  Rounds 3-5 had the shapes tuned by accident: the
        # cyclic clip's wrap-around every 8 frames read as a cut during the warm-up iterations.
        H_, W_px = clip.shape[0], clip.shape[1]
        drv.interpolate_stream(_Counting(model), _DevIO(DeviceClip(66, H_, W_px, 977, model.device, cut_at=[13, 27, 42, 58], pingpong=True),
                                                        SRC_FPS),
                               dst_fps, times=times, enable_scdet=True, to_inp=to_inp, to_out=to_out)
        torch.cuda.synchronize()
    cm = _Counting(model)
    io = _DevIO(clip, SRC_FPS)
    W_, K = args.warmup, args.steps
    st = {"t0": None, "t1": None, "g0": 0, "g1": 0, "w0": 0, "w1": 0, "first": None, "last": None, "host": 0.0, "stats0": {}, "path": None}
    ops.trace_begin()
    ops.trace_pause()

    def on_step(idx):  # idx = loop iterations completed (0 after the head); called before iteration j = idx - W_ runs
        j = idx - W_
        if j == 0:
            _quiet_gc()
            torch.cuda.synchronize()
            st["stats0"] = dict(getattr(model, "stats", None) or {})
            if os.environ.get("DRBA_BENCH_PROFILE"):  # diagnostic: where the host time of the timed iterations goes
                import cProfile
                st["prof"] = cProfile.Profile()
                st["prof"].enable()
            st["t0"], st["g0"], st["w0"] = time.perf_counter(), cm.generated, io.written
        if j == K:  # the timed iterations are done; the clip's remaining iterations are the roofline block (all traced)
            st["host"] = time.perf_counter() - st["t0"]  # every launch of the K iterations enqueued: the host side of the region
            if st.get("prof") is not None:
                import io as _io
                import pstats
                st["prof"].disable()
                buf = _io.StringIO()
                pstats.Stats(st["prof"], stream=buf).sort_stats("cumulative").print_stats(30)
                log("host profile of the timed iterations of: " + label[:60] + "\n" + buf.getvalue())
            torch.cuda.synchronize()
            st["t1"], st["g1"], st["w1"] = time.perf_counter(), cm.generated, io.written
            st["path"] = {k: v - st["stats0"].get(k, 0) for k, v in (getattr(model, "stats", None) or {}).items()} or None
            ops.trace_resume()
        if j == K + kTraceWarm:
            st["first"] = ops._trace_pos()
        if j == K + kTraceWarm + kTraceKeep:
            st["last"] = ops._trace_pos()
            ops.trace_pause()

    drv.interpolate_stream(cm, io, dst_fps, times=times, enable_scdet=scdet, to_inp=to_inp, to_out=to_out, on_step=on_step)
    torch.cuda.synchronize()
    recs = ops.trace_end()
    recs = recs[st["first"]:st["last"]] if st["last"] is not None else []
    dt = st["t1"] - st["t0"]
    gen, wr = st["g1"] - st["g0"], st["w1"] - st["w0"]
    # `path` (RIFE.stats over the timed iterations): how many of them ran as groups of steps, collected a step computed ahead, or
    # fell to the one-step path (around the planted cut: the iterations whose lookahead window holds the cut, and the two calls
    # inference_ts makes beside it); `host_ms_per_step`: launches enqueued, GPU possibly still working
    return {"workload": label, "value": round(gen / dt, 3), "unit": "frames/s", "steps": K, "warmup": W_,
            "ms_per_step": round(dt / K * 1e3, 3), "host_ms_per_step": round(st["host"] / K * 1e3, 3), "frames_generated": gen,
            "frames_written": wr, "path": st["path"], "group": int(getattr(model, "GROUP", 1)),
            "roofline": roofline_from_trace(recs, kTraceKeep, _traffic_table())}


def three_term_leg(args, dev):
    """The headline loop with every MFMA operand at 24 bits (kernel families 0-3: three bf16 terms per fp32 operand, six
    products per multiply) -- the arithmetic of rounds 1-3, driver-run beside the two-term default."""
    from drba_amd import ops
    from drba_amd.models.rife import RIFE
    fams = set(ops.CONV_FAMILIES)
    ops.set_precision({0, 1, 2, 3})
    try:
        (H, W), scale, desc = CONFIGS["1080p"]
        model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
        clip = DeviceClip(12, H, W, 1234, dev)
        dt, _, _, _, _ = step_loop(model, [clip[k] for k in range(len(clip))], args.warmup + args.steps, args, 1, trace=False)
        nb = LAST_BLOCKS["n"]
        return {"workload": desc + "; kernel families 0-3 only", "value": round(len(TS) * args.steps * nb / dt, 3), "unit": "frames/s",
                "steps": args.steps, "timed_blocks": nb, "ms_per_step": round(dt / (args.steps * nb) * 1e3, 3),
                "arithmetic": _arithmetic_note(), "path": dict(LAST_PATH)}
    finally:
        ops.set_precision(fams)


def extra_configs(args, dev):
    """BASELINE.json configs[2], [3], [4] at N = 1 (bounded: K steps each)."""
    from drba_amd.models.gmfss_union import GMFSS_UNION
    from drba_amd.models.rife import RIFE
    # W is the minimum (see step_loop); the clip legs time at least kClipSteps iterations, so that the region holds >= 4 whole groups
    # of steps on each side of the planted cut and the cut's recovery (cold calc_flow, one-step calls until a group is announced
    # again) is a bounded share of it, as it is of a real clip
    args = argparse.Namespace(**{**vars(args), "warmup": max(args.warmup, kClipWarmup), "steps": max(args.steps, kClipSteps)})
    n = args.warmup + args.steps + 3 + kTraceWarm + kTraceKeep + 1  # + the traced iterations of the roofline block
    out = {}
    cut = args.warmup + args.steps // 2 + 2
    m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=1.0, device=dev)
    log("extra: config 3")
    out["config3_rife_fps60_scdet_1080p"] = clip_leg(
        m, DeviceClip(n, 1080, 1920, 1234, dev, cut_at=cut, pingpong=True), 60.0, -1, True, args,
        f"rife -fps 60 (24 -> 60: ts alternate [0.6,1.0,1.4] / [0.8,1.2]), 1080p (net 1088x1920), scale 1.0, scdet on "
        f"(threshold 0.3), one planted cut at frame {cut}; driver loop incl. to_inp/to_out/check_scene on device")
    del m
    m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=0.5, device=dev)
    log("extra: config 5 (one GPU)")
    out["config5_rife_fps60_4k_scale0.5_one_gpu"] = clip_leg(
        m, DeviceClip(n, 2160, 3840, 1234, dev, cut_at=cut, pingpong=True), 60.0, -1, True, args,
        f"rife -fps 60, 4K (net 2176x3840), scale 0.5, scdet on, one planted cut at frame {cut}: the per-GPU work of the "
        "frame-sharded config (N = 1)")
    del m
    torch.cuda.empty_cache()
    g = GMFSS_UNION(weights=synth.gmfss_union_state_dicts(seed=0), scale=1.0, device=dev)
    log("extra: config 4 (model built)")
    out["config4_gmfss_union_fps60_1080p"] = clip_leg(
        g, DeviceClip(n, 1080, 1920, 4321, dev), 60.0, -1, False, args,
        "gmfss_union -fps 60 (24 -> 60), 1080p (net 1152x1920), scale 1.0 (GMFlow + softsplat + GridNet path)")
    del g
    torch.cuda.empty_cache()
    from drba_amd import ops
    if 4 in ops.CONV_FAMILIES and not args.no_lookahead:
        log("extra: headline loop with 24-bit MFMA operands")
        out["headline_three_term_1080p"] = three_term_leg(args, dev)
        torch.cuda.empty_cache()
    return out


def sharded_leg(model, clip, dst_fps, times, scdet, rank, world, dev):
    """One clip sharded over the ranks (drba_amd.parallel), frames streamed to rank 0 -> (seconds max over ranks,
    generated frames of all ranks, frames on the writer)."""
    from drba_amd import parallel
    cm = _Counting(model)
    to_inp, to_out = _dev_hooks()
    counts = parallel.emission_counts(len(clip), SRC_FPS, dst_fps, times, world)
    _quiet_gc()
    _fence(world)
    stats0 = dict(getattr(model, "stats", None) or {})
    t0 = time.perf_counter()
    sg = parallel.StreamedGather(rank, world, counts, chunk=4, device=_coll_device(dev), frame_shape=clip.shape)
    parallel.interpolate_shard(cm, clip, SRC_FPS, dst_fps, rank, world, times=times, enable_scdet=scdet,
                               to_inp=to_inp, to_out=to_out, sink=sg.push)
    allf = sg.finish()
    _fence(world)
    dt = time.perf_counter() - t0
    LAST_SHARD["rank_dt"] = dt
    # which path this rank's shard took: a shard starts cold (halo reuse, the first group computed in place before a staged one
    # exists), so its ramp-in -- steps outside staged groups -- separates the GROUP ramp from RCCL cost in a scaling record
    st = {k: v - stats0.get(k, 0) for k, v in (getattr(model, "stats", None) or {}).items()}
    grp = int(getattr(model, "GROUP", 1))
    LAST_SHARD["path"] = dict(st, group=grp, ramp_in_steps=st.get("single_steps", 0) + grp * (st.get("groups_formed", 0) - st.get("staged_groups", 0))) if st else None
    if world == 1:  # (--selftest-sharded: the same code on one GPU, no process group)
        return dt, cm.generated, (len(allf) if allf is not None else 0)
    import torch.distributed as dist
    t = torch.tensor([dt, float(cm.generated)], dtype=torch.float64, device=_coll_device(dev))
    mx = t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(mx[0].item()), int(round(float(t[1].item()))), (len(allf) if allf is not None else 0)


def sharded_warmup(model, H, W, dst_fps, times, scdet, rank, world, dev, cut=False):
    """Untimed: the sharded run on a short clip of the same frame size, so that everything a shard touches OUTSIDE the
    steady-state loop -- the head / tail `inference_ts` calls (batch 1), the halo's `warm_reuse`, a scene-cut step, the
    gather's buffers -- has been autotuned / allocated on every rank before the timed run."""
    n = 4 * world + 2
    clip = DeviceClip(n, H, W, 4321, dev, cut_at=(n // 2 if cut else None), pingpong=bool(cut))
    sharded_leg(model, clip, dst_fps, times, scdet, rank, world, dev)


def gpu_leg(args, rank, world):
    from drba_amd.models.rife import RIFE

    local = _local_device()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    (H, W), scale, desc = CONFIGS[args.config]
    model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
    n_total = args.warmup + args.steps
    r = {"desc": desc, "dev": dev, "model": model}
    clip = DeviceClip(min(n_total + 2, 12), H, W, 1234 + rank, dev)
    frames = [clip[k] for k in range(len(clip))]
    dt, t_host, recs, traced, dst = step_loop(model, frames, n_total, args, world, trace=not args.no_roofline)
    r["blocks"] = LAST_BLOCKS["n"]  # K-step blocks timed back to back (one region): every per-step figure divides by K * blocks
    r["dst_size"] = dst
    r["settle_steps"] = LAST_SETTLE["steps"]
    r["path"] = dict(LAST_PATH, group=int(getattr(model, "GROUP", 1)), what="RIFE.stats over the K timed calls of this rank")
    serial = None
    if recs and world == 1 and not args.no_lookahead:
        # the same steps on ONE stream (no lookahead / prefetch), every launch traced: each kernel's own duration, the
        # ranking of the symbols and what rocprofv3's table of this command should agree with
        a1 = argparse.Namespace(**{**vars(args), "steps": 0, "warmup": 2, "no_lookahead": True})
        _, _, recs1, traced1, _ = step_loop(model, frames, 2, a1, world, trace=True, settle=False)
        serial = _symbol_totals(recs1, traced1) if recs1 else None
        if serial and os.environ.get("DRBA_BENCH_SERIAL_TABLE"):  # the whole single-stream table (diagnostic)
            inst = _symbol_totals(recs, traced)
            for k, (ms, n, us, _) in sorted(serial.items(), key=lambda kv: -kv[1][0]):
                log(f"serial {ms:7.4f} ms/step {n:5.1f} x {us:7.1f} us | in-step {inst.get(k, (0, 0, 0))[2]:7.1f} us | {k[:110]}")
    r["roofline"] = roofline_from_trace(recs, traced, _traffic_table(), serial, workload=args.config) if recs else None
    if world == 1:
        r.update({"dt": dt, "host_dt": t_host, "frames": len(TS) * args.steps * r["blocks"]})
        if r["roofline"] and (r["roofline"].get("bound") == "mfma" or "stage_conv" in r["roofline"].get("kernel", "")):
            torch.cuda.synchronize()
            r["roofline"]["standalone"] = standalone_of(r["roofline"], dev)
            r["roofline"]["frac_standalone"] = (r["roofline"]["standalone"] or {}).get("frac")
        return r
    # ---- N > 1: the headline is ONE clip sharded over the ranks, K loop iterations per rank (weak scaling)
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device=_coll_device(dev))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    r["replica_loop"] = {"value": round(len(TS) * args.steps * world / float(t.item()), 3), "unit": "frames/s",
                         "ms_per_step": round(float(t.item()) / args.steps * 1e3, 3), "scaling": "weak",
                         "what": "every rank runs the N = 1 loop on its own clip: no halo, no collective"}
    big = DeviceClip(world * args.steps + 2, H, W, 1234, dev)
    sharded_warmup(model, H, W, SRC_FPS * 2, 2, False, rank, world, dev)
    sdt, gen, got = sharded_leg(model, big, SRC_FPS * 2, 2, False, rank, world, dev)
    r.update({"dt": sdt, "host_dt": None, "frames": gen, "writer_frames": got, "rank_dt": LAST_SHARD["rank_dt"], "rank_path": LAST_SHARD["path"]})
    if not args.no_extra:
        m5 = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=0.5, device=dev)
        n5 = 8 * args.steps + 2  # fixed clip whatever N is: strong scaling
        c5 = DeviceClip(n5, 2160, 3840, 1234, dev, cut_at=n5 // 2, pingpong=True)
        warm = [c5[k] for k in range(min(12, n5))]
        step_loop(m5, warm, 0, argparse.Namespace(**{**vars(args), "steps": 2, "warmup": 2}), world, trace=False, check_path=False)  # autotune / allocator warm-up
        sharded_warmup(m5, 2160, 3840, 60.0, -1, True, rank, world, dev, cut=True)
        sdt5, gen5, got5 = sharded_leg(m5, c5, 60.0, -1, True, rank, world, dev)
        r["config5_sharded"] = {"value": round(gen5 / sdt5, 3), "unit": "frames/s", "scaling": "strong", "seconds": round(sdt5, 4),
                                "frames_generated": gen5, "writer_frames": got5, "clip_source_frames": n5,
                                "workload": "rife -fps 60, 4K (net 2176x3840), scale 0.5, scdet on, one planted cut: ONE clip of "
                                            f"{n5} source frames sharded over {world} ranks, frames gathered on rank 0"}
    return r


def pcie_leg(args, model):
    """The N = 1 loop with the clip in pinned host memory and every output copied back to pinned host buffers
    (async H2D / D2H on the compute stream): what the CLI pays on top of the resident-in-HBM metric."""
    (H, W), _, _ = CONFIGS[args.config]
    frames = [torch.from_numpy(f).pin_memory() for f in make_frames_u8(min(args.warmup + args.steps + 2, 12), H, W, seed=1234)]
    dt, _, _, _, _ = step_loop(model, frames, args.warmup + args.steps, args, 1, trace=False, pcie=True)
    nb = LAST_BLOCKS["n"]
    return {"value": round(len(TS) * args.steps * nb / dt, 3), "unit": "frames/s", "ms_per_step": round(dt / (args.steps * nb) * 1e3, 3),
            "what": "same loop, uint8 frames read from pinned host memory and written back to pinned host memory (PCIe inclusive)"}


# ------------------------------------------------------------------------------------------------- CPU leg
def cpu_leg(args, model):
    """The CPU baseline: the fp32 oracle (port of the reference, pinned to it by tests/golden) on the same workload, and
    the parity figure of the metric: the HIP path run on the SAME uint8 frames, max-abs of the synthesised frames.
    Bounded sample: one untimed warm-up (a calc_flow to build `reuse` + one IFNet pass so oneDNN primitives exist), then
    `--cpu-steps` timed warm steps including to_inp/to_out on all cores, and one timed step on one thread."""
    import oracle  # the checker, timed here as the reported CPU baseline (never the product path)
    from drba_amd import ops
    (H, W), scale, _ = CONFIGS[args.config]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = avail if args.cpu_threads <= 0 else max(1, min(avail, args.cpu_threads))
    ora = oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), scale)
    from drba_amd.models.utils.tools import get_valid_net_inp_size
    dst = get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), scale, div=64)["dst_size"]
    n_ahead = 2 * int(getattr(model, "GROUP", 1)) - 1  # frames beyond the compared steps, so that the HIP side forms the groups it is timed with
    fr = make_frames_u8(3 + args.cpu_steps + max(n_ahead, 0), H, W, seed=1234)

    def to_inp(f):
        return oracle.ops.resize(torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0, dst)

    def to_out(x):
        return (oracle.ops.resize(x, (H, W))[0].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)

    state = {}

    def run(threads, steps, keep):
        torch.set_num_threads(threads)
        I = [to_inp(f) for f in fr[:steps + 3]]
        with torch.no_grad():
            if not state:  # untimed, once: what the previous step would have left behind + oneDNN primitive creation
                flow12, flow21, f1, f2 = ora.calc_flow(I[1], I[2])
                state["reuse"] = (flow21, flow12, f2, f1)
                oracle.ifnet.ifnet(ora.sd, torch.cat((I[1], I[2]), 1), 0.5, ora.scale_list, f0=f1, f1=f2)
            reuse = state["reuse"]
            t0 = time.perf_counter()
            n = 0
            for k in range(steps):
                out, reuse = ora.inference_ts_drba(I[k + 1], I[k + 2], I[k + 3], TS, reuse, True)
                u8 = [to_out(x) for x in out]
                if keep is not None:
                    keep.append((out, u8))
                n += len(out)
            return n, time.perf_counter() - t0

    kept = []
    n, dt = run(cores, args.cpu_steps, kept)
    log(f"cpu leg: {cores} threads {n / dt:.3f} frames/s ({dt:.1f} s)")
    n1, dt1 = run(1, 1, None)
    log(f"cpu leg: 1 thread {n1 / dt1:.4f} frames/s ({dt1:.1f} s)")
    base = {"value": round(n / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} warm inference_ts_drba step(s) = {n} frames at {dst[0]}x{dst[1]} incl. to_inp/to_out, after an "
                      f"untimed warm-up; torch {torch.__version__} CPU fp32, {cores} threads of {avail} available",
            "single_thread": {"value": round(n1 / dt1, 4), "unit": "frames/s", "cores": 1, "sample": f"1 warm step = {n1} frames"}}
    # ---- parity on the same frames: HIP path, same uint8 inputs, same step sequence, driven by THE loop of the timed region
    # (AnnouncedLoop: frames read ahead, encoders / coarse flows prefetched, the steps computed in groups of RIFE.GROUP)
    dev = model.device
    g = [ops.to_inp(torch.from_numpy(f).to(dev), dst) for f in fr]
    stats0 = dict(model.stats)
    loop = AnnouncedLoop(model, lambda k: g[k] if k < len(g) else None, lambda k: TS, reuse=model.warm_reuse(g[1], g[2]), first=1)
    worst, worst_lsb, n_cmp = 0.0, 0, 0
    for k, (want, want_u8) in enumerate(kept):
        out, _ = loop.step()
        for a, b, bu in zip(out, want, want_u8):
            worst = max(worst, float((a.cpu() - b).abs().max()))
            au = ops.to_out(a, (H, W)).cpu().numpy()
            worst_lsb = max(worst_lsb, int(np.abs(au.astype(np.int32) - bu.astype(np.int32)).max()))
            n_cmp += 1
    torch.cuda.synchronize()
    path = {k: v - stats0[k] for k, v in model.stats.items()}
    parity = {"value": worst, "frames": n_cmp, "u8_max_lsb": worst_lsb, "tolerance": 1e-3, "path": path,
              "what": "max |HIP - CPU oracle| over the synthesised fp32 frames of the cpu_baseline steps (same uint8 inputs, to_inp "
                      "on each side), the HIP side driven by the timed region's loop (frames announced ahead: `path` counts the "
                      "groups it formed); u8_max_lsb = largest difference of the written uint8 frames"}
    return base, parity


def describe_job(rank, world, r):
    """What torch.distributed saw: backend, world size, the device of every rank, the RCCL version, every rank's time for
    the timed sharded run -- so that a scaling record can be checked for "did RCCL run with N ranks on N GPUs"."""
    dev = r["dev"]
    prop = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", 0)), "device": f"cuda:{dev.index}", "name": prop.name,
          "arch": getattr(prop, "gcnArchName", None), "pci_bus_id": getattr(prop, "pci_bus_id", None),
          "uuid": str(getattr(prop, "uuid", "")) or None, "pid": os.getpid(),
          "rank_seconds": None if r.get("rank_dt") is None else round(r["rank_dt"], 5),
          "rank_path": r.get("rank_path")}  # this rank's groups formed / staged and ramp-in steps in the timed sharded run
    info = {"backend": "none (single process)", "world_size": 1, "devices": [me], "rccl_version": None,
            "visible_gpus": torch.cuda.device_count()}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001  (a torch build without the binding: reported as null)
        pass
    if world > 1:
        import torch.distributed as dist
        rows = [None] * world
        dist.all_gather_object(rows, me)
        info.update({"backend": dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else " (rehearsal, not a benchmark)"),
                     "world_size": dist.get_world_size(), "devices": rows,
                     "distinct_devices": len({(d["device"], d["uuid"], d["pci_bus_id"]) for d in rows})})
    return info


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed without a launcher (`python bench.py --gpus 8`): start the ranks ourselves, one process per GPU, exactly
        # as the driver's command line does; rank 0's JSON line is the only thing on stdout either way
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("self-launch: " + " ".join(cmd))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        os.execv(sys.executable, cmd)
    if args.gpus != world:
        print(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(_local_device())
        dist.init_process_group(backend=BACKEND)  # "nccl" = RCCL over xGMI
    if args.selftest_sharded:
        from drba_amd.models.rife import RIFE
        dev = torch.device("cuda", 0)
        (H, W), scale, _ = CONFIGS[args.config]
        model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
        sharded_warmup(model, H, W, SRC_FPS * 2, 2, False, 0, 1, dev)
        dt, gen, got = sharded_leg(model, DeviceClip(args.steps + 2, H, W, 1234, dev), SRC_FPS * 2, 2, False, 0, 1, dev)
        out = {"selftest": "sharded legs at world 1", "headline_clip": {"frames_generated": gen, "writer_frames": got, "frames_per_s": round(gen / dt, 2),
                                                                        "rank_path": LAST_SHARD["path"]}}
        m5 = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=0.5, device=dev)
        n5 = args.steps + 2
        sharded_warmup(m5, 2160, 3840, 60.0, -1, True, 0, 1, dev, cut=True)
        dt5, gen5, got5 = sharded_leg(m5, DeviceClip(n5, 2160, 3840, 1234, dev, cut_at=n5 // 2, pingpong=True), 60.0, -1, True, 0, 1, dev)
        out["config5_clip"] = {"frames_generated": gen5, "writer_frames": got5, "frames_per_s": round(gen5 / dt5, 2)}
        print(json.dumps(out))
        return
    r = gpu_leg(args, rank, world)
    log(f"gpu leg done: {r['frames'] / r['dt']:.1f} frames/s")
    dist_info = describe_job(rank, world, r)
    cpu = parity = extra = pcie = None
    if rank == 0 and world == 1:
        if not args.no_extra:
            pcie = pcie_leg(args, r["model"])
            log(f"pcie leg done: {pcie['value']} frames/s")
        if not args.no_cpu_baseline:
            cpu, parity = cpu_leg(args, r["model"])
            log(f"cpu leg done: {cpu['value']} frames/s on {cpu['cores']} threads, max_abs {parity['value']:.2e}")
        if not args.no_extra:
            r["model"] = None
            torch.cuda.empty_cache()
            extra = extra_configs(args, r["dev"])
    if rank == 0:
        wl = r["desc"] + "; warm inference_ts_drba ts=[0.75,1.25] + to_inp/to_out on device"
        if world > 1:
            wl += f"; ONE clip of {world * args.steps + 2} source frames sharded over {world} ranks (interpolate_shard), frames gathered on rank 0"
        line = {
            "metric": "interpolated frames/sec @1080p RIFE x2" if args.config == "1080p" else f"interpolated frames/sec @{args.config} RIFE x2",
            "value": round(r["frames"] / r["dt"], 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": r["settle_steps"], "timed_blocks": r.get("blocks", 1),
            "timed_region_s": round(r["dt"], 4), "ms_per_step": round(r["dt"] / (args.steps * r.get("blocks", 1)) * 1e3, 3),
            "host_ms_per_step": None if r["host_dt"] is None else round(r["host_dt"] / (args.steps * r.get("blocks", 1)) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": _dtype_note(), "data": "synthetic",
            "config": {"workload": wl, "net_size": list(r["dst_size"]), "frames_per_step": len(TS),
                       "weights": "seeded random IFNet 4.26-heavy", "parallelism": f"frame-sharded dp{world}"},
            "arithmetic": _arithmetic_note(),
            "max_abs_vs_oracle": parity, "roofline": r["roofline"], "cpu_baseline": cpu,
            "path": r.get("path"),
        }
        if args.config in FRAME_WORK and world == 1:  # SURVEY 8(d)'s whole-frame pair: the frame's algorithmic work x frames/s against the two roofs
            gf, gb = FRAME_WORK[args.config]
            line["frame_mfma_frac"] = round(gf * 1e9 * line["value"] / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
            line["frame_hbm_frac"] = round(gb * 1e9 * line["value"] / (HBM_PEAK_GBS * 1e9), 4)
            line["frame_work"] = {"gflop_per_frame": gf, "conv_boundary_gb_per_frame": gb, "mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                                  "hbm_peak_gbs": HBM_PEAK_GBS, "source": "SURVEY.md 8(d) / BASELINE.md 3 (fp32-MFMA peak: the reference's arithmetic)"}
        line["dist"] = dist_info
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        for k in ("replica_loop", "config5_sharded", "writer_frames"):
            if k in r:
                line[k] = r[k]
        if extra is not None:
            line["extra_configs"] = extra
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
