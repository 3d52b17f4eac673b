#!/usr/bin/env python3
"""A/B runs of bench.py between kernel variants / schedules of THIS library (never another backend): the switches that used
to be bench.py flags.  Everything not listed here is passed on to bench.py unchanged.

    python tools/ab_bench.py [--group G] [--conv-families 0,1] [--no-stage-conv] [--stage-conv-fp32] [--no-head-fused] [--no-lazy-flow]
                             [--no-lookahead] [--prefetch-priority P] [--enc-main] [--side-stages S] -- [bench.py flags]

  --group G            RIFE.GROUP, consecutive steps per stacked IFNet pass (1: off; negative: groups without the batched coarse flows)
  --conv-families L    kernel families the conv autotuner may pick from (0 fp32 MFMA, 1 split-bf16, 2 LDS-DMA 32 ch, 3 K-split)
  --no-stage-conv      the scale-1 stage input and conv0[0] as two kernels (ops.STAGE_CONV_FUSED = False)
  --stage-conv-fp32    the fused stage kernel in its exact-fp32 form (stage_conv.hip; ops.STAGE_CONV_TWO_TERM = False)
  --no-stage-conv-s2   the scale-2 stage as two kernels (gather + conv0[0]; ops.STAGE_CONV_S2 = False)
  --no-img-x4          the gathers read the planar frames, to_inp writes no [H,W,4] copy (ops.IMG_X4 = False; the scale-2 fusion needs the copies)
  --no-head-fused      IFNet's encoder layer by layer (ops.HEAD_FUSED = False)
  --no-lazy-flow       IFNet's running flow as a full-resolution tensor updated after every stage (ops.LAZY_FLOW = False)
  --no-lookahead       no side / prefetch streams (the single-stream loop)
  --prefetch-priority  HIP priority of the encoder / prefetch stream (-1 = high)
  --enc-main           RIFE.ENC_ON_MAIN, the encoders in the main stream
  --side-stages S      RIFE.SIDE_STAGES, IFNet stages of the next group staged on the side stream
Measured outcomes of each are recorded in DESIGN.md ("tried and rejected", "groups of steps")."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser(add_help=True)
    p.add_argument("--group", type=int, default=None)
    p.add_argument("--conv-families", default=None)
    p.add_argument("--no-stage-conv", action="store_true")
    p.add_argument("--stage-conv-fp32", action="store_true")
    p.add_argument("--no-stage-conv-s2", action="store_true")
    p.add_argument("--no-img-x4", action="store_true")
    p.add_argument("--no-head-fused", action="store_true")
    p.add_argument("--no-lazy-flow", action="store_true")
    p.add_argument("--no-lookahead", action="store_true")
    p.add_argument("--prefetch-priority", type=int, default=None)
    p.add_argument("--enc-main", action="store_true")
    p.add_argument("--side-stages", type=int, default=None)
    p.add_argument("--chain-split", type=float, default=None, help="ops.CHAIN_SPLIT_BYTES in MB: conv chains with a wider tensor run as two half-batches")
    p.add_argument("--cu-mask", default=None,
                   help="role=first:count[,role=...] CUs PER XCD (of 32) for the side / prefetch / main streams, e.g. "
                        "side=28:4,prefetch=28:4,main=0:28 (main: the whole benchmark runs inside a masked stream)")
    a, rest = p.parse_known_args()
    import bench
    from drba_amd import ops
    from drba_amd.models import lookahead
    from drba_amd.models.rife import RIFE
    if a.group is not None:
        RIFE.GROUP = abs(int(a.group))
        RIFE.BATCH_COARSE = int(a.group) > 0
    if a.conv_families is not None:
        ops.CONV_FAMILIES = {int(x) for x in a.conv_families.split(",")}
    if a.no_stage_conv:
        ops.STAGE_CONV_FUSED = False
    if a.stage_conv_fp32:
        ops.STAGE_CONV_TWO_TERM = False
    if a.no_stage_conv_s2:
        ops.STAGE_CONV_S2 = False
    if a.no_img_x4:
        ops.IMG_X4 = False
    if a.no_head_fused:
        ops.HEAD_FUSED = False
    if a.no_lazy_flow:
        ops.LAZY_FLOW = False
    if a.prefetch_priority is not None:
        lookahead.PRIORITY["prefetch"] = int(a.prefetch_priority)
    if a.enc_main:
        RIFE.ENC_ON_MAIN = True
    if a.side_stages is not None:
        RIFE.SIDE_STAGES = int(a.side_stages)
    if a.chain_split is not None:
        ops.CHAIN_SPLIT_BYTES = int(a.chain_split * 1e6)
    if a.cu_mask:
        for part in a.cu_mask.split(","):
            role, rng = part.split("=")
            first, count = (int(v) for v in rng.split(":"))
            lookahead.CU_MASK[role] = (first, count)
    bench.AB["no_lookahead"] = bool(a.no_lookahead)
    sys.argv = [os.path.join(ROOT, "bench.py")] + [x for x in rest if x != "--"]
    if "main" in lookahead.CU_MASK:
        import torch
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        with torch.cuda.stream(lookahead.masked_stream(dev, *lookahead.CU_MASK["main"])):
            bench.main()
    else:
        bench.main()


if __name__ == "__main__":
    main()
