#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (+ SQ_WAVES, TA_TA_BUSY ... whatever the pass
collected) over tools/pmc_targets.py -> a table of matrix-core utilisation per roofline target.
    python tools/pmc_mfma.py COUNTERS.csv|COUNTERS.db gpurun_out/pmc_manifest.json [table.md]
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE): the share of the launch during which
a SIMD's matrix pipe holds an instruction (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per
32x32x16 / ~16 per 16x16x32 bf16 MFMA; GRBM_GUI_ACTIVE = clocks of the launch).  The split-bf16 kernels issue 6 bf16 MFMAs
per fp32 product block, so "fraction of the bf16/6 roofline" x (clock / 2.4 GHz) is what this utilisation tops out at."""
import json
import sys

from pmc_traffic import per_target  # noqa: E402  (same segmentation of the dispatch order by marker launches)

CU, SIMD = 256, 4


def main():
    path, man = sys.argv[1:3]
    manifest = json.load(open(man))
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
             "TA_TA_BUSY_sum", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
    cols = {}
    for c in names:
        try:
            v = per_target(path, manifest, c)
        except AssertionError:
            continue
        if any(x is not None for x in v):
            cols[c] = v
    have = list(cols)
    lines = ["| target | kernel | launch | " + " | ".join(have) + " | MFMA busy % | launch us at 2.4 GHz |", "|---|---|---|" + "---|" * (len(have) + 2)]
    for i, m in enumerate(manifest):
        vals = [cols[c][i] for c in have]
        mf, ga = (cols.get("SQ_VALU_MFMA_BUSY_CYCLES") or [None] * len(manifest))[i], (cols.get("GRBM_GUI_ACTIVE") or [None] * len(manifest))[i]
        util = "" if not mf or not ga else f"{100.0 * mf / (CU * SIMD * ga):.1f}"
        us = "" if not ga else f"{ga / 2400.0:.1f}"
        lines.append(f"| {m['name']} | `{m['symbol']}` | {m['label']} | " + " | ".join("-" if v is None else f"{v:.0f}" for v in vals) + f" | {util} | {us} |")
    print("\n".join(lines))
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
