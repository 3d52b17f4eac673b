#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (+ SQ_WAVES, TA_TA_BUSY ... whatever the pass
collected) over tools/pmc_targets.py -> a table of matrix-core utilisation per roofline target.
    python tools/pmc_mfma.py COUNTERS.csv|COUNTERS.db gpurun_out/pmc_manifest.json [table.md]
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x clocks of the launch): the share of the launch during
which a SIMD's matrix pipe holds an instruction (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per
32x32x16 / ~16 per 16x16x32 bf16 MFMA).  rocprofv3 reports GRBM_GUI_ACTIVE SUMMED over the 8 XCDs (one GRBM each): the
calibration launch (512 MiB of traffic, ~100 us between its dispatch timestamps) reads 8 x 2.4e9 x 100e-6 of it, so
clocks of the launch = GRBM_GUI_ACTIVE / 8 (the first table of round 3 divided by the sum and read 8x too low).  The
"us" column is the dispatch's own End - Start timestamp in the same CSV, "MHz" the clock that follows from the two.
The split-bf16 kernels issue 6 bf16 MFMAs per fp32 product block, so "fraction of the bf16/6 roofline" x (2.4 GHz / clock)
is what this utilisation tops out at."""
import json
import sys

from pmc_traffic import per_target, short  # noqa: E402  (same segmentation of the dispatch order by marker launches)

CU, SIMD, XCD = 256, 4, 8


def durations(path, manifest):
    """Mean End - Start (us) of each target's launches, from the CSV's dispatch timestamps (None for a .db input)."""
    if not path.endswith(".csv"):
        return [None] * len(manifest)
    import csv
    seen, seq = set(), []
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        if d not in seen:
            seen.add(d)
            seq.append((d, r["Kernel_Name"], int(r["Grid_Size"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    segments, cur = [], None
    for _, kname, gsz, us in sorted(seq):
        if "affine_kernel" in kname and gsz == 256:
            cur = []
            segments.append(cur)
        elif cur is not None:
            cur.append((kname, us))
    if len(segments) != len(manifest):
        return [None] * len(manifest)
    out = []
    for m, seg in zip(manifest, segments):
        vals = [v for k, v in seg if short(k) == m["symbol"]][:m["launches"]]
        out.append(sum(vals) / len(vals) if vals else None)
    return out


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
    dur = durations(path, manifest)
    lines = ["| target | kernel | launch | " + " | ".join(have) + " | MFMA busy % | us | MHz |", "|---|---|---|" + "---|" * (len(have) + 3)]
    for i, m in enumerate(manifest):
        vals = [cols[c][i] for c in have]
        mf, ga = (cols.get("SQ_VALU_MFMA_BUSY_CYCLES") or [None] * len(manifest))[i], (cols.get("GRBM_GUI_ACTIVE") or [None] * len(manifest))[i]
        util = "" if not mf or not ga else f"{100.0 * mf / (CU * SIMD * ga / XCD):.1f}"
        us = "" if not dur[i] else f"{dur[i]:.1f}"
        mhz = "" if not ga or not dur[i] else f"{ga / XCD / dur[i]:.0f}"
        lines.append(f"| {m['name']} | `{m['symbol']}` | {m['label']} | " + " | ".join("-" if v is None else f"{v:.0f}" for v in vals) + f" | {util} | {us} | {mhz} |")
    print("\n".join(lines))
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
