"""Diagnostic parity report on a GPU box:  python -m tests.gpu_report [out.json]
Runs every GPU parity check without stopping at the first failure and prints a table."""
import json
import os
import sys
import time

import numpy as np
import torch

from tests import cases, gpu_checks
from tests.backends import HipBackend, OracleBackend

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def main():
    t0 = time.time()
    hip, ora = HipBackend(), OracleBackend()
    sections = []
    sections.append(("ops", gpu_checks.check_cases(cases.ops_cases(), hip, ora, 2e-5, np.load(os.path.join(GOLD, "ops.npz")))))
    sections.append(("drm", gpu_checks.check_cases(cases.drm_cases(), hip, ora, 2e-5, np.load(os.path.join(GOLD, "drm.npz")))))
    sections.append(("conv", gpu_checks.check_conv_layers(hip.dev)))
    sections.append(("glue", gpu_checks.check_glue(hip.dev)))
    sections.append(("window attention", gpu_checks.check_window_attention(hip.dev)))
    sections.append(("linear (split-bf16)", gpu_checks.check_linear_split(hip.dev)))
    sections.append(("feature splats", gpu_checks.check_splat_quad(hip.dev)))
    sections.append(("scdet", gpu_checks.check_scdet(hip, np.load(os.path.join(GOLD, "scdet.npz")))))
    gold = np.load(os.path.join(GOLD, "rife.npz"))
    for scale, size in cases.RIFE_CONFIGS:
        try:
            sections.append((f"rife s={scale} {size}", gpu_checks.check_rife(hip, ora, gold, scale, size)))
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            sections.append((f"rife s={scale}", [("EXC", float("inf"), 0.0, repr(e))]))
    if os.environ.get("DRBA_REPORT_GMFSS", "1") == "1":
        try:
            sections.append(("gmfss parts", gpu_checks.check_gmfss_parts(hip.dev)))
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            sections.append(("gmfss parts", [("EXC", float("inf"), 0.0, repr(e))]))
        gold = np.load(os.path.join(GOLD, "gmfss_union.npz"))
        for scale, size in cases.GMFSS_CONFIGS:
            try:
                sections.append((f"gmfss_union s={scale} {size}", gpu_checks.check_gmfss_union(hip, ora, gold, scale, size)))
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                sections.append((f"gmfss_union s={scale}", [("EXC", float("inf"), 0.0, repr(e))]))
        try:
            sections.append(("gmfss (non-union)", gpu_checks.check_gmfss_plain(hip, ora, np.load(os.path.join(GOLD, "gmfss.npz")))))
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            sections.append(("gmfss (non-union)", [("EXC", float("inf"), 0.0, repr(e))]))
    torch.cuda.synchronize()
    bad = 0
    out = {}
    for title, rows in sections:
        print(f"\n== {title}")
        for name, err, tol, extra in rows:
            ok = err <= tol
            bad += 0 if ok else 1
            print(f"  {'ok ' if ok else 'BAD'} {name:58s} err={err:.3e} tol={tol:.1e} {extra}")
            out[f"{title}/{name}"] = {"err": err if np.isfinite(err) else 1e30, "tol": tol, "extra": extra}
    print(f"\n{bad} failing checks, {time.time() - t0:.1f}s")
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(sys.argv[1]) or ".", exist_ok=True)
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
