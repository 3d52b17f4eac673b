"""Measured parity errors kept on file: when DRBA_PARITY_REPORT names a file, every GPU test appends the rows it checked
(name, measured error, tolerance, details) under its own heading -- the driver's pass/fail dots do not say how much of a
tolerance was used.  `profiles/rNN_parity_report.txt` is a copy of such a file from a full `pytest -m gpu` run."""
import os


def record(section, rows):
    path = os.environ.get("DRBA_PARITY_REPORT")
    if not path:
        return
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "a") as f:
        f.write(f"\n== {section}\n")
        for name, err, tol, extra in rows:
            f.write(f"  {'ok ' if err <= tol else 'BAD'} {name:58s} err={err:.3e} tol={tol:.1e} {extra}\n")
