"""CPU oracle for the DRBA per-frame hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain fp32 CPU torch, the algorithm of the reference
(routineLife1/DRBA @ 2025-03-10) for the path named in BASELINE.json `north_star`.
Every function cites the reference file:line it follows.  It is the checker the
HIP path is compared against and the `cpu_baseline` timed by bench.py.

Rules (enforced by tests/test_abi.py::test_oracle_is_not_imported_by_the_product):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it;
  * nothing under drba_amd/ (the product) imports or falls back to it;
  * it never touches a GPU and never reads /root/reference at run time.

Pinning: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, imported in the build
container by tests/golden/make_golden.py (decorators stripped -> the reference's own
functions evaluated in fp32).  That script asserts oracle == reference
bit-for-bit on every fixture it writes, and tests/test_oracle_golden.py re-checks the
oracle against the committed fixtures on any box.

Third-party arithmetic: torch (reference requirements.txt:2 `torch>=2.5.1`, unpinned
upper bound); the concrete oracle version is this image's torch 2.10.0 CPU kernels for
conv2d / conv_transpose2d / interpolate / grid_sample / index_add_.
"""
from . import ops, drm, ifnet, rife, scdet, gmflow, gmfss  # noqa: F401
