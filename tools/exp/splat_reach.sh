#!/bin/bash
# The fused RIFE splats with the per-tile reach map (halo 4 / 8 / 16 chosen per output tile) against the build before it
# (tools/exp/build/lib_base.so, same ABI): parity of the splat cases, one-stream table rows, bench.
python -m pytest tests/test_gpu_parity.py -q -x -k "glue or drm or splat or rife" 2>&1 | tail -3
tools/exp/ab_table.sh tools/exp/build/lib_base.so "splat"
tools/exp/ab_bench_libs.sh tools/exp/build/lib_base.so -- --no-cpu-baseline --no-roofline --no-extra
