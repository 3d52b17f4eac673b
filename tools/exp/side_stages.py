#!/usr/bin/env python3
"""A/B: how many IFNet stages of the NEXT step the lookahead stream runs.  python tools/exp/side_stages.py N [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drba_amd.models.rife import RIFE  # noqa: E402
RIFE.SIDE_STAGES = int(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402
bench.main()
