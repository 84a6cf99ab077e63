"""Steady-state ma_set_pair + icp under `rocprofv3 --kernel-trace --memory-copy-trace`: the last pair's kernels and copies
in time order (tools/pair_timeline.py prints them from the CSVs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
for _ in range(4):
    eng.set_pair(g, c); eng.icp(50, 0.0, 0.1)
