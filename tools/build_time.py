"""Steady-state cost of the per-pair setup: set_gold (grid build) and set_source, against the alignment itself (GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g, c, R0, t0 = syn.make_pair(0, n)
eng = mapalign.MapAligner(0)
for cell in (0.0, 0.0353):
    for rep in range(4):
        t0_ = time.perf_counter(); gi = eng.set_gold(g, cell); t1 = time.perf_counter()
        eng.set_source(c); t2 = time.perf_counter()
        r = eng.icp(50, 0.0, 0.1); t3 = time.perf_counter()
        print("cell %-6s rep %d: set_gold %.2f ms (device build %.2f) | set_source %.2f ms | icp %.2f ms" % (cell or "auto", rep, (t1 - t0_) * 1e3, gi["build_ms"], (t2 - t1) * 1e3, (t3 - t2) * 1e3))
