"""Where a warm mapalign.align(gold, comp) from host arrays spends its time (GPU box): the pieces one by one, the whole
call with the two staging calls (round 2's path) and with ma_set_pair (round 3), 1M-point pair, 50 iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, ROOT)
import numpy as np
import mapalign
from mapalign import synthetic as syn

g, c, R0, t0 = syn.make_pair(0, int(os.environ.get("N", 1_000_000)))
eng = mapalign.MapAligner(0)
for _ in range(2):
    eng.set_gold(g); eng.set_source(c); eng.icp(50, 0.0, 0.1)

def med(f, n=7):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[n // 2]

t_gold = med(lambda: eng.set_gold(g))
gi = eng.set_gold(g)
t_src = med(lambda: eng.set_source(c))
t_icp = med(lambda: eng.icp(50, 0.0, 0.1))
t_two = med(lambda: (eng.set_gold(g), eng.set_source(c), eng.icp(50, 0.0, 0.1)))
t_pair = med(lambda: (eng.set_pair(g, c), eng.icp(50, 0.0, 0.1)))
t_pair_only = med(lambda: eng.set_pair(g, c))
print(f"set_gold {t_gold:.2f} ms (device build {gi['build_ms']:.2f}) | set_source {t_src:.2f} | icp {t_icp:.2f} | "
      f"two calls + icp {t_two:.2f} | set_pair {t_pair_only:.2f} | set_pair + icp {t_pair:.2f}")
mapalign.align(g, c)
print(f"mapalign.align warm: {med(lambda: mapalign.align(g, c), 9):.2f} ms")
