"""How many host threads the KD-tree oracle wants on this box: seconds per 1M-point icp_step for several n_jobs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, ROOT)
import numpy as np
from mapalign import synthetic as syn
from oracle import icp_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g, c, R0, t0 = syn.make_pair(0, n)
print("cores", len(os.sched_getaffinity(0)), flush=True)
for nj in (-1, 128, 64, 32, 16):
    t = time.perf_counter(); ix = orc.GoldIndex(g, n_jobs=nj); tb = time.perf_counter() - t
    R, tt = np.eye(3), np.zeros(3)
    ts = []
    for k in range(3):
        t = time.perf_counter(); acc, _, _, R, tt, ok = orc.icp_step(ix, c, R, tt, 0.10); ts.append(time.perf_counter() - t)
    print(f"n_jobs {nj}: build {tb:.2f} s, steps {' '.join('%.2f' % v for v in ts)} s", flush=True)
