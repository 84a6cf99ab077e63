"""How an alignment's time depends on stray points far from the scene (reflections through windows, sensor noise):
the 1M-point benchmark pair with K stray points scattered over a box S times the scene's extent, in the gold map, in the
scan, or in both.  Prints grid, loop time and the error against the known transform for each case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np
import mapalign
from mapalign import synthetic as syn

n = int(os.environ.get("N", "1000000"))
gold, comp, R0, t0 = syn.make_pair(0, n)
rng = np.random.default_rng(7)
lo, hi = gold.min(0), gold.max(0)
ctr, ext = 0.5 * (lo + hi), (hi - lo)


def stray(k, scale):
    return (ctr + rng.uniform(-0.5, 0.5, size=(k, 3)) * ext.max() * scale).astype(np.float32)


eng = mapalign.MapAligner(0)
cases = [("clean", 0, 1.0, "none")]
for k, s in ((20, 5.0), (20, 50.0), (2000, 5.0), (2000, 50.0), (20, 1000.0)):
    for where in ("gold", "scan", "both"):
        cases.append((f"{k} stray points over {s:g}x the extent in {where}", k, s, where))
for name, k, s, where in cases:
    g = np.concatenate([gold, stray(k, s)]) if where in ("gold", "both") else gold
    c = np.concatenate([comp, stray(k, s)]) if where in ("scan", "both") else comp
    t_0 = time.perf_counter()
    grid = eng.set_gold(g)
    eng.set_source(c)
    t_build = (time.perf_counter() - t_0) * 1e3
    r = eng.icp(50, 0.0, 0.10)          # (the first run says what a run costs: a library without the robust box needs seconds here)
    ms = r.gpu_ms if r.gpu_ms > 200.0 else min(eng.icp(50, 0.0, 0.10).gpu_ms for _ in range(3))
    bulk = eng.grid_box()["bulk_only"] if hasattr(eng, "grid_box") else None
    print(f"{name:58s} cell {grid['cell']*100:7.2f} cm dims {tuple(grid['dims'])!s:20s} bulk box {bulk!s:5s} staging {t_build:7.2f} ms  loop {ms:8.3f} ms  lists read {int((eng.list_plan() == 2).sum()):2d}  "
          f"|dR| {np.linalg.norm(r.R - R0):.2e} |dt| {np.linalg.norm(r.t - t0):.2e} inliers {int(r.inliers)}", flush=True)
