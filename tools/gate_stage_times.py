"""Per-stage cost of the adaptor's gate schedule after the coarse stage (1M-point pair, scan 150 deg / 7 m off)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np
import mapalign
from mapalign import coarse, synthetic as syn

gold, _, _, _ = syn.make_pair(0, 1_000_000)
far = syn.make_pair(0, 1_000_000, angle_deg=150.0, t0=(7.0, -1.0, 0.5))[1]
eng = mapalign.MapAligner(0)
for rep in range(3):
    t0 = time.perf_counter(); eng.set_pair(gold, far); t1 = time.perf_counter()
    stages, T0, info = coarse.coarse_align(eng, gold, far, resident=True); t2 = time.perf_counter()
    line = [f"set_pair {1e3*(t1-t0):.2f}", f"coarse {1e3*(t2-t1):.2f} (bound {info['score_bound_m']:.2f} m)"]
    T = T0
    for gate, iters in ((0.8, 15), (0.4, 15), (0.2, 15)):
        t = time.perf_counter(); r = eng.icp(iters, 1e-5, gate, init=T); dt = time.perf_counter() - t
        T = r.T
        line.append(f"gate {gate}: {1e3*dt:.2f} ms wall, {r.gpu_ms:.2f} device, {r.iterations} it, rmse {r.rmse:.4f}")
    t = time.perf_counter(); r = eng.icp(50, 0.0, 0.1, init=T); dt = time.perf_counter() - t
    line.append(f"final 0.1: {1e3*dt:.2f} ms wall, {r.iterations} it, rmse {r.rmse:.4f}")
    print(" | ".join(line))
