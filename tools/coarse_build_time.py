import os, sys, time
sys.path.insert(0, "3d-localisation-and-mapping_amd")
import numpy as np, mapalign
from mapalign import synthetic as syn
gold, comp, R0, t0 = syn.make_pair(0, 1_000_000)
for rep in range(2):
    eng = mapalign.MapAligner(0); eng.set_gold(gold); eng.set_source(comp)
    eng.icp(2, 0.0, 0.1)
    t = time.perf_counter(); r1 = eng.icp(1, 0.0, np.inf); a = (time.perf_counter() - t) * 1e3      # builds levels 1-3
    t = time.perf_counter(); r2 = eng.icp(1, 0.0, np.inf); b = (time.perf_counter() - t) * 1e3
    print(f"first ungated icp(1) {a:.2f} ms wall (coarse levels built), second {b:.2f} ms -> build ~{a - b:.2f} ms; same T: {np.array_equal(r1.T, r2.T)}")
    r = eng.icp(50, 0.0, np.inf)
    print("   50 ungated iterations %.2f ms, |R-R0| %.2e" % (r.gpu_ms, np.linalg.norm(r.R - R0)))
    eng.close()
