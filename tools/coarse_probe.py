import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import coarse, synthetic as syn, transforms
gold, comp, R0, t0 = syn.make_pair(0, 1_000_000)
far = syn.make_pair(0, 1_000_000, angle_deg=150.0, t0=(7.0, -1.0, 0.5))[1]
eng = mapalign.MapAligner(0); eng.set_gold(gold)
def T(f, *a):
    t=time.perf_counter(); r=f(*a); return (time.perf_counter()-t)*1e3, r
for rep in range(2):
    tg, (cg, ag, _) = T(coarse.principal_frame, gold)
    tc, (cc, ac, _) = T(coarse.principal_frame, far)
    rng = np.random.default_rng(0)
    ts, sub = T(lambda: far[rng.choice(len(far), 20000, replace=False)])
    tss, _ = T(eng.set_source, sub.astype(np.float32))
    print("pca gold %.1f pca comp %.1f subsample %.1f set_source %.2f" % (tg, tc, ts, tss))
    for R in coarse.candidate_rotations(ag, ac):
        Tm = transforms.rt_to_matrix(np.eye(3), cg - cc) @ transforms.rotation_about_centre(R, cc)
        for gate in (np.inf, 1.0, 0.3):
            tn, (idx, d2) = T(eng.nn, Tm, gate)
            print("   nn gate %s: %.2f ms  rms(trunc 1m) %.3f" % (gate, tn, np.sqrt(np.mean(np.minimum(d2, 1.0)))))
