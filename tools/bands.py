"""Distribution of true nearest-neighbour distances at the transforms of a few ICP iterations (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0); eng.set_gold(g, 0.0); eng.set_source(c)
edges = [0, 0.02, 0.05, 0.08, 0.10, 0.115, 0.13, 0.17, 0.25, 10]
for k in (0, 2, 5, 10, 14):
    T = np.eye(4) if k == 0 else eng.icp(k, 0.0, 0.1).T
    idx, d2 = eng.nn(T, 0.5)
    d = np.sqrt(d2)
    h = np.histogram(np.where(np.isfinite(d), d, 9.0), bins=edges)[0] / len(d) * 100
    print("after %2d iterations: %% of points by NN distance (cm) " % k + " ".join("%s-%s:%.1f" % (int(a*100) if a*100 == int(a*100) else a*100, int(b*100) if b*100 == int(b*100) else b*100, v) for a, b, v in zip(edges[:-1], edges[1:], h)))
