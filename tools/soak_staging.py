"""Soak of the staging path (round 3: helper thread + second stream, pinned mailbox, own radix sort, probe tables that must
be all-zero between calls): pairs of different sizes and extents ALTERNATE through one context; every call's grid and
every transform must be bitwise equal to the first call with the same pair (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
pairs = [syn.make_pair(1, 10_000)[:2], syn.make_pair(2, 300_000, 250_000)[:2], syn.make_pair(3, 1_000_000)[:2],
         tuple(a.astype(np.float64) + np.array([4.5e5, 5.2e6, 100.0]) for a in syn.make_pair(4, 120_000)[:2]),
         tuple(5.0 * a for a in syn.make_pair(5, 50_000)[:2])]
eng = mapalign.MapAligner(0)
first, bad, calls = {}, 0, 0
rng = np.random.default_rng(0)
for k in range(int(os.environ.get("CALLS", 1500))):
    i = int(rng.integers(0, len(pairs)))
    g, c = pairs[i]
    eng.set_pair(g, c)
    gi = eng.grid_info()
    r = eng.icp(8, 0.0, 0.1, with_trace=True)
    key = (gi["cell"], tuple(gi["dims"]), gi["n_occupied"], r.T.tobytes(), r.trace.tobytes())
    if i not in first:
        first[i] = key
    elif key != first[i]:
        bad += 1
    calls += 1
print("%d set_pair + icp calls over %d alternating pairs (10k ... 1M points, one georeferenced fp64, one 5x larger): %d differ from the first call of their pair"
      % (calls, len(pairs), bad))
