"""One-off soak of concurrent contexts: the same pairs through align_many with 1, 2, 3 and 4 workers, repeatedly;
every 4x4 must equal the sequential result bit for bit (GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn
pairs = [syn.make_pair(100 + k, 80_000 + 7_000 * (k % 5), 60_000 + 3_000 * (k % 7))[:2] for k in range(40)]
ref = mapalign.align_many(pairs, workers=1)
bad = runs = 0
for rep in range(6):
    for w in (2, 3, 4):
        out = mapalign.align_many(pairs, workers=w)
        runs += len(out)
        bad += sum(not np.array_equal(a, b) for a, b in zip(out, ref))
print("%d alignments through 2-4 concurrent contexts, %d differ from the sequential result" % (runs, bad))
