"""One-off soak of the in-launch hand-off (k_reduce_solve) and the schedule: many alignments, every
result and every trace row must be bitwise equal to the first (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
for n, runs in ((10_000, 3000), (300_000, 600), (1_000_000, 300)):
    g, c, R0, t0 = syn.make_pair(5, n)
    eng = mapalign.MapAligner(0); eng.set_gold(g); eng.set_source(c)
    ref = eng.icp(50, 0.0, 0.1, with_trace=True)
    plan = eng.list_plan()          # what each iteration launched (search / search + write lists / list launch): the host's plan must repeat too
    bad = 0
    for k in range(runs):
        r = eng.icp(50, 0.0, 0.1, with_trace=True)
        if not (np.array_equal(r.T, ref.T) and np.array_equal(r.trace, ref.trace) and np.array_equal(eng.list_plan(), plan)):
            bad += 1
    print("n = %7d: %d alignments x 50 iterations, %d differ from the first (transform, trace rows or launch plan); plan %s" % (
        n, runs, bad, "".join("SWL"[int(v)] for v in plan)), flush=True)
