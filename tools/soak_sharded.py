"""One-off soak of the sharded-source loop's early exit (1-rank communicator, GPU box): the same run with a convergence
test many times — the transform, the stopping iteration AND the number of launches enqueued past it must repeat (what
keeps the collective counts of several ranks equal: the stop is decided from the iteration at which the loop stopped,
not from when the host looked)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
for n, runs in ((20_000, 1500), (300_000, 400)):
    g, c, R0, t0 = syn.make_pair(11, n, angle_deg=0.4, t0=(0.02, -0.015, 0.01))
    eng = mapalign.MapAligner(0); eng.set_pair(g, c)
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    for tol in (1e-4, 1e-6, 1e-8):
        ref = eng.icp_sharded(90, tol, 0.1, with_trace=True)
        enq, plan = eng.planner_diag()["enqueued"], eng.list_plan()
        bad = 0
        for k in range(runs):
            r = eng.icp_sharded(90, tol, 0.1, with_trace=True)
            same = (np.array_equal(r.T, ref.T) and r.iterations == ref.iterations and np.array_equal(r.trace, ref.trace)
                    and eng.planner_diag()["enqueued"] == enq and np.array_equal(eng.list_plan(), plan))
            bad += not same
        print("n = %7d tol %g: %d sharded runs, stopped after %d iterations with %d enqueued, %d differ from the first (transform, trace, "
              "stop, launches enqueued or plan)" % (n, tol, runs, ref.iterations, enq, bad), flush=True)
    eng.comm_destroy(); eng.close()
