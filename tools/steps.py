"""How far does each solve move the comparison cloud?  (What the list planner sees: list_planner.h `moved`.)

    python3 tools/steps.py [points] [iterations]

Re-forms every iteration's rigid step from the 17 sums of the trace (Kabsch on H = S - W mu_p mu_q^T) and prints the
largest displacement over the corners of the cloud's bounding box, the decay from step to step, and what remains to be
moved after each iteration — against the skin (cell edge / 7): lists are read while the motion since they were written
stays below 0.49 skin and written when a step is below 0.25 skin."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g, c, R0, t0 = syn.make_pair(0, n)
eng = mapalign.MapAligner(0)
grid = eng.set_gold(g); eng.set_source(c)
r = eng.icp(iters, 0.0, 0.1, with_trace=True, lists=False)
lo, hi = c.min(0).astype(np.float64), c.max(0).astype(np.float64)
corners = np.array([[(hi if (k >> a) & 1 else lo)[a] for a in range(3)] for k in range(8)])
R, t = np.eye(3), np.zeros(3)
steps = []
for k in range(iters):
    a = r.trace[k]
    W, mp, mq = a[0], a[1:4] / a[0], a[4:7] / a[0]
    H = a[7:16].reshape(3, 3) - W * np.outer(mp, mq)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    dR = Vt.T @ D @ U.T
    dt = mq - dR @ mp
    Rn, tn = dR @ R, dR @ t + dt
    d = (corners @ (Rn - R).T + (tn - t))
    steps.append(np.sqrt((d * d).sum(1)).max())
    R, t = Rn, tn
steps = np.array(steps)
skin = grid["cell"] / 7
print(f"cell {grid['cell'] * 100:.2f} cm, skin {skin * 1e3:.2f} mm: write below {0.25 * skin * 1e3:.2f} mm per step, read within {0.49 * skin * 1e3:.2f} mm")
rest = steps[::-1].cumsum()[::-1]
for k in range(iters):
    print(f"it {k:2d}: step {steps[k] * 1e3:8.3f} mm  ratio {steps[k] / steps[k - 1] if k else float('nan'):5.2f}  still to move after it {(rest[k] - steps[k]) * 1e3:8.3f} mm"
          f"  (= {(rest[k] - steps[k]) / skin:5.2f} skin)")
