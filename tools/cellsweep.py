import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, int(os.environ.get("N", "1000000")))
eng = mapalign.MapAligner(0)
for cell in [float(a) for a in sys.argv[1:]]:
    gi = eng.set_gold(g, cell); eng.set_source(c)
    for _ in range(2): eng.icp(50, 0.0, 0.1)
    ts = [eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(5)]
    r = eng.icp(50, 0.0, 0.1, count_work=True, with_trace=True)
    print("cell %.4f (%s) loop ms min %.3f | words/pt it0,10,49: %s cands/pt: %s | build %.2f ms" % (gi["cell"], gi["dims"], min(ts),
          np.round(r.trace[[0,10,49],18]/1e6,1), np.round(r.trace[[0,10,49],19]/1e6,1), gi["build_ms"]))
