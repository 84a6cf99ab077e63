"""Aggregate iterations/s when K alignments of resident 1M-point pairs iterate at the same time on one GPU
(one context = one stream per alignment, one host thread each) (GPU box)."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
engs = []
for k in range(4):
    g, c, R0, t0 = syn.make_pair(k, n)
    e = mapalign.MapAligner(0); e.set_gold(g); e.set_source(c); e.icp(50, 0.0, 0.10); engs.append((e, R0))
reps = 20
Ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 1, 2]
for K in Ks:
    errs = [0.0] * K
    def work(i):
        e, R0 = engs[i]
        for _ in range(reps):
            r = e.icp(50, 0.0, 0.10)
        errs[i] = float(np.linalg.norm(r.R - R0))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("%d alignments at once: %.0f ICP iterations/s in total (%.2f ms per 50-iteration alignment each; max |R-R0| %.1e)"
          % (K, K * reps * 50 / dt, dt / reps * 1e3, max(errs)))
