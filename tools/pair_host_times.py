"""Host-side time stamps of both threads of one warm ma_set_pair (debug build: make variant NAME=pt DEFS=-DMA_PAIR_TIMES=1,
run with MAPALIGN_LIB=.../libmapalign_pt.so).  Prints the stamps of the last three calls; FRESH=1 hands over new arrays
every call (no page-pin reuse inside the HIP runtime)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
fresh = os.environ.get("FRESH") == "1"
for k in range(6):
    if k >= 3: sys.stderr.write("--- call %d\n" % k); sys.stderr.flush()
    else: os.environ["X"] = "1"
    gg, cc = (g.copy(), c.copy()) if fresh else (g, c)
    eng.set_pair(gg, cc); eng.icp(50, 0.0, 0.1)
