"""One steady-state set_gold + set_source under rocprofv3 (kernel trace) -> per-kernel totals of the build."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
eng.set_gold(g); eng.set_source(c)        # allocations
eng.set_gold(g); eng.set_source(c)        # the measured one = the last occurrence of each kernel group
