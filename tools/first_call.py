"""First-call latency in a fresh process (what a one-shot run of the reference's pipeline pays): library load, context,
first staging of a 1M-point pair (allocations), first loop; then the second call for comparison (GPU box)."""
import os, sys, time
t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np
import mapalign
from mapalign import synthetic as syn
t_import = time.perf_counter() - t00
g, c, _, _ = syn.make_pair(0, 1_000_000)
t = time.perf_counter(); eng = mapalign.MapAligner(0); t_ctx = time.perf_counter() - t
t = time.perf_counter(); eng.set_pair(g, c); t_pair = time.perf_counter() - t
a0 = eng.mem_info()
t = time.perf_counter(); r = eng.icp(50, 0.0, 0.1); t_icp = time.perf_counter() - t
a1 = eng.mem_info()
t = time.perf_counter(); eng.set_pair(g, c); t_pair2 = time.perf_counter() - t
t = time.perf_counter(); eng.icp(50, 0.0, 0.1); t_icp2 = time.perf_counter() - t
print(f"import {t_import*1e3:.0f} ms | ma_create {t_ctx*1e3:.1f} ms | first set_pair {t_pair*1e3:.1f} ms ({a0['device_allocs']} device allocations, "
      f"{a0['bytes_live']/1e6:.0f} MB) | first icp {t_icp*1e3:.1f} ms (+{a1['device_allocs']-a0['device_allocs']} allocations) | "
      f"second set_pair {t_pair2*1e3:.2f} ms, second icp {t_icp2*1e3:.2f} ms")
