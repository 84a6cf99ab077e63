"""Inter-iteration timeline from the MA_WGTIMES debug build (GPU box only): where the time between
the last search workgroup of iteration k and the first one of iteration k+1 goes.

  make -C 3d-localisation-and-mapping_amd/csrc variant NAME=wgt DEFS=-DMA_WGTIMES=1
  python tools/rstimeline.py
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MAPALIGN_LIB", os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib", "libmapalign_wgt.so"))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign
from mapalign import synthetic as syn, _lib

g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
eng.set_gold(g, 0.0)
eng.set_source(c)
for _ in range(3):
    eng.icp(50, 0.0, 0.1)
lib = _lib.load()
times = np.zeros((64, 4096, 2), dtype=np.uint64)
hw = np.zeros((64, 4096, 2), dtype=np.uint32)
lib.ma_debug_wgtimes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
assert lib.ma_debug_wgtimes(eng._ctx, times.ctypes.data, hw.ctypes.data) == 0
rs = np.zeros((64, 20, 5), dtype=np.uint64)
lib.ma_debug_rstimes.argtypes = [C.c_void_p, C.c_void_p]
assert lib.ma_debug_rstimes(eng._ctx, rs.ctypes.data) == 0
G = int((times[0, :, 1] > 0).sum())
us = 0.01
print("all times in us relative to the END of the last search workgroup of the iteration (wall_clock64, 10 ns ticks)")
rows = []
for it in range(2, 49):
    t = times[it, :G].astype(np.int64)
    end_search = t[:, 1].max()
    r = rs[it].astype(np.int64)
    sums = r[:19]
    last = int(np.argmax(sums[:, 2]))              # the last arriver took its ticket last
    nxt = times[it + 1, :G].astype(np.int64)
    row = [(sums[:, 0].min() - end_search) * us, (sums[:, 0].max() - end_search) * us, (sums[:, 1].max() - end_search) * us,
           (sums[:, 2].max() - end_search) * us, (sums[last, 3] - end_search) * us, (sums[last, 4] - end_search) * us,
           (nxt[:, 0].min() - end_search) * us, (np.percentile(nxt[:, 0], 50) - end_search) * us,
           (r[19, 4] - end_search) * us, (r[19, 1] - end_search) * us, (r[19, 2] - end_search) * us, (r[19, 3] - end_search) * us]
    rows.append(row)
rows = np.array(rows)
names = ["first reduce wg enters", "last reduce wg enters", "last sum stored", "last ticket back", "19 sums read", "solve done (state stored)",
         "first search wg of next iteration starts", "median search wg of next iteration starts", "rank workgroup done",
         "solve: trace row written, Kabsch step begins", "solve: Kabsch step done", "solve: state, mirror and slot stores issued"]
for n, col in zip(names, rows.T):
    print(f"  {n:45s} median {np.median(col):6.2f}   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
