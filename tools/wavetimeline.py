"""Do the four wavefronts of a heavy workgroup of the early search launches finish their searches at different times?
(if the waves of the longest workgroups of iterations 0-5 differ by >= 25 %, a workgroup-shared work
queue could even them out; if not, the per-wavefront queues are as good as a shared one and k_nn_accum's search is closed.)

  make -C 3d-localisation-and-mapping_amd/csrc variant NAME=wgt DEFS=-DMA_WGTIMES=1
  MAPALIGN_AB_PARTIAL=1 python tools/wavetimeline.py

MA_WGTIMES debug build: every wavefront stamps the 100 MHz clock when its search (wave_search) is done, relative to its
workgroup's start.  Per iteration: over all workgroups and over the slowest tenth (by workgroup duration), the spread
(slowest wave - fastest wave) / slowest wave, and how much earlier the workgroup's search phase would end if the four
waves shared their work perfectly (mean of the four instead of their max).
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MAPALIGN_LIB"] = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib", "libmapalign_wgt.so")
os.environ["MAPALIGN_AB_PARTIAL"] = "1"
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign
from mapalign import synthetic as syn, _lib

g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
eng.set_gold(g)
eng.set_source(c)
for _ in range(3):
    eng.icp(50, 0.0, 0.1)
lib = _lib.load()
wt = np.zeros((16, 4096, 4), dtype=np.uint32)
lib.ma_debug_wavetimes.argtypes = [C.c_void_p, C.c_void_p]
assert lib.ma_debug_wavetimes(eng._ctx, wt.ctypes.data) == 0
times = np.zeros((64, 4096, 2), dtype=np.uint64)
hw = np.zeros((64, 4096, 2), dtype=np.uint32)
lib.ma_debug_wgtimes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
assert lib.ma_debug_wgtimes(eng._ctx, times.ctypes.data, hw.ctypes.data) == 0
G = int((times[0, :, 1] > 0).sum())
tick = 0.01
print(f"{G} workgroups of 4 wavefronts; times in us (100 MHz clock)")
for it in range(0, 12):
    w = wt[it, :G].astype(np.float64) * tick
    dur = (times[it, :G, 1].astype(np.int64) - times[it, :G, 0].astype(np.int64)) * tick
    span = (times[it, :G, 1].max() - times[it, :G, 0].min()) * tick
    mx, mn, mean = w.max(1), w.min(1), w.mean(1)
    spread = (mx - mn) / np.maximum(mx, 1e-9)
    heavy = dur >= np.percentile(dur, 90)
    def q(a): return f"{np.percentile(a, 50):5.2f} / {np.percentile(a, 90):5.2f} / {a.max():5.2f}"
    print(f"iter {it:2d}: launch {span:6.1f} us | wg duration p50 {np.percentile(dur, 50):5.1f} p90 {np.percentile(dur, 90):5.1f} max {dur.max():5.1f} | "
          f"wave spread (max-min)/max p50/p90/max: all {q(spread)}  slowest tenth {q(spread[heavy])} | "
          f"slowest tenth: search ends at {np.median(mx[heavy]):5.1f} us, at {np.median(mean[heavy]):5.1f} us if the 4 waves shared evenly "
          f"({100 * (1 - np.median(mean[heavy] / mx[heavy])):4.1f} % of the search phase, {100 * np.median((mx[heavy] - mean[heavy]) / dur[heavy]):4.1f} % of the workgroup)")
