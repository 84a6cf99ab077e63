"""Workgroup timeline of k_nn_accum from the MA_WGTIMES debug build (GPU box only).

  make -C 3d-localisation-and-mapping_amd/csrc variant NAME=wgt DEFS=-DMA_WGTIMES=1
  python tools/wgtimeline.py [cell]

Per chosen iteration: kernel span, workgroup duration quantiles, how many workgroups are
resident over time (deciles of the span) and how unevenly the XCDs finish.
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MAPALIGN_LIB"] = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib", "libmapalign_wgt.so")
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign
from mapalign import synthetic as syn, _lib

cell = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
eng.set_gold(g, cell)
eng.set_source(c)
for _ in range(3):
    eng.icp(50, 0.0, 0.1)
lib = _lib.load()
times = np.zeros((64, 4096, 2), dtype=np.uint64)
hw = np.zeros((64, 4096, 2), dtype=np.uint32)
lib.ma_debug_wgtimes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
rc = lib.ma_debug_wgtimes(eng._ctx, times.ctypes.data, hw.ctypes.data)
assert rc == 0, rc
ph = np.zeros((64, 4096, 4), dtype=np.uint32)
lib.ma_debug_wgphases.argtypes = [C.c_void_p, C.c_void_p]
assert lib.ma_debug_wgphases(eng._ctx, ph.ctypes.data) == 0
G = int((times[0, :, 1] > 0).sum())
print("workgroups", G)
for it in (0, 1, 2, 5, 12, 20, 45):
    t = times[it, :G].astype(np.int64)
    t0_, t1_ = t[:, 0].min(), t[:, 1].max()
    tick_us = 0.01  # wall_clock64: 100 MHz
    span = (t1_ - t0_) * tick_us
    dur = (t[:, 1] - t[:, 0]) * tick_us
    start = (t[:, 0] - t0_) * tick_us
    end = (t[:, 1] - t0_) * tick_us
    print(f"iter {it:2d}: span {span:6.1f} us | wg dur us p10 {np.percentile(dur,10):5.1f} p50 {np.percentile(dur,50):5.1f} "
          f"p90 {np.percentile(dur,90):5.1f} max {dur.max():5.1f} | sum(dur)/span = {dur.sum()/span:7.1f} resident avg "
          f"| last start at {start.max():5.1f} us")
    edges = np.linspace(0, span, 11)
    res = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)
        res.append(ov)
    print("          resident workgroups per decile of the span:", " ".join(f"{r:6.0f}" for r in res))
    p = ph[it, :G].astype(np.float64) * tick_us
    seg = np.stack([p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 1], p[:, 3] - p[:, 2], dur - p[:, 3]], axis=1)
    late = start > 0.5 * span if (start > 0.5 * span).sum() > 50 else start >= 0
    print("          wave 0 phases, median us (all | started in the 2nd half): head loads %.1f|%.1f  3x3 block %.1f|%.1f  rings %.1f|%.1f  winner+sums %.1f|%.1f  reduction+write %.1f|%.1f"
          % tuple(v for k in range(5) for v in (np.median(seg[:, k]), np.median(seg[late, k]))))
    by = []
    for a, b in zip(edges[:-1], edges[1:]):
        m = (start >= a) & (start < b)
        by.append(f"{np.median(dur[m]):6.1f}" if m.any() else "     -")
    print("          median duration of the workgroups STARTED in that decile:", " ".join(by))
    xcc = hw[it, :G, 1] & 0xF
    per = [end[xcc == x].max() if (xcc == x).any() else 0 for x in range(8)]
    cnt = [int((xcc == x).sum()) for x in range(8)]
    print("          per-XCD finish us:", " ".join(f"{p:5.1f}" for p in per), "| wgs per XCD:", cnt)
    if it > 0:
        # how well did the previous iteration's segment costs predict this one's? (the schedule uses them)
        def seg_cost(k):
            tk = times[k, :G].astype(np.int64)
            ch = (hw[k, :G, 1] >> 4).astype(np.int64)
            c = np.zeros(G); c[ch] = (tk[:, 1] - tk[:, 0])
            return c.reshape(-1, 16).sum(1)
        a, b = seg_cost(it - 1), seg_cost(it)
        ra, rb = np.argsort(np.argsort(-a)), np.argsort(np.argsort(-b))
        top = rb < len(b) // 4
        print("          segment cost vs previous iteration: Pearson %.3f, rank %.3f; of this iteration's top-quarter segments %.0f %% were predicted in the top quarter, worst predicted rank %d of %d"
              % (np.corrcoef(a, b)[0, 1], np.corrcoef(ra, rb)[0, 1], 100.0 * (ra[top] < len(b) // 4).mean(), ra[top].max(), len(b)))
    hwid = hw[it, :G, 0]
    cu = (hwid >> 8) & 0xF
    se = (hwid >> 13) & 0x7
    key = xcc.astype(np.int64) * 1000 + se * 16 + cu
    uniq, counts = np.unique(key, return_counts=True)
    print(f"          distinct (xcc,se,cu): {len(uniq)}; wgs per CU min {counts.min()} max {counts.max()}")
