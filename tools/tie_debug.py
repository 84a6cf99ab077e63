"""Which query makes lists on / off differ, and which of its candidates is the canonical one?  (debug build:
make variant NAME=dbg DEFS="-DMA_DEBUG_FETCH=1", run with MAPALIGN_LIB=.../libmapalign_dbg.so)

    python3 tools/tie_debug.py <fuzz case number> [...]

Replays cases of tests/test_gpu_round4.py's fuzz, finds the first iteration whose 17 sums differ, re-runs both arms up to
and including that iteration and fetches every source point's match (prev[], written by list launches too in the debug
build).  For every query whose match differs: both candidates' positions in the device order, their fp32 squared
distances re-computed on the host in the kernel's association, and which arm holds the smaller (distance, position)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, mapalign
from mapalign import _lib
import test_gpu_round4 as t4
lib = _lib.load()
lib.ma_debug_fetch.restype = C.c_int
lib.ma_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint64]
def fetch(eng, what, n):
    out = np.zeros(n, dtype=np.uint32)
    assert lib.ma_debug_fetch(eng._ctx, what, out.ctypes.data_as(C.POINTER(C.c_uint32)), n) == 0
    return out
want = [int(a) for a in sys.argv[1:]] or [95]
rng = np.random.default_rng(31)
eng = mapalign.MapAligner(0)
for case in range(max(want) + 1):
    gold, comp, w, gate, cell, iters = t4._fuzz_case(rng, case)
    if case not in want:
        continue
    eng.set_gold(gold, cell); eng.set_source(comp, w)
    on = eng.icp(iters, 0.0, gate, with_trace=True); plan = "".join("SWL"[int(v)] for v in eng.list_plan())
    off = eng.icp(iters, 0.0, gate, with_trace=True, lists=False)
    diff = (on.trace[:, :18] != off.trace[:, :18]).any(1)
    if not diff.any():
        print(f"case {case}: no difference"); continue
    k = int(np.argmax(diff))
    print(f"case {case}: gate {gate} cell {cell} iters {iters} n {len(comp)} m {len(gold)}; first differing iteration {k}; plan {plan}")
    cols = np.nonzero(on.trace[k, :18] != off.trace[k, :18])[0]
    print("   columns", cols, " W on/off", on.trace[k, 0], off.trace[k, 0], " E on - off %.3e" % (on.trace[k, 16] - off.trace[k, 16]),
          " sum(w q) on - off", on.trace[k, 4:7] - off.trace[k, 4:7])
    n, m = len(comp), len(gold)
    eng.icp(k + 1, 0.0, gate); plan_k = "".join("SWL"[int(v)] for v in eng.list_plan())
    m_on = fetch(eng, 0, n)
    eng.icp(k + 1, 0.0, gate, lists=False)
    m_off = fetch(eng, 0, n)
    sperm, gperm = fetch(eng, 1, n), fetch(eng, 2, m)
    print(f"   plan of the {k + 1}-iteration rerun {plan_k} (last launch: {plan_k[-1]})")
    d = np.nonzero(m_on != m_off)[0]
    print(f"   {len(d)} source slots with different matches")
    # the transform launch k ran with: compose the steps of iterations 0..k-1 from the (identical) sums of `off`
    R, t = np.eye(3), np.zeros(3)
    for i in range(k):
        a = off.trace[i]; Wt, mp, mq = a[0], a[1:4] / a[0], a[4:7] / a[0]
        H = a[7:16].reshape(3, 3) - Wt * np.outer(mp, mq)
        U, S, Vt = np.linalg.svd(H); D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
        dR = Vt.T @ D @ U.T; R, t = dR @ R, dR @ t + (mq - dR @ mp)
    sg, ss = eng.get_gold(), eng.get_source_stored()
    info = eng.grid_info()
    for slot in d[:10]:
        i = int(sperm[slot])
        p = (R @ ss[i] + t)
        for name, pos in (("on ", int(m_on[slot])), ("off", int(m_off[slot]))):
            if pos == 0xFFFFFFFF:
                print(f"     slot {slot} point {i}: {name} no match"); continue
            j = int(gperm[pos]); q = sg[j]
            df = (p - info["origin"]).astype(np.float32) - (q - info["origin"]).astype(np.float32)
            d2 = np.float32(df[2] * df[2] + np.float32(df[1] * df[1] + np.float32(df[0] * df[0])))
            print(f"     slot {slot} point {i}: {name} position {pos} (gold index {j}) d2 ~ {d2!r} (host re-computation, transform re-composed: approximate) |p-q| {np.linalg.norm(p - q):.9f}")
