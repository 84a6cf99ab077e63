"""Developer smoke: parity vs oracle at 10k, timing at 1M (run on the GPU box)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd")); sys.path.insert(0, ROOT)
import numpy as np
import mapalign
from mapalign import synthetic as syn
from oracle import icp_oracle as orc

def small():
    g, c, R0, t0 = syn.make_pair(0, 10000, resample=False)
    eng = mapalign.MapAligner(0)
    print("grid", eng.set_gold(g))
    eng.set_source(c)
    idx, d2 = eng.nn(None, np.inf)
    oi, od2 = orc.nn_bruteforce(g, c)
    print("nn ungated mismatches", int((idx != oi).sum()), "max d2 diff", float(np.abs(d2 - od2).max()))
    idx, d2 = eng.nn(None, 0.1)
    m = od2 <= 0.01
    print("nn gated: matched gpu", int((idx != 0xFFFFFFFF).sum()), "oracle", int(m.sum()),
          "mismatch among matched", int((idx[m] != oi[m]).sum()))
    res = eng.icp(50, 0.0, 0.1, with_trace=True, count_work=True)
    o = orc.icp_oracle(g, c, 50, 0.0, 0.1)
    print("icp err vs oracle", orc.rt_error(res.R, res.t, o["R"], o["t"]), "vs truth", orc.rt_error(res.R, res.t, R0, t0))
    print("iters", res.iterations, "rmse", res.rmse, o["rmse"], "inl", res.inliers, o["inliers"], "gpu_ms", res.gpu_ms)
    a0 = res.trace[0, :17]; b0 = o["trace"][0]["acc"]
    print("acc0 rel diff", float(np.abs(a0 - b0).max() / np.abs(b0).max()))
    print("work rows/cands per iter", res.trace[:3, 18:20])
    eng.close()

def big(n=1_000_000, cell=0.0):
    g, c, R0, t0 = syn.make_pair(0, n, resample=True)
    eng = mapalign.MapAligner(0)
    t = time.time(); gi = eng.set_gold(g, cell); t1 = time.time() - t
    t = time.time(); eng.set_source(c); t2 = time.time() - t
    print("grid", gi, "set_gold s", round(t1, 3), "set_source s", round(t2, 3))
    for rep in range(3):
        res = eng.icp(50, 0.0, 0.1)
        print("icp 50 it: gpu_ms %.3f -> %.1f it/s ; rmse %.5f inl %d err-vs-truth %s" % (
            res.gpu_ms, 50 / res.gpu_ms * 1e3, res.rmse, res.inliers, orc.rt_error(res.R, res.t, R0, t0)))
    res = eng.icp(50, 0.0, 0.1, time_kernels=True)
    kt = eng.kernel_times()
    print("kernel times: nn avg us %.1f solve avg us %.1f" % (kt["nn_ms"] / kt["nn_launches"] * 1e3, kt["solve_ms"] / kt["solve_launches"] * 1e3))
    res = eng.icp(50, 0.0, 0.1, count_work=True, with_trace=True)
    print("counted run gpu_ms", res.gpu_ms, "rows/cands per point it0,it10,it49:",
          res.trace[[0, 10, 49], 18:20] / n)
    eng.close()

if __name__ == "__main__":
    small()
    for cell in ([float(a) for a in sys.argv[1:]] or [0.0]):
        big(cell=cell)
