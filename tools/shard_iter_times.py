"""What one rank of the sharded-source loop sees: per-iteration launch times of the search / list launches and of the
reduce + solve launch when the context holds the whole 1M-point gold map but only 1/g of the comparison cloud
(g = 1, 2, 4, 8; rank 0's block of mapalign.dist's partition), and the 1-rank-communicator loop beside ma_icp.
The pieces of the predicted g-rank iteration of DESIGN.md 7 (the collective itself cannot be measured on one GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn, dist as mdist
g, c, R0, t0 = syn.make_pair(0, 1_000_000)
eng = mapalign.MapAligner(0)
eng.set_gold(g)
for ranks in (1, 2, 4, 8):
    lo, hi = mdist.shard_bounds(len(c), ranks, 0)
    eng.set_source(c[lo:hi])
    for _ in range(3): eng.icp(50, 0.0, 0.1)
    ms = sorted(eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(7))
    eng.icp(50, 0.0, 0.1, time_kernels=True)
    it = eng.iteration_times() * 1e3
    plan = "".join("SWL"[int(v)] for v in eng.list_plan())
    G, _ = eng.search_geometry()
    print(f"1/{ranks} of the scan ({hi - lo} points, {G} workgroups): loop {ms[0]:.3f} ms min, {ms[3]:.3f} med | plan {plan}")
    print("   search/list launch us: " + " ".join("%.0f" % v for v in it[:, 0]))
    print("   reduce+solve us mean %.1f | iterations 0-16 mean %.1f us, 17-27 %.1f, 28-49 %.1f" % (
        it[:, 1].mean(), it[:17, 0].mean(), it[17:28, 0].mean(), it[28:, 0].mean()))
eng.set_source(c)
ref = eng.icp(50, 0.0, 0.1)
eng.comm_init(eng.comm_unique_id(), 1, 0)
for _ in range(3): eng.icp_sharded(50, 0.0, 0.1)
a = sorted(eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(9)); b = sorted(eng.icp_sharded(50, 0.0, 0.1).gpu_ms for _ in range(9))
print(f"whole scan: ma_icp {a[0]:.3f} / {a[4]:.3f} ms, ma_icp_sharded (1-rank communicator: k_reduce19 + all-reduce + k_solve) {b[0]:.3f} / {b[4]:.3f} ms "
      f"-> {(b[4] - a[4]) / 50 * 1e3:.1f} us per iteration for the split launches and the (1-rank) collective")
eng.comm_destroy()
# the same whole scan through the peer-to-peer exchange with ONE rank (k_reduce_exchange_solve: the sums go to this rank's own
# row and come straight back — the launch structure of a g-rank run, everything but the peers' write latency)
eng.p2p_init([eng.p2p_export()], 0)
for _ in range(3): eng.icp_sharded(50, 0.0, 0.1)
p = sorted(eng.icp_sharded(50, 0.0, 0.1).gpu_ms for _ in range(9)); a = sorted(eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(9))
r = eng.icp_sharded(50, 0.0, 0.1)
print(f"whole scan: ma_icp {a[0]:.3f} / {a[4]:.3f} ms, ma_icp_sharded over the 1-rank peer-to-peer exchange {p[0]:.3f} / {p[4]:.3f} ms "
      f"-> {(p[4] - a[4]) / 50 * 1e3:.2f} us per iteration for the exchange inside the second launch; bitwise ma_icp: {np.array_equal(r.T, ref.T)}")
eng.p2p_destroy()
