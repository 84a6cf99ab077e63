"""Where Alignment(...).compare() on two 1M-point .ply files spends its time (GPU box): the pieces one by one."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np
import mapalign
from mapalign import coarse, ply, synthetic as syn, transforms

def T(f, *a, reps=5):
    f(*a)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(*a); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r

gold, _, _, _ = syn.make_pair(0, 1_000_000)
far = syn.make_pair(0, 1_000_000, angle_deg=150.0, t0=(7.0, -1.0, 0.5))[1]
d = tempfile.mkdtemp()
gp, sp = os.path.join(d, "gold.ply"), os.path.join(d, "scan.ply")
ply.write_ply(gp, gold); ply.write_ply(sp, far)
t_read, g = T(ply.read_ply, gp)
print(f"read_ply (one file, {g.dtype}): {t_read:.1f} ms")
t_read, g32 = T(lambda: ply.read_ply(gp, dtype=None))
print(f"read_ply (one file, dtype=None -> {g32.dtype}, a view: {not g32.flags.owndata}): {t_read:.1f} ms")
eng, lock = mapalign.api.cached_engine(0)
t, _ = T(eng.set_gold, g); print(f"set_gold: {t:.2f} ms")
c = ply.read_ply(sp)
t, (cg, ag, _) = T(coarse.principal_frame, g); print(f"principal_frame(gold): {t:.1f} ms")
t, (cc, ac, _) = T(coarse.principal_frame, c); print(f"principal_frame(scan): {t:.1f} ms")
t, (stages, T0, info) = T(coarse.coarse_align, eng, g, c, reps=3); print(f"coarse_align total: {t:.1f} ms")
t, _ = T(eng.set_source, c); print(f"set_source: {t:.2f} ms")
t, _ = T(eng.set_pair, g, c); print(f"set_pair: {t:.2f} ms")
t, _ = T(lambda: coarse.coarse_align(eng, g, c, resident=True), reps=3); print(f"coarse_align on the resident pair (device moments, sums-only scores): {t:.2f} ms")
t, r = T(lambda: eng.icp(50, 0.0, 0.10, init=T0)); print(f"icp 50 from the coarse start: {t:.2f} ms")
al = mapalign.Alignment(base_map_filepath=gp, comparison_map_filepath=sp)
t, _ = T(al.align_clouds, g, c, reps=3); print(f"align_clouds (coarse + gate schedule + final): {t:.1f} ms")
t, _ = T(lambda: mapalign.Alignment(base_map_filepath=gp, comparison_map_filepath=sp).compare(d), reps=3); print(f"compare(): {t:.1f} ms")
from concurrent.futures import ThreadPoolExecutor
def both():
    with ThreadPoolExecutor(max_workers=2) as pool:
        a, b = pool.submit(ply.read_ply, gp, dtype=None), pool.submit(ply.read_ply, sp, dtype=None)
        return a.result(), b.result()
t, _ = T(both); print(f"both .ply files in two threads: {t:.1f} ms")
import cProfile, pstats
al = mapalign.Alignment(base_map_filepath=gp, comparison_map_filepath=sp)
al.compare(d)
pr = cProfile.Profile(); pr.enable(); al.compare(d); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
