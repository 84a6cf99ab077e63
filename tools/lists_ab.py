"""A/B of the candidate lists (k_nn_accum list launches): every lib/libmapalign*.so, lists on and off in one process each.

    python3 tools/lists_ab.py [points]         (default 1000000)

Per build: loop time with lists on / off (MAPALIGN_LISTS=0), launches that wrote / read lists, the largest difference
between the two transforms, per-iteration launch times of the search kernel with lists on."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libdir = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib")
n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
names = [""] + sorted(f[len("libmapalign_"):-3] for f in os.listdir(libdir) if f.startswith("libmapalign_") and f.endswith(".so"))
code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, %s)
eng = mapalign.MapAligner(0); eng.set_gold(g, 0.0); eng.set_source(c)
out = {}
for mode in ("1", "0"):
    os.environ["MAPALIGN_LISTS"] = mode
    for _ in range(3): r = eng.icp(50, 0.0, 0.1)
    ts = sorted(eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(7))
    r = eng.icp(50, 0.0, 0.1)
    st = eng.list_stats()
    rt = eng.icp(50, 0.0, 0.1, time_kernels=True)
    it = eng.iteration_times()
    out[mode] = (ts, r.T.copy(), st, it, r.inliers, r.rmse)
on, off = out["1"], out["0"]
print("%%-8s lists on: min %%.3f med %%.3f ms (%%d wrote, %%d read) | off: min %%.3f med %%.3f ms | max|dT| %%.2e | inliers %%.0f/%%.0f rmse diff %%.1e | errR %%.2e" %% (
    os.environ.get("VAR", "default"), on[0][0], on[0][3], on[2]["builds"], on[2]["uses"], off[0][0], off[0][3],
    np.abs(on[1] - off[1]).max(), on[4], off[4], abs(on[5] - off[5]), np.linalg.norm(on[1][:3, :3] - R0)), flush=True)
print("   nn us/iteration, lists on : " + " ".join("%%.0f" %% (v * 1e3) for v in on[3][:, 0]))
print("   nn us/iteration, lists off: " + " ".join("%%.0f" %% (v * 1e3) for v in off[3][:, 0]))
print("   solve us mean on %%.2f off %%.2f" %% (on[3][:, 1].mean() * 1e3, off[3][:, 1].mean() * 1e3), flush=True)
''' % (ROOT, n)
skins = [a for a in sys.argv[2:]]          # optional: skin divisors to sweep with the default build (MAPALIGN_SKIN_DIV)
for rep in range(2):
    for nm in names:
        env = dict(os.environ, VAR=nm or "default")
        if nm:
            env["MAPALIGN_LIB"] = os.path.join(libdir, f"libmapalign_{nm}.so")
        subprocess.run([sys.executable, "-c", code], env=env, check=False, timeout=300)
    for sk in skins:
        env = dict(os.environ, VAR=f"skin/{sk}", MAPALIGN_SKIN_DIV=sk)
        subprocess.run([sys.executable, "-c", code], env=env, check=False, timeout=300)
