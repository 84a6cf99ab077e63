"""A/B of whole-loop time and per-iteration launch times between lib/libmapalign*.so builds, one process each.

    python3 tools/r4_ab.py [points] [iterations] [--skins 5,6,7] [--uses 2,3,5]

Per build: loop time (min / median of 7), launches that wrote / read the lists, per-iteration search-launch times with
lists on, the same loop with every launch searching (lists=False where the build knows the flag, MAPALIGN_LISTS=0 at
context creation otherwise), and whether both give the same bits.  --skins / --uses sweep the default build's list
configuration (ma_list_config)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libdir = os.path.join(ROOT, "3d-localisation-and-mapping_amd", "lib")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = args[0] if len(args) > 0 else "1000000"
iters = args[1] if len(args) > 1 else "50"
opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
names = [""] + sorted(f[len("libmapalign_"):-3] for f in os.listdir(libdir) if f.startswith("libmapalign_") and f.endswith(".so"))
code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
g, c, R0, t0 = syn.make_pair(0, %s)
IT = %s
old = os.environ.get("VAR", "") == "r3"
def run(lists):
    if old:
        os.environ["MAPALIGN_LISTS"] = "1" if lists else "0"
    eng = mapalign.MapAligner(0); eng.set_gold(g, 0.0); eng.set_source(c)
    sk, mu = os.environ.get("SKIN"), os.environ.get("USES")
    if not old and (sk or mu):
        eng.list_config(None, float(sk) if sk else None, int(mu) if mu else None)
    kw = {} if old else {"lists": lists}
    for _ in range(3): r = eng.icp(IT, 0.0, 0.1, **kw)
    ts = sorted(eng.icp(IT, 0.0, 0.1, **kw).gpu_ms for _ in range(9))
    r = eng.icp(IT, 0.0, 0.1, with_trace=True, **kw)
    st = eng.list_stats(); plan = "".join("SWL"[int(v)] for v in eng.list_plan())
    eng.icp(IT, 0.0, 0.1, time_kernels=True, **kw)
    it = eng.iteration_times()
    eng.close()
    return ts, r, st, it, plan
on, off = run(True), run(False)
same = np.array_equal(on[1].T, off[1].T) and np.array_equal(on[1].trace[:, :18], off[1].trace[:, :18])
print("%%-10s lists on: min %%.3f med %%.3f ms (%%d wrote, %%d read) | off: min %%.3f med %%.3f ms | on == off bitwise: %%s | errR %%.2e" %% (
    os.environ.get("VAR", "default") + os.environ.get("TAG", ""), on[0][0], on[0][4], on[2]["builds"], on[2]["uses"], off[0][0], off[0][4], same,
    np.linalg.norm(on[1].T[:3, :3] - R0)), flush=True)
print("   plan " + on[4])
print("   nn us/iteration, lists on : " + " ".join("%%.0f" %% (v * 1e3) for v in on[3][:, 0]))
kinds = np.array(list(on[4][:len(on[3])]))
for kd in "SWL":
    v = on[3][kinds == kd, 0] * 1e3
    if len(v): print("     %%s: %%d launches, mean %%.1f us (hipEvents: +~2 us each)" %% (kd, len(v), v.mean()))
print("   solve us mean on %%.2f off %%.2f" %% (on[3][:, 1].mean() * 1e3, off[3][:, 1].mean() * 1e3), flush=True)
''' % (ROOT, n, iters)
def go(env):
    subprocess.run([sys.executable, "-c", code], env=env, check=False, timeout=600)
for rep in range(int(opt.get("--reps", "2"))):
    for nm in names:
        env = dict(os.environ, VAR=nm or "default")
        if nm:
            env["MAPALIGN_LIB"] = os.path.join(libdir, f"libmapalign_{nm}.so")
        go(env)
for sk in [s for s in opt.get("--skins", "").split(",") if s]:
    for mu in [u for u in opt.get("--uses", "").split(",") if u] or [""]:
        go(dict(os.environ, VAR="default", TAG=f" skin/{sk} uses>={mu or 'dflt'}", SKIN=sk, USES=mu))
