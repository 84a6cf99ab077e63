"""Host wall of one eng.icp(50) call against the device time of its loop (hipEvents around the launches): what the
per-call setup and read-back cost (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn
g, c, _, _ = syn.make_pair(0, int(os.environ.get("N", 1_000_000)))
eng = mapalign.MapAligner(0); eng.set_gold(g); eng.set_source(c)
for _ in range(5): eng.icp(50, 0.0, 0.1)
wall, dev = [], []
t_all = time.perf_counter()
for _ in range(40):
    t = time.perf_counter(); r = eng.icp(50, 0.0, 0.1); wall.append((time.perf_counter() - t) * 1e3); dev.append(r.gpu_ms)
t_all = (time.perf_counter() - t_all) / 40 * 1e3
wall, dev = np.array(wall), np.array(dev)
print(f"icp(50): wall median {np.median(wall):.3f} ms, device loop median {np.median(dev):.3f} ms, overhead {1e3*np.median(wall - dev):.0f} us per call; "
      f"back-to-back {t_all:.3f} ms per call -> {50/t_all*1e3:.0f} it/s")
