"""Map-builder timing (row N4): N full-size depth frames -> one voxel-deduplicated map -> gold map (GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign, numpy as np
from mapalign import synthetic as syn
depth, K, T = syn.make_depth_frame(0)                      # 1280 x 960
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
poses = []
for k in range(frames):                                    # the same frame seen from a walk down the carriage
    P = T.copy(); P[:3, 3] += [0.5 * k, 0.02 * (k % 3), 0.0]
    poses.append(P)
eng = mapalign.MapAligner(0)
for rep in range(3):
    eng.map_begin(0.02, frames * depth.size)
    t0 = time.perf_counter()
    tot = 0
    for P in poses:
        tot += eng.map_add_depth(depth, K, P, 1000.0)
    eng._chk(eng._lib.ma_sync(eng._ctx)) if hasattr(eng._lib, "ma_sync") else None
    t1 = time.perf_counter()
    n = eng.map_finish(as_gold=True, fetch=False)
    t2 = time.perf_counter()
    print("rep %d: %d frames, %d valid pixels -> %d voxels of 2 cm | add_depth %.2f ms/frame (host wall, incl. 4.9 MB upload) | finish + gold grid %.1f ms"
          % (rep, frames, tot, n, (t1 - t0) / frames * 1e3, (t2 - t1) * 1e3))
# the same frames in ONE call (ma_map_add_depth_batch), host-resident and device-resident
import torch
stack = np.repeat(depth[None], frames, axis=0)
stack_dev = torch.as_tensor(stack).to("cuda:0")
torch.cuda.synchronize()
import ctypes as C
for name, arr in (("host frames", stack), ("device-resident frames", stack_dev)):
    for rep in range(3):
        eng.map_begin(0.02, frames * depth.size)
        t0 = time.perf_counter()
        tot = int(eng.map_add_depth_batch(arr, K, poses, 1000.0).sum())
        t1 = time.perf_counter()
        ms = C.c_double(0)
        eng._chk(eng._lib.ma_last_device_ms(eng._ctx, C.byref(ms)))
        n = eng.map_finish(as_gold=True, fetch=False)
        t2 = time.perf_counter()
        nbytes = frames * depth.size * (4 + 16)
        print("batch (%s) rep %d: %d frames, %d valid pixels -> %d voxels | one call %.2f ms host wall | k_unproject launch %.1f us = %.2f TB/s = %.2f of HBM peak | finish + gold grid %.1f ms"
              % (name, rep, frames, tot, n, (t1 - t0) * 1e3, ms.value * 1e3, nbytes / ms.value / 1e9, nbytes / ms.value / 1e9 / 8.0, (t2 - t1) * 1e3))
