"""Throughput of a stream of 1M-point pairs on one GPU: pair by pair against the pipelined align_many (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-localisation-and-mapping_amd"))
import numpy as np, mapalign
from mapalign import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
base = [syn.make_pair(s, n)[:2] for s in range(3)]
pairs = [(base[k % 3][0].copy(), base[k % 3][1].copy()) for k in range(24)]      # distinct array objects: no pair re-uses a gold map
for workers in (1, 2, 3, 4, 1, 3):
    mapalign.align_many(pairs[:4], workers=workers)
    t0 = time.perf_counter()
    Ts = mapalign.align_many(pairs, workers=workers)
    dt = time.perf_counter() - t0
    print("workers=%d %d pairs of %d points: %.2f ms per pair (%.0f pairs/s, %.0f ICP iterations/s incl. upload and grid build)"
          % (workers, len(pairs), n, dt / len(pairs) * 1e3, len(pairs) / dt, 50 * len(pairs) / dt))

# one gold map, many scans (the same gold array object in every pair)
scans = [(base[0][0], base[k % 3][1]) for k in range(24)]
for workers in (1, 3):
    mapalign.align_many(scans[:4], workers=workers)
    t0 = time.perf_counter()
    mapalign.align_many(scans, workers=workers)
    dt = time.perf_counter() - t0
    print("one gold map, workers=%d: %.2f ms per scan (%.0f scans/s)" % (workers, dt / len(scans) * 1e3, len(scans) / dt))
