"""Sweep of the list planner's two settings (ma_list_config: skin = cell edge / skin_div, min_uses) on resident pairs.

    python3 tools/planner_sweep.py [points=1000000] [seeds=0,1]      (a seed "depth": configs[2], the depth frame against the gold map)

The settings change speed only (a list launch proves every correspondence it takes from a list), so every row must show the
same transform; what differs is when the lists start, how often they are rewritten and how long a list launch takes (a wider
skin = longer lists).  One process, one context per pair; per setting 3 warm runs, then 15 timed ones: device time of the loop
(min / median), host wall per call (median), the launch plan.  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
import mapalign  # noqa: E402
from mapalign import synthetic as syn  # noqa: E402

points = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seeds = [s if s == "depth" else int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]      # "depth": configs[2]
SKINS = tuple(float(s) for s in os.environ.get("SWEEP_SKINS", "4,5,6,7,8,10").split(","))
USES = tuple(int(s) for s in os.environ.get("SWEEP_USES", "2,3,5,8").split(","))
KIND = "SWL"

for seed in seeds:
    eng = mapalign.MapAligner(0)
    if seed == "depth":      # configs[2] as bench.py sets it up: the 1280 x 960 frame, pose 1 degree / 4.4 cm off, against the gold map of seed 0
        g = syn.make_pair(0, points)[0]
        depth, K, T = syn.make_depth_frame(0)
        dT = np.eye(4)
        dT[:3, :3] = syn.rotation_about_axis([0.2, 1.0, 0.3], np.deg2rad(1.0))
        dT[:3, 3] = [0.03, -0.02, 0.025]
        eng.set_gold(g, 0.0)
        eng.set_source_from_depth(depth, K, dT @ T, 1000.0)
    else:
        g, c, R0, t0 = syn.make_pair(seed, points)
        eng.set_gold(g, 0.0)
        eng.set_source(c)
    ref = None
    print(f"== {points}-point pair, seed {seed}, 50 iterations, gate 0.10 m (default: skin_div 7, min_uses 5)", flush=True)
    for sd in SKINS:
        for mu in USES:
            eng.list_config(enabled=True, skin_div=sd, min_uses=mu)
            for _ in range(3):
                r = eng.icp(50, 0.0, 0.1)
            dev, wall = [], []
            for _ in range(15):
                t_0 = time.perf_counter()
                r = eng.icp(50, 0.0, 0.1)
                wall.append((time.perf_counter() - t_0) * 1e3)
                dev.append(r.gpu_ms)
            plan = "".join(KIND[int(k)] for k in eng.list_plan(50))
            if ref is None:
                ref = r.T.copy()
            same = bool(np.array_equal(ref, r.T))
            dev.sort()
            wall.sort()
            print(f"skin_div {sd:4.1f} min_uses {mu}: loop ms min {dev[0]:.3f} med {dev[7]:.3f} | host wall med {wall[7]:.3f} | "
                  f"{plan} | same bits {same}", flush=True)
    eng.list_config(enabled=False)
    r = eng.icp(50, 0.0, 0.1)
    dev = sorted(eng.icp(50, 0.0, 0.1).gpu_ms for _ in range(9))
    print(f"lists off              : loop ms min {dev[0]:.3f} med {dev[4]:.3f} | same bits {bool(np.array_equal(ref, r.T))}", flush=True)
    eng.close()
