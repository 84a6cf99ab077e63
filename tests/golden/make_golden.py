"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference holds no golden vectors for this path ("parity unpinned"),
so these are outputs of the repo's own fp64 oracle (oracle/icp_oracle.py) on seeded
inputs, committed so that (a) the oracle cannot drift silently (tests/test_oracle.py
re-derives them on CPU) and (b) the HIP path is checked against fixed numbers on the GPU
box, where /root/reference and this generator's inputs need not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "3d-localisation-and-mapping_amd"))
sys.path.insert(0, ROOT)

from mapalign import synthetic as syn  # noqa: E402
from oracle import icp_oracle as orc  # noqa: E402


def pair_fixture(name, seed, n_gold, n_comp, gate, iters, angle_deg, t0, weights=False):
    gold, comp, R0, t = syn.make_pair(seed, n_gold, n_comp, angle_deg=angle_deg, t0=np.array(t0))
    w = None
    if weights:
        w = np.random.default_rng(seed + 100).uniform(0.25, 2.0, size=n_comp).astype(np.float32)
    res = orc.icp_oracle(gold, comp, max_iter=iters, tol=0.0, max_corr_dist=gate, weights=w, keep_nn=True,
                         n_jobs=1)
    acc = np.stack([r["acc"] for r in res["trace"]])
    rmse = np.array([r["rmse"] for r in res["trace"]])
    nn0 = res["trace"][0]["idx"].astype(np.int64)
    d20 = res["trace"][0]["d2"]
    out = dict(gold=gold, comp=comp, gate=np.float64(gate), iters=np.int32(iters), R_true=R0, t_true=t,
               R=res["R"], t=res["t"], acc=acc, rmse=rmse, nn0=nn0, d20=d20)
    if w is not None:
        out["weights"] = w
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "iters", res["iterations"], "rmse", res["rmse"], "inliers", res["inliers"],
          "err vs truth", orc.rt_error(res["R"], res["t"], R0, t))


def depth_fixture(name, seed):
    depth, K, T = syn.make_depth_frame(seed, width=96, height=64, fx=80.0, fy=80.0, cx=47.5, cy=31.5)
    rng = np.random.default_rng(seed)
    bad = rng.random(depth.shape)
    depth = depth.copy()
    depth[bad < 0.05] = 0.0
    depth[(bad >= 0.05) & (bad < 0.07)] = np.nan
    depth[(bad >= 0.07) & (bad < 0.09)] = np.inf
    depth[(bad >= 0.09) & (bad < 0.10)] = -5.0
    pts, mask = orc.unproject_oracle(depth, K, T, 1000.0)
    np.savez_compressed(os.path.join(HERE, name), depth=depth, K=np.array(K), T=T, scale=np.float64(1000.0),
                        points=pts, mask=mask)
    print(name, "valid", int(mask.sum()), "of", mask.size)


if __name__ == "__main__":
    pair_fixture("pair_a.npz", seed=11, n_gold=6000, n_comp=5000, gate=0.30, iters=25, angle_deg=1.5,
                 t0=(0.08, -0.05, 0.03))
    pair_fixture("pair_weighted.npz", seed=12, n_gold=3000, n_comp=3500, gate=0.40, iters=15, angle_deg=1.0,
                 t0=(0.05, 0.04, -0.03), weights=True)
    depth_fixture("depth_a.npz", seed=13)
