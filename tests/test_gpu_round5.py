"""GPU parity tests added in round 5 (run with -m gpu on an MI355X), all through the C ABI:

* BASELINE configs[2] at EQUAL iteration count: the full 1280 x 960 depth frame (holes included) unprojected in place
  (``ma_set_source_from_depth`` without a count: the alignment is enqueued right behind it) and aligned against the
  1M-point gold map for 50 iterations, against the oracle fed the oracle's own unprojection of the same frame —
  per-iteration inlier counts and squared-distance sums, final (R, t), and candidate lists on = off bit for bit on this
  pixel-ordered, hole-carrying source (pose_processor.py:184-196,242-260, dataset.py:68-81 conventions);
* the count of valid pixels of a frame too large for the pinned count block (> 16.7 MP) decides what it decides for
  every other frame.
"""
import time

import numpy as np
import pytest

import mapalign
from mapalign import synthetic as syn
from oracle import icp_oracle as orc

from conftest import R_TOL_FRO, T_TOL_M, record_comparison, suite_seconds_left

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- configs[2] at equal iteration count
def test_config3_full_frame_50_iterations_vs_oracle_and_lists_on_off(engine):
    depth, K, T = syn.make_depth_frame(0)                     # 1280 x 960 float32 millimetres, fx = fy = 1000
    gold, _, _, _ = syn.make_pair(0, 1_000_000)
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.1, 0.9, 0.2], np.deg2rad(0.4))
    dT[:3, 3] = [0.025, -0.015, 0.02]
    pose = dT @ T
    opts, mask = orc.unproject_oracle(depth, K, pose, 1000.0)
    n_valid = int(mask.sum())
    assert 1_100_000 < n_valid < depth.size                   # the frame has holes (rays that leave the carriage)
    engine.set_gold(gold)
    full = 50
    # the oracle's 50 iterations on ITS OWN unprojection of the frame (fp64 points, valid pixels in row-major order)
    index = orc.GoldIndex(gold, n_jobs=-1, backend="ckdtree")
    R, t = np.eye(3), np.zeros(3)
    inliers, E = [], []
    t_start = time.perf_counter()
    iters = full
    k = 0
    while k < iters:
        acc, _, _, R, t, ok = orc.icp_step(index, opts, R, t, 0.10)
        assert ok
        inliers.append(acc[0])
        E.append(acc[16])
        k += 1
        if k == 4:      # never below 34 (past the first list launches), all 50 unless the suite is out of time
            per_it = (time.perf_counter() - t_start) / 4
            iters = int(min(full, max(34, k + (suite_seconds_left() - 150.0) / per_it)))
    record_comparison("config3", iters, full, "ckdtree", (time.perf_counter() - t_start) / iters)
    if iters != full:
        import warnings
        warnings.warn(f"config3: compared after {iters} iterations — SHORTENED from {full} by the suite's wall-clock budget")
    # the device: unprojection enqueued without a count, the loop right behind it
    assert engine.set_source_from_depth(depth, K, pose, 1000.0, count=False) is None
    res = engine.icp(iters, 0.0, 0.10, with_trace=True)
    assert res.iterations == iters and engine.n_source == n_valid
    st = engine.list_stats()
    assert st["builds"] >= 1 and st["uses"] >= 3, st           # the list launches of this source are part of the comparison
    eR, et = orc.rt_error(res.R, res.t, R, t)
    print(f"config3: full frame ({n_valid} valid pixels) vs 1M gold compared after {iters} iterations: |dR|_F = {eR:.2e}, "
          f"|dt| = {et:.2e} m; lists {st}")
    assert eR <= R_TOL_FRO and et <= T_TOL_M, (eR, et)
    assert eR <= 1e-5 and et <= 1e-4, (eR, et)
    # the same correspondences every iteration: counts agree to the handful of points within the fp32 storage error
    # (~2 um at 18 m range) of the gate; E follows (a flip moves it by gate^2)
    gpu_inl = res.trace[:, 0]
    d_inl = np.abs(gpu_inl - np.array(inliers))
    assert d_inl.max() <= 40, d_inl.max()
    E_gpu, E_cpu = res.trace[:, 16], np.array(E)
    assert (np.abs(E_gpu - E_cpu) <= (d_inl + 1.0) * 0.10 ** 2 + 2e-4 * E_cpu).all(), np.abs(E_gpu - E_cpu).max()
    # lists on = off, bit for bit, on the pixel-ordered source with holes
    off = engine.icp(iters, 0.0, 0.10, with_trace=True, lists=False)
    assert engine.list_stats() == {"builds": 0, "uses": 0}
    assert np.array_equal(off.T, res.T) and np.array_equal(off.trace[:, :18], res.trace[:, :18])
    # and the perturbation of the camera pose is undone (the frame samples the gold surfaces)
    if iters == full:
        err = np.linalg.inv(dT) @ np.linalg.inv(res.T)
        assert np.linalg.norm(err[:3, :3] - np.eye(3)) <= 2e-3 and np.linalg.norm(err[:3, 3]) <= 1e-2


# ----------------------------------------------------------------------------- frames above the pinned count block
def test_count_of_a_frame_above_16_7_megapixels_settles_the_source(engine):
    """a frame of more than 32 768 workgroups of counts (> 16.7 MP) takes the copied-counts path of
    unproject_frame; the count must still decide n_source, the holes and MA_ERR_EMPTY."""
    gold, _, _, _ = syn.make_pair(1, 50_000)
    engine.set_gold(gold)
    w, h = 4608, 3712                                          # 17.1 MP
    rng = np.random.default_rng(3)
    depth = np.zeros((h, w), dtype=np.uint16)
    ys, xs = rng.integers(0, h, 5000), rng.integers(0, w, 5000)
    depth[ys, xs] = rng.integers(500, 4000, 5000).astype(np.uint16)
    n_valid = int((depth > 0).sum())
    K = (3000.0, 3000.0, w / 2.0, h / 2.0)
    T = np.eye(4)
    assert engine.set_source_from_depth(depth, K, T, 1000.0, count=False) is None
    assert engine.n_source == n_valid                          # settled by the count, not "all pixels"
    pts = engine.get_source()
    opts, _ = orc.unproject_oracle(depth, K, T, 1000.0)
    assert pts.shape == (n_valid, 3) and np.abs(pts - opts).max() <= 2e-6
    assert engine.set_source_from_depth(depth, K, T, 1000.0) == n_valid
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.set_source_from_depth(np.zeros((h, w), dtype=np.uint16), K, T, 1000.0, count=False)
    assert e.value.status == mapalign._lib.MA_ERR_EMPTY
