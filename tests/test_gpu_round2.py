"""GPU parity tests added in round 2 (run with -m gpu on an MI355X), all through the C ABI:

* the headline configuration (BASELINE configs[1]: 1M-point pair, 50 iterations) against the
  CPU oracle at EQUAL iteration count, per-iteration inlier counts included, seeds 0-2;
* the schedule ranking with more than 4.2M source points (group count not a multiple of 4);
* invalidation of the stepping state, warm contexts allocate nothing, RCCL inside the library
  with a 1-rank communicator, the batched box lift and the batch-32 detector leg (configs[3]),
  trajectory-aided coarse stage, weights that do not sum to a count.
"""
import os
import time

import numpy as np
import pytest

import mapalign
from mapalign import _lib, coarse, synthetic as syn, transforms
from oracle import icp_oracle as orc

from conftest import R_TOL_FRO, T_TOL_M

pytestmark = pytest.mark.gpu

NO_MATCH = 0xFFFFFFFF


# ----------------------------------------------------------------------------- headline parity
# what the oracle-heavy tests that run AFTER the headline test need (tests/test_gpu_round4.py: two 2M-point pairs of >= 12
# oracle iterations each, the sharding and planner tests), seconds
RESERVE_AFTER_HEADLINE_S = 240.0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_config2_1m_50_iterations_vs_oracle(engine, seed):
    """north_star: "(R,t) must match the reference CPU path to <= 1e-4 Frobenius on R and
    <= 1 mm on t for identical inputs" — the 1M-point pair, all 50 iterations, tol = 0, gate
    0.10 m.  The oracle runs its 50 KD-tree iterations on the host cores
    (1.2-4 s each on the 256-core GPU boxes, depending on their load).  Seed 0 runs on the scikit-learn backend
    north_star names: all 50 iterations where the host allows (80 s on most boxes), and on a slow host (3.8 s per iteration
    seen: 190 s, which took the later seeds' time) at least the first 10, the rest from the same state on the oracle's
    cKDTree backend — the same nearest neighbours (tests/test_oracle.py), what bench.py's parity leg does — so that all 50
    are still compared.  Seeds 1 and 2 run on cKDTree (4-7 times faster); should a host be slower still, the budget
    (conftest.suite_seconds_left) shortens them — seed 1 never below 34, past the first list launches of this pair,
    seed 2 never below 20 — with a warning, and the terminal summary says what ran."""
    from conftest import record_comparison, suite_seconds_left
    gold, comp, R0, t0 = syn.make_pair(seed, 1_000_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    backend = "sklearn" if seed == 0 else "ckdtree"
    index = orc.GoldIndex(gold, n_jobs=-1, backend=backend)
    R, t = np.eye(3), np.zeros(3)
    inliers, rmse = [], []
    iters = 50
    n_sklearn = 50      # seed 0: iterations on the scikit-learn backend
    t_start = time.perf_counter()
    k = 0
    while k < iters:
        acc, _, _, R, t, ok = orc.icp_step(index, comp, R, t, 0.10)
        assert ok
        inliers.append(acc[0])
        rmse.append(np.sqrt(acc[16] / acc[0]))
        k += 1
        if seed == 0 and k == 5:
            # does the scikit-learn backend fit all 50?  Otherwise it keeps what fits (>= 10), and cKDTree takes over
            per_it = (time.perf_counter() - t_start) / 5
            room = suite_seconds_left() - RESERVE_AFTER_HEADLINE_S - 40.0      # 40 s: seeds 1 and 2 in full on cKDTree
            if (50 - k) * per_it > room:
                n_sklearn = int(max(10, min(50, k + (room - (50 - k) * 0.3) / per_it)))
        if seed == 0 and k == n_sklearn and k < iters:
            index = orc.GoldIndex(gold, n_jobs=-1, backend="ckdtree")
            backend = f"sklearn 0-{k - 1}, ckdtree {k}-49"
        if seed > 0 and k == 5:
            per_it = (time.perf_counter() - t_start) / 5
            # what is left for this seed: the budget minus the reserve of the later tests, minus (for seed 1) seed 2's floor
            floor = 34 if seed == 1 else 20        # seed 1 always reaches the list launches; seed 2 gives way first on a slow host
            room = suite_seconds_left() - RESERVE_AFTER_HEADLINE_S - (20 * per_it if seed == 1 else 0.0)
            iters = int(min(50, max(floor, k + room / per_it)))
    res = engine.icp(iters, 0.0, 0.10, with_trace=True)
    assert res.iterations == iters
    eR, et = orc.rt_error(res.R, res.t, R, t)
    # (which comparison actually ran goes into the log: pytest -s / -rP shows it, and the warning is always listed)
    msg = (f"config2 seed {seed}: compared after {iters} iterations ({(time.perf_counter() - t_start) / iters:.2f} s per oracle "
           f"iteration on {len(os.sched_getaffinity(0))} host cores): |dR|_F = {eR:.2e}, |dt| = {et:.2e} m")
    print(msg)
    record_comparison(f"config2[{seed}]", iters, 50, backend, (time.perf_counter() - t_start) / iters)
    if iters != 50:
        import warnings
        warnings.warn(msg + " — SHORTENED from 50: the suite's wall-clock budget has no room for the full oracle run of this seed")
    assert eR <= R_TOL_FRO and et <= T_TOL_M, (eR, et)
    # measured: 1e-7 / 1e-6 m; keep a tighter regression bar than the contract
    assert eR <= 1e-5 and et <= 1e-4, (eR, et)
    # the same correspondences every iteration: inlier counts agree to a handful of points
    # whose distance sits within the fp32 storage error (~1 um) of the gate (2e-5 of the cloud)
    gpu_inl = res.trace[:, 0]
    assert np.abs(gpu_inl - np.array(inliers)).max() <= 20, np.abs(gpu_inl - np.array(inliers)).max()
    # E = sum of squared distances: a point that flips across the gate moves it by gate^2, and
    # the ~1e-8 m difference in the transform that such a flip leaves behind moves E smoothly
    # (dE/dT ~ 2 sum d_i ~ 1e5 m per metre in the early iterations)
    E_gpu, E_cpu = res.trace[:, 16], np.array(rmse) ** 2 * np.array(inliers)
    slack = (np.abs(gpu_inl - np.array(inliers)) + 1.0) * 0.10 ** 2 + 2e-4 * E_cpu
    assert (np.abs(E_gpu - E_cpu) <= slack).all(), np.abs(E_gpu - E_cpu).max()
    if iters == 50:   # and the result is the ground truth of the synthetic pair
        eR0, et0 = orc.rt_error(res.R, res.t, R0, t0)
        assert eR0 <= R_TOL_FRO and et0 <= T_TOL_M


# ----------------------------------------------------------------------------- schedule, gshift >= 2
def test_5m_source_points_group_count_not_multiple_of_4(engine):
    """above ~4.19M source points the ranking workgroup orders 2^gshift-segment
    groups whose number is any integer (306 at 5M); the trailing count % 4 keys were never
    compared, two groups dropped out of the schedule and their chunks kept stale partials.
    The scheduled loop (ma_icp) must equal the unscheduled single steps (ma_icp_step) bit for
    bit, and iteration 0's sums must equal direct sums over the matched points."""
    n = 5_000_000
    gold, comp, R0, t0 = syn.make_pair(9, 600_000, n)
    engine.set_gold(gold)
    engine.set_source(comp)
    iters = 6
    res = engine.icp(iters, 0.0, 0.10, with_trace=True)
    # the same iterations through the stepping entry points: no schedule (chunks in cloud
    # order), same remembered matches, same fixed-order sums -> bit-identical state
    import torch
    acc = torch.zeros(17, dtype=torch.int64, device="cuda:0")      # the 17 fixed-point sums
    torch.cuda.synchronize()
    engine.set_sum_bounds(engine.sum_bounds())                      # a lone context: its own bounds (what ma_icp used)
    engine.state_set(None)
    for k in range(iters):
        engine.step_accumulate(acc.data_ptr(), 0.10)
        engine.step_solve(acc.data_ptr(), 0.0, 0.10)
        engine.sync()
        sums = engine.sums_to_float(acc.cpu().numpy(), 0.10)
        assert sums[0] == res.trace[k, 0], f"iteration {k}: scheduled and unscheduled inlier counts differ"
        assert sums[16] == res.trace[k, 16]                         # E: frame-independent, the same integer
    T, info = engine.state_get()
    assert info["iterations"] == iters and np.array_equal(T, res.T)
    # single steps from scratch (nothing remembered): the same correspondences up to
    # fp32 near-ties, so the sums agree to rounding and the counts to a few points
    T1 = np.eye(4)
    for k in range(3):
        a1, T1 = engine.icp_step(T1, 0.10)
        assert abs(a1[0] - res.trace[k, 0]) <= 3 and np.allclose(a1, res.trace[k, :17], rtol=1e-6)
    again = engine.icp(iters, 0.0, 0.10, with_trace=True)
    assert np.array_equal(again.trace, res.trace)
    idx, d2 = engine.nn(None, 0.10)
    m = idx != NO_MATCH
    assert res.trace[0, 0] == m.sum()
    assert np.allclose(res.trace[0, 1:4], comp[m].astype(np.float64).sum(0), rtol=1e-9)
    assert np.allclose(res.trace[0, 4:7], gold[idx[m]].astype(np.float64).sum(0), rtol=1e-9)
    # 60 iterations from the truth stay at the truth (every chunk contributes every iteration)
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = R0, t0
    far = engine.icp(20, 0.0, 0.10, init=T0)
    eR, et = orc.rt_error(far.R, far.t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M and far.inliers > 0.99 * n


# ----------------------------------------------------------------------------- stepping state
def test_stepping_state_is_invalidated_by_new_clouds(lib_built):
    import torch
    gold, comp, _, _ = syn.make_pair(4, 20_000, 30_000)
    acc = torch.zeros(17, dtype=torch.int64, device="cuda:0")
    with mapalign.MapAligner(0) as eng:
        eng.set_gold(gold)
        eng.set_source(comp[:5000])
        with pytest.raises(mapalign.MapAlignError) as e:          # split stepping needs agreed sum bounds
            eng.state_set(None)
            eng.step_accumulate(acc.data_ptr(), 0.10)
        assert e.value.status == _lib.MA_ERR_STATE
        eng.set_sum_bounds([30_000, 1.0, eng.sum_bounds()[2]])    # the whole cloud's: both sizes below scale alike
        eng.state_set(None)
        eng.step_accumulate(acc.data_ptr(), 0.10)
        eng.step_solve(acc.data_ptr(), 0.0, 0.10)
        eng.sync()
        w_small = float(acc[0])
        # a LARGER cloud after ma_state_set: partials / prev were sized for 5000 points
        eng.set_source(comp)
        eng.set_sum_bounds([30_000, 1.0, eng.sum_bounds()[2]])    # (a new source cloud clears the bounds)
        with pytest.raises(mapalign.MapAlignError) as e:
            eng.step_accumulate(acc.data_ptr(), 0.10)
        assert e.value.status == _lib.MA_ERR_STATE
        with pytest.raises(mapalign.MapAlignError):
            eng.step_solve(acc.data_ptr(), 0.0, 0.10)
        eng.state_set(None)
        eng.step_accumulate(acc.data_ptr(), 0.10)
        eng.sync()
        assert float(acc[0]) > w_small
        # a new gold map invalidates as well (prev[] holds positions in the old map)
        eng.set_gold(gold[:7000])
        with pytest.raises(mapalign.MapAlignError) as e:
            eng.step_accumulate(acc.data_ptr(), 0.10)
        assert e.value.status == _lib.MA_ERR_STATE
        # the one-shot entry points are unaffected
        eng.set_source(comp)
        assert eng.icp(5, 0.0, 0.10).iterations == 5


def test_warm_context_allocates_nothing():
    """align() keeps one context per device: the second call of the same size must perform zero
    device allocations, and GoldMap keeps the indexed gold map resident."""
    gold, comp, R0, t0 = syn.make_pair(6, 200_000)
    mapalign.release_cached_engines()
    R1, t1, i1 = mapalign.align(gold, comp, max_iter=30, return_info=True)
    R2, t2, i2 = mapalign.align(gold, comp, max_iter=30, return_info=True)
    assert i1["mem"]["device_allocs"] > 0
    assert i2["mem"]["device_allocs"] == i1["mem"]["device_allocs"] and i2["mem"]["device_frees"] == i1["mem"]["device_frees"]
    assert np.array_equal(R1, R2) and np.array_equal(t1, t2)
    # a smaller scan against the same gold map reuses the buffers as well
    _, _, i3 = mapalign.align(gold, comp[:150_000], max_iter=5, return_info=True)
    assert i3["mem"]["device_allocs"] == i1["mem"]["device_allocs"]
    with mapalign.GoldMap(gold) as gm:
        Ra, ta = mapalign.align(gm, comp, max_iter=30)
        before = gm.engine.mem_info()["device_allocs"]
        Rb, tb = mapalign.align(gm, comp[::-1].copy(), max_iter=30)
        assert gm.engine.mem_info()["device_allocs"] == before
        assert np.array_equal(Ra, R1) and np.allclose(Rb, R1, atol=1e-9) and np.allclose(tb, t1, atol=1e-9)
    mapalign.release_cached_engines()


def test_tol_early_exit_batches_match_the_oracle(engine):
    """tol > 0 with a large max_iter: the host stops enqueueing once the device reports `done`
    (batches of 8 iterations); the result is the oracle's early stop, not max_iter iterations."""
    gold, comp, _, _ = syn.make_pair(8, 30_000)
    # tol well above the ~1e-8 m rmse jitter that single points flipping across the gate cause,
    # so that the stopping iteration is a property of the convergence curve, not of rounding
    ref = orc.icp_oracle(gold, comp, max_iter=400, tol=1e-6, max_corr_dist=0.10, n_jobs=1)
    assert ref["converged"] and 24 < ref["iterations"] < 200      # past the first host-side check (iteration 16)
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(400, 1e-6, 0.10, with_trace=True)
    assert res.converged and abs(res.iterations - ref["iterations"]) <= 1
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= 1e-5 and et <= 1e-5
    # and the context is immediately reusable
    assert engine.icp(3, 0.0, 0.10).iterations == 3


def test_weights_that_do_not_sum_to_a_count(engine):
    """W is a sum of weights, not a correspondence count.  Normalised weights
    (sum = 1) over 50k matches must align, exactly like unit weights."""
    gold, comp, R0, t0 = syn.make_pair(12, 50_000, resample=False)
    engine.set_gold(gold)
    engine.set_source(comp, np.full(len(comp), 1.0 / len(comp), np.float32))
    res = engine.icp(40, 0.0, 0.10)
    eR, et = orc.rt_error(res.R, res.t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M and abs(res.inliers - 1.0) < 1e-3
    # two matches only (weights zero elsewhere): rank-deficient -> MA_ERR_DEGENERATE
    w = np.zeros(len(comp), np.float32)
    w[:2] = 1.0
    engine.set_source(comp, w)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.icp(5, 0.0, np.inf)
    assert e.value.status == _lib.MA_ERR_DEGENERATE


# ----------------------------------------------------------------------------- RCCL inside the library
def test_icp_sharded_one_rank_communicator_equals_icp_bitwise(lib_built):
    """ma_icp_sharded with a 1-rank RCCL communicator: librccl is loaded (dlopen), every
    iteration runs search -> 19-way sum -> ncclAllReduce -> solve from C on the ctx stream, and the
    result equals ma_icp bit for bit (same sums, same order)."""
    gold, comp, R0, t0 = syn.make_pair(17, 120_000, 100_003)
    with mapalign.MapAligner(0) as eng:
        eng.set_gold(gold)
        eng.set_source(comp)
        ref = eng.icp(25, 0.0, 0.10, with_trace=True)
        with pytest.raises(mapalign.MapAlignError) as e:     # no communicator yet
            eng.icp_sharded(5)
        assert e.value.status == _lib.MA_ERR_STATE
        uid = eng.comm_unique_id()
        assert len(uid) == _lib.MA_COMM_ID_BYTES and any(uid)
        eng.comm_init(uid, 1, 0)
        res = eng.icp_sharded(25, 0.0, 0.10, with_trace=True)
        assert np.array_equal(res.T, ref.T) and np.array_equal(res.trace[:, :18], ref.trace[:, :18])
        assert res.iterations == 25 and res.inliers == ref.inliers
        # with a convergence test the solve stops the state early while every rank keeps issuing
        # its max_iter all-reduces (collective counts must match across ranks): same result as ma_icp
        ref_tol = eng.icp(60, 1e-5, 0.10)
        res_tol = eng.icp_sharded(60, 1e-5, 0.10)
        assert ref_tol.converged and res_tol.converged and res_tol.iterations == ref_tol.iterations < 60
        assert np.array_equal(res_tol.T, ref_tol.T)
        # small host vectors through the same communicator (pair-mode gather, fleet statistics)
        g = eng.comm_allgather(np.arange(5.0))
        assert g.shape == (1, 5) and np.array_equal(g[0], np.arange(5.0))
        assert np.array_equal(eng.comm_allreduce([1.5, -2.0], "sum"), [1.5, -2.0])
        assert np.array_equal(eng.comm_allreduce([1.5, -2.0], "max"), [1.5, -2.0])
        eng.comm_destroy()
    from mapalign import dist as mdist
    R, t, info = mdist.sharded_align(gold, comp, max_iter=25, device=0, return_info=True)
    assert np.array_equal(R, ref.R) and np.array_equal(t, ref.t) and info["ranks"] == 1


# ----------------------------------------------------------------------------- a10, depth rule 1 (median over the box)
def test_lift_boxes_median_depth_rule_vs_oracle(engine):
    """MA_Z_MEDIAN — the reading of the comment at pose_processor.py:183 ("z-values from median over bbox (x, y)
    range"): one z per box = median of the valid depths over its pixel rectangle.  k_box_median (radix select,
    one workgroup per box) against numpy.median in the oracle: boxes with holes, boxes clamped into the image,
    one-pixel boxes, even and odd numbers of valid pixels, a box without any valid pixel, float32 and uint16
    frames, the per-frame call and the batched one, and the default rule unchanged beside it."""
    from mapalign import lifting
    rng = np.random.default_rng(21)
    w, h = 192, 256
    depth = rng.uniform(500.0, 9000.0, size=(h, w)).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0                       # holes everywhere
    depth[40:60, 50:80] = 0.0                                   # a block without a single valid pixel
    depth[5, 7] = np.nan
    depth[100:103, 20:22] = [[1000.0, 2000.0], [3000.0, 0.0], [4000.0, 5000.0]]     # a 3 x 2 patch with 5 valid pixels
    K = (150.0, 151.0, 95.5, 127.5)
    T = syn.pose_to_matrix([1.0, -2.0, 0.5, 0.1, -0.2, 0.3, 0.9])
    x1 = rng.uniform(-10, 170, 60); y1 = rng.uniform(-10, 230, 60)
    boxes = np.stack([x1, y1, x1 + rng.uniform(0, 90, 60), y1 + rng.uniform(0, 90, 60)], 1).astype(np.float32)
    boxes[0] = [50.2, 40.7, 79.9, 59.1]                         # inside the invalid block: no valid pixel -> four invalid corners
    boxes[1] = [20.0, 100.0, 21.9, 102.9]                       # the 3 x 2 patch: 5 valid (odd) -> the middle one, 3000
    boxes[2] = [20.0, 100.0, 21.9, 101.9]                       # its top 2 x 2: 3 valid
    boxes[3] = [20.0, 100.0, 20.9, 102.5]                       # left column: 1000, 3000, 4000
    boxes[4] = [20.0, 100.0, 21.5, 100.5]                       # top row: 1000, 2000 (even) -> 1500
    boxes[5] = [30.3, 30.8, 30.9, 30.9]                         # one pixel
    boxes[6] = [-50.0, -50.0, 900.0, 900.0]                     # the whole image (clamped): 49 152 pixels
    boxes[7] = [100.0, 200.0, 60.0, 150.0]                      # corners given the other way round
    depth[30, 30] = 1234.0
    oc, ov = orc.lift_boxes_oracle(boxes, depth, K, T, 1000.0, z_rule=1)
    assert not ov[0].any() and ov[1].all()
    cam1 = np.linalg.inv(T) @ np.append(oc[1, 0], 1.0)
    assert abs(cam1[2] - 3.0) < 1e-12                           # the oracle itself: median of {1,2,3,4,5} m
    cam4 = np.linalg.inv(T) @ np.append(oc[4, 0], 1.0)
    assert abs(cam4[2] - 1.5) < 1e-12                           # even count: mean of the two middle values
    corners, valid = engine.lift_boxes(boxes, K, T, depth, 1000.0, z_rule=lifting.Z_MEDIAN)
    assert np.array_equal(valid, ov)
    assert np.allclose(corners[valid], oc[ov], atol=1e-9) and np.isnan(corners[~valid]).all()
    # the four corners of a box share z in the camera frame
    cam = (corners[1] - T[:3, 3]) @ T[:3, :3]
    assert np.allclose(cam[:, 2], 3.0, atol=1e-12)
    # default rule beside it: unchanged, and different where the corner pixel is not the median
    c0, v0 = engine.lift_boxes(boxes, K, T, None, 1000.0)
    o0, ov0 = orc.lift_boxes_oracle(boxes, depth, K, T, 1000.0)
    assert np.array_equal(v0, ov0) and np.allclose(c0[v0], o0[ov0], atol=1e-9)
    assert not np.allclose(np.nan_to_num(c0), np.nan_to_num(corners))
    # uint16 frames
    d16 = np.clip(np.round(np.nan_to_num(depth)), 0, 65535).astype(np.uint16)
    oc16, ov16 = orc.lift_boxes_oracle(boxes, d16, K, T, 1000.0, z_rule=1)
    c16, v16 = engine.lift_boxes(boxes, K, T, d16, 1000.0, z_rule=1)
    assert np.array_equal(v16, ov16) and np.allclose(c16[v16], oc16[ov16], atol=1e-9)
    # the batched call: three frames, boxes spread over them
    frames = np.stack([depth, np.roll(depth, 17, axis=1), np.roll(depth, 31, axis=0)])
    poses = np.stack([T, syn.pose_to_matrix([0.0, 1.0, 2.0, 0.3, 0.1, -0.2, 0.9]), np.eye(4)])
    bf = (np.arange(len(boxes)) % 3).astype(np.uint32)
    cb, vb = engine.lift_boxes_batch(boxes, bf, frames, K, poses, 1000.0, z_rule=1)
    for f in range(3):
        m = bf == f
        of, ovf = orc.lift_boxes_oracle(boxes[m], frames[f], K, poses[f], 1000.0, z_rule=1)
        assert np.array_equal(vb[m], ovf) and np.allclose(cb[m][ovf], of[ovf], atol=1e-9)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.lift_boxes(boxes, K, T, depth, 1000.0, z_rule=2)
    assert e.value.status == _lib.MA_ERR_INVALID
    # the per-frame helper passes the rule on
    intr = {"image_width": 1920, "image_height": 2560, "fx": 1500.0, "fy": 1510.0, "cx": 955.0, "cy": 1275.0}
    dets = [[400.0, 500.0, 900.0, 1500.0, 2, 0.9, "exit"]]
    rows = lifting.lift_frame_bboxes(engine, dets, depth, intr, T, w, h, z_rule=lifting.Z_MEDIAN)
    sb = np.array([lifting.scale_bounding_box(dets[0][:4], (1920, 2560), (w, h))])
    om, _ = orc.lift_boxes_oracle(sb, depth, lifting.rescaled_intrinsics(intr, w), T, 1000.0, z_rule=1)
    assert np.allclose(np.array(rows[0][:4]), om[0], atol=1e-9)


# ----------------------------------------------------------------------------- configs[3]
def _frames(n_frames, w, h, seed):
    rng = np.random.default_rng(seed)
    depth = rng.uniform(500.0, 9000.0, size=(n_frames, h, w)).astype(np.float32)       # millimetres, dataset.py:77
    depth[rng.random(depth.shape) < 0.05] = 0.0                                         # holes
    depth[0, 3, 4] = np.nan
    poses = np.zeros((n_frames, 7))
    poses[:, :3] = rng.uniform(-5, 5, size=(n_frames, 3))
    q = rng.normal(size=(n_frames, 4))
    poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return depth, poses


def test_lift_boxes_batch_vs_oracle_and_per_frame_call(engine):
    n_frames, w, h = 7, 160, 120
    depth, poses = _frames(n_frames, w, h, 0)
    K = (140.0, 141.0, 80.5, 59.5)
    rng = np.random.default_rng(1)
    nb = 300
    x1, y1 = rng.uniform(-5, w - 2, nb), rng.uniform(-5, h - 2, nb)
    boxes = np.stack([x1, y1, x1 + rng.uniform(1, 40, nb), y1 + rng.uniform(1, 40, nb)], axis=1).astype(np.float32)
    frame = rng.integers(0, n_frames, nb).astype(np.uint32)
    T = np.stack([syn.pose_to_matrix(p) for p in poses])
    corners, valid = engine.lift_boxes_batch(boxes, frame, depth, K, T)
    for f in range(n_frames):
        sel = frame == f
        oc, ov = orc.lift_boxes_oracle(boxes[sel], depth[f], K, T[f])
        assert np.array_equal(valid[sel], ov)
        assert np.allclose(corners[sel][ov], oc[ov], rtol=0, atol=1e-9)
        assert np.isnan(corners[sel][~ov]).all()
        pc, pv = engine.lift_boxes(boxes[sel], K, T[f], depth[f])
        assert np.array_equal(pv, ov) and np.array_equal(pc[pv], corners[sel][ov])
    # uint16 frames, empty batch, bad frame index
    d16 = np.clip(np.nan_to_num(depth), 0, 65535).astype(np.uint16)
    c16, v16 = engine.lift_boxes_batch(boxes, frame, d16, K, T)
    oc, ov = orc.lift_boxes_oracle(boxes[frame == 2], d16[2], K, T[2])
    assert np.array_equal(v16[frame == 2], ov) and np.allclose(c16[frame == 2][ov], oc[ov], atol=1e-9)
    ce, ve = engine.lift_boxes_batch(np.zeros((0, 4), np.float32), np.zeros(0, np.uint32), depth, K, T)
    assert ce.shape == (0, 4, 3) and ve.shape == (0, 4)
    with pytest.raises(mapalign.MapAlignError):
        engine.lift_boxes_batch(boxes[:2], np.array([0, n_frames], np.uint32), depth, K, T)


def test_config4_batch32_detector_and_one_call_lift(engine):
    """BASELINE configs[3]: batch-32 detector inference on PyTorch-ROCm (random-init
    YOLOv10-shaped stand-in: SURVEY.md 8d allows it, weights do not exist here) at 3x640x640,
    rows in the detector.py:155 / :126-127 schema, and the boxes of all 32 frames lifted by ONE
    ma_lift_boxes_batch call, checked against the oracle frame by frame."""
    import torch
    from mapalign import detect

    torch.manual_seed(0)
    n_frames = 32
    model = detect.build_yolov10_like()
    det = detect.Detector(model, None, batch_size=32, conf_thresh=0.0, device="cuda:0", half=True)
    imgs = torch.randint(0, 256, (n_frames, 640, 640, 3), dtype=torch.uint8)
    raw, _ = det.infer(imgs)
    assert raw.shape == (n_frames, detect.MAX_DET, 6) and raw.is_cuda and torch.isfinite(raw).all()
    det.conf_thresh = float(torch.quantile(raw[:, :, 4].flatten().float(), 0.9))
    preds = det(imgs)
    n_rows = sum(len(v) for v in preds.values())
    assert 0 < n_rows < n_frames * detect.MAX_DET
    row = next(r for v in preds.values() for r in v)
    assert len(row) == 7 and isinstance(row[6], int) and row[4] == 0           # [x1,y1,x2,y2,damage,conf,label]
    depth, poses = _frames(n_frames, 256, 192, 5)                               # the reference's depth size, dataset.py:19-20
    intr = {"fx": 1500.0, "fy": 1500.0, "cx": 320.0, "cy": 320.0, "image_width": 640, "image_height": 640}
    before = engine.mem_info()["device_allocs"]
    out = detect.lift_batch(engine, preds, depth, intr, poses)
    assert sum(len(v) for v in out.values()) == n_rows
    from mapalign import lifting
    K = lifting.rescaled_intrinsics(intr, 256)
    for f in (0, 7, 31):
        if not preds[f]:
            continue
        scaled = np.array([lifting.scale_bounding_box(r[:4], (640, 640), (256, 192)) for r in preds[f]], np.float32)
        oc, ov = orc.lift_boxes_oracle(scaled, depth[f], K, syn.pose_to_matrix(poses[f]))
        got = np.array([[c for c in r[:4]] for r in out[f]])
        assert np.allclose(got[ov], oc[ov], atol=1e-9) and np.isnan(got[~ov]).all()
        assert [r[4:] for r in out[f]] == [r[-3:] for r in preds[f]]
    # a second batch of the same shape reuses the device buffers
    mid = engine.mem_info()["device_allocs"]
    detect.lift_batch(engine, preds, depth, intr, poses)
    assert engine.mem_info()["device_allocs"] == mid and mid >= before
    # the damage leg on the GPU: a BEiT classifier of the reference's architecture
    # (BeitForImageClassification, classifier.py:40-53; random init: the fine-tuned weights are not shipped) sees
    # every kept box through the reference's crop geometry (detector.py:197-231: box warped onto the whole frame,
    # then the processor's resize) and its class lands in column 4 of the rows (detector.py:126-127)
    beit = detect.build_beit_like(num_labels=3)
    if beit is None:
        pytest.skip("transformers has no BeitForImageClassification here")
    det2 = detect.Detector(model, beit, batch_size=32, conf_thresh=det.conf_thresh, device="cuda:0", half=True, crop="full_frame")
    preds2 = det2(imgs)
    # (fp16 convolutions are not bitwise repeatable from run to run, and the threshold sits ON the confidences'
    # 0.9 quantile: a frame may gain or lose a box against the first run)
    assert all(abs(len(preds2[b]) - len(preds[b])) <= 3 for b in range(n_frames))
    rows2 = [r for b in range(n_frames) for r in preds2[b]]
    assert rows2 and all(len(r) == 7 and r[4] in (0, 1, 2) and isinstance(r[6], int) for r in rows2)
    # the classes are those of the classifier run by hand on the same crops of the same raw output
    raw4, x4 = det2.infer(imgs)
    keep = raw4[:, :, 4] >= det2.conf_thresh
    labels = det2._damage(x4, raw4, keep).cpu().numpy()
    assert len(labels) == int(keep.sum())
    b_idx, k_idx = keep.nonzero(as_tuple=True)
    with torch.no_grad():
        crops = det2.full_frame_crops(x4[b_idx[:32]].float(), raw4[b_idx[:32], k_idx[:32], :4])
        assert crops.shape == (32, 3, 224, 224) and torch.isfinite(crops).all() and float(crops.min()) >= -1.0 and float(crops.max()) <= 1.0
        logits = beit(crops.to(det2.dtype)).logits.float()
    top2 = logits.topk(2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 0.02).cpu().numpy()          # random-init logits can tie to fp16 noise
    assert clear.any() and (labels[:32][clear] == logits.argmax(-1).cpu().numpy()[clear]).all()


# ----------------------------------------------------------------------------- N1 with trajectories
def test_coarse_stage_uses_trajectories_on_a_half_turn_symmetric_shell(lib_built):
    """A plain box shell is symmetric under a half turn about its vertical axis: the PCA
    candidates that differ by it score the same on the clouds.  The camera trajectories
    (task_def.py:176-177) — walked 0.45 m off the centre line and 0.4 m above mid-height — decide."""
    rng = np.random.default_rng(3)
    rects = syn.carriage_rectangles(rng, n_panels=0)
    gold = syn.sample_rectangles(rects, 60_000, rng).astype(np.float32)
    s = np.linspace(1.0, 19.0, 60)
    base_traj = np.stack([s, np.full_like(s, 1.95), np.full_like(s, 1.65)], axis=1)
    T_true = transforms.rt_to_matrix(syn.rotation_about_axis([0.05, -0.03, 1.0], np.deg2rad(131.0)), [6.0, -3.0, 0.8])
    inv = np.linalg.inv(T_true)
    comp = transforms.apply(inv, syn.sample_rectangles(rects, 50_000, rng)).astype(np.float32)
    comp_traj = transforms.apply(inv, base_traj)
    pose = lambda xyz: np.concatenate([np.arange(len(xyz))[:, None] * 0.1, xyz, np.tile([0, 0, 0, 1.0], (len(xyz), 1))], axis=1)  # noqa: E731
    al = mapalign.Alignment(base_pose_df=pose(base_traj), comparison_pose_df=pose(comp_traj), max_iter=60)
    stages, T, info = al.align_clouds(gold, comp)
    assert info["trajectory_scores"] is not None
    assert coarse.trajectory_residual(T, base_traj, comp_traj) < 0.02
    eR, et = orc.rt_error(T[:3, :3], T[:3, 3], T_true[:3, :3], T_true[:3, 3])
    assert eR <= 1e-3 and et <= 5e-3, (eR, et)
    # the two best cloud scores are the half-turn twins and nearly equal: geometry alone is a coin flip
    cs = np.sort(info["cloud_scores"])
    assert cs[1] - cs[0] < 0.25 * cs[0] + 1e-3
    ts = np.array(info["trajectory_scores"])
    assert ts.min() < 0.1 and np.sort(ts)[1] > 0.4
    # the emitted list replays through the visualiser's log-map interpolation (visualise.py:82-130)
    from visualiser_replay import assert_replayable
    assert all(isinstance(st, tuple) or np.shape(st) == (4, 4) for st in stages)
    assert_replayable(stages, comp[::500].astype(np.float64), T, atol=1e-6)
    mapalign.release_cached_engines()


# ----------------------------------------------------------------------------- a8: in-place unprojection, holes
def test_unprojection_in_place_with_holes_sizes_and_switching(engine):
    """k_unproject writes pixel i into slot i — a point or a hole — in one streaming pass with no exchange
    between workgroups (round 3).  What the caller sees is unchanged: the number of valid pixels, and through
    ma_get_source / ma_get_source_stored / ma_nn the valid pixels in row-major order (compacted lazily through
    a rank map).  Small and large frames, ragged last workgroups, 30-55 % holes, whole invalid rows, a frame
    without holes, and switching between sizes on one context."""
    gold, _, _, _ = syn.make_pair(2, 50_000)
    engine.set_gold(gold)
    rng = np.random.default_rng(11)
    K_small, K_big = (300.0, 301.0, 160.0, 120.0), (1500.0, 1500.0, 1024.0, 600.0)
    T = syn.default_camera_pose()

    def frame(w, h, holes):
        d = rng.uniform(800.0, 9000.0, size=(h, w)).astype(np.float32)
        d[rng.random((h, w)) < holes] = 0.0
        return d

    small = [frame(320, 240, 0.3) for _ in range(3)]
    small[1][100:140] = np.nan                                # forty invalid rows
    odd = frame(333, 17, 0.4)                                 # 5661 pixels: ragged last workgroup, rows not a multiple of 4
    full = frame(320, 240, 0.0)                               # no hole at all: no rank map is ever built
    big = frame(2048, 1200, 0.55)
    for depth, K in ((small[0], K_small), (small[1], K_small), (big, K_big), (odd, K_small), (small[2], K_small), (full, K_small),
                     (big, K_big), (small[0], K_small)):
        n = engine.set_source_from_depth(depth, K, T, 1000.0)
        ref, mask = orc.unproject_oracle(depth, K, T, 1000.0)
        assert n == int(mask.sum()) == len(ref)
        got = engine.get_source()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-5      # fp32 storage of coordinates up to ~10 m
        stored = engine.get_source_stored()
        assert stored.shape == ref.shape and np.abs(stored - ref).max() < 2e-5
        idx, d2 = engine.nn(None, 0.5)                          # per-point results come back for the valid pixels only, in order
        assert idx.shape == (n,) and d2.shape == (n,)
        oidx, od2 = orc.GoldIndex(gold, n_jobs=4).query(ref)
        m = od2 <= 0.25 - 1e-4
        assert ((idx != 0xFFFFFFFF) | ~m).all() and np.abs(d2[m] - od2[m]).max() < 2e-4
        res = engine.icp(3, 0.0, 0.5)                           # holes contribute nothing: W counts matched valid pixels only
        assert res.inliers <= n
    # an image without any valid pixel is an error, not an empty cloud
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.set_source_from_depth(np.zeros((240, 320), np.float32), K_small, T, 1000.0)
    assert e.value.status == _lib.MA_ERR_EMPTY


# ----------------------------------------------------------------------------- device-resident inputs
def test_device_resident_inputs_match_host_inputs(engine):
    """Clouds, weights and depth frames that already live in the GPU's memory (torch tensors,
    `__cuda_array_interface__`) go through the same C entry points (hipMemcpyDefault) and give
    bitwise the results of the host arrays — nothing is converted on the way."""
    import torch

    gold, comp, R0, t0 = syn.make_pair(3, 200_000)
    w = np.random.default_rng(3).uniform(0.25, 2.0, len(comp)).astype(np.float32)
    g_d, c_d, w_d = (torch.from_numpy(a).cuda() for a in (gold, comp, w))

    engine.set_gold(gold)
    engine.set_source(comp, w)
    host = engine.icp(max_iter=30, tol=0.0, max_corr_dist=0.10)
    engine.set_gold(g_d)
    engine.set_source(c_d, w_d)
    dev = engine.icp(max_iter=30, tol=0.0, max_corr_dist=0.10)
    assert np.array_equal(host.T, dev.T) and host.inliers == dev.inliers
    assert np.linalg.norm(dev.R - R0) < 1e-3

    # the drop-in call takes them too; fp64 tensors and strided views are converted on the device
    R1, t1 = mapalign.align(g_d, c_d.double()[::1], max_iter=30)
    R2, t2 = mapalign.align(gold, comp, max_iter=30)
    assert np.array_equal(R1, R2) and np.array_equal(t1, t2)
    view = torch.stack([c_d, c_d], dim=1)[:, 0, :]          # non-contiguous view of the same points
    assert not view.is_contiguous()
    R3, t3 = mapalign.align(g_d, view, max_iter=30)
    assert np.array_equal(R3, R2) and np.array_equal(t3, t2)

    # fp64 host arrays (Open3D's dtype) are converted on the device: same bits as the fp32 arrays they round to
    R4, t4 = mapalign.align(gold.astype(np.float64), comp.astype(np.float64), weights=w, max_iter=30)
    R5, t5 = mapalign.align(gold, comp, weights=w, max_iter=30)
    assert np.array_equal(R4, R5) and np.array_equal(t4, t5)
    with pytest.raises(mapalign.MapAlignError):
        bad = gold.astype(np.float64)
        bad[5, 1] = np.nan
        engine.set_gold(bad)

    # the fp64 entry points of the C ABI take device pointers too (here: torch float64 tensors)
    import ctypes as C
    g64_d, c64_d = g_d.double().contiguous(), c_d.double().contiguous()
    torch.cuda.synchronize()
    shift = np.zeros(3)
    f64p = C.POINTER(C.c_double)
    engine._chk(engine._lib.ma_set_gold_f64(engine._ctx, C.cast(C.c_void_p(g64_d.data_ptr()), f64p), len(gold), 0.0,
                                            shift.ctypes.data_as(f64p)))
    engine._chk(engine._lib.ma_set_source_f64(engine._ctx, C.cast(C.c_void_p(c64_d.data_ptr()), f64p), None, len(comp), None))
    engine._shift, engine.n_gold, engine.n_source = shift, len(gold), len(comp)
    raw64 = engine.icp(max_iter=30, tol=0.0, max_corr_dist=0.10)
    engine.set_gold(gold)
    engine.set_source(comp)
    assert not shift.any() and np.array_equal(raw64.T, engine.icp(max_iter=30, tol=0.0, max_corr_dist=0.10).T)

    # georeferenced fp64 tensors on the device: shifted and converted there, same bits as the host fp64 arrays
    off = np.array([412_345.0, 5_234_567.0, 120.0])
    gold_geo, comp_geo = gold.astype(np.float64) + off, comp.astype(np.float64) + off
    R6, t6 = mapalign.align(torch.from_numpy(gold_geo).cuda(), torch.from_numpy(comp_geo).cuda(), max_iter=30)
    R7, t7 = mapalign.align(gold_geo, comp_geo, max_iter=30)
    assert np.array_equal(R6, R7) and np.array_equal(t6, t7)
    with pytest.raises(ValueError):                      # an fp32 device scan cannot follow a gold map in a shifted frame
        engine.set_gold(gold_geo)
        engine.set_source(c_d)

    # depth frame on the device (float32 millimetres and uint16)
    depth, K, T = syn.make_depth_frame(0, width=320, height=240, fx=250.0, fy=250.0, cx=160.0, cy=120.0)
    engine.set_gold(gold)
    n_host = engine.set_source_from_depth(depth, K, T, 1000.0)
    a = engine.icp(max_iter=5, tol=0.0, max_corr_dist=0.10)
    n_dev = engine.set_source_from_depth(torch.from_numpy(depth).cuda(), K, T, 1000.0)
    b = engine.icp(max_iter=5, tol=0.0, max_corr_dist=0.10)
    assert n_host == n_dev > 0 and np.array_equal(a.T, b.T)
    d16 = np.clip(depth, 0, 65535).astype(np.uint16)
    assert engine.set_source_from_depth(torch.from_numpy(d16.astype(np.int32)).to(torch.uint16).cuda(), K, T, 1000.0) \
        == engine.set_source_from_depth(d16, K, T, 1000.0)

    with pytest.raises(ValueError):
        engine.set_gold(g_d[:, :2])
    with pytest.raises(ValueError):
        engine.set_source(c_d, w_d[:-1])


def test_map_builder_long_voxel_runs_and_repeatability(engine):
    """Voxels that hold hundreds to thousands of points: their runs span many of the 16-point
    blocks the centroid kernels work in (prefix pieces, open pieces, the sentinel of the last
    voxel); checked against the NumPy restatement, and two builds must agree bitwise."""
    K = (250.0, 250.0, 159.5, 119.5)
    rects = syn.carriage_rectangles(np.random.default_rng(8))
    poses = []
    for k, x in enumerate((-3.0, 2.5, 5.0, 7.5)):            # negative coordinates too
        T = syn.default_camera_pose()
        T[:3, 3] = (x, 1.4 + 0.05 * k, 1.3)
        poses.append(T)
    frames = [syn.render_depth(rects, T, 320, 240, *K) for T in poses]
    pts = np.concatenate([orc.unproject_oracle(d, K, T, 1000.0)[0] for d, T in zip(frames, poses)])
    origin = np.round(poses[0][:3, 3])
    for voxel in (0.4, 0.011):                                # very dense voxels; nearly one point per voxel
        clouds = []
        for _ in range(2):
            engine.map_begin(voxel, 4 * 320 * 240)
            assert sum(engine.map_add_depth(d, K, T) for d, T in zip(frames, poses)) == len(pts)
            clouds.append(engine.map_finish())
        assert np.array_equal(clouds[0], clouds[1])
        ref = orc.voxel_centroids_oracle(pts, voxel, origin)
        assert clouds[0].shape == ref.shape
        assert np.abs(clouds[0].astype(np.float64) - ref).max() <= 2e-5
    counts = np.unique(np.floor((pts - origin) / 0.4).astype(np.int64), axis=0, return_counts=True)[1]
    assert counts.max() > 500 and len(ref) > 0.5 * len(pts)   # the two regimes really were exercised


def test_clump_of_coincident_returns_in_both_clouds(engine):
    """Real scans carry clumps of coincident points (invalid returns mapped to one position): 20 000
    copies of one point in the gold map AND in the scan put a 20 000-point run into one grid cell,
    which every query near it has to scan chunk by chunk.  Must stay exact and finish promptly."""
    gold, comp, R0, t0 = syn.make_pair(11, 150_000, 120_000)
    spot = gold[777].copy()
    gold = np.concatenate([gold, np.tile(spot, (20_000, 1))]).astype(np.float32)
    comp_spot = (spot.astype(np.float64) - t0) @ R0
    comp = np.concatenate([comp, np.tile(comp_spot, (20_000, 1))]).astype(np.float32)
    engine.set_gold(gold)
    engine.set_source(comp)
    t_start = time.perf_counter()
    res = engine.icp(30, 0.0, 0.10)
    assert time.perf_counter() - t_start < 5.0
    ref = orc.icp_oracle(gold, comp, max_iter=30, max_corr_dist=0.10, n_jobs=-1)
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert abs(res.inliers - ref["inliers"]) <= 2
    idx, d2 = engine.nn(res.T, 0.10)
    # the scan's clump lands on the map's clump (30 iterations leave centimetres): all its points match within the gate
    clump = np.arange(len(comp) - 20_000, len(comp))
    assert np.all(np.isfinite(d2[clump])) and np.all(d2[clump] < 0.05 ** 2) and np.all(idx[clump] != NO_MATCH)


def test_alignment_gate_schedule_recovers_a_rough_coarse_start(lib_built):
    """The coarse stage is good to a few degrees at best on real scans.  From a start 5 degrees /
    0.8 m off, the 0.10 m gate alone stays wrong; the adaptor's default schedule (0.8 -> 0.4 ->
    0.2 m before the final gate) reaches the answer, and from a good start both agree."""
    gold, comp, R0, t0 = syn.make_pair(5, 200_000)
    T_true = transforms.rt_to_matrix(R0, t0)
    c = gold.mean(axis=0).astype(np.float64)
    dR = syn.rotation_about_axis([0.2, -0.4, 0.9], np.deg2rad(5.0))
    off = np.eye(4)
    off[:3, :3] = dR
    off[:3, 3] = c - dR @ c + np.array([0.6, -0.4, 0.3])
    rough = off @ T_true

    class RoughCoarse:                                   # stands in for a principal-axes start that is a few degrees off
        @staticmethod
        def coarse_align(eng, g, s, **kw):
            return [rough], rough, {}

    import mapalign.alignment as al_mod
    real = al_mod.coarse
    al_mod.coarse = RoughCoarse
    try:
        _, T_single, _ = mapalign.Alignment(gate_schedule=None).align_clouds(gold, comp)
        stages, T_sched, info = mapalign.Alignment().align_clouds(gold, comp)
    finally:
        al_mod.coarse = real
    eR1, et1 = orc.rt_error(T_single[:3, :3], T_single[:3, 3], R0, t0)
    eR2, et2 = orc.rt_error(T_sched[:3, :3], T_sched[:3, 3], R0, t0)
    assert et1 > 0.2 and eR2 <= 1e-3 and et2 <= 5e-3, (eR1, et1, eR2, et2)
    assert [g for g, _, _ in info["gate_schedule"]] == [0.8, 0.4, 0.2]
    assert np.allclose(transforms.compose(stages), T_sched, atol=1e-9)
    # the same schedule through align(): a start 5 degrees / 0.8 m off
    R_a, t_a, info_a = mapalign.align(gold, comp, init=rough, gate_schedule=[(0.8, 15), (0.4, 15), (0.2, 15)], return_info=True)
    eR3, et3 = orc.rt_error(R_a, t_a, R0, t0)
    assert eR3 <= 1e-3 and et3 <= 5e-3 and len(info_a["gate_schedule"]) == 3
    # from the truth-adjacent start of the benchmark pair the schedule changes nothing that matters
    _, T_a, _ = mapalign.Alignment(coarse_stage=False, gate_schedule=None).align_clouds(gold, comp)
    _, T_b, _ = mapalign.Alignment(coarse_stage=False, gate_schedule=[(0.8, 15), (0.4, 15), (0.2, 15)]).align_clouds(gold, comp)
    eR, et = orc.rt_error(T_a[:3, :3], T_a[:3, 3], T_b[:3, :3], T_b[:3, 3])
    assert eR <= 1e-4 and et <= 1e-3
    mapalign.release_cached_engines()


def test_align_many_pipelines_pairs_and_equals_align_pair_by_pair(lib_built):
    """A stream of pairs on one GPU: the next pair is staged in a second context while the current
    one iterates.  Every result equals mapalign.align of that pair bit for bit, in order."""
    pairs = [syn.make_pair(40 + k, 60_000 + 5_000 * k, 50_000)[:2] for k in range(5)]
    one_by_one = []
    for g, c in pairs:
        R, t = mapalign.align(g, c, max_iter=30)
        one_by_one.append(transforms.rt_to_matrix(R, t))
    for workers in (3, 2, 1):
        Ts = mapalign.align_many(iter(pairs), max_iter=30, workers=workers)
        assert len(Ts) == len(pairs)
        for a, b in zip(Ts, one_by_one):
            assert np.array_equal(a, b)
    # one gold map, many scans: the same gold OBJECT is indexed once per worker, results unchanged
    g0 = pairs[0][0]
    scans = [p[1] for p in pairs]
    same_gold = mapalign.align_many([(g0, s) for s in scans], max_iter=30, workers=2)
    for T, s in zip(same_gold, scans):
        R, t = mapalign.align(g0, s, max_iter=30)
        assert np.array_equal(T, transforms.rt_to_matrix(R, t))
    with mapalign.GoldMap(g0) as gm:                       # a resident gold map in the stream: its own context is used
        via_gm = mapalign.align_many([(gm, s) for s in scans[:3]], max_iter=30, workers=2)
    assert all(np.array_equal(a, b) for a, b in zip(via_gm, same_gold[:3]))
    assert mapalign.align_many([], max_iter=5) == []
    # an error in one pair surfaces as that pair's exception and leaves the contexts usable
    bad = [pairs[0], (pairs[1][0], np.full((10, 3), np.nan, np.float32)), pairs[2]]
    with pytest.raises(mapalign.MapAlignError):
        mapalign.align_many(bad, max_iter=5)
    assert np.array_equal(mapalign.align_many(pairs[:2], max_iter=30)[1], one_by_one[1])
    from mapalign import dist as mdist
    out = mdist.align_pairs(pairs, max_iter=30, device=0)       # single process: all pairs, pipelined
    assert all(np.array_equal(a, b) for a, b in zip(out, one_by_one))
    mapalign.release_cached_engines()


def test_icp_loop_fuzz_vs_oracle(engine):
    """The whole loop (search, sums, solve, compose) against the oracle on 60 seeded cases: volumes, slabs,
    clusters and carriage shells of 50-12 000 points, gates 0.03 m ... none, density-driven and fixed
    cell edges, a third of the cases weighted; 6 iterations each."""
    rng = np.random.default_rng(2027)
    for case in range(60):
        m, n = int(rng.integers(50, 12000)), int(rng.integers(50, 12000))
        kind = case % 4
        if kind == 0:
            gold = rng.uniform(-1, 1, (m, 3)) * rng.uniform(0.3, 5.0, 3)
        elif kind == 1:
            gold = rng.uniform(-3, 3, (m, 3))
            gold[:, rng.integers(0, 3)] *= 0.01
        elif kind == 2:
            c = rng.uniform(-2, 2, (max(m // 300, 2), 3))
            gold = c[rng.integers(0, len(c), m)] + rng.normal(0, 0.05, (m, 3))
        else:
            gold = syn.sample_rectangles(syn.carriage_rectangles(rng), m, rng)
        gold = gold.astype(np.float32)
        near = gold[rng.integers(0, m, n)].astype(np.float64) + rng.normal(0, 0.003, (n, 3))
        R0 = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, 0.02))
        t0 = rng.normal(0, 0.01, 3)
        comp = ((near - t0) @ R0).astype(np.float32)
        gate = [0.03, 0.1, 0.5, np.inf][int(rng.integers(0, 4))]
        w = rng.uniform(0.1, 3.0, n).astype(np.float32) if case % 3 == 0 else None
        cell = [0.0, 0.0, 0.02, 0.2, 1.5][int(rng.integers(0, 5))]
        engine.set_gold(gold, cell)
        engine.set_source(comp, w)
        res = engine.icp(6, 0.0, gate)
        ref = orc.icp_oracle(gold, comp, max_iter=6, max_corr_dist=gate, weights=w, n_jobs=4)
        eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
        # (a tenth of the parity bars: with a 3 cm gate a few points sit within rounding of the gate and flip)
        assert ref["ok"] and eR <= 1e-5 and et <= 1e-4, (case, kind, m, n, gate, cell, eR, et)
        assert abs(res.inliers - ref["inliers"]) <= 1e-3 * max(ref["inliers"], 1.0), (case, res.inliers, ref["inliers"])


def test_alignment_adaptor_with_georeferenced_fp64_clouds(lib_built):
    """The whole adaptor (coarse stage on a sub-sample, gate schedule, final loop) on UTM-like fp64
    coordinates: every stage must work in the shifted frame; the scan starts 150 degrees / 8 m off."""
    gold, comp, R0, t0 = syn.make_pair(19, 60_000, 50_000, angle_deg=150.0, t0=np.array([7.0, -4.0, 1.5]))
    off = np.array([412_345.0, 5_234_567.0, 120.0])
    g64, c64 = gold.astype(np.float64) + off, comp.astype(np.float64) + off
    stages, T, info = mapalign.Alignment().align_clouds(g64, c64)
    moved = c64[::53] @ T[:3, :3].T + T[:3, 3]
    expect = comp[::53].astype(np.float64) @ R0.T + t0 + off
    assert np.abs(moved - expect).max() < 0.02, np.abs(moved - expect).max()
    assert np.allclose(transforms.compose(stages), T, atol=1e-6)
    mapalign.release_cached_engines()


# ----------------------------------------------------------------------------- round 3: overlapped staging, shared contexts
def test_set_pair_equals_the_two_calls_bitwise_and_shared_context_users_do_not_mix_gold_maps(lib_built):
    """ma_set_pair (the scan uploaded by a helper thread while the gold map is indexed) leaves bitwise the state of
    ma_set_gold + ma_set_source; and align() between two align_many() calls on the same cached context
    cannot make a worker align against the wrong map."""
    rng = np.random.default_rng(8)
    gold, comp, _, _ = syn.make_pair(21, 80_000, 70_001)
    w = rng.uniform(0.2, 2.0, len(comp)).astype(np.float32)
    with mapalign.MapAligner(0) as a, mapalign.MapAligner(0) as b:
        ga = a.set_gold(gold)
        a.set_source(comp, w)
        ra = a.icp(12, 0.0, 0.10, with_trace=True)
        gb = b.set_pair(gold, comp, w)
        rb = b.icp(12, 0.0, 0.10, with_trace=True)
        assert np.array_equal(ra.T, rb.T) and np.array_equal(ra.trace, rb.trace)
        assert (ga["cell"], ga["dims"], ga["n_occupied"]) == (gb["cell"], gb["dims"], gb["n_occupied"])
        for _ in range(3):                                   # repeated on a warm context (buffers reused, helper thread per call)
            b.set_pair(gold, comp, w)
        assert np.array_equal(b.icp(12, 0.0, 0.10).T, ra.T)
        with pytest.raises(mapalign.MapAlignError) as e:
            b.set_pair(gold, np.full((5, 3), np.nan, np.float32))
        assert e.value.status == _lib.MA_ERR_INVALID
        # two float64 host arrays (what Open3D hands over) take ma_set_pair_f64: bitwise the two fp64 calls, georeferenced too
        for off in (np.zeros(3), np.array([4.5e5, 5.2e6, 120.0])):
            g64, c64 = gold.astype(np.float64) + off, comp.astype(np.float64) + off
            a.set_gold(g64)
            a.set_source(c64, w)
            r2 = a.icp(12, 0.0, 0.10, with_trace=True)
            b.set_pair(g64, c64, w)
            r3 = b.icp(12, 0.0, 0.10, with_trace=True)
            assert np.array_equal(r2.T, r3.T) and np.array_equal(r2.trace, r3.trace)
        a.set_gold(gold)
        a.set_source(comp)
        plain = a.icp(12, 0.0, 0.10)
        b.set_pair(gold, comp.astype(np.float64))            # mixed dtypes: the two calls underneath, the same cloud values
        assert np.array_equal(b.icp(12, 0.0, 0.10).T, plain.T)
    mapalign.release_cached_engines()
    gold2, comp2, _, _ = syn.make_pair(22, 60_000, 50_000)
    scans = [comp[: 60_000 + 500 * k] for k in range(4)]
    want = [np.vstack([np.hstack([R, t[:, None]]), [0, 0, 0, 1]]) for R, t in (mapalign.align(gold, s, max_iter=10) for s in scans)]
    first = mapalign.align_many([(gold, s) for s in scans[:2]], max_iter=10, workers=1)
    mapalign.align(gold2, comp2, max_iter=5)                  # another user of cached context 0 installs another gold map
    second = mapalign.align_many([(gold, s) for s in scans[2:]], max_iter=10, workers=1)
    for T, ref in zip(first + second, want):
        assert np.array_equal(np.asarray(T), ref)
    mapalign.release_cached_engines()


# ----------------------------------------------------------------------------- round 3: the device solve (a6) on crafted sums
def test_device_solve_vs_numpy_svd_through_step_solve(lib_built):
    """The device solve (kabsch.h: one-sided Jacobi SVD + Kabsch on one lane) through ma_step_solve: accumulators built from
    exact correspondences — generic, planar (third singular value 0), reflected-looking (det(V U^T) < 0 without the
    fix), nearly aligned, large rotations, weighted — must give the oracle's (numpy SVD) update; degenerate sums
    (collinear points, one pair, W = 0) must stop the state with the degenerate flag, not produce a transform.  The sums
    travel as the library's fixed-point int64 words (conftest.fixed_point_sums; a NaN cannot be written in them any more)."""
    import torch

    from conftest import fixed_point_sums

    rng = np.random.default_rng(31)
    gold, comp, _, _ = syn.make_pair(3, 2000, 1500)
    cases = []
    for k in range(40):
        n = int(rng.integers(3, 400))
        p = rng.normal(size=(n, 3)) * rng.uniform(0.1, 20.0, size=3) + rng.normal(size=3) * 5
        if k % 4 == 1:
            p[:, int(rng.integers(0, 3))] = 0.7                      # planar cloud
        R = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, np.pi) if k % 3 else rng.uniform(0, 1e-4))
        q = p @ R.T + rng.normal(size=3) + (rng.normal(0, 1e-3, size=p.shape) if k % 2 else 0.0)
        if k % 8 == 5:
            q = q * np.array([1.0, 1.0, -1.0])                       # mirrored target: the SVD alone would return a reflection
        w = rng.uniform(0.1, 3.0, n) if k % 5 == 0 else np.ones(n)
        cases.append(orc.accumulate(p, q, w, np.sum((p - q) ** 2, axis=1)))
    line = np.outer(np.linspace(-1, 1, 50), [1.0, 2.0, -0.5])
    bad = [orc.accumulate(line, line + 0.1, np.ones(50), np.zeros(50)),                  # collinear: rank 1
           orc.accumulate(np.ones((1, 3)), np.zeros((1, 3)), np.ones(1), np.zeros(1)),   # a single pair
           np.zeros(17)]
    with mapalign.MapAligner(0) as eng:
        eng.set_gold(gold)
        eng.set_source(comp)
        origin = eng.grid_info()["origin"]
        acc_dev = torch.zeros(17, dtype=torch.int64, device="cuda:0")

        def centred(acc):          # the library's sums live in the frame centred on the gold map's origin (engine._acc_shift, inverted)
            a = np.array(acc, dtype=np.float64)
            W, sp, sq = a[0], a[1:4] - a[0] * origin, a[4:7] - a[0] * origin
            S = a[7:16].reshape(3, 3) - np.outer(sp, origin) - np.outer(origin, sq) - W * np.outer(origin, origin)
            return np.concatenate([[W], sp, sq, S.ravel(), [a[16]]])

        for acc in cases:
            dR, dt, ok = orc.kabsch_from_acc(acc)
            assert ok
            acc_dev.copy_(torch.from_numpy(fixed_point_sums(eng, centred(acc))))
            eng.state_set(None)
            torch.cuda.synchronize()
            eng.step_solve(acc_dev.data_ptr())
            T, info = eng.state_get()
            assert info["iterations"] == 1 and not info["degenerate"]
            assert np.abs(T[:3, :3] - dR).max() < 1e-9 and np.abs(T[:3, 3] - dt).max() < 1e-8 * max(1.0, np.abs(dt).max())
            assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-12
        for acc in bad:
            acc_dev.copy_(torch.from_numpy(fixed_point_sums(eng, centred(acc))))
            eng.state_set(None)
            torch.cuda.synchronize()
            eng.step_solve(acc_dev.data_ptr())
            T, info = eng.state_get()
            assert info["degenerate"] and np.array_equal(T, np.eye(4))


def test_cloud_moments_and_truncated_rms_against_numpy(engine):
    """N1 on the device: ma_cloud_moments (count, centroid, covariance of a cloud as stored) against NumPy on the same
    points — float32 clouds, a georeferenced float64 pair (large-coordinate shift in force), a depth frame with holes —
    and MapAligner.truncated_rms (from the 17 sums of one gated launch) against the per-point distances of ma_nn."""
    gold, comp, _, _ = syn.make_pair(4, 60_000, 50_000)
    engine.set_pair(gold, comp)
    for which, cloud in ((0, gold), (1, comp)):
        n, c, C = engine.cloud_moments(which)
        p = cloud.astype(np.float64)
        assert n == len(p)
        assert np.abs(c - p.mean(0)).max() < 1e-6
        assert np.abs(C - np.cov(p.T)).max() < 1e-6 * np.abs(np.cov(p.T)).max() + 1e-9
    T = transforms.rt_to_matrix(syn.rotation_about_axis([0.2, 1.0, 0.1], 0.4), [0.5, -0.2, 0.1])
    for bound in (0.05, 0.19, 0.6):
        for M in (np.eye(4), T):
            _, d2 = engine.nn(M, bound)
            want = float(np.sqrt(np.mean(np.minimum(d2.astype(np.float64), bound * bound))))
            got = engine.truncated_rms(M, bound)
            assert abs(got - want) < 1e-6 * max(want, 1e-3), (bound, got, want)
    # nothing inside the bound at all: the score is the bound, not an error
    far = transforms.rt_to_matrix(np.eye(3), [500.0, 0.0, 0.0])
    assert abs(engine.truncated_rms(far, 0.1) - 0.1) < 1e-12
    # georeferenced float64 clouds: the centroid comes back in the caller's frame
    off = np.array([4.5e5, 5.2e6, 120.0])
    g64, c64 = gold.astype(np.float64) + off, comp.astype(np.float64) + off
    engine.set_pair(g64, c64)
    n, c, C = engine.cloud_moments(1)
    assert n == len(c64) and np.abs(c - c64.mean(0)).max() < 1e-4 and np.abs(C - np.cov(c64.T)).max() < 1e-4
    # a depth frame with holes as the source cloud: holes do not count
    depth, K, Tcw = syn.make_depth_frame(0, width=128, height=96, fx=100.0, fy=100.0, cx=63.5, cy=47.5)
    rng = np.random.default_rng(0)
    depth = depth.copy()
    depth[rng.random(depth.shape) < 0.3] = 0.0
    engine.set_gold(gold)
    engine.set_source_from_depth(depth, K, Tcw, 1000.0)
    pts = engine.get_source().astype(np.float64)
    n, c, C = engine.cloud_moments(1)
    assert n == len(pts) == int((depth > 0).sum())
    assert np.abs(c - pts.mean(0)).max() < 1e-5 and np.abs(C - np.cov(pts.T)).max() < 1e-5


def test_coarse_stage_on_the_resident_pair_equals_the_host_side_stage(engine, monkeypatch):
    """coarse_align(resident=True) — principal frames from the device's moment sums, candidates scored from the sums of
    one narrow-gated launch each over the whole comparison cloud — picks the same candidate as the host-side stage
    (sub-sample, per-point distances) and lands within a few millimetres / 0.1 degree of it; when the narrow round is
    undecided (forced here) it falls back to the wide bound and leaves the full comparison cloud in the engine."""
    gold, _, _, _ = syn.make_pair(0, 200_000)
    far = syn.make_pair(0, 200_000, angle_deg=150.0, t0=(7.0, -1.0, 0.5))[1]
    engine.set_gold(gold)
    stages_h, T_h, info_h = coarse.coarse_align(engine, gold, far)
    engine.set_pair(gold, far)
    stages_r, T_r, info_r = coarse.coarse_align(engine, gold, far, resident=True)
    assert info_r["chosen"] == info_h["chosen"] and info_r["score_bound_m"] < info_r["truncate_m"]
    assert engine.n_source == len(far)
    dR = T_r[:3, :3] @ T_h[:3, :3].T
    assert np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))) < 0.1 and np.linalg.norm(T_r[:3, 3] - T_h[:3, 3]) < 0.02
    r = engine.icp(30, 0.0, 0.1, init=T_r)              # the full cloud is what iterates
    assert r.inliers > 0.5 * len(far)
    monkeypatch.setattr(coarse, "DECISIVE", 0.0)         # never decisive: the wide round must run and tidy up after itself
    stages_w, T_w, info_w = coarse.coarse_align(engine, gold, far, resident=True)
    assert info_w["score_bound_m"] == info_w["truncate_m"] and info_w["chosen"] == info_h["chosen"]
    assert engine.n_source == len(far) and np.allclose(T_w, T_r)
