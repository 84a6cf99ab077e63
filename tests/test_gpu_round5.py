"""GPU parity tests added in round 5 (run with -m gpu on an MI355X), all through the C ABI:

* BASELINE configs[2] at EQUAL iteration count: the full 1280 x 960 depth frame (holes included) unprojected in place
  (``ma_set_source_from_depth`` without a count: the alignment is enqueued right behind it) and aligned against the
  1M-point gold map for 50 iterations, against the oracle fed the oracle's own unprojection of the same frame —
  per-iteration inlier counts and squared-distance sums, final (R, t), and candidate lists on = off bit for bit on this
  pixel-ordered, hole-carrying source (pose_processor.py:184-196,242-260, dataset.py:68-81 conventions);
* the count of valid pixels of a frame too large for the pinned count block (> 16.7 MP) decides what it decides for
  every other frame.
"""
import os
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


# ----------------------------------------------------------------------------- the peer-to-peer exchange (row e), rehearsed on one GPU
def _rehearse(*args, env=None, timeout=600):
    """tools/p2p_rehearsal.py in a FRESH process (its contexts' streams must sit on different hardware queues: see the
    script) -> its JSON line."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_rehearsal.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("ranks,points,tol", [(2, 300_000, "1e-7"), (4, 1_000_000, "0")])
def test_p2p_exchange_ranks_as_processes_return_the_single_context_bits(lib_built, ranks, points, tol):
    """ma_icp_sharded over the peer-to-peer exchange (k_reduce_exchange_solve: sum -> write the 17 tagged words into every
    peer's rows -> wait -> add -> solve, one launch), the ranks as PROCESSES sharing this GPU: rows mapped with
    hipIpcGetMemHandle / hipIpcOpenMemHandle, blobs carried by a gloo all-gather (mapalign.dist.bootstrap_p2p), bounds
    agreed over the exchange itself, block-partitioned comparison cloud.  Every rank must return ma_icp's result for the
    whole cloud BIT FOR BIT, run after run (the rows' tags go round more than once); with tol > 0 the ranks also have to
    stop together."""
    res = _rehearse("--mode", "procs", "--ranks", str(ranks), "--points", str(points), "--iters", "60", "--tol", tol, "--reps", "3")
    print(res)
    assert res["ok"] and res["bitwise_equal_to_one_context"], res


def test_p2p_exchange_guarded_kernel_of_nine_to_sixteen_ranks_with_three(lib_built):
    """Nine to sixteen ranks run the instantiation of k_reduce_exchange_solve that is unrolled over 15 possible peers with a
    guard each; no box holds that many ranks, so MAPALIGN_P2P_GENERIC=1 makes every context launch it with the ranks there
    are: three processes, a convergence test, the single context's bits."""
    res = _rehearse("--mode", "procs", "--ranks", "3", "--points", "200000", "--iters", "60", "--tol", "1e-7", "--reps", "3",
                    env={"MAPALIGN_P2P_GENERIC": "1"})
    print(res)
    assert res["ok"] and res["bitwise_equal_to_one_context"], res


def test_p2p_init_wants_a_fresh_export_and_this_process_own_blob(engine):
    """ma_p2p_export zeroes this rank's rows; a second ma_p2p_init on the same rows would start counting from 0 again over
    rows whose tags an earlier set-up left behind, so it is refused (MA_ERR_STATE) until the rows are exported again.  A blob
    of another process (its nonce differs even where the pid does not) is not accepted as this context's own."""
    blob = engine.p2p_export()
    engine.p2p_init([blob], 0)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.p2p_init([blob], 0)
    assert e.value.status == mapalign._lib.MA_ERR_STATE
    forged = bytearray(engine.p2p_export())
    forged[16] ^= 0xFF                                   # the process nonce (after magic and pid)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.p2p_init([bytes(forged)], 0)
    assert e.value.status == mapalign._lib.MA_ERR_INVALID
    engine.p2p_init([engine.p2p_export()], 0)            # and a proper bootstrap still works
    engine.p2p_destroy()


def test_p2p_exchange_contexts_of_one_process(lib_built):
    """The same with two CONTEXTS of one process (pointers used directly, one host thread each).  On one device this form
    needs the two contexts' streams on different hardware queues — the runtime decides that, and when it puts them into one
    queue a waiting exchange kernel sits in front of its peer's search launch and both ranks leave through the bounded wait
    (MA_ERR_COMM after 2 s, no hang: seen once in five runs of the 4-context form).  That outcome skips; anything else must
    be the single context's bits.  Ranks on different GPUs or in different processes have their own queues."""
    res = _rehearse("--mode", "threads", "--ranks", "2", "--points", "300000", "--iters", "50")
    print(res)
    if not res["ok"] and "did not arrive within" in res.get("error", ""):
        pytest.skip("the runtime mapped both contexts' streams to one hardware queue (one-process rehearsal artefact)")
    assert res["ok"] and res["bitwise_equal_to_one_context"] and res["iterations"] == 50, res


def test_p2p_exchange_times_out_instead_of_hanging(lib_built):
    """A rank whose peer never writes its rows: the solving wave gives up after the exchange's timeout (300 ms here),
    ma_icp_sharded returns MA_ERR_COMM, a second call refuses until the exchange is set up again, and the context still
    aligns on its own."""
    res = _rehearse("--mode", "timeout")
    print(res)
    assert res["ok"], res


# ----------------------------------------------------------------------------- row N4: the map builder's frames in one launch
def test_map_add_depth_batch_equals_single_calls_bitwise(engine):
    """ma_map_add_depth_batch: n frames, one upload, ONE k_unproject launch (blockIdx.y = frame).  The map it leaves must
    be the map of n ma_map_add_depth calls bit for bit (same voxel centroids from ma_map_finish), for float32 and uint16
    frames, host and device-resident input, with holes, and a batch appended behind single frames."""
    import torch
    K = (250.0, 250.0, 159.5, 119.5)
    rects = syn.carriage_rectangles(np.random.default_rng(8))
    poses = []
    for k, x in enumerate((-3.0, 1.0, 2.5, 5.0, 6.0, 7.5)):
        T = syn.default_camera_pose()
        T[:3, 3] = (x, 1.4 + 0.05 * k, 1.3)
        T[:3, :3] = T[:3, :3] @ syn.rotation_about_axis([0, 1, 0], 0.07 * k)
        poses.append(T)
    frames = np.stack([syn.render_depth(rects, T, 320, 240, *K) for T in poses])
    frames[2, 50:90, 100:180] = 0.0                                          # holes
    frames[4, 10, 10] = np.nan
    for cast in (lambda a: a, lambda a: np.clip(np.round(np.nan_to_num(a)), 0, 65535).astype(np.uint16)):
        fr = cast(frames)
        engine.map_begin(0.03, 6 * 320 * 240)
        single = [engine.map_add_depth(d, K, T) for d, T in zip(fr, poses)]
        ref = engine.map_finish()
        engine.map_begin(0.03, 6 * 320 * 240)
        counts = engine.map_add_depth_batch(fr, K, poses)
        assert list(counts) == single
        got = engine.map_finish()
        assert np.array_equal(got, ref)
        # device-resident frames; and a batch behind two single frames
        engine.map_begin(0.03, 6 * 320 * 240)
        engine.map_add_depth(fr[0], K, poses[0])
        engine.map_add_depth(fr[1], K, poses[1])
        dev = torch.as_tensor(np.ascontiguousarray(fr[2:])).to("cuda:0")
        torch.cuda.synchronize()
        counts2 = engine.map_add_depth_batch(dev, K, poses[2:])
        assert list(counts2) == single[2:]
        assert np.array_equal(engine.map_finish(), ref)
    pts = np.concatenate([orc.unproject_oracle(d, K, T, 1000.0)[0] for d, T in zip(frames, poses)])
    engine.map_begin(0.03, 6 * 320 * 240)
    assert int(engine.map_add_depth_batch(frames, K, poses).sum()) == len(pts)
    cloud = engine.map_finish()
    refo = orc.voxel_centroids_oracle(pts, 0.03, np.round(poses[0][:3, 3]))
    assert cloud.shape == refo.shape and np.abs(cloud.astype(np.float64) - refo).max() <= 2e-5
    with pytest.raises(mapalign.MapAlignError) as e:                         # capacity is checked for the whole batch
        engine.map_begin(0.03, 5 * 320 * 240)
        engine.map_add_depth_batch(frames, K, poses)
    assert e.value.status == mapalign._lib.MA_ERR_TOO_LARGE


def test_own_scan_and_64_bit_sort_on_a_large_map_and_a_5m_cloud(engine):
    """Round 5 took hipCUB / rocPRIM out of the product: the map builder's 64-bit voxel-key sort and its scan, the rank map
    of a cloud with holes and the cell sort of clouds above 4M points are the library's own three-kernel passes now.  A map
    of 24 frames (7.4M slots, keys of ~33 bits, with holes) against the NumPy restatement; and a 5M-point source cloud's
    sort through ma_nn against a brute-force check of a sample."""
    depth, K, T = syn.make_depth_frame(0, width=640, height=480, fx=500.0, fy=500.0, cx=320.0, cy=240.0)
    poses = []
    for k in range(24):
        P = T.copy()
        P[:3, 3] += [0.35 * k, 0.02 * (k % 3), 0.0]
        poses.append(P)
    frames = np.repeat(depth[None], 24, axis=0)
    engine.map_begin(0.02, 24 * depth.size)
    counts = engine.map_add_depth_batch(frames, K, poses)
    cloud = engine.map_finish()
    pts = np.concatenate([orc.unproject_oracle(depth, K, P, 1000.0)[0] for P in poses])
    assert int(counts.sum()) == len(pts)
    ref = orc.voxel_centroids_oracle(pts, 0.02, np.round(poses[0][:3, 3]))
    assert cloud.shape == ref.shape
    # the same voxels in the same order; centroids to fp32 storage precision — except where one of the 7.4M points sits
    # within an ulp of a voxel face and the device's fp32 coordinate (its own association of the pose arithmetic) falls on
    # the other side than the oracle's: that moves two centroids by a fraction of the voxel.  The carriage's walls are
    # planes at round coordinates (z = 2.5 m = face 125 of the 2 cm voxels), so such points come in rows: ~130 of 724k voxels.
    off = np.abs(cloud.astype(np.float64) - ref).max(axis=1)
    assert (off > 3e-5).sum() <= 1e-3 * len(off) and off.max() < 0.02, ((off > 3e-5).sum(), off.max())
    # the rank map (own scan) of a cloud with holes: get_source returns the valid pixels in row-major order
    gold, _, _, _ = syn.make_pair(2, 200_000)
    engine.set_gold(gold)
    n = engine.set_source_from_depth(depth, K, T, 1000.0)
    opts, _ = orc.unproject_oracle(depth, K, T, 1000.0)
    got = engine.get_source()
    assert n == len(opts) and got.shape == (n, 3) and np.abs(got - opts).max() <= 8e-6
    # a 5M-point cloud: the own sort beyond the old 4M hand-over to the library
    rng = np.random.default_rng(12)
    big = (gold[rng.integers(0, len(gold), 5_000_000)] + rng.normal(0, 0.01, (5_000_000, 3))).astype(np.float32)
    engine.set_source(big)
    idx, d2 = engine.nn(None, 0.10)
    sample = rng.integers(0, len(big), 400)
    bi, bd = orc.nn_bruteforce(gold, big[sample].astype(np.float64))
    m = bd <= 0.10 ** 2
    assert np.array_equal(idx[sample][m], bi[m]) or np.allclose(d2[sample][m], bd[m], rtol=1e-4)


def test_a_stopped_sharded_loop_stops_enqueueing_at_the_same_batch_whatever_its_list_settings(lib_built):
    """where a stopped sharded loop stops ENQUEUEING must not depend on anything rank-local — round 4
    tied the slots (which carry the exit decision) to `lists_wanted`, so a rank with lists switched off kept enqueueing
    max_iter iterations (each with a collective) while a peer with lists broke off at the next batch.  A loop that stops
    degenerate at iteration 0 under tol = 0, through the sharded loop (1-rank peer-to-peer exchange): 8 launches enqueued
    (two planning batches) with lists on, with lists off for the run, and with lists off for the context."""
    gold, comp, _, _ = syn.make_pair(5, 60_000)
    far = comp + np.array([0.0, 40.0, 0.0], dtype=np.float32)          # nothing within the gate: W = 0 at iteration 0
    counts = []
    for setting in ("on", "run_off", "ctx_off"):
        with mapalign.MapAligner(0) as eng:
            if setting == "ctx_off":
                eng.list_config(enabled=False)
            eng.set_gold(gold)
            eng.set_source(far)
            eng.p2p_init([eng.p2p_export()], 0)
            with pytest.raises(mapalign.MapAlignError) as e:
                eng.icp_sharded(40, 0.0, 0.10, lists=False if setting == "run_off" else None)
            assert e.value.status == mapalign._lib.MA_ERR_DEGENERATE
            counts.append(eng.planner_diag()["enqueued"])
            # and the context is not left broken: a good cloud aligns through the same exchange afterwards
            eng.set_source(comp)
            assert eng.icp_sharded(20, 0.0, 0.10).iterations == 20
    assert counts == [8, 8, 8], counts


# ----------------------------------------------------------------------------- row a5 as integer work: bit-exact
@pytest.mark.parametrize("weighted", [False, True])
def test_the_17_integer_sums_equal_an_exact_arithmetic_restatement(engine, weighted):
    """The sums of row a5 are 64-bit integers since round 5, and integer work has a bit-exact bar: the 17 integers the
    device forms for one iteration (ma_step_accumulate) against oracle/exact_sums.py — rational arithmetic, every fma one
    exact product-and-sum and one correctly rounded conversion — on the device's own stored coordinates, correspondences
    and centred-frame transform.  All 17 must be EQUAL, with and without weights, gated, for a general transform."""
    import torch
    from oracle import exact_sums as xs
    gold, comp, R0, t0 = syn.make_pair(21, 30_000, 2500)          # a dense enough map for ~2000 matches inside the gate
    w = np.random.default_rng(4).uniform(0.25, 3.0, len(comp)).astype(np.float32) if weighted else None
    engine.set_gold(gold)
    engine.set_source(comp, w)
    T = np.eye(4)                                                  # a general transform near the pair's truth: most points match
    T[:3, :3] = syn.rotation_about_axis([0.3, -0.5, 0.81], np.deg2rad(0.2)) @ R0
    T[:3, 3] = t0 + [0.01, -0.008, 0.005]
    gate = 0.10
    origin = np.asarray(engine.grid_info()["origin"], dtype=np.float64)
    # the state as the library centres it (host_math.h::world_to_centred, the same operations in the same order)
    R = T[:3, :3].copy()
    t = np.array([((R[r, 0] * origin[0] + R[r, 1] * origin[1]) + R[r, 2] * origin[2]) + T[r, 3] - origin[r] for r in range(3)])
    g_c = (engine.get_gold() - origin).astype(np.float32)             # exact: the stored fp32 coordinates
    s_c = (engine.get_source_stored() - origin).astype(np.float32)
    assert np.array_equal(g_c.astype(np.float64) + origin, engine.get_gold())
    idx, d2 = engine.nn(T, gate)
    match = np.where(idx == 0xFFFFFFFF, -1, idx.astype(np.int64))
    engine.set_sum_bounds(engine.sum_bounds())
    e = engine.sum_scale(gate)
    acc = torch.zeros(17, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    engine.state_set(T)
    engine.step_accumulate(acc.data_ptr(), gate)
    engine.sync()
    dev = [int(v) for v in acc.cpu().numpy()]
    ref = xs.sums(R, t, s_c, w, g_c, match, e)
    assert (match >= 0).sum() > 1000, (match >= 0).sum()
    assert dev == ref, [(a, d - r) for a, (d, r) in enumerate(zip(dev, ref)) if d != r]
    if not weighted:
        assert dev[0] == int((match >= 0).sum()) << int(e[0])


def test_order_of_the_comparison_cloud_does_not_change_a_bit_at_full_size(engine):
    """A size-independent law the exact sums buy, checked at BASELINE's full size (1M points, 50 iterations, list launches
    in use): the comparison cloud in ANY order gives the same transform, the same per-iteration sums and the same stop, bit
    for bit — the device sorts the scan into bricks of its own, a correspondence's 17 terms depend on the point, its match
    and the transform only, and integer addition does not care in which chunk, workgroup or launch a term is added.
    (The gold map's order is not free in the same way: an exact distance tie goes to the smaller gold position.)"""
    gold, comp, _, _ = syn.make_pair(0, 1_000_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    a = engine.icp(50, 0.0, 0.10, with_trace=True)
    uses = engine.list_stats()["uses"] if hasattr(engine, "list_stats") else None
    rng = np.random.default_rng(77)
    for rep in range(2):
        engine.set_source(comp[rng.permutation(len(comp))])
        b = engine.icp(50, 0.0, 0.10, with_trace=True)
        assert np.array_equal(a.T, b.T) and a.rmse == b.rmse and a.inliers == b.inliers
        assert np.array_equal(a.trace[:, :18], b.trace[:, :18])          # the 17 sums and the rmse of every iteration
    # a weighted, ungated, early-stopping run of a ragged size as well
    gold, comp, _, _ = syn.make_pair(5, 300_000, 212_345)
    w = rng.uniform(0.1, 4.0, len(comp)).astype(np.float32)
    engine.set_gold(gold)
    engine.set_source(comp, w)
    a = engine.icp(60, 1e-7, np.inf, with_trace=True)
    perm = rng.permutation(len(comp))
    engine.set_source(comp[perm], w[perm])
    b = engine.icp(60, 1e-7, np.inf, with_trace=True)
    assert a.iterations == b.iterations and np.array_equal(a.T, b.T) and np.array_equal(a.trace[:, :18], b.trace[:, :18])
    print(f"1M pair: 2 permutations bitwise equal over 50 iterations (lists {uses}); weighted ungated 212k: stop at {a.iterations} on both")


@pytest.mark.parametrize("gate,weighted", [(np.inf, False), (0.25, True)])
def test_uneven_shards_ungated_and_weighted_equal_one_context_bitwise(lib_built, gate, weighted):
    """The other bounds of the exact sums: an UNGATED run scales its terms from the comparison cloud's radius and the start
    (so the shards must agree on rs), a weighted one from the largest weight (so they must agree on wmax).  Three uneven
    random shards of a 90k-point cloud with the whole cloud's bounds installed, stepped with the sums added as integers:
    the transform after 20 iterations is the single context's, bit for bit."""
    import torch
    from mapalign import dist as mdist
    gold, comp, R0, t0 = syn.make_pair(13, 120_000, 90_000)
    rng = np.random.default_rng(8)
    w = rng.uniform(0.05, 7.5, len(comp)).astype(np.float32) if weighted else None
    iters = 20
    with mapalign.MapAligner(0) as full:
        full.set_gold(gold)
        full.set_source(comp, w)
        ref = full.icp(iters, 0.0, gate, with_trace=True)
        own = full.sum_bounds()
    owner = rng.choice(3, size=len(comp), p=[0.55, 0.35, 0.10])
    engs, accs = [], []
    for r in range(3):
        idx = np.flatnonzero(owner == r)
        a = mapalign.MapAligner(0)
        a.set_gold(gold)
        a.set_source(comp[idx], None if w is None else w[idx])
        he = mdist.HipShardEngine(a, gate, 0.0)
        he.bind_current_stream()
        engs.append(he)
        accs.append(he.new_acc())
    whole = mapalign.MapAligner.combine_sum_bounds([he.aligner.sum_bounds() for he in engs])
    assert np.array_equal(whole, own)          # (sum n, max wmax, max rs) over the shards: the whole cloud's own bounds, exactly
    for he in engs:
        he.aligner.set_sum_bounds(whole)
        he.state_set(None)
    for _ in range(iters):
        for he, acc in zip(engs, accs):
            he.accumulate_into(acc)
        total = accs[0] + accs[1] + accs[2]
        for he, acc in zip(engs, accs):
            acc.copy_(total)
            he.solve_from(acc)
    states = [he.state_get() for he in engs]
    for he in engs:
        he.aligner.use_own_stream()
        he.aligner.close()
    for T, info in states:
        assert np.array_equal(T, ref.T) and info["rmse"] == ref.rmse and info["inliers"] == ref.inliers
