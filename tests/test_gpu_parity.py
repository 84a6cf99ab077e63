"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(ctypes, mapalign.MapAligner), against the CPU oracle on identical seeded inputs, against
the committed golden fixtures, and — at BASELINE.json's full 1M-point size — through
size-independent properties (align(A, T.A) = T, cell-size invariance, determinism).

Tolerances (BASELINE.json:north_star): ||R - R_cpu||_F <= 1e-4, ||t - t_cpu||_2 <= 1 mm.
Index work (nearest-neighbour ids) must be exact, up to exact distance ties.
"""
import numpy as np
import pytest

import mapalign
from mapalign import synthetic as syn
from oracle import icp_oracle as orc

from conftest import R_TOL_FRO, T_TOL_M

pytestmark = pytest.mark.gpu

NO_MATCH = 0xFFFFFFFF
# inputs are re-centred and stored as fp32 on the device (<= 0.5 ulp(10 m) = 0.5 um per
# coordinate); fp64 sums over >= 1e3 points therefore agree to ~1e-8 relative
ACC_RTOL = 2e-7


# coordinates are stored re-centred in fp32 on the device: each one is within 0.5 ulp(10 m)
# ~ 0.5 um of the oracle's, so distances agree to a few um (NOT to a relative tolerance:
# d^2 of a 1 mm match moves by 2 d delta ~ 1e-9 m^2)
DIST_ATOL_M = 5e-6


def assert_nn_matches(idx, d2, oidx, od2, gate=np.inf):
    """Exact index parity; a differing index is accepted only as a tie at storage precision."""
    matched = od2 <= gate ** 2
    borderline = np.abs(np.sqrt(od2) - gate) < 2 * DIST_ATOL_M
    ok_unmatched = (idx == NO_MATCH) == (~matched)
    assert (ok_unmatched | borderline).all()
    m = matched & (idx != NO_MATCH)
    differ = m & (idx.astype(np.int64) != oidx)
    if m.any():
        # second, tolerance-based check against the oracle on the CALLER's fp64 inputs (which the device
        # rounds to fp32 about the centring origin): distances to 5 um, and the few indices that differ are
        # near-ties at that precision.  The exact index check is assert_nn_exact_on_stored.
        assert np.abs(np.sqrt(d2[m].astype(np.float64)) - np.sqrt(od2[m])).max() <= DIST_ATOL_M
        assert differ.mean() < 1e-3


def assert_nn_exact_on_stored(engine, T, gate, idx, d2, use_tree=False):
    """EXACT index parity (bar 1: bit-exact index work) on the coordinates the device actually stores.

    ``ma_get_gold`` / ``ma_get_source_stored`` return both clouds as stored (fp32 about the centring
    origin), so the oracle below works on bit for bit the device's inputs: the query is the fp32 image
    of the fp64 transform of the stored point (what k_nn_accum forms), the candidates are the stored
    gold points, distances in fp64.  Every index must then equal the device's.  The only admissible
    difference is a TIE IN THE DEVICE'S OWN ARITHMETIC: the device compares fp32 distances, so where the
    oracle's neighbour and the device's are closer than 4 ulp(fp32) in squared distance either may win;
    each such case is checked individually (no percentage of free mismatches)."""
    gi = engine.grid_info()
    o = gi["origin"]
    G = engine.get_gold() - o                     # exact: the stored fp32 values
    S = engine.get_source_stored() - o
    assert np.array_equal(G, G.astype(np.float32).astype(np.float64)) and np.array_equal(S, S.astype(np.float32).astype(np.float64))
    R, t = (np.eye(3), np.zeros(3)) if T is None else (np.asarray(T)[:3, :3], np.asarray(T)[:3, 3])
    tc = R @ o + t - o                              # host_math.h: world_to_centred
    P = (S @ R.T + tc).astype(np.float32)           # the fp32 image the search runs on
    P64 = P.astype(np.float64)
    if use_tree:
        oidx, od2 = orc.GoldIndex(G, n_jobs=-1).query(P64)
    else:
        oidx, od2 = orc.nn_bruteforce(G, P64)
    g2 = np.float32(gate) ** 2 if np.isfinite(gate) else np.float32(np.inf)

    def d2_f32(j):                                  # the device's arithmetic: fp32 differences, fp32 sum of squares
        d = P - G[j].astype(np.float32)
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)

    od2_dev = d2_f32(oidx)
    ulp4 = 4.0 * np.spacing(np.maximum(od2_dev, np.float32(1e-30)))
    near_gate = np.abs(od2_dev.astype(np.float64) - float(g2)) <= ulp4 if np.isfinite(gate) else np.zeros(len(P), bool)
    matched = od2_dev <= g2
    # the gate decision agrees except within 4 ulp of the gate itself
    assert (((idx != NO_MATCH) == matched) | near_gate).all()
    m = matched & (idx != NO_MATCH)
    differ = np.flatnonzero(m & (idx.astype(np.int64) != oidx))
    if len(differ):
        mine = d2_f32(idx.astype(np.int64).clip(0, len(G) - 1))[differ]
        assert (np.abs(mine.astype(np.float64) - od2_dev[differ]) <= ulp4[differ]).all(), \
            f"{len(differ)} indices differ from the oracle on the stored coordinates and are not fp32 ties"
    # reported squared distances: the device's fp32 value of its own winner (1 ulp of contraction freedom)
    if m.any():
        own = d2_f32(idx.astype(np.int64).clip(0, len(G) - 1))
        assert (np.abs(d2[m].astype(np.float64) - own[m]) <= ulp4[m] + 4.0 * np.spacing(own[m])).all()
    return len(differ)


def acc_close(a, b):
    scale = np.abs(b).max()
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= ACC_RTOL * scale


# ----------------------------------------------------------------------------- row a4
@pytest.mark.parametrize("n_gold,n_comp", [(10_000, 10_000), (4097, 1023), (1, 300), (257, 1), (5000, 7777)])
def test_nn_exact_vs_bruteforce(engine, n_gold, n_comp):
    gold, comp, _, _ = syn.make_pair(30, max(n_gold, 2), max(n_comp, 2))
    gold, comp = gold[:n_gold], comp[:n_comp]
    engine.set_gold(gold)
    engine.set_source(comp)
    oidx, od2 = orc.nn_bruteforce(gold, comp)
    idx, d2 = engine.nn(None, np.inf)               # ungated: ring expansion to the whole grid
    assert_nn_matches(idx, d2, oidx, od2)
    assert_nn_exact_on_stored(engine, None, np.inf, idx, d2)
    idx, d2 = engine.nn(None, 0.10)                 # gated
    assert_nn_matches(idx, d2, oidx, od2, gate=0.10)
    assert_nn_exact_on_stored(engine, None, 0.10, idx, d2)


@pytest.mark.parametrize("cell", [0.0, 0.01, 0.1, 1.0, 50.0])
def test_nn_independent_of_cell_size(engine, cell):
    gold, comp, R0, t0 = syn.make_pair(31, 20_000, 15_000)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R0, t0 + [0.3, -0.2, 0.1]
    engine.set_gold(gold, cell)
    engine.set_source(comp)
    p_t = comp.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    oidx, od2 = orc.GoldIndex(gold, n_jobs=4).query(p_t)
    idx, d2 = engine.nn(T, 0.25)
    assert_nn_matches(idx, d2, oidx, od2, gate=0.25)
    assert_nn_exact_on_stored(engine, T, 0.25, idx, d2, use_tree=True)
    idx, d2 = engine.nn(T, np.inf)
    assert_nn_matches(idx, d2, oidx, od2)
    assert_nn_exact_on_stored(engine, T, np.inf, idx, d2, use_tree=True)


def test_nn_far_outside_the_grid_and_duplicates(engine):
    rng = np.random.default_rng(32)
    gold = rng.uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    gold[100:200] = gold[100]                          # 100 exact duplicates
    comp = np.concatenate([rng.uniform(-1, 1, size=(500, 3)), rng.uniform(-1, 1, size=(500, 3)) * 40.0,
                           gold[100:101].repeat(5, 0)]).astype(np.float32)
    engine.set_gold(gold)
    engine.set_source(comp)
    oidx, od2 = orc.nn_bruteforce(gold, comp)
    idx, d2 = engine.nn(None, np.inf)
    m = np.ones(len(comp), bool)
    # duplicates: any of the duplicate ids is a correct answer
    dup = (oidx >= 100) & (oidx < 200)
    assert np.abs(np.sqrt(d2.astype(np.float64)) - np.sqrt(od2)).max() <= 2e-5   # coordinates up to 40 m
    assert (idx[~dup & m].astype(np.int64) != oidx[~dup & m]).mean() < 1e-3
    assert ((idx[dup] >= 100) & (idx[dup] < 200)).all()
    idx, d2 = engine.nn(None, 0.05)
    assert ((idx == NO_MATCH) == (od2 > 0.05 ** 2)).all()


def _fuzz_cloud(rng, n, kind):
    """Seeded shapes that stress the grid: a volume, a thin slab, a line, tight clusters."""
    if kind == 0:
        return rng.uniform(-1.0, 1.0, size=(n, 3)) * rng.uniform(0.2, 4.0, size=3)
    if kind == 1:                                              # slab: one axis nearly flat
        p = rng.uniform(-2.0, 2.0, size=(n, 3))
        p[:, rng.integers(0, 3)] *= 1e-3
        return p
    if kind == 2:                                              # line along a random direction
        d = rng.normal(size=3)
        return rng.uniform(-3.0, 3.0, size=(n, 1)) * (d / np.linalg.norm(d)) + rng.normal(0, 1e-3, size=(n, 3))
    centres = rng.uniform(-2.0, 2.0, size=(max(n // 200, 1), 3))    # clusters: crowded cells, long x-runs
    return centres[rng.integers(0, len(centres), size=n)] + rng.normal(0, 0.01, size=(n, 3))


def test_nn_fuzz_shapes_sizes_gates_cells(engine):
    """48 seeded cases through the C ABI against brute force: sizes 1..6000 (ragged wavefronts,
    queue overflow in crowded cells, empty surroundings), random rigid motion between the clouds,
    gates from 2 cm to none, cell edges from 1 cm to 2 m and density-driven."""
    rng = np.random.default_rng(77)
    for case in range(48):
        m, n = int(rng.integers(1, 6000)), int(rng.integers(1, 6000))
        gold = _fuzz_cloud(rng, m, case % 4).astype(np.float32)
        comp = _fuzz_cloud(rng, n, int(rng.integers(0, 4))).astype(np.float32)
        if case % 3 == 0:                                      # overlapping clouds (most neighbours close)
            comp[: min(n, m)] = gold[: min(n, m)] + rng.normal(0, 0.005, size=(min(n, m), 3)).astype(np.float32)
        T = np.eye(4)
        T[:3, :3] = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0, 0.3))
        T[:3, 3] = rng.normal(0, 0.05, size=3)
        gate = [0.02, 0.1, 0.5, np.inf][int(rng.integers(0, 4))]
        cell = [0.0, 0.0, 0.01, 0.05, 0.3, 2.0][int(rng.integers(0, 6))]
        engine.set_gold(gold, cell)
        engine.set_source(comp)
        p_t = comp.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        oidx, od2 = orc.nn_bruteforce(gold, p_t)
        idx, d2 = engine.nn(T, gate)
        assert_nn_matches(idx, d2, oidx, od2, gate=gate)
        assert_nn_exact_on_stored(engine, T, gate, idx, d2)        # exact on the stored coordinates, every case


@pytest.mark.parametrize("cell", [50.0, 0.3, 0.02])
def test_exact_duplicates_resolve_to_the_smallest_index_at_every_size(engine, cell):
    """Equal distances resolve to the smaller position of the cell-sorted gold array, and the cell sort is STABLE, so
    among exact duplicates the caller's smallest index wins — as in the oracle's first-minimum rule.  Sizes around
    the tile of the radix sort (2048 keys per workgroup), one cell for everything (every key equal: the sort must be
    the identity), a few cells, many cells."""
    rng = np.random.default_rng(5)
    for n_unique in (1, 683, 1365, 1366, 2048, 23456):
        base = rng.uniform(-1, 1, size=(n_unique, 3)).astype(np.float32)
        order = np.concatenate([rng.permutation(n_unique) for _ in range(3)])     # every point three times
        gold = base[order]
        first = np.full(n_unique, len(gold), dtype=np.int64)
        np.minimum.at(first, order, np.arange(len(gold)))
        engine.set_gold(gold, cell)
        engine.set_source(base)
        idx, d2 = engine.nn(None, np.inf)
        assert (d2 == 0).all()
        assert (idx.astype(np.int64) == first).all()


def test_nn_gated_queries_in_a_sparse_grid(engine):
    """The distance-to-occupied table (one byte per cell, three separable passes) dismisses gated queries that have
    nothing within reach: a few small clusters in a big, mostly empty box (grids with nx not a multiple of 4, one
    flat axis, one cell thick), queries all over the box, gates from 1.5 to 9 cells.  A table entry that is too
    large would dismiss a query that has a neighbour inside the gate: every result is checked against brute force."""
    rng = np.random.default_rng(2024)
    for case, (ext, cell) in enumerate([((4.0, 3.1, 2.3), 0.1), ((5.0, 0.05, 3.0), 0.07), ((1.3, 2.9, 0.02), 0.05),
                                        ((6.0, 6.0, 6.0), 0.25), ((0.9, 0.9, 0.9), 0.3)]):
        ext = np.asarray(ext)
        centres = rng.uniform(-0.5, 0.5, size=(12, 3)) * ext
        gold = (centres[rng.integers(0, 12, size=600)] + rng.normal(0, 0.3 * cell, size=(600, 3))).astype(np.float32)
        gold = np.clip(gold, -0.5 * ext, 0.5 * ext).astype(np.float32)
        gold[:8] = (np.array([[sx, sy, sz] for sx in (-.5, .5) for sy in (-.5, .5) for sz in (-.5, .5)]) * ext).astype(np.float32)  # corners span the box
        comp = (rng.uniform(-0.5, 0.5, size=(5000, 3)) * ext).astype(np.float32)
        engine.set_gold(gold, cell)
        engine.set_source(comp)
        oidx, od2 = orc.nn_bruteforce(gold, comp.astype(np.float64))
        for cells_wide in (1.5, 2.5, 4.0, 6.5, 9.0):
            gate = cells_wide * cell
            idx, d2 = engine.nn(None, gate)
            assert_nn_matches(idx, d2, oidx, od2, gate=gate)
            assert_nn_exact_on_stored(engine, np.eye(4), gate, idx, d2)


# ----------------------------------------------------------------------------- rows a4-a6
@pytest.mark.parametrize("gate", [0.10, 0.35, np.inf])
def test_one_iteration_accumulators_and_solve(engine, gate):
    gold, comp, _, _ = syn.make_pair(33, 12_000, 9_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    index = orc.GoldIndex(gold, n_jobs=4)
    T0 = np.eye(4)
    T0[:3, :3] = syn.rotation_about_axis([0, 0, 1], 0.004)
    T0[:3, 3] = [0.01, 0.02, -0.01]
    oacc, _, _, Rn, tn, ok = orc.icp_step(index, comp, T0[:3, :3], T0[:3, 3], gate)
    acc, Tn = engine.icp_step(T0, gate)
    assert ok and acc_close(acc, oacc)
    assert acc[0] == oacc[0]                                  # inlier count is integer work: exact
    eR, et = orc.rt_error(Tn[:3, :3], Tn[:3, 3], Rn, tn)
    assert eR <= 1e-7 and et <= 1e-6


def test_weights(engine, golden):
    g = golden("pair_weighted.npz")
    engine.set_gold(g["gold"])
    engine.set_source(g["comp"], g["weights"])
    res = engine.icp(int(g["iters"]), 0.0, float(g["gate"]), with_trace=True)
    eR, et = orc.rt_error(res.R, res.t, g["R"], g["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    for k in range(int(g["iters"])):
        assert acc_close(res.trace[k, :17], g["acc"][k]), k


def test_icp_loop_fuzz_vs_oracle(engine):
    """16 seeded small alignments (volumes, slabs, clusters; random weights on half of them;
    gates 5-50 cm) run for 12 iterations through ma_icp — remembered matches, the
    distance-to-occupied dismissal and the longest-first schedule all engaged — against the
    oracle loop: per-iteration inlier weight and accumulators, and the final transform."""
    rng = np.random.default_rng(78)
    for case in range(16):
        m = int(rng.integers(800, 9000))
        gold = _fuzz_cloud(rng, m, [0, 1, 3][case % 3]).astype(np.float32)
        n = int(rng.integers(500, m + 1))
        sel = rng.permutation(m)[:n]
        Rg = syn.rotation_about_axis(rng.normal(size=3), rng.uniform(0.0, 0.03))
        tg = rng.normal(0, 0.02, size=3)
        comp = ((gold[sel].astype(np.float64) + rng.normal(0, 0.002, size=(n, 3)) - tg) @ Rg).astype(np.float32)
        extra = _fuzz_cloud(rng, n // 4 + 1, 0).astype(np.float32) * 3.0      # points with nothing to match
        comp = np.concatenate([comp, extra])
        w = rng.uniform(0.2, 2.0, size=len(comp)).astype(np.float32) if case % 2 else None
        gate = [0.05, 0.1, 0.2, 0.5][case % 4]
        engine.set_gold(gold)
        engine.set_source(comp, w)
        res = engine.icp(12, 0.0, gate, with_trace=True)
        ref = orc.icp_oracle(gold, comp, max_iter=12, max_corr_dist=gate, weights=w, n_jobs=4)
        assert ref["ok"] and res.iterations == 12
        eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
        # (ten times inside the parity bars: a near-tie resolved differently at fp32 storage
        # precision moves an ill-conditioned case, e.g. a few tight clusters, by ~1e-6)
        assert eR <= 0.1 * R_TOL_FRO and et <= 0.01 * T_TOL_M, (case, eR, et)
        assert acc_close(res.trace[0, :17], ref["trace"][0]["acc"]), case       # same start: same sums
        for k in range(12):
            assert abs(res.trace[k, 0] - ref["trace"][k]["acc"][0]) <= 2e-3 * ref["trace"][k]["acc"][0], (case, k)


# ----------------------------------------------------------------------------- rows a4-a7
def test_config1_10k_pair_vs_oracle_and_truth(engine):
    gold, comp, R0, t0 = syn.make_pair(0, 10_000, resample=False)       # BASELINE configs[0]
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(50, 0.0, 0.10, with_trace=True)
    ref = orc.icp_oracle(gold, comp, max_iter=50, max_corr_dist=0.10, n_jobs=4)
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    eR, et = orc.rt_error(res.R, res.t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert res.iterations == 50 and res.inliers == ref["inliers"]
    for k in (0, 1, 5, 20):
        assert acc_close(res.trace[k, :17], ref["trace"][k]["acc"]), k


def test_resampled_pair_vs_oracle(engine):
    gold, comp, _, _ = syn.make_pair(1, 60_000, 50_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(30, 0.0, 0.10, with_trace=True)
    ref = orc.icp_oracle(gold, comp, max_iter=30, max_corr_dist=0.10, n_jobs=8)
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert abs(res.inliers - ref["inliers"]) <= 2          # fp32 vs fp64 gate on the boundary
    assert abs(res.rmse - ref["rmse"]) <= 1e-6


def test_golden_pair(engine, golden):
    g = golden("pair_a.npz")
    engine.set_gold(g["gold"])
    engine.set_source(g["comp"])
    idx, d2 = engine.nn(None, np.inf)
    assert_nn_matches(idx, d2, g["nn0"], g["d20"])
    res = engine.icp(int(g["iters"]), 0.0, float(g["gate"]), with_trace=True)
    eR, et = orc.rt_error(res.R, res.t, g["R"], g["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert res.trace.shape[0] == int(g["iters"])
    for k in range(int(g["iters"])):
        assert acc_close(res.trace[k, :17], g["acc"][k]), k
    assert np.allclose(res.trace[:, 17], g["rmse"], atol=1e-7)


def test_align_function_surface():
    gold, comp, R0, t0 = syn.make_pair(2, 8000, resample=False)
    R, t = mapalign.align(gold, comp)
    assert R.shape == (3, 3) and t.shape == (3,) and R.dtype == np.float64
    eR, et = orc.rt_error(R, t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    R2, t2, info = mapalign.align(gold.astype(np.float64), comp.tolist(), return_info=True, max_iter=5)
    assert info["iterations"] == 5 and info["trace"].shape == (5, 20)


def test_init_transform_and_tuple_form(engine):
    gold, comp, R0, t0 = syn.make_pair(3, 9000, resample=False)
    engine.set_gold(gold)
    engine.set_source(comp)
    # start from the true transform: must stay there
    res = engine.icp(3, 0.0, 0.10, init=(R0, t0))
    eR, et = orc.rt_error(res.R, res.t, R0, t0)
    assert eR <= 1e-6 and et <= 1e-5
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R0, t0
    res2 = engine.icp(3, 0.0, 0.10, init=T)
    assert np.array_equal(res.T, res2.T)


def test_tol_early_stop_matches_oracle(engine):
    gold, comp, _, _ = syn.make_pair(4, 6000, resample=False)
    engine.set_gold(gold)
    engine.set_source(comp)
    ref = orc.icp_oracle(gold, comp, max_iter=50, tol=1e-7, max_corr_dist=0.3, n_jobs=4)
    res = engine.icp(50, 1e-7, 0.3)
    assert res.converged and ref["converged"]
    assert abs(res.iterations - ref["iterations"]) <= 1 and res.iterations < 50
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M


def test_ungated_far_start(engine):
    gold, comp, R0, t0 = syn.make_pair(5, 5000, resample=False, angle_deg=4.0, t0=np.array([0.5, -0.3, 0.2]))
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(40, 0.0, np.inf)
    ref = orc.icp_oracle(gold, comp, max_iter=40, max_corr_dist=np.inf, n_jobs=4)
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M


def test_bitwise_deterministic(engine):
    gold, comp, _, _ = syn.make_pair(6, 50_000, 40_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    a = engine.icp(10, 0.0, 0.10, with_trace=True)
    b = engine.icp(10, 0.0, 0.10, with_trace=True)
    assert np.array_equal(a.T, b.T) and np.array_equal(a.trace, b.trace)
    with mapalign.MapAligner(0) as other:                   # fresh context, fresh grid build
        other.set_gold(gold)
        other.set_source(comp)
        c = other.icp(10, 0.0, 0.10, with_trace=True)
    assert np.array_equal(a.T, c.T) and np.array_equal(a.trace, c.trace)


def test_work_counters_do_not_change_results(engine):
    gold, comp, _, _ = syn.make_pair(7, 20_000, 20_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    a = engine.icp(5, 0.0, 0.10, with_trace=True)
    b = engine.icp(5, 0.0, 0.10, with_trace=True, count_work=True)
    assert np.array_equal(a.T, b.T) and np.array_equal(a.trace[:, :18], b.trace[:, :18])
    assert (a.trace[:, 18:] == 0).all() and (b.trace[:, 18] > 0).all() and (b.trace[:, 19] > 0).all()


# ----------------------------------------------------------------------------- edge cases
def test_error_paths(engine):
    gold, comp, _, _ = syn.make_pair(8, 2000, resample=False)
    E = mapalign.MapAlignError
    with pytest.raises(E) as e:
        engine.set_source(comp)                              # source before gold
    assert e.value.status == -5
    with pytest.raises(E) as e:
        engine.set_gold(np.zeros((0, 3), np.float32))
    assert e.value.status == -4
    bad = gold.copy()
    bad[5, 1] = np.nan
    with pytest.raises(E) as e:
        engine.set_gold(bad)
    assert e.value.status == -1
    engine.set_gold(gold)
    with pytest.raises(E) as e:
        engine.icp()                                         # gold but no source
    assert e.value.status == -5
    with pytest.raises(E) as e:
        engine.set_source(np.zeros((0, 3), np.float32))
    assert e.value.status == -4
    bad = comp.copy()
    bad[0, 0] = np.inf
    with pytest.raises(E):
        engine.set_source(bad)
    for w_bad in (-1.0, np.nan, np.inf):                     # weights: finite and >= 0
        w = np.ones(len(comp), np.float32)
        w[7] = w_bad
        with pytest.raises(E) as e:
            engine.set_source(comp, w)
        assert e.value.status == -1
    with pytest.raises(ValueError):
        engine.set_source(np.zeros((5, 2), np.float32))
    engine.set_source(comp)
    for kw in ({"max_iter": 0}, {"tol": -1.0}, {"max_corr_dist": 0.0}, {"max_corr_dist": np.nan}):
        with pytest.raises(E) as e:
            engine.icp(**kw)
        assert e.value.status == -1
    engine.set_source(comp + 100.0)                          # no correspondence inside the gate
    with pytest.raises(E) as e:
        engine.icp(5, 0.0, 0.10)
    assert e.value.status == -6
    engine.set_source(comp[:2])                              # fewer than 3 points
    with pytest.raises(E) as e:
        engine.icp(5, 0.0, np.inf)
    assert e.value.status == -6
    engine.set_gold(gold[:1])                                # rank-deficient: everything maps to one point
    engine.set_source(comp)
    with pytest.raises(E) as e:
        engine.icp(5, 0.0, np.inf)
    assert e.value.status == -6
    with pytest.raises(E):
        mapalign.MapAligner(99)                              # no such device


def test_context_reuse_across_clouds(engine):
    for seed, n in ((9, 3000), (10, 12_000), (11, 500)):
        gold, comp, R0, t0 = syn.make_pair(seed, n, resample=False)
        engine.set_gold(gold)
        engine.set_source(comp)
        res = engine.icp(50, 0.0, 0.25 if n < 1000 else 0.10)
        eR, et = orc.rt_error(res.R, res.t, R0, t0)
        assert eR <= R_TOL_FRO and et <= T_TOL_M, (seed, eR, et)


# ----------------------------------------------------------------------------- row a8
def test_golden_depth_unprojection(engine, golden):
    g = golden("depth_a.npz")
    gold, _, _, _ = syn.make_pair(13, 4000)
    engine.set_gold(gold)
    n = engine.set_source_from_depth(g["depth"], g["K"], g["T"], float(g["scale"]))
    assert n == int(g["mask"].sum())
    pts = engine.get_source()
    # world coordinates up to 20 m stored as centred fp32: 1 ulp(10 m) ~ 1e-6 m
    assert np.abs(pts.astype(np.float64) - g["points"]).max() <= 4e-6


@pytest.mark.parametrize("dtype", [np.float32, np.uint16])
def test_depth_unprojection_vs_oracle_and_pipeline(engine, dtype):
    # BASELINE configs[2] at reduced size: depth frame -> cloud -> align against the gold map
    depth, K, T = syn.make_depth_frame(0, width=320, height=240, fx=250.0, fy=250.0, cx=159.5, cy=119.5)
    if dtype == np.uint16:
        depth = np.clip(np.round(depth), 0, 65535).astype(np.uint16)
    gold, _, _, _ = syn.make_pair(0, 200_000)
    engine.set_gold(gold)
    n = engine.set_source_from_depth(depth, K, T, 1000.0)
    opts, mask = orc.unproject_oracle(depth, K, T, 1000.0)
    assert n == int(mask.sum()) and n > 0.9 * depth.size
    pts = engine.get_source()
    assert np.abs(pts.astype(np.float64) - opts).max() <= 4e-6
    # perturb the camera pose by a small rigid motion and recover it through the fused path
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.2, 1.0, 0.1], np.deg2rad(0.5))
    dT[:3, 3] = [0.03, -0.02, 0.02]
    n2 = engine.set_source_from_depth(depth, K, dT @ T, 1000.0)
    assert n2 == n
    res = engine.icp(40, 0.0, 0.10)
    ref = orc.icp_oracle(gold, engine.get_source(), max_iter=40, max_corr_dist=0.10, n_jobs=8)
    eR, et = orc.rt_error(res.R, res.t, ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    idx, d2 = engine.nn(None, 0.10)                          # unsorted (pixel-order) source path
    oidx, od2 = orc.GoldIndex(gold, n_jobs=8).query(engine.get_source())
    assert_nn_matches(idx, d2, oidx, od2, gate=0.10)


def test_depth_error_paths(engine):
    gold, _, _, _ = syn.make_pair(14, 1000)
    engine.set_gold(gold)
    K, T = (50.0, 50.0, 16.0, 12.0), np.eye(4)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.set_source_from_depth(np.zeros((24, 32), np.float32), K, T)
    assert e.value.status == -4                              # no valid pixel
    with pytest.raises(mapalign.MapAlignError):
        engine.set_source_from_depth(np.ones((24, 32), np.float32), (0.0, 50.0, 1.0, 1.0), T)
    with pytest.raises(ValueError):
        engine.set_source_from_depth(np.ones((24,), np.float32), K, T)


# ----------------------------------------------------------------------------- sharded stepping
def test_split_stepping_two_shards_equals_one_context(lib_built):
    """Sharded-source mode on ONE GPU: two contexts each hold half the source cloud; their 17
    fixed-point sums are added (what RCCL's all-reduce does across GPUs) and both solve.  With
    the whole cloud's sum bounds on both contexts the result is the single-context loop's, bit for bit."""
    import torch
    from mapalign import dist as mdist

    gold, comp, _, _ = syn.make_pair(15, 40_000, 30_001)
    with mapalign.MapAligner(0) as full:
        full.set_gold(gold)
        full.set_source(comp)
        ref = full.icp(12, 0.0, 0.10)
    engs, accs = [], []
    for r in range(2):
        lo, hi = mdist.shard_bounds(len(comp), 2, r)
        a = mapalign.MapAligner(0)
        a.set_gold(gold)
        a.set_source(comp[lo:hi])
        he = mdist.HipShardEngine(a, 0.10, 0.0)
        he.bind_current_stream()
        engs.append(he)
        accs.append(he.new_acc())
    whole = mapalign.MapAligner.combine_sum_bounds([he.aligner.sum_bounds() for he in engs])
    for he in engs:
        he.aligner.set_sum_bounds(whole)
        he.state_set(None)
    for _ in range(12):
        for he, acc in zip(engs, accs):
            he.accumulate_into(acc)
        total = accs[0] + accs[1]
        for he, acc in zip(engs, accs):
            acc.copy_(total)
            he.solve_from(acc)
    T0, i0 = engs[0].state_get()
    T1, i1 = engs[1].state_get()
    for he in engs:
        he.aligner.use_own_stream()
        he.aligner.close()
    assert np.array_equal(T0, T1) and i0["iterations"] == 12
    assert np.array_equal(T0, ref.T)                 # exact (integer) sums: the partition leaves no trace
    assert i0["inliers"] == ref.inliers and i0["rmse"] == ref.rmse


def test_sharded_icp_driver_single_rank(lib_built):
    from mapalign import dist as mdist
    gold, comp, R0, t0 = syn.make_pair(16, 8000, resample=False)
    R, t, info = mdist.sharded_align(gold, comp, max_iter=50, device=0, return_info=True)
    eR, et = orc.rt_error(R, t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M and info["iterations"] == 50


# ----------------------------------------------------------------------------- full size (1M)
def test_full_size_1m_known_transform_and_invariances(engine):
    """BASELINE configs[1] size.  The oracle needs minutes here, so parity is shown through
    properties: align(A, T.A) recovers T; the result does not depend on the grid cell;
    two runs are bitwise equal."""
    gold, comp, R0, t0 = syn.make_pair(0, 1_000_000, resample=False)
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(50, 0.0, 0.10, with_trace=True)
    eR, et = orc.rt_error(res.R, res.t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert res.inliers == 1_000_000 and res.rmse < 1e-5
    again = engine.icp(50, 0.0, 0.10)
    assert np.array_equal(res.T, again.T)
    engine.set_gold(gold, 0.10)
    engine.set_source(comp)
    coarse = engine.icp(50, 0.0, 0.10)
    assert np.allclose(coarse.T, res.T, atol=1e-9)
    # linearity of the accumulators: W, s_p, s_q of iteration 0 equal direct sums over matched points
    idx, d2 = engine.nn(None, 0.10)
    m = idx != NO_MATCH
    assert res.trace[0, 0] == m.sum()
    assert np.allclose(res.trace[0, 1:4], comp[m].astype(np.float64).sum(0), rtol=1e-9)
    assert np.allclose(res.trace[0, 4:7], gold[idx[m]].astype(np.float64).sum(0), rtol=1e-9)


def test_2p5m_cloud_grouped_schedule_known_transform(engine):
    """BASELINE configs[4] size class (2M+ points per pair): beyond 2M points the longest-first
    schedule ranks GROUPS of segments (more than 512 segments), and the grid is padded to whole
    groups.  align(A, T.A) must still recover T, bitwise the same on every run (the schedule
    follows measured durations, the sums must not), and the accumulators of iteration 0 must
    equal direct sums over the matched points."""
    n = 2_500_000
    gold, comp, R0, t0 = syn.make_pair(3, n, resample=False)
    engine.set_gold(gold)
    engine.set_source(comp)
    res = engine.icp(30, 0.0, 0.10, with_trace=True)
    eR, et = orc.rt_error(res.R, res.t, R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert res.inliers == n and res.rmse < 1e-5
    again = engine.icp(30, 0.0, 0.10, with_trace=True)
    assert np.array_equal(res.T, again.T) and np.array_equal(res.trace, again.trace)
    idx, d2 = engine.nn(None, 0.10)
    m = idx != NO_MATCH
    assert res.trace[0, 0] == m.sum()
    assert np.allclose(res.trace[0, 1:4], comp[m].astype(np.float64).sum(0), rtol=1e-9)
    assert np.allclose(res.trace[0, 4:7], gold[idx[m]].astype(np.float64).sum(0), rtol=1e-9)


def test_full_size_1m_resampled_pair_one_step_vs_oracle(engine):
    """One full-size iteration against the KD-tree oracle (a single 1M query is affordable)."""
    gold, comp, R0, t0 = syn.make_pair(0, 1_000_000)
    engine.set_gold(gold)
    engine.set_source(comp)
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = R0, t0                        # near the optimum: everything inside the gate
    index = orc.GoldIndex(gold, n_jobs=-1)
    oacc, oidx, od2, Rn, tn, ok = orc.icp_step(index, comp, R0, t0, 0.10)
    acc, Tn = engine.icp_step(T0, 0.10)
    assert acc[0] == oacc[0] and acc_close(acc, oacc)
    eR, et = orc.rt_error(Tn[:3, :3], Tn[:3, 3], Rn, tn)
    assert eR <= 1e-7 and et <= 1e-6
    idx, d2 = engine.nn(T0, 0.10)
    assert_nn_matches(idx, d2, oidx, od2, gate=0.10)
    # ... and EXACTLY, index by index, on the coordinates as stored (1M queries against the KD-tree)
    ties = assert_nn_exact_on_stored(engine, T0, 0.10, idx, d2, use_tree=True)
    print(f"1M one-step: {ties} of {len(idx)} indices differ from the oracle on the stored coordinates, all fp32 ties")


# ----------------------------------------------------------------------------- adaptor (a1-a3, N1-N3)
def test_alignment_adaptor_far_start_with_ply_and_pickle(tmp_path):
    """Reference call shape (task_def.py:175-187) end to end: .ply maps on disk, a maintenance
    scan that starts 150 degrees / several metres away (the README GIF's situation), boxes
    moved onto the gold map, the visualiser's three pickle keys written."""
    import pickle

    from mapalign import Alignment, boxes, ply, transforms

    gold, comp, R0, t0 = syn.make_pair(17, 60_000, 50_000, angle_deg=150.0, t0=np.array([7.0, -4.0, 1.5]))
    gold_path, comp_path = tmp_path / "gold.ply", tmp_path / "scan.ply"
    ply.write_ply(gold_path, gold)
    ply.write_ply(comp_path, comp)
    T_true = transforms.rt_to_matrix(R0, t0)
    sign = np.array([3.0, 0.0, 1.2])                         # a sign on the gold map ...
    d = 0.15
    gold_boxes = {0: [[sign + [-d, 0, -d], sign + [d, 0, -d], sign + [d, 0, d], sign + [-d, 0, d], 0, 0.9, "exit"]]}
    inv = np.linalg.inv(T_true)                              # ... as the maintenance scan saw it
    comp_boxes = {4: [[transforms.apply(inv, c[None])[0] for c in gold_boxes[0][0][:4]] + [1, 0.8, "exit"]]}
    pkl = tmp_path / "scan.pkl"
    with open(pkl, "wb") as f:
        pickle.dump({"pose_df": None}, f)
    al = Alignment(base_pose_df=None, comparison_pose_df=None, base_bboxes=gold_boxes, comparison_bboxes=comp_boxes,
                   visualise=False, base_map_filepath=str(gold_path), comparison_map_filepath=str(comp_path),
                   pickle_path=str(pkl), gate_schedule=None)   # one gate: the same loop the oracle runs below
    aligned, T, gold_pts, info = al.compare("scan")
    eR, et = orc.rt_error(T[:3, :3], T[:3, 3], R0, t0)
    assert eR <= 2e-3 and et <= 2e-2                         # independent resample: sampling noise, not parity
    ref = orc.icp_oracle(gold, comp, max_iter=50, max_corr_dist=0.10, init=transforms.compose(al.transformations[:2]),
                         n_jobs=8)
    eR, et = orc.rt_error(T[:3, :3], T[:3, 3], ref["R"], ref["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M                 # parity with the oracle from the same coarse start
    # staged list: (R, centre) tuple, 4x4 translation, 4x4 refinement — composes to T
    tr = al.transformations
    assert isinstance(tr[0], tuple) and tr[1].shape == (4, 4) and tr[2].shape == (4, 4)
    assert np.allclose(transforms.compose(tr), T, atol=1e-9) and transforms.is_rigid(T, 1e-8)
    # ... and through the only reference code that consumes the list (visualise.py:82-130, restated in
    # tests/visualiser_replay.py): 20 log-map steps per stage end on the composed transform, all increments real
    from visualiser_replay import assert_replayable
    assert_replayable(tr, comp[::500].astype(np.float64), T, atol=1e-6)
    with open(pkl, "rb") as f:
        state = pickle.load(f)
    assert set(state) >= {"transformations", "base_map_filepath", "comparison_map_filepath", "pose_df"}
    res = boxes.match_bboxes(gold_boxes, aligned, max_dist=0.1)
    assert res["matches"] and res["matches"][0][2] < 0.03 and res["damaged"] == [0] and not res["missing"]
    assert len(gold_pts) == len(gold)


# ----------------------------------------------------------------------------- row a10
def test_lift_boxes_vs_oracle_and_frame_helper(engine):
    from mapalign import lifting
    depth, K, T = syn.make_depth_frame(3, width=192, height=256, fx=150.0, fy=150.0, cx=95.5, cy=127.5)
    depth = depth.copy()
    depth[10, 10] = 0.0
    depth[200, 150] = np.nan
    rng = np.random.default_rng(0)
    x1 = rng.uniform(-5, 150, 40); y1 = rng.uniform(-5, 200, 40)
    boxes = np.stack([x1, y1, x1 + rng.uniform(1, 60, 40), y1 + rng.uniform(1, 70, 40)], 1).astype(np.float32)
    boxes[0] = [10.2, 10.9, 150.7, 200.1]                     # two corners on invalid pixels
    boxes[1] = [-20.0, -3.0, 500.0, 900.0]                    # clamped into the image
    oc, ov = orc.lift_boxes_oracle(boxes, depth, K, T, 1000.0)
    corners, valid = engine.lift_boxes(boxes, K, T, depth, 1000.0)
    assert np.array_equal(valid, ov) and not valid[0, 0] and not valid[0, 2]
    assert np.allclose(corners[valid], oc[ov], atol=1e-9) and np.isnan(corners[~valid]).all()
    c2, v2 = engine.lift_boxes(boxes, K, T, None, 1000.0)     # reuse the uploaded frame
    assert np.array_equal(np.nan_to_num(c2), np.nan_to_num(corners)) and np.array_equal(v2, valid)
    # uint16 depth and the per-frame helper mirroring pose_processor._3d_processing
    d16 = np.clip(np.round(np.nan_to_num(depth)), 0, 65535).astype(np.uint16)
    intr = {"image_width": 1920, "image_height": 2560, "fx": 1500.0, "fy": 1500.0, "cx": 955.0, "cy": 1275.0}
    pose = [1.0, 0.5, 1.2, 0.0, 0.0, 0.38268343, 0.92387953]
    dets = [[400.0, 500.0, 900.0, 1500.0, 2, 0.9, "exit"], [1000.0, 100.0, 1500.0, 700.0, 0, 0.5, "fire"]]
    rows = lifting.lift_frame_bboxes(engine, dets, d16, intr, pose, 192, 256)
    assert len(rows) == 2 and rows[0][4:] == [2, 0.9, "exit"] and len(rows[0]) == 7   # pose_processor.py:208
    Kd = lifting.rescaled_intrinsics(intr, 192)
    sb = np.array([lifting.scale_bounding_box(d[:4], (1920, 2560), (192, 256)) for d in dets])
    oc, _ = orc.lift_boxes_oracle(sb, d16, Kd, syn.pose_to_matrix(pose), 1000.0)
    assert np.allclose(np.array(rows[0][:4]), oc[0], atol=1e-9) and np.allclose(np.array(rows[1][:4]), oc[1], atol=1e-9)
    with pytest.raises(mapalign.MapAlignError):
        mapalign.MapAligner(0).lift_boxes(boxes, K, T, None)   # no frame uploaded yet


# ----------------------------------------------------------------------------- config C3 at full size
def test_config3_full_size_depth_frame_against_1m_gold(engine):
    """BASELINE configs[2]: 1280x960 depth image -> fused unprojection -> alignment against a
    1M-point gold map.  Unprojection: exact pixel count + coordinates vs the oracle; alignment:
    a perturbed camera pose is recovered, and 5 iterations match the oracle's 5."""
    depth, K, T = syn.make_depth_frame(0)                     # 1280 x 960, fx = fy = 1000
    gold, _, _, _ = syn.make_pair(0, 1_000_000)
    engine.set_gold(gold)
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.1, 0.9, 0.2], np.deg2rad(0.4))
    dT[:3, 3] = [0.025, -0.015, 0.02]
    n = engine.set_source_from_depth(depth, K, dT @ T, 1000.0)
    opts, mask = orc.unproject_oracle(depth, K, dT @ T, 1000.0)
    assert n == int(mask.sum()) and n > 1_100_000
    pts = engine.get_source()
    assert np.abs(pts.astype(np.float64) - opts).max() <= 8e-6      # camera sees up to ~18 m: 2 ulp fp32
    res5 = engine.icp(5, 0.0, 0.10)
    ref5 = orc.icp_oracle(gold, pts, max_iter=5, max_corr_dist=0.10, n_jobs=-1)
    eR, et = orc.rt_error(res5.R, res5.t, ref5["R"], ref5["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert abs(res5.inliers - ref5["inliers"]) <= 4
    res = engine.icp(50, 0.0, 0.10)
    # the frame samples the gold surfaces (analytic render, no jitter): the perturbation is undone
    err = np.linalg.inv(dT) @ np.linalg.inv(res.T)            # should be ~identity
    assert np.linalg.norm(err[:3, :3] - np.eye(3)) <= 2e-3 and np.linalg.norm(err[:3, 3]) <= 1e-2


def test_georeferenced_coordinates_fp64_inputs(engine):
    """UTM-like coordinates (1e5..1e7 m) as float64: the host shift keeps millimetre parity
    (plain fp32 storage would quantise them to 0.03..0.5 m)."""
    gold, comp, R0, t0 = syn.make_pair(18, 30_000, 25_000)
    off = np.array([412_345.0, 5_234_567.0, 120.0])
    gold64, comp64 = gold.astype(np.float64) + off, comp.astype(np.float64) + off
    gi = engine.set_gold(gold64)
    engine.set_source(comp64)
    assert np.abs(gi["origin"] - (off + [10, 1.5, 1.25])).max() < 1.0
    res = engine.icp(30, 0.0, 0.10, with_trace=True)
    # oracle on the un-offset clouds, conjugated by the offset (exact algebra).  The fp64 oracle
    # run directly at 5e6 m is itself only good to ~1e-2: its world-frame cross-covariance
    # S - W mu mu^T cancels 12 of its 16 digits — which is why the device works in a centred frame.
    ref = orc.icp_oracle(gold, comp, max_iter=30, max_corr_dist=0.10, n_jobs=8)
    # t at the world origin has a 5000 km lever arm (1e-10 in R moves it by 0.5 mm), so the
    # translation is compared where it acts: on the cloud
    t_ref = ref["t"] + off - ref["R"] @ off
    assert np.linalg.norm(res.R - ref["R"]) <= R_TOL_FRO
    moved = comp64[::97] @ res.R.T + res.t
    moved_ref = comp64[::97] @ ref["R"].T + t_ref
    assert np.abs(moved - moved_ref).max() <= T_TOL_M * 1e-2
    assert res.trace[0, 0] == ref["trace"][0]["acc"][0] and abs(res.trace[-1, 17] - ref["rmse"]) < 1e-6
    # same answer as the un-offset problem, conjugated by the offset
    engine.set_gold(gold)
    engine.set_source(comp)
    loc = engine.icp(30, 0.0, 0.10)
    assert np.allclose(res.R, loc.R, atol=1e-9)
    assert np.abs((comp64[::97] @ res.R.T + res.t) - (comp64[::97] @ loc.R.T + loc.t + off - loc.R @ off)).max() <= 1e-5
    idx64, d264 = (lambda e: (e.set_gold(gold64), e.set_source(comp64), e.nn(res.T, 0.1))[2])(engine)
    oi, od2 = orc.GoldIndex(gold64, n_jobs=8).query(comp64 @ res.R.T + res.t)
    assert_nn_matches(idx64, d264, oi, od2, gate=0.1)


def test_maximum_sizes_are_refused_not_truncated(engine):
    """Clouds beyond the 32-bit offset range fail loudly before any memory is touched."""
    import ctypes as C
    from mapalign import _lib
    lib, ctx = engine._lib, engine._ctx
    small = np.zeros((4, 3), np.float32)
    p = small.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.ma_set_gold(ctx, p, 1 << 28, 0.0) == _lib.MA_ERR_TOO_LARGE
    gold, comp, _, _ = syn.make_pair(19, 500)
    engine.set_gold(gold)
    assert lib.ma_set_source(ctx, p, None, 0xFFFFFFFE) == _lib.MA_ERR_TOO_LARGE
    K = (C.c_float * 4)(50, 50, 8, 8)
    T = (C.c_double * 16)(*np.eye(4).ravel())
    nv = C.c_uint64(0)
    assert lib.ma_set_source_from_depth(ctx, p, 0, 70000, 70000, K, T, 1000.0, C.byref(nv)) == _lib.MA_ERR_INVALID
    assert b"2^28" in lib.ma_last_error(ctx) or b"image" in lib.ma_last_error(ctx)


# ----------------------------------------------------------------------------- row N4: map builder
def test_map_builder_voxel_centroids_and_as_gold(engine):
    """Depth frames + poses -> one voxel-deduplicated cloud on the device, checked against the
    NumPy restatement; then used as the gold map for a fourth frame without a host round trip."""
    K = (250.0, 250.0, 159.5, 119.5)
    rects = syn.carriage_rectangles(np.random.default_rng(5))
    poses = []
    for k, x in enumerate((2.0, 4.0, 6.0)):
        T = syn.default_camera_pose()
        T[:3, 3] = (x, 1.4 + 0.1 * k, 1.3)
        T[:3, :3] = T[:3, :3] @ syn.rotation_about_axis([0, 1, 0], 0.1 * k)
        poses.append(T)
    frames = [syn.render_depth(rects, T, 320, 240, *K) for T in poses]
    voxel = 0.05
    engine.map_begin(voxel, 4 * 320 * 240)
    added = [engine.map_add_depth(d, K, T) for d, T in zip(frames, poses)]
    pts = np.concatenate([orc.unproject_oracle(d, K, T, 1000.0)[0] for d, T in zip(frames, poses)])
    assert sum(added) == len(pts)
    cloud = engine.map_finish()
    origin = np.round(poses[0][:3, 3])
    ref = orc.voxel_centroids_oracle(pts, voxel, origin)
    assert cloud.shape == ref.shape
    # same voxels in the same (z,y,x) order; centroids agree to fp32 storage precision
    assert np.abs(cloud.astype(np.float64) - ref).max() <= 2e-5
    assert len(cloud) < 0.5 * len(pts)                       # overlapping frames really were merged
    # a fourth frame aligned against the map built on the device
    n_vox = engine.map_finish(as_gold=True, fetch=False)
    assert n_vox == len(ref) and engine.grid_info()["n_occupied"] > 0
    T4 = syn.default_camera_pose()
    T4[:3, 3] = (3.0, 1.5, 1.3)
    d4 = syn.render_depth(rects, T4, 320, 240, *K)
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.3, 1.0, 0.2], np.deg2rad(0.5))
    dT[:3, 3] = [0.02, -0.015, 0.01]
    engine.set_source_from_depth(d4, K, dT @ T4)
    res = engine.icp(30, 0.0, 0.10)
    refi = orc.icp_oracle(cloud, engine.get_source(), max_iter=30, max_corr_dist=0.10, n_jobs=8)
    eR, et = orc.rt_error(res.R, res.t, refi["R"], refi["t"])
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    with pytest.raises(mapalign.MapAlignError):
        mapalign.MapAligner(0).map_add_depth(d4, K, T4)       # map_begin missing
    engine.map_begin(0.05, 1000)
    with pytest.raises(mapalign.MapAlignError) as e:
        engine.map_add_depth(d4, K, T4)                        # capacity exceeded
    assert e.value.status == -7
