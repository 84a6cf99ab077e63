"""CPU tests of the oracle: closed-form known answers + the committed golden vectors.

The reference has no tests or goldens for this path, so the oracle is
pinned by analytic answers: align(A, T.A) = T, align(A, A) = I, permutation invariance,
Kabsch on exact correspondences.
"""
import numpy as np
import pytest

from mapalign import synthetic as syn
from oracle import icp_oracle as orc

from conftest import R_TOL_FRO, T_TOL_M


def test_known_transform_recovered_config1():
    # BASELINE.json configs[0]: two synthetic 10k-point clouds with a known rigid transform
    gold, comp, R0, t0 = syn.make_pair(0, 10_000, resample=False)
    res = orc.icp_oracle(gold, comp, max_iter=50, max_corr_dist=0.10, n_jobs=1)
    eR, et = orc.rt_error(res["R"], res["t"], R0, t0)
    assert eR <= R_TOL_FRO and et <= T_TOL_M
    assert res["inliers"] == 10_000 and res["rmse"] < 1e-5


def test_identity_on_identical_clouds():
    gold, _, _, _ = syn.make_pair(1, 3000, resample=False)
    res = orc.icp_oracle(gold, gold, max_iter=5, n_jobs=1)
    assert np.allclose(res["R"], np.eye(3), atol=1e-12) and np.allclose(res["t"], 0, atol=1e-12)
    assert res["rmse"] < 1e-9


def test_permutation_invariance():
    gold, comp, _, _ = syn.make_pair(2, 4000, resample=False)
    a = orc.icp_oracle(gold, comp, max_iter=10, max_corr_dist=0.3, n_jobs=1)
    perm = np.random.default_rng(0).permutation(len(comp))
    b = orc.icp_oracle(gold, comp[perm], max_iter=10, max_corr_dist=0.3, n_jobs=1)
    assert np.allclose(a["R"], b["R"], atol=1e-10) and np.allclose(a["t"], b["t"], atol=1e-10)


def test_kabsch_exact_correspondences_and_reflection_guard():
    rng = np.random.default_rng(3)
    p = rng.normal(size=(500, 3))
    R0 = syn.rotation_about_axis([1, 2, 3], 0.7)
    t0 = np.array([0.3, -2.0, 1.1])
    q = p @ R0.T + t0
    acc = orc.accumulate(p, q, np.ones(len(p)), np.zeros(len(p)))
    dR, dt, ok = orc.kabsch_from_acc(acc)
    assert ok and np.allclose(dR, R0, atol=1e-12) and np.allclose(dt, t0, atol=1e-12)
    # planar correspondences: the naive SVD solution may be a reflection; the guard must give det +1
    p[:, 2] = 0.0
    q = p @ R0.T + t0
    dR, dt, ok = orc.kabsch_from_acc(orc.accumulate(p, q, np.ones(len(p)), np.zeros(len(p))))
    assert ok and np.isclose(np.linalg.det(dR), 1.0) and np.allclose(dR, R0, atol=1e-10)


def test_kabsch_degenerate_inputs():
    acc = np.zeros(17)
    assert orc.kabsch_from_acc(acc)[2] is False          # W = 0
    p = np.array([[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [3.0, 0, 0]])   # collinear -> rank 1
    assert orc.kabsch_from_acc(orc.accumulate(p, p, np.ones(4), np.zeros(4)))[2] is False


def test_kdtree_matches_bruteforce():
    gold, comp, _, _ = syn.make_pair(4, 2000, 1500)
    idx, d2 = orc.GoldIndex(gold, n_jobs=1).query(comp)
    bi, bd2 = orc.nn_bruteforce(gold, comp)
    assert np.array_equal(idx, bi) and np.allclose(d2, bd2, rtol=1e-12, atol=1e-15)


def test_gate_excludes_far_points_and_weights_scale():
    gold, comp, _, _ = syn.make_pair(5, 2000, resample=False)
    index = orc.GoldIndex(gold, n_jobs=1)
    acc, idx, d2, *_ = orc.icp_step(index, comp, np.eye(3), np.zeros(3), max_corr_dist=0.05)
    assert acc[0] == float((d2 <= 0.05 ** 2).sum())
    acc2, *_ = orc.icp_step(index, comp, np.eye(3), np.zeros(3), max_corr_dist=0.05, weights=np.full(len(comp), 2.0))
    assert np.allclose(acc2, 2.0 * acc)


def test_tol_stops_early_and_empty_inputs_raise():
    gold, comp, _, _ = syn.make_pair(6, 3000, resample=False)
    res = orc.icp_oracle(gold, comp, max_iter=50, tol=1e-9, max_corr_dist=0.3, n_jobs=1)
    assert res["converged"] and res["iterations"] < 50
    with pytest.raises(ValueError):
        orc.icp_oracle(gold, np.zeros((0, 3)))
    with pytest.raises(ValueError):
        orc.icp_oracle(np.zeros((0, 3)), comp)


def test_far_apart_clouds_are_degenerate_not_wrong():
    gold, comp, _, _ = syn.make_pair(7, 1000, resample=False)
    res = orc.icp_oracle(gold, comp + 100.0, max_iter=5, max_corr_dist=0.1, n_jobs=1)
    assert not res["ok"] and np.allclose(res["R"], np.eye(3))


def test_unproject_round_trip():
    # project known camera-frame points to pixels, unproject with the pinhole model of
    # pose_processor.py:184-196 / SURVEY §8a-8 and compare
    K = (80.0, 82.0, 31.5, 23.5)
    T = syn.pose_to_matrix([1.0, -2.0, 0.5, 0.1, -0.2, 0.3, 0.9])
    h, w = 48, 64
    rng = np.random.default_rng(8)
    depth = rng.uniform(500.0, 8000.0, size=(h, w)).astype(np.float32)
    depth[0, 0] = 0.0
    depth[1, 1] = np.nan
    pts, mask = orc.unproject_oracle(depth, K, T, 1000.0)
    assert mask.sum() == h * w - 2 and len(pts) == mask.sum()
    cam = (pts - T[:3, 3]) @ T[:3, :3]        # world -> camera
    u = cam[:, 0] / cam[:, 2] * K[0] + K[2]
    v = cam[:, 1] / cam[:, 2] * K[1] + K[3]
    ys, xs = np.nonzero(mask)
    assert np.allclose(u, xs, atol=1e-8) and np.allclose(v, ys, atol=1e-8)
    assert np.allclose(cam[:, 2] * 1000.0, depth[mask], rtol=1e-12)


def test_quaternion_pose_convention():
    # Hamilton (qx,qy,qz,qw), pose row tx ty tz qx qy qz qw (mapper/database_query.py:22)
    s = np.sin(np.pi / 4)
    R = syn.quat_to_rot([0, 0, s, s])            # +90 deg about z
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0], atol=1e-12)


@pytest.mark.parametrize("name", ["pair_a.npz", "pair_weighted.npz"])
def test_oracle_reproduces_golden_pairs(golden, name):
    g = golden(name)
    w = g.get("weights")
    res = orc.icp_oracle(g["gold"], g["comp"], max_iter=int(g["iters"]), tol=0.0, max_corr_dist=float(g["gate"]),
                         weights=w, keep_nn=True, n_jobs=1)
    assert np.allclose(res["R"], g["R"], atol=1e-12) and np.allclose(res["t"], g["t"], atol=1e-12)
    acc = np.stack([r["acc"] for r in res["trace"]])
    assert np.allclose(acc, g["acc"], rtol=1e-12, atol=1e-9)
    assert np.array_equal(res["trace"][0]["idx"], g["nn0"])


def test_oracle_reproduces_golden_depth(golden):
    g = golden("depth_a.npz")
    pts, mask = orc.unproject_oracle(g["depth"], g["K"], g["T"], float(g["scale"]))
    assert np.array_equal(mask, g["mask"]) and np.allclose(pts, g["points"], atol=1e-12)


# ---------------------------------------------------------------------------------------
# Independent second restatement: scipy cKDTree for the correspondences and Horn's
# closed-form QUATERNION solve for the rigid step (a different search structure and a
# different solver from the oracle's sklearn KD-tree + SVD Kabsch).  The oracle is unpinned by
# the reference (its align.py is absent), so agreement of two independently written
# restatements is the strongest cross-check available here.
# ---------------------------------------------------------------------------------------
def horn_quaternion_step(p, q, w):
    """Horn 1987: the rotation is the eigenvector of the largest eigenvalue of the 4x4
    matrix N built from the weighted cross-covariance; then t = mu_q - R mu_p."""
    W = w.sum()
    mu_p, mu_q = (w[:, None] * p).sum(0) / W, (w[:, None] * q).sum(0) / W
    M = ((p - mu_p) * w[:, None]).T @ (q - mu_q)          # M_ab = sum w (p-mu_p)_a (q-mu_q)_b
    Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz = M.ravel()
    N = np.array([[Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx],
                  [Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz],
                  [Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy],
                  [Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz]])
    vals, vecs = np.linalg.eigh(N)
    qw, qx, qy, qz = vecs[:, np.argmax(vals)]
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    return R, mu_q - R @ mu_p


def second_restatement_icp(gold, comp, max_iter, gate, weights=None):
    from scipy.spatial import cKDTree
    gold = np.asarray(gold, dtype=np.float64)
    comp = np.asarray(comp, dtype=np.float64)
    tree = cKDTree(gold)
    R, t = np.eye(3), np.zeros(3)
    inliers = []
    for _ in range(max_iter):
        moved = comp @ R.T + t
        d, j = tree.query(moved, k=1)
        w = (d <= gate).astype(np.float64) if np.isfinite(gate) else np.ones(len(d))
        if weights is not None:
            w = w * weights
        inliers.append(float(w.sum()))
        keep = w > 0
        dR, dt = horn_quaternion_step(moved[keep], gold[j[keep]], w[keep])
        R, t = dR @ R, dR @ t + dt
    return R, t, inliers


@pytest.mark.parametrize("seed,n,gate,resample", [(0, 10_000, 0.10, True), (3, 6000, 0.35, True), (5, 4000, np.inf, False),
                                                   (7, 8000, 0.10, True)])
def test_oracle_agrees_with_independent_ckdtree_horn_restatement(seed, n, gate, resample):
    gold, comp, R0, t0 = syn.make_pair(seed, n, resample=resample)
    iters = 25
    ref = orc.icp_oracle(gold, comp, max_iter=iters, max_corr_dist=gate, n_jobs=1)
    R2, t2, inl2 = second_restatement_icp(gold, comp, iters, gate)
    # same correspondences every iteration => same inlier counts, and transforms equal to rounding
    assert [row["inliers"] for row in ref["trace"]] == inl2
    eR, et = orc.rt_error(ref["R"], ref["t"], R2, t2)
    assert eR < 1e-9 and et < 1e-9, (eR, et)


def test_oracle_agrees_with_second_restatement_weighted():
    gold, comp, _, _ = syn.make_pair(11, 5000)
    w = np.random.default_rng(1).uniform(0.0, 2.0, len(comp))
    ref = orc.icp_oracle(gold, comp, max_iter=12, max_corr_dist=0.2, weights=w, n_jobs=1)
    R2, t2, _ = second_restatement_icp(gold, comp, 12, 0.2, weights=w)
    eR, et = orc.rt_error(ref["R"], ref["t"], R2, t2)
    assert eR < 1e-9 and et < 1e-9, (eR, et)


def test_kabsch_weights_are_not_counts():
    """Normalised weights (sum 1) over many exact correspondences must solve, not report
    'fewer than 3 correspondences' (the W >= 3 test confused weight with count)."""
    rng = np.random.default_rng(4)
    p = rng.normal(size=(500, 3))
    R0 = syn.rotation_about_axis([0.2, 0.9, -0.4], 0.3)
    q = p @ R0.T + np.array([0.3, -0.1, 0.2])
    w = np.full(500, 1.0 / 500)
    dR, dt, ok = orc.kabsch_from_acc(orc.accumulate(p, q, w, np.zeros(500)))
    assert ok and np.allclose(dR, R0, atol=1e-12) and np.allclose(dt, [0.3, -0.1, 0.2], atol=1e-12)
    # one or two pairs, or no weight at all, are degenerate whatever their weight
    for k in (0, 1, 2):
        wk = np.zeros(500)
        wk[:k] = 5.0
        assert not orc.kabsch_from_acc(orc.accumulate(p, q, wk, np.zeros(500)))[2]


def test_lift_boxes_oracle_median_rule_known_answers():
    """z_rule = 1 (the comment at pose_processor.py:183): median of the valid depths over the box's pixel
    rectangle, ends included; even counts average the middle pair; no valid pixel -> invalid corners."""
    depth = np.zeros((6, 8), dtype=np.float32)
    depth[1:4, 2:4] = [[1000.0, 2000.0], [3000.0, 0.0], [4000.0, 5000.0]]
    K, T = (10.0, 10.0, 4.0, 3.0), np.eye(4)
    c, v = orc.lift_boxes_oracle([[2.2, 1.9, 3.9, 3.1]], depth, K, T, 1000.0, z_rule=1)
    assert v.all() and np.allclose(c[0, :, 2], 3.0)                       # {1,2,3,4,5} m
    assert np.allclose(c[0, 0], [(2 - 4.0) * 3.0 / 10.0, (1 - 3.0) * 3.0 / 10.0, 3.0])   # each corner through its own pixel
    c, v = orc.lift_boxes_oracle([[2.0, 1.0, 3.5, 1.5]], depth, K, T, 1000.0, z_rule=1)
    assert np.allclose(c[0, :, 2], 1.5)                                   # {1,2} m -> 1.5
    c, v = orc.lift_boxes_oracle([[5.0, 4.0, 7.0, 5.0]], depth, K, T, 1000.0, z_rule=1)
    assert not v.any() and np.isnan(c).all()
    c0, v0 = orc.lift_boxes_oracle([[2.2, 1.9, 3.9, 3.1]], depth, K, T, 1000.0)      # rule 0 beside it: corner pixels
    assert np.allclose(c0[0, :, 2][v0[0]], [1.0, 2.0, 5.0, 4.0][:int(v0[0].sum())]) or v0[0].sum() == 4


@pytest.mark.parametrize("seed,n,gate", [(0, 10_000, 0.10), (2, 60_000, 0.10), (4, 20_000, 0.35), (6, 5000, np.inf)])
def test_ckdtree_backend_is_the_sklearn_oracle(seed, n, gate):
    """oracle.GoldIndex(backend="ckdtree") — what the long -m gpu comparisons run on — against the default
    scikit-learn backend: the same nearest neighbour for every query inside the gate (indices identical, squared distances
    to rounding) in a one-step query, and the same loop — identical inlier counts in every iteration, transforms equal to
    rounding — over 20 iterations.  (A query the gate cuts off comes back as (0, inf) and weighs nothing.)"""
    gold, comp, R0, t0 = syn.make_pair(seed, n)
    a = orc.GoldIndex(gold, n_jobs=1)
    b = orc.GoldIndex(gold, n_jobs=2, backend="ckdtree")
    ja, da = a.query(comp)
    jb, db = b.query(comp, radius=gate)
    inside = da <= gate ** 2 if np.isfinite(gate) else np.ones(len(da), bool)
    assert np.array_equal(ja[inside], jb[inside]) and np.allclose(da[inside], db[inside], rtol=1e-14, atol=0)
    assert np.isinf(db[~inside]).all()
    jc, dc = b.query(comp)                                    # without a bound: the full answer
    assert np.array_equal(ja, jc) and np.allclose(da, dc, rtol=1e-14, atol=0)
    ra = orc.icp_oracle(gold, comp, max_iter=20, max_corr_dist=gate, index=a)
    rb = orc.icp_oracle(gold, comp, max_iter=20, max_corr_dist=gate, index=b)
    assert [r["inliers"] for r in ra["trace"]] == [r["inliers"] for r in rb["trace"]]
    assert np.allclose([r["acc"] for r in ra["trace"]], [r["acc"] for r in rb["trace"]], rtol=1e-12, atol=1e-12)
    eR, et = orc.rt_error(ra["R"], ra["t"], rb["R"], rb["t"])
    assert eR < 1e-12 and et < 1e-12, (eR, et)


def test_exact_sums_restatement_against_float_arithmetic():
    """oracle/exact_sums.py (rational arithmetic) on a small case: its integers, read back as numbers, are the plain fp64
    sums to within n quanta, its fma is numpy's where no rounding is at stake, ties round to even."""
    from fractions import Fraction

    from oracle import exact_sums as xs
    assert xs.rne_int(Fraction(5, 2)) == 2 and xs.rne_int(Fraction(7, 2)) == 4 and xs.rne_int(Fraction(-5, 2)) == -2
    assert xs.rne_int(Fraction(-7, 3)) == -2 and xs.rne_int(Fraction(8, 3)) == 3
    assert xs.fma(3.0, 4.0, 0.5) == 12.5 and xs.fma(1.0 + 2.0 ** -52, 1.0 + 2.0 ** -52, -1.0) == 2.0 ** -51 + 2.0 ** -104
    rng = np.random.default_rng(2)
    n = 300
    src = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    gold = (src + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    R = syn.rotation_about_axis([0.2, 0.5, 0.8], 0.01)
    t = np.array([0.01, -0.02, 0.005])
    e = [40, 37, 37, 34, 47]
    k = xs.sums(R, t, src, None, gold, np.arange(n), e)
    P = src.astype(np.float64) @ R.T + t
    acc = orc.accumulate(P, gold.astype(np.float64), np.ones(n), ((P - gold) ** 2).sum(1))
    back = np.array([k[a] / 2.0 ** e[xs.GROUP[a]] for a in range(17)])
    quantum = np.array([2.0 ** -e[xs.GROUP[a]] for a in range(17)])
    assert (np.abs(back - acc) <= n * quantum + 1e-9 * np.abs(acc)).all()
    assert k[0] == n * 2 ** 40
