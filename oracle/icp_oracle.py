"""CPU oracle for the map-alignment hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``mapalign``) never does.

PARITY UNPINNED.  The reference's implementation of this path,
``src/map_alignment/align.py``, is absent from the snapshot (it is imported at
``/root/reference/task_def.py:10`` and called at ``task_def.py:175-187``, but
``src/map_alignment/`` holds only ``visualise.py``); the reference has no tests, golden
vectors or data for it either.  This file is therefore the textbook
point-to-point ICP that ``BASELINE.json:north_star`` names as "the reference's
NumPy/scikit-learn path", written in fp64 NumPy + ``sklearn.neighbors.NearestNeighbors``
and pinned by closed-form known-answer tests (``tests/test_oracle.py``), not by
reference outputs.  What the reference *does* fix, and what is followed here:

* transform conventions — a rigid transform is either a 4x4 homogeneous matrix applied
  as ``p' = T @ [p,1]`` or a ``(R, centre)`` tuple meaning ``p' = R (p - c) + c``
  (``src/map_alignment/visualise.py:82-96,118-130``);
* pinhole unprojection and pose conventions — intrinsics rescaled to depth resolution
  (``src/mapper/pose_processor.py:133-137``), ``_depth_to_3d(x, y, depth, fx, fy, cx, cy,
  scale_depth)`` call shape (``pose_processor.py:184-196``), camera->world
  ``g = (T @ [p,1])[:3]`` (``pose_processor.py:242-260``), depth in float32 millimetres
  (``src/detector/dataset.py:68-81``), pose row ``tx ty tz qx qy qz qw``
  (``src/mapper/database_query.py:20-25``).

Algorithm per iteration k (rows a4-a7 of SURVEY.md §8a), state (R_k, t_k):
  a4  p'_i = R_k p_i + t_k ;  j(i) = argmin_j |p'_i - q_j|^2 ,  d_i^2
  a5  w_i = weight_i * 1[d_i <= max_corr_dist]
      acc = [W, s_p(3), s_q(3), S(9 row-major, S_ab = sum w p'_a q_b), E]
  a6  mu_p = s_p/W, mu_q = s_q/W, H = S - W mu_p mu_q^T = U Sig V^T,
      dR = V diag(1,1,det(V U^T)) U^T, dt = mu_q - dR mu_p
  a7  R_{k+1} = dR R_k, t_{k+1} = dR t_k + dt ; rmse_k = sqrt(E/W) (before the update);
      stop after iteration k>=1 when |rmse_k - rmse_{k-1}| < tol (tol = 0 never stops).
"""
from __future__ import annotations

import numpy as np

ACC_LEN = 17


def kabsch_from_acc(acc):
    """Rows a6: 17-double accumulator -> (dR, dt, ok).

    ok=False when a sum is not finite, when there is no weight (W <= 0) or when rank(H) < 2 —
    what fewer than three non-collinear correspondences give.  W is a sum of weights, not a
    count, so there is no "W >= 3" test; H below 1e-12 of the size of the terms it is the
    difference of (W |mu_p| |mu_q|) is rounding noise and counts as rank 0."""
    acc = np.asarray(acc, dtype=np.float64)
    W = acc[0]
    if not np.isfinite(acc).all() or not W > 0.0:
        return np.eye(3), np.zeros(3), False
    mu_p = acc[1:4] / W
    mu_q = acc[4:7] / W
    H = acc[7:16].reshape(3, 3) - W * np.outer(mu_p, mu_q)
    U, S, Vt = np.linalg.svd(H)
    noise = W * np.linalg.norm(mu_p) * np.linalg.norm(mu_q)
    if S[1] <= 1e-12 * max(S[0], 1e-300) or S[0] <= 1e-12 * noise:
        return np.eye(3), mu_q - mu_p, False
    V = Vt.T
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(V @ U.T)) or 1.0])
    dR = V @ D @ U.T
    return dR, mu_q - dR @ mu_p, True


def accumulate(p_t, q, w, d2):
    """Row a5: fp64 accumulator from transformed source p_t, matched gold q, weights w."""
    p_t = np.asarray(p_t, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    acc = np.empty(ACC_LEN)
    acc[0] = w.sum()
    acc[1:4] = (w[:, None] * p_t).sum(axis=0)
    acc[4:7] = (w[:, None] * q).sum(axis=0)
    acc[7:16] = ((w[:, None] * p_t).T @ q).ravel()
    acc[16] = (w * d2).sum()
    return acc


class GoldIndex:
    """KD-tree over the gold cloud (built once per alignment; row a4).

    ``backend="sklearn"`` (default): ``sklearn.neighbors.NearestNeighbors(kd_tree)`` — the "NumPy/scikit-learn path"
    north_star names, and what ``bench.py`` times as the CPU baseline.  ``backend="ckdtree"``:
    ``scipy.spatial.cKDTree`` — the same exact nearest neighbour (tests/test_oracle.py holds the two to identical
    indices), 4-7 times faster per query on the same cores, and able to stop at the gate: with ``radius`` set, a query
    without a gold point inside it comes back as (index 0, d2 = inf), which the gate of ``icp_step`` gives weight 0
    either way.  The long equal-iteration-count comparisons of the -m gpu suite (1M and 2M points, 50 iterations) use it
    so that they fit the driver's time limit at FULL length; nothing about the arithmetic of rows a5-a7 changes."""

    def __init__(self, gold, n_jobs=-1, backend="sklearn"):
        self.gold = np.ascontiguousarray(gold, dtype=np.float64)
        if self.gold.ndim != 2 or self.gold.shape[1] != 3 or self.gold.shape[0] == 0:
            raise ValueError("gold cloud must be a non-empty [M,3] array")
        self.backend = backend
        self.n_jobs = n_jobs
        if backend == "sklearn":
            from sklearn.neighbors import NearestNeighbors

            self.nn = NearestNeighbors(n_neighbors=1, algorithm="kd_tree", n_jobs=n_jobs)
            self.nn.fit(self.gold)
        elif backend == "ckdtree":
            from scipy.spatial import cKDTree

            self.tree = cKDTree(self.gold, leafsize=16)
        else:
            raise ValueError("backend must be 'sklearn' or 'ckdtree'")

    def query(self, pts, radius=None):
        """(index of the nearest gold point, squared distance) per query.  ``radius`` (ckdtree only; sklearn ignores it
        and answers in full): nothing farther than this is looked for — such a query returns (0, inf)."""
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        if self.backend == "sklearn":
            d, j = self.nn.kneighbors(pts)
            return j[:, 0].astype(np.int64), d[:, 0] ** 2
        bound = np.inf if radius is None or not np.isfinite(radius) else float(radius) * (1.0 + 1e-9) + 1e-300
        d, j = self.tree.query(pts, k=1, workers=self.n_jobs if self.n_jobs else 1, distance_upper_bound=bound)
        j = j.astype(np.int64)
        j[j >= len(self.gold)] = 0          # "nothing inside the bound": d is inf there
        return j, d ** 2


def nn_bruteforce(gold, pts, chunk=2048):
    """Exhaustive argmin (small cases; checks the KD-tree itself)."""
    gold = np.asarray(gold, dtype=np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    idx = np.empty(len(pts), dtype=np.int64)
    d2 = np.empty(len(pts))
    for s in range(0, len(pts), chunk):
        diff = pts[s:s + chunk, None, :] - gold[None, :, :]
        dd = np.einsum("ijk,ijk->ij", diff, diff)
        idx[s:s + chunk] = dd.argmin(axis=1)
        d2[s:s + chunk] = dd.min(axis=1)
    return idx, d2


def as_matrix(init):
    """None | 4x4 | (R, t) -> (R, t) fp64."""
    if init is None:
        return np.eye(3), np.zeros(3)
    if isinstance(init, tuple):
        R, t = init
        return np.array(R, dtype=np.float64), np.array(t, dtype=np.float64)
    T = np.asarray(init, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("init must be None, a 4x4 matrix or an (R, t) tuple")
    return T[:3, :3].copy(), T[:3, 3].copy()


def icp_step(index, comp, R, t, max_corr_dist=0.10, weights=None):
    """One iteration at state (R, t): returns (acc17, idx, d2, R_next, t_next, ok)."""
    comp = np.asarray(comp, dtype=np.float64)
    p_t = comp @ R.T + t
    idx, d2 = index.query(p_t, radius=max_corr_dist)
    gate = d2 <= float(max_corr_dist) ** 2 if np.isfinite(max_corr_dist) else np.ones(len(d2), bool)
    w = gate.astype(np.float64)
    if weights is not None:
        w = w * np.asarray(weights, dtype=np.float64)
    acc = accumulate(p_t, index.gold[idx], w, np.where(gate, d2, 0.0))      # (a query cut off at the gate carries d2 = inf and w = 0)
    dR, dt, ok = kabsch_from_acc(acc)
    if not ok:
        return acc, idx, d2, R, t, False
    return acc, idx, d2, dR @ R, dR @ t + dt, True


def icp_oracle(gold, comp, max_iter=50, tol=0.0, max_corr_dist=0.10, init=None,
               weights=None, n_jobs=-1, index=None, keep_nn=False, backend="sklearn"):
    """Full loop.  Returns dict(R, t, iterations, rmse, inliers, converged, ok, trace).

    ``trace`` is a list of per-iteration dicts {acc, rmse, inliers[, idx, d2]} — the
    17-double accumulators are what the HIP path's ``ma_get_trace`` must reproduce.
    """
    comp = np.asarray(comp, dtype=np.float64)
    if comp.ndim != 2 or comp.shape[1] != 3 or comp.shape[0] == 0:
        raise ValueError("comparison cloud must be a non-empty [N,3] array")
    index = GoldIndex(gold, n_jobs=n_jobs, backend=backend) if index is None else index
    R, t = as_matrix(init)
    trace = []
    converged = False
    ok = True
    prev_rmse = None
    for _ in range(int(max_iter)):
        acc, idx, d2, Rn, tn, ok = icp_step(index, comp, R, t, max_corr_dist, weights)
        rmse = float(np.sqrt(acc[16] / acc[0])) if acc[0] > 0 else float("nan")
        row = {"acc": acc, "rmse": rmse, "inliers": float(acc[0])}
        if keep_nn:
            row["idx"], row["d2"] = idx, d2
        trace.append(row)
        if not ok:
            break
        R, t = Rn, tn
        if prev_rmse is not None and abs(rmse - prev_rmse) < tol:
            converged = True
            break
        prev_rmse = rmse
    last = trace[-1] if trace else {"rmse": float("nan"), "inliers": 0.0}
    return {"R": R, "t": t, "iterations": len(trace), "rmse": last["rmse"],
            "inliers": last["inliers"], "converged": converged, "ok": ok, "trace": trace}


def unproject_oracle(depth, K, T_cam_to_world, depth_scale=1000.0):
    """Row a8: depth image -> world points (valid pixels only, row-major pixel order).

    Z = depth[y,x]/depth_scale; X = (x-cx) Z/fx; Y = (y-cy) Z/fy; g = R [X,Y,Z]^T + t.
    Pixels with depth <= 0, NaN or inf are dropped.  Returns (points f64 [V,3], mask [H,W]).
    """
    depth = np.asarray(depth)
    fx, fy, cx, cy = (float(v) for v in K)
    T = np.asarray(T_cam_to_world, dtype=np.float64).reshape(4, 4)
    h, w = depth.shape
    z = depth.astype(np.float64) / float(depth_scale)
    mask = np.isfinite(z) & (z > 0)
    ys, xs = np.nonzero(mask)
    zz = z[ys, xs]
    cam = np.stack([(xs - cx) * zz / fx, (ys - cy) * zz / fy, zz], axis=1)
    return cam @ T[:3, :3].T + T[:3, 3], mask


def voxel_centroids_oracle(points, voxel, origin):
    """Row N4: centroid of the points of each voxel (Open3D voxel_down_sample semantics).
    Voxel index = floor((p - origin) / voxel) evaluated in float32 like the device does; output
    sorted by (z, y, x) voxel index."""
    p = np.asarray(points, dtype=np.float64)
    rel = (p - np.asarray(origin, dtype=np.float64)).astype(np.float32)
    idx = np.floor(rel * np.float32(1.0 / np.float32(voxel))).astype(np.int64)
    key = ((idx[:, 2] + (1 << 20)) << 42) | ((idx[:, 1] + (1 << 20)) << 21) | (idx[:, 0] + (1 << 20))
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.concatenate([[True], ks[1:] != ks[:-1]])
    seg = np.cumsum(heads) - 1
    out = np.zeros((seg[-1] + 1, 3))
    cnt = np.bincount(seg)
    relo = rel[order].astype(np.float64)
    for a in range(3):
        out[:, a] = np.bincount(seg, weights=relo[:, a]) / cnt
    return out + np.asarray(origin, dtype=np.float64)


def lift_boxes_oracle(boxes_xyxy, depth, K, T_cam_to_world, depth_scale=1000.0, z_rule=0):
    """Row a10: corners (x1,y1) (x2,y1) (x2,y2) (x1,y2) at int() pixels (clamped into the
    image) -> global points; call shape of pose_processor.py:181-201.  ``z_rule`` 0: the depth at
    the corner pixel (what the call at :184-196 is handed); 1: the median of the valid depths over
    the box's pixel rectangle, ends included (the comment at :183), the same z for the four
    corners.  Returns (corners [B,4,3], valid [B,4])."""
    depth = np.asarray(depth)
    h, w = depth.shape
    fx, fy, cx, cy = (float(v) for v in K)
    T = np.asarray(T_cam_to_world, dtype=np.float64).reshape(4, 4)
    b = np.asarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
    out = np.full((len(b), 4, 3), np.nan)
    valid = np.zeros((len(b), 4), dtype=bool)
    for i, (x1, y1, x2, y2) in enumerate(b):
        zmed = None
        if z_rule == 1:
            ua, ub = (min(max(int(x), 0), w - 1) for x in (x1, x2))
            va, vb = (min(max(int(y), 0), h - 1) for y in (y1, y2))
            patch = depth[min(va, vb):max(va, vb) + 1, min(ua, ub):max(ua, ub) + 1].astype(np.float64) / float(depth_scale)
            ok = np.isfinite(patch) & (patch > 0)
            zmed = float(np.median(patch[ok])) if ok.any() else float("nan")
        for k, (x, y) in enumerate(((x1, y1), (x2, y1), (x2, y2), (x1, y2))):
            u, v = min(max(int(x), 0), w - 1), min(max(int(y), 0), h - 1)
            z = zmed if zmed is not None else float(depth[v, u]) / float(depth_scale)
            if np.isfinite(z) and z > 0:
                cam = np.array([(u - cx) * z / fx, (v - cy) * z / fy, z])
                out[i, k] = T[:3, :3] @ cam + T[:3, 3]
                valid[i, k] = True
    return out, valid


def rt_error(R, t, R_ref, t_ref):
    """(||R-R_ref||_F, ||t-t_ref||_2) — the two parity figures of BASELINE.json."""
    return (float(np.linalg.norm(np.asarray(R) - np.asarray(R_ref))),
            float(np.linalg.norm(np.asarray(t) - np.asarray(t_ref))))
