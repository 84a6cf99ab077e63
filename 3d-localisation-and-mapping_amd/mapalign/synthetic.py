"""Seeded synthetic workloads for the map-alignment path.

The reference ships no scans (``/root/reference/.gitignore:2-11`` ignores every data
file), so the workloads are synthesised: a railway-carriage-like shell, elongated like
the maps in ``readme_img/alignment_visualisation.gif``.

* gold map: area-weighted uniform samples on the six faces of a 20 m x 3 m x 2.5 m box
  plus 40 interior axis-aligned rectangular panels (0.5-1.5 m sides) that break the
  symmetry, isotropic Gaussian jitter sigma = 1 mm, stored fp32;
* comparison map: an *independent* resample of the same surfaces moved by the inverse
  of the ground-truth transform (2 deg about (0.3,-0.5,0.81)/|.| through the gold
  centroid, then t0 = (0.10,-0.06,0.04) m), rows permuted;
* depth frame (config C3): analytic ray/rectangle render of the same shell from a
  camera inside the carriage, float32 millimetres exactly as
  ``src/detector/dataset.py:68-81`` hands them to the mapper.

This module is workload generation only: no alignment arithmetic lives here.
"""
from __future__ import annotations

import numpy as np

BOX = (20.0, 3.0, 2.5)
N_PANELS = 40
JITTER_SIGMA = 1e-3
GT_AXIS = np.array([0.3, -0.5, 0.81])
GT_ANGLE_DEG = 2.0
GT_T0 = np.array([0.10, -0.06, 0.04])


def rotation_about_axis(axis, angle_rad):
    """Rodrigues rotation matrix (fp64)."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
    return np.eye(3) + np.sin(angle_rad) * K + (1.0 - np.cos(angle_rad)) * (K @ K)


def carriage_rectangles(rng, box=BOX, n_panels=N_PANELS):
    """Rectangles as (origin[3], edge_u[3], edge_v[3]) rows: 6 box faces + interior panels."""
    L, W, H = box
    rects = []
    ext = np.array([L, W, H])
    for axis in range(3):
        u_ax, v_ax = [a for a in range(3) if a != axis]
        for side in (0.0, ext[axis]):
            o = np.zeros(3)
            o[axis] = side
            u = np.zeros(3)
            u[u_ax] = ext[u_ax]
            v = np.zeros(3)
            v[v_ax] = ext[v_ax]
            rects.append((o, u, v))
    for _ in range(n_panels):
        axis = int(rng.integers(0, 3))
        u_ax, v_ax = [a for a in range(3) if a != axis]
        su, sv = rng.uniform(0.5, 1.5, size=2)
        su = min(su, ext[u_ax] * 0.9)
        sv = min(sv, ext[v_ax] * 0.9)
        o = np.zeros(3)
        o[axis] = rng.uniform(0.1 * ext[axis], 0.9 * ext[axis])
        o[u_ax] = rng.uniform(0.0, ext[u_ax] - su)
        o[v_ax] = rng.uniform(0.0, ext[v_ax] - sv)
        u = np.zeros(3)
        u[u_ax] = su
        v = np.zeros(3)
        v[v_ax] = sv
        rects.append((o, u, v))
    return rects


def sample_rectangles(rects, n, rng, sigma=JITTER_SIGMA):
    """Area-weighted uniform samples on the rectangles + isotropic jitter -> (n,3) fp64."""
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in rects])
    counts = rng.multinomial(n, areas / areas.sum())
    out = np.empty((n, 3), dtype=np.float64)
    pos = 0
    for (o, u, v), k in zip(rects, counts):
        if k == 0:
            continue
        ab = rng.random((k, 2))
        out[pos:pos + k] = o + ab[:, :1] * u + ab[:, 1:] * v
        pos += k
    out += rng.normal(0.0, sigma, size=out.shape)
    return out


def ground_truth(gold_centroid, axis=GT_AXIS, angle_deg=GT_ANGLE_DEG, t0=GT_T0):
    """(R0, t) with gold ~= R0 @ comparison + t; rotation is about the gold centroid."""
    R0 = rotation_about_axis(axis, np.deg2rad(angle_deg))
    c = np.asarray(gold_centroid, dtype=np.float64)
    t = c + np.asarray(t0, dtype=np.float64) - R0 @ c
    return R0, t


def make_pair(seed, n_gold, n_comp=None, *, resample=True, angle_deg=GT_ANGLE_DEG,
              t0=GT_T0, sigma=JITTER_SIGMA, permute=True):
    """Gold/comparison cloud pair with a known rigid transform.

    Returns ``(gold f32 [M,3], comparison f32 [N,3], R0 f64 [3,3], t f64 [3])`` where
    ``gold ~= R0 @ comparison + t``. ``resample=False`` makes the comparison cloud the
    *same* points as the gold cloud (closed-form known answer); ``resample=True`` is the
    benchmark workload (independent resample of the same surfaces).
    """
    n_comp = n_gold if n_comp is None else n_comp
    rng = np.random.default_rng(seed)
    rects = carriage_rectangles(rng)
    gold = sample_rectangles(rects, n_gold, rng, sigma)
    centroid = gold.mean(axis=0)
    R0, t = ground_truth(centroid, angle_deg=angle_deg, t0=t0)
    if resample:
        comp_in_gold = sample_rectangles(rects, n_comp, rng, sigma)
    else:
        if n_comp != n_gold:
            raise ValueError("resample=False needs n_comp == n_gold")
        comp_in_gold = gold.copy()
    comp = (comp_in_gold - t) @ R0  # row form of R0^T (x - t)
    if permute:
        comp = comp[rng.permutation(n_comp)]
    return gold.astype(np.float32), comp.astype(np.float32), R0, t


from .transforms import pose_to_matrix, quat_to_rot  # noqa: E402,F401  (the pose math lives with the other transform helpers)


def default_camera_pose():
    """Camera inside the carriage, optical axis (+Z) looking down the long (+x) axis."""
    # camera x -> world -y, camera y -> world -z, camera z -> world +x
    Rcw = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    T = np.eye(4)
    T[:3, :3] = Rcw
    T[:3, 3] = (2.0, 1.4, 1.3)
    return T


def render_depth(rects, T_cam_to_world, width, height, fx, fy, cx, cy, *, chunk=1 << 18,
                 max_range=60.0):
    """Analytic z-buffer of the rectangle set -> float32 depth in MILLIMETRES (0 = no hit).

    Depth is the camera-frame Z of the nearest hit, the convention of the pinhole model in
    SURVEY.md §8a-8 (``Z = depth / scale_depth``).
    """
    T = np.asarray(T_cam_to_world, dtype=np.float64)
    Rcw, tcw = T[:3, :3], T[:3, 3]
    ys, xs = np.mgrid[0:height, 0:width]
    dirs_cam = np.stack([(xs.ravel() - cx) / fx, (ys.ravel() - cy) / fy,
                         np.ones(width * height)], axis=1)  # Z = 1 parametrisation
    depth = np.zeros(width * height, dtype=np.float64)
    for s in range(0, dirs_cam.shape[0], chunk):
        d = dirs_cam[s:s + chunk] @ Rcw.T  # world-frame direction per unit camera Z
        best = np.full(d.shape[0], np.inf)
        for o, u, v in rects:
            nrm = np.cross(u, v)
            denom = d @ nrm
            with np.errstate(divide="ignore", invalid="ignore"):
                lam = ((o - tcw) @ nrm) / denom
                hit = tcw + lam[:, None] * d - o
                a = (hit @ u) / (u @ u)
                b = (hit @ v) / (v @ v)
            ok = (np.abs(denom) > 1e-12) & (lam > 1e-3) & (lam < max_range) \
                & (a >= 0) & (a <= 1) & (b >= 0) & (b <= 1)
            best = np.where(ok & (lam < best), lam, best)
        depth[s:s + chunk] = np.where(np.isfinite(best), best, 0.0)
    return (depth.reshape(height, width) * 1000.0).astype(np.float32)


def make_depth_frame(seed, width=1280, height=960, fx=1000.0, fy=1000.0, cx=640.0,
                     cy=480.0, T_cam_to_world=None):
    """Config C3 input: (depth_mm f32 [H,W], K=(fx,fy,cx,cy), T_cam_to_world 4x4)."""
    rng = np.random.default_rng(seed)
    rects = carriage_rectangles(rng)
    T = default_camera_pose() if T_cam_to_world is None else np.asarray(T_cam_to_world)
    depth = render_depth(rects, T, width, height, fx, fy, cx, cy)
    return depth, (fx, fy, cx, cy), T
