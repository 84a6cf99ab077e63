"""Coarse pre-alignment stage: brings a maintenance scan that starts
far from the gold map inside ICP's basin, and emits the staged transforms the reference's
visualiser replays (``src/map_alignment/visualise.py:82-96``; README GIF: the comparison map
first ROTATES about a centre, then TRANSLATES onto the gold map, then is refined).

Stage 1  rotation about the comparison centroid that maps its principal axes onto the gold
         map's (carriages are strongly elongated, so the frames are well defined); the
         four proper sign choices are scored on the GPU — one ungated NN/accumulate step
         each (``ma_icp_step``) on a sub-sample — and the lowest rmse wins  -> (R, centre)
Stage 2  translation of the rotated centroid onto the gold centroid           -> 4x4
Stage 3  ICP refinement (mapalign.align)                                       -> 4x4

The camera trajectories the reference hands to ``Alignment`` (``base_pose_df`` /
``comparison_pose_df``, ``/root/reference/task_def.py:176-177``; columns ``timestamp tx ty tz qx
qy qz qw``, ``src/mapper/database_query.py:20-25``) decide what geometry alone cannot: a
carriage shell is nearly symmetric under a half turn about its vertical axis, so the PCA
candidates that differ by 180 degrees can score almost equally on the clouds.  Both scans
are walked through the vehicle along a trajectory lying INSIDE the cloud, so the candidate
must also map the comparison trajectory onto the gold one (direction of travel, which side
of the centre line the camera was carried on); the trajectory residual is added to the
cloud score.

The reference's own coarse stage (inside the absent align.py) is unknown; only its
output format is pinned.  This is host-side glue (3x3 eigen-decompositions); all cloud
work goes through the HIP engine.
"""
from __future__ import annotations

import numpy as np

from . import transforms


PCA_MAX_POINTS = 1 << 18
BLAS_THREADS = 8
NARROW_CELLS = 5.5      # first scoring bound, in cells of the gold grid (inside the engine's ring walk and distance table)
DECISIVE = 0.6          # the narrow scores decide if the best is below this fraction of the runner-up

_blas_controller = None


def _few_blas_threads():
    """Context manager capping the BLAS thread pool for the two small products below.  A [1M, 3] matrix-vector
    product needs no more than a few threads, but OpenBLAS fans it out over every hardware thread of the host
    (256 on an MI355X box); under a CPU quota (containers, batch schedulers) that burst exhausts the quota and
    the WHOLE process is throttled for the rest of the scheduling period — measured as a 40-80 ms stall in
    whichever call came 20-30 ms later (`tools/coarse_times.py`).  The controller is built once (it scans the
    loaded libraries); without threadpoolctl this is a no-op."""
    global _blas_controller
    import contextlib
    if _blas_controller is None:
        try:
            from threadpoolctl import ThreadpoolController
            _blas_controller = ThreadpoolController()
        except Exception:           # pragma: no cover - optional dependency
            _blas_controller = False
    if not _blas_controller:
        return contextlib.nullcontext()
    return _blas_controller.limit(limits=BLAS_THREADS, user_api="blas")


def principal_frame(points, max_points=PCA_MAX_POINTS):
    """(centroid, axes [3,3] columns by decreasing variance, right-handed, variances).

    Centroid and covariance are taken over every k-th point once the cloud holds more than
    ``max_points`` (a fixed stride, so the result is repeatable): the axes of a million-point scan do
    not change in the fourth digit, and the two fp64 passes over the full clouds were three quarters of
    the whole coarse stage (2 x 33 ms of 88 ms, 1M-point pair)."""
    with _few_blas_threads():
        return _principal_frame(points, max_points)


def _principal_frame(points, max_points):
    p = np.asarray(points)
    # every k-th point once the cloud holds more than max_points (a fixed stride: repeatable) — for the centroid as
    # well (round 3): the mean of 262 144 points of a 20 m scan is within a centimetre of the mean of all of them, the
    # stage only has to land inside the basin of the gated loop that follows, and the full-cloud mean was half of this
    # function's 5 ms per million points (a [N,3] axis-0 reduction runs at a tenth of NumPy's contiguous speed)
    if len(p) > max_points:
        p = p[::-(-len(p) // max_points)]
    p = np.ascontiguousarray(p, dtype=np.float64)
    c = (np.ones(len(p)) @ p) / max(len(p), 1)          # column sums as ONE matrix-vector product
    d = p - c
    # Stray points far from the scene (reflections, sensor noise: the reference's ingest removes none) would own the
    # covariance: 20 of them fifty scene-widths away outweigh a million points of the scene.  Where the bounding box
    # reaches beyond six standard deviations of the mean (a filled box reaches 1.7, a shell less), the frame is that of
    # the cloud's BULK — the rule of the engine's grid (csrc/host_math.h "Robust grid box"): per axis the central 90 % of
    # the points, widened by a quarter of its width.
    if len(p) >= 64:
        sd = np.sqrt((d * d).sum(axis=0) / len(p))
        if (np.abs(d).max(axis=0) > 6.0 * sd + 1e-12).any():
            lo, hi = np.quantile(p, 0.05, axis=0), np.quantile(p, 0.95, axis=0)
            pad = 0.25 * (hi - lo) + 1e-9
            keep = ((p >= lo - pad) & (p <= hi + pad)).all(axis=1)
            if 0.9 * len(p) <= keep.sum() < len(p):
                p = p[keep]
                c = (np.ones(len(p)) @ p) / len(p)
                d = p - c
    w, v = np.linalg.eigh(d.T @ d / max(len(d) - 1, 1))
    order = np.argsort(w)[::-1]
    axes = v[:, order]
    if np.linalg.det(axes) < 0:
        axes[:, 2] = -axes[:, 2]
    return c, axes, w[order]


def frame_from_moments(centroid, cov):
    """(centroid, axes, variances) as ``principal_frame`` from a centroid and a 3x3 covariance
    (``MapAligner.cloud_moments``: the sums taken on the device over the whole stored cloud)."""
    w, v = np.linalg.eigh(np.asarray(cov, dtype=np.float64))
    order = np.argsort(w)[::-1]
    axes = v[:, order]
    if np.linalg.det(axes) < 0:
        axes[:, 2] = -axes[:, 2]
    return np.asarray(centroid, dtype=np.float64), axes, w[order]


def candidate_rotations(axes_gold, axes_comp):
    """The four proper rotations taking the comparison principal frame onto the gold one."""
    out = []
    for sx, sy in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        S = np.diag([sx, sy, sx * sy])          # det +1
        out.append(axes_gold @ S @ axes_comp.T)
    return out


def trajectory_xyz(pose_df):
    """``[K,3]`` camera positions from a pose table: a DataFrame / mapping with ``tx ty tz``
    columns (``src/mapper/database_query.py:22``) or an array whose first three columns —
    after an optional leading timestamp column, as the reference slices ``row[1:]``
    (``pose_processor.py:109``) — are the position.  ``None`` -> ``None``."""
    if pose_df is None:
        return None
    if hasattr(pose_df, "columns") or isinstance(pose_df, dict):
        cols = [np.asarray(pose_df[k], dtype=np.float64) for k in ("tx", "ty", "tz")]
        xyz = np.stack(cols, axis=1)
    else:
        a = np.asarray(pose_df, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("pose array must be [K, >=3]")
        xyz = a[:, 1:4] if a.shape[1] in (8, 9) else a[:, :3]
    xyz = xyz[np.all(np.isfinite(xyz), axis=1)]
    return xyz if len(xyz) >= 2 else None


def resample_polyline(xyz, n=64):
    """``n`` points equally spaced in arc length along a trajectory (scans are recorded at
    different speeds, so samples are compared by the fraction of the path walked)."""
    seg = np.linalg.norm(np.diff(xyz, axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    if s[-1] <= 0.0:
        return np.repeat(xyz[:1], n, axis=0)
    u = np.linspace(0.0, s[-1], n)
    return np.stack([np.interp(u, s, xyz[:, a]) for a in range(3)], axis=1)


def trajectory_residual(T, base_traj, comp_traj, n=64):
    """rms distance between the gold trajectory and the transformed comparison trajectory,
    both resampled by arc length; the better of the two walking directions counts (a scan
    may be recorded from either end of the vehicle)."""
    a = resample_polyline(base_traj, n)
    b = resample_polyline(transforms.apply(T, comp_traj), n)
    fwd = np.sqrt(np.mean(np.sum((a - b) ** 2, axis=1)))
    rev = np.sqrt(np.mean(np.sum((a - b[::-1]) ** 2, axis=1)))
    return float(min(fwd, rev))


def coarse_align(engine, gold_cloud, comparison_cloud, sample=20000, seed=0, base_pose_df=None,
                 comparison_pose_df=None, truncate=None, resident=False):
    """Returns (stages, T_coarse, info); ``engine`` is a MapAligner with the gold map set.

    ``resident=True``: the engine already holds BOTH clouds (``set_pair``), and it still does when this returns.
    Then nothing of the stage walks the clouds on the host: the principal frames come from the device's moment
    sums over the whole stored clouds (``cloud_moments``; 30 us each against 5-6 ms per million points in NumPy),
    and the narrow-bound score of a candidate is one search + accumulate launch over the whole comparison cloud
    (``truncated_rms``: the sums come back, not a distance per point).  Only an undecided narrow round falls back
    to the sub-sample and the wide bound, and puts the full comparison cloud back afterwards.

    ``stages`` = [(R, centre), T_translate] in the reference's mixed format.  With both
    trajectories given, each candidate's score is ``sqrt(cloud_rms^2 + trajectory_rms^2)``.

    The cloud score of a candidate is the rms of its nearest-neighbour distances TRUNCATED at
    ``truncate`` metres (default: 5 % of the gold map's bounding-box diagonal): the right
    candidate leaves centimetres, a wrong one decimetres to metres, and the truncation both
    keeps a few outlying parts of a wrong candidate from dominating and bounds the search (an
    ungated exact search for points that lie metres from the map walks thousands of grid rows
    each: 5-9 ms per wrong candidate for 20k samples against a 1M-point map, 3-5 ms truncated
    at 1 m; the right candidate costs 0.15 ms either way).
    """
    gold = np.asarray(gold_cloud)
    comp = np.asarray(comparison_cloud)
    if resident:
        (_, cg, Cg), (_, cc, Cc) = engine.cloud_moments(0), engine.cloud_moments(1)
        (cg, ag, _), (cc, ac, _) = frame_from_moments(cg, Cg), frame_from_moments(cc, Cc)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as pool:          # the two frames side by side (NumPy releases the interpreter lock)
            fg, fc = pool.submit(principal_frame, gold), pool.submit(principal_frame, comp)
            (cg, ag, _), (cc, ac, _) = fg.result(), fc.result()

    def subsample():
        rng = np.random.default_rng(seed)
        return comp if len(comp) <= sample else comp[rng.choice(len(comp), sample, replace=False)]
    base_traj, comp_traj = trajectory_xyz(base_pose_df), trajectory_xyz(comparison_pose_df)
    use_traj = base_traj is not None and comp_traj is not None
    cands = candidate_rotations(ag, ac)
    gi = engine.grid_info()                                      # the gold map's box, as the engine gridded it
    if truncate is None:
        truncate = 0.05 * float(np.linalg.norm(np.asarray(gi["dims"], dtype=np.float64) * gi["cell"]))
    truncate = max(float(truncate), 1e-6)
    Ts = [transforms.rt_to_matrix(np.eye(3), cg - cc) @ transforms.rotation_about_centre(R, cc) for R in cands]
    traj_scores = [trajectory_residual(T, base_traj, comp_traj) if use_traj else 0.0 for T in Ts]

    def cloud_score(k, bound, sums_only):
        if sums_only:
            return engine.truncated_rms(Ts[k], bound)
        _, d2 = engine.nn(Ts[k], max_corr_dist=bound)           # inf where nothing lies within the bound
        return float(np.sqrt(np.mean(np.minimum(d2.astype(np.float64), bound * bound))))

    def all_scores(bound, sums_only=False):
        cs = [cloud_score(k, bound, sums_only) for k in range(len(cands))]
        return cs, [float(np.hypot(cs[k], traj_scores[k])) for k in range(len(cands))]

    # A wrong candidate is what costs: its points lie decimetres to metres from the map, and an exact search
    # bounded at `truncate` (1 m for a 20 m carriage) walks hundreds of grid rows per point — 3-4 ms per wrong
    # candidate, whether 2 000 or 20 000 points are scored (a chain of dependent reads per wavefront), plus the
    # coarser grid levels such a wide search makes the engine build.  Round 3: score first with the bound at
    # NARROW_CELLS cells of the gold grid.  There the engine's distance-to-occupied table settles "nothing within
    # the bound" with one byte per point, nothing wider than its ring walk is needed, and a candidate costs ~0.1 ms.
    # The narrow scores decide when they are decisive — the best below DECISIVE x the runner-up; otherwise (a sloppy
    # coarse pose leaves the right candidate decimetres off, and every score saturates at the bound) the wide
    # bound is used as before.
    narrow = min(truncate, NARROW_CELLS * float(gi["cell"]))
    if not resident:
        engine.set_source(np.ascontiguousarray(subsample()))    # dtype kept: fp64 georeferenced coordinates are shifted before they become fp32
    cloud_scores, scores = all_scores(narrow, sums_only=resident)
    ranked = sorted(scores)
    used = narrow
    if narrow < truncate and not (ranked[0] < DECISIVE * ranked[1]):
        if resident:
            engine.set_source(np.ascontiguousarray(subsample()))
        cloud_scores, scores = all_scores(truncate)
        used = truncate
        if resident:
            engine.set_source(comp)                             # as the caller left it
    best = int(np.argmin(scores))
    R = cands[best]
    # (a stage next to a half turn becomes two of half the angle: the visualiser's logm needs the principal branch)
    stages = transforms.split_half_turns([(R, cc)]) + [transforms.rt_to_matrix(np.eye(3), cg - cc)]
    return stages, transforms.compose(stages), {"scores": scores, "cloud_scores": cloud_scores,
                                                "trajectory_scores": traj_scores if use_traj else None,
                                                "chosen": best, "truncate_m": truncate, "score_bound_m": used}
