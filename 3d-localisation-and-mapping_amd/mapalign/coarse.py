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

The reference's own coarse stage (inside the absent align.py) is unknown; only its
output format is pinned.  This is host-side glue (3x3 eigen-decompositions); all cloud
work goes through the HIP engine.
"""
from __future__ import annotations

import numpy as np

from . import transforms


def principal_frame(points):
    """(centroid, axes [3,3] columns by decreasing variance, right-handed, variances)."""
    p = np.asarray(points, dtype=np.float64)
    c = p.mean(axis=0)
    w, v = np.linalg.eigh(np.cov((p - c).T))
    order = np.argsort(w)[::-1]
    axes = v[:, order]
    if np.linalg.det(axes) < 0:
        axes[:, 2] = -axes[:, 2]
    return c, axes, w[order]


def candidate_rotations(axes_gold, axes_comp):
    """The four proper rotations taking the comparison principal frame onto the gold one."""
    out = []
    for sx, sy in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        S = np.diag([sx, sy, sx * sy])          # det +1
        out.append(axes_gold @ S @ axes_comp.T)
    return out


def coarse_align(engine, gold_cloud, comparison_cloud, sample=20000, seed=0):
    """Returns (stages, T_coarse, info); ``engine`` is a MapAligner with the gold map set.

    ``stages`` = [(R, centre), T_translate] in the reference's mixed format.
    """
    gold = np.asarray(gold_cloud, dtype=np.float64)
    comp = np.asarray(comparison_cloud, dtype=np.float64)
    cg, ag, _ = principal_frame(gold)
    cc, ac, _ = principal_frame(comp)
    rng = np.random.default_rng(seed)
    sub = comp if len(comp) <= sample else comp[rng.choice(len(comp), sample, replace=False)]
    engine.set_source(sub.astype(np.float32))
    scores = []
    cands = candidate_rotations(ag, ac)
    for R in cands:
        T = transforms.rt_to_matrix(np.eye(3), cg - cc) @ transforms.rotation_about_centre(R, cc)
        acc, _ = engine.icp_step(T, max_corr_dist=np.inf)
        scores.append(float(np.sqrt(acc[16] / max(acc[0], 1.0))))
    best = int(np.argmin(scores))
    R = cands[best]
    stages = [(R, cc), transforms.rt_to_matrix(np.eye(3), cg - cc)]
    return stages, transforms.compose(stages), {"scores": scores, "chosen": best}
