"""``align(gold_cloud, comparison_cloud) -> (R, t)`` — the call surface BASELINE.json names.

The reference exposes the path as ``Alignment(...).compare(data_folder)``
(``/root/reference/task_def.py:175-187``); ``align`` is its arithmetic core on two point
arrays, and :mod:`mapalign.alignment` wraps it back into the reference's class shape.
"""
from __future__ import annotations

import numpy as np

from .engine import MapAligner

DEFAULT_MAX_ITER = 50
DEFAULT_MAX_CORR_DIST = 0.10   # metres; SURVEY.md §8d "Parameters"
DEFAULT_TOL = 0.0


def align(gold_cloud, comparison_cloud, *, max_iter=DEFAULT_MAX_ITER, tol=DEFAULT_TOL,
          max_corr_dist=DEFAULT_MAX_CORR_DIST, init=None, weights=None, cell=0.0, device=0,
          return_info=False):
    """Rigidly register ``comparison_cloud`` onto ``gold_cloud`` on one MI355X.

    Returns ``(R [3,3] f64, t [3] f64)`` with ``gold ~= R @ p + t`` for comparison points
    ``p`` (plus an info dict when ``return_info``).  Point-to-point ICP: per iteration an
    exact nearest-neighbour correspondence search gated at ``max_corr_dist``, fp64
    centroid / cross-covariance accumulation and a 3x3 SVD rigid solve, all in HIP kernels.

    Raises ``mapalign.MapAlignError`` for empty / non-finite clouds, when fewer than three
    correspondences fall inside the gate, and when no MI355X (or the built library) is
    available — there is no CPU fallback.
    """
    with MapAligner(device) as eng:
        grid = eng.set_gold(gold_cloud, cell)
        eng.set_source(comparison_cloud, weights)
        res = eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, init=init,
                      with_trace=return_info)
    if return_info:
        info = {"iterations": res.iterations, "rmse": res.rmse, "inliers": res.inliers,
                "converged": res.converged, "gpu_ms": res.gpu_ms, "grid": grid, "trace": res.trace,
                "T": res.T}
        return res.R, res.t, info
    return res.R, res.t


def compose(T_b, T_a):
    """4x4 of 'apply T_a, then T_b'."""
    return np.asarray(T_b, dtype=np.float64) @ np.asarray(T_a, dtype=np.float64)
