"""``align(gold_cloud, comparison_cloud) -> (R, t)`` — the call surface BASELINE.json names.

The reference exposes the path as ``Alignment(...).compare(data_folder)``
(``/root/reference/task_def.py:175-187``); ``align`` is its arithmetic core on two point
arrays, and :mod:`mapalign.alignment` wraps it back into the reference's class shape.

Contexts are cached per device: creating a context and growing its device buffers costs
~17 ms of ``hipMalloc`` the first time, an order of magnitude more than a 1M-point
alignment itself, so ``align`` keeps ONE :class:`MapAligner` per device for the life of the
process and a second call of similar size performs no device allocation at all
(``MapAligner.mem_info``; ``release_cached_engines`` frees them).  :class:`GoldMap` goes one
step further for the usual deployment — one gold map, many maintenance scans — and keeps the
indexed gold map resident between calls.
"""
from __future__ import annotations

import threading

import numpy as np

from .engine import MapAligner

DEFAULT_MAX_ITER = 50
DEFAULT_MAX_CORR_DIST = 0.10   # metres; SURVEY.md §8d "Parameters"
DEFAULT_TOL = 0.0

_engines = {}            # (device, slot) -> (MapAligner, lock)
_engines_guard = threading.Lock()


def cached_engine(device=0, slot=0):
    """The process-wide context of ``device`` and the lock that serialises its users
    (a context is not thread-safe; different devices run concurrently).  ``slot`` 1 is the
    second context :func:`align_many` stages the next pair in."""
    key = (int(device), int(slot))
    with _engines_guard:
        entry = _engines.get(key)
        if entry is None or entry[0]._ctx is None:
            entry = (MapAligner(key[0]), threading.Lock())
            _engines[key] = entry
        return entry


def release_cached_engines():
    """Destroy the cached contexts (frees their device memory)."""
    with _engines_guard:
        for eng, lock in _engines.values():
            with lock:
                eng.close()
        _engines.clear()


class GoldMap:
    """A gold ("base") map uploaded and indexed once, aligned against many times.

    ``align(GoldMap(gold), scan)`` skips the upload and the grid build of the gold map
    (1.2 ms for 1M points, plus its upload) on every call after the first.  Owns its own context."""

    def __init__(self, gold_cloud, cell=0.0, device=0):
        self.engine = MapAligner(device)
        self.lock = threading.Lock()
        self.grid = self.engine.set_gold(gold_cloud, cell)
        self.device = int(device)

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def align(gold_cloud, comparison_cloud, *, max_iter=DEFAULT_MAX_ITER, tol=DEFAULT_TOL,
          max_corr_dist=DEFAULT_MAX_CORR_DIST, init=None, weights=None, cell=0.0, device=0,
          return_info=False):
    """Rigidly register ``comparison_cloud`` onto ``gold_cloud`` on one MI355X.

    Returns ``(R [3,3] f64, t [3] f64)`` with ``gold ~= R @ p + t`` for comparison points
    ``p`` (plus an info dict when ``return_info``).  Point-to-point ICP: per iteration an
    exact nearest-neighbour correspondence search gated at ``max_corr_dist``, fp64
    centroid / cross-covariance accumulation and a 3x3 SVD rigid solve, all in HIP kernels.
    ``gold_cloud`` is an ``[M,3]`` array or a :class:`GoldMap`.

    Raises ``mapalign.MapAlignError`` for empty / non-finite clouds, when no correspondence
    falls inside the gate (or they are collinear), and when no MI355X (or the built library)
    is available — there is no CPU fallback.
    """
    if isinstance(gold_cloud, GoldMap):
        eng, lock, grid = gold_cloud.engine, gold_cloud.lock, gold_cloud.grid
        gold_cloud = None
    else:
        eng, lock = cached_engine(device)
        grid = None
    with lock:
        if gold_cloud is not None:
            grid = eng.set_gold(gold_cloud, cell)
        eng.set_source(comparison_cloud, weights)
        res = eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, init=init,
                      with_trace=return_info)
        mem = eng.mem_info() if return_info else None
    if return_info:
        info = {"iterations": res.iterations, "rmse": res.rmse, "inliers": res.inliers,
                "converged": res.converged, "gpu_ms": res.gpu_ms, "grid": grid, "trace": res.trace,
                "T": res.T, "mem": mem}
        return res.R, res.t, info
    return res.R, res.t


def align_many(pairs, *, max_iter=DEFAULT_MAX_ITER, tol=DEFAULT_TOL, max_corr_dist=DEFAULT_MAX_CORR_DIST,
               cell=0.0, device=0, pipeline=True):
    """Align a sequence of ``(gold_cloud, comparison_cloud)`` pairs on one GPU -> list of 4x4.

    With ``pipeline`` the upload, grid build and scan sort of pair k+1 (1.7 ms for 1M-point clouds,
    half of it host round trips and PCIe) run in a second context on a helper thread while pair
    k iterates (2.4 ms): the GPU's queue never drains between pairs, and a stream of pairs
    costs about the loop alone per pair instead of loop + setup.  Results are exactly those of
    ``align`` pair by pair (each pair is staged and iterated in one context, untouched by the other).
    """
    from concurrent.futures import ThreadPoolExecutor

    def stage(slot, pair):
        eng, lock = cached_engine(device, slot)
        lock.acquire()
        try:
            eng.set_gold(pair[0], cell)
            eng.set_source(pair[1])
        except BaseException:
            lock.release()
            raise
        return eng, lock

    out = []
    it = iter(pairs)
    first = next(it, None)
    if first is None:
        return out
    if not pipeline:
        for pair in [first, *it]:
            eng, lock = stage(0, pair)
            try:
                out.append(eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist).T)
            finally:
                lock.release()
        return out
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut, k = pool.submit(stage, 0, first), 0
        while fut is not None:
            eng, lock = fut.result()
            nxt = next(it, None)
            fut = pool.submit(stage, (k + 1) % 2, nxt) if nxt is not None else None
            try:
                out.append(eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist).T)
            except BaseException:
                if fut is not None:                      # do not leave the staged context locked
                    try:
                        fut.result()[1].release()
                    except Exception:
                        pass
                raise
            finally:
                lock.release()
            k += 1
    return out


def compose(T_b, T_a):
    """4x4 of 'apply T_a, then T_b'."""
    return np.asarray(T_b, dtype=np.float64) @ np.asarray(T_a, dtype=np.float64)
