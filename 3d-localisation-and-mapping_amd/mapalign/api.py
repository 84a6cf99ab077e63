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

from ._lib import MA_ERR_DEGENERATE, MapAlignError
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
          return_info=False, gate_schedule=None):
    """Rigidly register ``comparison_cloud`` onto ``gold_cloud`` on one MI355X.

    Returns ``(R [3,3] f64, t [3] f64)`` with ``gold ~= R @ p + t`` for comparison points
    ``p`` (plus an info dict when ``return_info``).  Point-to-point ICP: per iteration an
    exact nearest-neighbour correspondence search gated at ``max_corr_dist``, fp64
    centroid / cross-covariance accumulation and a 3x3 SVD rigid solve, all in HIP kernels.
    ``gold_cloud`` is an ``[M,3]`` array or a :class:`GoldMap`.

    ``gate_schedule``: optional ``[(gate_m, max_iterations), ...]`` run before the final stage, widest
    gate first (each stage stops when its rmse settles) — for starts further off than the final
    gate can recover (0.10 m alone: about 2 degrees / 15 cm on the benchmark pair; 0.8 -> 0.4 -> 0.2 m
    first: 6 degrees / 1 m).  The default, ``None``, is the plain loop the benchmark measures.

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
            grid = eng.set_pair(gold_cloud, comparison_cloud, weights, cell)    # the scan crosses PCIe while the gold map is indexed
        else:
            eng.set_source(comparison_cloud, weights)
        widened = []
        for gate, iters in (gate_schedule or ()):
            try:
                r = eng.icp(max_iter=int(iters), tol=1e-5, max_corr_dist=float(gate), init=init)
            except MapAlignError as e:
                if e.status != MA_ERR_DEGENERATE:    # a device fault or a call-order error is not "nothing inside this gate"
                    raise
                continue                             # nothing inside this gate: keep the start, try the next one
            init = r.T
            widened.append((float(gate), r.iterations, r.rmse))
        res = eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, init=init,
                      with_trace=return_info)
        mem = eng.mem_info() if return_info else None
        lists = eng.list_stats() if return_info else None     # launches that wrote / read the candidate lists
        box = eng.grid_box() if return_info else None         # which box the gold map's grid covers (stray points: DESIGN.md 3)
    if return_info:
        info = {"iterations": res.iterations, "rmse": res.rmse, "inliers": res.inliers,
                "converged": res.converged, "gpu_ms": res.gpu_ms, "grid": grid, "grid_box": box, "trace": res.trace,
                "T": res.T, "mem": mem, "gate_schedule": widened, "list_launches": lists}
        return res.R, res.t, info
    return res.R, res.t


def align_many(pairs, *, max_iter=DEFAULT_MAX_ITER, tol=DEFAULT_TOL, max_corr_dist=DEFAULT_MAX_CORR_DIST,
               cell=0.0, device=0, workers=3, pipeline=True):
    """Align a sequence of ``(gold_cloud, comparison_cloud)`` pairs on one GPU -> list of 4x4, in order.

    ``workers`` host threads each own one context (= one HIP stream) and take the next pair off the
    sequence as they finish: upload, grid build and scan sort of one pair overlap the iterations of
    the others, and the iterations of several pairs overlap each other.  One alignment leaves the GPU
    a third idle (its early launches are 57-66 % occupied and 8.5 us pass between two launches), so
    alignments running side by side raise the throughput: resident 1M-point pairs 20.8 k ICP
    iterations/s alone, 28.9 k with two and 33.4 k with four at once; from host arrays, upload and
    grid build included, 4.05 ms per pair one by one against ~2 ms with three workers
    (``tools/many_pairs.py``, ``tools/concurrent_pairs.py``).  Every pair is staged and iterated inside one
    context, so each result is exactly that of ``align`` on that pair.  ``pipeline=False`` (or
    ``workers=1``) runs the pairs one after the other.  Consecutive pairs that share their gold
    array OBJECT (``[(gold, s) for s in scans]``) re-use the indexed gold map of the worker's context
    (do not modify such an array in place while the call runs).
    """
    seq = iter(pairs)
    workers = 1 if not pipeline else max(1, int(workers))
    results, errors = {}, {}
    guard = threading.Lock()
    counter = [0]

    def take():
        with guard:
            if errors:
                return None
            pair = next(seq, None)
            if pair is None:
                return None
            i = counter[0]
            counter[0] += 1
            return i, pair

    def run(slot):
        eng, lock = cached_engine(device, slot)
        # Which gold array the context holds is recorded ON THE ENGINE (eng.gold_token, cleared by every
        # set_gold / set_pair / map_finish(as_gold), read and written under the engine's lock): the cached
        # contexts are shared with align() and with other align_many() calls, which may install another
        # gold map between two of this worker's jobs.
        while True:
            job = take()
            if job is None:
                return
            i, (g, c) = job
            try:
                if isinstance(g, GoldMap):        # a resident gold map brings its own context (its lock serialises its users)
                    with g.lock:
                        g.engine.set_source(c)
                        results[i] = g.engine.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist).T
                    continue
                with lock:
                    if eng.gold_token is not g:   # one gold map, many scans: every worker indexes it once
                        eng.set_pair(g, c, None, cell)
                        eng.gold_token = g        # (kept alive by the engine, compared by identity)
                    else:
                        eng.set_source(c)
                    results[i] = eng.icp(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist).T
            except BaseException as e:      # reported below, for the first failing pair
                with guard:
                    errors[i] = e
                return

    if workers == 1:
        run(0)
    else:
        threads = [threading.Thread(target=run, args=(slot,), name=f"mapalign-align-many-{slot}") for slot in range(workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[min(errors)]
    return [results[i] for i in range(counter[0])]


def compose(T_b, T_a):
    """4x4 of 'apply T_a, then T_b'."""
    return np.asarray(T_b, dtype=np.float64) @ np.asarray(T_a, dtype=np.float64)
