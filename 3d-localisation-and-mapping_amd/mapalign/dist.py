"""Multi-GPU scheduling of the alignment path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

Two modes, both without any per-point traffic between GPUs:

* **pair-per-GPU** (:func:`align_pairs`) — scan pairs are independent; rank r runs the
  whole ICP loop for pairs r, r+W, ... with zero communication inside the loop and one
  small gather of the 4x4 results at the end.  This is the node-level (weak-scaling) mode.
* **sharded source** (:func:`sharded_icp`) — one pair spread over the ranks of a group:
  the gold map and its grid are replicated, the comparison cloud is block-partitioned,
  and per iteration the ONLY exchange is one all-reduce(sum) of the 17-double
  centroid / cross-covariance accumulator (136 B, latency-bound; RCCL's LL protocol),
  after which every rank runs the identical 3x3 solve — no broadcast needed.

The engines are duck-typed so the CPU tests can drive the same scheduling code with a
stand-in engine over ``gloo``; the product engine is :class:`HipShardEngine`.
"""
from __future__ import annotations

import numpy as np

ACC_LEN = 17


def shard_bounds(n, world, rank):
    """Contiguous block partition of n items: rank -> [lo, hi)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pairs_for_rank(n_pairs, world, rank):
    """Round-robin pair schedule of the pair-per-GPU mode."""
    return list(range(rank, int(n_pairs), int(world)))


class HipShardEngine:
    """Sharded-source engine over one :class:`mapalign.MapAligner` (one GPU)."""

    def __init__(self, aligner, max_corr_dist=0.10, tol=0.0):
        self.aligner = aligner
        self.max_corr_dist = float(max_corr_dist)
        self.tol = float(tol)

    def new_acc(self):
        import torch
        return torch.zeros(ACC_LEN, dtype=torch.float64, device=f"cuda:{self.aligner.device}")

    def bind_current_stream(self):
        """Launch on torch's current stream so kernels and the RCCL all-reduce are ordered without host syncs."""
        import torch
        self.aligner.set_stream(torch.cuda.current_stream(self.aligner.device).cuda_stream)

    def state_set(self, T=None):
        self.aligner.state_set(T)

    def accumulate_into(self, acc):
        self.aligner.step_accumulate(acc.data_ptr(), self.max_corr_dist)

    def solve_from(self, acc):
        self.aligner.step_solve(acc.data_ptr(), self.tol, self.max_corr_dist)

    def state_get(self):
        return self.aligner.state_get()


def sharded_icp(engine, acc, max_iter=50, init=None, group=None, all_reduce=None):
    """Run the sharded-source loop on this rank.  ``acc`` is the 17-double exchange buffer
    (a tensor on the engine's device); ``all_reduce(acc)`` defaults to
    ``torch.distributed.all_reduce(acc, SUM, group)``.  Returns ``(T 4x4, info dict)``;
    identical on every rank of the group."""
    if all_reduce is None:
        import torch.distributed as dist

        def all_reduce(t):
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    engine.state_set(init)
    for _ in range(int(max_iter)):
        engine.accumulate_into(acc)   # a4 + a5 on the local shard
        all_reduce(acc)               # the one collective: 17 doubles
        engine.solve_from(acc)        # a6 + a7, redundantly on every rank
    return engine.state_get()


def sharded_align(gold_cloud, comparison_cloud, *, max_iter=50, tol=0.0, max_corr_dist=0.10, init=None,
                  cell=0.0, device=None, group=None):
    """One pair over all ranks of ``group`` (every rank passes the same full clouds)."""
    import torch
    import torch.distributed as dist

    from .engine import MapAligner

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = torch.cuda.current_device() if device is None else device
    comp = np.asarray(comparison_cloud)
    lo, hi = shard_bounds(len(comp), world, rank)
    if hi <= lo:
        raise ValueError("more ranks than comparison points")
    with MapAligner(device) as eng:
        eng.set_gold(gold_cloud, cell)
        eng.set_source(comp[lo:hi])
        he = HipShardEngine(eng, max_corr_dist, tol)
        he.bind_current_stream()
        acc = he.new_acc()
        T, info = sharded_icp(he, acc, max_iter=max_iter, init=init, group=group)
        eng.use_own_stream()
    return T[:3, :3].copy(), T[:3, 3].copy(), info


def align_pairs(pairs, *, max_iter=50, tol=0.0, max_corr_dist=0.10, cell=0.0, device=None, group=None,
                align_fn=None):
    """Pair-per-GPU: ``pairs`` is the same list of (gold, comparison) on every rank (or a
    callable ``i -> (gold, comparison)`` with ``len`` given by ``pairs.n``).  Rank r aligns
    pairs r, r+W, ...; the 4x4 results are gathered so every rank returns the full list."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_pairs = pairs.n if hasattr(pairs, "n") else len(pairs)
    if align_fn is None:
        import torch

        from .api import align
        dev = torch.cuda.current_device() if device is None else device

        def align_fn(g, c):
            R, t = align(g, c, max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, cell=cell, device=dev)
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R, t
            return T

    mine = {}
    for i in pairs_for_rank(n_pairs, world, rank):
        g, c = pairs(i) if callable(pairs) else pairs[i]
        mine[i] = align_fn(g, c)
    if world == 1:
        return [mine[i] for i in range(n_pairs)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    merged = {}
    for d in gathered:
        merged.update(d)
    return [merged[i] for i in range(n_pairs)]
