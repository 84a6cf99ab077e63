"""Multi-GPU scheduling of the alignment path: one process per GPU.

Two modes, both without any per-point traffic between GPUs:

* **pair-per-GPU** (:func:`align_pairs`) — scan pairs are independent; rank r runs the
  whole ICP loop for pairs r, r+W, ... with zero communication inside the loop and ONE
  tensor all-gather of the 4x4 results at the end.  This is the node-level (weak-scaling)
  mode and what ``bench.py --gpus N`` measures.
* **sharded source** (:func:`sharded_align`) — one pair spread over the ranks of a group:
  the gold map and its grid are replicated, the comparison cloud is block-partitioned, and
  per iteration the ONLY exchange is one all-reduce(sum) of the 17 centroid /
  cross-covariance sums (136 B, latency-bound) — 64-bit fixed-point INTEGERS, so the
  result does not depend on the order of the additions, nor on the partition: any number
  of ranks returns the single GPU's bits (``MapAligner.set_sum_bounds``: the ranks agree
  once per cloud on what bounds a term).  The collective lives INSIDE
  ``libmapalign.so`` (``ma_icp_sharded``: search -> 19-way sum -> ``ncclAllReduce`` ->
  solve, enqueued from C on the context's stream, no Python between iterations); Python
  only bootstraps the ``ncclUniqueId`` (:func:`bootstrap_comm`) over whatever
  ``torch.distributed`` group the job already has.

``sharded_icp`` is the same loop spelled out in Python over a duck-typed engine.  It exists
for the CPU tests (world-size-2 ``gloo`` with an oracle-backed stand-in engine) and for
callers that want their own collective between :meth:`MapAligner.step_accumulate` and
:meth:`MapAligner.step_solve`; the product path on GPUs is ``ma_icp_sharded``.
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


def _world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_ranks_ready(ok, group=None, device=None):
    """all-reduce(MIN) of a readiness flag over the torch group (any backend): True iff every rank passed True."""
    import torch
    import torch.distributed as dist

    world, _ = _world(group)
    if world == 1:
        return bool(ok)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = (f"cuda:{torch.cuda.current_device() if device is None else device}") if on_gpu else "cpu"
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return float(flag.item()) >= 1.0


def bootstrap_comm(aligner, group=None):
    """Give ``aligner`` (a :class:`mapalign.MapAligner`) an RCCL communicator spanning the
    ranks of ``group``: rank 0 draws the ``ncclUniqueId`` in the library, the 128 bytes travel
    through ``torch.distributed`` (any backend), every rank joins with ``ma_comm_init``.
    Without an initialised process group this makes a 1-rank communicator."""
    import torch.distributed as dist

    world, rank = _world(group)
    payload = [None]
    if rank == 0:
        try:
            payload[0] = aligner.comm_unique_id()
        except Exception as e:          # the other ranks are about to enter the broadcast: never leave them hanging
            payload[0] = f"{type(e).__name__}: {e}"       # (a plain string: survives pickling)
    if world > 1:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(payload, src=src, group=group)
    if not isinstance(payload[0], (bytes, bytearray)):
        raise RuntimeError(f"rank 0 could not create the RCCL bootstrap id: {payload[0]}")
    aligner.comm_init(payload[0], world, rank)
    return world, rank


def bootstrap_p2p(aligner, group=None):
    """Set up the peer-to-peer exchange of ``aligner`` over the ranks of ``group``: every rank exports its rows, the blobs
    travel through ``torch.distributed`` (any backend; the all-gather is also the barrier after which rows may be
    written), every rank maps all of them.  Without a process group: a 1-rank exchange."""
    import torch.distributed as dist

    world, rank = _world(group)
    mine = None
    try:
        mine = aligner.p2p_export()
    except Exception as e:              # the other ranks are about to enter the all-gather: never leave them hanging
        mine = f"{type(e).__name__}: {e}"
    blobs = [mine]
    if world > 1:
        blobs = [None] * world
        dist.all_gather_object(blobs, mine, group=group)
    bad = [f"rank {r}: {b}" for r, b in enumerate(blobs) if not isinstance(b, (bytes, bytearray))]
    if bad:
        raise RuntimeError("p2p bootstrap failed — " + "; ".join(bad))
    aligner.p2p_init(blobs, rank)
    return world, rank


class HipShardEngine:
    """Python-stepping view of one :class:`mapalign.MapAligner` (see ``sharded_icp``)."""

    def __init__(self, aligner, max_corr_dist=0.10, tol=0.0):
        self.aligner = aligner
        self.max_corr_dist = float(max_corr_dist)
        self.tol = float(tol)

    def new_acc(self):
        """The exchange buffer: 17 fixed-point sums, int64 (summing them is exact in any order)."""
        import torch
        return torch.zeros(ACC_LEN, dtype=torch.int64, device=f"cuda:{self.aligner.device}")

    def agree_sum_bounds(self, group=None):
        """Collective: every rank's (n, wmax, rs) -> the whole cloud's, installed on this rank's context, so that all
        ranks derive the same fixed-point scales (and the single GPU's).  Once per source cloud."""
        import torch
        import torch.distributed as dist

        mine = np.asarray(self.aligner.sum_bounds(), dtype=np.float64)
        world, _ = _world(group)
        if world > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.as_tensor(mine, device=f"cuda:{self.aligner.device}" if on_gpu else "cpu")
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t, group=group)
            mine = self.aligner.combine_sum_bounds(torch.stack(out).cpu().numpy())
        self.aligner.set_sum_bounds(mine)
        return mine

    def bind_current_stream(self):
        """Launch on torch's current stream so kernels and a torch collective are ordered without host syncs."""
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
    """The sharded-source loop in Python over a duck-typed engine (tests / custom collectives).
    ``acc`` is the exchange buffer of 17 sums (int64 fixed point for the HIP engine); ``all_reduce(acc)`` defaults to
    ``torch.distributed.all_reduce(acc, SUM, group)``.  Returns ``(T 4x4, info dict)``;
    identical on every rank of the group."""
    if all_reduce is None:
        import torch.distributed as dist

        def all_reduce(t):
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    if hasattr(engine, "agree_sum_bounds"):
        engine.agree_sum_bounds(group)
    engine.state_set(init)
    for _ in range(int(max_iter)):
        engine.accumulate_into(acc)   # a4 + a5 on the local shard
        all_reduce(acc)               # the one collective: 17 sums
        engine.solve_from(acc)        # a6 + a7, redundantly on every rank
    return engine.state_get()


def sharded_align(gold_cloud, comparison_cloud, *, max_iter=50, tol=0.0, max_corr_dist=0.10, init=None,
                  cell=0.0, device=None, group=None, return_info=False, transport="rccl"):
    """One pair over all ranks of ``group`` (every rank passes the same full clouds).  The
    loop, including the exchange of the 17 sums, runs inside the library (``ma_icp_sharded``):
    ``transport="rccl"`` an ``ncclAllReduce`` per iteration, ``"p2p"`` the peer-to-peer exchange
    inside the iteration's second kernel (no library collective).  Same bits either way."""
    if transport not in ("rccl", "p2p"):
        raise ValueError("transport must be 'rccl' or 'p2p'")
    import torch

    from .engine import MapAligner

    world, rank = _world(group)
    device = torch.cuda.current_device() if device is None else device
    comp = np.asarray(comparison_cloud)
    if len(comp) < world:            # the same on every rank (all pass the same cloud): nobody enters a collective
        raise ValueError(f"more ranks ({world}) than comparison points ({len(comp)})")
    lo, hi = shard_bounds(len(comp), world, rank)
    with MapAligner(device) as eng:
        # Rank-local staging can fail on ONE rank (its shard, its device, its librccl).  The ranks agree that
        # everybody is ready BEFORE anyone enters the communicator bootstrap or the loop's all-reduce, where a
        # missing rank would leave the others waiting for ever.
        err = None
        try:
            eng.set_gold(gold_cloud, cell)
            eng.set_source(comp[lo:hi])
            if rank != 0 and transport == "rccl":
                eng.comm_unique_id()     # can this rank load librccl at all?  (rank 0 finds out in bootstrap_comm)
        except Exception as e:           # noqa: BLE001 - reported on every rank below
            err = e
        if not all_ranks_ready(err is None, group, device):
            raise RuntimeError(f"sharded_align: staging failed on rank {rank}: {err}" if err is not None else
                               "sharded_align: staging failed on another rank of the group")
        if transport == "p2p":
            bootstrap_p2p(eng, group)
        else:
            bootstrap_comm(eng, group)
        # (an error returned from ma_icp_sharded mid-loop leaves the communicator unusable: its peers may still
        # sit in an all-reduce; tear the process group down rather than retrying on it.  The p2p exchange times out
        # instead — MA_ERR_COMM on every rank — and has to be set up again.)
        res = eng.icp_sharded(max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, init=init)
        if transport == "p2p":
            if world > 1:               # nobody unmaps rows a peer may still be writing its last iteration into
                import torch.distributed as dist
                dist.barrier(group=group)
            eng.p2p_destroy()
        else:
            eng.comm_destroy()
    if return_info:
        return res.R, res.t, {"iterations": res.iterations, "rmse": res.rmse, "inliers": res.inliers,
                              "converged": res.converged, "gpu_ms": res.gpu_ms, "ranks": world}
    return res.R, res.t


def _gather_rows(rows, n_cols, world, group, device):
    """All-gather of a [k, n_cols] float64 block per rank (k <= k_max, padded) as ONE tensor collective."""
    import torch
    import torch.distributed as dist

    k_max = max(1, -(-int(rows[1]) // world))
    block = torch.full((k_max, n_cols), float("nan"), dtype=torch.float64, device=device)
    if len(rows[0]):
        block[:len(rows[0])] = torch.as_tensor(np.asarray(rows[0], dtype=np.float64), device=device)
    out = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(out, block, group=group)
    return torch.cat(out).cpu().numpy()


def align_pairs(pairs, *, max_iter=50, tol=0.0, max_corr_dist=0.10, cell=0.0, device=None, group=None,
                align_fn=None):
    """Pair-per-GPU: ``pairs`` is the same list of (gold, comparison) on every rank (or a
    callable ``i -> (gold, comparison)`` with ``len`` given by ``pairs.n``).  Rank r aligns
    pairs r, r+W, ...; the 4x4 results travel in one tensor all-gather (17 doubles per pair:
    the pair index and the matrix) so every rank returns the full list."""
    import torch.distributed as dist

    world, rank = _world(group)
    n_pairs = pairs.n if hasattr(pairs, "n") else len(pairs)
    gather_device = "cpu"
    default_align_fn = None
    if align_fn is None:
        import torch

        from .api import align
        dev = torch.cuda.current_device() if device is None else device
        if world > 1 and dist.get_backend(group) == "nccl":
            gather_device = f"cuda:{dev}"

        def align_fn(g, c):
            R, t = align(g, c, max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, cell=cell, device=dev)
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R, t
            return T

        default_align_fn = align_fn

    mine = []
    my_ids = pairs_for_rank(n_pairs, world, rank)
    get = (lambda i: pairs(i)) if callable(pairs) else (lambda i: pairs[i])
    if align_fn is default_align_fn:
        # this rank's pairs as one pipelined stream: the next pair is uploaded and indexed while the current one iterates
        from .api import align_many
        Ts = align_many((get(i) for i in my_ids), max_iter=max_iter, tol=tol, max_corr_dist=max_corr_dist, cell=cell,
                        device=dev)
        mine = [np.concatenate([[float(i)], np.asarray(T, dtype=np.float64).reshape(16)]) for i, T in zip(my_ids, Ts)]
    else:
        for i in my_ids:
            g, c = get(i)
            mine.append(np.concatenate([[float(i)], np.asarray(align_fn(g, c), dtype=np.float64).reshape(16)]))
    if world == 1:
        table = np.asarray(mine, dtype=np.float64).reshape(-1, 17)
    else:
        table = _gather_rows((mine, n_pairs), 17, world, group, gather_device)
    out = [None] * n_pairs
    for row in table:
        if np.isfinite(row[0]):
            out[int(row[0])] = row[1:].reshape(4, 4).copy()
    return out
