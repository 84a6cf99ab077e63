"""world_size-2 `gloo` tests of the multi-GPU scheduling code (mapalign/dist.py) on CPU.

The product engine needs a GPU, so these drive the SAME scheduling functions with a
stand-in engine built on the oracle (tests may use the oracle; the product may not).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


@pytest.fixture(scope="module")
def shim_built():
    """tests/_build/list_planner_shim.so (csrc/list_planner.h behind a C interface), built by tests/test_list_planner.py's fixture code."""
    import test_list_planner as tlp
    return tlp.build_shim()


class OracleShardEngine:
    """CPU stand-in with the HipShardEngine interface."""

    def __init__(self, gold, comp_shard, gate):
        from oracle import icp_oracle as orc
        self.orc = orc
        self.index = orc.GoldIndex(gold, n_jobs=1)
        self.comp = np.asarray(comp_shard, dtype=np.float64)
        self.gate = gate
        self.iters = 0

    def state_set(self, T=None):
        self.R, self.t = self.orc.as_matrix(T)
        self.iters = 0

    def accumulate_into(self, acc):
        p_t = self.comp @ self.R.T + self.t
        idx, d2 = self.index.query(p_t)
        w = (d2 <= self.gate ** 2).astype(np.float64)
        acc.copy_(torch.from_numpy(self.orc.accumulate(p_t, self.index.gold[idx], w, d2)))

    def solve_from(self, acc):
        dR, dt, ok = self.orc.kabsch_from_acc(acc.numpy())
        assert ok
        self.R, self.t = dR @ self.R, dR @ self.t + dt
        self.iters += 1

    def state_get(self):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = self.R, self.t
        return T, {"iterations": self.iters}


def _worker(rank, world, port, out_dir):
    for p in (PKG, ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapalign import dist as mdist
    from mapalign import synthetic as syn

    # --- sharded source: all-reduce of the 17-double accumulator only
    gold, comp, _, _ = syn.make_pair(21, 3000, 2501, resample=False if False else True)
    lo, hi = mdist.shard_bounds(len(comp), world, rank)
    eng = OracleShardEngine(gold, comp[lo:hi], 0.4)
    acc = torch.zeros(mdist.ACC_LEN, dtype=torch.float64)
    T, info = mdist.sharded_icp(eng, acc, max_iter=6)
    # --- pair-per-GPU: no collective in the loop, one gather at the end
    calls = []

    def fake_align(g, c):
        calls.append(1)
        T = np.eye(4)
        T[0, 3] = float(len(g) + len(c))
        return T

    pairs = [(np.zeros((10 + i, 3)), np.zeros((5, 3))) for i in range(5)]
    res = mdist.align_pairs(pairs, align_fn=fake_align)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), T=T, iters=info["iterations"], pair_T=np.stack(res),
             n_calls=len(calls))
    dist.destroy_process_group()


class _FakeAligner:
    """Stands in for MapAligner in the RCCL bootstrap (no GPU here): records what it is told."""

    def __init__(self, fail):
        self.fail, self.joined = fail, None

    def comm_unique_id(self):
        if self.fail:
            raise RuntimeError("librccl could not be loaded")
        return bytes(range(128))

    def comm_init(self, uid, n_ranks, rank):
        self.joined = (bytes(uid), n_ranks, rank)


def _bootstrap_worker(rank, world, port, out_dir):
    for p in (PKG, ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapalign import dist as mdist

    ok = _FakeAligner(fail=False)
    w, r = mdist.bootstrap_comm(ok)
    bad = _FakeAligner(fail=True)
    try:
        mdist.bootstrap_comm(bad)
        raised = ""
    except RuntimeError as e:          # EVERY rank must get here: nobody may be left waiting in the broadcast
        raised = str(e)
    np.savez(os.path.join(out_dir, f"boot{rank}.npz"), w=w, r=r, uid=np.frombuffer(ok.joined[0], np.uint8),
             n=ok.joined[1], k=ok.joined[2], raised=raised, bad_joined=bad.joined is not None)
    dist.destroy_process_group()


def test_world2_rccl_bootstrap_id_travels_and_failures_reach_every_rank(tmp_path):
    world = 2
    mp.spawn(_bootstrap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        z = np.load(tmp_path / f"boot{rank}.npz")
        assert (int(z["w"]), int(z["r"]), int(z["n"]), int(z["k"])) == (2, rank, 2, rank)
        assert np.array_equal(z["uid"], np.arange(128, dtype=np.uint8))      # rank 0's id reached this rank
        assert "librccl could not be loaded" in str(z["raised"]) and not bool(z["bad_joined"])


def _stop_worker(rank, world, port, out_dir):
    """Two hosts of a sharded loop with a convergence test: each decides from ITS OWN looks at the pinned words where to
    stop enqueueing, and issues one all-reduce per iteration it enqueues — as ma_icp_sharded does.  Unequal counts would
    leave one rank waiting in a collective (the process group's timeout turns that into a failure here)."""
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import ctypes as C
    import test_list_planner as tlp
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "list_planner_shim.so"))
    for f in (lib.slot_wait_state, lib.stops_before_batch, lib.iterations_enqueued):
        f.restype = C.c_int
    rng = np.random.default_rng(100 + rank)              # every rank looks at different moments
    counts = []
    for max_iter, stop in ((40, 9), (40, 10), (40, 11), (40, 12), (40, 31), (40, None), (17, 3), (60, 58)):
        n = tlp.simulate_host(lib, max_iter, stop, rng)
        acc = torch.ones(17, dtype=torch.float64)
        for _ in range(n):                               # the collective of every enqueued iteration
            dist.all_reduce(acc)
            acc /= world
        counts.append(n)
    np.savez(os.path.join(out_dir, f"stop{rank}.npz"), counts=np.array(counts))
    dist.destroy_process_group()


def test_world2_ranks_of_a_stopped_sharded_loop_issue_the_same_collectives(tmp_path, shim_built):
    world = 2
    mp.spawn(_stop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c0, c1 = (np.load(tmp_path / f"stop{r}.npz")["counts"] for r in range(world))
    assert np.array_equal(c0, c1), (c0, c1)
    assert list(c0) == [16, 16, 16, 20, 36, 40, 8, 60]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_everything():
    from mapalign import dist as mdist
    for n in (1, 7, 100, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [mdist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert mdist.pairs_for_rank(8, 8, 3) == [3] and mdist.pairs_for_rank(5, 2, 1) == [1, 3]


def test_world2_sharded_and_pair_modes(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank ends with the identical transform ...
    assert np.array_equal(r0["T"], r1["T"]) and int(r0["iters"]) == 6
    # ... equal to the single-process oracle on the whole cloud (sum order differs: 1e-10)
    from mapalign import synthetic as syn
    from oracle import icp_oracle as orc
    gold, comp, _, _ = syn.make_pair(21, 3000, 2501)
    ref = orc.icp_oracle(gold, comp, max_iter=6, max_corr_dist=0.4, n_jobs=1)
    assert np.allclose(r0["T"][:3, :3], ref["R"], atol=1e-10) and np.allclose(r0["T"][:3, 3], ref["t"], atol=1e-10)
    # pair mode: work split 3 + 2, results complete and ordered on both ranks
    assert int(r0["n_calls"]) == 3 and int(r1["n_calls"]) == 2
    assert np.array_equal(r0["pair_T"], r1["pair_T"])
    assert [int(T[0, 3]) for T in r0["pair_T"]] == [15, 16, 17, 18, 19]
