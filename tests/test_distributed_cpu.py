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


# ----------------------------------------------------------------------------- round 5: integer sums, bounds agreement, p2p bootstrap
class FixedPointShardEngine(OracleShardEngine):
    """The stand-in with the HIP engine's EXCHANGE FORMAT (round 5): the 17 sums of a shard as int64 fixed point — every
    correspondence's terms rounded once to a quantum 2^-e — with the exponents derived from bounds the ranks agree on
    (the arithmetic of csrc/host_math.h::sum_scale, restated)."""
    GROUP = np.array([0, 1, 1, 1, 2, 2, 2] + [3] * 9 + [4])

    def __init__(self, gold, comp_shard, gate):
        super().__init__(gold, comp_shard, gate)
        self.qb = float(np.abs(self.index.gold).max())
        self.bounds = None

    def sum_bounds(self):
        return np.array([float(len(self.comp)), 1.0, float(np.linalg.norm(self.comp, axis=1).max())])

    def agree_sum_bounds(self, group=None):
        mine = torch.from_numpy(self.sum_bounds())
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        out = [torch.empty_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(out, mine, group=group)
        else:
            out = [mine]
        b = torch.stack(out).numpy()
        self.bounds = np.array([b[:, 0].sum(), b[:, 1].max(), b[:, 2].max()])

    def exponents(self):
        n, wmax, _ = self.bounds
        pb = self.qb + self.gate * 1.001 + 1e-6
        budget = min(50, 61 - int(np.frexp(n)[1]))
        term = [wmax, wmax * pb, wmax * self.qb, wmax * pb * self.qb, wmax * (self.gate ** 2 * 1.001 + 1e-12)]
        return np.array([budget - int(np.frexp(v)[1]) for v in term])

    def accumulate_into(self, acc):
        e = self.exponents()
        p = self.comp @ self.R.T + self.t
        idx, d2 = self.index.query(p)
        m = d2 <= self.gate ** 2
        p, q, d2 = p[m], self.index.gold[idx[m]], d2[m]
        terms = np.concatenate([np.ones((len(p), 1)), p, q, (p[:, :, None] * q[:, None, :]).reshape(len(p), 9), d2[:, None]], axis=1)
        k = np.rint(np.ldexp(terms, e[self.GROUP])).astype(np.int64)          # one rounding per term, ties to even
        acc.copy_(torch.from_numpy(k.sum(axis=0)))

    def solve_from(self, acc):
        e = self.exponents()
        a = np.ldexp(acc.numpy().astype(np.float64), -e[self.GROUP])
        dR, dt, ok = self.orc.kabsch_from_acc(a)
        assert ok
        self.R, self.t = dR @ self.R, dR @ self.t + dt
        self.iters += 1


class FakeP2PAligner:
    def __init__(self, rank, fail=False):
        self.rank, self.fail, self.inited = rank, fail, None

    def p2p_export(self):
        if self.fail:
            raise RuntimeError("no fine-grained memory today")
        return bytes([self.rank]) * 128

    def p2p_init(self, blobs, rank):
        self.inited = (list(blobs), rank)


def _round5_worker(rank, world, port, out_dir):
    for p in (PKG, ROOT):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapalign import dist as mdist
    from mapalign import synthetic as syn
    gold, comp, R0, t0 = syn.make_pair(3, 6000, 5000)
    # an UNEVEN partition (rank 0 holds 70 %): integer sums do not care
    cut = int(0.7 * len(comp))
    lo, hi = (0, cut) if rank == 0 else (cut, len(comp))
    eng = FixedPointShardEngine(gold, comp[lo:hi], 0.10)
    acc = torch.zeros(17, dtype=torch.int64)
    T, info = mdist.sharded_icp(eng, acc, max_iter=15)               # agree_sum_bounds, then 15 x (accumulate, all_reduce(int64), solve)
    # the same on ONE engine holding the whole cloud, bounds = its own: must be the same BITS
    whole = FixedPointShardEngine(gold, comp, 0.10)
    whole.bounds = whole.sum_bounds()
    assert np.array_equal(whole.bounds[:2], eng.bounds[:2])          # n_total and wmax: what the ranks agreed on
    acc1 = torch.zeros(17, dtype=torch.int64)
    whole.state_set(None)
    for _ in range(15):
        whole.accumulate_into(acc1)
        whole.solve_from(acc1)
    T1, _ = whole.state_get()
    # p2p bootstrap: blobs gathered in rank order; a failing export reaches every rank as an error, nobody hangs
    a = FakeP2PAligner(rank)
    mdist.bootstrap_p2p(a)
    ok_blobs = a.inited == ([bytes([r]) * 128 for r in range(world)], rank)
    raised = ""
    try:
        mdist.bootstrap_p2p(FakeP2PAligner(rank, fail=(rank == 1)))
    except RuntimeError as e:
        raised = str(e)
    np.savez(os.path.join(out_dir, f"r5_{rank}.npz"), T=T, T1=T1, ok_blobs=ok_blobs, raised=raised, iters=info["iterations"])
    dist.destroy_process_group()


def test_world2_integer_sums_any_partition_gives_the_single_process_bits_and_p2p_bootstrap(tmp_path):
    """Round 5 on the CPU: mapalign.dist.sharded_icp with the int64 exchange buffer over a 2-rank gloo group and a 70 / 30
    partition — agree_sum_bounds, integer all-reduce — ends on the bits of ONE engine holding the whole cloud (the property
    the HIP engine's exact sums exist for; the GPU tests check it on the device); and mapalign.dist.bootstrap_p2p gathers
    the ranks' blobs in rank order and turns one rank's failure into an error on every rank."""
    world = 2
    mp.spawn(_round5_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(tmp_path / f"r5_{r}.npz") for r in range(world)]
    assert np.array_equal(z[0]["T"], z[1]["T"]) and int(z[0]["iters"]) == 15
    assert np.array_equal(z[0]["T"], z[0]["T1"]) and np.array_equal(z[1]["T"], z[1]["T1"])      # sharded == whole, bit for bit
    assert bool(z[0]["ok_blobs"]) and bool(z[1]["ok_blobs"])
    for r in range(world):
        assert "p2p bootstrap failed" in str(z[r]["raised"]) and "rank 1" in str(z[r]["raised"])
