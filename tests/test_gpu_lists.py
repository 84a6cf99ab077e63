"""Candidate lists of the loop (rows a4-a7; DESIGN.md 5 "list launches"): once the transform moves the cloud by less than a
fraction of a cell per iteration, a search launch also writes, per query, every gold point within (match distance + skin),
and the following launches read those lists instead of searching.  Every correspondence taken from a list is proved by the
triangle inequality, so the results must be those of the searching loop BIT FOR BIT — that is what is tested here, with
``lists=False`` (``MA_FLAG_NO_LISTS``: every launch searches) as the other arm — and against the CPU oracle.  Since round 4
every path resolves equal fp32 distances the same way (the smaller position in the gold map's cell order), so there is no
admissible difference left: tests/test_gpu_round4.py holds the 200-case fuzz and the clouds made of ties."""
import numpy as np
import pytest

import mapalign
from mapalign import synthetic as syn
from oracle import icp_oracle as orc

pytestmark = pytest.mark.gpu


def _both(engine, monkeypatch, iters, gate=0.10, tol=0.0, **kw):
    out = {}
    for mode in (True, False):
        res = engine.icp(iters, tol, gate, with_trace=True, lists=mode, **kw)
        out[mode] = (res, engine.list_stats())
    return out[True], out[False]


def test_list_launches_equal_the_searching_loop_bitwise_and_the_oracle(engine, monkeypatch):
    # a start close enough for the loop to come to rest well inside the 60 iterations
    gold, comp, R0, t0 = syn.make_pair(11, 120_000, 100_000, angle_deg=0.4, t0=(0.02, -0.015, 0.01))
    engine.set_pair(gold, comp)
    (on, s_on), (off, s_off) = _both(engine, monkeypatch, 60)
    assert s_off == {"builds": 0, "uses": 0}
    assert s_on["builds"] >= 1 and s_on["uses"] >= 10, s_on
    assert np.array_equal(on.T, off.T)
    assert np.array_equal(on.trace[:, :18], off.trace[:, :18])          # all 17 sums and the rmse of every iteration
    assert on.inliers == off.inliers and on.iterations == off.iterations == 60
    # the same run twice, and in a fresh context: same plan, same bits
    again = engine.icp(60, 0.0, 0.10, with_trace=True)
    assert np.array_equal(again.T, on.T) and np.array_equal(again.trace, on.trace) and engine.list_stats() == s_on
    with mapalign.MapAligner(0) as other:
        other.set_pair(gold, comp)
        c = other.icp(60, 0.0, 0.10, with_trace=True)
        assert np.array_equal(c.T, on.T) and other.list_stats() == s_on
    ref = orc.icp_oracle(gold, comp, max_iter=60, max_corr_dist=0.10, n_jobs=8)
    eR, et = orc.rt_error(on.R, on.t, ref["R"], ref["t"])
    assert eR <= 1e-5 and et <= 1e-5, (eR, et)        # fp32-stored coordinates against the fp64 oracle, as for the searching loop
    assert abs(on.inliers - ref["inliers"]) <= 1


def test_list_launches_with_weights_unmatched_points_and_a_narrow_gate(engine, monkeypatch):
    """Queries without a gold point inside the gate keep proving that (their list is complete out to gate + skin), weights
    ride along, and points that sit at the edge of the gate may enter or leave it while the lists are in use."""
    rng = np.random.default_rng(5)
    gold, comp, _, _ = syn.make_pair(12, 90_000, 80_000, angle_deg=0.3, t0=(0.015, 0.01, -0.01))
    stray = comp[rng.choice(len(comp), 6000, replace=False)] + rng.normal(0.0, 0.06, (6000, 3)).astype(np.float32)
    far = comp[rng.choice(len(comp), 3000, replace=False)] + np.float32(0.5)
    comp2 = np.concatenate([comp, stray, far]).astype(np.float32)
    w = rng.uniform(0.2, 2.0, len(comp2)).astype(np.float32)
    w[rng.choice(len(w), 500, replace=False)] = 0.0
    engine.set_gold(gold)
    engine.set_source(comp2, weights=w)
    for gate in (0.10, 0.04):
        (on, s_on), (off, _) = _both(engine, monkeypatch, 48, gate=gate)
        assert s_on["uses"] >= 8, (gate, s_on)
        assert np.array_equal(on.T, off.T) and np.array_equal(on.trace[:, :18], off.trace[:, :18]), gate
        assert 0 < on.inliers < float(w.sum())


def test_list_launches_count_their_work_and_keep_results(engine, monkeypatch):
    gold, comp, _, _ = syn.make_pair(13, 60_000, 60_000, angle_deg=0.3, t0=(0.01, 0.01, 0.01))
    engine.set_pair(gold, comp)
    a = engine.icp(40, 0.0, 0.10, with_trace=True)
    st = engine.list_stats()
    b = engine.icp(40, 0.0, 0.10, with_trace=True, count_work=True)
    assert engine.list_stats() == st and st["uses"] >= 8
    assert np.array_equal(a.T, b.T) and np.array_equal(a.trace[:, :18], b.trace[:, :18])
    plan = engine.list_plan()
    assert len(plan) == 40 and int((plan == 2).sum()) == st["uses"] and int((plan == 1).sum()) == st["builds"]
    listed = plan == 2
    # a list launch examines the 1-8 gold points on each list; the 4-byte words it reads are the positions on the list
    # (one plane of 4 for every query, the second plane for the few with more than 4 entries) and the grid-table words
    # of the few queries that fail their check and search; a search launch reads several grid-table words for every query
    per_query = b.trace[listed, 19] / len(comp)
    assert (per_query >= 1.0).all() and (per_query <= 8.5).all()
    assert (b.trace[listed, 18] <= 8.0 * len(comp)).all() and (b.trace[~listed, 18] > 3.0 * len(comp)).all()
    assert b.trace[listed, 18].mean() < b.trace[~listed, 18].mean()


def test_no_lists_for_short_runs(engine, monkeypatch):
    gold, comp, _, _ = syn.make_pair(14, 50_000, 40_000, angle_deg=0.2, t0=(0.01, 0.0, 0.0))
    engine.set_pair(gold, comp)
    engine.icp(12, 0.0, 0.10)
    assert engine.list_stats() == {"builds": 0, "uses": 0}          # three planning batches or fewer
    engine.icp(12, 1e-3, 0.10)                                      # ... with a convergence test as well
    assert engine.list_stats() == {"builds": 0, "uses": 0}


@pytest.mark.parametrize("tol", [1e-4, 1e-6, 1e-8, 3e-10])
def test_tolerance_runs_plan_lists_too_and_stop_where_the_searching_loop_stops(engine, monkeypatch, tol):
    """tol > 0 (a7's convergence test): the planner's wait for a slot doubles as the early exit.  Whether the loop stops
    before the first lists, between list launches or not at all, it must stop at the searching loop's iteration with the
    searching loop's bits."""
    gold, comp, _, _ = syn.make_pair(11, 120_000, 100_000, angle_deg=0.4, t0=(0.02, -0.015, 0.01))
    engine.set_pair(gold, comp)
    (on, s_on), (off, s_off) = _both(engine, monkeypatch, 90, tol=tol)
    assert s_off == {"builds": 0, "uses": 0}
    assert on.iterations == off.iterations and on.converged == off.converged, (on.iterations, off.iterations)
    assert np.array_equal(on.T, off.T) and on.inliers == off.inliers
    k = on.iterations
    assert np.array_equal(on.trace[:k, :18], off.trace[:k, :18])
    again = engine.icp(90, tol, 0.10)
    assert np.array_equal(again.T, on.T) and again.iterations == k and engine.list_stats() == s_on
    plan = engine.list_plan()
    assert k <= len(plan) <= min(90, k + 8)                         # at most two planning batches enqueued past the stop
    assert on.converged == (k < 90) or k == 90
    if k >= 56:                                                     # the same pair uses lists from iteration < 50 on (first test)
        assert s_on["uses"] >= 1, s_on
    print(f"tol {tol:g}: stopped after {k} iterations, converged {on.converged}, lists {s_on}")


def test_list_launches_on_a_source_cloud_from_a_depth_frame(engine, monkeypatch):
    """configs[2]'s shape: one slot per pixel, holes where the depth is invalid; the cloud's box is found by the first run
    that is long enough to use it."""
    depth, K, T = syn.make_depth_frame(0, width=320, height=240, fx=250.0, fy=250.0, cx=159.5, cy=119.5)
    depth = depth.copy()
    depth[::7, ::5] = 0.0                                            # holes
    gold, _, _, _ = syn.make_pair(0, 200_000)
    engine.set_gold(gold)
    dT = np.eye(4)
    dT[:3, :3] = syn.rotation_about_axis([0.2, 1.0, 0.1], np.deg2rad(0.2))
    dT[:3, 3] = [0.01, -0.01, 0.01]
    n = engine.set_source_from_depth(depth, K, dT @ T, 1000.0)
    assert 0 < n < depth.size
    (on, s_on), (off, s_off) = _both(engine, monkeypatch, 48)
    assert s_off == {"builds": 0, "uses": 0} and s_on["uses"] >= 4, s_on
    assert np.array_equal(on.T, off.T) and np.array_equal(on.trace[:, :18], off.trace[:, :18])


def test_a_loop_that_stops_degenerate_does_not_leave_the_planner_waiting(engine, monkeypatch):
    gold, comp, _, _ = syn.make_pair(15, 20_000, 20_000)
    engine.set_gold(gold)
    engine.set_source(comp + np.float32(50.0))                      # nothing inside the gate: the first solve stops the loop
    with pytest.raises(mapalign.MapAlignError):
        engine.icp(40, 0.0, 0.10)
    engine.set_source(comp)                                         # and the context works on
    assert engine.icp(40, 0.0, 0.10).iterations == 40


def test_list_configuration_is_per_context_and_changes_no_bit(engine, monkeypatch):
    """ma_list_config: lists on / off, the skin divisor and the reads a write must be followed by are settings of a
    context (the environment only supplies the defaults at ma_create and is not read afterwards); whatever they are,
    the transform and every iteration's sums are the searching loop's."""
    from mapalign import _lib
    gold, comp, _, _ = syn.make_pair(11, 120_000, 100_000, angle_deg=0.4, t0=(0.02, -0.015, 0.01))
    engine.set_pair(gold, comp)
    ref = engine.icp(60, 0.0, 0.10, with_trace=True, lists=False)
    assert engine.list_stats() == {"builds": 0, "uses": 0}
    plans = set()
    for skin_div, min_uses in ((7, 5), (5, 3), (4, 1), (12, 8), (64, 64)):
        engine.list_config(True, skin_div, min_uses)
        r = engine.icp(60, 0.0, 0.10, with_trace=True)
        assert np.array_equal(r.T, ref.T) and np.array_equal(r.trace[:, :18], ref.trace[:, :18]), (skin_div, min_uses)
        plans.add("".join("SWL"[int(v)] for v in engine.list_plan()))
    assert len(plans) >= 3                                          # the settings do change the plan ...
    engine.list_config(False)
    monkeypatch.setenv("MAPALIGN_LISTS", "1")                       # ... and the environment does not, once the context exists
    r = engine.icp(60, 0.0, 0.10, with_trace=True)
    assert engine.list_stats() == {"builds": 0, "uses": 0} and np.array_equal(r.T, ref.T)
    engine.list_config(True, 7, 5)                                  # (None keeps a value: the sweep above ended on 64 / 64)
    monkeypatch.setenv("MAPALIGN_LISTS", "0")
    engine.icp(60, 0.0, 0.10)
    assert engine.list_stats()["uses"] >= 10
    with mapalign.MapAligner(0) as off:                             # a context CREATED with MAPALIGN_LISTS=0 starts without lists
        off.set_pair(gold, comp)
        r = off.icp(60, 0.0, 0.10)
        assert off.list_stats() == {"builds": 0, "uses": 0} and np.array_equal(r.T, ref.T)
        off.list_config(True)
        off.icp(60, 0.0, 0.10)
        assert off.list_stats()["uses"] >= 10
    for bad in ((True, 1.5, None), (True, 100.0, None), (True, None, 65), (True, None, -1)):
        with pytest.raises(mapalign.MapAlignError) as e:
            engine.list_config(*bad)
        assert e.value.status == _lib.MA_ERR_INVALID
    engine.list_config(True, 7, 5)
