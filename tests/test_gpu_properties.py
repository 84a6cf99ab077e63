"""Property tests (hypothesis) of the HIP path through the C ABI — size-independent laws the domain offers
(SURVEY.md 4): a rigid motion of BOTH clouds conjugates the answer, the order of the points does not matter,
an exactly known motion is recovered, and aligning the aligned cloud again is the identity."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import mapalign
from mapalign import synthetic as syn, transforms

pytestmark = pytest.mark.gpu

unit = st.floats(-1.0, 1.0, allow_nan=False)
axis3 = st.tuples(unit, unit, unit).filter(lambda v: np.linalg.norm(v) > 0.2)
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _T(axis, angle, t):
    return transforms.rt_to_matrix(syn.rotation_about_axis(axis, angle), t)


@settings(max_examples=10, **COMMON)
@given(axis3, st.floats(-3.0, 3.0), st.tuples(*[st.floats(-30.0, 30.0)] * 3), st.integers(0, 50))
def test_moving_both_clouds_conjugates_the_answer(engine, axis, angle, shift, seed):
    """align(G gold, G comp) = G align(gold, comp) G^-1 for any rigid G (the grid, the brick sort and the centring
    origin all change with G; the correspondences and therefore the motion must not)."""
    gold, comp, _, _ = syn.make_pair(seed, 6000, 5000)
    G = _T(axis, angle, shift)
    engine.set_pair(gold, comp)
    T = engine.icp(30, 0.0, 0.10).T
    g2 = transforms.apply(G, gold).astype(np.float32)
    c2 = transforms.apply(G, comp).astype(np.float32)
    engine.set_pair(g2, c2)
    T2 = engine.icp(30, 0.0, 0.10).T
    want = G @ T @ np.linalg.inv(G)
    # fp32 coordinates up to ~50 m after the move: a few um per point, and a point within that of the gate may flip
    assert np.abs(T2[:3, :3] - want[:3, :3]).max() < 2e-5 and np.abs(T2[:3, 3] - want[:3, 3]).max() < 2e-4


@settings(max_examples=8, **COMMON)
@given(st.integers(0, 10_000), st.integers(0, 30))
def test_point_order_does_not_matter(engine, perm_seed, seed):
    """Both clouds shuffled: the same transform (the device sorts them into its own order; only exact ties and the
    order of a few fp64 additions can differ)."""
    gold, comp, _, _ = syn.make_pair(seed, 7000, 6500)
    rng = np.random.default_rng(perm_seed)
    engine.set_pair(gold, comp)
    a = engine.icp(25, 0.0, 0.10)
    engine.set_pair(gold[rng.permutation(len(gold))], comp[rng.permutation(len(comp))])
    b = engine.icp(25, 0.0, 0.10)
    assert np.abs(a.T - b.T).max() < 1e-9 and a.inliers == b.inliers


@settings(max_examples=10, **COMMON)
@given(axis3, st.floats(-0.025, 0.025), st.tuples(*[st.floats(-0.06, 0.06)] * 3), st.integers(0, 40))
def test_a_known_motion_inside_the_basin_is_recovered(engine, axis, angle, t, seed):
    """comp = T0^-1 (a subset of gold), exact correspondences available: align returns T0."""
    gold, _, _, _ = syn.make_pair(seed, 8000)
    T0 = _T(axis, angle, t)
    rng = np.random.default_rng(seed)
    sub = gold[rng.choice(len(gold), 5000, replace=False)].astype(np.float64)
    centre = gold.mean(0).astype(np.float64)
    # rotate about the cloud's centre so that the angle does not turn into metres of offset at the ends
    M = transforms.rt_to_matrix(np.eye(3), centre) @ T0 @ transforms.rt_to_matrix(np.eye(3), -centre)
    comp = transforms.apply(np.linalg.inv(M), sub).astype(np.float32)
    R, tt = mapalign.align(gold, comp, max_iter=60)
    assert np.linalg.norm(R - M[:3, :3]) < 1e-4 and np.linalg.norm(tt - M[:3, 3]) < 1e-3      # BASELINE tolerance


@settings(max_examples=6, **COMMON)
@given(st.integers(0, 40))
def test_aligning_the_aligned_cloud_is_the_identity(engine, seed):
    """At a fixed gate, run to the fixed point (the gated inlier set stops changing and the update becomes the
    identity): the cloud moved by that answer is then already aligned.  (Not a law of align()'s gate SCHEDULE, which
    starts wide again and may walk to a neighbouring fixed point.)"""
    gold, comp, _, _ = syn.make_pair(seed, 6000, 6000)
    engine.set_pair(gold, comp)
    first = engine.icp(400, 1e-12, 0.10)
    assert first.converged
    moved = transforms.apply(first.T, comp).astype(np.float32)
    engine.set_pair(gold, moved)
    again = engine.icp(50, 0.0, 0.10).T
    # the moved cloud is rounded to fp32 (~1e-7 relative): a handful of pairs at the gate's edge may flip
    assert np.linalg.norm(again[:3, :3] - np.eye(3)) < 2e-5 and np.linalg.norm(again[:3, 3]) < 2e-4
