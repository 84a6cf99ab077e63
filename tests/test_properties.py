"""Property tests (hypothesis) of the algebra the path relies on — CPU only.

SURVEY.md §4: rigid invariances of the oracle, composition rules of the reference's mixed
transform list (visualise.py:82-96), PLY round trips, shard partitions."""
import numpy as np
from hypothesis import given, settings, strategies as st

from mapalign import dist as mdist, ply, synthetic as syn, transforms
from oracle import icp_oracle as orc

angles = st.floats(-3.0, 3.0, allow_nan=False)
vec3 = st.tuples(*[st.floats(-5.0, 5.0, allow_nan=False)] * 3)
axis3 = vec3.filter(lambda v: np.linalg.norm(v) > 1e-3)


@settings(max_examples=40, deadline=None)
@given(axis3, angles, vec3, axis3, angles, vec3)
def test_compose_is_associative_and_rigid(a1, g1, t1, a2, g2, c2):
    T1 = transforms.rt_to_matrix(syn.rotation_about_axis(a1, g1), t1)
    stage2 = (syn.rotation_about_axis(a2, g2), np.array(c2))           # (R, centre) tuple form
    T = transforms.compose([T1, stage2])
    assert transforms.is_rigid(T, 1e-9)
    p = np.array([[0.3, -1.0, 2.0], [4.0, 0.0, -0.5]])
    assert np.allclose(transforms.apply(T, p), transforms.apply(stage2, transforms.apply(T1, p)), atol=1e-9)
    assert np.allclose(transforms.compose([T1, np.linalg.inv(T1)]), np.eye(4), atol=1e-9)


@settings(max_examples=15, deadline=None)
@given(axis3, st.floats(-0.05, 0.05), st.tuples(*[st.floats(-0.05, 0.05)] * 3), st.integers(0, 10_000))
def test_oracle_kabsch_step_is_equivariant(axis, ang, t, seed):
    """One Kabsch step on exact correspondences recovers the motion, whatever it is."""
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(200, 3)) * [3.0, 1.0, 0.5]
    R0 = syn.rotation_about_axis(axis, ang)
    q = p @ R0.T + np.array(t)
    dR, dt, ok = orc.kabsch_from_acc(orc.accumulate(p, q, np.ones(len(p)), np.zeros(len(p))))
    assert ok and np.allclose(dR, R0, atol=1e-10) and np.allclose(dt, t, atol=1e-10)
    # moving BOTH clouds by a rigid G conjugates the solution: dR' = G dR G^T
    G = syn.rotation_about_axis([0.3, 0.2, 0.9], 0.7)
    dR2, _, _ = orc.kabsch_from_acc(orc.accumulate(p @ G.T, q @ G.T, np.ones(len(p)), np.zeros(len(p))))
    assert np.allclose(dR2, G @ dR @ G.T, atol=1e-9)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 400), st.booleans(), st.integers(0, 2**31 - 1))
def test_ply_round_trip(n, binary, seed):
    import tempfile, os
    rng = np.random.default_rng(seed)
    pts = (rng.normal(size=(n, 3)) * 50).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "c.ply")
        ply.write_ply(path, pts, binary=binary)
        out = ply.read_ply(path)
    assert out.shape == (n, 3) and np.allclose(out, pts.astype(np.float64), rtol=1e-6, atol=1e-6)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 10_000_000), st.integers(1, 64))
def test_shard_bounds_partition(n, world):
    spans = [mdist.shard_bounds(n, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(spans, spans[1:]))
    assert sorted(i for r in range(world) for i in mdist.pairs_for_rank(min(n, 50), world, r)) == list(range(min(n, 50)))
