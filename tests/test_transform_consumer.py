"""The staged output format against its only surviving consumer, on the CPU:
tests/visualiser_replay.py restates visualise.py:82-130 (20 steps of expm(logm(.)/20) per stage), and every
kind of list the adaptor emits must replay to the composed transform with all increments real."""
import numpy as np
import pytest

from mapalign import coarse, transforms
from mapalign import synthetic as syn

from visualiser_replay import assert_replayable, replay


def _points(seed=0, n=200):
    rng = np.random.default_rng(seed)
    return rng.uniform(-10, 10, size=(n, 3)) + np.array([30.0, -5.0, 2.0])


@pytest.mark.parametrize("angle_deg", [0.0, 2.0, 90.0, 150.0, 179.0, 179.99, 179.99995, 180.0])
def test_rotation_stage_replays_for_every_angle_after_the_half_turn_split(angle_deg):
    R = syn.rotation_about_axis([0.2, -0.5, 0.84], np.deg2rad(angle_deg))
    c = np.array([12.0, 1.5, 0.7])
    stages = transforms.split_half_turns([(R, c)]) + [transforms.rt_to_matrix(np.eye(3), [7.0, -1.0, 0.5])]
    near_pi = np.deg2rad(angle_deg) > np.pi - transforms.HALF_TURN_MARGIN
    assert len(stages) == (3 if near_pi else 2)
    assert np.allclose(transforms.compose(stages), transforms.compose([(R, c), stages[-1]]), atol=1e-12)
    assert_replayable(stages, _points(), transforms.compose(stages), atol=1e-6)


def test_an_exact_half_turn_is_what_breaks_the_consumer_without_the_split():
    """The coarse stage's candidates are axes_gold @ diag(+-1, +-1, .) @ axes_comp.T: with equal frames three
    of the four are exact half turns.  Unsplit, scipy's logm leaves the real branch (complex increment)."""
    for R in coarse.candidate_rotations(np.eye(3), np.eye(3))[1:]:
        axis, angle = transforms.axis_angle(R)
        assert abs(angle - np.pi) < 1e-12
        _, incs, _ = replay([(R, np.zeros(3))], _points())
        assert np.iscomplexobj(incs[0]) and np.abs(np.imag(incs[0])).max() > 1e-3
        split = transforms.split_half_turns([(R, np.zeros(3))])
        assert len(split) == 2
        assert_replayable(split, _points(), transforms.rotation_about_centre(R, np.zeros(3)))


def test_axis_angle_round_trip_and_halves_compose_exactly():
    rng = np.random.default_rng(5)
    for _ in range(200):
        ang = rng.choice([rng.uniform(0, np.pi), np.pi - rng.uniform(0, 2e-3), np.pi])
        R = syn.rotation_about_axis(rng.normal(size=3), ang)
        axis, angle = transforms.axis_angle(R)
        assert abs(angle - ang) < 1e-7
        assert np.allclose(transforms.rotation_from_axis_angle(axis, angle), R, atol=2e-7)
        parts = transforms.split_half_turns([(R, rng.normal(size=3))])
        M = np.eye(3)
        for Rk, _ in parts:
            M = Rk @ M
        assert np.allclose(M, R, atol=1e-14)                     # the halves compose to R as given


def test_full_staged_list_shape_of_the_adaptor_replays():
    """[(R, centre), 4x4 translation, 4x4 ICP refinement] — the mixed list of visualise.py:82-96 — for a far start."""
    T_true = transforms.rt_to_matrix(syn.rotation_about_axis([0.05, -0.03, 1.0], np.deg2rad(150.0)), [7.0, -4.0, 1.5])
    cc, cg = np.array([-3.0, 2.0, 0.4]), np.array([10.0, 0.0, 1.2])
    R = T_true[:3, :3] @ syn.rotation_about_axis([0.3, 0.1, 0.9], np.deg2rad(-1.5))       # coarse: 1.5 degrees off
    stages = transforms.split_half_turns([(R, cc)]) + [transforms.rt_to_matrix(np.eye(3), cg - cc)]
    T0 = transforms.compose(stages)
    stages.append(T_true @ np.linalg.inv(T0))                                            # the refinement stage
    assert_replayable(stages, _points(3), T_true, atol=1e-6)
