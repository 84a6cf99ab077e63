"""Restatement (test infrastructure) of the ONLY reference code that consumes the alignment's output:
``VisualiseAlignment._apply_incremental_transformation`` and its two helpers,
``/root/reference/src/map_alignment/visualise.py:82-130``.

  * a ``(rotation, centre)`` tuple is replayed as ``steps`` applications of Open3D's
    ``mesh.rotate(expm(logm(rotation) / steps), centre)``, i.e. ``p' = R_inc (p - c) + c`` (``:109-126``);
  * a 4x4 is replayed as ``steps`` applications of ``mesh.transform(expm(logm(T) / steps))`` (``:94-106,128-130``).

Only scipy.linalg's logm / expm, as the reference imports them.  Returns the points after every stage and
every incremental matrix, so that a test can require them to be REAL (a complex increment is what the
reference's Open3D call would choke on) and the end pose to equal the composed transform."""
import numpy as np
from scipy.linalg import expm, logm

STEPS = 20     # visualise.py:82 default


def replay(transformations, points, steps=STEPS):
    p = np.array(points, dtype=np.float64)
    increments, after_stage = [], []
    for tr in transformations:
        if isinstance(tr, tuple):                       # visualise.py:90-92 -> :109-126
            rotation, centre = tr
            inc = expm(logm(np.asarray(rotation)) / steps)
            increments.append(inc)
            c = np.asarray(centre, dtype=np.float64)
            for _ in range(steps):
                p = (p - c) @ np.real(inc).T + c        # open3d rotate(R, center)
        else:                                           # visualise.py:94-106,128-130
            inc = expm(logm(np.asarray(tr)) / steps)
            increments.append(inc)
            for _ in range(steps):
                p = p @ np.real(inc[:3, :3]).T + np.real(inc[:3, 3])   # open3d transform(T)
        after_stage.append(p.copy())
    return p, increments, after_stage


def assert_replayable(transformations, points, composed, atol=1e-6, steps=STEPS):
    """Every increment real and finite; end pose = composed transform applied to the points, to `atol` metres."""
    end, incs, _ = replay(transformations, points, steps)
    for k, inc in enumerate(incs):
        assert np.all(np.isfinite(inc)), f"stage {k}: non-finite increment"
        assert np.isrealobj(inc) or np.abs(np.imag(inc)).max() == 0.0, f"stage {k}: logm left the real branch (half turn?)"
    T = np.asarray(composed, dtype=np.float64)
    want = np.asarray(points, dtype=np.float64) @ T[:3, :3].T + T[:3, 3]
    err = np.abs(end - want).max()
    assert err <= atol, f"replayed end pose is {err:.2e} m from the composed transform"
    return err
