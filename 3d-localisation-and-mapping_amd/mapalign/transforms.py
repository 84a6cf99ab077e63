"""Rigid-transform containers in the reference's own formats.

The consumer of the alignment result, ``src/map_alignment/visualise.py``, accepts a list
whose elements are either a 4x4 homogeneous matrix (applied as ``mesh.transform(T)``,
``visualise.py:94-106,128-130``) or a ``(rotation_3x3, centre_3)`` tuple meaning Open3D's
``rotate(R, center)``: ``p' = R (p - c) + c`` (``visualise.py:82-92,118-124``).  These
helpers convert between that list and one composed 4x4.
"""
from __future__ import annotations

import numpy as np


def quat_to_rot(q):
    """Hamilton unit quaternion (qx,qy,qz,qw) -> 3x3, the pose layout of
    ``src/mapper/database_query.py:22`` (tx ty tz qx qy qz qw)."""
    x, y, z, w = (float(v) for v in q)
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def pose_to_matrix(pose7):
    """[tx,ty,tz,qx,qy,qz,qw] -> 4x4 camera->world (``pose_processor.py:242-260``)."""
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(pose7[3:7])
    T[:3, 3] = pose7[:3]
    return T


def rt_to_matrix(R, t):
    T = np.eye(4)
    T[:3, :3] = np.asarray(R, dtype=np.float64)
    T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    return T


def rotation_about_centre(R, centre):
    """(R, c) tuple -> 4x4 [R | c - R c]."""
    R = np.asarray(R, dtype=np.float64)
    c = np.asarray(centre, dtype=np.float64).reshape(3)
    return rt_to_matrix(R, c - R @ c)


def to_matrix(transformation):
    """One element of a reference-style ``transformations`` list -> 4x4."""
    if isinstance(transformation, tuple):
        return rotation_about_centre(*transformation)
    T = np.asarray(transformation, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("a transformation is a 4x4 matrix or a (rotation, centre) tuple")
    return T


def compose(transformations):
    """Apply-in-order composition of a reference-style list -> one 4x4."""
    T = np.eye(4)
    for tr in transformations:
        T = to_matrix(tr) @ T
    return T


def axis_angle(R):
    """Rotation matrix -> (unit axis, angle in [0, pi]).  Robust at both ends: the skew part gives
    the axis while sin(angle) is not tiny; next to a half turn, where it vanishes, the symmetric
    part does (R + I = 2 n n^T at angle pi)."""
    R = np.asarray(R, dtype=np.float64)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = 0.5 * np.linalg.norm(v), 0.5 * (np.trace(R) - 1.0)
    angle = float(np.arctan2(s, c))
    if s > 1e-6:
        return v / (2.0 * s), angle
    if c > 0.0:                                   # (almost) the identity: any axis
        return np.array([0.0, 0.0, 1.0]), angle
    M = 0.5 * (R + np.eye(3))                     # ~ n n^T
    n = M[:, int(np.argmax(np.diag(M)))]
    n = n / np.linalg.norm(n)
    if np.dot(n, v) < 0.0:                        # keep the sense of rotation the skew part still shows
        n = -n
    return n, angle


def rotation_from_axis_angle(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


HALF_TURN_MARGIN = 1e-3     # radians


def split_half_turns(transformations, margin=HALF_TURN_MARGIN):
    """The reference's visualiser replays a ``(R, centre)`` stage as 20 steps of
    ``expm(logm(R) / 20)`` (``src/map_alignment/visualise.py:109-126``).  The matrix logarithm of a
    rotation by pi is not real (eigenvalue -1 twice; scipy returns a complex matrix), and a coarse
    stage built from principal axes produces exact half turns whenever the two frames differ by a
    sign pattern only.  A stage within ``margin`` of a half turn is therefore emitted as TWO tuples of
    half the angle about the same axis and centre — same composition, and each logarithm on the
    principal branch with room to spare."""
    out = []
    for tr in transformations:
        if isinstance(tr, tuple):
            R, c = tr
            axis, angle = axis_angle(R)
            if angle > np.pi - margin:
                Rh = rotation_from_axis_angle(axis, 0.5 * angle)
                # second half = R Rh^T, so that the two compose to R exactly as given (not to a re-derived rotation)
                out.extend([(Rh, np.array(c, dtype=np.float64)), (np.asarray(R, dtype=np.float64) @ Rh.T, np.array(c, dtype=np.float64))])
                continue
        out.append(tr)
    return out


def apply(T, points):
    """Row-vector form of ``(T @ [p,1])[:3]`` (``pose_processor.py:254-260``)."""
    T = to_matrix(T)
    p = np.asarray(points, dtype=np.float64)
    out = p @ np.ascontiguousarray(T[:3, :3].T)
    out += T[:3, 3]
    return out


def is_rigid(T, atol=1e-9):
    T = to_matrix(T)
    R = T[:3, :3]
    return bool(np.allclose(R @ R.T, np.eye(3), atol=atol) and abs(np.linalg.det(R) - 1.0) < atol
                and np.allclose(T[3], [0, 0, 0, 1], atol=atol))
