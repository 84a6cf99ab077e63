"""Rigid-transform containers in the reference's own formats.

The consumer of the alignment result, ``src/map_alignment/visualise.py``, accepts a list
whose elements are either a 4x4 homogeneous matrix (applied as ``mesh.transform(T)``,
``visualise.py:94-106,128-130``) or a ``(rotation_3x3, centre_3)`` tuple meaning Open3D's
``rotate(R, center)``: ``p' = R (p - c) + c`` (``visualise.py:82-92,118-124``).  These
helpers convert between that list and one composed 4x4.
"""
from __future__ import annotations

import numpy as np


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
