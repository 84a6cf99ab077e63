"""Exact-arithmetic restatement of row a5's fixed-point sums — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` may import this module.  PARITY UNPINNED like the rest of ``oracle/`` (the reference ships no
implementation of this path: ``/root/reference/task_def.py:10`` imports ``src.map_alignment.align``, which is absent).

Since round 5 the device forms the 17 sums of an iteration {W, s_p, s_q, S, E} as 64-bit INTEGERS: every correspondence's
terms are rounded once to a multiple of a quantum 2^-e and added as integers (``csrc/host_math.h`` "Exact sums",
``csrc/kernels_search.hip::pair_terms``).  Integer work has a bit-exact bar, so this module restates those integers with
RATIONAL arithmetic (``fractions.Fraction``: every fused multiply-add is an exact product and sum followed by ONE correctly
rounded conversion to binary64, which is what ``v_fma_f64`` does) — slow pure Python, for clouds of a few thousand points.
Given the transform, the stored coordinates and the correspondences, the 17 integers must be the device's exactly.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np

GROUP = [0, 1, 1, 1, 2, 2, 2] + [3] * 9 + [4]          # accumulator -> exponent group (W, s_p, s_q, S, E)


def _fl(x: Fraction) -> float:
    """Correctly rounded binary64 of a rational (Python's Fraction -> float conversion is round-half-even)."""
    return float(x)


def fma(a: float, b: float, c: float) -> float:
    return _fl(Fraction(a) * Fraction(b) + Fraction(c))


def rne_int(x: Fraction) -> int:
    """Round to the nearest integer, ties to even — what fma(a, b, 1.5 * 2**52) leaves in its low mantissa bits."""
    f = x.numerator // x.denominator          # floor
    r = x - f
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and f % 2 == 1):
        return f + 1
    return f


def point_image(R, t, p):
    """fp64 image of a stored fp32 point under the centred-frame state (kernels_search.hip::point_image)."""
    x, y, z = (float(v) for v in p)
    return [fma(R[3 * r + 2], z, fma(R[3 * r + 1], y, fma(R[3 * r], x, t[r]))) for r in range(3)]


def pair_terms(w, P, q, e):
    """The 17 integers of one correspondence (kernels_search.hip::pair_terms); ``e`` = the five exponents."""
    w = float(w)
    qx, qy, qz = (float(v) for v in q)
    s = [w * 2.0 ** int(k) for k in e]         # w 2^e: exact (a power of two)
    out = [rne_int(Fraction(s[0]))]
    out += [rne_int(Fraction(s[1]) * Fraction(Pa)) for Pa in P]
    out += [rne_int(Fraction(s[2]) * Fraction(qa)) for qa in (qx, qy, qz)]
    for Pa in P:
        wPa = s[3] * Pa                        # one fp64 rounding, as on the device
        out += [rne_int(Fraction(wPa) * Fraction(qb)) for qb in (qx, qy, qz)]
    dx, dy, dz = P[0] - qx, P[1] - qy, P[2] - qz
    d2 = fma(dz, dz, fma(dy, dy, dx * dx))
    out.append(rne_int(Fraction(s[4]) * Fraction(d2)))
    return out


def sums(R, t, src_centred_f32, weights, gold_centred_f32, match, e):
    """17 exact integer sums over the correspondences ``match[i]`` (index into gold, < 0: none)."""
    tot = [0] * 17
    R = [float(v) for v in np.asarray(R, dtype=np.float64).ravel()]
    t = [float(v) for v in np.asarray(t, dtype=np.float64).ravel()]
    for i, j in enumerate(match):
        if j < 0:
            continue
        P = point_image(R, t, src_centred_f32[i])
        k = pair_terms(1.0 if weights is None else weights[i], P, gold_centred_f32[j], e)
        for a in range(17):
            tot[a] += k[a]
    return tot
