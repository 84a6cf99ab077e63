// kabsch.h — 3x3 SVD rigid-transform solve (rows a6/a7 of SURVEY.md §8a), fp64.
//
// The reference's solver (src/map_alignment/align.py) is absent from the snapshot; this
// is the textbook Kabsch step BASELINE.json:north_star names, restated from scratch:
//   mu_p = s_p/W, mu_q = s_q/W, H = S - W mu_p mu_q^T = U Sig V^T,
//   dR = V diag(1,1,det(V U^T)) U^T (flip on the smallest singular direction),
//   dt = mu_q - dR mu_p.
// Where the largest eigenvalue of Horn's 4x4 matrix stands well clear of the others — every iteration of a real
// alignment — dR comes from its eigenvector (rotation_qcp), without any singular vector.  Everything else takes the SVD,
// a one-sided (Hestenes) Jacobi iteration: rotations are applied to the
// columns of H until they are mutually orthogonal; it needs no 3x3 eigen-solver, has no
// branches on the data layout and converges quadratically (<= 6 sweeps in fp64 for a
// 3x3), which is what a single GPU lane wants.
//
// The same inline code is compiled for the device (solve kernel in mapalign.hip) and for
// the host (tests/ build it with g++ to check it against numpy.linalg.svd on a CPU box).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define MA_HD __host__ __device__
#else
#define MA_HD
#endif

namespace ma {

constexpr int kAccLen = 17;

struct Col3 { double x, y, z; };

MA_HD inline double dot3(const Col3& a, const Col3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MA_HD inline Col3 cross3(const Col3& a, const Col3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
MA_HD inline void swap_col(Col3& a, Col3& b) { const Col3 t = a; a = b; b = t; }

// The solve is ONE lane of dependent fp64 work on the device, so its time is its
// instruction count.  An IEEE divide or square root is ~30 fp64 instructions there (scaling,
// fix-ups); the hardware seeds (v_rcp_f64, v_rsq_f64: ~26 bits) plus one Newton step are 4-5
// and accurate to an ulp or two, which is all this needs (the rotation keeps c^2 + s^2 = 1 by
// construction, see jacobi_rotate).  Operands here are never 0, inf or denormal: every call
// site is guarded (W > 0, gamma^2 > 0, singular values checked).  The host build (unit
// test against numpy) uses the exact operations.
MA_HD inline double inv_f64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}
MA_HD inline double rsqrt_f64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  return fma(fma(-x * y, y, 1.0), 0.5 * y, y);
#else
  return 1.0 / sqrt(x);
#endif
}
MA_HD inline double sqrt_f64(double x) {   // x > 0
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x), s = x * y;
  return fma(fma(-s, s, x), 0.5 * y, s);
#else
  return sqrt(x);
#endif
}

// One Hestenes rotation: makes columns gp, gq of G orthogonal, applies the same rotation
// to the columns vp, vq of V.  Returns the squared relative size of the off-diagonal it
// removed.  tan(theta) = 2 gamma sgn(d) / (|d| + sqrt(d^2 + 4 gamma^2)) with d = beta -
// alpha is the smaller root (|theta| <= pi/4); c = rsqrt(1 + t^2), s = c t keeps
// c^2 + s^2 = 1 to rounding whatever the accuracy of t, so V stays orthogonal.
MA_HD inline double jacobi_rotate(Col3& gp, Col3& gq, Col3& vp, Col3& vq) {
  const double alpha = dot3(gp, gp), beta = dot3(gq, gq), gamma = dot3(gp, gq);
  const double ab = alpha * beta, g2 = gamma * gamma;
  if (g2 <= 1e-300 + 1e-32 * ab) return 0.0;  // already orthogonal
  const double d = beta - alpha;
  const double t = (d >= 0.0 ? 2.0 : -2.0) * gamma * inv_f64(fabs(d) + sqrt_f64(d * d + 4.0 * g2));
  const double c = rsqrt_f64(1.0 + t * t);
  const double s = c * t;
  const Col3 g0 = gp, g1 = gq, w0 = vp, w1 = vq;
  gp = {c * g0.x - s * g1.x, c * g0.y - s * g1.y, c * g0.z - s * g1.z};
  gq = {s * g0.x + c * g1.x, s * g0.y + c * g1.y, s * g0.z + c * g1.z};
  vp = {c * w0.x - s * w1.x, c * w0.y - s * w1.y, c * w0.z - s * w1.z};
  vq = {s * w0.x + c * w1.x, s * w0.y + c * w1.y, s * w0.z + c * w1.z};
  return g2 * inv_f64(ab);
}

// A = U Sig V^T for a 3x3 given by its COLUMNS g0,g1,g2 (overwritten by the columns of
// U Sig); v0,v1,v2 receive the columns of V.  No run-time array indexing: everything stays
// in registers on the device.
MA_HD inline void jacobi_svd3_cols(Col3& g0, Col3& g1, Col3& g2, Col3& v0, Col3& v1, Col3& v2) {
  v0 = {1.0, 0.0, 0.0};
  v1 = {0.0, 1.0, 0.0};
  v2 = {0.0, 0.0, 1.0};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = jacobi_rotate(g0, g1, v0, v1);
    off = fmax(off, jacobi_rotate(g0, g2, v0, v2));
    off = fmax(off, jacobi_rotate(g1, g2, v1, v2));
    if (off < 1e-30) break;  // squared relative off-diagonal
  }
}

// Row-major array front end (used by the host-side unit test): G is overwritten by U*Sig.
MA_HD inline void jacobi_svd3(double G[9], double V[9], double sig[3]) {
  Col3 g0{G[0], G[3], G[6]}, g1{G[1], G[4], G[7]}, g2{G[2], G[5], G[8]}, v0, v1, v2;
  jacobi_svd3_cols(g0, g1, g2, v0, v1, v2);
  G[0] = g0.x; G[3] = g0.y; G[6] = g0.z; G[1] = g1.x; G[4] = g1.y; G[7] = g1.z;
  G[2] = g2.x; G[5] = g2.y; G[8] = g2.z;
  V[0] = v0.x; V[3] = v0.y; V[6] = v0.z; V[1] = v1.x; V[4] = v1.y; V[7] = v1.z;
  V[2] = v2.x; V[5] = v2.y; V[8] = v2.z;
  sig[0] = sqrt(dot3(g0, g0)); sig[1] = sqrt(dot3(g1, g1)); sig[2] = sqrt(dot3(g2, g2));
}

// The rotation without an SVD, for the case every iteration of a real alignment is (Horn 1987; Theobald 2005): the unit
// quaternion of the optimal proper rotation is the eigenvector of the largest eigenvalue of the symmetric 4x4
//   N = [ Sxx+Syy+Szz   Syz-Szy        Szx-Sxz        Sxy-Syx      ]
//       [ .             Sxx-Syy-Szz    Sxy+Syx        Szx+Sxz      ]
//       [ .             .             -Sxx+Syy-Szz    Syz+Szy      ]
//       [ .             .              .             -Sxx-Syy+Szz  ]        S_ab = H_ab = sum w p'_a q_b (centred),
// whose characteristic polynomial has no cubic term (trace N = 0):  P(l) = l^4 + c2 l^2 + c1 l + c0,  c2 = -2 |H|_F^2,
// c1 = -8 det H,  c0 = det N.  Its largest root is (G - E_min) / 2, G = sum w (|p' - mu_p|^2 + |q - mu_q|^2), E_min the
// residual after the optimal rotation; l0 = G / 2 = E_c / 2 + trace H (E_c: the residual the sums hold, centred) is above
// it by (E_c - E_min) / 2 — next to nothing — and Newton's iteration comes down on it in 2-4 steps.  The eigenvector is a
// column of adj(N - l I); for rotations below half a turn its first column, (q0, qx, qy, qz) up to scale.
// ONE lane runs the solve on the device and its time is its instruction count (a lone wavefront gets one vector
// instruction through every ~9 cycles): this is ~200 instructions where the Jacobi SVD below is ~500.
// Accuracy falls with the gaps between the largest eigenvalue and the others (P'(l) / l^3, the size of the adjugate column):
// accepted only where both exceed 0.02 — |dR - dR_svd| <= 1e-10 there over 20 000 random problems, 2e-14 at the median;
// thin or line-like clouds, half turns and rank-deficient H fail the test and take the SVD.  (A mirrored H, det < 0, needs
// no special case: the quaternion's rotation is proper by construction — the one the SVD form reaches through its
// determinant correction.)
// Returns false (dR untouched) when it declines.
MA_HD inline bool rotation_qcp(const Col3& g0, const Col3& g1, const Col3& g2, const double frob2, const double l0, double dR[9]) {
  // columns of H: g_b = (H_0b, H_1b, H_2b)
  const double Sxx = g0.x, Syx = g0.y, Szx = g0.z, Sxy = g1.x, Syy = g1.y, Szy = g1.z, Sxz = g2.x, Syz = g2.y, Szz = g2.z;
  const double n00 = (Sxx + Syy) + Szz, n11 = (Sxx - Syy) - Szz, n22 = (Syy - Sxx) - Szz, n33 = (Szz - Sxx) - Syy;
  const double n01 = Syz - Szy, n02 = Szx - Sxz, n03 = Sxy - Syx, n12 = Sxy + Syx, n13 = Szx + Sxz, n23 = Syz + Szy;
  const double c2 = -2.0 * frob2;
  const double detH = Sxx * (Syy * Szz - Syz * Szy) + Sxy * (Syz * Szx - Syx * Szz) + Sxz * (Syx * Szy - Syy * Szx);
  const double c1 = -8.0 * detH;
  // det N through the 2x2 minors of rows {0,1} and {2,3} (Laplace expansion along the first two rows)
  const double a01 = n00 * n11 - n01 * n01, a02 = n00 * n12 - n02 * n01, a03 = n00 * n13 - n03 * n01;
  const double a12 = n01 * n12 - n02 * n11, a13 = n01 * n13 - n03 * n11, a23 = n02 * n13 - n03 * n12;
  const double b01 = n02 * n13 - n12 * n03, b02 = n02 * n23 - n22 * n03, b03 = n02 * n33 - n23 * n03;
  const double b12 = n12 * n23 - n22 * n13, b13 = n12 * n33 - n23 * n13, b23 = n22 * n33 - n23 * n23;
  const double c0 = ((a01 * b23 - a02 * b13) + (a03 * b12 + a12 * b03)) + (a23 * b01 - a13 * b02);
  const double c2x2 = 2.0 * c2;
  double l = l0, dP = 0.0, step = 0.0;
  bool settled = false;
  for (int it = 0; it < 8 && !settled; ++it) {
    const double l2 = l * l;
    const double P = ((l2 + c2) * l + c1) * l + c0;
    dP = (4.0 * l2 + c2x2) * l + c1;
    if (!(dP > 0.0)) return false;
    step = P * inv_f64(dP);
    l -= step;
    // the error squares with every step (times at most ~150 / l where the acceptance test below holds): the step
    // that is smaller than 1e-9 l lands within rounding of the root
    settled = fabs(step) <= 1e-9 * l;
  }
  if (!settled) return false;
  const double l3 = l * l * l;
  if (!(dP > 0.02 * l3)) return false;
  // first column of adj(N - l I): the cofactors of its first row
  const double m11 = n11 - l, m22 = n22 - l, m33 = n33 - l;
  const double p23 = m22 * m33 - n23 * n23, p13 = n12 * m33 - n23 * n13, p12 = n12 * n23 - m22 * n13;
  const double p03 = n02 * m33 - n23 * n03, p02 = n02 * n23 - m22 * n03, p01 = n02 * n13 - n12 * n03;
  const double q0 = (m11 * p23 - n12 * p13) + n13 * p12;
  const double qx = -((n01 * p23 - n12 * p03) + n13 * p02);
  const double qy = (n01 * p13 - m11 * p03) + n13 * p01;
  const double qz = -((n01 * p12 - m11 * p02) + n12 * p01);
  const double n2 = (q0 * q0 + qx * qx) + (qy * qy + qz * qz);
  if (!(n2 > 4e-4 * l3 * l3)) return false;           // |column| > 0.02 l^3
  const double in2 = inv_f64(n2);
  const double aa = q0 * q0 * in2, bb = qx * qx * in2, cc = qy * qy * in2, dd = qz * qz * in2;
  const double ab = 2.0 * q0 * qx * in2, ac = 2.0 * q0 * qy * in2, ad = 2.0 * q0 * qz * in2;
  const double bc = 2.0 * qx * qy * in2, bd = 2.0 * qx * qz * in2, cd = 2.0 * qy * qz * in2;
  dR[0] = (aa + bb) - (cc + dd); dR[1] = bc - ad;               dR[2] = bd + ac;
  dR[3] = bc + ad;               dR[4] = (aa - bb) + (cc - dd); dR[5] = cd - ab;
  dR[6] = bd - ac;               dR[7] = cd + ab;               dR[8] = (aa - bb) - (cc - dd);
  return true;
}

MA_HD inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// acc = {W, s_p[3], s_q[3], S[9] (S_ab = sum w p_a q_b), E}.  Writes dR (row-major) and dt.
// Returns 0 on success, 1 if degenerate, in which case dR = I, dt = 0.  Degenerate means: a
// non-finite sum, no weight at all (W <= 0), or rank(H) < 2 — which is what fewer than three
// non-collinear correspondences produce (0, 1 or 2 pairs, or any number on one line, give
// rank <= 1).  W is a sum of WEIGHTS, not a count: normalised or confidence weights may sum
// to less than 3 with millions of matches, so the test is on the rank, not on W >= 3.
MA_HD inline int kabsch_from_acc(const double* acc, double dR[9], double dt[3]) {
  dR[0] = 1.0; dR[1] = 0.0; dR[2] = 0.0; dR[3] = 0.0; dR[4] = 1.0; dR[5] = 0.0;
  dR[6] = 0.0; dR[7] = 0.0; dR[8] = 1.0;
  dt[0] = dt[1] = dt[2] = 0.0;
  const double W = acc[0];
  if (!(W > 0.0)) return 1;        // (NaN included; the other sums are looked at below, off the common path)
  const double iW = inv_f64(W);
  const Col3 mp{acc[1] * iW, acc[2] * iW, acc[3] * iW}, mq{acc[4] * iW, acc[5] * iW, acc[6] * iW};
  // H_ab = S_ab - W mp_a mq_b ; columns of H (index b)
  Col3 g0{acc[7] - W * mp.x * mq.x, acc[10] - W * mp.y * mq.x, acc[13] - W * mp.z * mq.x};
  Col3 g1{acc[8] - W * mp.x * mq.y, acc[11] - W * mp.y * mq.y, acc[14] - W * mp.z * mq.y};
  Col3 g2{acc[9] - W * mp.x * mq.z, acc[12] - W * mp.y * mq.z, acc[15] - W * mp.z * mq.z};
  const double noise2 = W * W * dot3(mp, mp) * dot3(mq, mq);
#ifndef MA_NO_QCP        // (A/B: -DMA_NO_QCP sends every solve through the SVD)
  {
    // rotation_qcp accepts only well-separated eigenvalues — the second singular value a fair share of the largest: rank
    // >= 2 with room to spare — so the rank test of the general path below holds wherever it does; the noise test is made here
    // (|H|_F^2 > 3e-24 noise2  =>  sigma_0^2 >= |H|_F^2 / 3 > 1e-24 noise2).
    const double frob2 = dot3(g0, g0) + dot3(g1, g1) + dot3(g2, g2);
    const Col3 dm{mp.x - mq.x, mp.y - mq.y, mp.z - mq.z};
    const double l0 = 0.5 * (acc[16] - W * dot3(dm, dm)) + ((g0.x + g1.y) + g2.z);
    if (frob2 > 3e-24 * noise2 && l0 > 0.0 && rotation_qcp(g0, g1, g2, frob2, l0, dR)) {
      dt[0] = mq.x - (dR[0] * mp.x + dR[1] * mp.y + dR[2] * mp.z);
      dt[1] = mq.y - (dR[3] * mp.x + dR[4] * mp.y + dR[5] * mp.z);
      dt[2] = mq.z - (dR[6] * mp.x + dR[7] * mp.y + dR[8] * mp.z);
      return 0;
    }
  }
#endif
  // a non-finite sum fails every comparison of the quaternion path (its tests are all of the form !(x > y)) and ends here
  bool finite = true;
  for (int i = 0; i < kAccLen; ++i) finite = finite && isfinite(acc[i]);
  if (!finite) return 1;
  Col3 v0, v1, v2;
  jacobi_svd3_cols(g0, g1, g2, v0, v1, v2);
  // squared singular values: their order and the rank test need no square root
  double s0 = dot3(g0, g0), s1 = dot3(g1, g1), s2 = dot3(g2, g2);
  // order the singular directions (largest first) with conditional swaps
  if (s0 < s1) { swap_col(g0, g1); swap_col(v0, v1); const double t = s0; s0 = s1; s1 = t; }
  if (s1 < s2) { swap_col(g1, g2); swap_col(v1, v2); const double t = s1; s1 = s2; s2 = t; }
  if (s0 < s1) { swap_col(g0, g1); swap_col(v0, v1); const double t = s0; s0 = s1; s1 = t; }
  if (!(s1 > 1e-24 * s0) || !(s0 > 0.0)) return 1;   // sigma_1 <= 1e-12 sigma_0: rank < 2
  // H = S - W mu_p mu_q^T is a difference of terms of size W |mu_p| |mu_q|; a result below
  // 1e-12 of that is rounding noise (a single correspondence gives H = 0 up to rounding, and
  // noise has no small singular value for the rank test to see)
  if (!(s0 > 1e-24 * noise2)) return 1;
  // U columns of the two leading directions; the third of BOTH frames is the cross product
  // of the first two.  That covers a vanishing third singular value (planar
  // correspondences) and makes both frames right-handed, so
  //   dR = v0 u0^T + v1 u1^T + (v0 x v1)(u0 x u1)^T
  // has det +1: it IS V diag(1,1,det(V U^T)) U^T with the flip on the smallest direction.
  const double i0 = rsqrt_f64(s0), i1 = rsqrt_f64(s1);
  const Col3 u0{g0.x * i0, g0.y * i0, g0.z * i0}, u1{g1.x * i1, g1.y * i1, g1.z * i1};
  const Col3 u2 = cross3(u0, u1), w2 = cross3(v0, v1);
  dR[0] = v0.x * u0.x + v1.x * u1.x + w2.x * u2.x;
  dR[1] = v0.x * u0.y + v1.x * u1.y + w2.x * u2.y;
  dR[2] = v0.x * u0.z + v1.x * u1.z + w2.x * u2.z;
  dR[3] = v0.y * u0.x + v1.y * u1.x + w2.y * u2.x;
  dR[4] = v0.y * u0.y + v1.y * u1.y + w2.y * u2.y;
  dR[5] = v0.y * u0.z + v1.y * u1.z + w2.y * u2.z;
  dR[6] = v0.z * u0.x + v1.z * u1.x + w2.z * u2.x;
  dR[7] = v0.z * u0.y + v1.z * u1.y + w2.z * u2.y;
  dR[8] = v0.z * u0.z + v1.z * u1.z + w2.z * u2.z;
  dt[0] = mq.x - (dR[0] * mp.x + dR[1] * mp.y + dR[2] * mp.z);
  dt[1] = mq.y - (dR[3] * mp.x + dR[4] * mp.y + dR[5] * mp.z);
  dt[2] = mq.z - (dR[6] * mp.x + dR[7] * mp.y + dR[8] * mp.z);
  return 0;
}

}  // namespace ma
