// kabsch.h — 3x3 SVD rigid-transform solve (rows a6/a7 of SURVEY.md §8a), fp64.
//
// The reference's solver (src/map_alignment/align.py) is absent from the snapshot; this
// is the textbook Kabsch step BASELINE.json:north_star names, restated from scratch:
//   mu_p = s_p/W, mu_q = s_q/W, H = S - W mu_p mu_q^T = U Sig V^T,
//   dR = V diag(1,1,det(V U^T)) U^T (flip on the smallest singular direction),
//   dt = mu_q - dR mu_p.
// The SVD is a one-sided (Hestenes) Jacobi iteration: rotations are applied to the
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
  bool finite = true;
  for (int i = 0; i < kAccLen; ++i) finite = finite && isfinite(acc[i]);
  const double W = acc[0];
  if (!finite || !(W > 0.0)) return 1;
  const double iW = inv_f64(W);
  const Col3 mp{acc[1] * iW, acc[2] * iW, acc[3] * iW}, mq{acc[4] * iW, acc[5] * iW, acc[6] * iW};
  // H_ab = S_ab - W mp_a mq_b ; columns of H (index b)
  Col3 g0{acc[7] - W * mp.x * mq.x, acc[10] - W * mp.y * mq.x, acc[13] - W * mp.z * mq.x};
  Col3 g1{acc[8] - W * mp.x * mq.y, acc[11] - W * mp.y * mq.y, acc[14] - W * mp.z * mq.y};
  Col3 g2{acc[9] - W * mp.x * mq.z, acc[12] - W * mp.y * mq.z, acc[15] - W * mp.z * mq.z};
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
  const double noise2 = W * W * dot3(mp, mp) * dot3(mq, mq);
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
