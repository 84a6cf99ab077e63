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
constexpr double kMinCorrespondences = 3.0;

// G (3x3 row-major) is overwritten by U*Sig (columns), V accumulates the right rotations.
MA_HD inline void jacobi_svd3(double G[9], double V[9], double sig[3]) {
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int r = 0; r < 3; ++r) {
          alpha += G[3 * r + p] * G[3 * r + p];
          beta += G[3 * r + q] * G[3 * r + q];
          gamma += G[3 * r + p] * G[3 * r + q];
        }
        const double scale = sqrt(alpha * beta);
        if (fabs(gamma) <= 1e-300 + 1e-16 * scale) continue;  // already orthogonal
        off = fmax(off, fabs(gamma) / scale);
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double gp = G[3 * r + p], gq = G[3 * r + q];
          G[3 * r + p] = c * gp - s * gq;
          G[3 * r + q] = s * gp + c * gq;
          const double vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - s * vq;
          V[3 * r + q] = s * vp + c * vq;
        }
      }
    }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j)
    sig[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
}

MA_HD inline double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// acc = {W, s_p[3], s_q[3], S[9] (S_ab = sum w p_a q_b), E}.  Writes dR (row-major) and dt.
// Returns 0 on success, 1 if degenerate (W < 3, non-finite input, or rank(H) < 2), in
// which case dR = I, dt = 0.
MA_HD inline int kabsch_from_acc(const double acc[kAccLen], double dR[9], double dt[3]) {
  for (int i = 0; i < 9; ++i) dR[i] = (i % 4 == 0) ? 1.0 : 0.0;
  dt[0] = dt[1] = dt[2] = 0.0;
  bool finite = true;
  for (int i = 0; i < kAccLen; ++i) finite = finite && isfinite(acc[i]);
  const double W = acc[0];
  if (!finite || !(W >= kMinCorrespondences)) return 1;
  double mp[3], mq[3];
  for (int a = 0; a < 3; ++a) {
    mp[a] = acc[1 + a] / W;
    mq[a] = acc[4 + a] / W;
  }
  double G[9], V[9], sig[3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) G[3 * a + b] = acc[7 + 3 * a + b] - W * mp[a] * mq[b];
  jacobi_svd3(G, V, sig);
  // order the singular directions: i0 largest, i2 smallest
  int i0 = 0, i1 = 1, i2 = 2;
  if (sig[i0] < sig[i1]) { int s = i0; i0 = i1; i1 = s; }
  if (sig[i1] < sig[i2]) { int s = i1; i1 = i2; i2 = s; }
  if (sig[i0] < sig[i1]) { int s = i0; i0 = i1; i1 = s; }
  if (!(sig[i1] > 1e-12 * sig[i0]) || !(sig[i0] > 0.0)) return 1;
  // U columns for the two leading directions; the third is their cross product, which
  // also covers a vanishing third singular value (planar correspondences).
  double u0[3], u1[3], u2[3], v0[3], v1[3], v2[3];
  for (int r = 0; r < 3; ++r) {
    u0[r] = G[3 * r + i0] / sig[i0];
    u1[r] = G[3 * r + i1] / sig[i1];
    v0[r] = V[3 * r + i0];
    v1[r] = V[3 * r + i1];
  }
  u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
  u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
  u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
  v2[0] = v0[1] * v1[2] - v0[2] * v1[1];
  v2[1] = v0[2] * v1[0] - v0[0] * v1[2];
  v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
  // With u2 = u0 x u1 and v2 = v0 x v1 both frames are right-handed, so
  // dR = v0 u0^T + v1 u1^T + v2 u2^T has det +1: this IS V diag(1,1,det(VU^T)) U^T.
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) dR[3 * a + b] = v0[a] * u0[b] + v1[a] * u1[b] + v2[a] * u2[b];
  for (int a = 0; a < 3; ++a)
    dt[a] = mq[a] - (dR[3 * a] * mp[0] + dR[3 * a + 1] * mp[1] + dR[3 * a + 2] * mp[2]);
  return 0;
}

}  // namespace ma
