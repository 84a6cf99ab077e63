// host_math.h — host-side arithmetic of libmapalign that needs no HIP: frame conversions
// between the caller's world frame and the context's centred frame, grid geometry, launch
// geometry of the search kernel.  Plain C++17, so that tests/test_host_math.py can build it
// with g++ -fsanitize=address,undefined on a CPU box (sanitizer row) and check
// it against NumPy; mapalign.hip includes the same header for the product build.
//
// Conventions (reference: src/map_alignment/visualise.py:94-96,128-130): a rigid transform
// is a 4x4 row-major matrix applied as p' = T @ [p,1].  The device keeps coordinates
// relative to a centring origin o (the gold map's bounding-box centre): p_c = p - o.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace ma {

struct GridP {
  float ox, oy, oz;  // min corner of the grid in the centred frame
  float c, inv_c;    // cell edge and its reciprocal
  float delta;       // boundary slack (a few ulp of the largest coordinate)
  int nx, ny, nz;
};

struct DevState {
  double R[9];
  double t[3];       // centred frame: p'_c = R p_c + t
  double prev_rmse;
  double last_rmse;
  double last_W;
  int iter;
  int done;
  int converged;
  int degenerate;
  unsigned long long clk;   // the device's 100 MHz clock when the solve that produced this state finished (ma_stats::gpu_ms)
};

// A/B knobs of the product build (csrc/Makefile `variant`); the defaults are the measured optimum.
#ifndef MA_NN_BLOCK
#define MA_NN_BLOCK 256
#endif
#ifndef MA_XCD_SEG
#define MA_XCD_SEG 16
#endif

constexpr uint64_t kMaxCells = 1ull << 27;
constexpr int kSearchBlock = MA_NN_BLOCK;     // points per workgroup of the search kernel
constexpr uint32_t kXcdSegment = MA_XCD_SEG;  // chunks per XCD segment
constexpr int kMaxSchedGroups = 512;  // scheduling groups ranked by cost

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// world 4x4 (row-major; nullptr = identity) -> centred-frame state
inline void world_to_centred(const double* T, const double o[3], DevState& s) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  if (T) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[3 * r + c] = T[4 * r + c];
      t[r] = T[4 * r + 3];
    }
  }
  for (int i = 0; i < 9; ++i) s.R[i] = R[i];
  for (int r = 0; r < 3; ++r)
    s.t[r] = R[3 * r] * o[0] + R[3 * r + 1] * o[1] + R[3 * r + 2] * o[2] + t[r] - o[r];
}

inline void centred_to_world(const DevState& s, const double o[3], double* T) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[4 * r + c] = s.R[3 * r + c];
    T[4 * r + 3] = s.t[r] - (s.R[3 * r] * o[0] + s.R[3 * r + 1] * o[1] + s.R[3 * r + 2] * o[2]) + o[r];
  }
  T[12] = T[13] = T[14] = 0.0;
  T[15] = 1.0;
}

// centred-frame accumulators -> world-frame accumulators (p = p_c + o, q = q_c + o)
inline void acc_to_world(const double* a, const double o[3], double* out) {
  const double W = a[0];
  out[0] = W;
  for (int i = 0; i < 3; ++i) {
    out[1 + i] = a[1 + i] + W * o[i];
    out[4 + i] = a[4 + i] + W * o[i];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      out[7 + 3 * r + c] = a[7 + 3 * r + c] + a[1 + r] * o[c] + o[r] * a[4 + c] + W * o[r] * o[c];
  out[16] = a[16];
}

// ---- Exact (associative) sums: the 17 accumulators as 64-bit fixed-point integers -----------------------------------------
// Row a5's sums are formed as INTEGERS: every correspondence's 17 terms are rounded once, each to a multiple of a
// per-accumulator quantum 2^-e (round to nearest even: one fma against 1.5 * 2^52), and everything after that — the
// workgroup's reduction, the 19-way sum, the all-reduce of a sharded loop — is integer addition, which is associative.
// The sums therefore do not depend on how the source cloud is cut into chunks, workgroups, launches, shards or ranks,
// nor on the order in which a collective adds them: ANY partition over ANY number of GPUs returns the single GPU's
// bits (the term of a correspondence depends only on the point, its match and the transform).  62 bits hold the sum:
// the exponents are chosen from bounds on what a term can be, so that n_total terms cannot overflow —
//   |q_a| <= qb            the gold map's box (centred frame; the same on every rank: it derives from the gold map alone)
//   |p'_a| <= pb           gated run: a matched p' lies within the gate of its gold point, pb = qb + gate;
//                          ungated: p' = R p + t with |p| <= rs gives rs + |t0| for the first iteration, and every later
//                          transform maps the matched points' centroid onto their gold points' centroid (inside the
//                          box): |p'| <= sqrt(3) qb + 2 rs
//   w <= wmax, d^2 <= gate^2 (gated) or 3 (pb + qb)^2
// — and they are the same on every rank of a sharded run once n_total, wmax and rs are the whole cloud's
// (ma_sum_bounds_get / _set; ma_icp_sharded agrees on them itself).  At 1M points, a 20 m map and a 0.10 m gate the
// quantum of S is 2^-34 m^2 per correspondence: finer than the rounding of an fp64 accumulation of the same sum.
constexpr int kSumW = 0, kSumP = 1, kSumQ = 2, kSumS = 3, kSumE = 4;
struct SumScale { int e[5]; };        // term * 2^e[group] is rounded to an integer; groups: W, s_p, s_q, S, E
struct SumBounds {
  double n_total;   // correspondences that can contribute, all ranks together (source slots)
  double wmax;      // largest weight (1 without weights)
  double rs;        // largest |p| of a source point (centred frame); only ungated runs look at it
};
// accumulator index (0..16) -> group
constexpr int sum_group(int k) { return k == 0 ? kSumW : (k < 4 ? kSumP : (k < 7 ? kSumQ : (k < 16 ? kSumS : kSumE))); }
// smallest E with x <= 2^E (x > 0; powers of two get one more: harmless headroom)
inline int ceil_log2(double x) {
  int E = 0;
  (void)std::frexp(x, &E);
  return E;
}
inline SumScale sum_scale(const SumBounds& b, double qb, double gate, double t0_norm) {
  const double n = std::max(b.n_total, 1.0), wmax = b.wmax > 0.0 && std::isfinite(b.wmax) ? b.wmax : 1.0;
  qb = std::max(qb, 1e-30);
  double pb, d2max;
  if (std::isfinite(gate)) {
    pb = qb + gate * 1.001 + 1e-6;
    d2max = gate * gate * 1.001 + 1e-12;
  } else {
    const double rs = std::max(b.rs, 0.0);
    pb = std::max(rs + t0_norm, 1.7320508075688772 * qb + 2.0 * rs) * 1.001 + 1e-6;
    d2max = 3.0 * (pb + qb) * (pb + qb);
  }
  // a term may use min(50, 61 - L) bits: each term stays inside the fma's exact window (|k| < 2^51) and n terms inside 2^62
  const int budget = std::min(50, 61 - ceil_log2(n));
  const double term[5] = {wmax, wmax * pb, wmax * qb, wmax * pb * qb, wmax * d2max};
  SumScale s;
  for (int g = 0; g < 5; ++g) s.e[g] = std::max(-900, std::min(900, budget - ceil_log2(term[g])));
  return s;
}
constexpr double kSumMagic = 6755399441055744.0;              // 1.5 * 2^52: fma(x, 2^e, magic) holds rne(x 2^e) in its low mantissa bits
constexpr unsigned long long kSumMagicBits = 0x4338000000000000ull;
// host model of what a correspondence term becomes, and back (tests; ma_icp_step's conversion)
inline long long sum_quantize(double term, int e) {
  const double r = std::fma(term, std::ldexp(1.0, e), kSumMagic);
  unsigned long long bits;
  std::memcpy(&bits, &r, 8);
  return (long long)(bits - kSumMagicBits);
}
inline double sum_to_double(long long v, int e) { return std::ldexp((double)v, -e); }

// ---- Peer-to-peer exchange of the sums (row e, comm.hip / kernels_solve.hip k_reduce_exchange_solve) ----------------------
// Every rank owns an array of rows [kXSlots][kXMaxRanks][kXRowWords] of 8-byte words in its own device memory, mapped into
// every peer.  Iteration number `seq` (counted over the life of the communicator, the same on every rank) uses slot
// seq % kXSlots; rank r WRITES its 17 sums into row [slot][r] of every rank's array (its own included), and reads the g rows
// of that slot in ITS OWN array.  A word is self-validating: the sums are 62-bit integers (host_math.h "Exact sums"), and
// the two bits above carry a tag that changes with every reuse of the slot — a naturally aligned 8-byte store is atomic,
// so a word is either the one this iteration wrote or the one of kXSlots iterations ago (different tag), never a mixture.
// No fence, no checksum, no ordering between words: the writer's stores are all in flight together, the reader polls.
constexpr int kXSlots = 4, kXMaxRanks = 16, kXRowWords = 32;      // a row: 17 words in use, 256 bytes apart
constexpr size_t kXBoundsOffset = (size_t)kXSlots * kXMaxRanks * kXRowWords;       // behind the rows: [kXMaxRanks][4] words for the bounds agreement
constexpr size_t kXWords = kXBoundsOffset + (size_t)kXMaxRanks * 4;
constexpr unsigned xword_tag(unsigned long long seq) { return (unsigned)((seq / (unsigned)kXSlots) + 1ull) & 3u; }      // (a zeroed buffer reads as tag 0: never the first generation's)
constexpr unsigned long long xword_encode(long long v, unsigned long long seq) {
  return ((unsigned long long)v & ((1ull << 62) - 1ull)) | ((unsigned long long)xword_tag(seq) << 62);
}
constexpr long long xword_value(unsigned long long w) { return (long long)(w << 2) >> 2; }      // sign-extend bit 61

// strict '<' acceptance against this value implements the inclusive gate d^2 <= gate^2
inline float open_gate2(double gate) {
  if (!std::isfinite(gate)) return std::numeric_limits<float>::infinity();
  const float g2 = (float)(gate * gate);
  return std::nextafterf(g2, std::numeric_limits<float>::infinity());
}

inline GridP make_grid(const float lo[3], const float hi[3], const double o[3], float c) {
  GridP g{};
  g.ox = (float)((double)lo[0] - o[0]);
  g.oy = (float)((double)lo[1] - o[1]);
  g.oz = (float)((double)lo[2] - o[2]);
  g.c = c;
  g.inv_c = 1.0f / c;
  const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1], ez = (double)hi[2] - lo[2];
  g.nx = (int)std::floor(ex / c) + 1;
  g.ny = (int)std::floor(ey / c) + 1;
  g.nz = (int)std::floor(ez / c) + 1;
  double maxabs = 0;
  for (int a = 0; a < 3; ++a) maxabs = std::max({maxabs, std::fabs((double)lo[a] - o[a]), std::fabs((double)hi[a] - o[a])});
  const float ulp = std::nextafterf((float)maxabs, INFINITY) - (float)maxabs;
  g.delta = 8.0f * ulp + 1e-6f * c;
  return g;
}

inline float clamp_cell_to_budget(const float lo[3], const float hi[3], float c, uint64_t max_cells) {
  for (int it = 0; it < 64; ++it) {
    const double nx = std::floor(((double)hi[0] - lo[0]) / c) + 1, ny = std::floor(((double)hi[1] - lo[1]) / c) + 1,
                 nz = std::floor(((double)hi[2] - lo[2]) / c) + 1;
    const double cells = nx * ny * nz;
    if (cells <= (double)max_cells && nx < 2e9 && ny < 2e9 && nz < 2e9) return c;
    // grow the edge by what the count calls for (not by a fixed factor: the result should sit
    // just under the budget, not up to 2x below it)
    c *= (float)std::min(2.0, std::max(1.004, std::cbrt(cells / (double)max_cells) * 1.002));
  }
  return c;
}

// ---- Robust grid box (stray points far from the scene) --------------------------------------------------------------------
// The grid is a dense table over a box.  A handful of stray points far from the scene (reflections through windows,
// sensor noise: RTAB-Map exports carry them) blow the bounding box up, the table budget then forces a coarse cell edge, and
// a search that examines hundreds of candidates per cell runs 20-10 000 times longer (tools/outlier_probe.py: 20 stray
// points over 5 / 50 times the scene's extent: 2.2 -> 43 ms / 7.9 s per alignment).  When the budget binds, the build
// therefore histograms the coordinates per axis and puts the grid over the box that holds the bulk of the points: per axis
// the central (1 - 2 kRobustTail) of the mass, widened by kRobustWiden of its width on either side (a uniformly filled
// axis [0, 1] gives [0.05, 0.95] -> [-0.175, 1.175]: a clean scene keeps its own box), cut to the true bounds.  Points
// outside it are STORED IN THE BOUNDARY CELLS (cell_of clamps its indices) with their true coordinates, and the search
// (kernels_search.hip, HALO instantiations) treats a boundary cell as reaching outwards without limit: gaps to it drop
// their outward term, cell indices are clamped for query and content alike (clamping both ends of a distance to an
// interval never lengthens it), distances are formed from the true coordinates — exact for every query.  The grid then
// goes over the bounds of the points INSIDE that box.  Any box is correct — the choice only decides how fast the search runs.
constexpr int kAxisBins = 128;                 // histogram bins per axis and pass (+ one count below, one above the range)
constexpr double kRobustTail = 0.05, kRobustWiden = 0.25;
constexpr double kRobustMaxOutside = 0.03;     // the box of the bulk is taken only if at most this share of the points lies outside it
// One axis: the bin-edge interval [a, b] of [lo, hi] that leaves at most `tail` of the n points on either side.
// h[0] = points below lo, h[1 .. bins] = the bins, h[bins + 1] = points above hi.
inline void central_interval(const uint64_t* h, int bins, double lo, double hi, uint64_t n, double tail, double* a, double* b) {
  const uint64_t allow = (uint64_t)std::floor(tail * (double)n);
  const double w = (hi - lo) / (double)bins;
  uint64_t cum = h[0];
  int first = 0;
  while (first < bins - 1 && cum + h[1 + first] <= allow) { cum += h[1 + first]; ++first; }
  cum = h[bins + 1];
  int last = bins - 1;
  while (last > first && cum + h[1 + last] <= allow) { cum += h[1 + last]; --last; }
  *a = lo + w * (double)first;
  *b = lo + w * (double)(last + 1);
}
// The box from the central intervals of the three axes (widened, cut to the true bounds); true if it is smaller than the
// true box along some axis by more than a twentieth — otherwise the true box stays (nothing to gain).
inline bool robust_box(const double ca[3], const double cb[3], const float lo[3], const float hi[3], float rlo[3], float rhi[3]) {
  bool smaller = false;
  for (int k = 0; k < 3; ++k) {
    const double w = std::max(cb[k] - ca[k], 0.0), pad = kRobustWiden * w + 1e-6;
    const double l = std::max((double)lo[k], ca[k] - pad), h = std::min((double)hi[k], cb[k] + pad);
    rlo[k] = (float)l;
    rhi[k] = (float)h;
    if (!(rhi[k] > rlo[k])) { rlo[k] = lo[k]; rhi[k] = hi[k]; }      // (a degenerate axis: all points in one plane)
    if (rlo[k] > lo[k]) rlo[k] = std::nextafterf(rlo[k], -INFINITY);
    if (rhi[k] < hi[k]) rhi[k] = std::nextafterf(rhi[k], INFINITY);
    if ((double)rhi[k] - rlo[k] < 0.95 * ((double)hi[k] - lo[k])) smaller = true;
  }
  return smaller;
}

// Launch geometry of the search kernel: one source point per thread, kSearchBlock points per
// chunk.  G (chunks, = workgroups) is a multiple of 8 * kXcdSegment (the XCD remap is a
// bijection) and of kXcdSegment << gshift, where 2^gshift segments form one scheduling group
// and there are at most kMaxSchedGroups groups (the surplus workgroups have no points).
inline uint32_t grid_for(uint32_t n, uint32_t* gshift_out = nullptr) {
  const uint32_t q = 8u * kXcdSegment;
  const uint32_t G0 = std::max<uint32_t>(q, (div_up(n, kSearchBlock) + q - 1u) / q * q);
  uint32_t gshift = 0, G = G0;
  for (;;) {
    const uint32_t unit = std::max<uint32_t>(q, kXcdSegment << gshift);   // both are powers of two
    G = (G0 + unit - 1u) / unit * unit;
    // rounding G up to the unit can add a group (67.1M points: 512 -> 513), so test AFTER it
    if (G / (kXcdSegment << gshift) <= (uint32_t)kMaxSchedGroups) break;
    ++gshift;
  }
  if (gshift_out) *gshift_out = gshift;
  return G;
}

// Number of scheduling groups the ranking workgroup of k_reduce_solve orders.
inline uint32_t sched_groups(uint32_t G, uint32_t gshift) { return G / (kXcdSegment << gshift); }

// Host model of the ranking workgroup: order[r] = the group with the r-th largest cost
// (ties: smaller index first).  Keys are padded with 0 up to a multiple of 4, exactly as the
// kernel pads its LDS array, so that the 16-byte compare loop sees every key.
inline void rank_groups_model(const uint32_t* cost, uint32_t ngroups, uint32_t* order) {
  uint32_t key[kMaxSchedGroups + 4] = {0};
  for (uint32_t g = 0; g < ngroups; ++g)
    key[g] = (std::min(cost[g], 0x7FFFFFu) << 9) | ((uint32_t)kMaxSchedGroups - 1u - g);
  const uint32_t n4 = (ngroups + 3u) / 4u;
  for (uint32_t g = 0; g < ngroups; ++g) {
    uint32_t rank = 0;
    for (uint32_t j = 0; j < 4u * n4; ++j) rank += key[j] > key[g] ? 1u : 0u;
    order[rank] = g;
  }
}

// N4: the voxels of a map's bounding box.  The voxel of a point is floor(p * inv_voxel) per axis; keys
// number the box's voxels z-major with indices relative to its first voxel, so the radix sort only looks at
// `key_bits` bits.  Returns false when an index or the voxel count does not fit 63 bits.
struct VoxelBox {
  long long q0[3];              // first voxel index per axis
  unsigned long long d[3];      // voxels per axis
};
inline bool voxel_box(const float lo[3], const float hi[3], float inv_voxel, VoxelBox& box, int& key_bits) {
  long double cells = 1.0L;
  for (int a = 0; a < 3; ++a) {
    if (!(std::fabs(lo[a] * inv_voxel) < 4.0e18f) || !(std::fabs(hi[a] * inv_voxel) < 4.0e18f)) return false;
    const long long q0 = (long long)std::floor(lo[a] * inv_voxel), q1 = (long long)std::floor(hi[a] * inv_voxel);
    if (q1 < q0) return false;
    box.q0[a] = q0;
    box.d[a] = (unsigned long long)(q1 - q0) + 1ull;
    cells *= (long double)box.d[a];
  }
  if (cells >= 9.0e18L) return false;
  const unsigned long long last = box.d[0] * box.d[1] * box.d[2] - 1ull;
  key_bits = 0;
  while (key_bits < 64 && (last >> key_bits) != 0ull) ++key_bits;
  key_bits = std::max(key_bits, 1);
  return true;
}

// fp64 clouds: georeferenced coordinates (any |coordinate| above `large`) lose centimetres in fp32, so an
// integer-metre shift to the box centre (half-to-even, as numpy.round) is taken out in fp64 first.
inline void choose_shift(const double lo[3], const double hi[3], double large, double shift[3]) {
  double big = 0.0;
  for (int a = 0; a < 3; ++a) big = std::max({big, std::fabs(lo[a]), std::fabs(hi[a])});
  for (int a = 0; a < 3; ++a) shift[a] = big > large ? std::nearbyint(0.5 * (lo[a] + hi[a])) : 0.0;
}

}  // namespace ma
