// kernels_build.hip — staging of clouds on the device (once per cloud; not part of the timed loop): bounds, the
// gold map's cell sort, cell table and distance-to-occupied table, the coarse levels, the source cloud's brick-major
// sort, fp64 staging, cloud moments (row N1).  Kernels first (anonymous namespace), then the host functions that
// launch them (namespace mai: ma_internal.h declares the ones other units call).
#include "ma_internal.h"

namespace {

// ---------------------------------------------------------------------------------------
// grid build (once per cloud; not part of the timed loop)
// ---------------------------------------------------------------------------------------
// Per-workgroup min / max / non-finite count of an n x 3 fp32 cloud.
// STRIDE floats per point: 3 for raw clouds, 4 for the float4 buffers.
// `within`: only the points inside this box count towards the bounds (the rest is counted in slot 7) — the bounds of a
// cloud's bulk once a box around it is known (host_math.h "Robust grid box"); (-inf, +inf) for the plain bounds.
struct BoundsBox { float lo[3], hi[3]; };
template <int STRIDE>
__global__ __launch_bounds__(kBlock) void k_minmax(const float* __restrict__ xyz, uint32_t n,
                                                   float* __restrict__ out /*[grid][8]*/, const int holes_ok, const BoundsBox within) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float bad = 0.f, outside = 0.f;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float x = xyz[STRIDE * (size_t)i], y = xyz[STRIDE * (size_t)i + 1], z = xyz[STRIDE * (size_t)i + 2];
    if (holes_ok && __float_as_uint(x) == kHoleBits) continue;     // a hole of the in-place unprojection (float4 buffers)
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      if (x < within.lo[0] || x > within.hi[0] || y < within.lo[1] || y > within.hi[1] || z < within.lo[2] || z > within.hi[2]) {
        outside += 1.f;
        continue;
      }
      lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
      lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
      lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    } else {
      bad += 1.f;
    }
  }
  __shared__ float red[kBlock / kWave][8];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, kWave));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, kWave));
    }
    bad += __shfl_down(bad, off, kWave);
    outside += __shfl_down(outside, off, kWave);
  }
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { red[wv][a] = lo[a]; red[wv][3 + a] = hi[a]; }
    red[wv][6] = bad; red[wv][7] = outside;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int k = threadIdx.x;
    float v = red[0][k];
    for (int w = 1; w < kBlock / kWave; ++w) {
      const float u = red[w][k];
      v = (k < 3) ? fminf(v, u) : (k < 6 ? fmaxf(v, u) : v + u);
    }
    out[blockIdx.x * 8 + k] = v;
  }
}

// What a HALO grid's boundary cells hold beyond their own interval (ma_internal.h HaloBox): per axis, the coordinate nearest
// to the grid's box T among the points outside T that lie beyond it ALONG THAT AXIS — a point outside T may lie inside T's
// range along two of the three axes.  out[b][0..2] = largest coordinate below T per axis (-inf: none), [3..5] = smallest
// above it (+inf: none).
__global__ __launch_bounds__(kBlock) void k_beyond(const float* __restrict__ xyz, uint32_t n, const BoundsBox T,
                                                   float* __restrict__ out /*[grid][8]*/) {
  float below[3] = {-INFINITY, -INFINITY, -INFINITY}, above[3] = {INFINITY, INFINITY, INFINITY};
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[3 * (size_t)i + a];
      if (v < T.lo[a]) below[a] = fmaxf(below[a], v);
      if (v > T.hi[a]) above[a] = fminf(above[a], v);
    }
  }
  __shared__ float red[kBlock / kWave][8];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      below[a] = fmaxf(below[a], __shfl_down(below[a], off, kWave));
      above[a] = fminf(above[a], __shfl_down(above[a], off, kWave));
    }
  }
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { red[wv][a] = below[a]; red[wv][3 + a] = above[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = red[0][k];
    for (int w = 1; w < kBlock / kWave; ++w) v = k < 3 ? fmaxf(v, red[w][k]) : fminf(v, red[w][k]);
    out[blockIdx.x * 8 + k] = v;
  }
}

// fp64 clouds (what Open3D / NumPy hand over by default): bounds in fp64 and the conversion to
// the engine's fp32 frame on the device, so that the host never walks the arrays.
__global__ __launch_bounds__(kBlock) void k_minmax_f64(const double* __restrict__ xyz, uint32_t n,
                                                       double* __restrict__ out /*[grid][8]*/) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double bad = 0.0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      lo[0] = fmin(lo[0], x); hi[0] = fmax(hi[0], x);
      lo[1] = fmin(lo[1], y); hi[1] = fmax(hi[1], y);
      lo[2] = fmin(lo[2], z); hi[2] = fmax(hi[2], z);
    } else {
      bad += 1.0;
    }
  }
  __shared__ double red[kBlock / kWave][8];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], off, kWave));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], off, kWave));
    }
    bad += __shfl_down(bad, off, kWave);
  }
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { red[wv][a] = lo[a]; red[wv][3 + a] = hi[a]; }
    red[wv][6] = bad; red[wv][7] = 0.0;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int k = threadIdx.x;
    double v = red[0][k];
    for (int w = 1; w < kBlock / kWave; ++w) {
      const double u = red[w][k];
      v = (k < 3) ? fmin(v, u) : (k < 6 ? fmax(v, u) : v + u);
    }
    out[blockIdx.x * 8 + k] = v;
  }
}

// raw[i] = (float)(xyz[i] - shift[axis]): the subtraction in fp64, one rounding to fp32
__global__ __launch_bounds__(kBlock) void k_f64_to_raw(const double* __restrict__ xyz, uint64_t n3, double s0, double s1,
                                                       double s2, float* __restrict__ raw) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n3) return;
  const uint32_t a = (uint32_t)(i % 3u);
  raw[i] = (float)(xyz[i] - (a == 0 ? s0 : (a == 1 ? s1 : s2)));
}

// key = cell id, val = original index; also the per-cell histogram (integer atomics:
// the counts are order-independent, hence deterministic).
#ifndef MA_BRICK_SHIFT
#define MA_BRICK_SHIFT 4
#endif
constexpr uint32_t kBrickShift = MA_BRICK_SHIFT;   // bricks of 2^shift cells per axis

// brick != 0: the key is brick-major — 16 x 16 x 16-cell bricks in x-fastest order, cells
// x-fastest inside a brick — so that 256 consecutive points of the sorted SOURCE cloud form
// a compact patch (small LDS tile) instead of a 2 m long line of cells.
__global__ __launch_bounds__(kBlock) void k_cell_keys(const float* __restrict__ xyz, uint32_t n,
                                                      double o0, double o1, double o2, GridP g, int brick,
                                                      uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals,
                                                      uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float x = (float)((double)xyz[3 * (size_t)i] - o0);
  const float y = (float)((double)xyz[3 * (size_t)i + 1] - o1);
  const float z = (float)((double)xyz[3 * (size_t)i + 2] - o2);
  uint32_t c = cell_of(x, y, z, g);
  if (brick) {
    constexpr uint32_t B = kBrickShift, M = (1u << B) - 1u;
    const uint32_t ix = c % (uint32_t)g.nx, iy = (c / (uint32_t)g.nx) % (uint32_t)g.ny, iz = c / ((uint32_t)g.nx * g.ny);
    const uint32_t nbx = ((uint32_t)g.nx + M) >> B, nby = ((uint32_t)g.ny + M) >> B;
    c = ((((iz >> B) * nby + (iy >> B)) * nbx + (ix >> B)) << (3 * B)) | ((iz & M) << (2 * B)) | ((iy & M) << B) | (ix & M);
  }
  keys[i] = c;
  vals[i] = i;
  if (counts) atomicAdd(&counts[c], 1u);
}

// Occupancy probes of the cell-edge / grid-phase search (12 per gold map): only WHETHER a cell
// holds a point matters, so a probe marks a byte per cell with plain stores (idempotent: no
// atomics, a quarter of the table bytes of a u32 histogram) and counts the marks 16 at a time.
// (The histogram kernels took 55 + 28 us per probe at 1M points, a third of the grid build.)
constexpr int kMaxProbes = 16;
constexpr uint32_t kProbeBlocks = 256;         // workgroups of k_count_marks per probe
struct ProbeSet {
  GridP g[kMaxProbes];
  uint32_t n16[kMaxProbes];     // 16-byte words of each probe's table
  uint32_t stride16;            // distance between two tables, in 16-byte words
  int count;
};

// ALL probes of a round trip in one launch (round 3; round 2 launched one pair of kernels per probe, 12.5 us each,
// for kernels that take 3-7 us of work): every point is read and centred once and marked in `count` tables.
__global__ __launch_bounds__(kBlock) void k_mark_cells(const float* __restrict__ xyz, uint32_t n, double o0, double o1,
                                                       double o2, const ProbeSet P, uint8_t* __restrict__ mark) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float x = (float)((double)xyz[3 * (size_t)i] - o0);
  const float y = (float)((double)xyz[3 * (size_t)i + 1] - o1);
  const float z = (float)((double)xyz[3 * (size_t)i + 2] - o2);
#pragma unroll 1
  for (int k = 0; k < P.count; ++k) mark[(size_t)k * P.stride16 * 16u + cell_of(x, y, z, P.g[k])] = 1;
}

// marks: 0 / 1 bytes, each table padded with zeros to a multiple of 16.  Counts them AND clears them (the tables
// are all-zero again afterwards: no memset, neither between probes nor between calls), blockIdx.y = probe, with
// kProbeBlocks large workgroups per probe.  Every workgroup stores its count with a plain store — straight into the
// pinned mailbox in host memory (see Mailbox below) — and the host adds them up: no atomic (round 2: one per
// workgroup on one address, and those queue), no cleared counter, no device-to-host copy.
__global__ __launch_bounds__(kBlock) void k_count_marks(uint4* __restrict__ mark16, const ProbeSet P,
                                                        unsigned long long* __restrict__ out_per_block) {
  const uint32_t n16 = P.n16[blockIdx.y];
  uint4* __restrict__ tab = mark16 + (size_t)blockIdx.y * P.stride16;
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n16; i += gridDim.x * kBlock) {
    const uint4 v = tab[i];
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    if (v.x | v.y | v.z | v.w) tab[i] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
  __shared__ uint32_t part[kBlock / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += part[w];
    out_per_block[(size_t)blockIdx.y * kProbeBlocks + blockIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------------------
// Stable LSD radix sort of (cell key, index) pairs for clouds of up to a few million points
// ---------------------------------------------------------------------------------------
// The library sorts (rocPRIM behind hipCUB) are built for large inputs: below 2^20 items a merge sort of 21 launches
// (150-190 us for the 1M-point benchmark cloud), above it Onesweep, whose passes are chained workgroup to workgroup by
// decoupled look-back — 26-29 us per pass at 1M points, where the chain's latency, not bandwidth, sets the time
// (`profiles/r03_variants_ab.txt`, staging).  With a few hundred workgroups the classic three-kernel pass is faster:
//   k_sort_hist     every workgroup counts the digits of its tile (2048 keys) -> hist[digit][workgroup]
//   k_sort_rows     one WAVEFRONT per digit: exclusive prefix along its row of workgroups, and the digit's total
//   k_sort_scatter  every workgroup: digit starts from the 256 totals (LDS scan), stable rank of every key inside the
//                   tile (match-by-ballot per wavefront, wavefront bases from LDS), and the scatter
// 8-bit digits, ceil(bits / 8) passes, no atomics on global memory, no spinning.  Stable: the order inside a tile is
// wavefront-major, then item, then lane — which is also its order in memory.
constexpr int kSortItems = 8;                         // keys per thread
constexpr int kSortTile = kBlock * kSortItems;        // keys per workgroup
constexpr int kSortBins = 256;
static_assert(kSortBins == kBlock, "one thread per digit in the base computation");

// (KEY: uint32_t — cell keys of the grid builds — or unsigned long long — the voxel keys of the map builder, row N4.)
template <typename KEY>
__global__ __launch_bounds__(kBlock) void k_sort_hist(const KEY* __restrict__ keys, const uint32_t n, const int shift,
                                                      const uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kSortBins];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + (uint32_t)j * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (kSortBins - 1)], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void k_sort_rows(uint32_t* __restrict__ hist, const uint32_t nblocks,
                                                      uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t bin = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  uint32_t* __restrict__ row = hist + (size_t)bin * nblocks;
  uint32_t run = 0u;
  for (uint32_t c0 = 0; c0 < nblocks; c0 += kWave) {
    const uint32_t i = c0 + (uint32_t)lane;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const uint32_t u = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += u;
    }
    if (i < nblocks) row[i] = run + incl - v;
    run += __shfl(incl, kWave - 1, kWave);
  }
  if (lane == 0) totals[bin] = run;
}

template <typename KEY>
__global__ __launch_bounds__(kBlock) void k_sort_scatter(const KEY* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                         const uint32_t n, const int shift, const uint32_t nblocks,
                                                         const uint32_t* __restrict__ row_prefix, const uint32_t* __restrict__ totals,
                                                         KEY* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  constexpr int kNW = kBlock / kWave;
  __shared__ uint32_t cnt[kNW][kSortBins];      // per-wavefront digit counts, then per-wavefront output bases
  __shared__ uint32_t wave_tot[kNW];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < kNW; ++k) cnt[k][threadIdx.x] = 0u;
  // digit starts: exclusive scan of the 256 totals (thread = digit)
  const uint32_t tot = totals[threadIdx.x];
  uint32_t incl = tot;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += u;
  }
  if (lane == kWave - 1) wave_tot[w] = incl;
  __syncthreads();
  uint32_t digit_start = incl - tot;
  for (int k = 0; k < w; ++k) digit_start += wave_tot[k];

  // stable ranks inside the wavefront's segment of the tile
  const uint32_t seg = blockIdx.x * (uint32_t)kSortTile + (uint32_t)w * (uint32_t)(kWave * kSortItems);
  const unsigned long long lt = (1ull << lane) - 1ull;
  KEY key[kSortItems];
  uint32_t rank[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = seg + (uint32_t)j * kWave + (uint32_t)lane;
    const bool valid = i < n;
    key[j] = valid ? keys_in[i] : (KEY)~(KEY)0;
    const uint32_t d = (uint32_t)(key[j] >> shift) & (kSortBins - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    uint32_t prior = 0u;
    if (valid) prior = cnt[w][d];
    __builtin_amdgcn_wave_barrier();            // every lane has read the count before the group's last lane replaces it
    rank[j] = prior + (uint32_t)__popcll(peers & lt);
    if (valid && (peers >> lane) == 1ull) cnt[w][d] = prior + (uint32_t)__popcll(peers);    // highest lane of the group
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {   // thread = digit: where each wavefront's keys of this digit go
    uint32_t run = digit_start + row_prefix[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int k = 0; k < kNW; ++k) {
      const uint32_t c = cnt[k][threadIdx.x];
      cnt[k][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = seg + (uint32_t)j * kWave + (uint32_t)lane;
    if (i < n) {
      const uint32_t pos = cnt[w][(uint32_t)(key[j] >> shift) & (kSortBins - 1)] + rank[j];
      keys_out[pos] = key[j];
      vals_out[pos] = vals_in[i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Exclusive prefix sum of 32-bit words (scan_exclusive_u32): tile sums, tile offsets, tiles
// ---------------------------------------------------------------------------------------
// A thread owns kSortItems CONSECUTIVE words of its workgroup's tile (two 16-byte loads), so its local prefix is a short
// serial chain; the threads' totals are scanned by wavefront (shuffles) and across the four wavefronts through LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(const uint32_t mine, uint32_t* sh_wave /*[kBlock / kWave]*/, uint32_t& block_total) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += u;
  }
  if (lane == kWave - 1) sh_wave[w] = incl;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kBlock / kWave; ++k) {
    const uint32_t t = sh_wave[k];
    before += k < w ? t : 0u;
    total += t;
  }
  block_total = total;
  return before + incl - mine;
}
__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(const uint32_t* __restrict__ in, const uint32_t n, uint32_t* __restrict__ tile) {
  const uint32_t first = blockIdx.x * (uint32_t)kSortTile + threadIdx.x * (uint32_t)kSortItems;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) s += first + j < n ? in[first + j] : 0u;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
  __shared__ uint32_t part[kBlock / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += part[w];
    tile[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kBlock) void k_scan_offsets(uint32_t* __restrict__ tile, const uint32_t ntiles) {
  __shared__ uint32_t sh_wave[kBlock / kWave];
  uint32_t run = 0;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += kBlock) {
    const uint32_t i = t0 + threadIdx.x;
    const uint32_t v = i < ntiles ? tile[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, sh_wave, total);
    if (i < ntiles) tile[i] = run + ex;
    run += total;
    __syncthreads();                    // sh_wave is reused by the next trip
  }
}
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const uint32_t* __restrict__ in, const uint32_t n, const uint32_t* __restrict__ tile,
                                                       uint32_t* __restrict__ out) {
  __shared__ uint32_t sh_wave[kBlock / kWave];
  const uint32_t first = blockIdx.x * (uint32_t)kSortTile + threadIdx.x * (uint32_t)kSortItems;
  uint32_t v[kSortItems], s = 0;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) { v[j] = first + j < n ? in[first + j] : 0u; s += v[j]; }
  uint32_t total;
  uint32_t run = tile[blockIdx.x] + block_exclusive_scan(s, sh_wave, total);
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    if (first + j < n) out[first + j] = run;
    run += v[j];
  }
}

// Coarser levels are built on demand from the level-0 copy (already centred, float4):
// cell ids of level l (+ histogram), and the gather that keeps .w = position in level 0.
__global__ __launch_bounds__(kBlock) void k_cell_keys4(const float4* __restrict__ pts, uint32_t n, GridP g,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const uint32_t c = cell_of(p.x, p.y, p.z, g);
  keys[i] = c;
  vals[i] = i;
  if (counts) atomicAdd(&counts[c], 1u);
}

__global__ __launch_bounds__(kBlock) void k_gather4(const float4* __restrict__ gold0, const uint32_t* __restrict__ perm,
                                                    uint32_t n, float4* __restrict__ out) {
  const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n) return;
  const uint32_t j = perm[pos];
  float4 v = gold0[j];
  v.w = __uint_as_float(j);
  out[pos] = v;
}

// Cell table straight from the sorted keys (round 3).  Round 2 counted the points of every cell with one integer
// atomic per point beside the key computation (k_cell_keys: 54 us with, 8 us without, for 1M points) and summed the
// counts over all cells (hipCUB, 28 us for 3.8M cells); the first form of this round scattered the run ends into a
// zeroed table and max-scanned it (memset 6 + 5 + 28 us).  Neither pass over the cells is needed: cstart[c] = number
// of points with key < c, so the run that STARTS at sorted position i (keys[i] != keys[i - 1]) settles every cell in
// (keys[i - 1], keys[i]] — its own cell and the empty ones in front of it — to the value i; item n (one past the
// last point) settles the cells behind the last occupied one, padding included, to n.  A gap of a few cells is
// written by its own thread; a longer one (row ends, empty slabs: thousands of cells) by the whole wavefront, 64
// cells per store; a HUGE one (empty layers of the box: 10^5 cells and more — one wavefront filling it alone took
// 26 of this kernel's 28 us) is put on a short list and filled by a whole grid (k_fill_gaps).  Reads 4 MB of keys,
// writes the table once.  The number of occupied cells (ma_grid_info: informational) is the number of run starts:
// one count per workgroup.
constexpr uint32_t kGapShort = 4, kGapHuge = 1024, kGapListCap = 4096;
struct GapList {
  uint32_t count;                 // zero between builds (k_fill_gaps' successor, k_sum_u32, clears it)
  uint32_t pad[3];
  uint32_t first[kGapListCap], len[kGapListCap], value[kGapListCap];
};
__global__ __launch_bounds__(kBlock) void k_cell_starts(const uint32_t* __restrict__ sorted_keys, const uint32_t n,
                                                        const uint32_t table_len, uint32_t* __restrict__ cstart,
                                                        uint32_t* __restrict__ starts_per_block, GapList* __restrict__ gaps) {
  // workgroups are dealt round-robin over the 8 XCDs: give every XCD ONE contiguous eighth of the sorted array, so that
  // the cache lines of the table (neighbouring runs write neighbouring cells) are completed inside one L2 instead of
  // being written back in pieces by several (gridDim.x is a multiple of 8)
  const uint32_t blk = (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u;
  const uint32_t i = blk * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  bool start = false;
  uint32_t first = 0, len = 0;
  if (i < n) {
    const uint32_t k = sorted_keys[i];
    if (i == 0) { start = true; first = 0; len = k + 1u; }
    else {
      const uint32_t p = sorted_keys[i - 1];
      if (p != k) { start = true; first = p + 1u; len = k - p; }
    }
  } else if (i == n) {            // behind the last occupied cell
    first = sorted_keys[n - 1] + 1u;
    len = table_len - first;
  }
  if (len != 0u && len <= kGapShort) {
    for (uint32_t j = 0; j < len; ++j) cstart[first + j] = i;
  }
  if (len > kGapHuge) {           // rare (a few per grid): one atomic each is no queue
    const uint32_t slot = atomicAdd(&gaps->count, 1u);
    if (slot < kGapListCap) {
      gaps->first[slot] = first; gaps->len[slot] = len; gaps->value[slot] = i;
      len = 0u;                   // handed over
    }                             // (list full: filled here after all, as a wide gap)
  }
  unsigned long long wide = __ballot(len > kGapShort);
  while (wide) {
    const int src = __ffsll((long long)wide) - 1;
    wide &= wide - 1ull;
    const uint32_t f = __shfl(first, src, kWave), l = __shfl(len, src, kWave), v = __shfl(i, src, kWave);
    // four stores per trip: a lone wavefront pays ~25 ns per trip of this loop, and the longest chains of gaps set
    // the kernel's time
    for (uint32_t j = (uint32_t)lane; j < l; j += 4u * kWave) {
      cstart[f + j] = v;
      if (j + kWave < l) cstart[f + j + kWave] = v;
      if (j + 2u * kWave < l) cstart[f + j + 2u * kWave] = v;
      if (j + 3u * kWave < l) cstart[f + j + 3u * kWave] = v;
    }
  }
  const uint32_t c = (uint32_t)__popcll(__ballot(start));
  __shared__ uint32_t part[kBlock / kWave];
  if (lane == 0) part[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += part[w];
    starts_per_block[blk] = t;
  }
}

// the listed gaps, every one by the whole grid
__global__ __launch_bounds__(kBlock) void k_fill_gaps(const GapList* __restrict__ gaps, uint32_t* __restrict__ cstart) {
  const uint32_t count = min(gaps->count, kGapListCap);
  for (uint32_t g = 0; g < count; ++g) {
    const uint32_t f = gaps->first[g], l = gaps->len[g], v = gaps->value[g];
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < l; j += gridDim.x * kBlock) cstart[f + j] = v;
  }
}

// sum of n words -> one 64-bit word (ONE workgroup; the destination is the pinned mailbox): the run-start counts of
// k_cell_starts = the number of occupied cells.  Also leaves the gap list empty for the next build.
__global__ __launch_bounds__(kBlock) void k_sum_u32(const uint32_t* __restrict__ v, const uint32_t n,
                                                    unsigned long long* __restrict__ out, uint32_t* __restrict__ clear) {
  if (threadIdx.x == 0 && clear) *clear = 0u;
  unsigned long long c = 0;
  for (uint32_t i = threadIdx.x; i < n; i += kBlock) c += v[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
  __shared__ unsigned long long part[kBlock / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += part[w];
    *out = t;
  }
}

// First and second moments of a stored cloud (centred float4; holes skipped): per workgroup {count, x, y, z, xx, xy,
// xz, yy, yz, zz} in fp64, written to the pinned mailbox; the host adds the rows.  What the coarse stage's principal
// axes need (N1) — 30 us on the device against 5-6 ms per million points for the same sums in NumPy.
constexpr int kMomLen = 10;
constexpr uint32_t kMomBlocks = 256;
// `within`: only the points inside this box count (the box of a cloud's bulk where one was taken — a few stray points a
// hundred scene-widths away would own the covariance; (-inf, +inf) otherwise).
__global__ __launch_bounds__(kBlock) void k_moments(const float4* __restrict__ pts, const uint32_t n, double* __restrict__ out,
                                                    const BoundsBox within) {
  double a[kMomLen];
#pragma unroll
  for (int k = 0; k < kMomLen; ++k) a[k] = 0.0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
    if (p.x < within.lo[0] || p.x > within.hi[0] || p.y < within.lo[1] || p.y > within.hi[1] || p.z < within.lo[2] || p.z > within.hi[2]) continue;
    const double x = p.x, y = p.y, z = p.z;
    a[0] += 1.0; a[1] += x; a[2] += y; a[3] += z;
    a[4] += x * x; a[5] += x * y; a[6] += x * z; a[7] += y * y; a[8] += y * z; a[9] += z * z;
  }
  __shared__ double part[kBlock / kWave][kMomLen];
#pragma unroll
  for (int k = 0; k < kMomLen; ++k) {
    const double v = wave_sum(a[k]);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) part[threadIdx.x / kWave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kMomLen) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) v += part[w][threadIdx.x];
    out[(size_t)blockIdx.x * kMomLen + threadIdx.x] = v;
  }
}

// number of weights that are not finite and >= 0 (one integer atomic per workgroup)
__global__ __launch_bounds__(kBlock) void k_count_bad_weights(const float* __restrict__ w, uint32_t n,
                                                              unsigned long long* __restrict__ out) {
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float v = w[i];
    c += !(v >= 0.f) || !isfinite(v);
  }
  const unsigned long long any = __ballot(c != 0u);
  __shared__ uint32_t part[kBlock / kWave];
  uint32_t tot = c;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) tot += __shfl_down(tot, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = any ? tot : 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; ++k) t += part[k];
    if (t) atomicAdd(out, (unsigned long long)t);
  }
  // the largest weight (row a5's sum bounds, host_math.h): non-negative floats order like their bit patterns
  float m = 0.f;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) m = fmaxf(m, w[i]);      // (fmaxf drops NaNs: those are counted above)
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(out + 1), __float_as_uint(m));
}

// sorted[pos] = centred(xyz[perm[pos]]).  w_mode 0: .w = weight (source cloud);
// 1: .w = pos (level 0 of the gold map: its own position).  (The coarser levels are gathered
// from level 0 by k_gather4.)
__global__ __launch_bounds__(kBlock) void k_gather(const float* __restrict__ xyz,
                                                   const float* __restrict__ weights,
                                                   const uint32_t* __restrict__ perm, uint32_t n,
                                                   double o0, double o1, double o2, int w_mode,
                                                   const uint32_t* __restrict__ lut,
                                                   float4* __restrict__ out) {
  const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n) return;
  const uint32_t i = perm ? perm[pos] : pos;
  float4 v;
  v.x = (float)((double)xyz[3 * (size_t)i] - o0);
  v.y = (float)((double)xyz[3 * (size_t)i + 1] - o1);
  v.z = (float)((double)xyz[3 * (size_t)i + 2] - o2);
  if (w_mode == 0) v.w = weights ? weights[i] : 1.0f;
  else v.w = __uint_as_float(pos);
  out[pos] = v;
}

// Chebyshev distance (in cells, capped at kOccCap) from every level-0 cell to the nearest
// occupied one.  A query whose cell has distance k cannot have a gold point closer than
// (k-1)*c: one byte read settles "nothing within the gate" for points of the scan that the
// gold map does not cover, instead of a full ring walk.
// Built in three separable passes (round 2; round 1 relaxed the 26-neighbourhood eight times,
// 27 reads per cell and pass: 0.38 of the 1.5 ms grid build):
//   d(p) = min_q max(|dx|,|dy|,|dz|) = min_z' max(|dz|, min_y' max(|dy|, min_x' |dx|))
// because max(u, min_v f(v)) = min_v max(u, f(v)).  Pass x measures the distance along the
// row from the cell table itself; passes y and z each combine at most 2 x 8 neighbours along
// their axis and stop as soon as the offset reaches the best value found (near the surfaces,
// i.e. almost everywhere it matters, after one or two taps).
constexpr int kOccCap = 9;
constexpr int kOccReach = kOccCap - 1;             // farthest offset looked at
constexpr int kOccXOut = kBlock - 2 * kOccReach;   // cells a workgroup of k_occ_x writes

// Pass x (round 3): occupancy comes from the cell table with ONE coalesced read per cell (round 2: up to 32 per
// cell — most of the grid is empty and an empty cell looked at all 16 neighbours, two table words each: 37 us for
// 3.8M cells).  Thread t of a workgroup looks at cell base - 8 + t, the wavefronts' ballots form a 256-bit
// occupancy map in LDS, and the distance to the nearest set bit inside the cell's own row is two bit scans.
__global__ __launch_bounds__(kBlock) void k_occ_x(const uint32_t* __restrict__ cstart, const int nx, const uint32_t ncells,
                                                  uint8_t* __restrict__ d) {
  __shared__ unsigned long long bits[kBlock / kWave + 1];
  const long long cell = (long long)blockIdx.x * kOccXOut - kOccReach + (long long)threadIdx.x;
  bool occ = false;
  if (cell >= 0 && cell < (long long)ncells) occ = cstart[cell + 1] != cstart[cell];
  const unsigned long long m = __ballot(occ);
  if ((threadIdx.x & (kWave - 1)) == 0) bits[threadIdx.x / kWave] = m;
  if (threadIdx.x == 0) bits[kBlock / kWave] = 0ull;
  __syncthreads();
  if (threadIdx.x < kOccReach || threadIdx.x >= kBlock - kOccReach || cell >= (long long)ncells) return;
  const int lo = (int)threadIdx.x - kOccReach, w = lo >> 6, sh = lo & 63;
  unsigned long long win = bits[w] >> sh;
  if (sh) win |= bits[w + 1] << (64 - sh);
  // bit kOccReach of the window is the cell itself; keep what lies inside its row
  const int x = (int)((uint32_t)cell % (uint32_t)nx);
  const int left = min(x, kOccReach), right = min(nx - 1 - x, kOccReach);
  uint32_t wb = (uint32_t)win & ((2u << (kOccReach + right)) - 1u) & ~((1u << (kOccReach - left)) - 1u);
  int best = kOccCap;
  if (wb) {
    const uint32_t lower = wb & ((2u << kOccReach) - 1u), upper = wb >> kOccReach;
    const int dl = lower ? kOccReach - (31 - __clz((int)lower)) : kOccCap;
    const int dr = upper ? __ffs((int)upper) - 1 : kOccCap;
    best = min(dl, dr);
  }
  d[cell] = (uint8_t)best;
}

// out(p) = min over offsets k along the axis (stride in cells, n cells long) of max(in(p + k), |k|).
// Round 3: FOUR cells per thread — consecutive along x inside one row, so that they share their position along the
// axis — read with one (unaligned) 32-bit load per tap and combined as two pairs of 16-bit lanes (v_pk_max_u16 /
// v_pk_min_u16): a quarter of the load instructions of the byte-per-thread form (40 us per pass for 3.8M cells,
// bound by the L1's one-request-per-lane rate, not by bytes).  Tables are padded by 16 bytes: the last group of a
// row reads (never writes) up to 3 bytes beyond it.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__global__ __launch_bounds__(kBlock) void k_occ_axis(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                     const uint32_t nrows, const uint32_t nx, const uint32_t stride,
                                                     const uint32_t rows_per_step, const int n) {
  // a "row" is a run of nx cells along x; rows_per_step rows make one step along the axis (1 for y, ny for z)
  const uint32_t per_row = (nx + 3u) / 4u;
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t row = t / per_row, x = (t - row * per_row) * 4u;
  if (row >= nrows) return;
  const int pos = (int)((row / rows_per_step) % (uint32_t)n);
  const size_t i = (size_t)row * nx + x;
  const uint32_t w = ld_u32_unaligned(in + i);
  const uint32_t kLo = 0x00FF00FFu;
  us2 be = __builtin_bit_cast(us2, w & kLo), bo = __builtin_bit_cast(us2, (w >> 8) & kLo);
  auto widest = [&]() -> int {
    const us2 m = __builtin_elementwise_max(be, bo);
    return max((int)m.x, (int)m.y);
  };
  int kmax = widest();
  for (int k = 1; k < kmax; ++k) {       // an offset of `best` or more cannot improve on best
    const us2 ks = {(unsigned short)k, (unsigned short)k};
    if (pos - k >= 0) {
      const uint32_t u = ld_u32_unaligned(in + i - (size_t)k * stride);
      be = __builtin_elementwise_min(be, __builtin_elementwise_max(__builtin_bit_cast(us2, u & kLo), ks));
      bo = __builtin_elementwise_min(bo, __builtin_elementwise_max(__builtin_bit_cast(us2, (u >> 8) & kLo), ks));
    }
    if (pos + k < n) {
      const uint32_t u = ld_u32_unaligned(in + i + (size_t)k * stride);
      be = __builtin_elementwise_min(be, __builtin_elementwise_max(__builtin_bit_cast(us2, u & kLo), ks));
      bo = __builtin_elementwise_min(bo, __builtin_elementwise_max(__builtin_bit_cast(us2, (u >> 8) & kLo), ks));
    }
    kmax = widest();
  }
  const uint32_t r = __builtin_bit_cast(uint32_t, be) | (__builtin_bit_cast(uint32_t, bo) << 8);
  if (x + 4u <= nx) {
    __builtin_memcpy(out + i, &r, 4);
  } else {
    for (uint32_t j = 0; x + j < nx; ++j) out[i + j] = (uint8_t)(r >> (8u * j));
  }
}

static_assert(kMaxProbes * kProbeBlocks * 8 <= kMailWord - kMailProbe, "probe region of the mailbox");

// Per-axis coordinate histograms of a cloud over the ranges [lo, lo + bins w] (host_math.h "Robust grid box"): slot 0 counts
// what lies below a range, slot bins + 1 what lies above.  Integer LDS atomics, one partial histogram per workgroup
// written straight into the pinned mailbox (the bounds region: the bounds have been read by then) and added up by the
// host.  Only run when a cloud's bounding box is too large for the cell edge its density asks for — never for a clean scene.
constexpr int kHistBlocks = 40, kHistSlots = ma::kAxisBins + 2;
static_assert((size_t)kHistBlocks * 3 * kHistSlots * 4 <= kMailProbe, "bounds region of the mailbox");
struct AxisRange { double lo[3], inv_w[3]; };
__global__ __launch_bounds__(kBlock) void k_axis_hist(const float* __restrict__ xyz, const uint32_t n, const uint32_t step,
                                                      const AxisRange r, uint32_t* __restrict__ out /*[grid][3][kHistSlots]*/) {
  __shared__ uint32_t h[3][kHistSlots];
  for (int k = threadIdx.x; k < 3 * kHistSlots; k += kBlock) (&h[0][0])[k] = 0u;
  __syncthreads();
  for (uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * step; i < n; i += (uint64_t)gridDim.x * kBlock * step) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double t = ((double)xyz[3 * (size_t)i + a] - r.lo[a]) * r.inv_w[a];
      const int slot = t < 0.0 ? 0 : (t >= (double)ma::kAxisBins ? ma::kAxisBins + 1 : 1 + (int)t);
      atomicAdd(&h[a][slot], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 3 * kHistSlots; k += kBlock) out[(size_t)blockIdx.x * 3 * kHistSlots + k] = (&h[0][0])[k];
}

}  // namespace

namespace mai {

// bounding box (and finiteness check) of a device-resident n x 3 cloud
int cloud_bounds(ma_ctx* ctx, const float* d_xyz, uint32_t n, float lo[3], float hi[3], int stride, bool holes_ok, const Lane* lane,
                 const float* within_lo, const float* within_hi, double* n_outside) {
  const Lane ml = main_lane(ctx);
  const Lane& L = lane ? *lane : ml;
  const int nb = std::min<uint32_t>(div_up(n, kBlock), 1024);
  float* h = reinterpret_cast<float*>(L.mail);          // [nb][8], written by the kernel
  BoundsBox within{{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
  if (within_lo && within_hi) for (int a = 0; a < 3; ++a) { within.lo[a] = within_lo[a]; within.hi[a] = within_hi[a]; }
  if (stride == 4) hipLaunchKernelGGL(k_minmax<4>, dim3(nb), dim3(kBlock), 0, L.stream, d_xyz, n, h, holes_ok ? 1 : 0, within);
  else hipLaunchKernelGGL(k_minmax<3>, dim3(nb), dim3(kBlock), 0, L.stream, d_xyz, n, h, 0, within);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(L.stream));
  double bad = 0, outside = 0;
  for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  for (int b = 0; b < nb; ++b) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], h[(size_t)b * 8 + a]);
      hi[a] = std::max(hi[a], h[(size_t)b * 8 + 3 + a]);
    }
    bad += h[(size_t)b * 8 + 6];
    outside += h[(size_t)b * 8 + 7];
  }
  if (n_outside) *n_outside = outside;
  if (bad > 0) return ctx->fail(MA_ERR_INVALID, "cloud has non-finite coordinates (filter them first, as the reference's read_point_cloud(remove_nan_points=True, remove_infinite_points=True) does)");
  return MA_OK;
}
// k_beyond over a raw cloud: below[a] / above[a] as described there
static int nearest_beyond(ma_ctx* ctx, const float* d_xyz, uint32_t n, const float lo[3], const float hi[3], float below[3], float above[3]) {
  const int nb = std::min<uint32_t>(div_up(n, kBlock), 1024);
  float* h = reinterpret_cast<float*>(ctx->mailbox);
  BoundsBox T{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
  hipLaunchKernelGGL(k_beyond, dim3(nb), dim3(kBlock), 0, ctx->stream, d_xyz, n, T, h);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int a = 0; a < 3; ++a) { below[a] = -INFINITY; above[a] = INFINITY; }
  for (int b = 0; b < nb; ++b)
    for (int a = 0; a < 3; ++a) {
      below[a] = std::max(below[a], h[(size_t)b * 8 + a]);
      above[a] = std::min(above[a], h[(size_t)b * 8 + 3 + a]);
    }
  return MA_OK;
}

// Central interval of the cloud's mass along every axis (host_math.h "Robust grid box"): histogram passes over the true
// bounds, zooming into the interval found while it is much narrower than the range it was found in (a stray point a
// thousand scene-widths away leaves the whole scene in one bin of the first pass).  <= 4 passes, one synchronisation each.
static int axis_central(ma_ctx* ctx, const float* d_xyz, uint32_t n, const float lo[3], const float hi[3], const Lane& L,
                        double ca[3], double cb[3]) {
  double rlo[3], rhi[3];
  for (int a = 0; a < 3; ++a) { rlo[a] = lo[a]; rhi[a] = hi[a]; ca[a] = lo[a]; cb[a] = hi[a]; }
  uint32_t* part = reinterpret_cast<uint32_t*>(L.mail);
  const uint32_t step = std::max<uint32_t>(1u, n / (2u << 20));      // every step-th point: a sample of <= ~2M points says where the bulk is
  for (int pass = 0; pass < 4; ++pass) {
    AxisRange r{};
    for (int a = 0; a < 3; ++a) {
      const double w = (rhi[a] - rlo[a]) / (double)ma::kAxisBins;
      r.lo[a] = rlo[a];
      r.inv_w[a] = w > 0.0 ? 1.0 / w : 0.0;       // (a flat axis: every point in slot 1)
    }
    hipLaunchKernelGGL(k_axis_hist, dim3(kHistBlocks), dim3(kBlock), 0, L.stream, d_xyz, n, step, r, part);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(L.stream));
    bool zoom = false;
    for (int a = 0; a < 3; ++a) {
      uint64_t h[kHistSlots] = {0};
      for (int b = 0; b < kHistBlocks; ++b)
        for (int k = 0; k < kHistSlots; ++k) h[k] += part[((size_t)b * 3 + a) * kHistSlots + k];
      if (!(rhi[a] > rlo[a])) continue;
      uint64_t sampled = 0;
      for (int k = 0; k < kHistSlots; ++k) sampled += h[k];
      ma::central_interval(h, ma::kAxisBins, rlo[a], rhi[a], sampled, ma::kRobustTail, &ca[a], &cb[a]);
      if (cb[a] - ca[a] < (rhi[a] - rlo[a]) / 8.0 && cb[a] - ca[a] > 1e-9 * (std::fabs(ca[a]) + std::fabs(cb[a]) + 1e-30)) {
        rlo[a] = ca[a]; rhi[a] = cb[a];
        zoom = true;
      }
    }
    if (!zoom) break;
  }
  return MA_OK;
}
// Stable radix sort of (key, index) pairs, 32- or 64-bit keys: the three-kernel passes above.  Two-call pattern:
// tmp == nullptr asks for the temporary bytes.  The inputs are left intact.
// (Round 4 handed clouds above 4M points to rocPRIM's Onesweep: with the scans below that was the last use of a library
// primitive in the product; round 5 measures the own passes there too — profiles/r05_variants_ab.txt — and drops it.)
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
template <typename KEY>
static hipError_t sort_pairs_impl(void* tmp, size_t& tmp_bytes, const KEY* keys_in, KEY* keys_out, const uint32_t* vals_in,
                                  uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream) {
  const uint32_t nblocks = div_up(n, (uint32_t)kSortTile);
  const size_t key_bytes = align256((size_t)n * sizeof(KEY)), val_bytes = align256((size_t)n * 4),
               hist_bytes = align256((size_t)kSortBins * nblocks * 4);
  const size_t need = key_bytes + val_bytes + hist_bytes + align256(kSortBins * 4);
  if (!tmp) { tmp_bytes = need; return hipSuccess; }
  if (tmp_bytes < need) return hipErrorInvalidValue;
  char* base = static_cast<char*>(tmp);
  KEY* t_keys = reinterpret_cast<KEY*>(base);
  uint32_t* t_vals = reinterpret_cast<uint32_t*>(base + key_bytes);
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + key_bytes + val_bytes);
  uint32_t* totals = reinterpret_cast<uint32_t*>(base + key_bytes + val_bytes + hist_bytes);
  const int passes = std::max(1, (bits + 7) / 8);
  const KEY* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;       // the last pass writes the caller's output; alternate backwards from there
    KEY* dst_k = to_out ? keys_out : t_keys;
    uint32_t* dst_v = to_out ? vals_out : t_vals;
    hipLaunchKernelGGL(k_sort_hist<KEY>, dim3(nblocks), dim3(kBlock), 0, stream, src_k, n, 8 * p, nblocks, hist);
    hipLaunchKernelGGL(k_sort_rows, dim3(kSortBins / (kBlock / kWave)), dim3(kBlock), 0, stream, hist, nblocks, totals);
    hipLaunchKernelGGL(k_sort_scatter<KEY>, dim3(nblocks), dim3(kBlock), 0, stream, src_k, src_v, n, 8 * p, nblocks,
                       (const uint32_t*)hist, (const uint32_t*)totals, dst_k, dst_v);
    src_k = dst_k;
    src_v = dst_v;
  }
  return hipGetLastError();
}
hipError_t sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream) {
  return sort_pairs_impl<uint32_t>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, bits, stream);
}
hipError_t sort_pairs_u64(void* tmp, size_t& tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream) {
  return sort_pairs_impl<unsigned long long>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, bits, stream);
}

// Exclusive prefix sum of n 32-bit words (out[i] = in[0] + ... + in[i-1]; out may be in), three launches: every workgroup
// sums its tile of 2048 words, ONE workgroup turns the tile sums into tile offsets (a running total carried through its
// loop), every workgroup scans its tile from its offset.  tmp: one word per tile (scan_tmp_bytes).  What the rank map of a
// cloud with holes and the voxel numbering of the map builder need (both once per cloud, outside the loop).
size_t scan_tmp_bytes(uint32_t n) { return align256((size_t)div_up(std::max<uint32_t>(n, 1), (uint32_t)kSortTile) * 4); }
hipError_t scan_exclusive_u32(void* tmp, const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint32_t ntiles = div_up(n, (uint32_t)kSortTile);
  uint32_t* tile = static_cast<uint32_t*>(tmp);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(kBlock), 0, stream, in, n, tile);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(kBlock), 0, stream, tile, ntiles);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kBlock), 0, stream, in, n, (const uint32_t*)tile, out);
  return hipGetLastError();
}

// keys/vals for `n` raw points + optional histogram; then a stable radix sort by cell id.
// On return vals_b holds the permutation (sorted position -> original index).
static int sort_by_cell(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP& g, const double o[3],
                 uint32_t* d_counts, int brick = 0, const Lane* lane = nullptr) {
  const Lane ml = main_lane(ctx);
  const Lane& L = lane ? *lane : ml;
  HIPCHK(L.keys_a->ensure((size_t)n * 4));
  HIPCHK(L.keys_b->ensure((size_t)n * 4));
  HIPCHK(L.vals_a->ensure((size_t)n * 4));
  HIPCHK(L.vals_b->ensure((size_t)n * 4));
  hipLaunchKernelGGL(k_cell_keys, dim3(div_up(n, kBlock)), dim3(kBlock), 0, L.stream, d_xyz, n, o[0],
                     o[1], o[2], g, brick, L.keys_a->as<uint32_t>(), L.vals_a->as<uint32_t>(), d_counts);
  HIPCHK(hipGetLastError());
  const int bm = (1 << kBrickShift) - 1;
  const uint64_t ncells = brick ? ((uint64_t)((g.nx + bm) >> kBrickShift) * ((g.ny + bm) >> kBrickShift) *
                                   ((g.nz + bm) >> kBrickShift)) << (3 * kBrickShift)
                                : (uint64_t)g.nx * g.ny * g.nz;
  int bits = 1;
  while (bits < 32 && (1ull << bits) < ncells) ++bits;
  size_t tmp_bytes = 0;
  HIPCHK(sort_pairs_u32(nullptr, tmp_bytes, L.keys_a->as<uint32_t>(), L.keys_b->as<uint32_t>(), L.vals_a->as<uint32_t>(),
                        L.vals_b->as<uint32_t>(), n, bits, L.stream));
  HIPCHK(L.cub_tmp->ensure(tmp_bytes));
  HIPCHK(sort_pairs_u32(L.cub_tmp->p, tmp_bytes, L.keys_a->as<uint32_t>(), L.keys_b->as<uint32_t>(), L.vals_a->as<uint32_t>(),
                        L.vals_b->as<uint32_t>(), n, bits, L.stream));
  return MA_OK;
}

// byte marks of the raw gold points in each of `count` (<= 16) candidate grids -> numbers of occupied cells.
// One marking launch and one counting launch for all probes whose tables fit kProbeBudget together (all of them,
// unless the grids are huge), ONE synchronisation.  The mark tables live in their own buffer, which is all-zero
// whenever this function is not running (the counting kernel clears what it counts): it is cleared once after
// every (re)allocation and after a failed call, never per call.
constexpr size_t kProbeBudget = 256u << 20;
static int occupied_cells_multi(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP* grids, int count, const double o[3],
                         uint64_t* out) {
  if (count > kMaxProbes) return ctx->fail(MA_ERR_INVALID, "too many occupancy probes");
  uint64_t n16_max = 0;
  for (int k = 0; k < count; ++k) n16_max = std::max<uint64_t>(n16_max, ((uint64_t)grids[k].nx * grids[k].ny * grids[k].nz + 15) / 16);
  const size_t table = n16_max * 16;
  const int per_launch = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, kProbeBudget / std::max<size_t>(table, 1)));
  const size_t bytes = table * (size_t)per_launch;
  if (ctx->probe_marks.cap < bytes) ctx->probe_clean = 0;
  HIPCHK(ctx->probe_marks.ensure(bytes));
  if (ctx->probe_clean < bytes) {
    HIPCHK(hipMemsetAsync(ctx->probe_marks.p, 0, ctx->probe_marks.cap, ctx->stream));
  }
  ctx->probe_clean = 0;                    // until this call has come to its end
  unsigned long long* mail = reinterpret_cast<unsigned long long*>(ctx->mailbox + kMailProbe);   // [count][kProbeBlocks]
  for (int k0 = 0; k0 < count; k0 += per_launch) {
    ProbeSet P{};
    P.count = std::min(per_launch, count - k0);
    P.stride16 = (uint32_t)n16_max;
    for (int k = 0; k < P.count; ++k) {
      P.g[k] = grids[k0 + k];
      P.n16[k] = (uint32_t)(((uint64_t)P.g[k].nx * P.g[k].ny * P.g[k].nz + 15) / 16);
    }
    hipLaunchKernelGGL(k_mark_cells, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_xyz, n, o[0], o[1], o[2], P,
                       ctx->probe_marks.as<uint8_t>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_count_marks, dim3(kProbeBlocks, P.count), dim3(kBlock), 0, ctx->stream,
                       ctx->probe_marks.as<uint4>(), P, mail + (size_t)k0 * kProbeBlocks);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->probe_clean = ctx->probe_marks.cap;
  for (int k = 0; k < count; ++k) {
    uint64_t t = 0;
    for (uint32_t b = 0; b < kProbeBlocks; ++b) t += mail[(size_t)k * kProbeBlocks + b];
    out[k] = t;
  }
  return MA_OK;
}
static int occupied_cells(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP& g, const double o[3], uint64_t* out) {
  return occupied_cells_multi(ctx, d_xyz, n, &g, 1, o, out);
}
// Levels 1..3 (cell edges 2c, 4c, 8c over the same box) from the level-0 copy.  Stable sort by
// cell id of level l: inside a cell the points keep their level-0 order.
// Round 4: the same launches as level 0's build — cell keys without a histogram (no atomics), the own radix sort, the cell
// table straight from the sorted keys (k_cell_starts + k_fill_gaps) — instead of a histogram with one atomic per point, a
// memset and a library scan per level; and no synchronisation between the levels: everything is on one stream, the
// scratch a level reuses is free by the time its launches run.  (An ungated run's first iteration, which is where these
// levels are built, used to spend most of its 3.7 ms here.)
static int build_coarse_levels(ma_ctx* ctx) {
  const uint32_t n = ctx->m;
  HIPCHK(ctx->keys_a.ensure((size_t)n * 4));
  HIPCHK(ctx->keys_b.ensure((size_t)n * 4));
  HIPCHK(ctx->vals_a.ensure((size_t)n * 4));
  HIPCHK(ctx->vals_b.ensure((size_t)n * 4));
  const uint32_t nblk = (div_up(n + 1u, kBlock) + 7u) & ~7u;   // item n settles the cells behind the last occupied one; 8 | grid
  HIPCHK(ctx->occ_b.ensure((size_t)nblk * 4));
  if (!ctx->gap_list.p || ctx->gap_list_dirty) {
    HIPCHK(ctx->gap_list.ensure(sizeof(GapList)));
    HIPCHK(hipMemsetAsync(ctx->gap_list.p, 0, sizeof(GapList), ctx->stream));
  }
  for (int l = 1; l < kLevels; ++l) {
    const GridP g = make_grid(ctx->grid_lo, ctx->grid_hi, ctx->origin, ctx->grid.c * (float)(1 << l));
    const uint64_t ncells = (uint64_t)g.nx * g.ny * g.nz;
    HIPCHK(ctx->cstart[l].ensure((ncells + 8) * 4));
    hipLaunchKernelGGL(k_cell_keys4, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->gold[0].as<float4>(), n, g,
                       ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), (uint32_t*)nullptr);
    HIPCHK(hipGetLastError());
    int bits = 1;
    while (bits < 32 && (1ull << bits) < ncells) ++bits;
    size_t tmp_bytes = 0;
    HIPCHK(sort_pairs_u32(nullptr, tmp_bytes, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(), ctx->vals_a.as<uint32_t>(),
                          ctx->vals_b.as<uint32_t>(), n, bits, ctx->stream));
    HIPCHK(ctx->cub_tmp.ensure(tmp_bytes));
    HIPCHK(sort_pairs_u32(ctx->cub_tmp.p, tmp_bytes, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(), ctx->vals_a.as<uint32_t>(),
                          ctx->vals_b.as<uint32_t>(), n, bits, ctx->stream));
    ctx->gap_list_dirty = true;
    hipLaunchKernelGGL(k_cell_starts, dim3(nblk), dim3(kBlock), 0, ctx->stream, ctx->keys_b.as<uint32_t>(), n,
                       (uint32_t)(ncells + 8), ctx->cstart[l].as<uint32_t>(), ctx->occ_b.as<uint32_t>(),
                       ctx->gap_list.as<GapList>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_fill_gaps, dim3(256), dim3(kBlock), 0, ctx->stream, ctx->gap_list.as<GapList>(),
                       ctx->cstart[l].as<uint32_t>());
    HIPCHK(hipGetLastError());
    // (the run-start counts are not wanted here: k_sum_u32 runs for its other duty, clearing the gap list's counter;
    // its sum goes to a mailbox word nobody reads)
    hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->occ_b.as<uint32_t>(), nblk,
                       reinterpret_cast<unsigned long long*>(ctx->mailbox + kMailWord + 128), &ctx->gap_list.as<GapList>()->count);
    HIPCHK(hipGetLastError());
    ctx->gap_list_dirty = false;
    HIPCHK(ctx->gold[l].ensure(((size_t)n + 8) * sizeof(float4)));
    hipLaunchKernelGGL(k_gather4, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->gold[0].as<float4>(),
                       ctx->vals_b.as<uint32_t>(), n, ctx->gold[l].as<float4>());
    HIPCHK(hipGetLastError());
    LevelSet& L = ctx->levels;
    L.gold[l] = ctx->gold[l].as<float4>();
    L.cs[l] = ctx->cstart[l].as<uint32_t>();
    L.c[l] = g.c; L.inv_c[l] = g.inv_c;
    L.nx[l] = g.nx; L.ny[l] = g.ny; L.nz[l] = g.nz;
  }
  ctx->coarse_built = true;
  return MA_OK;
}

// A search sphere wider than 6 level-0 cells leaves the wavefront-cooperative level-0 search for
// the multi-level one (wave_search: kWalk): only runs gated that widely, or ungated, need levels 1..3.
int ensure_levels_for(ma_ctx* ctx, double max_corr_dist) {
  if (ctx->coarse_built || !ctx->have_gold) return MA_OK;
  const double reach = 6.0 * (double)ctx->grid.c;
  if (std::isfinite(max_corr_dist) && max_corr_dist <= reach) return MA_OK;
  return build_coarse_levels(ctx);
}

int build_gold_from_raw(ma_ctx* ctx, uint32_t n, float cell) {
  float lo[3], hi[3];
  int rc = cloud_bounds(ctx, ctx->raw.as<float>(), n, lo, hi);
  if (rc) return rc;
  MA_PT("gold: bounds read");
  const float true_lo[3] = {lo[0], lo[1], lo[2]}, true_hi[3] = {hi[0], hi[1], hi[2]};
  bool halo = false;            // the grid covers the bulk of the map only: points outside it sit in its boundary cells
  float wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0};      // ... per axis, the nearest coordinate of such a point below / above the grid's box (raw frame)
  float c = cell;
  // Two attempts at most: over the bounding box, and — when the table budget forced a coarser cell edge than the map's
  // density (or the caller) asked for — over the box that holds the bulk of the points (host_math.h "Robust grid box").
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool binding = false;         // the table budget decided the cell edge
    // centring origin: the box's centre, rounded to fp32 so that (x - o) is exact-ish
    for (int a = 0; a < 3; ++a) ctx->origin[a] = (double)(float)(0.5 * ((double)lo[a] + (double)hi[a]));

    c = cell;
    if (!(c > 0.f)) {
      // density-driven cell edge: aim at ~4.5 points per occupied cell (measured optimum of
      // table reads vs candidates on MI355X).  Scans are surfaces,
      // so occupancy grows ~ c^2; two histogram probes estimate the exponent.
      const double target = 4.5;
      // table budget of the density-driven choice: at most 32 cells (128 B of table) per point,
      // and at most 2^24 cells up to 4M points, 4 cells per point beyond, 2^25 from 8M points on
      // (the level-0 table has to stay in the Infinity Cache, but a cloud that dense pays for coarse
      // cells with candidates: 8M / 16M / 32M points, 50 iterations: 2.1 cm (2^24 cells) 20.9 / 59.4 /
      // 188.5 ms, 1.7 cm (31M cells) 19.9 / 51.8 / 153.0, 1.4 cm (55M cells) 24.6 / 61.4 / 177.6,
      // 1.1 cm (114M cells) - / 62.3 / 171.6 - tools/big_cell_sweep.py)
      // The relaxed cap stops at 1.7 cm where it would cross it: a search sphere wider than 6 cells leaves the
      // wavefront-cooperative search (wave_search: kWalk), and with the default 0.10 m gate that is every unmatched
      // query once c < 1.67 cm (16M points: 1.75 / 1.70 / 1.66 / 1.60 cm -> 51.8 / 52.0 / 61.6 / 60.2 ms).
      const uint64_t cap_hi = std::min<uint64_t>(1ull << 25, std::max<uint64_t>(1ull << 24, 4ull * n));
      uint64_t cell_cap = 1ull << 24;
      if (cap_hi > cell_cap) {
        const double ext0 = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-6});
        const float seed = (float)(ext0 / 4096.0);
        const float c_lo = clamp_cell_to_budget(lo, hi, seed, cap_hi), c_hi = clamp_cell_to_budget(lo, hi, seed, cell_cap);
        constexpr float kGateFloor = 0.017f;
        if (c_lo >= kGateFloor || c_hi <= kGateFloor) {
          cell_cap = cap_hi;                       // the whole relaxed range lies on one side of the floor
        } else {                                   // the range brackets it: the cell count of a 1.7 cm grid
          const GridP gf = make_grid(lo, hi, ctx->origin, kGateFloor);
          cell_cap = std::min<uint64_t>(cap_hi, std::max<uint64_t>(cell_cap, (uint64_t)gf.nx * gf.ny * gf.nz));
        }
      }
      const uint64_t auto_cells = std::min<uint64_t>(cell_cap, std::max<uint64_t>(1ull << 20, 32ull * n));
      const double ext = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-6});
      float c0 = clamp_cell_to_budget(lo, hi, (float)(ext / 256.0), auto_cells);
      uint64_t occ = 0;
      GridP g0 = make_grid(lo, hi, ctx->origin, c0);
      if ((rc = occupied_cells(ctx, ctx->raw.as<float>(), n, g0, ctx->origin, &occ))) return rc;
      double rho0 = (double)n / (double)std::max<uint64_t>(occ, 1);
      double expo = 2.0;
      float c1 = (float)(c0 * std::pow(target / rho0, 1.0 / expo));
      c1 = std::min(std::max(c1, c0 / 16.f), c0 * 16.f);
      float wanted = c1;
      c1 = clamp_cell_to_budget(lo, hi, c1, auto_cells);
      if (std::fabs(c1 / c0 - 1.f) > 0.05f) {
        GridP g1 = make_grid(lo, hi, ctx->origin, c1);
        if ((rc = occupied_cells(ctx, ctx->raw.as<float>(), n, g1, ctx->origin, &occ))) return rc;
        const double rho1 = (double)n / (double)std::max<uint64_t>(occ, 1);
        if (rho1 != rho0 && rho1 > 0 && rho0 > 0) {
          const double e = std::log(rho1 / rho0) / std::log((double)c1 / c0);
          if (e > 0.5 && e < 3.5) expo = e;
        }
        float c2 = (float)(c1 * std::pow(target / rho1, 1.0 / expo));
        c2 = std::min(std::max(c2, c1 / 4.f), c1 * 4.f);
        wanted = c2;
        c = clamp_cell_to_budget(lo, hi, c2, auto_cells);
      } else {
        c = c1;
      }
      binding = c > 1.15f * wanted;
    }
    {
      const float fits = clamp_cell_to_budget(lo, hi, c, kMaxCells);
      binding = binding || fits > 1.0001f * c;
      c = fits;
    }
    if (!(c > 0.f) || !std::isfinite(c)) return ctx->fail(MA_ERR_INVALID, "could not choose a cell edge");
    if (!binding || attempt == 1) break;
    // The budget, not the map, decided the cell edge: is the bounding box that of a few stray points?
    double ca[3], cb[3];
    float rlo[3], rhi[3];
    if ((rc = axis_central(ctx, ctx->raw.as<float>(), n, lo, hi, main_lane(ctx), ca, cb))) return rc;
    if (!ma::robust_box(ca, cb, lo, hi, rlo, rhi)) break;      // no: a large or dense scene, the box stays
    // yes: the grid goes over the bounds of the points INSIDE that box — the scene's own extent where the strays lie beyond it
    float tlo[3], thi[3];
    double n_out = 0;
    if ((rc = cloud_bounds(ctx, ctx->raw.as<float>(), n, tlo, thi, 3, false, nullptr, rlo, rhi, &n_out))) return rc;
    // (only for a FEW points out there — the boundary cells are not made to hold a tenth of the map: a scene whose density
    // varies that much between its parts keeps its bounding box and pays for it with candidates, tools/shape_probe.py)
    if (!(n_out > 0) || n_out > ma::kRobustMaxOutside * (double)n || !(thi[0] >= tlo[0])) break;
    for (int a = 0; a < 3; ++a) { lo[a] = tlo[a]; hi[a] = thi[a]; }
    // ... and, per axis, how near to that box the points outside it come along that axis (HaloBox)
    if ((rc = nearest_beyond(ctx, ctx->raw.as<float>(), n, lo, hi, wlo, whi))) return rc;
    halo = true;
    MA_PT("gold: robust box chosen");
  }
  MA_PT("gold: cell edge chosen");
  if (ctx->pair_handoff) ctx->pair_handoff->set(1, c);     // ma_set_pair: the scan's sort may start (origin and cell edge are final)

  // Grid phase: a surface that runs along a cell boundary (walls and floors of man-made
  // scenes are axis-aligned once the scan is levelled) puts its points into two layers of
  // cells, and every query near it then needs both.  Per axis, try the grid shifted by
  // 0, 1/4, 1/2 and 3/4 of a cell and keep the shift with the fewest occupied cells.
  if (cell <= 0.f) {
    // (round 3: the unshifted grid and the nine shifted ones — three per axis, each axis against the unshifted
    // grid — are probed back to back and read with ONE synchronisation: 3 host round trips per gold map instead
    // of 7.  Walls along different axes fill different layers of cells, so choosing the axes independently is a
    // good stand-in for the axis-after-axis descent of round 2; the phase only decides how fast the search runs,
    // never what it finds.)
    float best_lo[3] = {lo[0], lo[1], lo[2]};
    GridP trial_grid[10];
    float trial_lo[10];
    int trial_axis[10], nt = 0;
    trial_grid[nt] = make_grid(best_lo, hi, ctx->origin, c);
    trial_axis[nt] = -1; trial_lo[nt++] = 0.f;
    for (int a = 0; a < 3; ++a) {
      for (int ph = 1; ph < 4; ++ph) {
        float trial[3] = {lo[0], lo[1], lo[2]};
        trial[a] = lo[a] - 0.25f * (float)ph * c;
        const GridP g = make_grid(trial, hi, ctx->origin, c);
        if ((uint64_t)g.nx * g.ny * g.nz > kMaxCells) continue;
        trial_grid[nt] = g;
        trial_axis[nt] = a;
        trial_lo[nt++] = trial[a];
      }
    }
    uint64_t occ[10] = {0};
    if ((rc = occupied_cells_multi(ctx, ctx->raw.as<float>(), n, trial_grid, nt, ctx->origin, occ))) return rc;
    uint64_t best_axis_occ[3] = {occ[0], occ[0], occ[0]};
    for (int k = 1; k < nt; ++k) {
      const int a = trial_axis[k];
      if (occ[k] < best_axis_occ[a]) { best_axis_occ[a] = occ[k]; best_lo[a] = trial_lo[k]; }
    }
    for (int a = 0; a < 3; ++a) lo[a] = best_lo[a];
    MA_PT("gold: phase probes read");
  }

  // one grid per level: cell edges c, 2c, 4c, 8c over the same box
  LevelSet L{};
  GridP g0{};
  uint64_t ncells0 = 0;
  for (int l = 0; l < 1; ++l) {     // level 0 now; levels 1..3 in build_coarse_levels, on demand
    const float cl = c * (float)(1 << l);
    const GridP g = make_grid(lo, hi, ctx->origin, cl);
    const uint64_t ncells = (uint64_t)g.nx * g.ny * g.nz;
    if (l == 0) { g0 = g; ncells0 = ncells; }
    HIPCHK(ctx->cstart[l].ensure((ncells + 8) * 4));   // + padding for the 16-byte table reads
    // stable radix sort by cell id -> keys_b sorted, perm0 = permutation (sorted pos -> original index);
    // run starts of the sorted keys -> cell table (k_cell_starts)
    Lane gl = main_lane(ctx);
    gl.vals_b = &ctx->perm0;                            // the permutation is sorted straight into its final place
    if ((rc = sort_by_cell(ctx, ctx->raw.as<float>(), n, g, ctx->origin, nullptr, 0, &gl))) return rc;
    const uint32_t nblk = (div_up(n + 1u, kBlock) + 7u) & ~7u;   // item n settles the cells behind the last occupied one; 8 | grid
    HIPCHK(ctx->occ_b.ensure((size_t)nblk * 4));
    if (!ctx->gap_list.p || ctx->gap_list_dirty) {      // the list's counter is zero whenever no build runs (a build that
      HIPCHK(ctx->gap_list.ensure(sizeof(GapList)));    // failed half-way is the exception: cleared here)
      HIPCHK(hipMemsetAsync(ctx->gap_list.p, 0, sizeof(GapList), ctx->stream));
    }
    ctx->gap_list_dirty = true;
    hipLaunchKernelGGL(k_cell_starts, dim3(nblk), dim3(kBlock), 0, ctx->stream, ctx->keys_b.as<uint32_t>(), n,
                       (uint32_t)(ncells + 8), ctx->cstart[l].as<uint32_t>(), ctx->occ_b.as<uint32_t>(),
                       ctx->gap_list.as<GapList>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_fill_gaps, dim3(256), dim3(kBlock), 0, ctx->stream, ctx->gap_list.as<GapList>(),
                       ctx->cstart[l].as<uint32_t>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->occ_b.as<uint32_t>(), nblk,
                       reinterpret_cast<unsigned long long*>(ctx->mailbox + kMailWord), &ctx->gap_list.as<GapList>()->count);
    HIPCHK(hipGetLastError());
    ctx->gap_list_dirty = false;                        // the clearing launch is in the stream
    HIPCHK(ctx->gold[l].ensure(((size_t)n + 8) * sizeof(float4)));   // + 8 pad entries (level 0)
    hipLaunchKernelGGL(k_gather, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->raw.as<float>(),
                       (const float*)nullptr, ctx->perm0.as<uint32_t>(), n, ctx->origin[0], ctx->origin[1],
                       ctx->origin[2], 1, (const uint32_t*)nullptr, ctx->gold[l].as<float4>());
    HIPCHK(hipGetLastError());
    if (l == 0) {
      // eight far-away pad entries behind the last point: the candidate scan reads whole groups of
      // consecutive positions without clamping to the end of the array
      static const float4 kFar[8] = {{3e18f, 3e18f, 3e18f, 0.f}, {3e18f, 3e18f, 3e18f, 0.f}, {3e18f, 3e18f, 3e18f, 0.f},
                                     {3e18f, 3e18f, 3e18f, 0.f}, {3e18f, 3e18f, 3e18f, 0.f}, {3e18f, 3e18f, 3e18f, 0.f},
                                     {3e18f, 3e18f, 3e18f, 0.f}, {3e18f, 3e18f, 3e18f, 0.f}};
      HIPCHK(hipMemcpyAsync(ctx->gold[0].as<float4>() + n, kFar, sizeof(kFar), hipMemcpyHostToDevice, ctx->stream));
    }
    // (no synchronisation here: the number of occupied cells is read from the mailbox after the build's last one)
    L.gold[l] = ctx->gold[l].as<float4>();
    L.cs[l] = ctx->cstart[l].as<uint32_t>();
    L.c[l] = g.c; L.inv_c[l] = g.inv_c;
    L.nx[l] = g.nx; L.ny[l] = g.ny; L.nz[l] = g.nz;
    L.ox = g.ox; L.oy = g.oy; L.oz = g.oz;
    if (l == 0) L.delta = g.delta;
  }
  // Until the coarser levels exist their slots alias level 0: a search that reached them would
  // still be exact (every level holds every point), only slower.
  for (int l = 1; l < kLevels; ++l) {
    L.gold[l] = L.gold[0]; L.cs[l] = L.cs[0];
    L.c[l] = L.c[0]; L.inv_c[l] = L.inv_c[0];
    L.nx[l] = L.nx[0]; L.ny[l] = L.ny[0]; L.nz[l] = L.nz[0];
  }
  ctx->coarse_built = false;
  for (int a = 0; a < 3; ++a) { ctx->grid_lo[a] = lo[a]; ctx->grid_hi[a] = hi[a]; }
  {   // largest |coordinate| a gold point can have in the centred frame (row a5's sum bounds): of the TRUE bounds — a
      // point beyond the grid's box is a gold point like any other
    double qb = 0.0;
    for (int a = 0; a < 3; ++a) {
      qb = std::max({qb, std::fabs((double)std::min(lo[a], true_lo[a]) - ctx->origin[a]), std::fabs((double)std::max(hi[a], true_hi[a]) - ctx->origin[a])});
    }
    ctx->gold_qb = qb * (1.0 + 1e-6) + 1e-6;
  }
  ctx->grid_halo = halo;      // launch_nn picks the kernels that know what the boundary cells hold beyond their own interval
  for (int a = 0; a < 3; ++a) {
    // (centred like the stored points — the same subtraction, rounded to fp32 as k_gather rounds them — and moved one
    // ulp towards the grid: what the boundary cells hold beyond their interval lies at or beyond these values)
    ctx->halo_box.lo[a] = (halo && std::isfinite(wlo[a])) ? std::nextafterf((float)((double)wlo[a] - ctx->origin[a]), INFINITY) : -INFINITY;
    ctx->halo_box.hi[a] = (halo && std::isfinite(whi[a])) ? std::nextafterf((float)((double)whi[a] - ctx->origin[a]), -INFINITY) : INFINITY;
  }
  // distance-to-occupied table of level 0: three separable passes, x (from the cell table), y, z
  {
    const uint32_t nc = (uint32_t)ncells0;
    HIPCHK(ctx->occ_a.ensure((size_t)nc + 16));     // (+ padding: k_occ_axis reads whole 32-bit words)
    HIPCHK(ctx->occ_b.ensure((size_t)nc + 16));
    hipLaunchKernelGGL(k_occ_x, dim3(div_up(nc, (uint32_t)kOccXOut)), dim3(kBlock), 0, ctx->stream, ctx->cstart[0].as<uint32_t>(),
                       g0.nx, nc, ctx->occ_a.as<uint8_t>());
    HIPCHK(hipGetLastError());
    const uint32_t nrows = (uint32_t)g0.ny * (uint32_t)g0.nz, nthreads = nrows * (((uint32_t)g0.nx + 3u) / 4u);
    hipLaunchKernelGGL(k_occ_axis, dim3(div_up(nthreads, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->occ_a.as<uint8_t>(),
                       ctx->occ_b.as<uint8_t>(), nrows, (uint32_t)g0.nx, (uint32_t)g0.nx, 1u, g0.ny);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_occ_axis, dim3(div_up(nthreads, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->occ_b.as<uint8_t>(),
                       ctx->occ_a.as<uint8_t>(), nrows, (uint32_t)g0.nx, (uint32_t)g0.nx * (uint32_t)g0.ny, (uint32_t)g0.ny, g0.nz);
    HIPCHK(hipGetLastError());
    L.occ_dist = ctx->occ_a.as<uint8_t>();
  }
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  MA_PT("gold: build synchronised");
  ctx->noccupied = *reinterpret_cast<const unsigned long long*>(ctx->mailbox + kMailWord);
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->build_ms = ms;
  ctx->grid = g0;
  ctx->levels = L;
  ctx->ncells = ncells0;
  ctx->m = n;
  ctx->have_gold = true;
  return MA_OK;
}
// d_raw (default ctx->raw) holds the n x 3 fp32 source points (relative to the host-side shift, if any)
int build_source_from_raw(ma_ctx* ctx, uint32_t n, const float* weights, const float* d_raw, float gold_cell,
                                 const Lane* lane) {
  const Lane ml = main_lane(ctx);
  const Lane& L = lane ? *lane : ml;
  if (!d_raw) d_raw = ctx->raw.as<float>();
  if (!(gold_cell > 0.f)) gold_cell = ctx->grid.c;
  const float* d_w = nullptr;
  if (weights) {
    HIPCHK(ctx->wts.ensure((size_t)n * 4));
    HIPCHK(hipMemcpyAsync(ctx->wts.p, weights, (size_t)n * 4, hipMemcpyDefault, L.stream));
    d_w = ctx->wts.as<float>();
    HIPCHK(L.misc->ensure(64));
    HIPCHK(hipMemsetAsync(L.misc->p, 0, 16, L.stream));
    hipLaunchKernelGGL(k_count_bad_weights, dim3(std::min<uint32_t>(div_up(n, kBlock), 1024)), dim3(kBlock), 0,
                       L.stream, d_w, n, L.misc->as<unsigned long long>());
    HIPCHK(hipGetLastError());
    unsigned long long* bad = reinterpret_cast<unsigned long long*>(L.mail + kMailWord + 64);
    HIPCHK(hipMemcpyAsync(bad, L.misc->p, 16, hipMemcpyDeviceToHost, L.stream));     // into pinned memory: {bad weights, bits of the largest}
    HIPCHK(hipStreamSynchronize(L.stream));
    if (*bad) return ctx->fail(MA_ERR_INVALID, "weights must be finite and >= 0");
    {
      float wmax;
      const uint32_t bits = (uint32_t)bad[1];
      std::memcpy(&wmax, &bits, 4);
      ctx->src_wmax = wmax > 0.f ? (double)wmax : 1.0;
    }
  }
  float lo[3], hi[3];
  int rc = cloud_bounds(ctx, d_raw, n, lo, hi, 3, false, &L);
  if (rc) return rc;
  MA_PT("scan: bounds read");
  // sort in the cloud's own frame with the gold map's cell edge (coarsened to fit 2^27
  // cells), brick-major: 256 consecutive points then form a compact patch, so that a
  // workgroup's queries share one small block of the gold grid under any rigid motion.
  float c = clamp_cell_to_budget(lo, hi, gold_cell, kMaxCells);
  // The box the sort, the list planner and the lists' rounding allowance work with: the cloud's bounding box — or, when
  // that box is too large for the gold map's cell edge (stray points far from the scene: host_math.h "Robust grid box"),
  // the box of the bulk of the points.  Points outside it get the keys of its boundary bricks (cell_of clamps) and
  // keep no candidate list (k_nn_accum, list-writing launch); they are queries like any other.
  float blo[3] = {lo[0], lo[1], lo[2]}, bhi[3] = {hi[0], hi[1], hi[2]};
  bool bulk = false;
  if (c > 1.0001f * gold_cell) {
    double ca[3], cb[3];
    float rlo[3], rhi[3];
    if ((rc = axis_central(ctx, d_raw, n, lo, hi, L, ca, cb))) return rc;
    if (ma::robust_box(ca, cb, lo, hi, rlo, rhi)) {
      float tlo[3], thi[3];
      double n_out = 0;
      if ((rc = cloud_bounds(ctx, d_raw, n, tlo, thi, 3, false, &L, rlo, rhi, &n_out))) return rc;
      if (n_out > 0 && n_out <= ma::kRobustMaxOutside * (double)n && thi[0] >= tlo[0]) {
        bulk = true;
        for (int a = 0; a < 3; ++a) { blo[a] = tlo[a]; bhi[a] = thi[a]; }
        c = clamp_cell_to_budget(blo, bhi, gold_cell, kMaxCells);
      }
    }
  }
  const GridP g = make_grid(blo, bhi, ctx->origin, c);
  {
    double r2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double l = (double)blo[a] - ctx->origin[a], h = (double)bhi[a] - ctx->origin[a], pad = 1e-4 + 1e-6 * (std::fabs(l) + std::fabs(h));
      ctx->src_lo[a] = (float)(l - pad);
      ctx->src_hi[a] = (float)(h + pad);
      r2 += std::max(l * l, h * h);
      // (the sums' bound on |p| is that of EVERY point: the true bounds)
      ctx->src_amax = std::max({ctx->src_amax, std::fabs((double)lo[a] - ctx->origin[a]), std::fabs((double)hi[a] - ctx->origin[a])});
    }
    ctx->src_rs = (float)(std::sqrt(r2) * 1.001 + 1e-4);
    ctx->src_box_pending = false;
    ctx->src_bulk_box = bulk;
  }
  HIPCHK(ctx->src.ensure((size_t)n * sizeof(float4)));
  HIPCHK(ctx->src_perm.ensure((size_t)n * 4));
  Lane sl = L;
  sl.vals_b = &ctx->src_perm;                           // the permutation is sorted straight into its final place
  if ((rc = sort_by_cell(ctx, d_raw, n, g, ctx->origin, nullptr, /*brick=*/1, &sl))) return rc;
  hipLaunchKernelGGL(k_gather, dim3(div_up(n, kBlock)), dim3(kBlock), 0, L.stream, d_raw, d_w,
                     ctx->src_perm.as<uint32_t>(), n, ctx->origin[0], ctx->origin[1], ctx->origin[2], 0,
                     (const uint32_t*)nullptr, ctx->src.as<float4>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(L.stream));
  MA_PT("scan: build synchronised");
  ctx->n = n;
  ctx->n_pts = n;
  ctx->src_sorted = true;
  ctx->src_holes = false;
  ctx->src_map_ready = true;
  ctx->have_src = true;
  return MA_OK;
}

// fp64 staging: upload (host or device memory), bounds in fp64; `shift` is chosen here when
// choose_shift (gold map) and applied as given otherwise (source cloud)
int stage_f64(ma_ctx* ctx, const double* xyz, uint32_t n, bool choose_shift, double shift[3]) {
  HIPCHK(ctx->raw64.ensure((size_t)n * 24));
  HIPCHK(ctx->raw.ensure((size_t)n * 12));
  HIPCHK(hipMemcpyAsync(ctx->raw64.p, xyz, (size_t)n * 24, hipMemcpyDefault, ctx->stream));
  if (choose_shift) {
    const int nb = std::min<uint32_t>(div_up(n, kBlock), 1024);
    static_assert(1024 * 8 * sizeof(double) <= kMailProbe, "bounds region of the mailbox");
    double* h = reinterpret_cast<double*>(ctx->mailbox);      // [nb][8], written by the kernel
    hipLaunchKernelGGL(k_minmax_f64, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->raw64.as<double>(), n, h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, bad = 0;
    for (int b = 0; b < nb; ++b) {
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], h[(size_t)b * 8 + a]); hi[a] = std::max(hi[a], h[(size_t)b * 8 + 3 + a]); }
      bad += h[(size_t)b * 8 + 6];
    }
    if (bad > 0) return ctx->fail(MA_ERR_INVALID, "cloud has non-finite coordinates (filter them first, as the reference's read_point_cloud(remove_nan_points=True, remove_infinite_points=True) does)");
    ma::choose_shift(lo, hi, MA_LARGE_COORDINATE_M, shift);   // integer-metre shift for georeferenced coordinates
  }
  const uint64_t n3 = 3ull * n;
  hipLaunchKernelGGL(k_f64_to_raw, dim3((uint32_t)((n3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                     ctx->raw64.as<double>(), n3, shift[0], shift[1], shift[2], ctx->raw.as<float>());
  HIPCHK(hipGetLastError());
  return MA_OK;
}

hipError_t launch_f64_to_raw(const double* xyz, uint64_t n3, const double shift[3], float* raw, hipStream_t stream) {
  hipLaunchKernelGGL(k_f64_to_raw, dim3((uint32_t)((n3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, xyz, n3, shift[0],
                     shift[1], shift[2], raw);
  return hipGetLastError();
}

}  // namespace mai

extern "C" {

int ma_cloud_moments(ma_ctx* ctx, int which, double out[13]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return MA_ERR_INVALID;
  if (which != 0 && which != 1) return ctx->fail(MA_ERR_INVALID, "which: 0 = gold map, 1 = source cloud");
  if (which == 0 ? !ctx->have_gold : !ctx->have_src) return ctx->fail(MA_ERR_STATE, which == 0 ? "gold map not set" : "source cloud not set");
  HIPCHK(hipSetDevice(ctx->device));
  const float4* pts = which == 0 ? ctx->gold[0].as<float4>() : ctx->src.as<float4>();
  const uint32_t n = which == 0 ? ctx->m : ctx->n;
  const uint32_t nb = std::min<uint32_t>(div_up(n, kBlock), kMomBlocks);
  static_assert(kMomBlocks * kMomLen * sizeof(double) <= kMailProbe, "bounds region of the mailbox");
  double* rows = reinterpret_cast<double*>(ctx->mailbox);
  // the moments of the cloud's BULK where its grid / its sort took the box of the bulk (host_math.h "Robust grid box")
  BoundsBox within{{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
  if (which == 0 && ctx->grid_halo) {
    for (int a = 0; a < 3; ++a) {
      within.lo[a] = (float)((double)ctx->grid_lo[a] - ctx->origin[a] - (double)ctx->grid.c);
      within.hi[a] = (float)((double)ctx->grid_hi[a] - ctx->origin[a] + (double)ctx->grid.c);
    }
  } else if (which == 1 && ctx->src_bulk_box) {
    for (int a = 0; a < 3; ++a) { within.lo[a] = ctx->src_lo[a]; within.hi[a] = ctx->src_hi[a]; }
  }
  hipLaunchKernelGGL(k_moments, dim3(nb), dim3(kBlock), 0, ctx->stream, pts, n, rows, within);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double a[kMomLen] = {0};
  for (uint32_t b = 0; b < nb; ++b)
    for (int k = 0; k < kMomLen; ++k) a[k] += rows[(size_t)b * kMomLen + k];
  if (!(a[0] >= 1.0)) return ctx->fail(MA_ERR_EMPTY, "the cloud holds no valid point");
  const double cnt = a[0], mx = a[1] / cnt, my = a[2] / cnt, mz = a[3] / cnt, den = cnt > 1.0 ? cnt - 1.0 : 1.0;
  out[0] = cnt;
  out[1] = ctx->origin[0] + mx; out[2] = ctx->origin[1] + my; out[3] = ctx->origin[2] + mz;
  const double cxx = (a[4] - cnt * mx * mx) / den, cxy = (a[5] - cnt * mx * my) / den, cxz = (a[6] - cnt * mx * mz) / den;
  const double cyy = (a[7] - cnt * my * my) / den, cyz = (a[8] - cnt * my * mz) / den, czz = (a[9] - cnt * mz * mz) / den;
  out[4] = cxx; out[5] = cxy; out[6] = cxz; out[7] = cxy; out[8] = cyy; out[9] = cyz; out[10] = cxz; out[11] = cyz; out[12] = czz;
  return MA_OK;
  });
}

}  // extern "C"
