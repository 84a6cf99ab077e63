// mapalign.hip — MI355X (gfx950 / CDNA4) map-alignment engine: kernels + C ABI.
//
// Hot path (the reference body src/map_alignment/align.py is
// absent, call site /root/reference/task_def.py:175-187):
//   k_nn_accum   a4+a5  per source point: rigid transform (fp64), exact nearest gold
//                       point through a uniform grid stored in HBM (wavefront-cooperative
//                       row / chunk work queues in LDS, ring by ring with distance
//                       pruning), distance gate, fp64 accumulation of {W, s_p, s_q, S, E};
//                       DPP + LDS reduction -> one fp64 partial per workgroup (no float
//                       atomics => bitwise reproducible)
//   k_reduce_solve a5-a7 fixed-order sum of the partials, one workgroup per accumulator; the
//                       last-arriving workgroup runs the 3x3 Jacobi-SVD Kabsch step, the
//                       transform composition and the convergence test
//                       on the device, so the loop never synchronises with the host
//   k_reduce19, k_solve the same two halves as separate launches (sharded-source stepping)
//   k_unproject  a8     depth image -> world points written straight into the source
//                       cloud buffer, pixel i in slot i (invalid pixels leave a hole the search skips)
// Everything is written for 64-wide wavefronts; there is no MFMA because the path is a
// gather + reduction (<0.5 flop/B), and no CPU fallback.
//
// Memory layout in HBM (all owned by the ctx):
//   gold[l] float4[M]  four levels (cell edge c, 2c, 4c, 8c), each cell-sorted (x fastest),
//                      coordinates relative to the centring origin, .w = the point's
//                      position in level 0 (bit pattern); perm0 maps that to the caller's index
//   cstart[l] u32[ncells+1] exclusive prefix of points per cell: the points of the x-run
//                      [x0,x1] of row (y,z) are gold[cstart[row+x0] .. cstart[row+x1+1])
//   occ_dist u8[ncells] level 0: Chebyshev distance (cells, capped) to the nearest occupied cell
//   src    float4[N]   brick-major sorted in its own frame (256 consecutive points = a compact
//                      patch), .w = per-point weight; prev u32[N] = the match of the last SEARCH launch
//   lists  ListPlanes  per source slot: 8 gold positions in two planes of 4 + the radius the list is complete for
//                      (list launches: DESIGN.md 5)
//   partials f64[19][G] one per 256-point chunk, state (R,t,flags), trace f64[iters][20],
//   wg_cost u32[G] + seg_order u32[<=512]  workgroup durations and the longest-first schedule
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded with dlopen at ma_comm_init

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "host_math.h"
#include "kabsch.h"
#include "list_planner.h"
#include "mapalign.h"

namespace {

constexpr int kBlock = 256;           // 4 wavefronts
constexpr int kNNBlock = ma::kSearchBlock;  // workgroup of the search kernel (MA_NN_BLOCK)
constexpr int kWave = 64;
constexpr int kNAcc = 19;             // 17 accumulators + rows visited + candidates
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;
constexpr uint32_t kHoleBitsEarly = 0x7FC00000u;   // = kHoleBits (k_unproject): x of a hole, a quiet NaN

// Candidate lists (k_nn_accum, "list launches"): once the transform moves the cloud by less than a fraction of
// a cell per iteration, a search launch also WRITES, per query, every gold point within (match distance + skin) of
// the query's position; while a query has moved less than skin / 2 since then, its nearest neighbour is provably
// in that list (triangle inequality, checked per query), and the launch reads up to 8 gold positions instead of searching.
constexpr int kListLen = 8;
// Storage, every plane indexed by source slot.  The build in force (MA_LIST_INLINE = 0) keeps POSITIONS only: plane b0
// holds entries 0-3 (16 bytes per query, read by every list launch), plane b1 entries 4-7 (read by the few queries with
// more than four), plane rad the radius the list is complete for (its low 4 bits: the number of entries).
// MA_LIST_INLINE = 2 (A/B, csrc/Makefile `variant`) is the form VERDICT r03 asked for: entries 0 and 1 INLINE — the gold
// point's coordinates and its position, 16 bytes each, planes a0 / a1 — so that the common list (<= 2 entries) costs no
// gather; b0 then holds entries 2-5, b1 entries 6-7; an unused inline entry is (inf, inf, inf, none).  Measured SLOWER,
// 25.7 against 22.8 us per list launch: 16 more streamed bytes per query cost more than the gathers they save, which hit
// the XCD's L2 (profiles/r04_variants_ab.txt).
#ifndef MA_LIST_INLINE
#define MA_LIST_INLINE 0
#endif
constexpr int kListInline = MA_LIST_INLINE;
static_assert(kListInline == 0 || kListInline == 2, "inline entries");
struct ListPlanes {
  float4* __restrict__ a0;
  float4* __restrict__ a1;
  uint32_t* __restrict__ b0;     // [slots][4]: entries kListInline .. kListInline + 3
  uint32_t* __restrict__ b1;     // [slots][4]: the entries behind them (2 or 4 in use)
  float* __restrict__ rad;       // radius the list is complete for; low 4 bits: entries
};
constexpr size_t kListBytesPerSlot = 2 * sizeof(float4) + 4 * sizeof(uint32_t) + 4 * sizeof(uint32_t) + sizeof(float);
using ma::kPlan;                     // iterations per planning batch of ma_icp (list_planner.h)
constexpr int kMaxSlots = 4096;      // iterations of a run whose states the planner can read (longer runs search after that)
struct ListCtl {
  int have;          // lists exist (written by the launch that ran with the transform Rb, tb)
  float skin;        // metres
  float rs;          // largest distance of a corner of the source cloud's box from the centring origin; <= 0: no lists for this cloud
  float blo[3], bhi[3];   // that box (source coordinates, centred frame)
  float slack;       // rounding allowance of the per-query displacement, metres
  double Rb[9], tb[3];
};
// What every solve of a loop that may use lists leaves in pinned host memory for the host's planner: the state it
// produced, in a slot of its own (iteration k -> slot k), so that the host reads the transform of a fixed iteration
// whatever the device has done since.  Posted writes: the solve does not wait for them.
struct StateSlot {
  ma::DevState s;
  volatile unsigned long long chk;   // slot_check(R, t, seq): the host takes a slot only when it matches what it reads —
  volatile int seq;                  // iterations solved (= s.iter) — so the solve needs no fence between the posted writes
  int pad;                           // (a system-scope fence there cost 1.4 us of every iteration)
};
__host__ __device__ inline unsigned long long slot_check(const double* R, const double* t, int seq);
// host: has the solve of iteration `want - 1` written this slot completely?  (every field read through volatile)
inline bool slot_ready(const volatile StateSlot* sl, int want);
// (The solving lane pays ~4 ns per instruction, 50 times per alignment: an exclusive-or per 32-bit half and word — every
// second word with its halves swapped — instead of a 64-bit multiply and twelve 64-bit rotations.  What it has to catch is a
// slot whose words come from two different solves, and the slots of a new run start out invalid.)
__host__ __device__ inline unsigned long long slot_check(const double* R, const double* t, int seq) {
  unsigned long long x = ((unsigned long long)(unsigned)seq << 32 | (unsigned)~seq) ^ 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 12; ++i) {
    unsigned long long b;
    memcpy(&b, i < 9 ? &R[i] : &t[i - 9], 8);
    x ^= (i & 1) ? (b << 32 | b >> 32) : b;
  }
  return x;
}
inline bool slot_ready(const volatile StateSlot* sl, int want) {
  if (sl->seq != want) return false;
  double R[9], t[3];
  for (int i = 0; i < 9; ++i) R[i] = sl->s.R[i];
  for (int i = 0; i < 3; ++i) t[i] = sl->s.t[i];
  return sl->chk == slot_check(R, t, want);
}

// plain structs and host arithmetic shared with the CPU-side sanitizer test (host_math.h)
using ma::DevState;
using ma::GridP;
using ma::kMaxCells;
using ma::div_up;
using ma::world_to_centred;
using ma::centred_to_world;
using ma::acc_to_world;
using ma::open_gate2;
using ma::make_grid;
using ma::clamp_cell_to_budget;
using ma::grid_for;

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
// fp64 add of a DPP-shifted copy (two 32-bit DPP moves + one add; no LDS traffic).
// bound_ctrl = 1 makes lanes without a source read 0, and every row is enabled, so no
// "old" operand has to be materialised (that cost two extra moves per step before).
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
  return v + __hiloint2double(hi2, lo2);
}
// Wavefront sum on the VALU only (DPP row shifts + row broadcasts; no LDS permutes, which
// measured 6 us per launch slower).  Only lane 63 holds the total: the broadcasts go to
// all rows, so the other lanes end with partial sums nobody reads.
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0x111>(v);  // row_shr:1
  v = dpp_add<0x112>(v);  // row_shr:2
  v = dpp_add<0x114>(v);  // row_shr:4
  v = dpp_add<0x118>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_add<0x142>(v);  // row_bcast:15: row r += row sum of row r-1
  v = dpp_add<0x143>(v);  // row_bcast:31: rows 2,3 += lane 31 -> lane 63 = total
  return v;
}
constexpr int kSumLane = kWave - 1;


// hardware square root (1 ulp, one instruction instead of the ~18 of the correctly rounded
// sqrtf).  Only used for search radii, which carry the boundary slack `delta` anyway.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ uint32_t cell_of(float x, float y, float z, const GridP& g) {
  int ix = (int)floorf((x - g.ox) * g.inv_c);
  int iy = (int)floorf((y - g.oy) * g.inv_c);
  int iz = (int)floorf((z - g.oz) * g.inv_c);
  ix = min(max(ix, 0), g.nx - 1);
  iy = min(max(iy, 0), g.ny - 1);
  iz = min(max(iz, 0), g.nz - 1);
  return (uint32_t)((iz * g.ny + iy) * g.nx + ix);
}

// ---------------------------------------------------------------------------------------
// grid build (once per cloud; not part of the timed loop)
// ---------------------------------------------------------------------------------------
// Per-workgroup min / max / non-finite count of an n x 3 fp32 cloud.
// STRIDE floats per point: 3 for raw clouds, 4 for the float4 buffers.
template <int STRIDE>
__global__ __launch_bounds__(kBlock) void k_minmax(const float* __restrict__ xyz, uint32_t n,
                                                   float* __restrict__ out /*[grid][8]*/, const int holes_ok) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float bad = 0.f;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float x = xyz[STRIDE * (size_t)i], y = xyz[STRIDE * (size_t)i + 1], z = xyz[STRIDE * (size_t)i + 2];
    if (holes_ok && __float_as_uint(x) == kHoleBitsEarly) continue;     // a hole of the in-place unprojection (float4 buffers)
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
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
  }
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { red[wv][a] = lo[a]; red[wv][3 + a] = hi[a]; }
    red[wv][6] = bad; red[wv][7] = 0.f;
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

__global__ __launch_bounds__(kBlock) void k_sort_hist(const uint32_t* __restrict__ keys, const uint32_t n, const int shift,
                                                      const uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kSortBins];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = base + (uint32_t)j * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & (kSortBins - 1)], 1u);
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

__global__ __launch_bounds__(kBlock) void k_sort_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                         const uint32_t n, const int shift, const uint32_t nblocks,
                                                         const uint32_t* __restrict__ row_prefix, const uint32_t* __restrict__ totals,
                                                         uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
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
  uint32_t key[kSortItems], rank[kSortItems];
#pragma unroll
  for (int j = 0; j < kSortItems; ++j) {
    const uint32_t i = seg + (uint32_t)j * kWave + (uint32_t)lane;
    const bool valid = i < n;
    key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
    const uint32_t d = (key[j] >> shift) & (kSortBins - 1);
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
      const uint32_t pos = cnt[w][(key[j] >> shift) & (kSortBins - 1)] + rank[j];
      keys_out[pos] = key[j];
      vals_out[pos] = vals_in[i];
    }
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
__global__ __launch_bounds__(kBlock) void k_moments(const float4* __restrict__ pts, const uint32_t n, double* __restrict__ out) {
  double a[kMomLen];
#pragma unroll
  for (int k = 0; k < kMomLen; ++k) a[k] = 0.0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
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

// ---------------------------------------------------------------------------------------
// a4: exact nearest neighbour through a multi-resolution grid
// ---------------------------------------------------------------------------------------
// kLevels uniform grids over the same box, cell edges c0, 2 c0, 4 c0, 8 c0, each with its own
// cell-sorted (x fastest) copy of the gold map.  A query with search radius r is answered at
// the finest level whose cell edge is >= r: there the sphere cannot leave the 3 x 3 rows
// around the query's row, nor the 3 cells around its cell inside a row, so the whole search
// is nine independent row tests, at most nine 16-byte table reads (issued together) and a
// few short contiguous runs of candidates — a fixed, shallow dependency chain instead of a
// data-dependent ring walk.  r comes from the previous iteration's match (temporal
// coherence) or, without one, from a probe of the finest level; if a pass ends with a best
// distance larger than its level's cell edge, the query moves up to the level that covers
// it.  Beyond the coarsest level (ungated far-apart clouds) an exact ring walk finishes.
constexpr int kLevels = 4;

struct LevelSet {
  const float4* gold[kLevels];     // .w = position of the point in level 0's array
  const uint32_t* cs[kLevels];     // exclusive prefix of points per cell (+4 words padding)
  float c[kLevels], inv_c[kLevels];
  int nx[kLevels], ny[kLevels], nz[kLevels];
  float ox, oy, oz;                // common min corner (centred frame)
  float delta;                     // boundary slack (a few ulp of the largest coordinate)
  const uint8_t* occ_dist;         // level 0: Chebyshev distance (cells, capped) to the nearest occupied cell
};

struct LevelView {
  const float4* gold;
  const uint32_t* cs;
  float c, inv_c;
  int nx, ny, nz;
};

struct u4_unaligned { uint32_t x, y, z, w; } __attribute__((packed, aligned(4)));

// base + 32-bit byte offset: with a wave-uniform base this is the scalar-base addressing
// form (global_load ... v_offset, s[base]) — one shift per address instead of 64-bit
// per-lane pointer arithmetic.  Valid because M < 2^28 points and < 2^30 cells.
__device__ __forceinline__ float4 ld_point(const float4* __restrict__ base, uint32_t j) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (j << 4));
}
// Coordinates only (12 of the 16 bytes).  The candidate scan is bound by the L1 return path
// (a wavefront-wide 16-byte load is 16 cycles of it, a 12-byte one 12: forcing full float4
// loads cost 10 % of the whole alignment), and in level 0 the fourth word is the point's own
// position, i.e. the index it was loaded from.
__device__ __forceinline__ float3 ld_xyz(const float4* __restrict__ base, uint32_t j) {
  return *reinterpret_cast<const float3*>(reinterpret_cast<const char*>(base) + (j << 4));
}
__device__ __forceinline__ uint32_t ld_cell(const uint32_t* __restrict__ base, uint32_t c) {
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + (c << 2));
}
__device__ __forceinline__ u4_unaligned ld_cell4(const uint32_t* __restrict__ base, uint32_t c) {
  return *reinterpret_cast<const u4_unaligned*>(reinterpret_cast<const char*>(base) + (c << 2));
}

// The squared distance, in ONE association everywhere (search, per-lane search, list launches): which of two gold
// points is nearer is decided on these bits, so every path must form them the same way — spelled out, with
// contraction off, instead of left to what the compiler makes of `dx*dx + dy*dy + dz*dz` in each context.
// Form (A/B in profiles/r04_variants_ab.txt): (x, y) as ONE packed pair — v_pk_add_f32 for the two differences,
// v_pk_mul_f32 for their squares (IEEE products) — then one fma and one add:  d = fma(dz, dz, dx*dx) + dy*dy.
// A chain of three fused multiply-adds is one instruction per candidate more and cost the converged search launches
// 10 % when the association was first pinned that way.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef MA_DIST_FORM
#define MA_DIST_FORM 1
#endif
__device__ __forceinline__ float dist2(const float px, const float py, const float pz, const float qx, const float qy,
                                       const float qz) {
#pragma clang fp contract(off)
#if MA_DIST_FORM == 1
  const f32x2 d = f32x2{px, py} - f32x2{qx, qy};
  const f32x2 sq = d * d;
  const float dz = pz - qz;
  return __builtin_fmaf(dz, dz, sq.x) + sq.y;
#else
  const float dx = px - qx, dy = py - qy, dz = pz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
#endif
}

// ONE tie rule for every path (round 4): the nearest neighbour is the gold point with the smallest
// (fp32 squared distance, position in level 0) in lexicographic order — the order of the packed 64-bit keys of the
// cooperative search.  Which of two gold points at the SAME fp32 distance wins therefore no longer depends on the
// order in which a search came across them, on last iteration's match, or on whether the launch searched or read
// candidate lists.  (The open gate is a distance no real match may equal: a tie against it, with no match held, loses.)
//
// Four candidates per trip: the run is contiguous, so the four loads are in flight together
// (indices past the end are clamped to the last point: a duplicate changes nothing).
// Only (distance, position) of the winner are tracked; its coordinates are re-read once.
__device__ __forceinline__ void scan_run(const float px, const float py, const float pz,
                                         const float4* __restrict__ gold, const uint32_t a,
                                         const uint32_t b, float& best2, uint32_t& bestj) {
  if (a >= b) return;
  // The tie rule as ONE 64-bit comparison per candidate: keys (distance bits, position) — squared distances are
  // non-negative, so their bit patterns order like the values.  "Nothing held, gate g" is the key (bits(g) - 1, none):
  // a candidate strictly inside the gate is smaller, one AT the open gate is not.
  unsigned long long key = ((unsigned long long)(__float_as_uint(best2) - (bestj == kNoMatch ? 1u : 0u)) << 32) | bestj;
  for (uint32_t j = a; j < b; j += 4u) {
    float4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = ld_point(gold, min(j + (uint32_t)u, b - 1u));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d = dist2(px, py, pz, q[u].x, q[u].y, q[u].z);
      const unsigned long long ku = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q[u].w);
      key = ku < key ? ku : key;
    }
  }
  bestj = (uint32_t)key;
  best2 = __uint_as_float((uint32_t)(key >> 32) + (bestj == kNoMatch ? 1u : 0u));
}

// One pass at one level: the 3 x 3 rows around the query's row (rings 0 and 1), each over
// the x-run the current best sphere reaches.  Exact for every gold point within
// (c - delta) of the query.
template <bool COUNT>
__device__ __forceinline__ void level_pass(const float px, const float py, const float pz,
                                           const LevelView& V, const LevelSet& S, float& best2,
                                           uint32_t& bestj, uint32_t& nrows, uint32_t& ncand) {
  const float uy = fminf(fmaxf((py - S.oy) * V.inv_c, -1.f), (float)V.ny);
  const float uz = fminf(fmaxf((pz - S.oz) * V.inv_c, -1.f), (float)V.nz);
  const int cy = (int)floorf(uy), cz = (int)floorf(uz);
  // three z-slabs (centre first so that the bound shrinks early); a slab the best sphere
  // cannot reach costs no memory access at all
  for (int sl = 0; sl < 3; ++sl) {
    const int z = cz + (sl == 0 ? 0 : (sl == 1 ? -1 : 1));
    if ((unsigned)z >= (unsigned)V.nz) continue;
    const float zlo = S.oz + (float)z * V.c;
    const float dzq = fmaxf(0.f, fmaxf(zlo - S.delta - pz, pz - (zlo + V.c + S.delta)));
    if (dzq * dzq > best2) continue;      // (>, <= below: a tie at distance 0 must still be examined)
    uint32_t a[3], b[3];
    // phase 1: the slab's three row tests + table reads, independent => one round trip
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int y = cy + (k == 0 ? 0 : (k == 1 ? -1 : 1));
      a[k] = 0u;
      b[k] = 0u;
      if ((unsigned)y < (unsigned)V.ny) {
        const float ylo = S.oy + (float)y * V.c;
        const float dyq = fmaxf(0.f, fmaxf(ylo - S.delta - py, py - (ylo + V.c + S.delta)));
        const float d2yz = dyq * dyq + dzq * dzq;
        if (d2yz <= best2) {
          const float rx = fast_sqrt(best2 - d2yz) + S.delta;  // +inf while ungated and unmatched
          const int x0 = (int)fmaxf(floorf((px - rx - S.ox) * V.inv_c), 0.f);
          const int x1 = (int)fminf(floorf((px + rx - S.ox) * V.inv_c), (float)(V.nx - 1));
          if (x0 <= x1) {
            const uint32_t cell = (uint32_t)((z * V.ny + y) * V.nx + x0);
            const int span = x1 - x0;  // cells beyond the first
            if (span <= 2) {           // the usual case: one 16-byte read gives both ends
              const u4_unaligned v = ld_cell4(V.cs, cell);
              a[k] = v.x;
              b[k] = span == 0 ? v.y : (span == 1 ? v.z : v.w);
            } else {
              a[k] = ld_cell(V.cs, cell);
              b[k] = ld_cell(V.cs, cell + (uint32_t)span + 1u);
            }
            if (COUNT) { nrows += span <= 2 ? 4u : 2u; ncand += b[k] - a[k]; }  // table words read
          }
        }
      }
    }
    // phase 2: candidates
#pragma unroll
    for (int k = 0; k < 3; ++k) scan_run(px, py, pz, V.gold, a[k], b[k], best2, bestj);
  }
}

// Exact ring walk on one level, from ring r_start outwards.  Rows (y,z) are visited in
// square rings of growing Chebyshev radius r around the query's row; every point of ring r
// is at least (r-1)*c away, so the walk stops as soon as that bound reaches the best
// distance found.  Inside a row only the x-run the best sphere can reach is scanned — one
// contiguous slice of the level's array.  Rows are handled four at a time: their tests
// and table reads are independent, so a batch costs one memory round trip, not four.
template <bool COUNT>
__device__ __forceinline__ void ring_walk(const float px, const float py, const float pz,
                                          const LevelView& V, const LevelSet& S, const int r_start,
                                          float& best2, uint32_t& bestj, uint32_t& nrows,
                                          uint32_t& ncand) {
  // centre row, clamped to one cell outside the grid: for a query outside the grid the
  // ring bound (r-1)*c then under-estimates the true distance, which keeps it valid.
  const float uy = fminf(fmaxf((py - S.oy) * V.inv_c, -1.f), (float)V.ny);
  const float uz = fminf(fmaxf((pz - S.oz) * V.inv_c, -1.f), (float)V.nz);
  const int cy = (int)floorf(uy), cz = (int)floorf(uz);
  const int r0 = max(r_start, max(max(-cy, cy - (V.ny - 1)), max(-cz, cz - (V.nz - 1))));
  const int rmax = max(max(cy, V.ny - 1 - cy), max(cz, V.nz - 1 - cz));
  for (int r = r0; r <= rmax; ++r) {
    if (r >= 2) {
      const float lb = (float)(r - 1) * V.c - S.delta;
      if (lb * lb >= best2) break;
    }
    const int nside = r ? 4 : 1, len = r ? 2 * r : 1;
    for (int s = 0; s < nside; ++s) {
      for (int o4 = 0; o4 < len; o4 += 4) {
        uint32_t a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] = 0u;
          b[u] = 0u;
          const int o = o4 + u;
          if (o < len) {
            int dy, dz;
            if (s == 0)      { dy = o - r; dz = -r; }
            else if (s == 1) { dy = r;     dz = o - r; }
            else if (s == 2) { dy = r - o; dz = r; }
            else             { dy = -r;    dz = r - o; }
            const int y = cy + dy, z = cz + dz;
            if ((unsigned)y < (unsigned)V.ny && (unsigned)z < (unsigned)V.nz) {
              const float ylo = S.oy + (float)y * V.c, zlo = S.oz + (float)z * V.c;
              const float dyq = fmaxf(0.f, fmaxf(ylo - S.delta - py, py - (ylo + V.c + S.delta)));
              const float dzq = fmaxf(0.f, fmaxf(zlo - S.delta - pz, pz - (zlo + V.c + S.delta)));
              const float d2yz = dyq * dyq + dzq * dzq;
              if (d2yz <= best2) {
                const float rx = fast_sqrt(best2 - d2yz) + S.delta;
                const int x0 = (int)fmaxf(floorf((px - rx - S.ox) * V.inv_c), 0.f);
                const int x1 = (int)fminf(floorf((px + rx - S.ox) * V.inv_c), (float)(V.nx - 1));
                if (x0 <= x1) {
                  const uint32_t row = (uint32_t)((z * V.ny + y) * V.nx);
                  a[u] = ld_cell(V.cs, row + (uint32_t)x0);
                  b[u] = ld_cell(V.cs, row + (uint32_t)x1 + 1u);
                  if (COUNT) { nrows += 2u; ncand += b[u] - a[u]; }  // table words read
                }
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) scan_run(px, py, pz, V.gold, a[u], b[u], best2, bestj);
      }
    }
  }
}

// Driver: best2 / bestj come in initialised by the caller — (gate, none) or the previous
// iteration's match, which is a valid candidate and usually already the answer.
// The level is a WAVE-UNIFORM loop variable: all lanes still searching work on the same
// level, so its array bases and grid constants live in scalar registers.
template <bool COUNT>
__device__ __forceinline__ void nn_find(const float px, const float py, const float pz,
                                        const LevelSet& S, float& best2, uint32_t& bestj,
                                        uint32_t& nrows, uint32_t& ncand) {
  // 1. Probe the 3 x 3 rows of the finest level: the true neighbour of a point on a
  //    scanned surface is almost always there, and the pass is then already exact.
  // 2. If the best sphere is still larger than the cell but only a few cells wide (the
  //    usual case for a gated search), walk the outer rings of the SAME level: fine cells
  //    keep the number of candidates at what the sphere really contains.
  // 3. Only a sphere many cells wide (ungated, far-apart clouds) climbs to a coarser level,
  //    where the same two steps cover eight times the volume per table read.
  constexpr float kWalkCells = 6.f;
  bool active = isfinite(px) && isfinite(py) && isfinite(pz);
  // 0. A search sphere wider than ~1.5 cells (an early iteration, or a point with nothing
  //    remembered) first asks the distance-to-occupied table: if even the nearest occupied
  //    cell is farther than the sphere, there is nothing to find (the remembered match, if
  //    any, stays) — one byte instead of a ring walk through empty space.
  {
    const float c0 = S.c[0], wide0 = 1.5f * c0;
    if (active && best2 > wide0 * wide0) {
      const float ux = (px - S.ox) * S.inv_c[0], uy = (py - S.oy) * S.inv_c[0], uz = (pz - S.oz) * S.inv_c[0];
      if (ux >= 0.f && uy >= 0.f && uz >= 0.f && ux < (float)S.nx[0] && uy < (float)S.ny[0] && uz < (float)S.nz[0]) {
        const uint32_t cell = ((uint32_t)uz * (uint32_t)S.ny[0] + (uint32_t)uy) * (uint32_t)S.nx[0] + (uint32_t)ux;
        const int k = S.occ_dist[cell];
        const float lb = (float)max(k - 1, 0) * c0 - S.delta;
        if (lb > 0.f && lb * lb >= best2) active = false;
      }
    }
  }
#pragma unroll 1
  for (int l = 0; l < kLevels; ++l) {
    if (!__any(active)) break;
    const LevelView V{S.gold[l], S.cs[l], S.c[l], S.inv_c[l], S.nx[l], S.ny[l], S.nz[l]};
    if (active) {
      level_pass<COUNT>(px, py, pz, V, S, best2, bestj, nrows, ncand);
      const float lim = V.c - S.delta, wide = kWalkCells * V.c;
      if (best2 <= lim * lim) {
        active = false;  // the pass covered the whole best sphere: exact
      } else if (l == kLevels - 1 || best2 <= wide * wide) {
        // rings 0 and 1 are done (the pass scanned the full x-run of each of its rows)
        ring_walk<COUNT>(px, py, pz, V, S, 2, best2, bestj, nrows, ncand);
        active = false;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Wavefront-cooperative search: work queues in LDS instead of per-lane control flow
// ---------------------------------------------------------------------------------------
// The per-lane search wastes about half of its VALU lane-slots: a wavefront executes every
// row test and every candidate trip that ANY of its 64 queries needs.  Here the wavefront
// turns the work into ITEMS and spreads them evenly over its lanes:
//   phase 1  every query tests its candidate rows geometrically (cheap, all lanes busy) and
//            pushes the rows its sphere reaches into a row queue (ballot-prefix compaction);
//   phase 2  the row items are dealt to the lanes: x-range, table read, and the run is cut
//            into chunks of <= 8 candidates pushed into a chunk queue;
//   phase 3  the chunk items are dealt to the lanes: distances, and a 64-bit LDS atomic min
//            on (distance bits << 32 | position) per query — order-independent, hence
//            deterministic (equal distances resolve to the smaller position).
// Rounds: the 3 x 3 rows first, then one ring per round while some query's sphere reaches it.
// capacities per wavefront (A/B on MI355X: doubling them costs occupancy, 3.39 vs 3.08 ms;
// halving 3.12; chunks of 4, 6, 12 candidates: 3.39, 3.28, 3.30 ms against 3.08 for 8)
constexpr int kRowCap = 128, kChunkCap = 256, kChunkLen = 8;
#ifndef MA_ROWS_PER_TRIP
#define MA_ROWS_PER_TRIP 2
#endif

struct WaveLds {
  float4 pos[kWave];
  unsigned long long best[kWave];
  uint32_t row_cell[kRowCap];
  float row_d2yz[kRowCap];
  uint32_t chunk_a[kChunkCap];
  uint8_t row_lane[kRowCap];
  uint8_t chunk_lane[kChunkCap];
  uint8_t chunk_cnt[kChunkCap];
  uint32_t minx[kWave];      // list-writing launches: smallest squared distance (bits) among the candidates a full list turned away
#if MA_WGTIMES
  unsigned long long dbg_round0;
#endif
};

__device__ __forceinline__ unsigned long long pack_best(float d2, uint32_t j) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | j;
}
// The key a wavefront keeps per query in LDS: (distance bits + 1, position) for a match held, (open gate, none) without
// one.  The +1 — one ulp — is what lets a candidate at EXACTLY the distance of the match held compete on its position
// (the tie rule, scan_run) at no cost in the candidate loop: the loop's bound is the key's high word as it stands, a
// candidate passes it with d <= the distance held, and the 64-bit minimum over keys of the same form decides.  A tie
// against the open gate, with no match held, still loses.
__device__ __forceinline__ unsigned long long pack_held(float d2, uint32_t j) {
  return ((unsigned long long)(__float_as_uint(d2) + (j != kNoMatch ? 1u : 0u)) << 32) | j;
}
__device__ __forceinline__ float held_d2(unsigned long long key) {
  return __uint_as_float((uint32_t)(key >> 32) - ((uint32_t)key != kNoMatch ? 1u : 0u));
}
__device__ __forceinline__ uint32_t lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool COUNT, bool BUILD>
__device__ __forceinline__ void wq_flush_chunks(WaveLds& W, const int lane, const float4* __restrict__ gold0,
                                                uint32_t& nchunk, uint32_t& ncand, const float skin,
                                                const ListPlanes& lp, const uint32_t slot0) {
  for (uint32_t j0 = 0; j0 < nchunk; j0 += kWave) {
    const uint32_t j = j0 + lane;
    if (j < nchunk) {
      const uint32_t lq = W.chunk_lane[j], a = W.chunk_a[j], cnt = W.chunk_cnt[j];
      const float4 p = W.pos[lq];
      const unsigned long long cur = W.best[lq];
      // the bound as the key holds it: one ulp above the distance of the match held (pack_held), so that a candidate at
      // exactly that distance passes and competes on its position in the 64-bit minimum below
      const float b2_in = __uint_as_float((uint32_t)(cur >> 32));
      float b2 = b2_in;
      float reach2 = 0.f;      // list-writing launch: everything within (current best distance + skin) goes on the query's list
      if (BUILD) { const float r = fast_sqrt(b2_in) + skin; reach2 = r * r; }
      // Four candidates per trip at consecutive positions a+t .. a+t+3, WITHOUT clamping to the
      // chunk: the positions past its end hold other gold points (the next cells of the sorted
      // array; eight far-away pad entries follow the last point), and any gold point is a
      // legitimate candidate — the minimum over a superset that contains the true neighbour is
      // still the true neighbour.  One address per trip (the other three loads use immediate
      // offsets) and the winner is tracked as a small offset from `a`: 10 instead of 14 VALU
      // instructions per candidate.  Positions ascend and only a strictly nearer candidate replaces
      // the one held, so of equal candidates inside the chunk the smallest position stays.
      uint32_t bo = 0xFFFFFFFFu;
      for (uint32_t t = 0; t < cnt; t += 4u) {
        const char* __restrict__ at = reinterpret_cast<const char*>(gold0) + ((a + t) << 4);
        float3 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float3*>(at + 16 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = dist2(p.x, p.y, p.z, q[u].x, q[u].y, q[u].z);
          if (d < b2) { b2 = d; bo = t + (uint32_t)u; }
          if (BUILD) {
            if (d <= reach2 && t + (uint32_t)u < cnt) {      // real members of the chunk only
              const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t*>(&W.pos[lq]) + 3, 1u);
              const uint32_t pos = a + t + (uint32_t)u, at_q = slot0 + lq;
              constexpr uint32_t kIn = (uint32_t)kListInline;
              if (slot < kIn) (slot == 0u ? lp.a0 : lp.a1)[at_q] = make_float4(q[u].x, q[u].y, q[u].z, __uint_as_float(pos));
              else if (slot < kIn + 4u) lp.b0[at_q * 4u + (slot - kIn)] = pos;
              else if (slot < (uint32_t)kListLen) lp.b1[at_q * 4u + (slot - kIn - 4u)] = pos;
              else atomicMin(&W.minx[lq], __float_as_uint(d));
            }
          }
        }
      }
      if (COUNT) ncand += cnt;
      if (b2 < b2_in) {             // something at most as far as the match held (a real gold point: its key carries the +1)
        const unsigned long long mine = ((unsigned long long)(__float_as_uint(b2) + 1u) << 32) | (a + bo);
        if (mine < cur) atomicMin(&W.best[lq], mine);      // (the chunk that holds the match itself finds it again: equal keys, no atomic)
      }
    }
  }
  nchunk = 0;
}


template <bool COUNT, bool BUILD>
__device__ __forceinline__ void wq_flush_rows(WaveLds& W, const int lane, const LevelSet& S, uint32_t& nrow,
                                              uint32_t& nchunk, uint32_t& nrows_cnt, uint32_t& ncand, const float skin,
                                              const ListPlanes& lp, const uint32_t slot0) {
  const float4* __restrict__ gold0 = S.gold[0];
  const uint32_t* __restrict__ cs0 = S.cs[0];
  const float inv_c = S.inv_c[0];
  const int nx = S.nx[0];
  // MA_ROWS_PER_TRIP row items per lane and trip: their table reads are in flight together
  // (a lone wavefront in the tail of a launch has nobody else to hide that latency).
  // Measured: 1 / 2 / 3 / 4 items -> 2.95 / 2.77 / 2.96 / 3.04 ms (3 and 4 cross 64 VGPRs);
  // two chunk items per trip in wq_flush_chunks: 118 VGPRs, 3.62 ms.
  for (uint32_t i0 = 0; i0 < nrow; i0 += MA_ROWS_PER_TRIP * kWave) {
    uint32_t a[MA_ROWS_PER_TRIP], b[MA_ROWS_PER_TRIP], lq[MA_ROWS_PER_TRIP];
#pragma unroll
    for (int u = 0; u < MA_ROWS_PER_TRIP; ++u) {
      const uint32_t i = i0 + (uint32_t)u * kWave + lane;
      a[u] = 0; b[u] = 0; lq[u] = 0;
      if (i < nrow) {
        lq[u] = W.row_lane[i];
        const float d2yz = W.row_d2yz[i];
        const float4 p = W.pos[lq[u]];
        float b2 = __uint_as_float((uint32_t)(W.best[lq[u]] >> 32));
        if (BUILD) { const float r = fast_sqrt(b2) + skin; b2 = r * r; }
        if (d2yz <= b2) {
          const float rx = fast_sqrt(b2 - d2yz) + S.delta;
          const int x0 = (int)fmaxf(floorf((p.x - rx - S.ox) * inv_c), 0.f);
          const int x1 = (int)fminf(floorf((p.x + rx - S.ox) * inv_c), (float)(nx - 1));
          if (x0 <= x1) {
            const uint32_t cell = W.row_cell[i] + (uint32_t)x0;
            const int span = x1 - x0;
            if (span <= 2) {
              const u4_unaligned v = ld_cell4(cs0, cell);
              a[u] = v.x;
              b[u] = span == 0 ? v.y : (span == 1 ? v.z : v.w);
            } else {
              a[u] = ld_cell(cs0, cell);
              b[u] = ld_cell(cs0, cell + (uint32_t)span + 1u);
            }
            if (COUNT) nrows_cnt += span <= 2 ? 4u : 2u;
          }
        }
      }
    }
    // cut [a,b) into chunks; trip k pushes chunk k of every lane that still has one
#pragma unroll
    for (int u = 0; u < MA_ROWS_PER_TRIP; ++u) {
      for (uint32_t k = 0;; ++k) {
        const bool has = a[u] + k * kChunkLen < b[u];
        const unsigned long long mask = __ballot(has);
        if (mask == 0ull) break;
        const uint32_t cnt = (uint32_t)__popcll(mask);
        if (nchunk + cnt > (uint32_t)kChunkCap) wq_flush_chunks<COUNT, BUILD>(W, lane, gold0, nchunk, ncand, skin, lp, slot0);
        if (has) {
          const uint32_t pos = nchunk + lane_prefix(mask), ca = a[u] + k * kChunkLen;
          W.chunk_a[pos] = ca;
          W.chunk_lane[pos] = (uint8_t)lq[u];
          W.chunk_cnt[pos] = (uint8_t)min((uint32_t)kChunkLen, b[u] - ca);
        }
        nchunk += cnt;
      }
    }
  }
  nrow = 0;
}

// All 64 lanes of the wavefront must call this together.
template <bool COUNT, bool BUILD>
__device__ __forceinline__ void wave_search(WaveLds& W, const int lane, const bool has_query, const float px,
                                            const float py, const float pz, const LevelSet& S, float& best2,
                                            uint32_t& bestj, uint32_t& nrows_cnt, uint32_t& ncand,
                                            uint32_t& ndismissed, const float skin = 0.f,
                                            const ListPlanes& lp = ListPlanes{}, const uint32_t slot0 = 0) {
  const float c = S.c[0], inv_c = S.inv_c[0];
  const int nx = S.nx[0], ny = S.ny[0], nz = S.nz[0];
  const float4* __restrict__ gold0 = S.gold[0];
  const bool finite = has_query && isfinite(px) && isfinite(py) && isfinite(pz);
  W.pos[lane] = make_float4(px, py, pz, 0.f);
  W.best[lane] = pack_held(best2, bestj);
  if (BUILD) W.minx[lane] = 0x7F800000u;
  // list-writing launch: every pruning decision uses the sphere widened by the skin, so that everything within
  // (final match distance + skin) is examined — and put on the list by wq_flush_chunks
  auto reach = [&](const float b2) -> float {
    if (!BUILD) return b2;
    const float r = fast_sqrt(b2) + skin;
    return r * r;
  };
  float bi = reach(best2);
  const float uy = fminf(fmaxf((py - S.oy) * inv_c, -1.f), (float)ny);
  const float uz = fminf(fmaxf((pz - S.oz) * inv_c, -1.f), (float)nz);
  const int cy = (int)floorf(uy), cz = (int)floorf(uz);
  const int rmax = max(max(cy, ny - 1 - cy), max(cz, nz - 1 - cz));
  uint32_t nrow = 0, nchunk = 0;

  auto push_row = [&](const bool pred, const int y, const int z, const float d2yz) {
    const unsigned long long mask = __ballot(pred);
    const uint32_t cnt = (uint32_t)__popcll(mask);
    if (cnt == 0u) return;
    if (nrow + cnt > (uint32_t)kRowCap) wq_flush_rows<COUNT, BUILD>(W, lane, S, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
    if (pred) {
      const uint32_t pos = nrow + lane_prefix(mask);
      W.row_cell[pos] = (uint32_t)((z * ny + y) * nx);
      W.row_d2yz[pos] = d2yz;
      W.row_lane[pos] = (uint8_t)lane;
    }
    nrow += cnt;
  };

  // ---- round 0: the 3 x 3 rows (rings 0 and 1), centre row first.  The row test is
  // separable: squared gap to the slab of cells at offset d along y, and along z (6 values),
  // then one add and one compare per row (the direct form costs 22 VALU ops per row, a
  // quarter of a converged iteration).  Same bounds up to rounding, which `delta` covers.
  const float fy = py - (S.oy + (float)cy * c), fz = pz - (S.oz + (float)cz * c);
  auto gap2 = [&](const int d, const float f, const int ci, const int n) -> float {
    const float lo = ((float)d * c - S.delta) - f, hi = f - ((float)(d + 1) * c + S.delta);
    const float t = fmaxf(0.f, fmaxf(lo, hi));
    return (unsigned)(ci + d) < (unsigned)n ? t * t : NAN;     // outside the grid: no comparison with it holds (<= included)
  };
  {
    const float gy[3] = {gap2(0, fy, cy, ny), gap2(-1, fy, cy, ny), gap2(1, fy, cy, ny)};
    const float gz[3] = {gap2(0, fz, cz, nz), gap2(-1, fz, cz, nz), gap2(1, fz, cz, nz)};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int dy = (k % 3 == 0) ? 0 : (k % 3 == 1 ? -1 : 1), dz = (k / 3 == 0) ? 0 : (k / 3 == 1 ? -1 : 1);
      const float d2yz = gy[k % 3] + gz[k / 3];
      push_row(finite && d2yz <= bi, cy + dy, cz + dz, d2yz);
    }
  }
  wq_flush_rows<COUNT, BUILD>(W, lane, S, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
  wq_flush_chunks<COUNT, BUILD>(W, lane, gold0, nchunk, ncand, skin, lp, slot0);
#if MA_WGTIMES
  if (lane == 0) W.dbg_round0 = wall_clock64();
#endif
  {
    const unsigned long long key = W.best[lane];
    best2 = held_d2(key);
    bestj = (uint32_t)key;
    bi = reach(best2);
  }
  // ---- further rounds: ring r while some query's sphere reaches it (same level; a sphere
  // wider than kWalk cells is left to the per-lane multi-level search)
  constexpr float kWalk = 6.f;
  const float lim = c - S.delta, wide = kWalk * c;
  bool more = finite && bi > lim * lim && best2 <= wide * wide;
  const bool fallback = finite && best2 > wide * wide;
  // A sphere that is still wide after the 3 x 3 block (an early iteration: nothing, or only
  // something far, has been found) asks the distance-to-occupied table before walking rings:
  // if even the nearest occupied cell is farther than the sphere there is nothing more to
  // find, for one byte instead of ~40 empty rows.  (Never reached once the clouds are aligned.)
  ndismissed = 0;
  if (__any(more)) {
    const float ux = (px - S.ox) * inv_c;
    bool dismissed = false;
    if (more && ux >= 0.f && ux < (float)nx && (unsigned)cy < (unsigned)ny && (unsigned)cz < (unsigned)nz) {
      const int k = S.occ_dist[(uint32_t)((cz * ny + cy) * nx) + (uint32_t)ux];
      const float lbk = (float)max(k - 1, 0) * c - S.delta;
      dismissed = lbk > 0.f && lbk * lbk > bi;   // (strictly: an equal distance may still win on position)
    }
    more = more && !dismissed;
    ndismissed = (uint32_t)__popcll(__ballot(dismissed));
  }
  for (int r = 2; __any(more); ++r) {
    const float lb = (float)(r - 1) * c - S.delta;
    more = more && r <= rmax && lb * lb < bi;
    if (!__any(more)) break;
    // the four sides of the ring; along a side one offset is fixed, so a row costs one gap
    // evaluation, one add and one compare
    for (int side = 0; side < 4; ++side) {
      const bool zfixed = (side & 1) == 0;                    // sides 0, 2: dz = -r, +r; sides 1, 3: dy = +r, -r
      const int dfix = (side == 0 || side == 3) ? -r : r;
      const float ffix = zfixed ? fz : fy, fvar = zfixed ? fy : fz;
      const int cfix = zfixed ? cz : cy, cvar = zfixed ? cy : cz, nfix = zfixed ? nz : ny, nvar = zfixed ? ny : nz;
      const float gfix = gap2(dfix, ffix, cfix, nfix);
      for (int o = 0; o < 2 * r; ++o) {
        const int dvar = side < 2 ? o - r : r - o;
        const float d2yz = gfix + gap2(dvar, fvar, cvar, nvar);
        const int dy = zfixed ? dvar : dfix, dz = zfixed ? dfix : dvar;
        push_row(more && d2yz <= bi, cy + dy, cz + dz, d2yz);
      }
    }
    wq_flush_rows<COUNT, BUILD>(W, lane, S, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
    wq_flush_chunks<COUNT, BUILD>(W, lane, gold0, nchunk, ncand, skin, lp, slot0);
    const unsigned long long key = W.best[lane];
    best2 = held_d2(key);
    bestj = (uint32_t)key;
    bi = reach(best2);
  }
  if (fallback) nn_find<COUNT>(px, py, pz, S, best2, bestj, nrows_cnt, ncand);
  if (BUILD) {
    // The list is complete inside the radius L: everything within (match distance + skin) was examined — the
    // block and the rings cover the widened sphere, a dismissed query has nothing that near at all — and put on
    // the list unless the list was full; then L stops at the nearest candidate that was turned away.  A query
    // that left for the multi-level search has no list (L = 0: it searches again in every list launch); a hole
    // never matches (L = inf, nothing to check).  The low 4 bits of L carry the number of entries (work counters).
    const uint32_t cnt = reinterpret_cast<const uint32_t*>(&W.pos[lane])[3];
    float L = fallback ? 0.f : fast_sqrt(best2) + skin;
    if (cnt > (uint32_t)kListLen) L = fminf(L, fast_sqrt(__uint_as_float(W.minx[lane])));
    L *= 1.f - 4e-6f;
    if (!finite) L = INFINITY;
    const uint32_t kept = min(cnt, (uint32_t)kListLen);
    if (has_query) {
      const uint32_t at_q = slot0 + (uint32_t)lane;
      // unused inline entries never pass the gate; unused positions of a plane that is read repeat the match (any gold
      // point may sit on a list: a superset stays exact)
      constexpr uint32_t kIn = (uint32_t)kListInline;
      const float4 none = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(kNoMatch));
      if (kIn > 0u && kept < 1u) lp.a0[at_q] = none;
      if (kIn > 1u && kept < 2u) lp.a1[at_q] = none;
      const uint32_t fill = bestj != kNoMatch ? bestj : 0u;
      // plane b0 is read when the list has more than kIn entries (always, without inline entries), b1 beyond kIn + 4
      if (kIn == 0u || kept > kIn) for (uint32_t sl = max(kept, kIn); sl < kIn + 4u; ++sl) lp.b0[at_q * 4u + (sl - kIn)] = fill;
      if (kept > kIn + 4u) for (uint32_t sl = kept; sl < (uint32_t)kListLen; ++sl) lp.b1[at_q * 4u + (sl - kIn - 4u)] = fill;
      lp.rad[at_q] = isfinite(L) ? __uint_as_float((__float_as_uint(L) & ~0xFu) | kept) : L;
    }
  }
}

// XCD-aware chunk order: workgroups are dealt round-robin over the 8 XCDs, so workgroups b
// and b+8 share an L2.  The sorted source cloud is cut into segments of kXcdSeg consecutive
// chunks and the segments are dealt over the XCDs: every XCD works on contiguous pieces
// (neighbouring workgroups share gold cells in its private L2) but gets pieces from all
// over the cloud, so that an expensive region (e.g. a part of the scan the gold map does
// not cover) is not one XCD's problem alone.
constexpr uint32_t kXcdSeg = ma::kXcdSegment;   // MA_XCD_SEG
#ifndef MA_SCHED
#define MA_SCHED 1
#endif
constexpr int kMaxGroups = ma::kMaxSchedGroups;   // scheduling groups (of 2^gshift segments) ranked by cost

// Longest-first schedule.  While the clouds are still apart, the workgroups whose points find
// no gold point inside the gate walk whole gate spheres and run 3-5x longer than the median;
// dispatched in cloud order they left the second half of a launch at < 30 % occupancy (tail
// of 50 us in a 100-us launch, tools/wgtimeline.py).  Every workgroup therefore records its
// duration, k_reduce_solve ranks the segments (in groups) by the previous iteration's cost,
// and the next launch hands them out most expensive first — dealt over the XCDs in that
// order, which also evens out the XCDs.  The schedule decides only WHEN a chunk is
// processed: partials are indexed by chunk, so sums (and results) do not depend on it.
struct Sched {
  uint32_t* __restrict__ wg_cost;           // [G] duration of the workgroup that processed chunk g (100 MHz ticks)
  const uint32_t* __restrict__ order;       // [G / kXcdSeg >> gshift] group ids, most expensive first; null: cloud order
  uint32_t gshift;
};

__device__ __forceinline__ uint32_t xcd_chunk(const uint32_t b, const Sched& sc, const bool use_order) {
  const uint32_t xcd = b % 8u, slot = b / 8u;            // slot: index among this XCD's workgroups
  uint32_t segpos = (slot / kXcdSeg) * 8u + xcd;         // position of the segment in the schedule
  if (use_order) {
    const uint32_t gp = segpos >> sc.gshift;
    segpos = (sc.order[gp] << sc.gshift) + (segpos - (gp << sc.gshift));
  }
  return segpos * kXcdSeg + slot % kXcdSeg;
}

// a4 + a5 fused.  MODE 0: accumulate only; MODE 1: per-point (idx, d2) only (ma_nn).
// One source point per thread: many short-lived wavefronts hide the search's memory latency
// best (2 and 4 points per thread measured 25-70 % slower), and the 17 fp64 accumulator terms
// are formed only after the search, so they hold no registers during it.

// The fp64 image of a source point under the state's transform, in ONE association (every launch kind must arrive at
// the same bits: a search launch forms it twice, a list launch once).
__device__ __forceinline__ void point_image(const DevState* __restrict__ st, const float4 s, double& Px, double& Py, double& Pz) {
  const double x = s.x, y = s.y, z = s.z;
  Px = __fma_rn(st->R[2], z, __fma_rn(st->R[1], y, __fma_rn(st->R[0], x, st->t[0])));
  Py = __fma_rn(st->R[5], z, __fma_rn(st->R[4], y, __fma_rn(st->R[3], x, st->t[1])));
  Pz = __fma_rn(st->R[8], z, __fma_rn(st->R[7], y, __fma_rn(st->R[6], x, st->t[2])));
}
// a5: the 17 terms of one correspondence (w, w p', w q, w p' q^T row-major, w |p' - q|^2)
__device__ __forceinline__ void pair_terms(double (&acc)[kNAcc], const double w, const double Px, const double Py, const double Pz,
                                           const float3 q) {
  const double qx = q.x, qy = q.y, qz = q.z;
  const double dx = Px - qx, dy = Py - qy, dz = Pz - qz;
  const double wPx = w * Px, wPy = w * Py, wPz = w * Pz;
  acc[0] = w;
  acc[1] = wPx; acc[2] = wPy; acc[3] = wPz;
  acc[4] = w * qx; acc[5] = w * qy; acc[6] = w * qz;
  acc[7] = wPx * qx;  acc[8] = wPx * qy;  acc[9] = wPx * qz;
  acc[10] = wPy * qx; acc[11] = wPy * qy; acc[12] = wPy * qz;
  acc[13] = wPz * qx; acc[14] = wPz * qy; acc[15] = wPz * qz;
  acc[16] = w * __fma_rn(dz, dz, __fma_rn(dy, dy, dx * dx));
}
// Workgroup reduction of the live accumulators of one 256-point chunk, fixed order (deterministic), shared by every
// launch kind so that their partials are the same bits:
//   1. two DPP steps fold each group of 4 lanes (6 VALU ops per value; a full wavefront
//      reduction in DPP costs 18 and was a quarter of the search kernel's VALU work);
//   2. the 16 group sums per wavefront go to LDS; 8 threads per accumulator each add
//      their share in index order;
//   3. one thread per accumulator adds the 8 partial sums and writes the partial.
// Without work counting only the 17 real accumulators take part (the counter slots stay 0).
// WV: per-wavefront LDS area with a member v[kNAcc][16]; two barriers inside.
constexpr int kRedParts = 8;
template <bool COUNT, typename WV>
__device__ __forceinline__ void reduce_chunk(const double (&acc)[kNAcc], WV* __restrict__ sh_w, double (*sh_part)[kRedParts],
                                             double* __restrict__ partials, const uint32_t G, const uint32_t chunk,
                                             const int lane, const int wv) {
  constexpr int kNW = kNNBlock / kWave;
  constexpr int kLive = COUNT ? kNAcc : ma::kAccLen;
  constexpr int kVals = kNW * 16, kPer = kVals / kRedParts;
#pragma unroll
  for (int k = 0; k < kLive; ++k) {
    double v = dpp_add<0x111>(acc[k]);
    v = dpp_add<0x112>(v);
    if ((lane & 3) == 3) sh_w[wv].v[k][lane >> 2] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLive * kRedParts) {
    const int k = threadIdx.x / kRedParts, part = threadIdx.x % kRedParts;
    double v = 0.0;
#pragma unroll
    for (int t = 0; t < kPer; ++t) {
      const int idx = part * kPer + t;
      v += sh_w[idx / 16].v[k][idx % 16];
    }
    sh_part[k][part] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNAcc) {
    const int k = threadIdx.x;
    double v = 0.0;
    if (k < kLive) {
#pragma unroll
      for (int t = 0; t < kRedParts; ++t) v += sh_part[k][t];
    }
    partials[(size_t)k * G + chunk] = v;
  }
}

#if MA_WGTIMES   // debug variant only (tools/wgtimeline.py): start/end clock and placement of every workgroup
__device__ unsigned long long g_wgt[64][4096][2];
__device__ uint32_t g_wghw[64][4096][2];
__device__ uint32_t g_wgph[64][4096][4];   // wave 0: ticks from the start to: search begins, 3x3 block done, search done, sums formed
__device__ unsigned long long g_rs[64][20][5];   // k_reduce_solve, per iteration and workgroup: entry, sum stored, ticket back, 19 sums read, solve done
#endif

// Waves per SIMD the register allocator must make room for: 8 (<= 64 VGPRs) for the search and the list launch — every
// build that let the search cross 64 VGPRs lost 5-10 % — and whatever it takes for the list-writing launch (2 of an
// alignment's 50 launches), the work-counting instantiations and ma_nn's (forced to 64 VGPRs they spill).
// MA_NN_WAVES overrides all of them (A/B).
#ifdef MA_NN_WAVES
#define MA_NN_ATTR __attribute__((amdgpu_waves_per_eu(MA_NN_WAVES, MA_NN_WAVES)))
#else
#define MA_NN_ATTR __attribute__((amdgpu_waves_per_eu((MODE == 0 && LISTS != 1 && !COUNT) ? 8 : 4, 8)))
#endif
// LISTS (compile time, so that the plain search launch keeps its 63 VGPRs — a run-time switch between the three bodies
// cost 6 VGPRs, a wavefront per SIMD and 10 % of every search launch): 0 search; 1 search and write the candidate lists;
// 2 read the lists.  The host picks the kernel per iteration from the step sizes the solve reports (ma_icp).
template <bool COUNT, int MODE, int LISTS>
// (Argument order: the first 16 dwords are preloaded into scalar registers at dispatch — csrc/Makefile — so the state,
// the source cloud, the remembered matches and the schedule, which the first instructions of a workgroup need, come
// first; the level set, 200 bytes by value, is fetched while the first loads are in flight.)
__global__ __launch_bounds__(kNNBlock) MA_NN_ATTR void k_nn_accum(
    const DevState* __restrict__ st, const float4* __restrict__ src, const uint32_t n, const int warm_hint,
    uint32_t* __restrict__ prev, const Sched sched, double* __restrict__ partials, const float best2_init,
    const LevelSet S, const uint32_t* __restrict__ perm0, uint32_t* __restrict__ nn_idx,
    float* __restrict__ nn_d2, const ListCtl* __restrict__ lc, const ListPlanes lp) {
  if (MODE == 0 && st->done) return;      // a stopped loop: nothing to do
  constexpr int lmode = MODE == 0 ? LISTS : 0;
  const unsigned long long wg_start = wall_clock64();
  constexpr int kNW = kNNBlock / kWave;
  const uint32_t G = gridDim.x;
  // iteration 0 of a run: no schedule yet, and prev[] holds another run's matches.  The loop knows which iteration it
  // enqueues and says so (warm_hint 0 / 1: one dependent scalar load less in front of the first data load); the stepping
  // entry points leave it to the state (-1)
  const bool warm = warm_hint >= 0 ? warm_hint != 0 : st->iter > 0;
  // The chunk of this workgroup: XCD segments in longest-first order.  List launches keep the mapping of the launch that
  // wrote the lists (their work is uniform, the order itself buys them nothing, but chunks in plain launch order — every
  // XCD touching gold lines from all over the map — cost 2.5 us of a 23-us launch, and segments in cloud order without
  // the schedule read measured 0.4 us slower than with it: profiles/r04_variants_ab.txt).
  const uint32_t chunk = xcd_chunk(blockIdx.x, sched, sched.order != nullptr && warm);
  const uint32_t i = chunk * kNNBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  // per-wavefront LDS: the work queues during the search, the group sums afterwards
  union WaveSearch {
    WaveLds q;
    double v[kNAcc][16];
  };
  struct WaveSums {            // a list launch has no work queues
    double v[kNAcc][16];
  };
  using WaveShared = typename std::conditional<lmode == 2, WaveSums, WaveSearch>::type;
  __shared__ WaveShared sh_w[kNW];

  const bool has = i < n;
  // the point and its fp64 image are formed twice, before the search (only the fp32 image
  // is kept) and again after it: ten registers that the search does not have to carry
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  double Px = 0.0, Py = 0.0, Pz = 0.0;
  auto image = [&](const uint32_t at) {
    s = src[at];
    point_image(st, s, Px, Py, Pz);
  };
  float px = 0.f, py = 0.f, pz = 0.f, best2 = best2_init;
  uint32_t bestj = kNoMatch, nrows = 0, ncand = 0;
  // list launch: the query's list (first plane, or the two inline entries) and its radius, issued beside the source
  // point (issuing them ahead of the `done` test as well changed nothing: profiles/r04_variants_ab.txt)
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
  uint4 lb0 = make_uint4(0u, 0u, 0u, 0u);
  float Lq = 0.f;
  if constexpr (lmode == 2) {
    if (has) {
      if (kListInline > 0) { e0 = lp.a0[i]; e1 = lp.a1[i]; }
      else lb0 = reinterpret_cast<const uint4*>(lp.b0)[i];           // without inline entries the first plane is read by every query
      Lq = lp.rad[i];
    }
  }
  if (has) {
    image(i);
    px = (float)Px; py = (float)Py; pz = (float)Pz;
    if (prev && warm && lmode != 2) {
      // temporal coherence: last iteration's match bounds the search radius from the start
      const uint32_t pj = prev[i];
      if (pj != kNoMatch) {
        const float3 c = ld_xyz(S.gold[0], pj);
        const float d2 = dist2(px, py, pz, c.x, c.y, c.z);     // (the one association: a remembered match must not tie-break on a rounding of its own)
        if (COUNT) ncand += 1u;
        if (d2 < best2) { best2 = d2; bestj = pj; }
      }
    }
  }
  uint32_t ndismissed = 0;
#if MA_WGTIMES
  const unsigned long long dbg_t1 = wall_clock64();
#endif
  float wqx = 0.f, wqy = 0.f, wqz = 0.f;      // list launch: the winner's coordinates, carried with the minimum
  if constexpr (lmode == 2) {
    // ---- list launch: the nearest of the (at most 8) gold points on the query's list IS the nearest neighbour
    // while (its distance, capped at the gate) + (how far the query has moved since the list was written) stays
    // inside the radius the list is complete for; a query that fails the check searches (per-lane, exact).
    // Streams: the source point, the two inline entries and the radius — 52 bytes per query, every one of them
    // coalesced; only a list of more than two entries reads positions and gathers.  No remembered match is read or
    // written: the tie rule needs none, and a later search launch may start from an older one (any gold point is a
    // valid first candidate).
    if (has) {
      const bool finite = isfinite(px) && isfinite(py) && isfinite(pz);
      constexpr uint32_t kIn = (uint32_t)kListInline;
      const uint32_t nlist = __float_as_uint(Lq) & 0xFu;     // entries on the list
      // how far this query has moved since its list was written: (R - Rb) s + (t - tb), wave-uniform differences
      float dRf[9], dtf[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) dRf[k] = (float)(st->R[k] - lc->Rb[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) dtf[k] = (float)(st->t[k] - lc->tb[k]);
      const float mx = dRf[0] * s.x + dRf[1] * s.y + dRf[2] * s.z + dtf[0];
      const float my = dRf[3] * s.x + dRf[4] * s.y + dRf[5] * s.z + dtf[1];
      const float mz = dRf[6] * s.x + dRf[7] * s.y + dRf[8] * s.z + dtf[2];
      const float moved = fast_sqrt(mx * mx + my * my + mz * mz) + lc->slack;
      unsigned long long key = pack_best(best2, kNoMatch);
      auto consider = [&](const float qx, const float qy, const float qz, const uint32_t pos) {
        const float d = dist2(px, py, pz, qx, qy, qz);
        const unsigned long long ku = pack_best(d, pos);
        const bool take = (d < best2_init) & (ku < key);      // the tie rule: smallest (distance, position); selects, not branches
        key = take ? ku : key;
        wqx = take ? qx : wqx; wqy = take ? qy : wqy; wqz = take ? qz : wqz;
      };
      if (kIn > 0u) {
        consider(e0.x, e0.y, e0.z, __float_as_uint(e0.w));
        consider(e1.x, e1.y, e1.z, __float_as_uint(e1.w));
      }
      if (kIn == 0u || nlist > kIn) {
        if (kIn > 0u) lb0 = reinterpret_cast<const uint4*>(lp.b0)[i];
        const uint32_t idx[4] = {lb0.x, lb0.y, lb0.z, lb0.w};
        float3 q[4];
        // (all four, whatever the list holds — free slots repeat the match, an L1 hit: gathering only the entries held
        // is four exec-masked loads and measured 2 us slower per launch, profiles/r04_variants_ab.txt)
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = ld_xyz(S.gold[0], idx[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) consider(q[u].x, q[u].y, q[u].z, idx[u]);
        if (nlist > kIn + 4u) {
          const uint4 lb1 = reinterpret_cast<const uint4*>(lp.b1)[i];
          const uint32_t idx1[4] = {lb1.x, lb1.y, kIn == 0u ? lb1.z : lb1.x, kIn == 0u ? lb1.w : lb1.y};     // (two in use beside inline entries)
          float3 r[4];
#pragma unroll
          for (int u = 0; u < (kIn == 0u ? 4 : 2); ++u) r[u] = ld_xyz(S.gold[0], idx1[u]);
#pragma unroll
          for (int u = 0; u < (kIn == 0u ? 4 : 2); ++u) consider(r[u].x, r[u].y, r[u].z, idx1[u]);
        }
      }
      best2 = __uint_as_float((uint32_t)(key >> 32));
      bestj = (uint32_t)key;
      // work counters of a list launch: candidates = entries on the list; 4-byte words = the positions read from the
      // planes b0 / b1 (each is followed by one 12-byte gather), plus what a failed check's search reads
      if (COUNT) { ncand += nlist; nrows += nlist > kIn + 4u ? kListLen - kIn : ((kIn == 0u || nlist > kIn) ? 4u : 0u); }
      if (!finite) {
        bestj = kNoMatch;
      } else if (!(fast_sqrt(best2) + moved < Lq)) {
        nn_find<COUNT>(px, py, pz, S, best2, bestj, nrows, ncand);
        if (bestj != kNoMatch) { const float3 q = ld_xyz(S.gold[0], bestj); wqx = q.x; wqy = q.y; wqz = q.z; }
      }
    }
  } else if constexpr (lmode == 1) {
    wave_search<COUNT, true>(sh_w[wv].q, lane, has, px, py, pz, S, best2, bestj, nrows, ncand, ndismissed, lc->skin,
                             lp, i - (uint32_t)lane);
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the transform these lists belong to
      ListCtl* __restrict__ w = const_cast<ListCtl*>(lc);
#pragma unroll
      for (int k = 0; k < 9; ++k) w->Rb[k] = st->R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) w->tb[k] = st->t[k];
      w->have = 1;
    }
  } else {
    wave_search<COUNT, false>(sh_w[wv].q, lane, has, px, py, pz, S, best2, bestj, nrows, ncand, ndismissed);
  }
#if MA_WGTIMES
  unsigned long long dbg_t2 = 0;
  if constexpr (lmode != 2) dbg_t2 = sh_w[wv].q.dbg_round0;
  const unsigned long long dbg_t3 = wall_clock64();
  unsigned long long dbg_t4 = 0;
#endif

  double acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) acc[k] = 0.0;
  if (has) {
#if MA_DEBUG_FETCH      // debug build (tools/tie_debug.py): list launches leave their matches behind too
    if (prev) prev[i] = bestj;
#else
    if (prev && lmode != 2) prev[i] = bestj;
#endif
    if (MODE == 0 && lmode != 2) {       // (a list launch has no search in between: its point stays in registers)
      uint32_t again = i;
      asm volatile("" : "+v"(again));   // opaque copy of the index: the compiler must not keep s, P live instead
      image(again);
    }
    if (MODE == 1) {
      nn_idx[i] = bestj != kNoMatch ? perm0[bestj] : kNoMatch;
      nn_d2[i] = bestj != kNoMatch ? best2 : INFINITY;
    }
    if (MODE == 0 && bestj != kNoMatch) {
      float3 q;
      if constexpr (lmode == 2) q = make_float3(wqx, wqy, wqz);     // carried with the minimum: the very bits of the gold array
      else q = ld_xyz(S.gold[0], bestj);           // the winner's coordinates (level 0 holds every point)
      pair_terms(acc, (double)s.w, Px, Py, Pz, q);
    }
    if (COUNT) { acc[17] = (double)nrows; acc[18] = (double)ncand; }
  }
  if (MODE == 1) return;  // ma_nn: per-point outputs only
#if MA_WGTIMES
  asm volatile("" :: "v"(acc[16]), "v"(acc[0]));
  dbg_t4 = wall_clock64();
#endif

  // workgroup reduction of the live accumulators -> one partial per accumulator and chunk (reduce_chunk)
  __shared__ double sh_part[kNAcc][kRedParts];
  __shared__ uint32_t sh_dismissed[kNW];
  if (lane == 0) sh_dismissed[wv] = ndismissed;
  reduce_chunk<COUNT>(acc, sh_w, sh_part, partials, G, chunk, lane, wv);
  // Cost for the next launch's schedule: this workgroup's duration, plus what its dismissed
  // points would have cost had they walked their rings (~0.5 us per point of one wavefront;
  // four wavefronts run side by side) — as the clouds approach each other such points come
  // within reach from one iteration to the next, and a segment that was cheap only because
  // it was dismissed must not be scheduled last.
  if (threadIdx.x == 0 && sched.wg_cost) {
    uint32_t dismissed = 0;
#pragma unroll
    for (int w = 0; w < kNW; ++w) dismissed += sh_dismissed[w];
    sched.wg_cost[chunk] = (uint32_t)(wall_clock64() - wg_start) + (48u / kNW) * dismissed;
  }
#if MA_WGTIMES
  if (threadIdx.x == 0 && MODE == 0 && st->iter < 64 && blockIdx.x < 4096) {
    g_wgt[st->iter][blockIdx.x][0] = wg_start;
    g_wgt[st->iter][blockIdx.x][1] = wall_clock64();
    g_wghw[st->iter][blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    g_wghw[st->iter][blockIdx.x][1] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) | (chunk << 4);   // XCC_ID, chunk
    g_wgph[st->iter][blockIdx.x][0] = (uint32_t)(dbg_t1 - wg_start);
    g_wgph[st->iter][blockIdx.x][1] = (uint32_t)(dbg_t2 - wg_start);
    g_wgph[st->iter][blockIdx.x][2] = (uint32_t)(dbg_t3 - wg_start);
    g_wgph[st->iter][blockIdx.x][3] = (uint32_t)(dbg_t4 - wg_start);
  }
#endif
}

// Longest-first schedule, ranking step (one workgroup): order[r] = the scheduling group with
// the r-th largest cost recorded by the search kernel, ties by index (rank by counting).
__device__ __forceinline__ void rank_groups(const int G, const uint32_t* __restrict__ wg_cost,
                                            uint32_t* __restrict__ order, const uint32_t gshift, const int done) {
  // key = (cost, reversed index): unique, so the rank is a plain count of larger keys.
  // ngroups = G / (16 << gshift) is ANY integer <= kMaxGroups once gshift >= 2 (sources above
  // ~4.2M points: 306 groups at 5M), so the array is padded with zeros (smaller than or equal
  // to every real key, and only strictly larger keys are counted) up to the next multiple of
  // 4 and the 16-byte compare loop runs over the padded length: every key is compared.
  __shared__ __attribute__((aligned(16))) uint32_t key[kMaxGroups + 4];
  const uint32_t per4 = (kXcdSeg << gshift) / 4u, ngroups = (uint32_t)G / (kXcdSeg << gshift);   // <= kMaxGroups
  const uint32_t n4 = (ngroups + 3u) / 4u;
  const uint4* __restrict__ cost4 = reinterpret_cast<const uint4*>(wg_cost);
  for (uint32_t g = threadIdx.x; g < 4u * n4; g += kBlock) {
    uint32_t c = 0;
    if (g < ngroups) {
      for (uint32_t t = 0; t < per4; t += 4u) {     // per4 is a multiple of 4: four 16-byte reads in flight
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = cost4[g * per4 + t + (uint32_t)u];
#pragma unroll
        for (int u = 0; u < 4; ++u) c += (q[u].x >> gshift) + (q[u].y >> gshift) + (q[u].z >> gshift) + (q[u].w >> gshift);
      }
    }
    key[g] = g < ngroups ? (min(c, 0x7FFFFFu) << 9) | ((uint32_t)kMaxGroups - 1u - g) : 0u;
  }
  if (done) return;
  __syncthreads();
  const uint4* key4 = reinterpret_cast<const uint4*>(key);
  for (uint32_t g = threadIdx.x; g < ngroups; g += kBlock) {
    const uint32_t mine = key[g];
    uint32_t rank = 0;
#pragma unroll 4
    for (uint32_t j = 0; j < n4; ++j) {
      const uint4 kj = key4[j];
      rank += (kj.x > mine ? 1u : 0u) + (kj.y > mine ? 1u : 0u) + (kj.z > mine ? 1u : 0u) + (kj.w > mine ? 1u : 0u);
    }
    order[rank] = g;
  }
}

// a5 (final sum): one workgroup PER ACCUMULATOR sums that accumulator's G partials — 19
// CUs pull 31 KB each instead of one CU pulling 0.6 MB (one CU's load bandwidth, not the
// arithmetic, is what a single-workgroup reduction waits for).  Fixed assignment partial ->
// thread, fixed order within a thread, DPP + LDS tree: deterministic.
__global__ __launch_bounds__(kBlock) void k_reduce19(const double* __restrict__ partials, const int G,
                                                     const DevState* __restrict__ st,
                                                     double* __restrict__ acc_out,
                                                     const uint32_t* __restrict__ wg_cost,
                                                     uint32_t* __restrict__ order, const uint32_t gshift) {
  if (blockIdx.x == kNAcc) {   // launched only with a schedule (ma_icp_sharded): rank the groups for the next search
    rank_groups(G, wg_cost, order, gshift, st->done);
    return;
  }
  const int k = blockIdx.x;
  if (st->done) {
    // a stopped loop: the search launch returned at once and left old partials behind.  The sharded loop keeps
    // issuing its all-reduces (every rank must issue the same number), so the sums are ZEROED rather than left
    // stale: in place, they would be multiplied by the number of ranks in every remaining iteration.
    if (threadIdx.x == 0) acc_out[k] = 0.0;
    return;
  }
  const double* __restrict__ p = partials + (size_t)k * G;
  double v = 0.0;
  for (int b0 = threadIdx.x; b0 < G; b0 += 8 * kBlock) {
    double u[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {     // eight independent loads in flight per thread
      const int b = b0 + t * kBlock;
      u[t] = b < G ? p[b] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) v += u[t];
  }
  __shared__ double red[kBlock / kWave];
  v = wave_sum(v);
  if ((threadIdx.x & (kWave - 1)) == kSumLane) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) acc_out[k] = ((red[0] + red[1]) + red[2]) + red[3];
}

// a6 + a7 for one thread: Kabsch step from the 19 reduced values, transform composition,
// convergence test, trace row.  (Kept out of the search kernel: its fp64 Jacobi code needs
// ~110 VGPRs and would halve the occupancy of the search.)
// The solve is ONE lane, and a lone wavefront gets one vector instruction through every 7.5-9.5 cycles whatever it
// depends on (profiles/r04_valu_issue_rate.txt, W = 1): the solve's time is its instruction count.  Hence the quaternion
// form of the Kabsch step (kabsch.h: ~200 instructions for the SVD's ~500), the rmse from the hardware seeds (an IEEE
// divide and square root are ~55 instructions, these 9, to an ulp or two), and in k_reduce_solve the 19 summing workgroups
// writing their own sums into the trace row (SUMS_IN_ROW false) instead of this lane writing all of them.
template <bool SUMS_IN_ROW = true>
__device__ __forceinline__ void solve_step(const double (&a)[kNAcc], DevState s0, DevState* __restrict__ st,
                                           double* __restrict__ trace, const int trace_cap, const double tol,
                                           int* __restrict__ host_done, DevState* __restrict__ host_mirror = nullptr,
                                           StateSlot* __restrict__ slot = nullptr) {
  const int it = s0.iter;
  const double W = a[0];
  const double rmse = W > 0.0 ? (a[16] > 0.0 ? ma::sqrt_f64(a[16] * ma::inv_f64(W)) : 0.0) : NAN;
  if (it < trace_cap) {
    double* row = trace + (size_t)it * MA_TRACE_STRIDE;
    if (SUMS_IN_ROW) {
#pragma unroll
      for (int k = 0; k < ma::kAccLen; ++k) row[k] = a[k];
      row[18] = a[17]; row[19] = a[18];
    }
    row[17] = rmse;
  }
  s0.iter = it + 1;
  s0.last_rmse = rmse;
  s0.last_W = W;
  double dR[9], dt[3];
#if MA_WGTIMES
  if (it < 64) g_rs[it][19][1] = wall_clock64();
#endif
  const int degenerate = ma::kabsch_from_acc(a, dR, dt);
#if MA_WGTIMES
  asm volatile("" :: "v"(dR[0]), "v"(dt[0]));
  if (it < 64) g_rs[it][19][2] = wall_clock64();
#endif
  if (degenerate) {
    s0.degenerate = 1;
    s0.done = 1;
    *st = s0;
    if (host_mirror) *host_mirror = s0;
    if (host_done) *host_done = s0.iter;     // iterations run when the loop stopped (never 0)
    if (slot) {                              // every iteration up to and including the one that stops the loop writes its slot
      *const_cast<unsigned long long*>(&slot->chk) = slot_check(s0.R, s0.t, s0.iter);
      *const_cast<int*>(&slot->seq) = s0.iter;
    }
    return;
  }
  double Rn[9], tn[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rn[3 * r + c] = dR[3 * r] * s0.R[c] + dR[3 * r + 1] * s0.R[3 + c] + dR[3 * r + 2] * s0.R[6 + c];
    tn[r] = dR[3 * r] * s0.t[0] + dR[3 * r + 1] * s0.t[1] + dR[3 * r + 2] * s0.t[2] + dt[r];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) s0.R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) s0.t[i] = tn[i];
  if (it >= 1 && fabs(rmse - s0.prev_rmse) < tol) {
    s0.converged = 1;
    s0.done = 1;
    // pinned host word: lets ma_icp stop enqueueing launches that would only find `done` set; its value is the number
    // of iterations run, so that every rank of a sharded loop stops enqueueing at the same iteration
    if (host_done) *host_done = s0.iter;
  }
  s0.prev_rmse = rmse;
  *st = s0;
  // pinned host copy of the state (ma_icp): written with every solve, read by the host after its one
  // synchronisation — no device-to-host copy at the end of the loop (posted writes: the solve does not wait for them)
  if (host_mirror) *host_mirror = s0;
  if (slot) {      // candidate lists: host_mirror is this iteration's own pinned slot, and the host's planner reads it (ma_icp)
    // plain stores (the fields are volatile for the HOST's reads): a volatile store makes the wavefront wait for the
    // posted write to be acknowledged, 0.7 us each; the checksum makes their order irrelevant
    *const_cast<unsigned long long*>(&slot->chk) = slot_check(s0.R, s0.t, s0.iter);
    *const_cast<int*>(&slot->seq) = s0.iter;
  }
#if MA_WGTIMES
  if (it < 64) g_rs[it][19][3] = wall_clock64();
#endif
}

// The solve alone (sharded-source stepping: the 17 sums come back from the all-reduce).
__global__ __launch_bounds__(kWave) void k_solve(const double* __restrict__ acc_in, DevState* __restrict__ st,
                                                 double* __restrict__ trace, const int trace_cap,
                                                 const double tol, int* __restrict__ host_done,
                                                 DevState* __restrict__ host_mirror, StateSlot* __restrict__ slot) {
  if (st->done || threadIdx.x != 0) return;
  double a[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) a[k] = acc_in[k];
  solve_step(a, *st, st, trace, trace_cap, tol, host_done, host_mirror, slot);
}

// a5 (final sum) + a6 + a7 in ONE launch: the 19 reducing workgroups of k_reduce19, and the
// one whose arrival ticket comes back last runs the solve (saves a dependent launch, ~4 us
// of every iteration).  Hand-off across CUs / XCDs in the memory-model form: each sum is stored
// with an agent-scope atomic store, the same lane takes its ticket with an agent-scope ACQ_REL
// fetch-add (release: the sum is visible before the ticket; acquire: the last arriver sees all
// nineteen), and the last arriver reads the sums with agent-scope atomic loads.  (Round 1 used
// relaxed atomics + explicit s_waitcnt, which relied on gfx950's write-through behaviour; the
// fenced form measured the same: 2.366 vs 2.362 ms per alignment.)
//
// One more workgroup (blockIdx kNAcc, beside the sums, not part of the hand-off) ranks the
// scheduling groups by the cost the search kernel just recorded: order[r] = the group with
// the r-th largest cost, ties by index (rank by counting, <= 512 groups).
__global__ __launch_bounds__(kBlock) void k_reduce_solve(const double* __restrict__ partials, const int G,
                                                         DevState* __restrict__ st, double* __restrict__ acc_out,
                                                         uint32_t* __restrict__ ticket, double* __restrict__ trace,
                                                         const int trace_cap, const double tol,
                                                         const uint32_t* __restrict__ wg_cost,
                                                         uint32_t* __restrict__ order, const uint32_t gshift,
                                                         int* __restrict__ host_done, DevState* __restrict__ host_mirror,
                                                         StateSlot* __restrict__ slot) {
  // `done` is written only by the last arriver, after every workgroup has read it.  It is
  // tested AFTER the first loads have been issued: every dependent memory round trip of
  // this small kernel is ~1 us of the iteration.
  const int done = st->done;
#if MA_WGTIMES
  const int dbg_it = st->iter;
  if (threadIdx.x == 0 && dbg_it < 64) g_rs[dbg_it][blockIdx.x][0] = wall_clock64();
#endif
  if (blockIdx.x == kNAcc) {
    rank_groups(G, wg_cost, order, gshift, done);
#if MA_WGTIMES
    if (threadIdx.x == 0 && dbg_it < 64) g_rs[dbg_it][blockIdx.x][4] = wall_clock64();
#endif
    return;
  }
  const int k = blockIdx.x;
  const int it_now = st->iter;
  DevState s0;
  if (threadIdx.x == 0) s0 = *st;   // whoever arrives last will need it: read it now, not after the hand-off
  const double* __restrict__ p = partials + (size_t)k * G;
  double v = 0.0;
  for (int b0 = threadIdx.x; b0 < G; b0 += 16 * kBlock) {
    double u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {    // sixteen independent loads in flight per thread
      const int b = b0 + t * kBlock;
      u[t] = b < G ? p[b] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) v += u[t];
  }
  if (done) return;
  __shared__ double red[kBlock / kWave];
  v = wave_sum(v);
  if ((threadIdx.x & (kWave - 1)) == kSumLane) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double sum = ((red[0] + red[1]) + red[2]) + red[3];
  __hip_atomic_store(&acc_out[k], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // this accumulator's entry of the iteration's trace row (columns 0-16 the sums, 17 the rmse — the solve's —, 18 and 19
  // the two work counters)
  if (it_now < trace_cap) trace[(size_t)it_now * MA_TRACE_STRIDE + (k < ma::kAccLen ? k : k + 1)] = sum;
#if MA_WGTIMES
  if (dbg_it < 64) g_rs[dbg_it][k][1] = wall_clock64();
#endif
  // memory-model form: release on the ticket, acquire by the last arriver
  const uint32_t arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#if MA_WGTIMES
  if (dbg_it < 64) g_rs[dbg_it][k][2] = wall_clock64();
#endif
  if (arrived != (uint32_t)(kNAcc - 1)) return;
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  double a[kNAcc];
#pragma unroll
  for (int t = 0; t < kNAcc; ++t) a[t] = __hip_atomic_load(&acc_out[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if MA_WGTIMES
  asm volatile("" :: "v"(a[0]), "v"(a[18]));
  if (dbg_it < 64) g_rs[dbg_it][k][3] = wall_clock64();
#endif
  solve_step<false>(a, s0, st, trace, trace_cap, tol, host_done, host_mirror, slot);
#if MA_WGTIMES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (dbg_it < 64) g_rs[dbg_it][k][4] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------
// a8: depth -> world points, order-preserving compaction
// ---------------------------------------------------------------------------------------
// pixels per thread of k_unproject: 1 / 2 / 4 / 8 -> 11.1 / 11.4 / 13.9 / 14.2 us for a 1280 x 960 frame and 160 / 163 / 196 us
// for 8192 x 4096 (rocprofv3; more, lighter threads hide the load latency of a kernel that is one load and a few
// stores long); non-temporal stores (four dword stores per point): 24 us.  2 keeps the per-workgroup count array half as long as 1.
#ifndef MA_UNPROJ_PX
#define MA_UNPROJ_PX 2
#endif
constexpr int kPxPerThread = MA_UNPROJ_PX;
constexpr int kPxPerBlock = kBlock * kPxPerThread;

template <typename T>
__device__ __forceinline__ double depth_z(const T* __restrict__ depth, uint32_t i, double inv_scale) {
  return (double)depth[i] * inv_scale;
}

struct Pinhole {
  double fx_inv, fy_inv, cx, cy, inv_scale;
  double T[12];  // camera->world rows 0..2, translation already minus the centring origin
};

// ONE streaming pass, no exchange between workgroups (round 3).  Every pixel is written IN PLACE:
// slot i of the output holds the point of pixel i, or a hole — x = quiet NaN, w = 0 — where the
// depth is <= 0 / NaN / inf.  The search kernel skips non-finite queries (wave_search: `finite`),
// the map builder's voxel sort sends holes behind every real key, and the calls that hand points
// back to the caller (ma_get_source, ma_nn, ...) compact lazily through a rank map (src_map).
// The only other output is one count of valid pixels per workgroup, written with a plain store and
// added up by the host: an atomic on one address completes every ~200 ns on this chip
// (profiles/r03_variants_ab.txt, call 1), so 1200 of them would cost ten times the kernel.
// Rounds 1-2 compacted in the kernel (order-preserving, first with a scan between two passes, then
// single-pass with a decoupled look-back / all-predecessors exchange): 18-24 us for a 1280 x 960
// frame, all of it waiting for other workgroups' counts (~2 us per coherent load, O(grid^2) of
// them) to keep 99.9999 % valid pixels where they already were.
// Algorithmic bytes: w h sizeof(depth) read + 16 w h written (24.6 MB for configs[2]).
constexpr uint32_t kHoleBits = 0x7FC00000u;      // quiet NaN

template <typename T>
__global__ __launch_bounds__(kBlock) void k_unproject(const T* __restrict__ depth, const int width,
                                                      const uint32_t npx, const Pinhole cam,
                                                      uint32_t* __restrict__ counts, float4* __restrict__ out) {
  const uint32_t first = blockIdx.x * kPxPerBlock + threadIdx.x * kPxPerThread;
  double z[kPxPerThread];
  if (first + kPxPerThread <= npx && (reinterpret_cast<uintptr_t>(depth + first) & (sizeof(T) * kPxPerThread - 1)) == 0) {
    // the thread's pixels in one load
    struct alignas(sizeof(T) * kPxPerThread) Quad { T v[kPxPerThread]; };
    const Quad q = *reinterpret_cast<const Quad*>(depth + first);
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) z[k] = (double)q.v[k] * cam.inv_scale;
  } else {
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) z[k] = first + k < npx ? depth_z(depth, first + k, cam.inv_scale) : 0.0;
  }
  uint32_t cnt = 0;
  // (u, v) of the first pixel by one division; the others step along the row
  uint32_t u = first % (uint32_t)width, v = first / (uint32_t)width;
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const uint32_t i = first + k;
    const bool ok = isfinite(z[k]) && z[k] > 0.0;
    if (i < npx) {
      float4 p;
      if (ok) {
        const double X = ((double)u - cam.cx) * z[k] * cam.fx_inv;
        const double Y = ((double)v - cam.cy) * z[k] * cam.fy_inv;
        p.x = (float)(cam.T[0] * X + cam.T[1] * Y + cam.T[2] * z[k] + cam.T[3]);
        p.y = (float)(cam.T[4] * X + cam.T[5] * Y + cam.T[6] * z[k] + cam.T[7]);
        p.z = (float)(cam.T[8] * X + cam.T[9] * Y + cam.T[10] * z[k] + cam.T[11]);
        p.w = 1.0f;
        ++cnt;
      } else {
        p = make_float4(__uint_as_float(kHoleBits), 0.f, 0.f, 0.f);
      }
      out[i] = p;
    }
    if (++u == (uint32_t)width) { u = 0; ++v; }
  }
  __shared__ uint32_t wsum[kBlock / kWave];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += wsum[w];
    counts[blockIdx.x] = t;
  }
}

// Lazy compaction of a cloud with holes: map[i] = number of real points in front of slot i (its
// index in the caller's view: valid pixels in row-major order), kNoMatch for a hole.  flags -> scan
// (hipCUB) -> this kernel; built only when a call has to hand points or per-point results back.
__global__ __launch_bounds__(kBlock) void k_hole_flags(const float4* __restrict__ pts, uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) flags[i] = isfinite(pts[i].x) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_rank_map(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan,
                                                     uint32_t n, uint32_t* __restrict__ map) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) map[i] = flags[i] ? scan[i] : kNoMatch;
}

// N4: voxel keys of the accumulated map points.  The voxel of a point is floor(p / voxel) per axis
// (anchored at the map origin); the key numbers the voxels of the map's bounding box z-major,
// (vz * dims.y + vy) * dims.x + vx with indices relative to the box's first voxel, so that the
// radix sort only has to look at the ~30 bits a real map uses instead of 63.
using ma::VoxelBox;              // csrc/host_math.h (its arithmetic is host code, tested under sanitizers)
__global__ __launch_bounds__(kBlock) void k_voxel_keys(const float4* __restrict__ pts, uint32_t n, float inv_voxel,
                                                       const VoxelBox box, const unsigned long long hole_key,
                                                       unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  if (!isfinite(p.x)) {            // a hole of the in-place unprojection: sorts behind every voxel of the box
    keys[i] = hole_key;
    vals[i] = i;
    return;
  }
  auto q = [&](float v, int a) -> unsigned long long {
    long long c = (long long)floorf(v * inv_voxel) - box.q0[a];
    c = c < 0 ? 0 : (c > (long long)box.d[a] - 1 ? (long long)box.d[a] - 1 : c);
    return (unsigned long long)c;
  };
  keys[i] = (q(p.z, 2) * box.d[1] + q(p.y, 1)) * box.d[0] + q(p.x, 0);
  vals[i] = i;
}

__global__ __launch_bounds__(kBlock) void k_voxel_heads(const unsigned long long* __restrict__ keys, uint32_t n,
                                                        uint32_t* __restrict__ flags /*[n+1]*/) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i > n) return;
  flags[i] = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}

// fp64 centroid per voxel, in two steps whose work per thread is bounded (a thread per voxel
// walking its whole run took 3.3 ms of a 7 ms map build: a wavefront waits for its longest run).
// Step 1, one thread per kVoxRun consecutive sorted points: the points before the block's first
// voxel head belong to a voxel that started in an earlier block -> pre[block]; every head opens a
// piece that ends at the next head or at the block's end -> acc[voxel] (written exactly once, by
// the block that holds the head).  Step 2, one thread per voxel: acc[voxel] + pre[] of the blocks
// its run continues into, in block order.  The summation order depends only on the sorted order
// of the points (stable sort = insertion order within a voxel): deterministic.
constexpr uint32_t kVoxRun = 16;
struct alignas(32) VoxelSum { double x, y, z, n; };
__global__ __launch_bounds__(kBlock) void k_voxel_pieces(const uint32_t* __restrict__ vals,
                                                         const uint32_t* __restrict__ flags,
                                                         const uint32_t* __restrict__ segid, const uint32_t n,
                                                         const float4* __restrict__ pts, VoxelSum* __restrict__ acc,
                                                         VoxelSum* __restrict__ pre, uint32_t* __restrict__ heads) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  const uint64_t i0 = (uint64_t)t * kVoxRun;
  if (i0 >= n) return;
  const uint32_t cnt = (uint32_t)min((uint64_t)kVoxRun, (uint64_t)n - i0);
  VoxelSum s = {0.0, 0.0, 0.0, 0.0};
  long long cur = -1;   // voxel of the piece being summed; -1: the block's prefix piece
  auto close = [&]() {
    if (cur < 0) pre[t] = s; else acc[cur] = s;
  };
  for (uint32_t k0 = 0; k0 < cnt; k0 += 4u) {
    uint32_t v[4], f[4];
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = k0 + u < cnt;
      v[u] = in ? vals[i0 + k0 + u] : 0u;
      f[u] = in ? flags[i0 + k0 + u] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = pts[v[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k0 + u >= cnt) break;
      if (f[u]) {
        close();
        cur = (long long)segid[i0 + k0 + u];
        heads[cur] = (uint32_t)(i0 + k0 + u);
        s = {0.0, 0.0, 0.0, 0.0};
      }
      s.x += p[u].x; s.y += p[u].y; s.z += p[u].z; s.n += 1.0;
    }
  }
  close();
  if (i0 + cnt == n) heads[segid[n]] = n;   // sentinel: one past the last voxel's run (segid[n] = number of voxels)
}

__global__ __launch_bounds__(kBlock) void k_voxel_finish(const VoxelSum* __restrict__ acc, const VoxelSum* __restrict__ pre,
                                                         const uint32_t* __restrict__ heads, const uint32_t nvox,
                                                         const double o0, const double o1, const double o2,
                                                         float* __restrict__ xyz) {
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= nvox) return;
  VoxelSum s = acc[v];
  const uint32_t t0 = heads[v] / kVoxRun, t1 = (heads[v + 1] - 1u) / kVoxRun;
  for (uint32_t t = t0 + 1u; t <= t1; ++t) {
    const VoxelSum q = pre[t];
    s.x += q.x; s.y += q.y; s.z += q.z; s.n += q.n;
  }
  xyz[3 * (size_t)v] = (float)(s.x / s.n + o0);
  xyz[3 * (size_t)v + 1] = (float)(s.y / s.n + o1);
  xyz[3 * (size_t)v + 2] = (float)(s.z / s.n + o2);
}

// a10, depth rule 1 (MA_Z_MEDIAN): the comment at pose_processor.py:183 says the corners' z comes "from median
// over bbox (x, y) range"; the body that would say how (Transforms._depth_to_3d) is absent.  This is that
// reading: z = median of the valid depths (finite, > 0) over the box's pixel rectangle
// [int(x1) .. int(x2)] x [int(y1) .. int(y2)] (ends included, clamped into the image), the same z for all
// four corners; an even count averages the two middle values (numpy.median).  No valid pixel -> NaN.
// One workgroup per box: radix select over the depth's bit pattern (positive floats and uint16 order
// like unsigned integers), four 8-bit passes with a 256-bin LDS histogram each; exact, deterministic.
template <typename T> __device__ __forceinline__ uint32_t depth_key(T v);
template <> __device__ __forceinline__ uint32_t depth_key<float>(float v) { return (isfinite(v) && v > 0.f) ? __float_as_uint(v) : 0u; }
template <> __device__ __forceinline__ uint32_t depth_key<uint16_t>(uint16_t v) { return (uint32_t)v; }
template <typename T> __device__ __forceinline__ double key_depth(uint32_t k);
template <> __device__ __forceinline__ double key_depth<float>(uint32_t k) { return (double)__uint_as_float(k); }
template <> __device__ __forceinline__ double key_depth<uint16_t>(uint32_t k) { return (double)k; }

template <typename T>
__global__ __launch_bounds__(kBlock) void k_box_median(const T* __restrict__ depth, const int width, const int height,
                                                       const uint32_t n_frames, const float* __restrict__ boxes,
                                                       const uint32_t* __restrict__ box_frame, const double inv_scale,
                                                       double* __restrict__ zmed) {
  const uint32_t b = blockIdx.x;
  const uint32_t f = box_frame ? min(box_frame[b], n_frames - 1u) : 0u;
  const T* __restrict__ img = depth + (size_t)f * width * height;
  const float x1 = boxes[4 * b], y1 = boxes[4 * b + 1], x2 = boxes[4 * b + 2], y2 = boxes[4 * b + 3];
  const int ua = min(max((int)x1, 0), width - 1), ub = min(max((int)x2, 0), width - 1);     // int(x), clamped: the corner pixels
  const int va = min(max((int)y1, 0), height - 1), vb = min(max((int)y2, 0), height - 1);
  const int u0 = min(ua, ub), u1 = max(ua, ub), v0 = min(va, vb), v1 = max(va, vb);
  const uint32_t rw = (uint32_t)(u1 - u0 + 1), npx = rw * (uint32_t)(v1 - v0 + 1);
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_rank, sh_count;
  uint32_t result[2] = {0u, 0u};
  uint32_t count = 0;
  for (int which = 0; which < 2; ++which) {         // lower and upper middle element (the same one for an odd count)
    uint32_t prefix = 0, mask = 0, rank = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      hist[threadIdx.x] = 0u;
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < npx; t += kBlock) {
        const uint32_t key = depth_key<T>(img[(size_t)(v0 + (int)(t / rw)) * width + (u0 + (int)(t % rw))]);
        if (key != 0u && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        if (shift == 24) {                          // first pass: every valid pixel is in the histogram
          uint32_t c = 0;
          for (int k = 0; k < 256; ++k) c += hist[k];
          sh_count = c;
          rank = c ? (which == 0 ? (c - 1u) / 2u : c / 2u) : 0u;
        } else {
          rank = sh_rank;
        }
        uint32_t cum = 0;
        int bin = 0;
        for (; bin < 255; ++bin) {
          if (cum + hist[bin] > rank) break;
          cum += hist[bin];
        }
        sh_prefix = prefix | ((uint32_t)bin << shift);
        sh_rank = rank - cum;
      }
      __syncthreads();
      prefix = sh_prefix;
      mask |= 255u << shift;
      count = sh_count;
      if (count == 0u) break;
    }
    result[which] = prefix;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    zmed[b] = count ? 0.5 * (key_depth<T>(result[0]) * inv_scale + key_depth<T>(result[1]) * inv_scale) : (double)NAN;
}

// a10: one thread per box corner.  cam.T here is the plain camera->world matrix (world frame).
// zmed == null: depth rule 0, the corner pixel's own depth (the call shape of pose_processor.py:184-196:
// _depth_to_3d(int(x), int(y), depth_image, ...)); otherwise the box's median depth (k_box_median).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lift_boxes(const T* __restrict__ depth, const int width,
                                                       const int height, const Pinhole cam,
                                                       const float* __restrict__ boxes, const uint32_t ncorners,
                                                       const double* __restrict__ zmed,
                                                       double* __restrict__ out, uint8_t* __restrict__ valid) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncorners) return;
  const uint32_t b = i >> 2, k = i & 3u;
  const float bx = boxes[4 * b + ((k == 1u || k == 2u) ? 2 : 0)];   // x1 x2 x2 x1
  const float by = boxes[4 * b + ((k >= 2u) ? 3 : 1)];               // y1 y1 y2 y2
  const int u = min(max((int)bx, 0), width - 1), v = min(max((int)by, 0), height - 1);  // int(x), int(y)
  double z = zmed ? zmed[b] : (double)depth[(size_t)v * width + u] * cam.inv_scale;
  const bool ok = isfinite(z) && z > 0.0;
  if (!ok) z = NAN;
  const double X = ((double)u - cam.cx) * z * cam.fx_inv, Y = ((double)v - cam.cy) * z * cam.fy_inv;
  out[3 * (size_t)i] = cam.T[0] * X + cam.T[1] * Y + cam.T[2] * z + cam.T[3];
  out[3 * (size_t)i + 1] = cam.T[4] * X + cam.T[5] * Y + cam.T[6] * z + cam.T[7];
  out[3 * (size_t)i + 2] = cam.T[8] * X + cam.T[9] * Y + cam.T[10] * z + cam.T[11];
  if (valid) valid[i] = ok ? 1 : 0;
}

// a10 over a batch of frames: corner i belongs to box i/4 of frame box_frame[i/4]; frames are
// stored back to back, poses as 12 doubles per frame (rows 0..2 of camera->world).  One
// launch lifts the boxes of all frames of a detector batch (configs[3]).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lift_boxes_batch(const T* __restrict__ depth, const int width,
                                                             const int height, const uint32_t n_frames,
                                                             const double fx_inv, const double fy_inv,
                                                             const double cx, const double cy, const double inv_scale,
                                                             const double* __restrict__ poses,
                                                             const float* __restrict__ boxes,
                                                             const uint32_t* __restrict__ box_frame,
                                                             const uint32_t ncorners, const double* __restrict__ zmed,
                                                             double* __restrict__ out, uint8_t* __restrict__ valid) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncorners) return;
  const uint32_t b = i >> 2, k = i & 3u;
  const uint32_t f = min(box_frame[b], n_frames - 1u);   // (validated on the host; clamp keeps the read in bounds)
  const float bx = boxes[4 * b + ((k == 1u || k == 2u) ? 2 : 0)];   // x1 x2 x2 x1
  const float by = boxes[4 * b + ((k >= 2u) ? 3 : 1)];               // y1 y1 y2 y2
  const int u = min(max((int)bx, 0), width - 1), v = min(max((int)by, 0), height - 1);  // int(x), int(y)
  double z = zmed ? zmed[b] : (double)depth[(size_t)f * width * height + (size_t)v * width + u] * inv_scale;
  const bool ok = isfinite(z) && z > 0.0;
  if (!ok) z = NAN;
  const double* __restrict__ T12 = poses + 12 * (size_t)f;
  const double X = ((double)u - cx) * z * fx_inv, Y = ((double)v - cy) * z * fy_inv;
  out[3 * (size_t)i] = T12[0] * X + T12[1] * Y + T12[2] * z + T12[3];
  out[3 * (size_t)i + 1] = T12[4] * X + T12[5] * Y + T12[6] * z + T12[7];
  out[3 * (size_t)i + 2] = T12[8] * X + T12[9] * Y + T12[10] * z + T12[11];
  if (valid) valid[i] = ok ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_uncentre(const float4* __restrict__ src, const uint32_t* __restrict__ map,
                                                     uint32_t n, double o0, double o1, double o2,
                                                     float* __restrict__ xyz) {
  const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n) return;
  const size_t i = map ? map[pos] : pos;
  if (map && map[pos] == kNoMatch) return;       // a hole
  const float4 s = src[pos];
  xyz[3 * i] = (float)((double)s.x + o0);
  xyz[3 * i + 1] = (float)((double)s.y + o1);
  xyz[3 * i + 2] = (float)((double)s.z + o2);
}

// The stored coordinates as the search sees them, exactly: world = fp32 (centred) + origin, evaluated in
// fp64 (exact: both terms are fp32 numbers), written in the CALLER's order (perm: sorted position ->
// original index; null = storage order is the caller's order).
__global__ __launch_bounds__(kBlock) void k_export_f64(const float4* __restrict__ pts, const uint32_t* __restrict__ perm,
                                                       uint32_t n, double o0, double o1, double o2,
                                                       double* __restrict__ xyz) {
  const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n) return;
  const float4 p = pts[pos];
  const size_t i = perm ? perm[pos] : pos;
  if (perm && perm[pos] == kNoMatch) return;     // a hole
  xyz[3 * i] = (double)p.x + o0;
  xyz[3 * i + 1] = (double)p.y + o1;
  xyz[3 * i + 2] = (double)p.z + o2;
}

// out[perm[pos]] = in[pos] (undo the source cell sort for ma_nn)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_unpermute(const T* __restrict__ in,
                                                      const uint32_t* __restrict__ perm, uint32_t n,
                                                      T* __restrict__ out) {
  const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n) return;
  if (perm && perm[pos] == kNoMatch) return;     // a hole
  out[perm ? perm[pos] : pos] = in[pos];
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// Device allocations of one context: counted so that callers (and the tests) can check that a
// warm context — the second and later alignments through the cached engine of mapalign.api —
// allocates nothing (ma_mem_info_get).
struct MemStats {
  uint64_t allocs = 0, frees = 0, bytes_live = 0, bytes_peak = 0;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  MemStats* stats = nullptr;
  // Grows only; 12.5 % headroom so that clouds of slightly different sizes (successive scans of
  // the same vehicle) reuse the allocation.
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) {
      cap = want;
      if (stats) {
        ++stats->allocs;
        stats->bytes_live += want;
        stats->bytes_peak = std::max(stats->bytes_peak, stats->bytes_live);
      }
    } else {
      p = nullptr;
    }
    return e;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
      if (stats) { ++stats->frees; stats->bytes_live -= cap; }
    }
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

thread_local std::string g_create_error;
// the message of the calling THREAD's last ma_ctx::fail: ma_set_pair runs two threads on one context, and which of two
// failures a context reports must not depend on which thread wrote its message last
thread_local std::string t_last_fail;

}  // namespace

struct ma_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t evu0 = nullptr, evu1 = nullptr;   // around the last k_unproject launch (its count — and time — may be settled after other work has re-used ev0 / ev1)
  std::string err;

  // gold map
  bool have_gold = false;
  uint32_t m = 0;
  double origin[3] = {0, 0, 0};
  GridP grid{};                 // level 0 (finest)
  uint64_t ncells = 0, noccupied = 0;
  double build_ms = 0;
  DevBuf gold[kLevels], cstart[kLevels], perm0, occ_a, occ_b;
  LevelSet levels{};
  // levels 1..3 serve only search spheres wider than 6 level-0 cells (ungated or widely gated
  // runs): they are built when such a run first asks for them, not with every gold map
  bool coarse_built = false;
  float grid_lo[3] = {0, 0, 0}, grid_hi[3] = {0, 0, 0};   // box of the grid (after the phase shift)

  // source cloud
  bool have_src = false;
  bool src_sorted = false;
  uint32_t n = 0;                // slots of the source buffer the search kernel covers
  uint32_t n_pts = 0;            // points the caller sees (= n, or the valid pixels of a depth frame: its slots may be holes)
  bool src_holes = false;        // src has holes (x = NaN): per-point results are compacted through src_perm, built on demand
  bool src_map_ready = false;
  DevBuf src, src_perm;          // src_perm: slot -> caller's index (the sort's permutation, or the rank map of a cloud with holes)

  // last uploaded depth frame (ma_set_source_from_depth / ma_lift_boxes)
  int depth_w = 0, depth_h = 0, depth_dtype = -1;

  // map builder (N4)
  DevBuf map_pts, map_keys_a, map_keys_b, map_flags, map_seg, map_acc, map_pre, map_heads;
  uint64_t map_n = 0, map_cap = 0, map_valid = 0;   // slots used / capacity / real points among the slots
  float map_voxel = 0.f;
  double map_origin[3] = {0, 0, 0};
  bool map_has_origin = false;

  // scratch for builds
  DevBuf raw, raw64, wts, keys_a, keys_b, vals_a, vals_b, cub_tmp, misc, depth_in;
  DevBuf raw_src, raw64_src;     // ma_set_pair{,_f64}: the scan is uploaded here while the gold map is being built from `raw`
  DevBuf akeys_a, akeys_b, avals_a, avals_b, acub_tmp, amisc;     // ... and sorted with its own scratch on its own stream
  hipStream_t aux_stream = nullptr;
  DevBuf gap_list;               // k_cell_starts' list of huge runs of empty cells (GapList)
  bool gap_list_dirty = false;
  DevBuf probe_marks;            // mark tables of the occupancy probes: all-zero between calls (occupied_cells_multi)
  size_t probe_clean = 0;        //   bytes of it known to be zero
  char* mailbox = nullptr;       // pinned host memory, 3 x kMailBytes: the small results of the build kernels land here (Mailbox below); the third block: k_unproject's counts

  // batched box lifting (a10 over many frames)
  DevBuf lift_poses, lift_frames;
  DevBuf unproj_cnt;             // valid pixels per workgroup of the last k_unproject launch (frames too large for the pinned block)
  bool unproj_pending = false;   // a k_unproject launch whose count of valid pixels has not been added up yet (finish_unproject)
  bool unproj_is_source = false; //   ... and it wrote the source cloud: the count settles n_pts / src_holes / MA_ERR_EMPTY
  uint32_t unproj_blocks = 0, unproj_total = 0;

  // loop state
  DevBuf state, partials, trace, acc19, ticket, wg_cost, seg_order, nn_idx, nn_d2, nn_tmp, prev;
  // candidate lists of the loop (k_nn_accum): 8 gold positions + one radius per source slot, and the control block
  DevBuf lists, listctl;      // lists: the five planes of ListPlanes, back to back
  float src_rs = 0.f;                          // largest distance of a corner of the source cloud's box from the centring origin; 0: unknown, no lists
  float src_lo[3] = {0, 0, 0}, src_hi[3] = {0, 0, 0};   // that box (centred frame)
  bool src_box_pending = false;                // source from a depth frame: the box is computed by the first run that can use it
  bool lists_on = false;                       // set by the loop (icp_loop) around its launches
  bool cfg_lists = true;                       // ma_list_config: lists allowed at all; skin = cell edge / cfg_skin_div; a write must
  float cfg_skin_div = ma::kSkinDivDefault;    //   be followed by >= cfg_min_uses predicted uses
  int cfg_min_uses = ma::kMinUsesDefault;
  int cfg_planner_timeout_us = 20 * 1000 * 1000;   // how long the planner waits for a slot before it falls back (MAPALIGN_PLANNER_TIMEOUT_US: tests)
  int32_t plan_diag[4] = {0, 0, 0, 0};         // last loop: planner waits, waits that left the spin for yields, timeouts, iterations enqueued
  ListCtl* host_lc = nullptr;                  // pinned: the control block a run starts from
  StateSlot* host_slots = nullptr;             // pinned, kMaxSlots slots: the state every solve of a list-planning loop produced
  int list_builds = 0, list_uses = 0;          // launches of the last ma_icp that wrote / read the lists
  std::vector<uint8_t> list_plan;              // per iteration of the last ma_icp: 0 searched, 1 searched and wrote the lists, 2 read them
  // The stepping state (DevState, prev[], partials, ...) is sized for the clouds that were set
  // when it was uploaded; installing another gold map or source cloud invalidates it
  // (ma_step_accumulate / ma_step_solve then return MA_ERR_STATE until ma_state_set).
  bool state_valid = false;
  int* host_done = nullptr;      // pinned host word the solve sets when the loop stops (tol > 0 runs); first word of a 1 KB pinned block:
  DevState* host_state_in = nullptr;    //   + 256: the initial state of a run, uploaded from here (a pageable source blocks the caller while it is staged)
  DevState* host_state_out = nullptr;   //   + 512: the state as the last solve left it (k_reduce_solve mirrors it here)
  std::vector<hipEvent_t> batch_ev;
  MemStats mem;
  int n_cus = 256;               // compute units of the device (how many workgroups are resident at once)
  double last_op_ms = 0;         // device time of the last unprojection (ma_last_device_ms)
  bool oom = false;              // a C++ allocation failed inside the last call (no message storage)
  // RCCL communicator of the sharded-source mode (ma_comm_init); opaque here
  void* comm = nullptr;
  int comm_ranks = 0, comm_rank = 0;
  int trace_cap = 0;
  int last_G = 0;
  uint32_t last_gshift = 0;
  std::vector<hipEvent_t> kev;   // MA_FLAG_TIME_KERNELS
  double ktimes[4] = {0, 0, 0, 0};
  std::vector<float> iter_ms;    // MA_FLAG_TIME_KERNELS: per iteration {search launch, reduce + solve launch}

  std::mutex err_mu;             // ma_set_pair stages the scan on a helper thread: both threads may report
  int fail(int code, const std::string& msg) {
    t_last_fail = msg;
    std::lock_guard<std::mutex> lock(err_mu);
    err = msg;
    return code;
  }
  // ma_set_pair: the gold build tells the helper thread the cell edge as soon as it is chosen, so that the
  // scan's own sort (which uses it) runs beside the rest of the gold build; state 0 waiting, 1 go, 2 abort
  struct PairHandoff {
    std::mutex m;
    std::condition_variable cv;
    int state = 0;
    float cell = 0.f;
    void set(int st, float c) {
      { std::lock_guard<std::mutex> lock(m); if (state != 0) return; state = st; cell = c; }
      cv.notify_all();
    }
  };
  PairHandoff* pair_handoff = nullptr;
  std::vector<DevBuf*> all_bufs() {
    std::vector<DevBuf*> v = {&map_pts, &map_keys_a, &map_keys_b, &map_flags, &map_seg, &map_acc, &map_pre, &map_heads, &perm0, &occ_a, &occ_b, &probe_marks, &gap_list,
                              &src, &src_perm, &raw, &raw_src, &raw64_src, &akeys_a, &akeys_b, &avals_a, &avals_b, &acub_tmp, &amisc, &raw64, &wts, &keys_a, &keys_b, &vals_a, &vals_b, &cub_tmp, &misc,
                              &depth_in, &lift_poses, &lift_frames, &unproj_cnt, &state, &partials, &trace, &acc19, &ticket,
                              &wg_cost, &seg_order, &nn_idx, &nn_d2, &nn_tmp, &prev, &lists, &listctl};
    for (int l = 0; l < kLevels; ++l) { v.push_back(&gold[l]); v.push_back(&cstart[l]); }
    return v;
  }
};

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ctx->fail(MA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

namespace {

// ---------------------------------------------------------------------------------------
// RCCL (sharded-source mode): loaded on first use so that single-GPU users
// need no librccl at load time, and so that a process that already holds an RCCL (PyTorch
// bundles one under the same soname) shares it instead of getting a second copy.
// ---------------------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

RcclApi g_rccl;

RcclApi* rccl_api() {
  static const bool ok = [] {   // C++11 static initialisation is thread-safe
    RcclApi& api = g_rccl;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      const char* e = dlerror();
      api.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
      return false;
    }
    auto sym = [&](const char* name) -> void* {
      void* f = dlsym(api.handle, name);
      if (!f && api.error.empty()) api.error = std::string("librccl lacks ") + name;
      return f;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return api.error.empty();
  }();
  return ok ? &g_rccl : nullptr;
}

#define NCCLCHK(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return ctx->fail(MA_ERR_COMM, std::string(#expr) + ": " + rccl_api()->GetErrorString(r_));   \
  } while (0)

void comm_destroy(ma_ctx* ctx) {
  RcclApi* api = rccl_api();
  if (api && ctx->comm) (void)api->CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_ranks = 0;
  ctx->comm_rank = 0;
}

int check_params(ma_ctx* ctx, const ma_params* p) {
  if (!p) return ctx->fail(MA_ERR_INVALID, "params is NULL");
  if (p->max_iter < 1) return ctx->fail(MA_ERR_INVALID, "max_iter must be >= 1");
  if (!(p->tol >= 0.0) || !std::isfinite(p->tol)) return ctx->fail(MA_ERR_INVALID, "tol must be finite and >= 0");
  if (!(p->max_corr_dist > 0.0)) return ctx->fail(MA_ERR_INVALID, "max_corr_dist must be > 0 (or +inf)");
  return MA_OK;
}

int need_ready(ma_ctx* ctx) {
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "gold map not set (ma_set_gold)");
  if (!ctx->have_src) return ctx->fail(MA_ERR_STATE, "source cloud not set (ma_set_source)");
  return MA_OK;
}

// A lane = one stream plus the scratch buffers the staging steps of a cloud use on it.  Everything runs on
// the main lane (the context's stream) except the scan of ma_set_pair, which is staged on the aux lane by a
// helper thread while the main lane builds the gold map.
#if MA_PAIR_TIMES   // debug variant only (tools/pair_host_times.py): host-side time stamps of one ma_set_pair, both threads
std::mutex g_pt_mu;
std::vector<std::pair<const char*, double>> g_pt;
void pt(const char* what) {
  const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  std::lock_guard<std::mutex> lock(g_pt_mu);
  g_pt.emplace_back(what, t);
}
void pt_dump() {
  std::lock_guard<std::mutex> lock(g_pt_mu);
  std::sort(g_pt.begin(), g_pt.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
  for (const auto& e : g_pt) std::fprintf(stderr, "  %9.1f us  %s\n", e.second - g_pt.front().second, e.first);
  g_pt.clear();
}
#define MA_PT(x) pt(x)
#else
#define MA_PT(x) ((void)0)
#endif
//
// Mailbox: the few numbers a staging step hands back to the host (bounds, occupancy counts, the count of bad
// weights) are WRITTEN BY THE KERNELS INTO PINNED HOST MEMORY and read after the stream synchronisation the step
// has anyway.  Round 2 copied them device-to-host into pageable vectors; next to the other thread's 12 MB pageable
// upload such a copy waited for that upload to finish (rocprofv3 memory-copy trace: the gold map's 32 KB of bounds
// arrived 230 us after the kernel that produced them, right behind the scan's upload).
constexpr size_t kMailBytes = 128 * 1024;      // per lane: [0, 64K) bounds, [64K, 96K) probe counts, [96K, ...) single words
constexpr size_t kMailProbe = 64 * 1024, kMailWord = 96 * 1024;
static_assert(kMaxProbes * kProbeBlocks * 8 <= kMailWord - kMailProbe, "probe region of the mailbox");
struct Lane {
  hipStream_t stream;
  DevBuf *keys_a, *keys_b, *vals_a, *vals_b, *cub_tmp, *misc;
  char* mail;
};
Lane main_lane(ma_ctx* ctx) { return {ctx->stream, &ctx->keys_a, &ctx->keys_b, &ctx->vals_a, &ctx->vals_b, &ctx->cub_tmp, &ctx->misc, ctx->mailbox}; }
Lane aux_lane(ma_ctx* ctx) { return {ctx->aux_stream, &ctx->akeys_a, &ctx->akeys_b, &ctx->avals_a, &ctx->avals_b, &ctx->acub_tmp, &ctx->amisc, ctx->mailbox + kMailBytes}; }

// bounding box (and finiteness check) of a device-resident n x 3 cloud
int cloud_bounds(ma_ctx* ctx, const float* d_xyz, uint32_t n, float lo[3], float hi[3], int stride = 3, bool holes_ok = false,
                 const Lane* lane = nullptr) {
  const Lane ml = main_lane(ctx);
  const Lane& L = lane ? *lane : ml;
  const int nb = std::min<uint32_t>(div_up(n, kBlock), 1024);
  float* h = reinterpret_cast<float*>(L.mail);          // [nb][8], written by the kernel
  if (stride == 4) hipLaunchKernelGGL(k_minmax<4>, dim3(nb), dim3(kBlock), 0, L.stream, d_xyz, n, h, holes_ok ? 1 : 0);
  else hipLaunchKernelGGL(k_minmax<3>, dim3(nb), dim3(kBlock), 0, L.stream, d_xyz, n, h, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(L.stream));
  double bad = 0;
  for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  for (int b = 0; b < nb; ++b) {
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], h[(size_t)b * 8 + a]);
      hi[a] = std::max(hi[a], h[(size_t)b * 8 + 3 + a]);
    }
    bad += h[(size_t)b * 8 + 6];
  }
  if (bad > 0) return ctx->fail(MA_ERR_INVALID, "cloud has non-finite coordinates (filter them first, as the reference's read_point_cloud(remove_nan_points=True, remove_infinite_points=True) does)");
  return MA_OK;
}

// Stable radix sort of (cell key, index) pairs: the three-kernel passes above for clouds of up to kSortMaxN points,
// rocPRIM's Onesweep (one look-back-chained launch per 8 key bits; the library's choice above 2^20 items anyway)
// beyond.  Two-call pattern of the library: tmp == nullptr asks for the temporary bytes.  The inputs are left intact.
constexpr uint32_t kSortMaxN = 4u << 20;
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                              (size_t)65536>;
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
hipError_t sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream) {
  if (n > kSortMaxN)
    return rocprim::radix_sort_pairs<SortConfig>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                                 (unsigned)bits, stream);
  const uint32_t nblocks = div_up(n, (uint32_t)kSortTile);
  const size_t pair_bytes = align256((size_t)n * 4), hist_bytes = align256((size_t)kSortBins * nblocks * 4);
  const size_t need = 2 * pair_bytes + hist_bytes + align256(kSortBins * 4);
  if (!tmp) { tmp_bytes = need; return hipSuccess; }
  if (tmp_bytes < need) return hipErrorInvalidValue;
  char* base = static_cast<char*>(tmp);
  uint32_t* t_keys = reinterpret_cast<uint32_t*>(base);
  uint32_t* t_vals = reinterpret_cast<uint32_t*>(base + pair_bytes);
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + 2 * pair_bytes);
  uint32_t* totals = reinterpret_cast<uint32_t*>(base + 2 * pair_bytes + hist_bytes);
  const int passes = std::max(1, (bits + 7) / 8);
  const uint32_t* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;       // the last pass writes the caller's output; alternate backwards from there
    uint32_t* dst_k = to_out ? keys_out : t_keys;
    uint32_t* dst_v = to_out ? vals_out : t_vals;
    hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(kBlock), 0, stream, src_k, n, 8 * p, nblocks, hist);
    hipLaunchKernelGGL(k_sort_rows, dim3(kSortBins / (kBlock / kWave)), dim3(kBlock), 0, stream, hist, nblocks, totals);
    hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(kBlock), 0, stream, src_k, src_v, n, 8 * p, nblocks,
                       (const uint32_t*)hist, (const uint32_t*)totals, dst_k, dst_v);
    src_k = dst_k;
    src_v = dst_v;
  }
  return hipGetLastError();
}

// keys/vals for `n` raw points + optional histogram; then a stable radix sort by cell id.
// On return vals_b holds the permutation (sorted position -> original index).
int sort_by_cell(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP& g, const double o[3],
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
int occupied_cells_multi(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP* grids, int count, const double o[3],
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
int occupied_cells(ma_ctx* ctx, const float* d_xyz, uint32_t n, const GridP& g, const double o[3], uint64_t* out) {
  return occupied_cells_multi(ctx, d_xyz, n, &g, 1, o, out);
}

// Candidate lists are a per-context setting (ma_list_config; MA_FLAG_NO_LISTS switches them off for one run).  The
// environment only supplies the defaults a context is created with: MAPALIGN_LISTS=0 every launch searches,
// MAPALIGN_SKIN_DIV=<d> skin = cell edge / d, MAPALIGN_MIN_USES=<n> (sweeps).  Nothing is read from the environment
// after ma_create.
using ma::kSkinDivDefault;
using ma::kMinUsesDefault;
void list_config_from_env(ma_ctx* ctx) {
  const char* e = std::getenv("MAPALIGN_LISTS");
  ctx->cfg_lists = !(e && e[0] == '0');
  e = std::getenv("MAPALIGN_SKIN_DIV");
  const float v = e ? (float)std::atof(e) : 0.f;
  ctx->cfg_skin_div = v >= 2.f && v <= 64.f ? v : kSkinDivDefault;
  e = std::getenv("MAPALIGN_MIN_USES");
  const int u = e ? std::atoi(e) : 0;
  ctx->cfg_min_uses = u >= 1 && u <= 64 ? u : kMinUsesDefault;
  e = std::getenv("MAPALIGN_PLANNER_TIMEOUT_US");       // fault injection for the tests: make the planner's wait give up early
  const long t = e ? std::atol(e) : 0;
  if (t >= 1 && t <= 60L * 1000 * 1000) ctx->cfg_planner_timeout_us = (int)t;
}
bool lists_enabled(const ma_ctx* ctx) { return ctx->cfg_lists; }
float list_skin_div(const ma_ctx* ctx) { return ctx->cfg_skin_div; }

int ensure_loop_buffers(ma_ctx* ctx, int trace_rows) {
  HIPCHK(ctx->state.ensure(sizeof(DevState)));
  HIPCHK(ctx->partials.ensure((size_t)kNAcc * grid_for(std::max<uint32_t>(ctx->n, 1)) * sizeof(double)));
  HIPCHK(ctx->acc19.ensure(kNAcc * sizeof(double)));
  HIPCHK(ctx->wg_cost.ensure((size_t)grid_for(std::max<uint32_t>(ctx->n, 1)) * sizeof(uint32_t)));
  HIPCHK(ctx->listctl.ensure(sizeof(ListCtl)));
  if (ctx->have_src && lists_enabled(ctx)) {      // candidate lists: sized with the other loop buffers, so that no later run allocates
    const size_t slots = (size_t)grid_for(std::max<uint32_t>(ctx->n, 1)) * kNNBlock;
    HIPCHK(ctx->lists.ensure(slots * kListBytesPerSlot));
  }
  if (!ctx->seg_order.p) {  // never read before the first ranking wrote it (order is used from iteration 1 on); zeroed anyway
    HIPCHK(ctx->seg_order.ensure((size_t)kMaxGroups * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(ctx->seg_order.p, 0, (size_t)kMaxGroups * sizeof(uint32_t), ctx->stream));
  }
  if (!ctx->ticket.p) {   // arrival counter of k_reduce_solve: every launch leaves it at 0 again
    HIPCHK(ctx->ticket.ensure(sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(ctx->ticket.p, 0, sizeof(uint32_t), ctx->stream));
  }
  if (trace_rows > ctx->trace_cap) {
    HIPCHK(ctx->trace.ensure((size_t)trace_rows * MA_TRACE_STRIDE * sizeof(double)));
    ctx->trace_cap = trace_rows;
  }
  return MA_OK;
}

// The state and the list control block a run starts from, handed over BY VALUE in the kernel's arguments: one tiny
// launch instead of two small host-to-device copies in front of every loop (each copy is a DMA or blit of its own on
// the stream, ~5 us of host time and as much on the device's).
__global__ __launch_bounds__(kWave) void k_begin_run(const DevState s, const ListCtl c, DevState* __restrict__ st,
                                                     ListCtl* __restrict__ lc) {
  if (threadIdx.x == 0) { *st = s; *lc = c; }
}

int upload_state(ma_ctx* ctx, const double* T) {
  int rc = ensure_loop_buffers(ctx, std::max(ctx->trace_cap, 64));
  if (rc) return rc;
  DevState s{};
  world_to_centred(T, ctx->origin, s);
  s.prev_rmse = NAN; s.last_rmse = NAN; s.last_W = 0;
  s.iter = 0; s.done = 0; s.converged = 0; s.degenerate = 0;
  // from the pinned slot (every earlier use of it has completed: each entry point that runs the loop ends with a sync)
  *ctx->host_state_in = s;
  *ctx->host_state_out = s;
  // a new run starts without remembered matches: the search ignores prev[] while DevState::iter == 0 and has
  // written every entry by the end of that iteration — no 4 MB fill per run
  if (ctx->have_src) HIPCHK(ctx->prev.ensure((size_t)ctx->n * 4));
  {
    // candidate lists: none yet; the skin is a seventh of the cell edge (the density rule puts ~4.5 points into an
    // occupied cell, so a list then holds 2-3 points on average and overflows 8 for < 0.2 % of the queries)
    ListCtl c{};
    c.skin = ctx->grid.c / list_skin_div(ctx);
    c.rs = (ctx->have_src && lists_enabled(ctx)) ? ctx->src_rs : 0.f;
    for (int a = 0; a < 3; ++a) { c.blo[a] = ctx->src_lo[a]; c.bhi[a] = ctx->src_hi[a]; }
    c.slack = 1e-5f + 4e-7f * ctx->src_rs;
    *ctx->host_lc = c;
    hipLaunchKernelGGL(k_begin_run, dim3(1), dim3(kWave), 0, ctx->stream, s, c, ctx->state.as<DevState>(), ctx->listctl.as<ListCtl>());
    HIPCHK(hipGetLastError());
  }
  *ctx->host_done = 0;   // every earlier launch has completed: each entry point that runs the loop ends with a sync
  ctx->state_valid = ctx->have_src;
  return MA_OK;
}



// `scheduled`: inside the ma_icp loop (costs recorded, longest-first order from iteration 1 on)
template <int MODE>
int launch_nn(ma_ctx* ctx, bool count, float best2_init, uint32_t* nn_idx, float* nn_d2, bool scheduled = false,
              int list_kind = 0, int warm_hint = -1) {
  uint32_t gshift = 0;
  const uint32_t G = grid_for(ctx->n, &gshift);
  ctx->last_G = (int)G;
  ctx->last_gshift = gshift;
  Sched sched{nullptr, nullptr, gshift};
  if (scheduled) { sched.wg_cost = ctx->wg_cost.as<uint32_t>(); sched.order = ctx->seg_order.as<uint32_t>(); }
  const bool with_lists = MODE == 0 && ctx->lists_on && list_kind != 0;
  ListPlanes lp{};
  if (with_lists) {      // one list and one radius per slot of the launched grid
    const size_t slots = (size_t)G * kNNBlock;
    HIPCHK(ctx->lists.ensure(slots * kListBytesPerSlot));
    char* base = ctx->lists.as<char>();
    lp.a0 = reinterpret_cast<float4*>(base);
    lp.a1 = reinterpret_cast<float4*>(base + slots * 16);
    lp.b0 = reinterpret_cast<uint32_t*>(base + slots * 32);
    lp.b1 = reinterpret_cast<uint32_t*>(base + slots * 48);
    lp.rad = reinterpret_cast<float*>(base + slots * 64);
  }
  auto args = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(G), dim3(kNNBlock), 0, ctx->stream, ctx->state.as<DevState>(), ctx->src.as<float4>(), ctx->n,
                       warm_hint, MODE == 0 ? ctx->prev.as<uint32_t>() : (uint32_t*)nullptr, sched,
                       ctx->partials.as<double>(), best2_init, ctx->levels, ctx->perm0.as<uint32_t>(), nn_idx, nn_d2,
                       with_lists ? ctx->listctl.as<ListCtl>() : (const ListCtl*)nullptr, lp);
  };
  if (MODE == 0 && with_lists && list_kind == 1) {
    if (count) args(k_nn_accum<true, MODE == 0 ? 0 : 1, MODE == 0 ? 1 : 0>);
    else args(k_nn_accum<false, MODE == 0 ? 0 : 1, MODE == 0 ? 1 : 0>);
  } else if (MODE == 0 && with_lists && list_kind == 2) {
    if (count) args(k_nn_accum<true, MODE == 0 ? 0 : 1, MODE == 0 ? 2 : 0>);
    else args(k_nn_accum<false, MODE == 0 ? 0 : 1, MODE == 0 ? 2 : 0>);
  } else {
    if (count) args(k_nn_accum<true, MODE, 0>);
    else args(k_nn_accum<false, MODE, 0>);
  }
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// partials -> ctx->acc19 (19 workgroups), then optionally the solve (one wavefront)
// with_rank: one more workgroup orders the scheduling groups for the next search launch
int launch_reduce(ma_ctx* ctx, bool with_rank = false) {
  hipLaunchKernelGGL(k_reduce19, dim3(kNAcc + (with_rank ? 1 : 0)), dim3(kBlock), 0, ctx->stream,
                     ctx->partials.as<double>(), ctx->last_G, ctx->state.as<DevState>(), ctx->acc19.as<double>(),
                     ctx->wg_cost.as<uint32_t>(), ctx->seg_order.as<uint32_t>(), ctx->last_gshift);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// the same sum plus the solve by the last-arriving workgroup, one launch
int launch_reduce_solve(ma_ctx* ctx, double tol, StateSlot* slot = nullptr) {
  hipLaunchKernelGGL(k_reduce_solve, dim3(kNAcc + 1), dim3(kBlock), 0, ctx->stream, ctx->partials.as<double>(),
                     ctx->last_G, ctx->state.as<DevState>(), ctx->acc19.as<double>(), ctx->ticket.as<uint32_t>(),
                     ctx->trace.as<double>(), ctx->trace_cap, tol, ctx->wg_cost.as<uint32_t>(),
                     ctx->seg_order.as<uint32_t>(), ctx->last_gshift, ctx->host_done,
                     slot ? &slot->s : ctx->host_state_out, slot);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// in_loop: inside the sharded loop (pinned `done` word, pinned mirror of the state, planner slot)
int launch_solve(ma_ctx* ctx, double tol, bool in_loop = false, StateSlot* slot = nullptr) {
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(kWave), 0, ctx->stream, ctx->acc19.as<double>(),
                     ctx->state.as<DevState>(), ctx->trace.as<double>(), ctx->trace_cap, tol,
                     in_loop ? ctx->host_done : (int*)nullptr,
                     in_loop ? (slot ? &slot->s : ctx->host_state_out) : (DevState*)nullptr, slot);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

int fetch_state(ma_ctx* ctx, DevState* s) {
  HIPCHK(hipMemcpyAsync(s, ctx->state.p, sizeof(DevState), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
}

void fill_stats(const DevState& s, double gpu_ms, ma_stats* st) {
  if (!st) return;
  st->iterations = s.iter;
  st->converged = s.converged;
  st->degenerate = s.degenerate;
  st->reserved = 0;
  st->rmse = s.last_rmse;
  st->inliers = s.last_W;
  st->gpu_ms = gpu_ms;
}

// Depth frame already in ctx->depth_in -> one slot per pixel at `out` (k_unproject: points and holes).
// evu0/evu1 bracket the device work.  The kernel writes its per-workgroup counts of valid pixels straight into pinned
// host memory (the third mailbox block), so the NUMBER of valid pixels costs no device-to-host copy — and no
// synchronisation at all unless somebody asks for it: total != nullptr waits and adds the counts up; with nullptr the
// call only enqueues, and finish_unproject() settles the count when it is first needed (round 3: a pageable copy
// of the count array and a stream sync in every call, 0.21 ms of host wall for 14 us of device work).
int finish_unproject(ma_ctx* ctx) {
  if (!ctx->unproj_pending) return MA_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->unproj_pending = false;
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->evu0, ctx->evu1));
  ctx->last_op_ms = ms;
  const uint32_t* h = reinterpret_cast<const uint32_t*>(ctx->mailbox + 2 * kMailBytes);
  uint64_t t = 0;
  for (uint32_t b = 0; b < ctx->unproj_blocks; ++b) t += h[b];
  ctx->unproj_total = (uint32_t)t;
  if (ctx->unproj_is_source) {          // the frame is the source cloud: what the count decides
    ctx->n_pts = ctx->unproj_total;
    ctx->src_holes = ctx->unproj_total != ctx->n;
    if (ctx->unproj_total == 0) {
      ctx->have_src = false;
      return ctx->fail(MA_ERR_EMPTY, "depth image has no valid pixel");
    }
  }
  return MA_OK;
}

int unproject_frame(ma_ctx* ctx, int dtype, int width, uint32_t npx, const Pinhole& cam, float4* out, uint32_t* total,
                    bool is_source) {
  const uint32_t nb = div_up(npx, kPxPerBlock);
  uint32_t* counts = reinterpret_cast<uint32_t*>(ctx->mailbox + 2 * kMailBytes);       // pinned: the kernel stores there
  const bool pinned = (size_t)nb * 4 <= kMailBytes;
  if (!pinned) {                     // a frame above 16.7 MP: counts in device memory, copied back
    HIPCHK(ctx->unproj_cnt.ensure((size_t)nb * 4));
    counts = ctx->unproj_cnt.as<uint32_t>();
  }
  HIPCHK(hipEventRecord(ctx->evu0, ctx->stream));
  if (dtype == 0)
    hipLaunchKernelGGL(k_unproject<float>, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<float>(), width, npx, cam,
                       counts, out);
  else
    hipLaunchKernelGGL(k_unproject<uint16_t>, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<uint16_t>(), width, npx,
                       cam, counts, out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evu1, ctx->stream));
  if (!pinned) {
    std::vector<uint32_t> h(nb);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->unproj_cnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->evu0, ctx->evu1));
    ctx->last_op_ms = ms;
    uint64_t t = 0;
    for (uint32_t c : h) t += c;
    ctx->unproj_total = (uint32_t)t;
    ctx->unproj_pending = false;
    if (total) *total = ctx->unproj_total;
    return MA_OK;
  }
  ctx->unproj_pending = true;
  ctx->unproj_blocks = nb;
  ctx->unproj_is_source = is_source;
  if (total) {
    const bool was_source = ctx->unproj_is_source;
    ctx->unproj_is_source = false;          // the caller looks at the total itself
    int rc = finish_unproject(ctx);
    ctx->unproj_is_source = was_source;
    if (rc) return rc;
    *total = ctx->unproj_total;
  }
  return MA_OK;
}

// slot -> caller's index for everything that hands per-point data back: the sort's permutation, the rank
// map of a cloud with holes (built here, once per cloud, when first needed), or null (identity).
int source_map(ma_ctx* ctx, const uint32_t** map) {
  *map = nullptr;
  { const int rc0 = finish_unproject(ctx); if (rc0) return rc0; }     // whether the cloud has holes is the count's to say
  if (ctx->src_sorted) { *map = ctx->src_perm.as<uint32_t>(); return MA_OK; }
  if (!ctx->src_holes) return MA_OK;
  if (!ctx->src_map_ready) {
    const uint32_t n = ctx->n;
    HIPCHK(ctx->keys_a.ensure((size_t)n * 4));
    HIPCHK(ctx->keys_b.ensure((size_t)n * 4));
    HIPCHK(ctx->src_perm.ensure((size_t)n * 4));
    hipLaunchKernelGGL(k_hole_flags, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->src.as<float4>(), n,
                       ctx->keys_a.as<uint32_t>());
    HIPCHK(hipGetLastError());
    size_t tmp_bytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(), (int)n, ctx->stream));
    HIPCHK(ctx->cub_tmp.ensure(tmp_bytes));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cub_tmp.p, tmp_bytes, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(), (int)n, ctx->stream));
    hipLaunchKernelGGL(k_rank_map, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->keys_a.as<uint32_t>(),
                       ctx->keys_b.as<uint32_t>(), n, ctx->src_perm.as<uint32_t>());
    HIPCHK(hipGetLastError());
    ctx->src_map_ready = true;
  }
  *map = ctx->src_perm.as<uint32_t>();
  return MA_OK;
}

Pinhole make_pinhole(const float K[4], const double T[16], float depth_scale, const double origin[3]) {
  Pinhole cam{};
  cam.fx_inv = 1.0 / (double)K[0];
  cam.fy_inv = 1.0 / (double)K[1];
  cam.cx = K[2];
  cam.cy = K[3];
  cam.inv_scale = 1.0 / (double)depth_scale;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) cam.T[4 * r + c] = T[4 * r + c];
    cam.T[4 * r + 3] = T[4 * r + 3] - origin[r];
  }
  return cam;
}

// Levels 1..3 (cell edges 2c, 4c, 8c over the same box) from the level-0 copy.  Stable sort by
// cell id of level l: inside a cell the points keep their level-0 order.
// Round 4: the same launches as level 0's build — cell keys without a histogram (no atomics), the own radix sort, the cell
// table straight from the sorted keys (k_cell_starts + k_fill_gaps) — instead of a histogram with one atomic per point, a
// memset and a library scan per level; and no synchronisation between the levels: everything is on one stream, the
// scratch a level reuses is free by the time its launches run.  (An ungated run's first iteration, which is where these
// levels are built, used to spend most of its 3.7 ms here.)
int build_coarse_levels(ma_ctx* ctx) {
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

// No C++ exception crosses the C ABI: std::vector / std::string allocations inside the entry
// points can throw std::bad_alloc, which is turned into a status here.
template <typename F>
int guarded(ma_ctx* ctx, F&& body) noexcept {
  try {
    if (ctx) ctx->oom = false;
    return body();
  } catch (const std::bad_alloc&) {
    if (ctx) ctx->oom = true;
    return MA_ERR_NO_MEMORY;
  } catch (...) {
    return MA_ERR_INVALID;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
// Builds the gold map (grids of all levels, distance table) from the n x 3 fp32 world-frame
// points already resident in ctx->raw; ev0 must have been recorded by the caller.
static int build_gold_from_raw(ma_ctx* ctx, uint32_t n, float cell);
static int build_source_from_raw(ma_ctx* ctx, uint32_t n, const float* weights, const float* d_raw = nullptr,
                                 float gold_cell = 0.f, const Lane* lane = nullptr);
static int stage_f64(ma_ctx* ctx, const double* xyz, uint32_t n, bool choose_shift, double shift[3]);

extern "C" {

int ma_version(void) { return MA_VERSION; }

const char* ma_status_string(int status) {
  switch (status) {
    case MA_OK: return "ok";
    case MA_ERR_INVALID: return "invalid argument";
    case MA_ERR_NO_DEVICE: return "no usable HIP device";
    case MA_ERR_HIP: return "HIP runtime error";
    case MA_ERR_EMPTY: return "empty cloud";
    case MA_ERR_STATE: return "call-order error";
    case MA_ERR_DEGENERATE: return "degenerate correspondences";
    case MA_ERR_TOO_LARGE: return "cloud too large";
    case MA_ERR_COMM: return "RCCL error";
    case MA_ERR_NO_MEMORY: return "out of host memory";
    default: return "unknown status";
  }
}

const char* ma_last_error(const ma_ctx* ctx) {
  if (ctx && ctx->oom) return "out of host memory inside the library";
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int ma_create(int device, ma_ctx** out) {
  return guarded(nullptr, [&]() -> int {
  if (!out) { g_create_error = "out is NULL"; return MA_ERR_INVALID; }
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) +
                     "); libmapalign has no CPU fallback";
    return MA_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    g_create_error = "device ordinal out of range";
    return MA_ERR_NO_DEVICE;
  }
  ma_ctx* ctx = new (std::nothrow) ma_ctx();
  if (!ctx) { g_create_error = "out of host memory"; return MA_ERR_INVALID; }
  ctx->device = device;
  auto bail = [&](const char* what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evu0) (void)hipEventDestroy(ctx->evu0);
    if (ctx->evu1) (void)hipEventDestroy(ctx->evu1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->host_done) (void)hipHostFree(ctx->host_done);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    delete ctx;
    return MA_ERR_HIP;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  ctx->stream = ctx->own_stream;
  if (hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cus <= 0) ctx->n_cus = 256;
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->evu0)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->evu1)) != hipSuccess) return bail("hipEventCreate", e);
  static_assert(sizeof(DevState) <= 256, "pinned state slots are 256 bytes apart");
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->host_done), 1280 + kMaxSlots * sizeof(StateSlot), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
  std::memset(ctx->host_done, 0, 1024);
  ctx->host_state_in = reinterpret_cast<DevState*>(reinterpret_cast<char*>(ctx->host_done) + 256);
  ctx->host_state_out = reinterpret_cast<DevState*>(reinterpret_cast<char*>(ctx->host_done) + 512);
  static_assert(sizeof(ListCtl) <= 256 && sizeof(DevState) <= 256, "pinned block slots");
  ctx->host_lc = reinterpret_cast<ListCtl*>(reinterpret_cast<char*>(ctx->host_done) + 1024);
  ctx->host_slots = reinterpret_cast<StateSlot*>(reinterpret_cast<char*>(ctx->host_done) + 1280);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->mailbox), 3 * kMailBytes, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
  std::memset(ctx->mailbox, 0, 3 * kMailBytes);
  for (DevBuf* b : ctx->all_bufs()) b->stats = &ctx->mem;
  list_config_from_env(ctx);
  *out = ctx;
  return MA_OK;
  });
}

int ma_destroy(ma_ctx* ctx) {
  if (!ctx) return MA_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) comm_destroy(ctx);
  try {
    for (DevBuf* b : ctx->all_bufs()) b->release();
  } catch (...) {   // (the buffer list is a std::vector: never let its allocation failure cross the ABI)
  }
  if (ctx->host_done) (void)hipHostFree(ctx->host_done);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  for (hipEvent_t e : ctx->batch_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->kev) (void)hipEventDestroy(e);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->evu0) (void)hipEventDestroy(ctx->evu0);
  if (ctx->evu1) (void)hipEventDestroy(ctx->evu1);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return MA_OK;
}

int ma_set_stream(ma_ctx* ctx, void* hip_stream, int external) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->stream = external ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return MA_OK;
  });
}

int ma_sync(ma_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
  });
}

int ma_set_gold(ma_ctx* ctx, const float* xyz, uint64_t m, float cell) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!xyz && m) return ctx->fail(MA_ERR_INVALID, "xyz is NULL");
  if (m == 0) return ctx->fail(MA_ERR_EMPTY, "gold cloud is empty");
  if (m >= (1ull << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "gold cloud has 2^28 or more points (32-bit byte offsets into the cell-sorted arrays)");
  if (std::isnan(cell) || std::isinf(cell)) return ctx->fail(MA_ERR_INVALID, "cell must be finite");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_gold = false;
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  const uint32_t n = (uint32_t)m;
  HIPCHK(ctx->raw.ensure((size_t)n * 12));
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream));   // host OR device memory (include/mapalign.h)
  return build_gold_from_raw(ctx, n, cell);
  });
}

static int build_gold_from_raw(ma_ctx* ctx, uint32_t n, float cell) {
  float lo[3], hi[3];
  int rc = cloud_bounds(ctx, ctx->raw.as<float>(), n, lo, hi);
  if (rc) return rc;
  MA_PT("gold: bounds read");
  // centring origin: bounding-box centre, rounded to fp32 so that (x - o) is exact-ish
  for (int a = 0; a < 3; ++a) ctx->origin[a] = (double)(float)(0.5 * ((double)lo[a] + (double)hi[a]));

  float c = cell;
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
      c = clamp_cell_to_budget(lo, hi, c2, auto_cells);
    } else {
      c = c1;
    }
  }
  c = clamp_cell_to_budget(lo, hi, c, kMaxCells);
  if (!(c > 0.f) || !std::isfinite(c)) return ctx->fail(MA_ERR_INVALID, "could not choose a cell edge");
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

int ma_grid_info_get(const ma_ctx* ctx, ma_grid_info* out) {
  if (!ctx || !out) return MA_ERR_INVALID;
  if (!ctx->have_gold) return MA_ERR_STATE;
  out->cell = ctx->grid.c;
  out->nx = ctx->grid.nx; out->ny = ctx->grid.ny; out->nz = ctx->grid.nz;
  out->n_cells = ctx->ncells;
  out->n_occupied = ctx->noccupied;
  for (int a = 0; a < 3; ++a) out->origin[a] = ctx->origin[a];
  out->build_ms = ctx->build_ms;
  return MA_OK;
}

int ma_set_source(ma_ctx* ctx, const float* xyz, const float* weights, uint64_t n64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "set the gold map before the source cloud");
  if (!xyz && n64) return ctx->fail(MA_ERR_INVALID, "xyz is NULL");
  if (n64 == 0) return ctx->fail(MA_ERR_EMPTY, "source cloud is empty");
  if (n64 >= (1ull << 31)) return ctx->fail(MA_ERR_TOO_LARGE, "source cloud has 2^31 or more points (32-bit item counts of the sort)");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  const uint32_t n = (uint32_t)n64;
  HIPCHK(ctx->raw.ensure((size_t)n * 12));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream));   // host OR device memory (include/mapalign.h)
  return build_source_from_raw(ctx, n, weights);
  });
}

// Both clouds of a pair in one call: the scan crosses PCIe WHILE the gold map is being built.  A copy from
// pageable host memory occupies the calling thread for its whole length (the runtime stages it chunk by chunk),
// so the scan's upload is issued from a helper thread on a second stream; the gold build's kernels and host
// round trips (ctx->stream) run beside it, and the scan's sort follows once both are done.  Same state afterwards
// as ma_set_gold + ma_set_source (bitwise: the same kernels on the same data), 0.3-0.4 ms sooner for a 1M-point pair.
static int set_pair_impl(ma_ctx* ctx, const void* gold_xyz, uint64_t m, float cell, const void* src_xyz, const float* weights,
                         uint64_t n64, const bool f64, double shift_out[3]) {
  if ((!gold_xyz && m) || (!src_xyz && n64) || (f64 && !shift_out)) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (m == 0) return ctx->fail(MA_ERR_EMPTY, "gold cloud is empty");
  if (n64 == 0) return ctx->fail(MA_ERR_EMPTY, "source cloud is empty");
  if (m >= (1ull << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "gold cloud has 2^28 or more points (32-bit byte offsets into the cell-sorted arrays)");
  if (n64 >= (1ull << 31)) return ctx->fail(MA_ERR_TOO_LARGE, "source cloud has 2^31 or more points (32-bit item counts of the sort)");
  if (std::isnan(cell) || std::isinf(cell)) return ctx->fail(MA_ERR_INVALID, "cell must be finite");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_gold = false;
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  const uint32_t mg = (uint32_t)m, n = (uint32_t)n64;
  const size_t in_bytes = f64 ? 24 : 12;
  if (!ctx->aux_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  // every buffer the helper thread will touch is sized HERE, on the calling thread (allocation counters and
  // hipMalloc stay single-threaded; a warm context allocates nothing)
  HIPCHK(ctx->raw.ensure((size_t)mg * 12));
  HIPCHK(ctx->raw_src.ensure((size_t)n * 12));
  if (f64) {
    HIPCHK(ctx->raw64.ensure((size_t)mg * 24));
    HIPCHK(ctx->raw64_src.ensure((size_t)n * 24));
  }
  HIPCHK(ctx->akeys_a.ensure((size_t)n * 4));
  HIPCHK(ctx->akeys_b.ensure((size_t)n * 4));
  HIPCHK(ctx->avals_a.ensure((size_t)n * 4));
  HIPCHK(ctx->amisc.ensure(1024 * 8 * sizeof(float)));
  HIPCHK(ctx->src.ensure((size_t)n * sizeof(float4)));
  HIPCHK(ctx->src_perm.ensure((size_t)n * 4));
  if (weights) HIPCHK(ctx->wts.ensure((size_t)n * 4));
  {
    size_t tmp_bytes = 0;
    HIPCHK(sort_pairs_u32(nullptr, tmp_bytes, ctx->akeys_a.as<uint32_t>(), ctx->akeys_b.as<uint32_t>(), ctx->avals_a.as<uint32_t>(),
                          ctx->src_perm.as<uint32_t>(), n, 32, ctx->aux_stream));
    HIPCHK(ctx->acub_tmp.ensure(tmp_bytes));
  }
  MA_PT("set_pair: buffers sized");
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  double shift[3] = {0.0, 0.0, 0.0};
  ma_ctx::PairHandoff handoff;
  struct HandoffScope {       // the context points at a stack object: detached again on EVERY way out, a throwing std::thread included
    ma_ctx* ctx;
    explicit HandoffScope(ma_ctx* c, ma_ctx::PairHandoff* h) : ctx(c) { ctx->pair_handoff = h; }
    ~HandoffScope() { ctx->pair_handoff = nullptr; }
  } handoff_scope(ctx, &handoff);
  // The helper is started FIRST (creating a thread and its first HIP call take 60-100 us) and then waits for the
  // gold map's upload call to return before it issues its own: the two uploads share one PCIe link, and the gold
  // map's build is the longer chain, so its 12 MB go first and undisturbed.
  std::atomic<int> gold_uploaded{0};      // 0: wait, 1: go (shift[] is final), 2: give up
  int rc_src = MA_OK;
  std::string msg_src;             // the helper's own failure message (reported once, by the calling thread, after the join)
  std::thread helper([&]() {
    struct KeepMessage { std::string& m; int& rc; ~KeepMessage() { if (rc != MA_OK) m = t_last_fail; } } keep{msg_src, rc_src};
    try {
      MA_PT("scan: helper thread runs");
      hipError_t e = hipSetDevice(ctx->device);
      MA_PT("scan: hipSetDevice returned");
      int go;
      while ((go = gold_uploaded.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (go != 1) return;
      void* dst = f64 ? ctx->raw64_src.p : ctx->raw_src.p;
      if (e == hipSuccess) e = hipMemcpyAsync(dst, src_xyz, (size_t)n * in_bytes, hipMemcpyDefault, ctx->aux_stream);
      MA_PT("scan: upload call returned");
      if (e != hipSuccess) { rc_src = ctx->fail(MA_ERR_HIP, std::string("upload of the source cloud: ") + hipGetErrorString(e)); return; }
      if (f64) {      // convert behind the upload, on the aux stream, with the gold map's shift
        const uint64_t n3 = 3ull * n;
        hipLaunchKernelGGL(k_f64_to_raw, dim3((uint32_t)((n3 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->aux_stream,
                           ctx->raw64_src.as<double>(), n3, shift[0], shift[1], shift[2], ctx->raw_src.as<float>());
        if ((e = hipGetLastError()) != hipSuccess) { rc_src = ctx->fail(MA_ERR_HIP, std::string("k_f64_to_raw: ") + hipGetErrorString(e)); return; }
      }
      float cell_edge = 0.f;
      {
        std::unique_lock<std::mutex> lock(handoff.m);
        handoff.cv.wait(lock, [&] { return handoff.state != 0; });
        if (handoff.state != 1) return;            // the gold build failed before it chose a cell edge
        cell_edge = handoff.cell;
      }
      MA_PT("scan: cell edge received");
      const Lane aux = aux_lane(ctx);
      rc_src = build_source_from_raw(ctx, n, weights, ctx->raw_src.as<float>(), cell_edge, &aux);
    } catch (const std::bad_alloc&) {
      rc_src = MA_ERR_NO_MEMORY;
    } catch (...) {
      rc_src = MA_ERR_INVALID;
    }
  });
  struct Joiner {             // whatever ends the build below — a status or an exception — the helper is released and joined
    std::thread& t;
    std::atomic<int>& uploaded;
    ma_ctx::PairHandoff& h;
    ~Joiner() {
      int expected = 0;
      uploaded.compare_exchange_strong(expected, 2);     // (no effect once the upload has said "go")
      h.set(2, 0.f);          // (no effect once the build has said "go")
      if (t.joinable()) t.join();
    }
  };
  int rc_gold;
  {
    Joiner joiner{helper, gold_uploaded, handoff};
    rc_gold = [&]() -> int {
      if (f64) {          // upload, fp64 bounds and the large-coordinate shift of the gold map, conversion to the fp32 frame
        int rc = stage_f64(ctx, static_cast<const double*>(gold_xyz), mg, true, shift);
        if (rc) return rc;
        for (int a = 0; a < 3; ++a) shift_out[a] = shift[a];
      } else {
        HIPCHK(hipMemcpyAsync(ctx->raw.p, gold_xyz, (size_t)mg * 12, hipMemcpyDefault, ctx->stream));
      }
      MA_PT("gold: upload call returned");
      gold_uploaded.store(1, std::memory_order_release);
      return build_gold_from_raw(ctx, mg, cell);
    }();
  }
  MA_PT("set_pair: helper joined");
#if MA_PAIR_TIMES
  pt_dump();
#endif
  if (rc_gold || rc_src) {
    // ONE report, on the calling thread: the gold map's failure if it had one (its message is this thread's last), else
    // the scan's, with the message the helper thread left behind
    if (rc_gold) (void)ctx->fail(rc_gold, t_last_fail);
    else if (!msg_src.empty()) (void)ctx->fail(rc_src, msg_src);
    ctx->have_src = false;
    ctx->unproj_pending = false;
    if (rc_gold) ctx->have_gold = false;
    return rc_gold ? rc_gold : rc_src;
  }
  return MA_OK;
}

int ma_set_pair(ma_ctx* ctx, const float* gold_xyz, uint64_t m, float cell, const float* src_xyz, const float* weights,
                uint64_t n64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  return set_pair_impl(ctx, gold_xyz, m, cell, src_xyz, weights, n64, false, nullptr);
  });
}

int ma_set_pair_f64(ma_ctx* ctx, const double* gold_xyz, uint64_t m, float cell, double shift_out[3], const double* src_xyz,
                    const float* weights, uint64_t n64) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  return set_pair_impl(ctx, gold_xyz, m, cell, src_xyz, weights, n64, true, shift_out);
  });
}

// d_raw (default ctx->raw) holds the n x 3 fp32 source points (relative to the host-side shift, if any)
static int build_source_from_raw(ma_ctx* ctx, uint32_t n, const float* weights, const float* d_raw, float gold_cell,
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
    HIPCHK(hipMemsetAsync(L.misc->p, 0, 8, L.stream));
    hipLaunchKernelGGL(k_count_bad_weights, dim3(std::min<uint32_t>(div_up(n, kBlock), 1024)), dim3(kBlock), 0,
                       L.stream, d_w, n, L.misc->as<unsigned long long>());
    HIPCHK(hipGetLastError());
    unsigned long long* bad = reinterpret_cast<unsigned long long*>(L.mail + kMailWord + 64);
    HIPCHK(hipMemcpyAsync(bad, L.misc->p, 8, hipMemcpyDeviceToHost, L.stream));     // into pinned memory
    HIPCHK(hipStreamSynchronize(L.stream));
    if (*bad) return ctx->fail(MA_ERR_INVALID, "weights must be finite and >= 0");
  }
  float lo[3], hi[3];
  int rc = cloud_bounds(ctx, d_raw, n, lo, hi, 3, false, &L);
  if (rc) return rc;
  MA_PT("scan: bounds read");
  // sort in the cloud's own frame with the gold map's cell edge (coarsened to fit 2^27
  // cells), brick-major: 256 consecutive points then form a compact patch, so that a
  // workgroup's queries share one small block of the gold grid under any rigid motion.
  const float c = clamp_cell_to_budget(lo, hi, gold_cell, kMaxCells);
  const GridP g = make_grid(lo, hi, ctx->origin, c);
  {
    double r2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double l = (double)lo[a] - ctx->origin[a], h = (double)hi[a] - ctx->origin[a], pad = 1e-4 + 1e-6 * (std::fabs(l) + std::fabs(h));
      ctx->src_lo[a] = (float)(l - pad);
      ctx->src_hi[a] = (float)(h + pad);
      r2 += std::max(l * l, h * h);
    }
    ctx->src_rs = (float)(std::sqrt(r2) * 1.001 + 1e-4);
    ctx->src_box_pending = false;
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
static int stage_f64(ma_ctx* ctx, const double* xyz, uint32_t n, bool choose_shift, double shift[3]) {
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

int ma_set_gold_f64(ma_ctx* ctx, const double* xyz, uint64_t m, float cell, double shift_out[3]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if ((!xyz && m) || !shift_out) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (m == 0) return ctx->fail(MA_ERR_EMPTY, "gold cloud is empty");
  if (m >= (1ull << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "gold cloud has 2^28 or more points (32-bit byte offsets into the cell-sorted arrays)");
  if (std::isnan(cell) || std::isinf(cell)) return ctx->fail(MA_ERR_INVALID, "cell must be finite");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_gold = false;
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  int rc = stage_f64(ctx, xyz, (uint32_t)m, true, shift_out);
  if (rc) return rc;
  return build_gold_from_raw(ctx, (uint32_t)m, cell);
  });
}

int ma_set_source_f64(ma_ctx* ctx, const double* xyz, const float* weights, uint64_t n64, const double shift[3]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "set the gold map before the source cloud");
  if (!xyz && n64) return ctx->fail(MA_ERR_INVALID, "xyz is NULL");
  if (n64 == 0) return ctx->fail(MA_ERR_EMPTY, "source cloud is empty");
  if (n64 >= (1ull << 31)) return ctx->fail(MA_ERR_TOO_LARGE, "source cloud has 2^31 or more points (32-bit item counts of the sort)");
  double s[3] = {0.0, 0.0, 0.0};
  if (shift) for (int a = 0; a < 3; ++a) { if (!std::isfinite(shift[a])) return ctx->fail(MA_ERR_INVALID, "non-finite shift"); s[a] = shift[a]; }
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  int rc = stage_f64(ctx, xyz, (uint32_t)n64, false, s);
  if (rc) return rc;
  return build_source_from_raw(ctx, (uint32_t)n64, weights);   // (non-finite coordinates are caught by its fp32 bounds pass)
  });
}

int ma_set_source_from_depth(ma_ctx* ctx, const void* depth, int dtype, int width, int height,
                             const float K[4], const double T[16], float depth_scale, uint64_t* n_valid) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "set the gold map before the source cloud");
  if (!depth || !K || !T) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (dtype != 0 && dtype != 1) return ctx->fail(MA_ERR_INVALID, "dtype must be 0 (float32) or 1 (uint16)");
  if (width <= 0 || height <= 0 || (uint64_t)width * height >= (1ull << 31)) return ctx->fail(MA_ERR_INVALID, "bad image size");
  if (!(K[0] != 0.f) || !(K[1] != 0.f) || !(depth_scale > 0.f) || !std::isfinite(depth_scale)) return ctx->fail(MA_ERR_INVALID, "bad intrinsics / depth_scale");
  for (int i = 0; i < 16; ++i) if (!std::isfinite(T[i])) return ctx->fail(MA_ERR_INVALID, "non-finite camera pose");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_src = false;
  ctx->unproj_pending = false;
  ctx->state_valid = false;
  const uint32_t npx = (uint32_t)((uint64_t)width * height);
  const size_t esz = dtype == 0 ? 4 : 2;
  HIPCHK(ctx->depth_in.ensure((size_t)npx * esz));
  HIPCHK(hipMemcpyAsync(ctx->depth_in.p, depth, (size_t)npx * esz, hipMemcpyDefault, ctx->stream));
  ctx->depth_w = width; ctx->depth_h = height; ctx->depth_dtype = dtype;
  const Pinhole cam = make_pinhole(K, T, depth_scale, ctx->origin);
  HIPCHK(ctx->src.ensure((size_t)npx * sizeof(float4)));
  ctx->n = npx;                       // one slot per pixel; the invalid ones are holes the search skips
  ctx->n_pts = npx;                   // (until the count is in: finish_unproject)
  ctx->src_sorted = false;
  ctx->src_holes = true;
  ctx->src_map_ready = false;
  // n_valid == NULL: the caller does not want the number now — nothing is waited for, the alignment can be enqueued
  // right behind the unprojection, and an image without a valid pixel shows as MA_ERR_EMPTY when the count is first
  // needed (or as MA_ERR_DEGENERATE in ma_icp, which needs no count: the search skips holes)
  int rc = unproject_frame(ctx, dtype, width, npx, cam, ctx->src.as<float4>(), nullptr, true);
  if (rc) return rc;
  if (n_valid) {
    ctx->have_src = true;             // (finish_unproject takes it back for an empty frame)
    if ((rc = finish_unproject(ctx))) return rc;
    *n_valid = ctx->n_pts;
  }
  ctx->src_rs = 0.f;                  // no bounds of this cloud are known yet: ma_icp finds them when a run is long enough for candidate lists
  ctx->src_box_pending = true;
  ctx->have_src = true;
  return MA_OK;
  });
}

int ma_map_begin(ma_ctx* ctx, float voxel, uint64_t capacity_points) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!(voxel > 0.f) || !std::isfinite(voxel)) return ctx->fail(MA_ERR_INVALID, "voxel must be > 0");
  if (capacity_points == 0 || capacity_points >= (1ull << 31)) return ctx->fail(MA_ERR_INVALID, "capacity must be in [1, 2^31)");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(ctx->map_pts.ensure(capacity_points * sizeof(float4)));
  ctx->map_cap = capacity_points;
  ctx->map_n = 0;
  ctx->map_valid = 0;
  ctx->map_voxel = voxel;
  ctx->map_has_origin = false;
  return MA_OK;
  });
}

int ma_map_add_depth(ma_ctx* ctx, const void* depth, int dtype, int width, int height, const float K[4],
                     const double T[16], float depth_scale, uint64_t* n_added) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (ctx->map_cap == 0) return ctx->fail(MA_ERR_STATE, "ma_map_begin first");
  if (!depth || !K || !T) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (dtype != 0 && dtype != 1) return ctx->fail(MA_ERR_INVALID, "dtype must be 0 (float32) or 1 (uint16)");
  if (width <= 0 || height <= 0 || (uint64_t)width * height >= 0xFFFFFFFEull) return ctx->fail(MA_ERR_INVALID, "bad image size");
  if (!(K[0] != 0.f) || !(K[1] != 0.f) || !(depth_scale > 0.f) || !std::isfinite(depth_scale)) return ctx->fail(MA_ERR_INVALID, "bad intrinsics / depth_scale");
  for (int i = 0; i < 16; ++i) if (!std::isfinite(T[i])) return ctx->fail(MA_ERR_INVALID, "non-finite camera pose");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t npx = (uint32_t)((uint64_t)width * height);
  if (ctx->map_n + npx > ctx->map_cap) return ctx->fail(MA_ERR_TOO_LARGE, "map capacity exceeded (ma_map_begin capacity_points)");
  if (!ctx->map_has_origin) {   // points are stored relative to the first camera position (fp32 precision)
    for (int r = 0; r < 3; ++r) ctx->map_origin[r] = std::round(T[4 * r + 3]);
    ctx->map_has_origin = true;
  }
  const size_t esz = dtype == 0 ? 4 : 2;
  HIPCHK(ctx->depth_in.ensure((size_t)npx * esz));
  HIPCHK(hipMemcpyAsync(ctx->depth_in.p, depth, (size_t)npx * esz, hipMemcpyDefault, ctx->stream));
  ctx->depth_w = width; ctx->depth_h = height; ctx->depth_dtype = dtype;
  const Pinhole cam = make_pinhole(K, T, depth_scale, ctx->map_origin);
  uint32_t total = 0;
  { const int rc0 = finish_unproject(ctx); if (rc0) return rc0; }      // an earlier frame's count still pending in the pinned block
  int rc = unproject_frame(ctx, dtype, width, npx, cam, ctx->map_pts.as<float4>() + ctx->map_n, &total, false);
  if (rc) return rc;
  ctx->map_n += npx;                  // one slot per pixel (holes included: the voxel sort drops them)
  ctx->map_valid += total;
  if (n_added) *n_added = total;
  return MA_OK;
  });
}

int ma_map_finish(ma_ctx* ctx, int as_gold, float cell, float* xyz_out, uint64_t* n_io) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !n_io) return MA_ERR_INVALID;
  if (ctx->map_cap == 0) return ctx->fail(MA_ERR_STATE, "ma_map_begin first");
  if (ctx->map_valid == 0) return ctx->fail(MA_ERR_EMPTY, "the map holds no point");
  if (std::isnan(cell) || std::isinf(cell)) return ctx->fail(MA_ERR_INVALID, "cell must be finite");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t nslots = (uint32_t)ctx->map_n;       // slots: points and holes
  const uint32_t n = (uint32_t)ctx->map_valid;        // real points: the first n entries after the sort
  HIPCHK(ctx->map_keys_a.ensure((size_t)nslots * 8));
  HIPCHK(ctx->map_keys_b.ensure((size_t)nslots * 8));
  HIPCHK(ctx->vals_a.ensure((size_t)nslots * 4));
  HIPCHK(ctx->vals_b.ensure((size_t)nslots * 4));
  HIPCHK(ctx->map_flags.ensure((size_t)(n + 1) * 4));
  HIPCHK(ctx->map_seg.ensure((size_t)(n + 1) * 4));
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  // voxel box of the map (holes skipped) and the key width; a hole's key is one past the box's last voxel
  float lo[3], hi[3];
  int rc = cloud_bounds(ctx, ctx->map_pts.as<float>(), nslots, lo, hi, 4, /*holes_ok=*/true);
  if (rc) return rc;
  const float inv_voxel = 1.0f / ctx->map_voxel;
  VoxelBox box;
  int key_bits = 0;
  if (!ma::voxel_box(lo, hi, inv_voxel, box, key_bits))
    return ctx->fail(MA_ERR_TOO_LARGE, "the voxel indices of the map's bounding box do not fit 63 bits (coordinates / voxel too large)");
  const bool holes = nslots != n;
  const unsigned long long hole_key = key_bits >= 64 ? ~0ull : (1ull << key_bits);     // (voxel_box keeps real keys below 9e18 < 2^63)
  const int sort_bits = holes ? std::min(key_bits + 1, 64) : key_bits;
  hipLaunchKernelGGL(k_voxel_keys, dim3(div_up(nslots, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->map_pts.as<float4>(), nslots,
                     inv_voxel, box, hole_key, ctx->map_keys_a.as<unsigned long long>(), ctx->vals_a.as<uint32_t>());
  HIPCHK(hipGetLastError());
  size_t tmp_bytes = 0;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ctx->map_keys_a.as<unsigned long long>(),
                                            ctx->map_keys_b.as<unsigned long long>(), ctx->vals_a.as<uint32_t>(),
                                            ctx->vals_b.as<uint32_t>(), (int)nslots, 0, sort_bits, ctx->stream));
  HIPCHK(ctx->cub_tmp.ensure(tmp_bytes));
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp.p, tmp_bytes, ctx->map_keys_a.as<unsigned long long>(),
                                            ctx->map_keys_b.as<unsigned long long>(), ctx->vals_a.as<uint32_t>(),
                                            ctx->vals_b.as<uint32_t>(), (int)nslots, 0, sort_bits, ctx->stream));
  hipLaunchKernelGGL(k_voxel_heads, dim3(div_up((uint64_t)n + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     ctx->map_keys_b.as<unsigned long long>(), n, ctx->map_flags.as<uint32_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, ctx->map_flags.as<uint32_t>(), ctx->map_seg.as<uint32_t>(),
                                          (int)(n + 1), ctx->stream));
  HIPCHK(ctx->cub_tmp.ensure(tmp_bytes));
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(ctx->cub_tmp.p, tmp_bytes, ctx->map_flags.as<uint32_t>(),
                                          ctx->map_seg.as<uint32_t>(), (int)(n + 1), ctx->stream));
  uint32_t nvox = 0;
  HIPCHK(hipMemcpyAsync(&nvox, ctx->map_seg.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx->raw.ensure((size_t)nvox * 12));
  const uint32_t nblk = div_up(n, kVoxRun);
  HIPCHK(ctx->map_acc.ensure((size_t)nvox * sizeof(VoxelSum)));
  HIPCHK(ctx->map_pre.ensure((size_t)nblk * sizeof(VoxelSum)));
  HIPCHK(ctx->map_heads.ensure(((size_t)nvox + 1) * 4));
  hipLaunchKernelGGL(k_voxel_pieces, dim3(div_up(nblk, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->vals_b.as<uint32_t>(),
                     ctx->map_flags.as<uint32_t>(), ctx->map_seg.as<uint32_t>(), n, ctx->map_pts.as<float4>(),
                     ctx->map_acc.as<VoxelSum>(), ctx->map_pre.as<VoxelSum>(), ctx->map_heads.as<uint32_t>());
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(k_voxel_finish, dim3(div_up(nvox, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->map_acc.as<VoxelSum>(),
                     ctx->map_pre.as<VoxelSum>(), ctx->map_heads.as<uint32_t>(), nvox, ctx->map_origin[0],
                     ctx->map_origin[1], ctx->map_origin[2], ctx->raw.as<float>());
  HIPCHK(hipGetLastError());
  if (xyz_out) {
    if (*n_io < nvox) { *n_io = nvox; return ctx->fail(MA_ERR_INVALID, "output capacity too small"); }
    HIPCHK(hipMemcpyAsync(xyz_out, ctx->raw.p, (size_t)nvox * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  *n_io = nvox;
  if (as_gold) {
    if (nvox >= (1u << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "map has 2^28 or more voxels");
    ctx->have_gold = false;
    ctx->have_src = false;
    ctx->unproj_pending = false;
    ctx->state_valid = false;
    return build_gold_from_raw(ctx, nvox, cell);   // ctx->raw already holds the voxel centroids
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
  });
}

int ma_lift_boxes(ma_ctx* ctx, const void* depth, int dtype, int width, int height, const float K[4],
                  const double T[16], float depth_scale, const float* boxes_xyxy, uint64_t n_boxes, int z_rule,
                  double* corners_out, uint8_t* valid_out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (z_rule != MA_Z_CORNER && z_rule != MA_Z_MEDIAN) return ctx->fail(MA_ERR_INVALID, "z_rule must be MA_Z_CORNER or MA_Z_MEDIAN");
  if (!K || !T || (!boxes_xyxy && n_boxes) || (!corners_out && n_boxes)) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (n_boxes == 0) return MA_OK;
  if (n_boxes > (1ull << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "too many boxes");
  if (!(K[0] != 0.f) || !(K[1] != 0.f) || !(depth_scale > 0.f) || !std::isfinite(depth_scale)) return ctx->fail(MA_ERR_INVALID, "bad intrinsics / depth_scale");
  for (int i = 0; i < 16; ++i) if (!std::isfinite(T[i])) return ctx->fail(MA_ERR_INVALID, "non-finite camera pose");
  HIPCHK(hipSetDevice(ctx->device));
  if (depth) {
    if (dtype != 0 && dtype != 1) return ctx->fail(MA_ERR_INVALID, "dtype must be 0 (float32) or 1 (uint16)");
    if (width <= 0 || height <= 0 || (uint64_t)width * height >= 0xFFFFFFFEull) return ctx->fail(MA_ERR_INVALID, "bad image size");
    const size_t bytes = (size_t)width * height * (dtype == 0 ? 4 : 2);
    HIPCHK(ctx->depth_in.ensure(bytes));
    HIPCHK(hipMemcpyAsync(ctx->depth_in.p, depth, bytes, hipMemcpyDefault, ctx->stream));
    ctx->depth_w = width; ctx->depth_h = height; ctx->depth_dtype = dtype;
  } else if (ctx->depth_dtype < 0) {
    return ctx->fail(MA_ERR_STATE, "no depth frame uploaded yet");
  }
  Pinhole cam{};
  cam.fx_inv = 1.0 / (double)K[0];
  cam.fy_inv = 1.0 / (double)K[1];
  cam.cx = K[2];
  cam.cy = K[3];
  cam.inv_scale = 1.0 / (double)depth_scale;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) cam.T[4 * r + c] = T[4 * r + c];
  const uint32_t nc = (uint32_t)n_boxes * 4u;
  HIPCHK(ctx->raw.ensure((size_t)n_boxes * 16));
  HIPCHK(ctx->misc.ensure((size_t)nc * 24 + nc + 8 + (size_t)n_boxes * 8));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, boxes_xyxy, (size_t)n_boxes * 16, hipMemcpyHostToDevice, ctx->stream));
  double* d_out = ctx->misc.as<double>();
  double* d_zmed = d_out + (size_t)nc * 3;                               // one median depth per box (depth rule 1)
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_zmed + n_boxes);
  const double* zmed = z_rule == MA_Z_MEDIAN ? d_zmed : nullptr;
  if (ctx->depth_dtype == 0) {
    if (zmed) hipLaunchKernelGGL(k_box_median<float>, dim3((uint32_t)n_boxes), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<float>(),
                                 ctx->depth_w, ctx->depth_h, 1u, ctx->raw.as<float>(), (const uint32_t*)nullptr, cam.inv_scale, d_zmed);
    hipLaunchKernelGGL(k_lift_boxes<float>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<float>(),
                       ctx->depth_w, ctx->depth_h, cam, ctx->raw.as<float>(), nc, zmed, d_out, d_valid);
  } else {
    if (zmed) hipLaunchKernelGGL(k_box_median<uint16_t>, dim3((uint32_t)n_boxes), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<uint16_t>(),
                                 ctx->depth_w, ctx->depth_h, 1u, ctx->raw.as<float>(), (const uint32_t*)nullptr, cam.inv_scale, d_zmed);
    hipLaunchKernelGGL(k_lift_boxes<uint16_t>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, ctx->stream,
                       ctx->depth_in.as<uint16_t>(), ctx->depth_w, ctx->depth_h, cam, ctx->raw.as<float>(), nc, zmed, d_out, d_valid);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(corners_out, d_out, (size_t)nc * 24, hipMemcpyDeviceToHost, ctx->stream));
  if (valid_out) HIPCHK(hipMemcpyAsync(valid_out, d_valid, nc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
  });
}

int ma_get_source(ma_ctx* ctx, float* xyz_out, uint64_t* n) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !n) return MA_ERR_INVALID;
  if (!ctx->have_src) return ctx->fail(MA_ERR_STATE, "source cloud not set");
  if (ctx->unproj_pending) { HIPCHK(hipSetDevice(ctx->device)); const int rc0 = finish_unproject(ctx); if (rc0) return rc0; }
  if (!xyz_out) { *n = ctx->n_pts; return MA_OK; }
  if (*n < ctx->n_pts) { *n = ctx->n_pts; return ctx->fail(MA_ERR_INVALID, "output capacity too small"); }
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t* map = nullptr;      // storage order; a cloud from a depth frame is compacted (valid pixels, row-major)
  if (ctx->src_holes) { int rc = source_map(ctx, &map); if (rc) return rc; }
  HIPCHK(ctx->raw.ensure((size_t)ctx->n_pts * 12));
  hipLaunchKernelGGL(k_uncentre, dim3(div_up(ctx->n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->src.as<float4>(), map,
                     ctx->n, ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->raw.as<float>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(xyz_out, ctx->raw.p, (size_t)ctx->n_pts * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n = ctx->n_pts;
  return MA_OK;
  });
}

static int export_stored(ma_ctx* ctx, const float4* pts, const uint32_t* perm, uint32_t slots, uint32_t count, double* xyz_out,
                         uint64_t* n) {
  if (!xyz_out) { *n = count; return MA_OK; }
  if (*n < count) { *n = count; return ctx->fail(MA_ERR_INVALID, "output capacity too small"); }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(ctx->raw64.ensure((size_t)count * 24));
  hipLaunchKernelGGL(k_export_f64, dim3(div_up(slots, kBlock)), dim3(kBlock), 0, ctx->stream, pts, perm, slots,
                     ctx->origin[0], ctx->origin[1], ctx->origin[2], ctx->raw64.as<double>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(xyz_out, ctx->raw64.p, (size_t)count * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n = count;
  return MA_OK;
}

int ma_get_gold(ma_ctx* ctx, double* xyz_out, uint64_t* n) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !n) return MA_ERR_INVALID;
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "gold map not set");
  return export_stored(ctx, ctx->gold[0].as<float4>(), ctx->perm0.as<uint32_t>(), ctx->m, ctx->m, xyz_out, n);
  });
}

int ma_get_source_stored(ma_ctx* ctx, double* xyz_out, uint64_t* n) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !n) return MA_ERR_INVALID;
  if (!ctx->have_src) return ctx->fail(MA_ERR_STATE, "source cloud not set");
  if (ctx->unproj_pending) { HIPCHK(hipSetDevice(ctx->device)); const int rc0 = finish_unproject(ctx); if (rc0) return rc0; }
  const uint32_t* map = nullptr;
  if (xyz_out) { HIPCHK(hipSetDevice(ctx->device)); int rc = source_map(ctx, &map); if (rc) return rc; }
  return export_stored(ctx, ctx->src.as<float4>(), map, ctx->n, ctx->n_pts, xyz_out, n);
  });
}

// The loop of ma_icp and ma_icp_sharded (rows a4-a7).  One pair on this context; `sharded`: this context holds a shard
// of the comparison cloud and the 17 sums of every iteration are all-reduced over the communicator's ranks.
//   per iteration, all on the ctx stream, no host synchronisation:
//     unsharded   k_nn_accum (search | search + write lists | read lists)  ->  k_reduce_solve
//     sharded     k_nn_accum (the same three kinds, local shard)  ->  k_reduce19 (+ schedule ranking)
//                 ->  ncclAllReduce(sum, 17 x f64, in place)  ->  k_solve (redundantly on every rank)
// Identical sums give identical states on all ranks, so nothing is broadcast; each rank plans its own list launches
// (its shard has its own box) — which kernel a rank launches changes no bit of its sums, and every rank issues exactly
// one collective per iteration.
static int icp_loop(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16], ma_stats* stats,
                    const bool sharded) {
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if ((rc = need_ready(ctx))) return rc;
  if (sharded && !ctx->comm) return ctx->fail(MA_ERR_STATE, "no communicator (ma_comm_init)");
  if (!T_out) return ctx->fail(MA_ERR_INVALID, "T_out is NULL");
  if (init) for (int i = 0; i < 16; ++i) if (!std::isfinite(init[i])) return ctx->fail(MA_ERR_INVALID, "non-finite init");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, params->max_corr_dist))) return rc;
  if ((rc = ensure_loop_buffers(ctx, params->max_iter))) return rc;
  const bool lists_allowed = lists_enabled(ctx) && !(params->flags & MA_FLAG_NO_LISTS);
  const bool long_run = params->max_iter > 3 * kPlan && params->max_iter <= kMaxSlots;
  if (ctx->src_box_pending && lists_allowed && long_run) {
    // a source cloud from a depth frame: its box (stored, centred coordinates; holes skipped) for the candidate lists
    float lo[3], hi[3];
    if ((rc = cloud_bounds(ctx, ctx->src.as<float>(), ctx->n, lo, hi, 4, true))) return rc;
    double r2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double pad = 1e-4 + 1e-6 * (std::fabs((double)lo[a]) + std::fabs((double)hi[a]));
      ctx->src_lo[a] = (float)((double)lo[a] - pad);
      ctx->src_hi[a] = (float)((double)hi[a] + pad);
      r2 += std::max((double)lo[a] * lo[a], (double)hi[a] * hi[a]);
    }
    ctx->src_rs = (float)(std::sqrt(r2) * 1.001 + 1e-4);
    ctx->src_box_pending = false;
  }
  if ((rc = upload_state(ctx, init))) return rc;
  const float best2 = open_gate2(params->max_corr_dist);
  const bool count = (params->flags & MA_FLAG_COUNT_WORK) != 0;
  const bool timek = (params->flags & MA_FLAG_TIME_KERNELS) != 0 && !sharded;
  if (timek) {
    while ((int)ctx->kev.size() < 2 * params->max_iter + 1) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      ctx->kev.push_back(e);
    }
  }
  RcclApi* api = sharded ? rccl_api() : nullptr;
  ncclComm_t comm = sharded ? static_cast<ncclComm_t>(ctx->comm) : nullptr;
  // With a convergence test (tol > 0) the loop usually stops long before max_iter, and every launch after that only
  // finds `done` set (3-6 us each).  Two ways out:
  //  * SLOTS (runs of more than 3 kPlan iterations; the same mechanism plans the candidate lists): every solve leaves
  //    the state it produced in a pinned slot of its own (iteration k -> slot k; posted writes the solve does not wait
  //    for).  Before batch b (first iteration k = b kPlan) is enqueued the host waits until the last solve of batch b-2
  //    — iteration j = k - kPlan - 1 — has written its slot; batch b-1 stays queued behind it, so the device never
  //    idles.  The loop stops enqueueing before batch b exactly when it stopped at an iteration <= j: a rule on the
  //    STATES, not on when the host happened to look, so every rank of a sharded loop issues the same launches and
  //    the same collectives (a rank that enqueued one all-reduce more than its peers would hang them all).
  //  * BATCHES of 8 with an event (short unsharded runs): before batch b, wait for the end of batch b-2 and read the
  //    pinned word the solve sets when it stops.  A short sharded run simply runs its <= 12 iterations.
  // tol = 0 without lists (and MA_FLAG_TIME_KERNELS runs) take part in neither.
  constexpr int kBatch = 8;
  const bool early_exit = params->tol > 0.0 && !timek;
  // (lists stop paying where the clouds outgrow the caches: at 16M points a list launch costs what a converged search
  // launch costs — its gathers miss — and the two list-writing launches are pure loss: 55.0 against 53.2 ms per 50
  // iterations; 8M gains 1 %, 1M-4M 3-4 %.  profiles/r04_variants_ab.txt, call 24)
  constexpr uint32_t kListMaxPoints = 12u << 20;
  const bool lists_wanted = ctx->src_rs > 0.f && lists_allowed && long_run && ctx->n <= kListMaxPoints && ctx->m <= kListMaxPoints;
  const bool slots_used = long_run && (lists_wanted || early_exit);    // the solves mirror their state into the slots, not into host_state_out
  const bool batch_exit = early_exit && !slots_used && !sharded;
  if (batch_exit) {
    const int need = (params->max_iter + kBatch - 1) / kBatch;
    while ((int)ctx->batch_ev.size() < need) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->batch_ev.push_back(e);
    }
  }
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  if (timek) HIPCHK(hipEventRecord(ctx->kev[0], ctx->stream));
  struct ListsScope { ma_ctx* c; ~ListsScope() { c->lists_on = false; } } lists_scope{ctx};
  // Candidate lists (k_nn_accum<.,.,1|2>).  WHICH kernel an iteration launches is decided here, on the host, in
  // batches of kPlan iterations, from the slots: how far each solve moved the cloud, and how far it has moved since the
  // lists were written (largest displacement over the corners of the source cloud's bounding box: |dR s + dt| is
  // convex in s, so the box's maximum is at a corner).  The slots belong to fixed iterations, so the plan is the same in
  // every run.  It only has to be good, not right: a list launch proves every correspondence it takes from a list and
  // searches for the others, and lists on / off give the same transform bit for bit (tests/test_gpu_lists.py).
  ctx->lists_on = lists_wanted;
  ctx->list_builds = ctx->list_uses = 0;
  ctx->list_plan.clear();
  for (int i = 0; i < 4; ++i) ctx->plan_diag[i] = 0;
  if (slots_used) {
    for (int i = 0; i < params->max_iter; ++i) {      // nothing of an earlier run may pass for this run's slot
      StateSlot& sl = ctx->host_slots[i];
      sl.seq = -1; sl.s.iter = -1; sl.chk = ~0ull; sl.s.R[0] = NAN;
    }
  }
  int launched = 0;
  const double skin = (double)ctx->grid.c / (double)list_skin_div(ctx);
  ma::ListPlanner planner;       // list_planner.h: the decisions themselves (CPU-tested by tests/test_list_planner.py)
  planner.configure(skin, params->max_iter, ctx->src_lo, ctx->src_hi, ctx->cfg_min_uses);
  const DevState T_first = *ctx->host_state_in;     // the transform launch 0 runs with
  auto T_of = [&](int k) -> const DevState& {        // the transform launch k runs with (k <= the last slot read + 1)
    return k == 0 ? T_first : const_cast<const DevState&>(ctx->host_slots[k - 1].s);
  };
  volatile int* const host_done = ctx->host_done;
  int plan[kPlan] = {0, 0, 0, 0};
  bool slots_live = slots_used;   // false once a slot failed to arrive (or the loop stopped under MA_FLAG_TIME_KERNELS)
  bool sync_mode = false;         // after a planner timeout: drain the stream before every batch
  for (int k = 0; k < params->max_iter; ++k) {
    if (batch_exit && k % kBatch == 0 && k >= 2 * kBatch) {
      HIPCHK(hipEventSynchronize(ctx->batch_ev[k / kBatch - 2]));
      if (*host_done) break;
    }
    if (slots_live && k % kPlan == 0) {
      for (int i = 0; i < kPlan; ++i) plan[i] = 0;
      if (k >= 2 * kPlan) {
        const int j = k - kPlan - 1;               // the last iteration of batch b-2
        const StateSlot& sl = ctx->host_slots[j];
        // Wait for slot j: spin for ~20 us (an event wait wakes up later than the queued batch lasts), then yield the
        // core between looks — under a CPU quota or with several contexts per core a spinning planner starves the
        // very threads that feed the other streams; a descheduled planner only idles the device two batches later.
        ++ctx->plan_diag[0];
        const auto t_wait = std::chrono::steady_clock::now();
        int state = 0;                             // 1: slot j is here; 2: the loop stopped before j; 3: timeout
        bool yielding = false;
        for (uint32_t spins = 0;; ++spins) {
          // (list_planner.h: slot_wait_state — the slot, or the loop stopped BEFORE j and the slot never comes; stopped
          // AT j its slot is still to come, with `done` set in it)
          const ma::SlotWait w = ma::slot_wait_state(slot_ready(&sl, j + 1), *host_done, j);
          if (w == ma::SlotWait::kSlotHere) { state = 1; break; }
          if (w == ma::SlotWait::kStoppedBefore) { state = 2; break; }
          if (yielding || (spins & 63u) == 63u) {
            const auto waited = std::chrono::steady_clock::now() - t_wait;
            if (waited > std::chrono::microseconds(ctx->cfg_planner_timeout_us)) { state = 3; break; }
            if (waited > std::chrono::microseconds(20)) {
              if (!yielding) { yielding = true; ++ctx->plan_diag[1]; }
              if (waited > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
              else std::this_thread::yield();
            }
          }
          __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        bool stop_here = false;                    // the loop stopped at an iteration <= j
        if (state == 3) {
          // no sign of iteration j for 20 s: the posted writes are not arriving (or the device is wedged).  Not
          // silent: counted in the diagnostics (ma_get_planner_diag); the device is waited for, and the rest of the
          // run searches and decides its exit with everything enqueued so far complete (sync_mode).
          ++ctx->plan_diag[2];
          planner.stop();
          ctx->lists_on = false;
          slots_live = false;
          sync_mode = true;
        } else if (ma::stops_before_batch(state == 2 ? ma::SlotWait::kStoppedBefore : ma::SlotWait::kSlotHere,
                                          const_cast<const DevState&>(sl.s).done != 0)) {
          stop_here = true;
        } else if (ctx->lists_on) {
          planner.plan_batch(k, T_of, plan);
        }
        if (stop_here) {
          if (!timek) break;                       // (a timed run keeps its launch count: the rest only find `done` set)
          planner.stop();
          ctx->lists_on = false;
          slots_live = false;
        }
      }
    }
    if (sync_mode && k % kPlan == 0 && k >= 2 * kPlan) {
      // after a planner timeout: the same rule as above — stop before this batch iff the loop stopped at an iteration
      // <= j — decided with the stream drained, so that a sharded rank still stops where its peers stop
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (ma::iterations_enqueued(params->max_iter, *host_done, true) <= k && !timek) break;
    }
    const int kind = ctx->lists_on ? plan[k % kPlan] : 0;
    ctx->list_plan.push_back((uint8_t)kind);
    if ((rc = launch_nn<0>(ctx, count, best2, nullptr, nullptr, MA_SCHED != 0, kind, k > 0 ? 1 : 0))) return rc;
    if (timek) HIPCHK(hipEventRecord(ctx->kev[2 * k + 1], ctx->stream));
    StateSlot* slot = slots_used ? &ctx->host_slots[k] : nullptr;
    if (sharded) {
      if ((rc = launch_reduce(ctx, MA_SCHED != 0))) return rc;
      NCCLCHK(api->AllReduce(ctx->acc19.p, ctx->acc19.p, MA_ACC_LEN, ncclDouble, ncclSum, comm, ctx->stream));
      if ((rc = launch_solve(ctx, params->tol, true, slot))) return rc;
    } else {
      if ((rc = launch_reduce_solve(ctx, params->tol, slot))) return rc;
    }
    launched = k + 1;
    if (timek) HIPCHK(hipEventRecord(ctx->kev[2 * k + 2], ctx->stream));
    if (batch_exit && (k + 1) % kBatch == 0) HIPCHK(hipEventRecord(ctx->batch_ev[k / kBatch], ctx->stream));
  }
  ctx->plan_diag[3] = launched;
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (slots_used) {      // the state as the last solve that ran left it (the solve that stops a loop writes the last slot)
    *ctx->host_state_out = *ctx->host_state_in;
    for (int i = launched - 1; i >= 0; --i)
      if (ctx->host_slots[i].s.iter == i + 1) { *ctx->host_state_out = const_cast<const DevState&>(ctx->host_slots[i].s); break; }
  }
  const DevState s = *ctx->host_state_out;     // the last solve's pinned copy of the state: no device-to-host copy
  // launches that ran as iterations (a loop stopped by its convergence test leaves up to 2 kPlan enqueued launches that
  // only find `done` set)
  for (int i = 0; i < s.iter && i < (int)ctx->list_plan.size(); ++i) {
    ctx->list_builds += ctx->list_plan[(size_t)i] == 1;
    ctx->list_uses += ctx->list_plan[(size_t)i] == 2;
  }
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (timek) {
    ctx->ktimes[0] = ctx->ktimes[1] = ctx->ktimes[2] = ctx->ktimes[3] = 0;
    ctx->iter_ms.clear();
    for (int k = 0; k < s.iter && k < params->max_iter; ++k) {   // launches after `done` are no-ops: skip them
      float a = 0, b = 0;
      HIPCHK(hipEventElapsedTime(&a, ctx->kev[2 * k], ctx->kev[2 * k + 1]));
      HIPCHK(hipEventElapsedTime(&b, ctx->kev[2 * k + 1], ctx->kev[2 * k + 2]));
      ctx->iter_ms.push_back(a);
      ctx->iter_ms.push_back(b);
      ctx->ktimes[0] += a; ctx->ktimes[1] += 1;
      ctx->ktimes[2] += b; ctx->ktimes[3] += 1;
    }
  }
  centred_to_world(s, ctx->origin, T_out);
  fill_stats(s, ms, stats);
  if (s.degenerate) return ctx->fail(MA_ERR_DEGENERATE, "no correspondence inside max_corr_dist, or rank-deficient covariance (fewer than 3 non-collinear pairs)");
  return MA_OK;
}

int ma_icp(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16], ma_stats* stats) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  return icp_loop(ctx, params, init, T_out, stats, false);
  });
}

int ma_list_config(ma_ctx* ctx, int enabled, float skin_div, int min_uses) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (skin_div != 0.f && !(skin_div >= 2.f && skin_div <= 64.f)) return ctx->fail(MA_ERR_INVALID, "skin_div must be 0 (keep) or in [2, 64]");
  if (min_uses != 0 && (min_uses < 1 || min_uses > 64)) return ctx->fail(MA_ERR_INVALID, "min_uses must be 0 (keep) or in [1, 64]");
  if (enabled >= 0) ctx->cfg_lists = enabled != 0;
  if (skin_div != 0.f) ctx->cfg_skin_div = skin_div;
  if (min_uses != 0) ctx->cfg_min_uses = min_uses;
  return MA_OK;
  });
}

int ma_get_planner_diag(ma_ctx* ctx, int32_t out[4]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return MA_ERR_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = ctx->plan_diag[i];
  return MA_OK;
  });
}

int ma_get_list_stats(ma_ctx* ctx, int32_t out[2]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return MA_ERR_INVALID;
  out[0] = ctx->list_builds;
  out[1] = ctx->list_uses;
  return MA_OK;
  });
}

int ma_get_list_plan(ma_ctx* ctx, uint8_t* kinds, int max_iterations) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !kinds || max_iterations < 0) return MA_ERR_INVALID;
  const int n = std::min<int>(max_iterations, (int)ctx->list_plan.size());
  for (int i = 0; i < n; ++i) kinds[i] = ctx->list_plan[(size_t)i];
  return n;
  });
}

int ma_get_trace(ma_ctx* ctx, double* out, int max_rows) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out || max_rows < 0) return MA_ERR_INVALID;
  if (!ctx->state.p || !ctx->trace.p) return ctx->fail(MA_ERR_STATE, "no iteration has run");
  HIPCHK(hipSetDevice(ctx->device));
  DevState s;
  int rc = fetch_state(ctx, &s);
  if (rc) return rc;
  const int rows = std::min({s.iter, max_rows, ctx->trace_cap});
  if (rows == 0) return 0;
  std::vector<double> h((size_t)rows * MA_TRACE_STRIDE);
  HIPCHK(hipMemcpy(h.data(), ctx->trace.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int r = 0; r < rows; ++r) {
    const double* in = &h[(size_t)r * MA_TRACE_STRIDE];
    double* o = out + (size_t)r * MA_TRACE_STRIDE;
    acc_to_world(in, ctx->origin, o);
    o[17] = in[17]; o[18] = in[18]; o[19] = in[19];
  }
  return rows;
  });
}

#if MA_DEBUG_FETCH   // debug build only (tools/tie_debug.py): what 0 = prev[] (match of every source slot, position in the gold map's
// cell order), 1 = src_perm (slot -> caller's index; identity if unsorted), 2 = perm0 (position -> caller's gold index)
int ma_debug_fetch(ma_ctx* ctx, int what, uint32_t* out, uint64_t count) {
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const void* src = what == 0 ? ctx->prev.p : (what == 1 ? ctx->src_perm.p : ctx->perm0.p);
  if (!src) return ctx->fail(MA_ERR_STATE, "nothing to fetch");
  HIPCHK(hipMemcpy(out, src, count * 4, hipMemcpyDeviceToHost));
  return MA_OK;
}
#endif

#if MA_WGTIMES
int ma_debug_wgphases(ma_ctx* ctx, uint32_t* ph) {
  HIPCHK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_wgph), sizeof(g_wgph)));
  return MA_OK;
}
int ma_debug_rstimes(ma_ctx* ctx, unsigned long long* rs) {
  HIPCHK(hipMemcpyFromSymbol(rs, HIP_SYMBOL(g_rs), sizeof(g_rs)));
  return MA_OK;
}
int ma_debug_wgtimes(ma_ctx* ctx, unsigned long long* times, uint32_t* hw) {
  HIPCHK(hipMemcpyFromSymbol(times, HIP_SYMBOL(g_wgt), sizeof(g_wgt)));
  HIPCHK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wghw), sizeof(g_wghw)));
  return MA_OK;
}
#endif

int ma_get_kernel_times(ma_ctx* ctx, double out[4]) {
  if (!ctx || !out) return MA_ERR_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = ctx->ktimes[i];
  return MA_OK;
}

int ma_get_iteration_times(ma_ctx* ctx, float* out, int max_iterations) {
  if (!ctx || !out || max_iterations < 0) return MA_ERR_INVALID;
  const int n = std::min<int>(max_iterations, (int)(ctx->iter_ms.size() / 2));
  for (int i = 0; i < 2 * n; ++i) out[i] = ctx->iter_ms[(size_t)i];
  return n;
}

int ma_icp_step(ma_ctx* ctx, const ma_params* params, const double T_in[16], double acc_out[MA_ACC_LEN],
                double T_out[16]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if ((rc = need_ready(ctx))) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, params->max_corr_dist))) return rc;
  if ((rc = upload_state(ctx, T_in))) return rc;
  if ((rc = launch_nn<0>(ctx, (params->flags & MA_FLAG_COUNT_WORK) != 0, open_gate2(params->max_corr_dist), nullptr, nullptr))) return rc;
  if ((rc = launch_reduce(ctx))) return rc;
  if ((rc = launch_solve(ctx, params->tol))) return rc;
  double a[kNAcc];
  HIPCHK(hipMemcpyAsync(a, ctx->acc19.p, sizeof(a), hipMemcpyDeviceToHost, ctx->stream));
  DevState s;
  if ((rc = fetch_state(ctx, &s))) return rc;
  if (acc_out) acc_to_world(a, ctx->origin, acc_out);
  if (T_out) centred_to_world(s, ctx->origin, T_out);
  if (s.degenerate) return ctx->fail(MA_ERR_DEGENERATE, "no correspondence inside max_corr_dist, or rank-deficient covariance (fewer than 3 non-collinear pairs)");
  return MA_OK;
  });
}

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
  hipLaunchKernelGGL(k_moments, dim3(nb), dim3(kBlock), 0, ctx->stream, pts, n, rows);
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

int ma_nn(ma_ctx* ctx, const double T[16], double max_corr_dist, uint32_t* idx_out, float* d2_out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  int rc;
  if ((rc = need_ready(ctx))) return rc;
  if (!(max_corr_dist > 0.0)) return ctx->fail(MA_ERR_INVALID, "max_corr_dist must be > 0 (or +inf)");
  if (!idx_out && !d2_out) return ctx->fail(MA_ERR_INVALID, "no output requested");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, max_corr_dist))) return rc;
  if ((rc = upload_state(ctx, T))) return rc;
  if ((rc = finish_unproject(ctx))) return rc;
  const size_t n = ctx->n, n_out = ctx->n_pts;     // slots searched; points reported (holes of a depth frame are not)
  HIPCHK(ctx->nn_idx.ensure(n * 4));
  HIPCHK(ctx->nn_d2.ensure(n * 4));
  HIPCHK(ctx->nn_tmp.ensure(n * 4));
  if ((rc = launch_nn<1>(ctx, false, open_gate2(max_corr_dist), ctx->nn_idx.as<uint32_t>(), ctx->nn_d2.as<float>()))) return rc;
  const uint32_t* perm = nullptr;
  if ((rc = source_map(ctx, &perm))) return rc;
  if (idx_out) {
    hipLaunchKernelGGL(k_unpermute<uint32_t>, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       ctx->nn_idx.as<uint32_t>(), perm, (uint32_t)n, ctx->nn_tmp.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(idx_out, ctx->nn_tmp.p, n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (d2_out) {
    hipLaunchKernelGGL(k_unpermute<float>, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream,
                       ctx->nn_d2.as<float>(), perm, (uint32_t)n, ctx->nn_tmp.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d2_out, ctx->nn_tmp.p, n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return MA_OK;
  });
}

int ma_state_set(ma_ctx* ctx, const double T[16]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->have_gold) return ctx->fail(MA_ERR_STATE, "gold map not set");
  if (T) for (int i = 0; i < 16; ++i) if (!std::isfinite(T[i])) return ctx->fail(MA_ERR_INVALID, "non-finite transform");
  HIPCHK(hipSetDevice(ctx->device));
  return upload_state(ctx, T);
  });
}

int ma_state_get(ma_ctx* ctx, double T_out[16], ma_stats* stats) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->state.p) return ctx->fail(MA_ERR_STATE, "no state (ma_state_set / ma_icp first)");
  HIPCHK(hipSetDevice(ctx->device));
  DevState s;
  int rc = fetch_state(ctx, &s);
  if (rc) return rc;
  if (T_out) centred_to_world(s, ctx->origin, T_out);
  fill_stats(s, 0.0, stats);
  return MA_OK;
  });
}

int ma_step_accumulate(ma_ctx* ctx, const ma_params* params, void* acc_dev) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if ((rc = need_ready(ctx))) return rc;
  if (!ctx->state_valid) return ctx->fail(MA_ERR_STATE, "no valid stepping state: call ma_state_set after the clouds are set (a new gold map or source cloud invalidates it)");
  if (!acc_dev) return ctx->fail(MA_ERR_INVALID, "acc_dev is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, params->max_corr_dist))) return rc;
  if ((rc = launch_nn<0>(ctx, (params->flags & MA_FLAG_COUNT_WORK) != 0, open_gate2(params->max_corr_dist), nullptr, nullptr))) return rc;
  if ((rc = launch_reduce(ctx))) return rc;
  HIPCHK(hipMemcpyAsync(acc_dev, ctx->acc19.p, MA_ACC_LEN * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return MA_OK;
  });
}

int ma_step_solve(ma_ctx* ctx, const ma_params* params, const void* acc_dev) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if (!ctx->state_valid) return ctx->fail(MA_ERR_STATE, "no valid stepping state: call ma_state_set after the clouds are set (a new gold map or source cloud invalidates it)");
  if (!acc_dev) return ctx->fail(MA_ERR_INVALID, "acc_dev is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  // rows 17,18 (work counters) keep the local values of the accumulate step
  HIPCHK(hipMemcpyAsync(ctx->acc19.p, acc_dev, MA_ACC_LEN * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return launch_solve(ctx, params->tol);
  });
}

int ma_search_geometry(const ma_ctx* ctx, uint32_t* n_workgroups, uint32_t* block_threads) {
  if (!ctx || !n_workgroups || !block_threads) return MA_ERR_INVALID;
  if (!ctx->have_src) return MA_ERR_STATE;
  *n_workgroups = grid_for(ctx->n);
  *block_threads = (uint32_t)kNNBlock;
  return MA_OK;
}

int ma_mem_info_get(const ma_ctx* ctx, ma_mem_info* out) {
  if (!ctx || !out) return MA_ERR_INVALID;
  out->device_allocs = ctx->mem.allocs;
  out->device_frees = ctx->mem.frees;
  out->bytes_live = ctx->mem.bytes_live;
  out->bytes_peak = ctx->mem.bytes_peak;
  return MA_OK;
}

int ma_last_device_ms(const ma_ctx* ctx_c, double* ms) {
  ma_ctx* ctx = const_cast<ma_ctx*>(ctx_c);
  return guarded(ctx, [&]() -> int {
  if (!ctx || !ms) return MA_ERR_INVALID;
  if (ctx->unproj_pending) {          // the launch may still be running: wait for it (and settle its count on the way)
    HIPCHK(hipSetDevice(ctx->device));
    const int rc = finish_unproject(ctx);
    if (rc && rc != MA_ERR_EMPTY) return rc;
  }
  *ms = ctx->last_op_ms;
  return MA_OK;
  });
}

int ma_lift_boxes_batch(ma_ctx* ctx, const void* depth_frames, int dtype, int width, int height, uint32_t n_frames,
                        const float K[4], const double* T_cam_to_world, float depth_scale, const float* boxes_xyxy,
                        const uint32_t* box_frame, uint64_t n_boxes, int z_rule, double* corners_out, uint8_t* valid_out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (z_rule != MA_Z_CORNER && z_rule != MA_Z_MEDIAN) return ctx->fail(MA_ERR_INVALID, "z_rule must be MA_Z_CORNER or MA_Z_MEDIAN");
  if (!depth_frames || !K || !T_cam_to_world || (!boxes_xyxy && n_boxes) || (!box_frame && n_boxes) || (!corners_out && n_boxes))
    return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (dtype != 0 && dtype != 1) return ctx->fail(MA_ERR_INVALID, "dtype must be 0 (float32) or 1 (uint16)");
  if (width <= 0 || height <= 0 || n_frames == 0 || (uint64_t)width * height * n_frames >= (1ull << 40)) return ctx->fail(MA_ERR_INVALID, "bad image size / frame count");
  if (n_boxes > (1ull << 28)) return ctx->fail(MA_ERR_TOO_LARGE, "too many boxes");
  if (!(K[0] != 0.f) || !(K[1] != 0.f) || !(depth_scale > 0.f) || !std::isfinite(depth_scale)) return ctx->fail(MA_ERR_INVALID, "bad intrinsics / depth_scale");
  for (uint64_t i = 0; i < 16ull * n_frames; ++i) if (!std::isfinite(T_cam_to_world[i])) return ctx->fail(MA_ERR_INVALID, "non-finite camera pose");
  for (uint64_t b = 0; b < n_boxes; ++b) if (box_frame[b] >= n_frames) return ctx->fail(MA_ERR_INVALID, "box_frame entry out of range");
  if (n_boxes == 0) return MA_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t esz = dtype == 0 ? 4 : 2, frame_bytes = (size_t)width * height * esz;
  std::vector<double> poses((size_t)n_frames * 12);
  for (uint32_t f = 0; f < n_frames; ++f)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) poses[(size_t)f * 12 + 4 * r + c] = T_cam_to_world[(size_t)f * 16 + 4 * r + c];
  const uint32_t nc = (uint32_t)n_boxes * 4u;
  HIPCHK(ctx->lift_frames.ensure(frame_bytes * n_frames));
  HIPCHK(ctx->lift_poses.ensure(poses.size() * 8 + n_boxes * 4));
  HIPCHK(ctx->raw.ensure((size_t)n_boxes * 16));
  HIPCHK(ctx->misc.ensure((size_t)nc * 24 + nc + 8 + (size_t)n_boxes * 8));
  // ONE upload of the whole batch of frames, one of the poses + frame indices, one of the boxes
  HIPCHK(hipMemcpyAsync(ctx->lift_frames.p, depth_frames, frame_bytes * n_frames, hipMemcpyDefault, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->lift_poses.p, poses.data(), poses.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  uint32_t* d_frame = reinterpret_cast<uint32_t*>(ctx->lift_poses.as<double>() + poses.size());
  HIPCHK(hipMemcpyAsync(d_frame, box_frame, n_boxes * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, boxes_xyxy, (size_t)n_boxes * 16, hipMemcpyHostToDevice, ctx->stream));
  double* d_out = ctx->misc.as<double>();
  double* d_zmed = d_out + (size_t)nc * 3;                               // one median depth per box (depth rule 1)
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_zmed + n_boxes);
  const double* zmed = z_rule == MA_Z_MEDIAN ? d_zmed : nullptr;
  const double fx_inv = 1.0 / (double)K[0], fy_inv = 1.0 / (double)K[1], inv_scale = 1.0 / (double)depth_scale;
  if (dtype == 0) {
    if (zmed) hipLaunchKernelGGL(k_box_median<float>, dim3((uint32_t)n_boxes), dim3(kBlock), 0, ctx->stream, ctx->lift_frames.as<float>(),
                                 width, height, n_frames, ctx->raw.as<float>(), d_frame, inv_scale, d_zmed);
    hipLaunchKernelGGL(k_lift_boxes_batch<float>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, ctx->stream,
                       ctx->lift_frames.as<float>(), width, height, n_frames, fx_inv, fy_inv, (double)K[2], (double)K[3],
                       inv_scale, ctx->lift_poses.as<double>(), ctx->raw.as<float>(), d_frame, nc, zmed, d_out, d_valid);
  } else {
    if (zmed) hipLaunchKernelGGL(k_box_median<uint16_t>, dim3((uint32_t)n_boxes), dim3(kBlock), 0, ctx->stream, ctx->lift_frames.as<uint16_t>(),
                                 width, height, n_frames, ctx->raw.as<float>(), d_frame, inv_scale, d_zmed);
    hipLaunchKernelGGL(k_lift_boxes_batch<uint16_t>, dim3(div_up(nc, kBlock)), dim3(kBlock), 0, ctx->stream,
                       ctx->lift_frames.as<uint16_t>(), width, height, n_frames, fx_inv, fy_inv, (double)K[2], (double)K[3],
                       inv_scale, ctx->lift_poses.as<double>(), ctx->raw.as<float>(), d_frame, nc, zmed, d_out, d_valid);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(corners_out, d_out, (size_t)nc * 24, hipMemcpyDeviceToHost, ctx->stream));
  if (valid_out) HIPCHK(hipMemcpyAsync(valid_out, d_valid, nc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // the one synchronisation of the batch
  return MA_OK;
  });
}

// ---- RCCL: communicator bootstrap + the sharded-source loop --------------------------------
int ma_comm_unique_id(void* id_out) {
  return guarded(nullptr, [&]() -> int {
  if (!id_out) return MA_ERR_INVALID;
  RcclApi* api = rccl_api();
  if (!api) { g_create_error = g_rccl.error; return MA_ERR_COMM; }
  ncclUniqueId id;
  const ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return MA_ERR_COMM; }
  static_assert(sizeof(ncclUniqueId) == MA_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id_out, &id, sizeof(id));
  return MA_OK;
  });
}

int ma_comm_init(ma_ctx* ctx, const void* id, int n_ranks, int rank) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return ctx->fail(MA_ERR_INVALID, "bad communicator arguments");
  RcclApi* api = rccl_api();
  if (!api) return ctx->fail(MA_ERR_COMM, g_rccl.error);
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->comm) comm_destroy(ctx);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  NCCLCHK(api->CommInitRank(&comm, n_ranks, uid, rank));
  ctx->comm = comm;
  ctx->comm_ranks = n_ranks;
  ctx->comm_rank = rank;
  return MA_OK;
  });
}

int ma_comm_destroy(ma_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  comm_destroy(ctx);
  return MA_OK;
  });
}

int ma_comm_allgather_f64(ma_ctx* ctx, const double* in, uint64_t count, double* out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->comm) return ctx->fail(MA_ERR_STATE, "no communicator (ma_comm_init)");
  if (!in || !out || count == 0 || count > (1ull << 24)) return ctx->fail(MA_ERR_INVALID, "bad gather arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t bytes = count * sizeof(double), total = bytes * (size_t)ctx->comm_ranks;
  HIPCHK(ctx->nn_tmp.ensure(bytes + total));
  char* base = static_cast<char*>(ctx->nn_tmp.p);
  HIPCHK(hipMemcpyAsync(base, in, bytes, hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(rccl_api()->AllGather(base, base + bytes, count, ncclDouble, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
  HIPCHK(hipMemcpyAsync(out, base + bytes, total, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
  });
}

int ma_comm_allreduce_f64(ma_ctx* ctx, double* inout, uint64_t count, int op) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!ctx->comm) return ctx->fail(MA_ERR_STATE, "no communicator (ma_comm_init)");
  if (!inout || count == 0 || count > (1ull << 24) || op < 0 || op > 1) return ctx->fail(MA_ERR_INVALID, "bad reduce arguments");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t bytes = count * sizeof(double);
  HIPCHK(ctx->nn_tmp.ensure(bytes));
  HIPCHK(hipMemcpyAsync(ctx->nn_tmp.p, inout, bytes, hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(rccl_api()->AllReduce(ctx->nn_tmp.p, ctx->nn_tmp.p, count, ncclDouble, op == 0 ? ncclSum : ncclMax,
                                static_cast<ncclComm_t>(ctx->comm), ctx->stream));
  HIPCHK(hipMemcpyAsync(inout, ctx->nn_tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return MA_OK;
  });
}

// One pair over the ranks of the communicator: gold map + grids replicated (every rank called
// ma_set_gold with the same cloud), this rank's shard of the comparison cloud in ma_set_source.
// The loop is icp_loop above — the same planner, list launches, pinned slots and early exit as ma_icp, with
//   k_reduce19 (+ schedule ranking) -> ncclAllReduce(sum, 17 x f64, in place) -> k_solve
// in the place of k_reduce_solve.
int ma_icp_sharded(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16], ma_stats* stats) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  return icp_loop(ctx, params, init, T_out, stats, true);
  });
}

}  // extern "C"
