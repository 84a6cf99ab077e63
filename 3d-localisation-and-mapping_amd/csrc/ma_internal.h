// ma_internal.h — what the translation units of libmapalign.so share: constants, the plain structs that cross the
// host/device line, the small device helpers every kernel file uses, the context (struct ma_ctx), and the host
// functions one unit offers the others.  Internal: nothing here is part of the C ABI (include/mapalign.h).
//
// MI355X (gfx950 / CDNA4) map-alignment engine.  Hot path (the reference body
// src/map_alignment/align.py is absent, call site /root/reference/task_def.py:175-187):
//   kernels_search.hip  k_nn_accum  a4+a5  per source point: rigid transform (fp64), exact nearest gold point through
//                       a uniform grid in HBM (wavefront-cooperative row / chunk work queues in LDS, ring by ring with
//                       distance pruning) or from the query's candidate list, distance gate, the 17 sums
//                       {W, s_p, s_q, S, E}; DPP + LDS reduction -> one partial per workgroup and sum (no float
//                       atomics => bitwise reproducible)
//   kernels_solve.hip   k_reduce_solve a5-a7  fixed-order sum of the partials, one workgroup per accumulator; the
//                       last-arriving workgroup runs the Kabsch step (kabsch.h), the transform composition and the
//                       convergence test on the device, so the loop never synchronises with the host;
//                       k_reduce19 + k_solve: the same two halves as separate launches (sharded-source stepping)
//   kernels_depth.hip   k_unproject a8  depth image -> world points written straight into the source cloud buffer,
//                       pixel i in slot i (invalid pixels leave a hole the search skips); box lifting (a10); the map
//                       builder's voxel kernels (N4)
//   kernels_build.hip   the gold map's grid (cell sort, cell table, distance-to-occupied table), the source cloud's
//                       brick-major sort, fp64 staging
//   host_ctx.hip        context lifetime, cloud upload entry points
//   host_loop.hip       icp_loop: the ICP loop of ma_icp / ma_icp_sharded with the list planner and the pinned slots
//   comm.hip            RCCL shim (dlopen) and the sharded entry points
// Everything is written for 64-wide wavefronts; there is no MFMA because the path is a gather + reduction
// (<0.5 flop/B), and no CPU fallback.
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
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
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
#include "kabsch.h"          // (header-only: kAccLen here, the solve itself in kernels_solve.hip)
#include "list_planner.h"

#ifndef MA_REDUCE_ONEPASS
#define MA_REDUCE_ONEPASS 0      // 1: reduce_chunk with both DPP steps per accumulator (A/B; kernels_search.hip)
#endif
#ifndef MA_SCHED
#define MA_SCHED 1      // longest-first schedule of the search launch's chunks (0: cloud order; A/B)
#endif
#include "mapalign.h"

namespace mai {

constexpr int kBlock = 256;           // 4 wavefronts
constexpr int kNNBlock = ma::kSearchBlock;  // workgroup of the search kernel (MA_NN_BLOCK)
constexpr int kWave = 64;
constexpr int kNAcc = 19;             // 17 accumulators + rows visited + candidates
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;
constexpr uint32_t kXcdSeg = ma::kXcdSegment;     // chunks per XCD segment of the search launch (MA_XCD_SEG)
constexpr int kMaxGroups = ma::kMaxSchedGroups;   // scheduling groups (of 2^gshift segments) ranked by cost

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
using ma::StateSlot;                // list_planner.h: the pinned slot every solve of a planned loop reports into
using ma::slot_check;
using ma::slot_flags;
using ma::slot_ready;

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

// The same two for 64-bit INTEGERS (the exact sums of row a5, host_math.h "Exact sums"): an add of a DPP-shifted copy
// and the wavefront total in lane 63.  Lanes without a source read 0 (bound_ctrl), the identity of the sum.
// Two instructions: the low halves with carry out, the high halves with carry in, each taking its first operand through
// the DPP network (v_add_co_u32_dpp / v_addc_co_u32_dpp: VOP2, so the shifted copy costs no move of its own — the fp64
// form above is three).  Spelled in assembly because the compiler turns the C form into two moves, two zero-extensions
// and two 64-bit adds; the s_nop covers the two wait states a DPP read of a just-written VGPR needs (the scheduler does
// not look inside an asm block).
#define MA_DPP_ADD_U64(NAME, CTRL_TEXT)                                                                                  \
  __device__ __forceinline__ unsigned long long NAME(const unsigned long long v) {                                       \
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);                                                            \
    uint32_t rlo, rhi;                                                                                                    \
    asm volatile("s_nop 1\n\t"                                                                                            \
                 "v_add_co_u32_dpp %0, vcc, %2, %2 " CTRL_TEXT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"              \
                 "v_addc_co_u32_dpp %1, vcc, %3, %3, vcc " CTRL_TEXT " row_mask:0xf bank_mask:0xf bound_ctrl:1"            \
                 : "=&v"(rlo), "=&v"(rhi)                                                                                 \
                 : "v"(lo), "v"(hi)                                                                                       \
                 : "vcc");                                                                                                \
    return ((unsigned long long)rhi << 32) | rlo;                                                                         \
  }
// ... and without the wait states, for callers that keep the producer of `v` at least two instructions away (reduce_chunk's
// two passes over 17 values)
#define MA_DPP_ADD_U64_FAR(NAME, CTRL_TEXT)                                                                              \
  __device__ __forceinline__ unsigned long long NAME(const unsigned long long v) {                                       \
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);      /* in place: an instruction reads every lane's operands before it writes */ \
    asm volatile("v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL_TEXT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"              \
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL_TEXT " row_mask:0xf bank_mask:0xf bound_ctrl:1"            \
                 : "+v"(lo), "+v"(hi)                                                                                     \
                 :                                                                                                        \
                 : "vcc");                                                                                                \
    return ((unsigned long long)hi << 32) | lo;                                                                           \
  }
MA_DPP_ADD_U64_FAR(dpp_add_u64_far_shr1, "row_shr:1")
MA_DPP_ADD_U64_FAR(dpp_add_u64_far_shr2, "row_shr:2")
#undef MA_DPP_ADD_U64_FAR
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_add_u64_far(const unsigned long long v) {
  static_assert(CTRL == 0x111 || CTRL == 0x112, "DPP control");
  if constexpr (CTRL == 0x111) return dpp_add_u64_far_shr1(v);
  else return dpp_add_u64_far_shr2(v);
}
MA_DPP_ADD_U64(dpp_add_u64_shr1, "row_shr:1")
MA_DPP_ADD_U64(dpp_add_u64_shr2, "row_shr:2")
MA_DPP_ADD_U64(dpp_add_u64_shr4, "row_shr:4")
MA_DPP_ADD_U64(dpp_add_u64_shr8, "row_shr:8")
MA_DPP_ADD_U64(dpp_add_u64_bcast15, "row_bcast:15")
MA_DPP_ADD_U64(dpp_add_u64_bcast31, "row_bcast:31")
#undef MA_DPP_ADD_U64
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_add_u64(const unsigned long long v) {
  static_assert(CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118 || CTRL == 0x142 || CTRL == 0x143, "DPP control");
  if constexpr (CTRL == 0x111) return dpp_add_u64_shr1(v);
  else if constexpr (CTRL == 0x112) return dpp_add_u64_shr2(v);
  else if constexpr (CTRL == 0x114) return dpp_add_u64_shr4(v);
  else if constexpr (CTRL == 0x118) return dpp_add_u64_shr8(v);
  else if constexpr (CTRL == 0x142) return dpp_add_u64_bcast15(v);
  else return dpp_add_u64_bcast31(v);
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  v = dpp_add_u64<0x111>(v);  // row_shr:1
  v = dpp_add_u64<0x112>(v);  // row_shr:2
  v = dpp_add_u64<0x114>(v);  // row_shr:4
  v = dpp_add_u64<0x118>(v);  // row_shr:8
  v = dpp_add_u64<0x142>(v);  // row_bcast:15
  v = dpp_add_u64<0x143>(v);  // row_bcast:31 -> lane 63 = total
  return v;
}
using ma::SumScale;
using ma::kSumMagic;
using ma::kSumMagicBits;
// 2^e as a double from its exponent (|e| < 1000): two scalar operations, no table
__device__ __forceinline__ double pow2_f64(const int e) { return __hiloint2double((1023 + e) << 20, 0); }


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

// HALO grids (host_math.h "Robust grid box"), centred frame: per axis, the coordinate nearest to the grid's box among the
// gold points stored in its boundary cells that lie beyond the box ALONG THAT AXIS — lo[a] the largest below it (-inf:
// none), hi[a] the smallest above it (+inf: none).  Along axis a the content of a boundary cell is its own interval plus
// a part of (-inf, lo[a]] or [hi[a], +inf), with nothing in between.
struct HaloBox { float lo[3], hi[3]; };

struct Pinhole {
  double fx_inv, fy_inv, cx, cy, inv_scale;
  double T[12];  // camera->world rows 0..2, translation already minus the centring origin
};
constexpr uint32_t kHoleBits = 0x7FC00000u;      // quiet NaN: x of a hole of the in-place unprojection

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

extern thread_local std::string g_create_error;      // (defined in host_ctx.hip)
// the message of the calling THREAD's last ma_ctx::fail: ma_set_pair runs two threads on one context, and which of two
// failures a context reports must not depend on which thread wrote its message last
extern thread_local std::string t_last_fail;


}  // namespace mai
using namespace mai;

struct ma_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned long long* host_clk0 = nullptr;     // pinned: k_begin_run's clock stamp (the start of a loop's device time)
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
  HaloBox halo_box{{0, 0, 0}, {0, 0, 0}};     // with grid_halo (kernels_search.hip HALO)
  bool grid_halo = false;                      // the grid covers the bulk of the gold map only (host_math.h "Robust grid box"): strays sit in its boundary cells
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
  char* batch_stage = nullptr;   // pinned staging of ma_map_add_depth_batch: the frames' poses going in, their counts coming back
  size_t batch_stage_cap = 0;
  bool unproj_pending = false;   // a k_unproject launch whose count of valid pixels has not been added up yet (finish_unproject)
  bool unproj_is_source = false; //   ... and it wrote the source cloud: the count settles n_pts / src_holes / MA_ERR_EMPTY
  uint32_t unproj_blocks = 0, unproj_total = 0;

  // loop state
  DevBuf state, partials, trace, acc19, ticket, wg_cost, seg_order, nn_idx, nn_d2, nn_tmp, prev;
  // candidate lists of the loop (k_nn_accum): 8 gold positions + one radius per source slot, and the control block
  DevBuf lists, listctl;      // lists: the five planes of ListPlanes, back to back
  float src_rs = 0.f;                          // largest distance of a corner of the source cloud's box from the centring origin; 0: unknown, no lists
  float src_lo[3] = {0, 0, 0}, src_hi[3] = {0, 0, 0};   // that box (centred frame)
  bool src_bulk_box = false;                   // src_lo / src_hi are the box of the cloud's bulk (stray points lie outside it)
  bool src_box_pending = false;                // source from a depth frame: the box is computed by the first run that can use it
  // exact sums (host_math.h): what bounds a correspondence's terms
  double src_wmax = 1.0;                       // largest weight of the source cloud (1 without weights)
  double src_amax = 0.0;                       // largest |coordinate| of the source cloud (centred frame; with the box): what its radius for the sum
                                               //   bounds derives from — a MAXIMUM, so that the largest over the shards of a cloud is the whole cloud's, exactly
  double gold_qb = 0.0;                        // largest |coordinate| of the gold map's grid box, centred frame
  double run_t0 = 0.0;                         // |t| of the transform the current run / stepping sequence started from (centred frame)
  bool sum_bounds_set = false;                 // ma_sum_bounds_set: the bounds of the WHOLE cloud of a sharded run ...
  ma::SumBounds sum_bounds{0.0, 1.0, 0.0};     // ... (n_total, wmax, rs); until then this context's own
  bool lists_on = false;                       // set by the loop (icp_loop) around its launches
  bool cfg_lists = true;                       // ma_list_config: lists allowed at all; skin = cell edge / cfg_skin_div; a write must
  float cfg_skin_div = ma::kSkinDivDefault;    //   be followed by >= cfg_min_uses predicted uses
  int cfg_min_uses = ma::kMinUsesDefault;
  double cfg_build_frac = ma::kBuildFracDefault, cfg_rho_min = ma::kRhoMinDefault;      // the planner's prediction constants (environment at ma_create: sweeps)
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
  // peer-to-peer exchange of the sums (ma_p2p_export / ma_p2p_init; comm.hip): this rank's rows, every rank's rows as
  // mapped here, and the number of iterations the ranks have solved together so far
  unsigned long long* xchg = nullptr;          // kXWords words of this device's memory (fine-grained where the runtime allows)
  bool xchg_finegrained = false;
  unsigned long long* xchg_peer[ma::kXMaxRanks] = {};
  bool xchg_peer_ipc[ma::kXMaxRanks] = {};     // opened with hipIpcOpenMemHandle (to be closed)
  int xchg_ranks = 0, xchg_rank = 0;           // 0 ranks: no exchange set up
  unsigned long long xchg_seq = 0;
  unsigned long long xchg_agree_gen = 0;       // bounds agreements done over the exchange (the tag of the next one)
  bool xchg_broken = false;                    // a run ended in a communication failure: the ranks' counters may differ; set up again
  int comm_transport = 0;                      // ma_comm_config: 0 = RCCL all-reduce, 1 = the peer-to-peer exchange
  int xchg_timeout_ms = 2000;                  // how long the solving wave waits for its peers' rows (MAPALIGN_P2P_TIMEOUT_MS at ma_create)
  bool xchg_fresh = false;                     // rows zeroed by ma_p2p_export and not yet used by a ma_p2p_init (one export per init:
                                               // rows of an earlier set-up carry tags a new count from 0 would accept)
  bool xchg_force_generic = false;             // MAPALIGN_P2P_GENERIC=1 at ma_create: always the guarded up-to-15-peers kernel (tests)
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

namespace mai {

// Mailbox: the few numbers a staging step hands back to the host (bounds, occupancy counts, the count of bad
// weights) are WRITTEN BY THE KERNELS INTO PINNED HOST MEMORY and read after the stream synchronisation the step
// has anyway.  Round 2 copied them device-to-host into pageable vectors; next to the other thread's 12 MB pageable
// upload such a copy waited for that upload to finish (rocprofv3 memory-copy trace: the gold map's 32 KB of bounds
// arrived 230 us after the kernel that produced them, right behind the scan's upload).
constexpr size_t kMailBytes = 128 * 1024;      // per lane: [0, 64K) bounds, [64K, 96K) probe counts, [96K, ...) single words
constexpr size_t kMailProbe = 64 * 1024, kMailWord = 96 * 1024;
struct Lane {
  hipStream_t stream;
  DevBuf *keys_a, *keys_b, *vals_a, *vals_b, *cub_tmp, *misc;
  char* mail;
};
inline Lane main_lane(ma_ctx* ctx) { return {ctx->stream, &ctx->keys_a, &ctx->keys_b, &ctx->vals_a, &ctx->vals_b, &ctx->cub_tmp, &ctx->misc, ctx->mailbox}; }
inline Lane aux_lane(ma_ctx* ctx) { return {ctx->aux_stream, &ctx->akeys_a, &ctx->akeys_b, &ctx->avals_a, &ctx->avals_b, &ctx->acub_tmp, &ctx->amisc, ctx->mailbox + kMailBytes}; }
inline bool lists_enabled(const ma_ctx* ctx) { return ctx->cfg_lists; }
inline float list_skin_div(const ma_ctx* ctx) { return ctx->cfg_skin_div; }
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

// ---- what the translation units offer each other -------------------------------------------------------------------
// kernels_build.hip: staging of clouds on the device
// bounding box (and finiteness check) of a device-resident n x 3 (stride 3) or float4 (stride 4) cloud
int cloud_bounds(ma_ctx* ctx, const float* d_xyz, uint32_t n, float lo[3], float hi[3], int stride = 3, bool holes_ok = false,
                 const Lane* lane = nullptr, const float* within_lo = nullptr, const float* within_hi = nullptr,
                 double* n_outside = nullptr);
// stable radix sort of (key, index) pairs; tmp == nullptr asks for the temporary bytes
hipError_t sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream);
hipError_t sort_pairs_u64(void* tmp, size_t& tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, int bits, hipStream_t stream);
// exclusive prefix sum of n 32-bit words (out may be in); tmp: scan_tmp_bytes(n) bytes
size_t scan_tmp_bytes(uint32_t n);
hipError_t scan_exclusive_u32(void* tmp, const uint32_t* in, uint32_t* out, uint32_t n, hipStream_t stream);
int ensure_levels_for(ma_ctx* ctx, double max_corr_dist);
// Builds the gold map (level-0 grid, distance table) from the n x 3 fp32 world-frame points already resident in
// ctx->raw; ev0 must have been recorded by the caller.
int build_gold_from_raw(ma_ctx* ctx, uint32_t n, float cell);
// d_raw (default ctx->raw) holds the n x 3 fp32 source points (relative to the host-side shift, if any)
int build_source_from_raw(ma_ctx* ctx, uint32_t n, const float* weights, const float* d_raw = nullptr, float gold_cell = 0.f,
                          const Lane* lane = nullptr);
int stage_f64(ma_ctx* ctx, const double* xyz, uint32_t n, bool choose_shift, double shift[3]);
hipError_t launch_f64_to_raw(const double* xyz, uint64_t n3, const double shift[3], float* raw, hipStream_t stream);

// kernels_search.hip: one launch of k_nn_accum.  mode 0: accumulate (the loop), 1: per-point (idx, d2) only (ma_nn);
// `scheduled`: inside the loop (costs recorded, longest-first order from iteration 1 on); list_kind 0 search, 1 search
// and write the candidate lists, 2 read them
int launch_nn(ma_ctx* ctx, int mode, bool count, float best2_init, uint32_t* nn_idx, float* nn_d2, const SumScale& sc,
              bool scheduled = false, int list_kind = 0, int warm_hint = -1);

// kernels_solve.hip
int launch_begin_run(ma_ctx* ctx, const DevState& s, const ListCtl& c);
int launch_reduce(ma_ctx* ctx, bool with_rank = false);
int launch_reduce_solve(ma_ctx* ctx, double tol, const SumScale& sc, StateSlot* slot = nullptr);
int launch_solve(ma_ctx* ctx, double tol, const SumScale& sc, bool in_loop = false, StateSlot* slot = nullptr);
// the sharded iteration's second launch with the exchange inside: sum, write to every peer, wait, add, solve
int launch_reduce_exchange_solve(ma_ctx* ctx, double tol, const SumScale& sc, StateSlot* slot);

// kernels_depth.hip
int finish_unproject(ma_ctx* ctx);
int source_map(ma_ctx* ctx, const uint32_t** map);
// out[perm[pos]] = in[pos] for 4-byte items (perm null: identity; kNoMatch: a hole, skipped)
int launch_unpermute4(ma_ctx* ctx, const void* in, const uint32_t* perm, uint32_t n, void* out);

// host_loop.hip
int check_params(ma_ctx* ctx, const ma_params* p);
// the box of a source cloud made from a depth frame, computed when first needed (candidate lists, ungated sum bounds)
int ensure_source_box(ma_ctx* ctx);
// the fixed-point scale of the 17 sums for a run gated at `gate` on the clouds and bounds in force (host_math.h)
int run_sum_scale(ma_ctx* ctx, double gate, SumScale* out);
// a new source cloud (or gold map): whatever was agreed for the old one no longer holds
inline void source_replaced(ma_ctx* ctx) { ctx->sum_bounds_set = false; ctx->src_wmax = 1.0; ctx->src_amax = 0.0; }
// radius of the source cloud for the sum bounds: |p| <= sqrt(3) max|coordinate| (a little padding for the fp32 rounding of the
// stored coordinates; monotone in src_amax, so the maximum over shards stays the whole cloud's)
inline double sum_rs(const ma_ctx* ctx) { return 1.7320508075688772 * ctx->src_amax * (1.0 + 1e-6) + 1e-6; }
int need_ready(ma_ctx* ctx);
int ensure_loop_buffers(ma_ctx* ctx, int trace_rows);
int upload_state(ma_ctx* ctx, const double* T);
int fetch_state(ma_ctx* ctx, DevState* s);
void fill_stats(const DevState& s, double gpu_ms, ma_stats* st);
int icp_loop(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16], ma_stats* stats, bool sharded);

// comm.hip
void comm_destroy(ma_ctx* ctx);
// the loop's collective: all-reduce (sum) of the MA_ACC_LEN sums in ctx->acc19, in place, on the ctx stream
int comm_allreduce_sums(ma_ctx* ctx);
// every rank's (n, wmax, rs) -> the whole cloud's, installed as this context's sum bounds (one small collective per source cloud)
int comm_agree_sum_bounds(ma_ctx* ctx);
void p2p_release(ma_ctx* ctx);
// does a sharded run on this context exchange peer to peer (else: RCCL)?
inline bool use_p2p(const ma_ctx* ctx) { return ctx->xchg_ranks > 0 && (ctx->comm_transport == 1 || !ctx->comm); }

// host_ctx.hip
void list_config_from_env(ma_ctx* ctx);
#if MA_PAIR_TIMES
void pt(const char* what);
void pt_dump();
#define MA_PT(x) ::mai::pt(x)
#else
#define MA_PT(x) ((void)0)
#endif

}  // namespace mai
