// kernels_search.hip — rows a4 + a5: k_nn_accum, the fused nearest-neighbour search / candidate-list read and the
// per-workgroup sums, with everything only it uses (the per-lane multi-level search, the wavefront-cooperative work
// queues, the XCD-aware longest-first chunk order), and launch_nn, the one host function that launches it.
#include "ma_internal.h"

namespace {

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

// HALO (host_math.h "Robust grid box"): the grid covers only the bulk of the gold map, and the points beyond its box are
// stored in its BOUNDARY cells (cell_of clamps its indices) with their true coordinates.  Along an axis the content of a
// boundary cell is then its own interval plus a part of the half-line that starts at the nearest such point's coordinate
// (HaloBox: w_lo / w_hi), with nothing in between, and every test of cell geometry knows it:
//   * the gap between a query and a row (or slab) of boundary index is the smaller of the gap to the row's own interval
//     and the gap to that half-line — for a query near the scene the former, exactly what an ordinary grid computes
//     (stray points are usually far), for a stray query out there zero;
//   * along x a sphere that ends beyond the row's end reaches the end cell only if it reaches that half-line;
//   * ring bounds and the distance-to-occupied table work on cell indices, which are clamped for query and content
//     alike — clamping both ends of a distance to an interval never lengthens it — and hold as they are.
// Distances are formed from the true coordinates, so the search stays exact for every query, inside the box or not.
// A compile-time switch: for an ordinary map (every point inside its cell) the kernels are instruction for instruction
// what they were.
template <bool HALO>
__device__ __forceinline__ float row_gap(const float below, const float above, const int idx, const int n, const float q,
                                         const float w_lo, const float w_hi) {
  // below = (row's lower edge - slack) - q, above = q - (row's upper edge + slack): the gap is the positive one, if any
  float t = fmaxf(0.f, fmaxf(below, above));
  if (HALO) {
    if (idx == n - 1) t = fminf(t, fmaxf(0.f, w_hi - q));
    if (idx == 0) t = fminf(t, fmaxf(0.f, q - w_lo));
  }
  return t;
}
template <bool HALO>
__device__ __forceinline__ void x_cells(const float px, const float rx, const float ox, const float inv_c, const int nx, int& x0, int& x1,
                                        const float w_lo, const float w_hi) {
  const float f0 = floorf((px - rx - ox) * inv_c), f1 = floorf((px + rx - ox) * inv_c);
  x0 = (int)fmaxf(f0, 0.f);
  x1 = (int)fminf(f1, (float)(nx - 1));
  if (HALO) {
    // a sphere that lies beyond the row's end still reaches what the end cell holds out there, if it reaches the nearest of it
    if (f0 > (float)(nx - 1) && px + rx >= w_hi) x0 = x1 = nx - 1;
    if (f1 < 0.f && px - rx <= w_lo) x0 = x1 = 0;
  }
}

// One pass at one level: the 3 x 3 rows around the query's row (rings 0 and 1), each over
// the x-run the current best sphere reaches.  Exact for every gold point within
// (c - delta) of the query.
template <bool COUNT, bool HALO>
__device__ __forceinline__ void level_pass(const float px, const float py, const float pz,
                                           const LevelView& V, const LevelSet& S, const HaloBox& hb, float& best2,
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
    const float dzq = row_gap<HALO>(zlo - S.delta - pz, pz - (zlo + V.c + S.delta), z, V.nz, pz, hb.lo[2], hb.hi[2]);
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
        const float dyq = row_gap<HALO>(ylo - S.delta - py, py - (ylo + V.c + S.delta), y, V.ny, py, hb.lo[1], hb.hi[1]);
        const float d2yz = dyq * dyq + dzq * dzq;
        if (d2yz <= best2) {
          const float rx = fast_sqrt(best2 - d2yz) + S.delta;  // +inf while ungated and unmatched
          int x0, x1;
          x_cells<HALO>(px, rx, S.ox, V.inv_c, V.nx, x0, x1, hb.lo[0], hb.hi[0]);
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
template <bool COUNT, bool HALO>
__device__ __forceinline__ void ring_walk(const float px, const float py, const float pz,
                                          const LevelView& V, const LevelSet& S, const HaloBox& hb, const int r_start,
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
              const float dyq = row_gap<HALO>(ylo - S.delta - py, py - (ylo + V.c + S.delta), y, V.ny, py, hb.lo[1], hb.hi[1]);
              const float dzq = row_gap<HALO>(zlo - S.delta - pz, pz - (zlo + V.c + S.delta), z, V.nz, pz, hb.lo[2], hb.hi[2]);
              const float d2yz = dyq * dyq + dzq * dzq;
              if (d2yz <= best2) {
                const float rx = fast_sqrt(best2 - d2yz) + S.delta;
                int x0, x1;
                x_cells<HALO>(px, rx, S.ox, V.inv_c, V.nx, x0, x1, hb.lo[0], hb.hi[0]);
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
template <bool COUNT, bool HALO>
__device__ __forceinline__ void nn_find(const float px, const float py, const float pz,
                                        const LevelSet& S, const HaloBox& hb, float& best2, uint32_t& bestj,
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
      level_pass<COUNT, HALO>(px, py, pz, V, S, hb, best2, bestj, nrows, ncand);
      const float lim = V.c - S.delta, wide = kWalkCells * V.c;
      if (best2 <= lim * lim) {
        active = false;  // the pass covered the whole best sphere: exact
      } else if (l == kLevels - 1 || best2 <= wide * wide) {
        // rings 0 and 1 are done (the pass scanned the full x-run of each of its rows)
        ring_walk<COUNT, HALO>(px, py, pz, V, S, hb, 2, best2, bestj, nrows, ncand);
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


template <bool COUNT, bool BUILD, bool HALO>
__device__ __forceinline__ void wq_flush_rows(WaveLds& W, const int lane, const LevelSet& S, const HaloBox& hb, uint32_t& nrow,
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
          int x0, x1;
          x_cells<HALO>(p.x, rx, S.ox, inv_c, nx, x0, x1, hb.lo[0], hb.hi[0]);
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
template <bool COUNT, bool BUILD, bool HALO>
__device__ __forceinline__ void wave_search(WaveLds& W, const int lane, const bool has_query, const float px,
                                            const float py, const float pz, const LevelSet& S, const HaloBox& hb, float& best2,
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
    if (nrow + cnt > (uint32_t)kRowCap) wq_flush_rows<COUNT, BUILD, HALO>(W, lane, S, hb, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
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
  auto gap2 = [&](const int d, const float f, const int ci, const int n, const int axis /* 1 = y, 2 = z */) -> float {
    const float lo = ((float)d * c - S.delta) - f, hi = f - ((float)(d + 1) * c + S.delta);
    const bool is_y = axis == 1;
    const float t = row_gap<HALO>(lo, hi, ci + d, n, is_y ? py : pz, is_y ? hb.lo[1] : hb.lo[2], is_y ? hb.hi[1] : hb.hi[2]);
    return (unsigned)(ci + d) < (unsigned)n ? t * t : NAN;     // outside the grid: no comparison with it holds (<= included)
  };
  {
    const float gy[3] = {gap2(0, fy, cy, ny, 1), gap2(-1, fy, cy, ny, 1), gap2(1, fy, cy, ny, 1)};
    const float gz[3] = {gap2(0, fz, cz, nz, 2), gap2(-1, fz, cz, nz, 2), gap2(1, fz, cz, nz, 2)};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int dy = (k % 3 == 0) ? 0 : (k % 3 == 1 ? -1 : 1), dz = (k / 3 == 0) ? 0 : (k / 3 == 1 ? -1 : 1);
      const float d2yz = gy[k % 3] + gz[k / 3];
      push_row(finite && d2yz <= bi, cy + dy, cz + dz, d2yz);
    }
  }
  wq_flush_rows<COUNT, BUILD, HALO>(W, lane, S, hb, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
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
      const float gfix = gap2(dfix, ffix, cfix, nfix, zfixed ? 2 : 1);
      for (int o = 0; o < 2 * r; ++o) {
        const int dvar = side < 2 ? o - r : r - o;
        const float d2yz = gfix + gap2(dvar, fvar, cvar, nvar, zfixed ? 1 : 2);
        const int dy = zfixed ? dvar : dfix, dz = zfixed ? dfix : dvar;
        push_row(more && d2yz <= bi, cy + dy, cz + dz, d2yz);
      }
    }
    wq_flush_rows<COUNT, BUILD, HALO>(W, lane, S, hb, nrow, nchunk, nrows_cnt, ncand, skin, lp, slot0);
    wq_flush_chunks<COUNT, BUILD>(W, lane, gold0, nchunk, ncand, skin, lp, slot0);
    const unsigned long long key = W.best[lane];
    best2 = held_d2(key);
    bestj = (uint32_t)key;
    bi = reach(best2);
  }
  if (fallback) nn_find<COUNT, HALO>(px, py, pz, S, hb, best2, bestj, nrows_cnt, ncand);
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
    // The query's own rounding allowance: its fp32 image is within an ulp of where the fp64 transform puts it, now and in
    // the launches that read the list — 1.2e-7 |p| each.  ListCtl::slack covers that for a point inside the cloud's box
    // (or the box of its bulk, build_source_from_raw); a stray point a kilometre away pays for itself here.
    L = fmaxf(L - 1.5e-6f * (fabsf(px) + fabsf(py) + fabsf(pz)), 0.f);
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
// a5: the 17 terms of one correspondence (w, w p', w q, w p' q^T row-major, w |p' - q|^2), each rounded ONCE to a multiple
// of its accumulator's quantum 2^-e (host_math.h "Exact sums"): fma(x, 2^e, 1.5 * 2^52) leaves rne(x 2^e) in the low
// mantissa bits of the result, and the sums from here on add those bit patterns as 64-bit integers — 1.5 * 2^52 included,
// once per lane, taken out again when the workgroup writes its partial.  Every path (search, list launch, any shard of any
// partition) forms a correspondence's terms with exactly these operations, so the integer sums are the same whatever the
// partition: this is what makes a sharded run return the single GPU's bits.
__device__ __forceinline__ void pair_terms(unsigned long long (&acc)[kNAcc], const double w, const double Px, const double Py,
                                           const double Pz, const float3 q, const SumScale& sc) {
  const double qx = q.x, qy = q.y, qz = q.z;
  const double dx = Px - qx, dy = Py - qy, dz = Pz - qz;
  // w 2^e is exact (a power of two), so each term is one rounding of an exact product against the magic constant
  const double wW = w * pow2_f64(sc.e[ma::kSumW]), wP = w * pow2_f64(sc.e[ma::kSumP]), wQ = w * pow2_f64(sc.e[ma::kSumQ]);
  const double wS = w * pow2_f64(sc.e[ma::kSumS]), wE = w * pow2_f64(sc.e[ma::kSumE]);
  const double wPx = wS * Px, wPy = wS * Py, wPz = wS * Pz;
  auto q64 = [](const double a, const double b) -> unsigned long long {
    return (unsigned long long)__double_as_longlong(__fma_rn(a, b, kSumMagic));
  };
  acc[0] = q64(wW, 1.0);
  acc[1] = q64(wP, Px); acc[2] = q64(wP, Py); acc[3] = q64(wP, Pz);
  acc[4] = q64(wQ, qx); acc[5] = q64(wQ, qy); acc[6] = q64(wQ, qz);
  acc[7] = q64(wPx, qx);  acc[8] = q64(wPx, qy);  acc[9] = q64(wPx, qz);
  acc[10] = q64(wPy, qx); acc[11] = q64(wPy, qy); acc[12] = q64(wPy, qz);
  acc[13] = q64(wPz, qx); acc[14] = q64(wPz, qy); acc[15] = q64(wPz, qz);
  acc[16] = q64(wE, __fma_rn(dz, dz, __fma_rn(dy, dy, dx * dx)));
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
__device__ __forceinline__ void reduce_chunk(const unsigned long long (&acc)[kNAcc], WV* __restrict__ sh_w,
                                             unsigned long long (*sh_part)[kRedParts], long long* __restrict__ partials,
                                             const uint32_t G, const uint32_t chunk, const int lane, const int wv) {
  constexpr int kNW = kNNBlock / kWave;
  constexpr int kLive = COUNT ? kNAcc : ma::kAccLen;
  constexpr int kVals = kNW * 16, kPer = kVals / kRedParts;
  // two passes over the accumulators instead of two steps per accumulator: the DPP read of a value then comes >= 16
  // instruction pairs after the add that wrote it, so the adds need no wait states of their own (dpp_add_u64_far: the
  // general form without its s_nop; tests/test_isa_hazards.py checks the distance in the built code).
  // MA_REDUCE_ONEPASS=1 keeps the round's first form (both steps per accumulator, s_nop each) for tools/variants.py.
  unsigned long long v[kLive];
#if MA_REDUCE_ONEPASS
#pragma unroll
  for (int k = 0; k < kLive; ++k) v[k] = dpp_add_u64<0x112>(dpp_add_u64<0x111>(acc[k]));
#else
#pragma unroll
  for (int k = 0; k < kLive; ++k) v[k] = dpp_add_u64_far<0x111>(acc[k]);
#pragma unroll
  for (int k = 0; k < kLive; ++k) v[k] = dpp_add_u64_far<0x112>(v[k]);
#endif
#pragma unroll
  for (int k = 0; k < kLive; ++k)
    if ((lane & 3) == 3) sh_w[wv].v[k][lane >> 2] = v[k];
  __syncthreads();
  if (threadIdx.x < kLive * kRedParts) {
    const int k = threadIdx.x / kRedParts, part = threadIdx.x % kRedParts;
    unsigned long long v = 0ull;
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
    unsigned long long v = 0ull;
    if (k < kLive) {
#pragma unroll
      for (int t = 0; t < kRedParts; ++t) v += sh_part[k][t];
      // every lane of the workgroup carried the magic constant once (matched or not): out it goes, modulo 2^64
      if (k < ma::kAccLen) v -= (unsigned long long)kNNBlock * kSumMagicBits;
    }
    partials[(size_t)k * G + chunk] = (long long)v;
  }
}

#if MA_WGTIMES   // debug variant only (tools/wgtimeline.py): start/end clock and placement of every workgroup
__device__ unsigned long long g_wgt[64][4096][2];
__device__ uint32_t g_wghw[64][4096][2];
__device__ uint32_t g_wgph[64][4096][4];   // wave 0: ticks from the start to: search begins, 3x3 block done, search done, sums formed
__device__ uint32_t g_wvt[16][4096][4];    // iterations 0-15, EVERY wavefront: ticks from the workgroup's start to the end of its search (tools/wavetimeline.py)
#endif

// Waves per SIMD the register allocator must make room for: 8 (<= 64 VGPRs) for the search and the list launch — every
// build that let the search cross 64 VGPRs lost 5-10 % — and whatever it takes for the list-writing launch (2 of an
// alignment's 50 launches), the work-counting instantiations and ma_nn's (forced to 64 VGPRs they spill).
// MA_NN_WAVES overrides all of them (A/B).
#ifdef MA_NN_WAVES
#define MA_NN_ATTR __attribute__((amdgpu_waves_per_eu(MA_NN_WAVES, MA_NN_WAVES)))
#else
// (HALO instantiations, A/B knob MA_HALO_MIN_WAVES: forced to 8 waves the boundary-cell tests spill 8-9 dwords)
#ifndef MA_HALO_MIN_WAVES
#define MA_HALO_MIN_WAVES 7
#endif
#define MA_NN_ATTR __attribute__((amdgpu_waves_per_eu((MODE == 0 && LISTS != 1 && !COUNT) ? (HALO ? MA_HALO_MIN_WAVES : 8) : 4, 8)))
#endif
// LISTS (compile time, so that the plain search launch keeps its 63 VGPRs — a run-time switch between the three bodies
// cost 6 VGPRs, a wavefront per SIMD and 10 % of every search launch): 0 search; 1 search and write the candidate lists;
// 2 read the lists.  The host picks the kernel per iteration from the step sizes the solve reports (ma_icp).
template <bool COUNT, int MODE, int LISTS, bool HALO>
// (Argument order: the first 16 dwords are preloaded into scalar registers at dispatch — csrc/Makefile — so the state,
// the source cloud, the remembered matches and the schedule, which the first instructions of a workgroup need, come
// first; the level set, 200 bytes by value, is fetched while the first loads are in flight.)
__global__ __launch_bounds__(kNNBlock) MA_NN_ATTR void k_nn_accum(
    const DevState* __restrict__ st, const float4* __restrict__ src, const uint32_t n, const int warm_hint,
    uint32_t* __restrict__ prev, const Sched sched, long long* __restrict__ partials, const float best2_init,
    const LevelSet S, const uint32_t* __restrict__ perm0, uint32_t* __restrict__ nn_idx,
    float* __restrict__ nn_d2, const ListCtl* __restrict__ lc, const ListPlanes lp, const SumScale sc, const HaloBox hb) {
  // (hb, the last argument: only the HALO instantiations read it — an ordinary map's kernels never fetch those bytes)
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
    unsigned long long v[kNAcc][16];
  };
  struct WaveSums {            // a list launch has no work queues
    unsigned long long v[kNAcc][16];
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
        nn_find<COUNT, HALO>(px, py, pz, S, hb, best2, bestj, nrows, ncand);
        if (bestj != kNoMatch) { const float3 q = ld_xyz(S.gold[0], bestj); wqx = q.x; wqy = q.y; wqz = q.z; }
      }
    }
  } else if constexpr (lmode == 1) {
    wave_search<COUNT, true, HALO>(sh_w[wv].q, lane, has, px, py, pz, S, hb, best2, bestj, nrows, ncand, ndismissed, lc->skin,
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
    wave_search<COUNT, false, HALO>(sh_w[wv].q, lane, has, px, py, pz, S, hb, best2, bestj, nrows, ncand, ndismissed);
  }
#if MA_WGTIMES
  unsigned long long dbg_t2 = 0;
  if constexpr (lmode != 2) dbg_t2 = sh_w[wv].q.dbg_round0;
  const unsigned long long dbg_t3 = wall_clock64();
  unsigned long long dbg_t4 = 0;
  if (lane == 0 && MODE == 0 && st->iter < 16 && blockIdx.x < 4096 && wv < 4) g_wvt[st->iter][blockIdx.x][wv] = (uint32_t)(dbg_t3 - wg_start);
#endif

  // the 17 terms as integer bit patterns (pair_terms); a lane without a correspondence carries the bare magic constant
  unsigned long long acc[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) acc[k] = k < ma::kAccLen ? kSumMagicBits : 0ull;
  // The slot index once more, formed anew from the chunk (a scalar, made opaque) and the thread id: neither the index
  // nor the point and its image have to live in vector registers through the search (64 VGPRs are all there are at 8
  // waves per SIMD; with the index carried the allocator spilled it).
  uint32_t chunk_again = chunk;
  asm volatile("" : "+s"(chunk_again));
  const uint32_t i2 = lmode == 2 ? i : chunk_again * kNNBlock + threadIdx.x;      // (a list launch has no search in between)
  if (has) {
#if MA_DEBUG_FETCH      // debug build (tools/tie_debug.py): list launches leave their matches behind too
    if (prev) prev[i2] = bestj;
#else
    if (prev && lmode != 2) prev[i2] = bestj;
#endif
    if (MODE == 0 && lmode != 2) image(i2);       // (a list launch's point stays in registers)
    if (MODE == 1) {
      nn_idx[i2] = bestj != kNoMatch ? perm0[bestj] : kNoMatch;
      nn_d2[i2] = bestj != kNoMatch ? best2 : INFINITY;
    }
    if (MODE == 0 && bestj != kNoMatch) {
      float3 q;
      if constexpr (lmode == 2) q = make_float3(wqx, wqy, wqz);     // carried with the minimum: the very bits of the gold array
      else q = ld_xyz(S.gold[0], bestj);           // the winner's coordinates (level 0 holds every point)
      pair_terms(acc, (double)s.w, Px, Py, Pz, q, sc);
    }
    if (COUNT) { acc[17] = nrows; acc[18] = ncand; }
  }
  if (MODE == 1) return;  // ma_nn: per-point outputs only
#if MA_WGTIMES
  asm volatile("" :: "v"(acc[16]), "v"(acc[0]));
  dbg_t4 = wall_clock64();
#endif

  // workgroup reduction of the live accumulators -> one partial per accumulator and chunk (reduce_chunk)
  __shared__ unsigned long long sh_part[kNAcc][kRedParts];
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
}  // namespace

namespace mai {

int launch_nn(ma_ctx* ctx, const int mode, bool count, float best2_init, uint32_t* nn_idx, float* nn_d2, const SumScale& sc,
              bool scheduled, int list_kind, int warm_hint) {
  uint32_t gshift = 0;
  const uint32_t G = grid_for(ctx->n, &gshift);
  ctx->last_G = (int)G;
  ctx->last_gshift = gshift;
  Sched sched{nullptr, nullptr, gshift};
  if (scheduled) { sched.wg_cost = ctx->wg_cost.as<uint32_t>(); sched.order = ctx->seg_order.as<uint32_t>(); }
  const bool with_lists = mode == 0 && ctx->lists_on && list_kind != 0;
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
                       warm_hint, mode == 0 ? ctx->prev.as<uint32_t>() : (uint32_t*)nullptr, sched,
                       ctx->partials.as<long long>(), best2_init, ctx->levels, ctx->perm0.as<uint32_t>(), nn_idx, nn_d2,
                       with_lists ? ctx->listctl.as<ListCtl>() : (const ListCtl*)nullptr, lp, sc, ctx->halo_box);
  };
  // (HALO: the instantiations for a grid that covers the bulk of the gold map only — strays in its boundary cells)
  auto pick = [&](auto halo) {
    constexpr bool H = decltype(halo)::value;
    if (mode == 1) {                       // ma_nn: per-point outputs, always a search
      if (count) args(k_nn_accum<true, 1, 0, H>);
      else args(k_nn_accum<false, 1, 0, H>);
    } else if (with_lists && list_kind == 1) {
      if (count) args(k_nn_accum<true, 0, 1, H>);
      else args(k_nn_accum<false, 0, 1, H>);
    } else if (with_lists && list_kind == 2) {
      if (count) args(k_nn_accum<true, 0, 2, H>);
      else args(k_nn_accum<false, 0, 2, H>);
    } else {
      if (count) args(k_nn_accum<true, 0, 0, H>);
      else args(k_nn_accum<false, 0, 0, H>);
    }
  };
  if (ctx->grid_halo) pick(std::true_type{});
  else pick(std::false_type{});
  HIPCHK(hipGetLastError());
  return MA_OK;
}

}  // namespace mai

extern "C" {

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
int ma_debug_wavetimes(ma_ctx* ctx, uint32_t* wt) {
  HIPCHK(hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_wvt), sizeof(g_wvt)));
  return MA_OK;
}
int ma_debug_wgtimes(ma_ctx* ctx, unsigned long long* times, uint32_t* hw) {
  HIPCHK(hipMemcpyFromSymbol(times, HIP_SYMBOL(g_wgt), sizeof(g_wgt)));
  HIPCHK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wghw), sizeof(g_wghw)));
  return MA_OK;
}
#endif

}  // extern "C"
