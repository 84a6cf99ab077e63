// kernels_solve.hip — rows a5 (final sum), a6, a7: the 19-way sum of the per-workgroup partials, the Kabsch step
// (kabsch.h), transform composition and convergence test on the device, the ranking of the search launch's
// scheduling groups, and the tiny launch that installs a run's initial state; then the host functions that launch them.
#include "ma_internal.h"
#include "kabsch.h"

namespace {

#if MA_WGTIMES   // debug variant only (tools/rstimeline.py)
__device__ unsigned long long g_rs[64][20][5];   // k_reduce_solve, per iteration and workgroup: entry, sum stored, ticket back, 19 sums read, solve done
#endif

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
// The sums are 64-bit fixed-point INTEGERS (host_math.h "Exact sums"): integer addition is associative, so neither the
// assignment of partials to threads nor the tree below matters for the result — only for the time.
__device__ __forceinline__ double sum_value(const long long v, const int k, const SumScale& sc) {
  // accumulator k as a number: v 2^-e (exact up to 53 bits); the two work counters (k >= 17) are plain counts
  const double x = (double)v;
  return k < ma::kAccLen ? x * pow2_f64(-sc.e[ma::sum_group(k)]) : x;
}

__global__ __launch_bounds__(kBlock) void k_reduce19(const long long* __restrict__ partials, const int G,
                                                     const DevState* __restrict__ st,
                                                     long long* __restrict__ acc_out,
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
    if (threadIdx.x == 0) acc_out[k] = 0;
    return;
  }
  const long long* __restrict__ p = partials + (size_t)k * G;
  unsigned long long v = 0ull;
  for (int b0 = threadIdx.x; b0 < G; b0 += 8 * kBlock) {
    long long u[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {     // eight independent loads in flight per thread
      const int b = b0 + t * kBlock;
      u[t] = b < G ? p[b] : 0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) v += (unsigned long long)u[t];
  }
  __shared__ unsigned long long red[kBlock / kWave];
  v = wave_sum_u64(v);
  if ((threadIdx.x & (kWave - 1)) == kSumLane) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) acc_out[k] = (long long)(red[0] + red[1] + red[2] + red[3]);
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
    s0.clk = wall_clock64();
    *st = s0;
    if (host_mirror) *host_mirror = s0;
    if (host_done) *host_done = s0.iter;     // iterations run when the loop stopped (never 0)
    if (slot) {                              // every iteration up to and including the one that stops the loop writes its slot
      *const_cast<unsigned long long*>(&slot->chk) = slot_check(s0.R, s0.t, s0.iter, slot_flags(s0.done, s0.converged, s0.degenerate));
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
  s0.clk = wall_clock64();
  *st = s0;
  // pinned host copy of the state (ma_icp): written with every solve, read by the host after its one
  // synchronisation — no device-to-host copy at the end of the loop (posted writes: the solve does not wait for them)
  if (host_mirror) *host_mirror = s0;
  if (slot) {      // candidate lists: host_mirror is this iteration's own pinned slot, and the host's planner reads it (ma_icp)
    // plain stores (the fields are volatile for the HOST's reads): a volatile store makes the wavefront wait for the
    // posted write to be acknowledged, 0.7 us each; the checksum — over the transform, the count AND the flags the host's stop
    // rule reads — makes their order irrelevant
    *const_cast<unsigned long long*>(&slot->chk) = slot_check(s0.R, s0.t, s0.iter, slot_flags(s0.done, s0.converged, s0.degenerate));
    *const_cast<int*>(&slot->seq) = s0.iter;
  }
#if MA_WGTIMES
  if (it < 64) g_rs[it][19][3] = wall_clock64();
#endif
}

// The solve alone (sharded-source stepping: the 17 sums come back from the all-reduce).
__global__ __launch_bounds__(kWave) void k_solve(const long long* __restrict__ acc_in, DevState* __restrict__ st,
                                                 double* __restrict__ trace, const int trace_cap,
                                                 const double tol, int* __restrict__ host_done,
                                                 DevState* __restrict__ host_mirror, StateSlot* __restrict__ slot,
                                                 const SumScale sc) {
  if (st->done || threadIdx.x != 0) return;
  double a[kNAcc];
#pragma unroll
  for (int k = 0; k < kNAcc; ++k) a[k] = sum_value(acc_in[k], k, sc);
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
__global__ __launch_bounds__(kBlock) void k_reduce_solve(const long long* __restrict__ partials, const int G,
                                                         DevState* __restrict__ st, long long* __restrict__ acc_out,
                                                         uint32_t* __restrict__ ticket, double* __restrict__ trace,
                                                         const int trace_cap, const double tol,
                                                         const uint32_t* __restrict__ wg_cost,
                                                         uint32_t* __restrict__ order, const uint32_t gshift,
                                                         int* __restrict__ host_done, DevState* __restrict__ host_mirror,
                                                         StateSlot* __restrict__ slot, const SumScale sc) {
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
  // this accumulator's exponent: fetched now (a scalar load beside the partials' loads), used after the sum
  int e_k = k < ma::kAccLen ? sc.e[ma::sum_group(k)] : 0;
  const long long* __restrict__ p = partials + (size_t)k * G;
  unsigned long long v = 0ull;
  for (int b0 = threadIdx.x; b0 < G; b0 += 16 * kBlock) {
    long long u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {    // sixteen independent loads in flight per thread
      const int b = b0 + t * kBlock;
      u[t] = b < G ? p[b] : 0;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) v += (unsigned long long)u[t];
  }
  if (done) return;
  asm volatile("" : "+s"(e_k));
  __shared__ unsigned long long red[kBlock / kWave];
  v = wave_sum_u64(v);
  if ((threadIdx.x & (kWave - 1)) == kSumLane) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const long long sum = (long long)(red[0] + red[1] + red[2] + red[3]);
  // The sum leaves this workgroup as a NUMBER already (v 2^-e, the bits of the double in the 8-byte slot): nineteen lanes
  // on nineteen CUs convert side by side, where the solving lane would convert one after the other — on a lone wavefront,
  // 19 x (int64 -> fp64, scale) is 0.3-0.4 us of the iteration (profiles/r05_variants_ab.txt, call 3: 8.1 -> 8.4 us).
  const double value = k < ma::kAccLen ? (double)sum * pow2_f64(-e_k) : (double)sum;
  __hip_atomic_store(&acc_out[k], __double_as_longlong(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // this accumulator's entry of the iteration's trace row (columns 0-16 the sums, 17 the rmse — the solve's —, 18 and 19
  // the two work counters)
  if (it_now < trace_cap) trace[(size_t)it_now * MA_TRACE_STRIDE + (k < ma::kAccLen ? k : k + 1)] = value;
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
  for (int t = 0; t < kNAcc; ++t) a[t] = __longlong_as_double(__hip_atomic_load(&acc_out[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
// The state and the list control block a run starts from, handed over BY VALUE in the kernel's arguments: one tiny
// launch instead of two small host-to-device copies in front of every loop (each copy is a DMA or blit of its own on
// the stream, ~5 us of host time and as much on the device's).
// It also stamps the run's start (the device's 100 MHz clock, into pinned host memory): the loop's device time is the last
// solve's stamp minus this one — no timing events around the loop (two host calls and ~10 us of a 2.2 ms alignment,
// profiles/r04_variants_ab.txt call 57).
__global__ __launch_bounds__(kWave) void k_begin_run(const DevState s, const ListCtl c, DevState* __restrict__ st,
                                                     ListCtl* __restrict__ lc, unsigned long long* __restrict__ host_clk0) {
  if (threadIdx.x == 0) { *host_clk0 = wall_clock64(); *st = s; *lc = c; }
}
// Row e without a library collective: a5 (final sum) + the EXCHANGE + a6 + a7 in one launch — a sharded iteration is two
// launches, as an unsharded one is.  The 19 summing workgroups are k_reduce_solve's; the first wavefront of the one
// that arrives last then
//   1. writes this rank's 17 sums, one 8-byte word per lane, into its row of the iteration's slot in EVERY rank's array
//      (host_math.h "Peer-to-peer exchange": tagged, self-validating words; system-scope stores, all in flight together);
//   2. polls the g rows of that slot in its OWN array until every word carries this iteration's tag — bounded: after
//      `timeout_ticks` of the 100 MHz clock the loop stops with degenerate = 2 (the host reports MA_ERR_COMM), so every
//      wave leaves the kernel whatever the peers do;
//   3. adds the g rows (integers: any order gives the same bits), converts, and lane 0 solves.
// Every rank computes the same sums, hence the same state: nothing else is exchanged.
// NP = the number of PEERS (ranks - 1) as a template parameter, 0 ... 7 exact, 15 = "up to 15, guarded": the loops over the
// peers are unrolled without guards, the peers' row pointers are plain kernel arguments fetched with the others at the top,
// and the g - 1 stores, then the g - 1 polling loads, are in flight together — one peer round trip per poll.  (A lone
// wavefront issues an instruction every ~8 cycles: the first form, unrolled over 16 possible ranks with a uniform guard
// each, spent 1.2 us of every iteration on guards; a rolled loop waited for each peer's load before issuing the next.)
// A rank neither writes nor reads its own row: it holds those sums in registers (a 1-rank exchange touches no row).
struct XPeers {
  unsigned long long* rows[ma::kXMaxRanks - 1];      // the PEERS' arrays as mapped here, in rank order without this rank
  int rank_of[ma::kXMaxRanks - 1];                   // their ranks (which row of MY array each of them writes)
  int n_peers, rank;
};
template <int NP>
__global__ __launch_bounds__(kBlock) void k_reduce_exchange_solve(
    const long long* __restrict__ partials, const int G, DevState* __restrict__ st, long long* __restrict__ acc_out,
    uint32_t* __restrict__ ticket, double* __restrict__ trace, const int trace_cap, const double tol,
    const uint32_t* __restrict__ wg_cost, uint32_t* __restrict__ order, const uint32_t gshift, int* __restrict__ host_done,
    DevState* __restrict__ host_mirror, StateSlot* __restrict__ slot, const SumScale sc, const XPeers xp,
    const unsigned long long* __restrict__ my_rows, const unsigned long long seq_base, const unsigned long long timeout_ticks) {
  const int done = st->done;
  const unsigned long long t_launch = wall_clock64();      // (read here, beside the first loads: the wait below is timed from the launch)
  if (blockIdx.x == kNAcc) {
    rank_groups(G, wg_cost, order, gshift, done);
    return;
  }
  const int k = blockIdx.x;
  const int it_now = st->iter;
  DevState s0;
  if (threadIdx.x == 0) s0 = *st;
  const long long* __restrict__ p = partials + (size_t)k * G;
  unsigned long long v = 0ull;
  for (int b0 = threadIdx.x; b0 < G; b0 += 16 * kBlock) {
    long long u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int b = b0 + t * kBlock;
      u[t] = b < G ? p[b] : 0;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) v += (unsigned long long)u[t];
  }
  if (done) return;
  __shared__ unsigned long long red[kBlock / kWave];
  __shared__ uint32_t sh_last;
  v = wave_sum_u64(v);
  if ((threadIdx.x & (kWave - 1)) == kSumLane) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long sum = (long long)(red[0] + red[1] + red[2] + red[3]);
    __hip_atomic_store(&acc_out[k], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = arrived == (uint32_t)(kNAcc - 1) ? 1u : 0u;
    if (sh_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
  __syncthreads();
  if (!sh_last || threadIdx.x >= kWave) return;
  // ---- the first wavefront of the last arriver: exchange and solve
  const int lane = threadIdx.x;
  const unsigned long long seq = seq_base + (unsigned long long)it_now;
  const uint32_t xslot = (uint32_t)(seq % (unsigned long long)ma::kXSlots);
  const uint32_t tag = ma::xword_tag(seq);
  const long long mine = lane < kNAcc ? __hip_atomic_load(&acc_out[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  const int word = lane < ma::kAccLen ? lane : 0;
  const int np = NP <= 7 ? NP : xp.n_peers;
  if (NP > 0 && lane < ma::kAccLen) {
    const unsigned long long w = ma::xword_encode(mine, seq);
    const size_t at = ((size_t)xslot * ma::kXMaxRanks + (size_t)xp.rank) * ma::kXRowWords + (size_t)lane;
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (NP <= 7 || j < np) __hip_atomic_store(xp.rows[j] + at, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned long long total = (unsigned long long)mine;
  bool arrived_all = true;
  if (NP > 0) {
    const unsigned long long* my = my_rows + (size_t)xslot * ma::kXMaxRanks * ma::kXRowWords + (size_t)word;
    arrived_all = false;
    for (;;) {
      unsigned long long w[NP > 0 ? NP : 1];
#pragma unroll
      for (int j = 0; j < NP; ++j)
        w[j] = (NP <= 7 || j < np) ? __hip_atomic_load(my + (size_t)xp.rank_of[j] * ma::kXRowWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                   : ((unsigned long long)tag << 62);      // an absent peer: a valid zero
      bool all = true;
      total = (unsigned long long)mine;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        all = all && (uint32_t)(w[j] >> 62) == tag;
        total += (unsigned long long)ma::xword_value(w[j]);
      }
      if (__ballot(!all && lane < ma::kAccLen) == 0ull) { arrived_all = true; break; }
      if (wall_clock64() - t_launch > timeout_ticks) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  const double mine_value = lane < ma::kAccLen ? sum_value((long long)total, lane, sc) : (double)mine;     // lanes 17, 18: this rank's work counters
  // lane t's value to the solving lane: v_readlane into scalar registers (a shuffle is an LDS round trip each)
  double a[kNAcc];
  const int mv_lo = __double2loint(mine_value), mv_hi = __double2hiint(mine_value);
#pragma unroll
  for (int t = 0; t < kNAcc; ++t) a[t] = __hiloint2double(__builtin_amdgcn_readlane(mv_hi, t), __builtin_amdgcn_readlane(mv_lo, t));
  if (lane != 0) return;
  if (!arrived_all) {
    // a peer's rows did not come: stop the loop where it stands (every rank whose peers are silent does the same after
    // its own wait) and say why — degenerate = 2 is MA_ERR_COMM on the host
    s0.iter = it_now + 1;
    s0.degenerate = 2;
    s0.done = 1;
    s0.clk = wall_clock64();
    *st = s0;
    if (host_mirror) *host_mirror = s0;
    if (host_done) *host_done = s0.iter;
    if (slot) {
      *const_cast<unsigned long long*>(&slot->chk) = slot_check(s0.R, s0.t, s0.iter, slot_flags(s0.done, s0.converged, s0.degenerate));
      *const_cast<int*>(&slot->seq) = s0.iter;
    }
    return;
  }
  solve_step<true>(a, s0, st, trace, trace_cap, tol, host_done, host_mirror, slot);
}

}  // namespace

namespace mai {

int launch_reduce_exchange_solve(ma_ctx* ctx, double tol, const SumScale& sc, StateSlot* slot) {
  XPeers xp{};
  int np = 0;
  for (int r = 0; r < ctx->xchg_ranks; ++r) {
    if (r == ctx->xchg_rank) continue;
    xp.rows[np] = ctx->xchg_peer[r];
    xp.rank_of[np] = r;
    ++np;
  }
  xp.n_peers = np;
  xp.rank = ctx->xchg_rank;
  const unsigned long long ticks = (unsigned long long)ctx->xchg_timeout_ms * 100000ull;      // wall_clock64: 100 MHz
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(kNAcc + 1), dim3(kBlock), 0, ctx->stream, ctx->partials.as<long long>(),
                       ctx->last_G, ctx->state.as<DevState>(), ctx->acc19.as<long long>(), ctx->ticket.as<uint32_t>(),
                       ctx->trace.as<double>(), ctx->trace_cap, tol, ctx->wg_cost.as<uint32_t>(), ctx->seg_order.as<uint32_t>(),
                       ctx->last_gshift, ctx->host_done, slot ? &slot->s : ctx->host_state_out, slot, sc, xp,
                       (const unsigned long long*)ctx->xchg, ctx->xchg_seq, ticks);
  };
  switch (ctx->xchg_force_generic && np > 0 ? 15 : np) {
    case 0: go(k_reduce_exchange_solve<0>); break;
    case 1: go(k_reduce_exchange_solve<1>); break;
    case 2: go(k_reduce_exchange_solve<2>); break;
    case 3: go(k_reduce_exchange_solve<3>); break;
    case 4: go(k_reduce_exchange_solve<4>); break;
    case 5: go(k_reduce_exchange_solve<5>); break;
    case 6: go(k_reduce_exchange_solve<6>); break;
    case 7: go(k_reduce_exchange_solve<7>); break;
    default: go(k_reduce_exchange_solve<15>); break;
  }
  HIPCHK(hipGetLastError());
  return MA_OK;
}

int launch_begin_run(ma_ctx* ctx, const DevState& s, const ListCtl& c) {
  hipLaunchKernelGGL(k_begin_run, dim3(1), dim3(kWave), 0, ctx->stream, s, c, ctx->state.as<DevState>(), ctx->listctl.as<ListCtl>(),
                     ctx->host_clk0);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// partials -> ctx->acc19 (19 workgroups), then optionally the solve (one wavefront)
// with_rank: one more workgroup orders the scheduling groups for the next search launch
int launch_reduce(ma_ctx* ctx, bool with_rank) {
  hipLaunchKernelGGL(k_reduce19, dim3(kNAcc + (with_rank ? 1 : 0)), dim3(kBlock), 0, ctx->stream,
                     ctx->partials.as<long long>(), ctx->last_G, ctx->state.as<DevState>(), ctx->acc19.as<long long>(),
                     ctx->wg_cost.as<uint32_t>(), ctx->seg_order.as<uint32_t>(), ctx->last_gshift);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// the same sum plus the solve by the last-arriving workgroup, one launch
int launch_reduce_solve(ma_ctx* ctx, double tol, const SumScale& sc, StateSlot* slot) {
  hipLaunchKernelGGL(k_reduce_solve, dim3(kNAcc + 1), dim3(kBlock), 0, ctx->stream, ctx->partials.as<long long>(),
                     ctx->last_G, ctx->state.as<DevState>(), ctx->acc19.as<long long>(), ctx->ticket.as<uint32_t>(),
                     ctx->trace.as<double>(), ctx->trace_cap, tol, ctx->wg_cost.as<uint32_t>(),
                     ctx->seg_order.as<uint32_t>(), ctx->last_gshift, ctx->host_done,
                     slot ? &slot->s : ctx->host_state_out, slot, sc);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// in_loop: inside the sharded loop (pinned `done` word, pinned mirror of the state, planner slot)
int launch_solve(ma_ctx* ctx, double tol, const SumScale& sc, bool in_loop, StateSlot* slot) {
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(kWave), 0, ctx->stream, ctx->acc19.as<long long>(),
                     ctx->state.as<DevState>(), ctx->trace.as<double>(), ctx->trace_cap, tol,
                     in_loop ? ctx->host_done : (int*)nullptr,
                     in_loop ? (slot ? &slot->s : ctx->host_state_out) : (DevState*)nullptr, slot, sc);
  HIPCHK(hipGetLastError());
  return MA_OK;
}

}  // namespace mai

#if MA_WGTIMES
extern "C" int ma_debug_rstimes(ma_ctx* ctx, unsigned long long* rs) {
  HIPCHK(hipMemcpyFromSymbol(rs, HIP_SYMBOL(g_rs), sizeof(g_rs)));
  return MA_OK;
}
#endif
