// host_loop.hip — rows a4-a7 on the host side: icp_loop, the loop of ma_icp and ma_icp_sharded (which kernel every
// iteration launches — list_planner.h —, the pinned slots the solves report into, the early exit), and the entry points
// that step, trace and inspect it.  No kernels here: launches go through kernels_search.hip / kernels_solve.hip.
#include "ma_internal.h"
#include "list_planner.h"

namespace mai {

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
int ensure_source_box(ma_ctx* ctx) {
  if (!ctx->src_box_pending) return MA_OK;
  // a source cloud from a depth frame: its box (stored, centred coordinates; holes skipped)
  float lo[3], hi[3];
  const int rc = cloud_bounds(ctx, ctx->src.as<float>(), ctx->n, lo, hi, 4, true);
  if (rc) return rc;
  double r2 = 0.0;
  for (int a = 0; a < 3; ++a) {
    const double pad = 1e-4 + 1e-6 * (std::fabs((double)lo[a]) + std::fabs((double)hi[a]));
    ctx->src_lo[a] = (float)((double)lo[a] - pad);
    ctx->src_hi[a] = (float)((double)hi[a] + pad);
    r2 += std::max((double)lo[a] * lo[a], (double)hi[a] * hi[a]);
    ctx->src_amax = std::max({ctx->src_amax, std::fabs((double)lo[a]), std::fabs((double)hi[a])});
  }
  ctx->src_rs = (float)(std::sqrt(r2) * 1.001 + 1e-4);
  ctx->src_box_pending = false;
  return MA_OK;
}

int run_sum_scale(ma_ctx* ctx, double gate, SumScale* out) {
  ma::SumBounds b = ctx->sum_bounds;
  if (!ctx->sum_bounds_set) {
    if (!std::isfinite(gate)) { const int rc = ensure_source_box(ctx); if (rc) return rc; }     // only an ungated run's bounds look at rs
    b.n_total = (double)ctx->n;
    b.wmax = ctx->src_wmax;
    b.rs = sum_rs(ctx);
  }
  *out = ma::sum_scale(b, ctx->gold_qb, gate, ctx->run_t0);
  return MA_OK;
}

int ensure_loop_buffers(ma_ctx* ctx, int trace_rows) {
  HIPCHK(ctx->state.ensure(sizeof(DevState)));
  HIPCHK(ctx->partials.ensure((size_t)kNAcc * grid_for(std::max<uint32_t>(ctx->n, 1)) * sizeof(double)));
  HIPCHK(ctx->acc19.ensure(kNAcc * sizeof(long long)));
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
int upload_state(ma_ctx* ctx, const double* T) {
  int rc = ensure_loop_buffers(ctx, std::max(ctx->trace_cap, 64));
  if (rc) return rc;
  DevState s{};
  world_to_centred(T, ctx->origin, s);
  s.prev_rmse = NAN; s.last_rmse = NAN; s.last_W = 0;
  s.iter = 0; s.done = 0; s.converged = 0; s.degenerate = 0;
  ctx->run_t0 = std::sqrt(s.t[0] * s.t[0] + s.t[1] * s.t[1] + s.t[2] * s.t[2]);      // (an ungated run's sum bounds)
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
    { const int rc_b = launch_begin_run(ctx, s, c); if (rc_b) return rc_b; }
  }
  *ctx->host_done = 0;   // every earlier launch has completed: each entry point that runs the loop ends with a sync
  ctx->state_valid = ctx->have_src;
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
// The loop of ma_icp and ma_icp_sharded (rows a4-a7).  One pair on this context; `sharded`: this context holds a shard
// of the comparison cloud and the 17 sums of every iteration are all-reduced over the communicator's ranks.
//   per iteration, all on the ctx stream, no host synchronisation:
//     unsharded   k_nn_accum (search | search + write lists | read lists)  ->  k_reduce_solve
//     sharded     k_nn_accum (the same three kinds, local shard)  ->  k_reduce19 (+ schedule ranking)
//                 ->  ncclAllReduce(sum, 17 x f64, in place)  ->  k_solve (redundantly on every rank)
// Identical sums give identical states on all ranks, so nothing is broadcast; each rank plans its own list launches
// (its shard has its own box) — which kernel a rank launches changes no bit of its sums, and every rank issues exactly
// one collective per iteration.
int icp_loop(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16], ma_stats* stats,
                    const bool sharded) {
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if ((rc = need_ready(ctx))) return rc;
  if (sharded && !ctx->comm && ctx->xchg_ranks == 0) return ctx->fail(MA_ERR_STATE, "no communicator (ma_comm_init, or ma_p2p_export + ma_p2p_init)");
  const bool p2p = sharded && use_p2p(ctx);
  if (!T_out) return ctx->fail(MA_ERR_INVALID, "T_out is NULL");
  if (init) for (int i = 0; i < 16; ++i) if (!std::isfinite(init[i])) return ctx->fail(MA_ERR_INVALID, "non-finite init");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, params->max_corr_dist))) return rc;
  if ((rc = ensure_loop_buffers(ctx, params->max_iter))) return rc;
  const bool lists_allowed = lists_enabled(ctx) && !(params->flags & MA_FLAG_NO_LISTS);
  const bool long_run = params->max_iter > 3 * kPlan && params->max_iter <= kMaxSlots;
  if (lists_allowed && long_run && (rc = ensure_source_box(ctx))) return rc;      // (a depth frame's box: the candidate lists need it)
  if ((rc = upload_state(ctx, init))) return rc;
  SumScale sc;
  if ((rc = run_sum_scale(ctx, params->max_corr_dist, &sc))) return rc;
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
  // the solves mirror their state into the slots, not into host_state_out.  A long SHARDED run always uses them: where a
  // stopped loop (converged — or degenerate, which tol = 0 does not rule out) stops enqueueing must not depend on
  // anything rank-local, and `lists_wanted` is (a shard's size, its box, its list settings): a rank that broke off at the
  // next batch while a peer went on enqueueing all-reduces would hang the node
  const bool slots_used = long_run && (lists_wanted || early_exit || sharded);
  const bool batch_exit = early_exit && !slots_used && !sharded;
  if (batch_exit) {
    const int need = (params->max_iter + kBatch - 1) / kBatch;
    while ((int)ctx->batch_ev.size() < need) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->batch_ev.push_back(e);
    }
  }
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
  planner.configure(skin, params->max_iter, ctx->src_lo, ctx->src_hi, ctx->cfg_min_uses, ctx->cfg_build_frac, ctx->cfg_rho_min);
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
    if ((rc = launch_nn(ctx, 0, count, best2, nullptr, nullptr, sc, MA_SCHED != 0, kind, k > 0 ? 1 : 0))) return rc;
    if (timek) HIPCHK(hipEventRecord(ctx->kev[2 * k + 1], ctx->stream));
    StateSlot* slot = slots_used ? &ctx->host_slots[k] : nullptr;
    if (p2p) {
      if ((rc = launch_reduce_exchange_solve(ctx, params->tol, sc, slot))) return rc;
    } else if (sharded) {
      if ((rc = launch_reduce(ctx, MA_SCHED != 0))) return rc;
      if ((rc = comm_allreduce_sums(ctx))) return rc;
      if ((rc = launch_solve(ctx, params->tol, sc, true, slot))) return rc;
    } else {
      if ((rc = launch_reduce_solve(ctx, params->tol, sc, slot))) return rc;
    }
    launched = k + 1;
    if (timek) HIPCHK(hipEventRecord(ctx->kev[2 * k + 2], ctx->stream));
    if (batch_exit && (k + 1) % kBatch == 0) HIPCHK(hipEventRecord(ctx->batch_ev[k / kBatch], ctx->stream));
  }
  ctx->plan_diag[3] = launched;
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
  // device time of the loop: from k_begin_run's clock stamp to the stamp of the last solve that ran (100 MHz, one clock for
  // the whole device) — launches enqueued past a stop only find `done` set and are not part of it
  const unsigned long long clk0 = *ctx->host_clk0;
  const double ms = s.iter > 0 && s.clk > clk0 ? (double)(s.clk - clk0) * 1e-5 : 0.0;
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
  if (p2p) {
    // iterations the ranks have solved together: the next run's rows go on from here (the same number on every rank —
    // identical sums, identical states, identical stops)
    ctx->xchg_seq += (unsigned long long)std::max(s.iter, 0);
    if (s.degenerate == 2) {
      ctx->xchg_broken = true;
      return ctx->fail(MA_ERR_COMM, "peer-to-peer exchange: a peer's sums did not arrive within " + std::to_string(ctx->xchg_timeout_ms) +
                                    " ms (iteration " + std::to_string(s.iter) + "); the loop was stopped");
    }
  }
  if (s.degenerate) return ctx->fail(MA_ERR_DEGENERATE, "no correspondence inside max_corr_dist, or rank-deficient covariance (fewer than 3 non-collinear pairs)");
  return MA_OK;
}

}  // namespace mai

extern "C" {

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
  SumScale sc;
  if ((rc = run_sum_scale(ctx, params->max_corr_dist, &sc))) return rc;
  if ((rc = launch_nn(ctx, 0, (params->flags & MA_FLAG_COUNT_WORK) != 0, open_gate2(params->max_corr_dist), nullptr, nullptr, sc))) return rc;
  if ((rc = launch_reduce(ctx))) return rc;
  if ((rc = launch_solve(ctx, params->tol, sc))) return rc;
  long long ai[kNAcc];
  HIPCHK(hipMemcpyAsync(ai, ctx->acc19.p, sizeof(ai), hipMemcpyDeviceToHost, ctx->stream));
  DevState s;
  if ((rc = fetch_state(ctx, &s))) return rc;
  double a[kNAcc];
  for (int k = 0; k < kNAcc; ++k) a[k] = k < MA_ACC_LEN ? ma::sum_to_double(ai[k], sc.e[ma::sum_group(k)]) : (double)ai[k];
  if (acc_out) acc_to_world(a, ctx->origin, acc_out);
  if (T_out) centred_to_world(s, ctx->origin, T_out);
  if (s.degenerate) return ctx->fail(MA_ERR_DEGENERATE, "no correspondence inside max_corr_dist, or rank-deficient covariance (fewer than 3 non-collinear pairs)");
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
  if ((rc = launch_nn(ctx, 1, false, open_gate2(max_corr_dist), ctx->nn_idx.as<uint32_t>(), ctx->nn_d2.as<float>(), SumScale{}))) return rc;
  const uint32_t* perm = nullptr;
  if ((rc = source_map(ctx, &perm))) return rc;
  if (idx_out) {
    if ((rc = launch_unpermute4(ctx, ctx->nn_idx.p, perm, (uint32_t)n, ctx->nn_tmp.p))) return rc;
    HIPCHK(hipMemcpyAsync(idx_out, ctx->nn_tmp.p, n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (d2_out) {
    if ((rc = launch_unpermute4(ctx, ctx->nn_d2.p, perm, (uint32_t)n, ctx->nn_tmp.p))) return rc;
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

int ma_sum_bounds_get(ma_ctx* ctx, double out[3]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return MA_ERR_INVALID;
  if (!ctx->have_src) return ctx->fail(MA_ERR_STATE, "source cloud not set");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_source_box(ctx))) return rc;
  out[0] = (double)ctx->n;
  out[1] = ctx->src_wmax;
  out[2] = sum_rs(ctx);
  return MA_OK;
  });
}

int ma_sum_bounds_set(ma_ctx* ctx, const double in[3]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!in) { ctx->sum_bounds_set = false; return MA_OK; }
  if (!ctx->have_src) return ctx->fail(MA_ERR_STATE, "source cloud not set (the bounds belong to a source cloud: a new one clears them)");
  if (!(in[0] >= (double)ctx->n) || !(in[0] < 9.0e15) || !(in[1] >= ctx->src_wmax) || !std::isfinite(in[1]) ||
      !(in[2] >= 0.0) || !std::isfinite(in[2]))
    return ctx->fail(MA_ERR_INVALID, "sum bounds must cover this context's own (n_total >= n, wmax >= this cloud's, rs >= 0, all finite)");
  ctx->sum_bounds = ma::SumBounds{in[0], in[1], in[2]};
  ctx->sum_bounds_set = true;
  return MA_OK;
  });
}

int ma_sum_scale_get(ma_ctx* ctx, const ma_params* params, int32_t out[5]) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !out) return MA_ERR_INVALID;
  int rc;
  if ((rc = check_params(ctx, params))) return rc;
  if ((rc = need_ready(ctx))) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  SumScale sc;
  if ((rc = run_sum_scale(ctx, params->max_corr_dist, &sc))) return rc;
  for (int g = 0; g < 5; ++g) out[g] = sc.e[g];
  return MA_OK;
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
  if (!ctx->sum_bounds_set)
    return ctx->fail(MA_ERR_STATE, "split stepping needs agreed sum bounds: ma_sum_bounds_set with the whole cloud's (n_total, wmax, rs) on every "
                                   "shard's context (a lone context: its own ma_sum_bounds_get) — integers summed across contexts mean "
                                   "something only if all contexts scale them alike");
  HIPCHK(hipSetDevice(ctx->device));
  if ((rc = ensure_levels_for(ctx, params->max_corr_dist))) return rc;
  SumScale sc;
  if ((rc = run_sum_scale(ctx, params->max_corr_dist, &sc))) return rc;
  if ((rc = launch_nn(ctx, 0, (params->flags & MA_FLAG_COUNT_WORK) != 0, open_gate2(params->max_corr_dist), nullptr, nullptr, sc))) return rc;
  if ((rc = launch_reduce(ctx))) return rc;
  HIPCHK(hipMemcpyAsync(acc_dev, ctx->acc19.p, MA_ACC_LEN * sizeof(long long), hipMemcpyDeviceToDevice, ctx->stream));
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
  if (!ctx->sum_bounds_set)
    return ctx->fail(MA_ERR_STATE, "split stepping needs agreed sum bounds: ma_sum_bounds_set with the whole cloud's (n_total, wmax, rs) on every "
                                   "shard's context (a lone context: its own ma_sum_bounds_get) — integers summed across contexts mean "
                                   "something only if all contexts scale them alike");
  HIPCHK(hipSetDevice(ctx->device));
  // rows 17,18 (work counters) keep the local values of the accumulate step
  SumScale sc;
  if ((rc = run_sum_scale(ctx, params->max_corr_dist, &sc))) return rc;
  HIPCHK(hipMemcpyAsync(ctx->acc19.p, acc_dev, MA_ACC_LEN * sizeof(long long), hipMemcpyDeviceToDevice, ctx->stream));
  return launch_solve(ctx, params->tol, sc);
  });
}

}  // extern "C"
