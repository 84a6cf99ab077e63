// host_ctx.hip — the C entry points that create and destroy a context and install clouds in it (ma_create ...
// ma_set_pair_f64); no kernels here: the staging work is kernels_build.hip's.
#include "ma_internal.h"

namespace mai {

thread_local std::string g_create_error;
// the message of the calling THREAD's last ma_ctx::fail: ma_set_pair runs two threads on one context, and which of two
// failures a context reports must not depend on which thread wrote its message last
thread_local std::string t_last_fail;

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
#endif
//
// Candidate lists are a per-context setting (ma_list_config; MA_FLAG_NO_LISTS switches them off for one run).  The
// environment only supplies the defaults a context is created with: MAPALIGN_LISTS=0 every launch searches,
// MAPALIGN_SKIN_DIV=<d> skin = cell edge / d, MAPALIGN_MIN_USES=<n>, MAPALIGN_BUILD_FRAC / MAPALIGN_RHO_MIN (the planner's
// prediction constants; sweeps).  Nothing is read from the environment
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
  e = std::getenv("MAPALIGN_BUILD_FRAC");
  const double bf = e ? std::atof(e) : 0.0;
  ctx->cfg_build_frac = bf >= 0.05 && bf <= 0.49 ? bf : ma::kBuildFracDefault;
  e = std::getenv("MAPALIGN_RHO_MIN");
  const double rm = e ? std::atof(e) : 0.0;
  ctx->cfg_rho_min = rm >= 0.3 && rm <= 1.0 ? rm : ma::kRhoMinDefault;
  e = std::getenv("MAPALIGN_PLANNER_TIMEOUT_US");       // fault injection for the tests: make the planner's wait give up early
  const long t = e ? std::atol(e) : 0;
  if (t >= 1 && t <= 60L * 1000 * 1000) ctx->cfg_planner_timeout_us = (int)t;
  e = std::getenv("MAPALIGN_P2P_TIMEOUT_MS");           // how long a solving wave waits for its peers' sums (tests shorten it)
  const long tp = e ? std::atol(e) : 0;
  if (tp >= 1 && tp <= 30000) ctx->xchg_timeout_ms = (int)tp;
  e = std::getenv("MAPALIGN_P2P_GENERIC");              // tests: exercise the instantiation that 9-16 ranks use with fewer ranks
  ctx->xchg_force_generic = e && std::atoi(e) == 1;
}

}  // namespace mai

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
  ctx->host_clk0 = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->host_done) + 64);
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
  p2p_release(ctx);
  if (ctx->xchg) { (void)hipFree(ctx->xchg); ctx->xchg = nullptr; }
  try {
    for (DevBuf* b : ctx->all_bufs()) b->release();
  } catch (...) {   // (the buffer list is a std::vector: never let its allocation failure cross the ABI)
  }
  if (ctx->host_done) (void)hipHostFree(ctx->host_done);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->batch_stage) (void)hipHostFree(ctx->batch_stage);
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
  source_replaced(ctx);
  const uint32_t n = (uint32_t)m;
  HIPCHK(ctx->raw.ensure((size_t)n * 12));
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, xyz, (size_t)n * 12, hipMemcpyDefault, ctx->stream));   // host OR device memory (include/mapalign.h)
  return build_gold_from_raw(ctx, n, cell);
  });
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

int ma_grid_box_get(const ma_ctx* ctx, double lo[3], double hi[3], int* bulk_only) {
  if (!ctx || !lo || !hi) return MA_ERR_INVALID;
  if (!ctx->have_gold) return MA_ERR_STATE;
  for (int a = 0; a < 3; ++a) { lo[a] = (double)ctx->grid_lo[a]; hi[a] = (double)ctx->grid_hi[a]; }
  if (bulk_only) *bulk_only = ctx->grid_halo ? 1 : 0;
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
  source_replaced(ctx);
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
  source_replaced(ctx);
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
        if ((e = launch_f64_to_raw(ctx->raw64_src.as<double>(), n3, shift, ctx->raw_src.as<float>(), ctx->aux_stream)) != hipSuccess) { rc_src = ctx->fail(MA_ERR_HIP, std::string("k_f64_to_raw: ") + hipGetErrorString(e)); return; }
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
  source_replaced(ctx);
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
  source_replaced(ctx);
  int rc = stage_f64(ctx, xyz, (uint32_t)n64, false, s);
  if (rc) return rc;
  return build_source_from_raw(ctx, (uint32_t)n64, weights);   // (non-finite coordinates are caught by its fp32 bounds pass)
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

}  // extern "C"
