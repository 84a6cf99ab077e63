// comm.hip — row e: the RCCL shim (librccl loaded with dlopen on first use), the communicator entry points and
// ma_icp_sharded, the loop of host_loop.hip with the iteration's 17 sums all-reduced over the communicator's ranks.
#include "ma_internal.h"
#include <random>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded with dlopen at ma_comm_init
#include <dlfcn.h>
#include <unistd.h>

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

}  // namespace

namespace mai {

void comm_destroy(ma_ctx* ctx) {
  RcclApi* api = rccl_api();
  if (api && ctx->comm) (void)api->CommDestroy(static_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_ranks = 0;
  ctx->comm_rank = 0;
}


// The 17 sums are 64-bit fixed-point integers (host_math.h "Exact sums"): whatever order the collective adds them in —
// ring, tree, any topology — the result is the same bits, and the same bits as one GPU summing the whole cloud.
int comm_allreduce_sums(ma_ctx* ctx) {
  NCCLCHK(rccl_api()->AllReduce(ctx->acc19.p, ctx->acc19.p, MA_ACC_LEN, ncclInt64, ncclSum, static_cast<ncclComm_t>(ctx->comm),
                                ctx->stream));
  return MA_OK;
}

// What bounds a correspondence's terms must be the same on every rank, or their fixed-point scales (and with them their
// integers) differ: the ranks gather each other's (n, wmax, rs) once per source cloud and every one installs
// (sum n, max wmax, max rs) — the bounds of the whole cloud, which is also what a single GPU holding all of it derives.
// The same agreement over the peer-to-peer exchange (no RCCL in the process): every rank copies its three numbers, then a
// tag, into its entry of the bounds area behind the rows of every rank's array, and reads its own array until all g
// entries carry this agreement's tag.  Host-driven (bootstrap time: once per source cloud), bounded by the exchange's
// timeout x 5.
static int p2p_agree_sum_bounds(ma_ctx* ctx, const double mine[3], std::vector<double>& all) {
  const int g = ctx->xchg_ranks;
  const unsigned long long gen = ++ctx->xchg_agree_gen;
  unsigned long long entry[4];
  std::memcpy(entry, mine, 24);
  entry[3] = gen;
  const size_t at = ma::kXBoundsOffset + (size_t)ctx->xchg_rank * 4;
  for (int r = 0; r < g; ++r) HIPCHK(hipMemcpyAsync(ctx->xchg_peer[r] + at, entry, 24, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));      // the numbers are in place everywhere before any tag is
  for (int r = 0; r < g; ++r) HIPCHK(hipMemcpyAsync(ctx->xchg_peer[r] + at + 3, entry + 3, 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<unsigned long long> got((size_t)g * 4);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    HIPCHK(hipMemcpyAsync(got.data(), ctx->xchg + ma::kXBoundsOffset, got.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bool ok = true;
    for (int r = 0; r < g; ++r) ok = ok && got[(size_t)r * 4 + 3] == gen;
    if (ok) break;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5 * (long long)ctx->xchg_timeout_ms)) {
      ctx->xchg_broken = true;
      return ctx->fail(MA_ERR_COMM, "peer-to-peer exchange: a rank's sum bounds did not arrive (are all ranks in ma_icp_sharded on the same source generation?)");
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  for (int r = 0; r < g; ++r) std::memcpy(&all[(size_t)r * 3], &got[(size_t)r * 4], 24);
  return MA_OK;
}

int comm_agree_sum_bounds(ma_ctx* ctx) {
  if (ctx->sum_bounds_set) return MA_OK;
  int rc;
  if ((rc = ensure_source_box(ctx))) return rc;
  const bool p2p = use_p2p(ctx);
  const int g = p2p ? ctx->xchg_ranks : ctx->comm_ranks;
  double mine[3] = {(double)ctx->n, ctx->src_wmax, sum_rs(ctx)};
  std::vector<double> all(3 * (size_t)std::max(g, 1));
  if (g > 1 && p2p) {
    if ((rc = p2p_agree_sum_bounds(ctx, mine, all))) return rc;
  } else if (g > 1) {
    const size_t bytes = sizeof(mine);
    HIPCHK(ctx->nn_tmp.ensure(bytes * (size_t)(g + 1)));
    char* base = static_cast<char*>(ctx->nn_tmp.p);
    HIPCHK(hipMemcpyAsync(base, mine, bytes, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(rccl_api()->AllGather(base, base + bytes, 3, ncclDouble, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    HIPCHK(hipMemcpyAsync(all.data(), base + bytes, bytes * (size_t)g, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    for (int i = 0; i < 3; ++i) all[i] = mine[i];
  }
  ma::SumBounds b{0.0, 0.0, 0.0};
  for (int r = 0; r < std::max(g, 1); ++r) {
    b.n_total += all[3 * r];
    b.wmax = std::max(b.wmax, all[3 * r + 1]);
    b.rs = std::max(b.rs, all[3 * r + 2]);
  }
  ctx->sum_bounds = b;
  ctx->sum_bounds_set = true;
  return MA_OK;
}


// the blob of ma_p2p_export (include/mapalign.h: MA_P2P_BLOB_BYTES opaque bytes)
struct P2PBlob {
  unsigned long long magic;
  long long pid;
  unsigned long long nonce;              // drawn once per process: equal pids in different pid namespaces are not one process
  int device, finegrained;
  unsigned long long ptr;
  hipIpcMemHandle_t handle;
};
static_assert(sizeof(P2PBlob) <= MA_P2P_BLOB_BYTES, "blob size");
constexpr unsigned long long kP2PMagic = 0x4D41503250763032ull;      // "MAP2Pv02"
unsigned long long process_nonce() {
  static const unsigned long long nonce = [] {
    std::random_device rd;
    unsigned long long v = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
    v ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    return v ? v : 1ull;
  }();
  return nonce;
}

void p2p_release(ma_ctx* ctx) {
  for (int r = 0; r < ma::kXMaxRanks; ++r) {
    if (ctx->xchg_peer[r] && ctx->xchg_peer_ipc[r]) (void)hipIpcCloseMemHandle(ctx->xchg_peer[r]);
    ctx->xchg_peer[r] = nullptr;
    ctx->xchg_peer_ipc[r] = false;
  }
  ctx->xchg_ranks = 0;
  ctx->xchg_rank = 0;
}

}  // namespace mai

extern "C" {

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

// ---- peer-to-peer exchange: bootstrap ----------------------------------------------------------------------------------------
// ma_p2p_export: allocate (once) and zero this context's array of rows and describe it in a blob any channel can carry;
// ma_p2p_init: map every rank's array — a pointer of this very process is used as it is (contexts of one process, on one
// device or on peers with access enabled), another process's through hipIpcOpenMemHandle.
int ma_p2p_export(ma_ctx* ctx, void* blob_out) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !blob_out) return MA_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  p2p_release(ctx);                     // (a new bootstrap: whatever was mapped before is let go)
  if (!ctx->xchg) {
    // fine-grained device memory where the runtime has it (coherent for peers' writes without cache maintenance);
    // otherwise ordinary device memory — every access to the rows is a system-scope atomic either way
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, ma::kXWords * 8, hipDeviceMallocFinegrained) == hipSuccess && p) {
      ctx->xchg_finegrained = true;
    } else {
      (void)hipGetLastError();
      p = nullptr;
      HIPCHK(hipMalloc(&p, ma::kXWords * 8));
      ctx->xchg_finegrained = false;
    }
    ctx->xchg = static_cast<unsigned long long*>(p);
  }
  HIPCHK(hipMemsetAsync(ctx->xchg, 0, ma::kXWords * 8, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->xchg_seq = 0;
  ctx->xchg_agree_gen = 0;
  ctx->xchg_broken = false;
  ctx->xchg_fresh = true;
  P2PBlob b{};
  b.magic = kP2PMagic;
  b.pid = (long long)getpid();
  b.nonce = process_nonce();
  b.device = ctx->device;
  b.finegrained = ctx->xchg_finegrained ? 1 : 0;
  b.ptr = (unsigned long long)reinterpret_cast<uintptr_t>(ctx->xchg);
  const hipError_t e = hipIpcGetMemHandle(&b.handle, ctx->xchg);
  if (e != hipSuccess) {                // same-process ranks need no handle: leave it zeroed and say so only when it is asked for
    (void)hipGetLastError();
    std::memset(&b.handle, 0, sizeof(b.handle));
    b.finegrained |= 2;                 // bit 1: no IPC handle
  }
  std::memset(blob_out, 0, MA_P2P_BLOB_BYTES);
  std::memcpy(blob_out, &b, sizeof(b));
  return MA_OK;
  });
}

int ma_p2p_init(ma_ctx* ctx, const void* blobs, int n_ranks, int rank) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (!blobs || n_ranks < 1 || n_ranks > ma::kXMaxRanks || rank < 0 || rank >= n_ranks)
    return ctx->fail(MA_ERR_INVALID, "bad arguments (1 <= n_ranks <= 16, 0 <= rank < n_ranks)");
  if (!ctx->xchg || !ctx->xchg_fresh)
    return ctx->fail(MA_ERR_STATE, "ma_p2p_export before every ma_p2p_init (the export zeroes this rank's rows; rows of an earlier set-up "
                                   "would pass for the new one's)");
  HIPCHK(hipSetDevice(ctx->device));
  p2p_release(ctx);
  const char* base = static_cast<const char*>(blobs);
  for (int r = 0; r < n_ranks; ++r) {
    P2PBlob b;
    std::memcpy(&b, base + (size_t)r * MA_P2P_BLOB_BYTES, sizeof(b));
    if (b.magic != kP2PMagic) { p2p_release(ctx); return ctx->fail(MA_ERR_INVALID, "blob " + std::to_string(r) + " is not a ma_p2p_export blob"); }
    const bool same_process = b.pid == (long long)getpid() && b.nonce == process_nonce();
    if (r == rank) {
      if (!same_process || b.ptr != (unsigned long long)reinterpret_cast<uintptr_t>(ctx->xchg)) {
        p2p_release(ctx);
        return ctx->fail(MA_ERR_INVALID, "blob [rank] is not this context's own export");
      }
      ctx->xchg_peer[r] = ctx->xchg;
    } else if (same_process) {
      if (b.device != ctx->device) {     // another device of this process: direct access over xGMI
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, ctx->device, b.device);
        if (!can) { p2p_release(ctx); return ctx->fail(MA_ERR_COMM, "no peer access between devices " + std::to_string(ctx->device) + " and " + std::to_string(b.device)); }
        const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { p2p_release(ctx); return ctx->fail(MA_ERR_COMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); }
        (void)hipGetLastError();
      }
      ctx->xchg_peer[r] = reinterpret_cast<unsigned long long*>(static_cast<uintptr_t>(b.ptr));
    } else {
      if (b.finegrained & 2) { p2p_release(ctx); return ctx->fail(MA_ERR_COMM, "rank " + std::to_string(r) + " could not export an IPC handle for its rows (hipIpcGetMemHandle failed there)"); }
      void* p = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess || !p) { (void)hipGetLastError(); p2p_release(ctx); return ctx->fail(MA_ERR_COMM, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e)); }
      ctx->xchg_peer[r] = static_cast<unsigned long long*>(p);
      ctx->xchg_peer_ipc[r] = true;
    }
  }
  ctx->xchg_ranks = n_ranks;
  ctx->xchg_rank = rank;
  ctx->xchg_fresh = false;
  ctx->xchg_seq = 0;
  ctx->xchg_agree_gen = 0;
  ctx->xchg_broken = false;
  ctx->sum_bounds_set = false;          // (bounds agreed over another group of ranks do not carry over)
  return MA_OK;
  });
}

int ma_p2p_destroy(ma_ctx* ctx) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  p2p_release(ctx);
  return MA_OK;
  });
}

int ma_comm_config(ma_ctx* ctx, int transport) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (transport != MA_TRANSPORT_RCCL && transport != MA_TRANSPORT_P2P) return ctx->fail(MA_ERR_INVALID, "transport: MA_TRANSPORT_RCCL or MA_TRANSPORT_P2P");
  if (transport != ctx->comm_transport) ctx->sum_bounds_set = ctx->sum_bounds_set && (ctx->comm_ranks == ctx->xchg_ranks);
  ctx->comm_transport = transport;
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
  if ((ctx->comm || ctx->xchg_ranks > 0) && ctx->have_src) {
    if (use_p2p(ctx) && ctx->xchg_broken)
      return ctx->fail(MA_ERR_COMM, "the peer-to-peer exchange ended a run in a communication failure: set it up again (ma_p2p_export / ma_p2p_init on every rank)");
    HIPCHK(hipSetDevice(ctx->device));
    const int rc = comm_agree_sum_bounds(ctx);
    if (rc) return rc;
  }
  return icp_loop(ctx, params, init, T_out, stats, true);
  });
}

}  // extern "C"
