// kernels_depth.hip — rows a8 (depth -> 3-D unprojection in place), a10 (2-D box -> 3-D global box), N4 (full-frame
// map builder: voxel keys, sort, per-voxel centroids) and the kernels that hand stored points back to the caller; then
// the host functions and C entry points that launch them.
#include "ma_internal.h"

namespace {

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

// BATCH (row N4, ma_map_add_depth_batch): blockIdx.y is the frame — frames stored back to back, outputs back to back,
// one camera->world pose per frame read from `poses` (12 doubles each, the translation already minus the map origin) —
// so that the n frames of a map cost one launch.  The point of a pixel is formed by the same spelled-out operations
// either way, so a batch of n frames writes the bits of n single calls.
template <typename T, bool BATCH>
__global__ __launch_bounds__(kBlock) void k_unproject(const T* __restrict__ depth, const int width,
                                                      const uint32_t npx, const Pinhole cam, const double* __restrict__ poses,
                                                      uint32_t* __restrict__ counts, float4* __restrict__ out) {
  double T12[12];
  if (BATCH) {
    depth += (size_t)blockIdx.y * npx;
    out += (size_t)blockIdx.y * npx;
    counts += (size_t)blockIdx.y * gridDim.x;
#pragma unroll
    for (int k = 0; k < 12; ++k) T12[k] = poses[12 * (size_t)blockIdx.y + k];
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) T12[k] = cam.T[k];
  }
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
        p.x = (float)__fma_rn(T12[0], X, __fma_rn(T12[1], Y, __fma_rn(T12[2], z[k], T12[3])));
        p.y = (float)__fma_rn(T12[4], X, __fma_rn(T12[5], Y, __fma_rn(T12[6], z[k], T12[7])));
        p.z = (float)__fma_rn(T12[8], X, __fma_rn(T12[9], Y, __fma_rn(T12[10], z[k], T12[11])));
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
}  // namespace

namespace mai {

int launch_unpermute4(ma_ctx* ctx, const void* in, const uint32_t* perm, uint32_t n, void* out) {
  hipLaunchKernelGGL(k_unpermute<uint32_t>, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream,
                     static_cast<const uint32_t*>(in), perm, n, static_cast<uint32_t*>(out));
  HIPCHK(hipGetLastError());
  return MA_OK;
}

// Depth frame already in ctx->depth_in -> one slot per pixel at `out` (k_unproject: points and holes).
// evu0/evu1 bracket the device work.  The kernel writes its per-workgroup counts of valid pixels straight into pinned
// host memory (the third mailbox block), so the NUMBER of valid pixels costs no device-to-host copy — and no
// synchronisation at all unless somebody asks for it: total != nullptr waits and adds the counts up; with nullptr the
// call only enqueues, and finish_unproject() settles the count when it is first needed (round 3: a pageable copy
// of the count array and a stream sync in every call, 0.21 ms of host wall for 14 us of device work).
// the frame is the source cloud: what its count of valid pixels decides
static int settle_source_count(ma_ctx* ctx) {
  ctx->n_pts = ctx->unproj_total;
  ctx->src_holes = ctx->unproj_total != ctx->n;
  if (ctx->unproj_total == 0) {
    ctx->have_src = false;
    return ctx->fail(MA_ERR_EMPTY, "depth image has no valid pixel");
  }
  return MA_OK;
}

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
  return ctx->unproj_is_source ? settle_source_count(ctx) : MA_OK;
}

static int unproject_frame(ma_ctx* ctx, int dtype, int width, uint32_t npx, const Pinhole& cam, float4* out, uint32_t* total,
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
    hipLaunchKernelGGL((k_unproject<float, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<float>(), width, npx, cam,
                       (const double*)nullptr, counts, out);
  else
    hipLaunchKernelGGL((k_unproject<uint16_t, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->depth_in.as<uint16_t>(), width, npx,
                       cam, (const double*)nullptr, counts, out);
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
    if (total) *total = ctx->unproj_total;      // the caller looks at the total itself (map builder)
    else if (is_source) return settle_source_count(ctx);
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
    HIPCHK(ctx->cub_tmp.ensure(scan_tmp_bytes(n)));
    HIPCHK(scan_exclusive_u32(ctx->cub_tmp.p, ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>(), n, ctx->stream));
    hipLaunchKernelGGL(k_rank_map, dim3(div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, ctx->keys_a.as<uint32_t>(),
                       ctx->keys_b.as<uint32_t>(), n, ctx->src_perm.as<uint32_t>());
    HIPCHK(hipGetLastError());
    ctx->src_map_ready = true;
  }
  *map = ctx->src_perm.as<uint32_t>();
  return MA_OK;
}

static Pinhole make_pinhole(const float K[4], const double T[16], float depth_scale, const double origin[3]) {
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

}  // namespace mai

extern "C" {

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
  source_replaced(ctx);
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
  ctx->src_bulk_box = false;
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

// Row N4 for a whole run of frames (pose_processor.py:88-122 is the reference's per-frame loop): ONE upload of n_frames
// frames stored back to back, one upload of their poses, ONE k_unproject launch (blockIdx.y = frame), one copy of the
// per-workgroup counts, one synchronisation — a single frame is launch-bound (24.6 MB in 11 us: 0.28 of the HBM roofline),
// a batch of them is not.  The map holds exactly what n_frames ma_map_add_depth calls would have put there.
int ma_map_add_depth_batch(ma_ctx* ctx, const void* depth_frames, int dtype, int width, int height, uint32_t n_frames,
                           const float K[4], const double* T_cam_to_world, float depth_scale, uint64_t* n_added) {
  return guarded(ctx, [&]() -> int {
  if (!ctx) return MA_ERR_INVALID;
  if (ctx->map_cap == 0) return ctx->fail(MA_ERR_STATE, "ma_map_begin first");
  if (!depth_frames || !K || !T_cam_to_world) return ctx->fail(MA_ERR_INVALID, "NULL argument");
  if (dtype != 0 && dtype != 1) return ctx->fail(MA_ERR_INVALID, "dtype must be 0 (float32) or 1 (uint16)");
  if (width <= 0 || height <= 0 || (uint64_t)width * height >= 0xFFFFFFFEull || n_frames == 0 || n_frames > 65535u)
    return ctx->fail(MA_ERR_INVALID, "bad image size / frame count (1..65535 frames)");
  if (!(K[0] != 0.f) || !(K[1] != 0.f) || !(depth_scale > 0.f) || !std::isfinite(depth_scale)) return ctx->fail(MA_ERR_INVALID, "bad intrinsics / depth_scale");
  for (uint64_t i = 0; i < 16ull * n_frames; ++i) if (!std::isfinite(T_cam_to_world[i])) return ctx->fail(MA_ERR_INVALID, "non-finite camera pose");
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t npx = (uint32_t)((uint64_t)width * height);
  const uint64_t total_px = (uint64_t)npx * n_frames;
  if (ctx->map_n + total_px > ctx->map_cap) return ctx->fail(MA_ERR_TOO_LARGE, "map capacity exceeded (ma_map_begin capacity_points)");
  if (!ctx->map_has_origin) {   // points are stored relative to the first camera position (fp32 precision)
    for (int r = 0; r < 3; ++r) ctx->map_origin[r] = std::round(T_cam_to_world[4 * r + 3]);
    ctx->map_has_origin = true;
  }
  { const int rc0 = finish_unproject(ctx); if (rc0) return rc0; }      // an earlier frame's count still pending in the pinned block
  const size_t esz = dtype == 0 ? 4 : 2, frame_bytes = (size_t)npx * esz;
  const uint32_t nb = div_up(npx, kPxPerBlock);
  const size_t n_counts = (size_t)nb * n_frames;
  // the poses as the kernel reads them (make_pinhole's rows, translation minus the map origin), and the counts coming
  // back, through one pinned staging block
  const size_t pose_bytes = (size_t)n_frames * 12 * sizeof(double), stage_bytes = pose_bytes + n_counts * 4;
  if (ctx->batch_stage_cap < stage_bytes) {
    if (ctx->batch_stage) (void)hipHostFree(ctx->batch_stage);
    ctx->batch_stage = nullptr;
    ctx->batch_stage_cap = 0;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->batch_stage), stage_bytes + stage_bytes / 4, hipHostMallocDefault));
    ctx->batch_stage_cap = stage_bytes + stage_bytes / 4;
  }
  double* h_poses = reinterpret_cast<double*>(ctx->batch_stage);
  uint32_t* h_counts = reinterpret_cast<uint32_t*>(ctx->batch_stage + pose_bytes);
  Pinhole cam0{};
  for (uint32_t f = 0; f < n_frames; ++f) {
    const Pinhole cam = make_pinhole(K, T_cam_to_world + 16 * (size_t)f, depth_scale, ctx->map_origin);
    if (f == 0) cam0 = cam;
    for (int k = 0; k < 12; ++k) h_poses[(size_t)f * 12 + k] = cam.T[k];
  }
  HIPCHK(ctx->lift_frames.ensure(frame_bytes * n_frames));
  HIPCHK(ctx->lift_poses.ensure(pose_bytes));
  HIPCHK(ctx->unproj_cnt.ensure(n_counts * 4));
  HIPCHK(hipMemcpyAsync(ctx->lift_frames.p, depth_frames, frame_bytes * n_frames, hipMemcpyDefault, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->lift_poses.p, h_poses, pose_bytes, hipMemcpyHostToDevice, ctx->stream));
  float4* out = ctx->map_pts.as<float4>() + ctx->map_n;
  HIPCHK(hipEventRecord(ctx->evu0, ctx->stream));
  if (dtype == 0)
    hipLaunchKernelGGL((k_unproject<float, true>), dim3(nb, n_frames), dim3(kBlock), 0, ctx->stream, ctx->lift_frames.as<float>(), width, npx,
                       cam0, ctx->lift_poses.as<double>(), ctx->unproj_cnt.as<uint32_t>(), out);
  else
    hipLaunchKernelGGL((k_unproject<uint16_t, true>), dim3(nb, n_frames), dim3(kBlock), 0, ctx->stream, ctx->lift_frames.as<uint16_t>(), width,
                       npx, cam0, ctx->lift_poses.as<double>(), ctx->unproj_cnt.as<uint32_t>(), out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evu1, ctx->stream));
  HIPCHK(hipMemcpyAsync(h_counts, ctx->unproj_cnt.p, n_counts * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->evu0, ctx->evu1));
  ctx->last_op_ms = ms;                  // the one launch (ma_last_device_ms)
  uint64_t all = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    uint64_t t = 0;
    for (uint32_t b = 0; b < nb; ++b) t += h_counts[(size_t)f * nb + b];
    if (n_added) n_added[f] = t;
    all += t;
  }
  ctx->map_n += total_px;
  ctx->map_valid += all;
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
  HIPCHK(sort_pairs_u64(nullptr, tmp_bytes, ctx->map_keys_a.as<unsigned long long>(), ctx->map_keys_b.as<unsigned long long>(),
                        ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), nslots, sort_bits, ctx->stream));
  HIPCHK(ctx->cub_tmp.ensure(std::max(tmp_bytes, scan_tmp_bytes(n + 1))));
  HIPCHK(sort_pairs_u64(ctx->cub_tmp.p, tmp_bytes, ctx->map_keys_a.as<unsigned long long>(), ctx->map_keys_b.as<unsigned long long>(),
                        ctx->vals_a.as<uint32_t>(), ctx->vals_b.as<uint32_t>(), nslots, sort_bits, ctx->stream));
  hipLaunchKernelGGL(k_voxel_heads, dim3(div_up((uint64_t)n + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                     ctx->map_keys_b.as<unsigned long long>(), n, ctx->map_flags.as<uint32_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(scan_exclusive_u32(ctx->cub_tmp.p, ctx->map_flags.as<uint32_t>(), ctx->map_seg.as<uint32_t>(), n + 1, ctx->stream));
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
    source_replaced(ctx);
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

}  // extern "C"
