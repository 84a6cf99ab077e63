/*
 * mapalign.h — C ABI of libmapalign.so, the MI355X (gfx950) map-alignment engine.
 *
 * Drop-in boundary for the map-alignment path of ben-sanati/3d-localisation-and-mapping.
 * The reference has no FFI: its boundary is the Python call
 *     Alignment(base_pose_df, comparison_pose_df, base_bboxes, comparison_bboxes,
 *               visualise).compare(data_folder)                 (task_def.py:175-187)
 * whose body (src/map_alignment/align.py) is absent from the snapshot.  The entry points
 * below are what a ctypes binding for that body's hot loop binds; each cites the row of
 * SURVEY.md §8a / the reference line whose arithmetic it replaces.
 *
 * Conventions
 *   - return 0 (MA_OK) on success, a negative ma_status otherwise; ma_last_error(ctx)
 *     gives the message; no C++ exception crosses this boundary.
 *   - the caller owns every host buffer (row-major, contiguous, borrowed only for the
 *     duration of the call); the library owns every device buffer inside the opaque ctx.
 *   - one ctx = one HIP device + one stream.  A ctx is not thread-safe; different ctxs
 *     may be driven from different host threads.  No global state but the HIP runtime.
 *   - rigid transforms are 4x4 row-major doubles mapping comparison -> gold coordinates,
 *     p' = T @ [p,1], the matrix form consumed by src/map_alignment/visualise.py:94-96,128-130.
 *   - there is NO CPU fallback: without a usable HIP device ma_create fails with
 *     MA_ERR_NO_DEVICE.
 */
#ifndef MAPALIGN_H
#define MAPALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MA_VERSION 500 /* 0.5.0: the 17 sums are exact 64-bit fixed-point integers (any sharding returns the single GPU's bits): ma_sum_bounds_get/_set, ma_sum_scale_get; ma_step_accumulate hands over int64 */
#define MA_ACC_LEN 17  /* W, s_p[3], s_q[3], S[9] row-major (S_ab = sum w p'_a q_b), E */
#define MA_TRACE_STRIDE 20 /* per-iteration trace row: acc[17], rmse, table_words_read, candidates */
#define MA_NO_MATCH 0xFFFFFFFFu
#define MA_FLAG_COUNT_WORK 1   /* also count 4-byte grid-table words read / candidates examined (slower kernel) */
#define MA_FLAG_TIME_KERNELS 2 /* bracket every kernel launch of ma_icp with hipEvents (perturbs the loop) */
#define MA_FLAG_NO_LISTS 4     /* this run: every launch of the loop searches (no candidate lists); results are the same bits */

typedef struct ma_ctx ma_ctx;

typedef enum ma_status {
  MA_OK = 0,
  MA_ERR_INVALID = -1,   /* bad argument (null pointer, non-finite parameter, ...) */
  MA_ERR_NO_DEVICE = -2, /* no HIP device / device index out of range */
  MA_ERR_HIP = -3,       /* a HIP runtime call failed; see ma_last_error */
  MA_ERR_EMPTY = -4,     /* empty cloud / no valid depth pixel */
  MA_ERR_STATE = -5,     /* call order: gold map or source cloud not set */
  MA_ERR_DEGENERATE = -6,/* no correspondence inside the gate (W = 0) or rank-deficient covariance (< 3 non-collinear pairs) */
  MA_ERR_TOO_LARGE = -7, /* gold map >= 2^28 points, source cloud / depth image >= 2^31 points */
  MA_ERR_COMM = -8,      /* RCCL: librccl could not be loaded, or a collective failed; see ma_last_error */
  MA_ERR_NO_MEMORY = -9  /* a host allocation failed inside the call */
} ma_status;

/* Loop parameters (rows a4-a7).  The reference exposes no alignment hyper-parameters
 * (SURVEY.md §5 "Config / flags"), so these are the build's, mirrored by the oracle. */
typedef struct ma_params {
  int32_t max_iter;     /* iteration cap (benchmark: 50) */
  int32_t flags;        /* MA_FLAG_* bits */
  double tol;           /* stop after iteration k>=1 when |rmse_k - rmse_{k-1}| < tol; 0 = never */
  double max_corr_dist; /* correspondence gate in metres (w_i = 1[d_i <= gate]); +inf = ungated */
} ma_params;

typedef struct ma_stats {
  int32_t iterations;   /* iterations actually run */
  int32_t converged;    /* 1 if the tol test fired */
  int32_t degenerate;   /* 1 if an iteration had < 3 correspondences / rank < 2 (loop stopped) */
  int32_t reserved;
  double rmse;          /* sqrt(E/W) of the last iteration run (before its update) */
  double inliers;       /* W of the last iteration run */
  double gpu_ms;        /* device time of the whole loop: first kernel of the run to the end of its last solve (the device's clock) */
} ma_stats;

typedef struct ma_grid_info {
  float cell;           /* cell edge actually used (m) */
  int32_t nx, ny, nz;   /* grid dimensions */
  uint64_t n_cells;
  uint64_t n_occupied;  /* non-empty cells */
  double origin[3];     /* centring origin subtracted before fp32 storage (world frame) */
  double build_ms;      /* device time of the last grid build */
} ma_grid_info;

/* Device allocations made by one context since ma_create (hipMalloc / hipFree calls and bytes).
 * Buffers only grow, so a context that has seen clouds of a given size aligns further clouds
 * of that size without allocating: mapalign.api keeps one context per device for that reason. */
typedef struct ma_mem_info {
  uint64_t device_allocs;
  uint64_t device_frees;
  uint64_t bytes_live;
  uint64_t bytes_peak;
} ma_mem_info;

#define MA_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

int ma_version(void);
const char* ma_status_string(int status);
/* Message of the last failure on ctx (or of the last failed ma_create when ctx == NULL). */
const char* ma_last_error(const ma_ctx* ctx);

/* Lifetime.  `device` is a HIP device ordinal.  A context is one HIP stream plus its device
 * buffers and is not thread-safe; DIFFERENT contexts — also on the same device — may be driven
 * from different host threads at the same time, and doing so pays: one alignment leaves about a
 * third of an MI355X idle, two to four at once raise the iterations/s of the device by 40-60 %
 * (mapalign.align_many does this for a stream of pairs). */
int ma_create(int device, ma_ctx** out);
int ma_destroy(ma_ctx* ctx);
/* external != 0: run on the caller's hipStream_t `hip_stream` (e.g. torch's current
 * stream; NULL is then the legacy default stream, which is what torch hands out for its
 * default stream).  external == 0: return to the ctx's own stream (hip_stream ignored). */
int ma_set_stream(ma_ctx* ctx, void* hip_stream, int external);
int ma_sync(ma_ctx* ctx);

/* Bulk inputs (point arrays, weights, depth frames) may live in HOST memory or in the memory
 * of the context's device (a torch tensor's data_ptr, the output of a decoder on the GPU): the
 * library copies them with hipMemcpyDefault on the context's stream, so a device-resident
 * input never crosses PCIe.  A device input must be complete before the call (or produced on
 * the stream given to ma_set_stream); small arguments (K, poses, boxes, params) are host memory. */

/* Gold ("base") map: upload M x 3 fp32 points and build the uniform grid used by the
 * nearest-neighbour search (row a4; replaces loading the gold .ply and whatever index
 * align.py built — mapping.py:63-68 is the ingest precedent).  cell <= 0 picks the cell
 * edge from the point density.  Invalidates a previously set source cloud. */
int ma_set_gold(ma_ctx* ctx, const float* xyz, uint64_t m, float cell);
int ma_grid_info_get(const ma_ctx* ctx, ma_grid_info* out);
/* The box the gold map's grid covers (world frame) and whether that is the box of the map's BULK only (*bulk_only = 1).
 * A grid is a dense table over a box; a few stray points far from the scene (reflections through windows, sensor noise —
 * real RTAB-Map exports carry them; the reference reads its clouds with o3d.io.read_point_cloud, mapping.py:63-68, which
 * removes NaN / infinite points but no strays) would blow a bounding box up until the table budget forces cells of
 * decimetres or metres, and an alignment then takes 20 to 10 000 times longer.  When the budget binds, the build puts
 * the grid over the central 90 % of the points per axis, widened by a quarter of its width and cut to the true bounds;
 * the points beyond it are stored in the grid's boundary cells and found like any other — results are those of an
 * exhaustive search either way, bit for bit; the box only decides how fast.  The comparison cloud's sort is treated
 * the same way. */
int ma_grid_box_get(const ma_ctx* ctx, double lo[3], double hi[3], int* bulk_only);

/* Comparison ("maintenance") cloud: N x 3 fp32 points, optional per-point weights
 * (NULL = 1; each finite and >= 0, else MA_ERR_INVALID).  Needs the gold map first (points
 * are stored relative to its origin).  Weights scale a correspondence's contribution to the
 * sums (w = 1 is one plain correspondence); they need not sum to anything in particular. */
int ma_set_source(ma_ctx* ctx, const float* xyz, const float* weights, uint64_t n);

/* ma_set_gold + ma_set_source of one pair in ONE call, overlapped: the comparison cloud is uploaded (helper
 * thread, second stream) while the gold map is being indexed.  Same arguments, same errors, bitwise the same
 * state afterwards; what mapalign.align() uses for two host arrays.  std::thread creation can fail with an
 * exception (resource exhaustion): reported as MA_ERR_INVALID like every other C++ exception at this boundary. */
int ma_set_pair(ma_ctx* ctx, const float* gold_xyz, uint64_t m, float cell, const float* comparison_xyz,
                const float* weights, uint64_t n);
/* The same for two fp64 clouds (ma_set_gold_f64 + ma_set_source_f64 with the shift the gold map chose, reported in
 * shift_out): the reference's own arrays are float64 (np.asarray(pcd.points)), so this is the drop-in path. */
int ma_set_pair_f64(ma_ctx* ctx, const double* gold_xyz, uint64_t m, float cell, double shift_out[3],
                    const double* comparison_xyz, const float* weights, uint64_t n);

/* The same for fp64 clouds — what Open3D and NumPy hand over by default (the reference's
 * arrays are np.asarray(pcd.points), float64).  Upload, bounds and the conversion to the engine's
 * fp32 frame happen on the device; the host never walks the arrays.  Georeferenced coordinates
 * (any |coordinate| above MA_LARGE_COORDINATE_M) would lose centimetres in fp32, so
 * ma_set_gold_f64 then takes an integer-metre shift (the box centre, rounded half-to-even) out in
 * fp64 first and reports it in shift_out (zeros otherwise): the context works in the shifted
 * frame, i.e. every transform it returns is T_local with T_world = S T_local S^-1, S = translation
 * by the shift — the caller passes the same shift to ma_set_source_f64 (NULL = zeros) and
 * conjugates results (mapalign/engine.py does). */
#define MA_LARGE_COORDINATE_M 1000.0
int ma_set_gold_f64(ma_ctx* ctx, const double* xyz, uint64_t m, float cell, double shift_out[3]);
int ma_set_source_f64(ma_ctx* ctx, const double* xyz, const float* weights, uint64_t n, const double shift[3]);
int ma_mem_info_get(const ma_ctx* ctx, ma_mem_info* out);
/* Launch geometry of the search kernel (k_nn_accum) for the current source cloud: workgroups per launch
 * (padded to whole XCD segments and scheduling groups) and threads (= points) per workgroup.  What a roofline
 * computation needs for the per-workgroup terms (19 fp64 partials each). */
int ma_search_geometry(const ma_ctx* ctx, uint32_t* n_workgroups, uint32_t* block_threads);
/* Candidate lists of the last ma_icp / ma_icp_sharded run (rows a4-a7): out = {launches that searched and
 * wrote the per-query candidate lists, launches that read them instead of searching}.  Both 0 when the loop never came
 * to rest enough for lists to pay, in a run of <= 12 iterations, with MA_FLAG_NO_LISTS, or on a context whose lists are
 * switched off (ma_list_config).  Results do not depend on them: a list launch proves per query that the nearest
 * neighbour is on its list (triangle inequality over the motion since the list was written) and searches where it
 * cannot, and every path resolves equal fp32 distances the same way (the smaller position in the gold map's cell order),
 * so lists on and off give the same sums in every iteration and the same transform, bit for bit. */
int ma_get_list_stats(ma_ctx* ctx, int32_t out[2]);
/* What each iteration of the last ma_icp run launched: 0 a search, 1 a search that also wrote the candidate lists,
 * 2 a list launch.  Returns the number of entries written (<= max_iterations; launches enqueued, which for a loop that
 * stopped early may exceed the iterations performed) or a negative status. */
int ma_get_list_plan(ma_ctx* ctx, uint8_t* kinds, int max_iterations);
/* Per-context configuration of the candidate lists.  enabled: 1 / 0, or -1 to keep; skin_div: the lists' skin is the
 * grid's cell edge / skin_div (2..64), 0 to keep; min_uses: a launch writes lists only if at least this many reads are
 * predicted to follow (1..64), 0 to keep.  A context is created with lists enabled, cell / 10 and 3 — or with what
 * MAPALIGN_LISTS (0 = off), MAPALIGN_SKIN_DIV and MAPALIGN_MIN_USES say in the environment AT ma_create; nothing is read
 * from the environment afterwards.  The settings change how fast the loop runs, never what it returns. */
int ma_list_config(ma_ctx* ctx, int enabled, float skin_div, int min_uses);
/* Diagnostics of the host-side planner of the last ma_icp / ma_icp_sharded run (the thread that decides, every 4
 * iterations, which kernel the next ones launch, from the states the solves leave in pinned host memory):
 * out = {waits for a state, waits that outlasted the ~20 us spin and went on yielding the core, waits that TIMED OUT,
 * iterations enqueued}.  A timeout (20 s without the state of an iteration enqueued two batches earlier) is not an
 * error — the run goes on searching, drains the stream before every further batch and returns the same bits — but it
 * means posted writes to pinned memory are not arriving, and it is reported here rather than passed over. */
int ma_get_planner_diag(ma_ctx* ctx, int32_t out[4]);
/* Row N1 (coarse pre-alignment): number of valid points, centroid (frame of the coordinates handed in, minus the
 * large-coordinate shift if one is in force) and 3x3 sample covariance (row-major) of a cloud AS STORED on the
 * device: which = 0 the gold map, 1 the source cloud (holes of a depth frame are skipped).  out = {n, c[3], C[9]}.
 * The principal axes of both clouds give the (R, centre) candidates of the first stage of the transformations
 * list (src/map_alignment/visualise.py:82-92); the sums are one pass over data that is in HBM anyway. */
int ma_cloud_moments(ma_ctx* ctx, int which, double out[13]);

/* Row a8: fused depth -> 3-D unprojection written straight into the source-cloud device
 * buffer.  Pinhole model of pose_processor.py:133-137,184-196,242-260:
 *   Z = depth[y][x] / depth_scale;  X = (x-cx) Z / fx;  Y = (y-cy) Z / fy;
 *   g = (T_cam_to_world @ [X,Y,Z,1])[:3]
 * K = {fx, fy, cx, cy} already rescaled to the depth resolution.  dtype: 0 = float32
 * (the reference's format, dataset.py:68-81: float32 millimetres, depth_scale = 1000),
 * 1 = uint16.  Pixels with depth <= 0 / NaN / inf are dropped: *n_valid receives the number of
 * valid pixels, and everything that hands points or per-point results back (ma_get_source,
 * ma_get_source_stored, ma_nn) reports exactly those, in row-major pixel order.  (Inside, every
 * pixel keeps its slot and an invalid one is a hole the search skips; the compaction happens
 * only where points are handed back.)
 * n_valid == NULL: the call only ENQUEUES the upload and the kernel and returns without waiting for
 * the device — an alignment can follow on the same stream at once (ma_icp needs no count).  The
 * count is then settled by the first call that needs it (ma_get_source(ctx, NULL, &n) returns it),
 * and an image without a single valid pixel is reported there (MA_ERR_EMPTY) or by ma_icp
 * (MA_ERR_DEGENERATE: nothing to match).
 * Lifetime of the depth buffer.  With n_valid != NULL the call waits for the device, so the buffer is free when it
 * returns.  With n_valid == NULL the copy into the context is only ENQUEUED (hipMemcpyAsync on the ctx stream): a frame in
 * DEVICE memory or in PINNED host memory must stay valid and unmodified until the next call on this context that waits
 * for the stream (ma_sync, ma_icp, ma_get_source, ...), and a producer on another stream must not overwrite it before
 * then; a frame in pageable host memory is staged by the runtime before hipMemcpyAsync returns, as for every pageable
 * source (mapalign/engine.py keeps a temporary it made alive until that next wait either way). */
int ma_set_source_from_depth(ma_ctx* ctx, const void* depth, int dtype, int width, int height,
                             const float K[4], const double T_cam_to_world[16],
                             float depth_scale, uint64_t* n_valid);
/* Device time (ms, hipEvents on the ctx stream) of the last ma_set_source_from_depth /
 * ma_map_add_depth: the k_unproject launch (one streaming pass, pixel i -> slot i). */
int ma_last_device_ms(const ma_ctx* ctx, double* ms);
/* Row N4: full-frame map builder — what RTAB-Map's .ply export does for the
 * reference, on the device: every depth frame is unprojected with its pose (same pinhole
 * model as ma_set_source_from_depth, pose_processor.py:88-122,262-271) into one buffer, and
 * ma_map_finish reduces the points of each voxel of edge `voxel` to their centroid (the
 * semantics of Open3D's voxel_down_sample; deterministic: stable sort by voxel key, fixed
 * summation order).  as_gold != 0 installs the result as the gold map of this ctx without a
 * host round trip (cell as in ma_set_gold).  xyz_out may be NULL; *n: in = capacity of
 * xyz_out in points, out = number of voxels. */
int ma_map_begin(ma_ctx* ctx, float voxel, uint64_t capacity_points);
int ma_map_add_depth(ma_ctx* ctx, const void* depth, int dtype, int width, int height, const float K[4],
                     const double T_cam_to_world[16], float depth_scale, uint64_t* n_added);
/* The same for a run of n_frames frames of one size in ONE call (the reference walks its frames one by one,
 * pose_processor.py:88-122): frames stored back to back, n_frames x 16 doubles of poses, one intrinsics set; one upload,
 * one kernel launch over all frames, one synchronisation.  n_added (optional): n_frames counts of valid pixels.  The map
 * afterwards holds exactly — bit for bit — what n_frames ma_map_add_depth calls in the same order would have left. */
int ma_map_add_depth_batch(ma_ctx* ctx, const void* depth_frames, int dtype, int width, int height, uint32_t n_frames,
                           const float K[4], const double* T_cam_to_world, float depth_scale, uint64_t* n_added);
int ma_map_finish(ma_ctx* ctx, int as_gold, float cell, float* xyz_out, uint64_t* n);

/* Row a10: lift 2-D box corners to 3-D global coordinates (pose_processor.py:172-208):
 * for each of the n_boxes boxes [x1,y1,x2,y2] (pixels AT DEPTH RESOLUTION, i.e. after the
 * reference's scale_bounding_box), the four corners (x1,y1) (x2,y1) (x2,y2) (x1,y2) are
 * truncated to integer pixels, clamped into the image, unprojected with the pinhole model
 * of ma_set_source_from_depth (_depth_to_3d call shape, pose_processor.py:184-196) and
 * moved to the global frame (_transform_to_global, pose_processor.py:242-260).
 * depth == NULL reuses the frame of the last ma_set_source_from_depth / ma_lift_boxes call.
 * corners_out: n_boxes x 4 x 3 doubles; valid_out (optional): n_boxes x 4 bytes, 0 where the
 * depth used for the corner is <= 0 / NaN / inf (the corner is then NaN).
 *
 * z_rule — which depth a corner gets.  The reference's two statements disagree and the body that
 * would settle it (Transforms._depth_to_3d) is absent from the snapshot:
 *   MA_Z_CORNER (0)  the depth AT the corner pixel — what the call does: _depth_to_3d(int(x), int(y),
 *                    depth_image_cv, fx, fy, cx, cy, scale_depth) is handed one pixel and no box
 *                    (pose_processor.py:184-196);
 *   MA_Z_MEDIAN (1)  what the comment above that call says, "z-values from median over bbox (x, y)
 *                    range" (pose_processor.py:183): the median of the valid depths over the box's pixel
 *                    rectangle [int(x1)..int(x2)] x [int(y1)..int(y2)] (clamped into the image), the
 *                    SAME z for the four corners, each still unprojected through its own pixel (u, v);
 *                    an even count averages the two middle values (numpy.median); a box without a
 *                    valid pixel gives four invalid corners.
 * An integrator picks the one their Transforms implemented; everything else of the row is identical. */
#define MA_Z_CORNER 0
#define MA_Z_MEDIAN 1
int ma_lift_boxes(ma_ctx* ctx, const void* depth, int dtype, int width, int height, const float K[4],
                  const double T_cam_to_world[16], float depth_scale, const float* boxes_xyxy,
                  uint64_t n_boxes, int z_rule, double* corners_out, uint8_t* valid_out);

/* Row a10 for a whole detector batch (BASELINE configs[3]: batch-32 inference, detector.py:82-90,
 * then pose_processor.py:172-208 per frame): n_frames depth images of the same size stored back
 * to back, one pose per frame (n_frames x 16 doubles), one intrinsics set, and n_boxes boxes
 * [x1,y1,x2,y2] (depth-resolution pixels) with box_frame[b] = index of the frame box b was
 * detected in.  One upload, ONE kernel launch (two with MA_Z_MEDIAN) and one synchronisation for the
 * whole batch; same corner order, truncation, clamping, z_rule and validity rule as ma_lift_boxes. */
int ma_lift_boxes_batch(ma_ctx* ctx, const void* depth_frames, int dtype, int width, int height,
                        uint32_t n_frames, const float K[4], const double* T_cam_to_world,
                        float depth_scale, const float* boxes_xyxy, const uint32_t* box_frame,
                        uint64_t n_boxes, int z_rule, double* corners_out, uint8_t* valid_out);

/* Copy the current source cloud back (world frame, storage order); *n in = capacity in
 * points, out = points written.  For parity tests of the unprojection kernel. */
int ma_get_source(ma_ctx* ctx, float* xyz_out, uint64_t* n);

/* The clouds AS STORED — what the nearest-neighbour search (row a4) actually compares: every coordinate
 * is kept as fp32 relative to the centring origin of the gold map, so the stored point is
 * world = fp32(x - origin) + origin, returned here in fp64 (exact) and in the CALLER's order (the
 * order of the arrays given to ma_set_gold / ma_set_source; pixel order of the valid pixels for a cloud
 * made from a depth frame).  An oracle run on these arrays sees bit for bit the device's inputs, so its
 * nearest-neighbour indices must equal ma_nn's except between candidates whose fp32 distances tie
 * (tests/test_gpu_parity.py: assert_nn_exact_on_stored).  xyz_out == NULL only queries the count;
 * *n: in = capacity in points, out = points written. */
int ma_get_gold(ma_ctx* ctx, double* xyz_out, uint64_t* n);
int ma_get_source_stored(ma_ctx* ctx, double* xyz_out, uint64_t* n);

/* Rows a4-a7: the whole ICP loop on the device.  init = NULL means identity.
 * T_out receives the composed comparison->gold transform.
 * A run of more than 12 iterations lets the launches of its later iterations read per-query candidate lists instead
 * of searching once the cloud has come to rest (DESIGN.md 5 "List launches"; ma_get_list_stats / ma_get_list_plan
 * report them).  Every correspondence taken from a list is proved, and the result is the searching loop's bit for bit
 * — with tol > 0 as well: such a run stops at the searching loop's iteration, and at most 8 launches that only find
 * the loop stopped are enqueued past it.  MA_FLAG_NO_LISTS makes every launch of one run search; ma_list_config sets
 * the context's default. */
int ma_icp(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16],
           ma_stats* stats);
/* Per-iteration trace of the last ma_icp / stepping sequence: row k =
 * {acc[17] in WORLD coordinates, rmse_k, table_words_read_k, candidates_k}; the last two are
 * zero unless MA_FLAG_COUNT_WORK.  Returns the number of rows written (<= max_rows) or a
 * negative status. */
int ma_get_trace(ma_ctx* ctx, double* out, int max_rows);

/* Per-kernel device time of the last ma_icp run with MA_FLAG_TIME_KERNELS:
 * out = {nn_accum total ms, nn_accum launches, reduce_solve total ms, reduce_solve launches}. */
int ma_get_kernel_times(ma_ctx* ctx, double out[4]);
/* The same run, per iteration: out[2k] = ms of iteration k's search launch, out[2k+1] = ms of its
 * reduce + solve launch.  Returns the number of iterations written (<= max_iterations). */
int ma_get_iteration_times(ma_ctx* ctx, float* out, int max_iterations);

/* One iteration at transform T_in (NULL = identity): accumulators (world frame) and the
 * updated transform.  For parity tests (rows a4-a6). */
int ma_icp_step(ma_ctx* ctx, const ma_params* params, const double T_in[16],
                double acc_out[MA_ACC_LEN], double T_out[16]);

/* Row a4 alone: nearest gold point of every source point under T (NULL = identity), in
 * the ORIGINAL order of both clouds; unmatched (d > gate) -> MA_NO_MATCH / +inf. */
int ma_nn(ma_ctx* ctx, const double T[16], double max_corr_dist, uint32_t* idx_out,
          float* d2_out);

/* Exact sums (row a5).  Every correspondence's 17 terms are rounded ONCE, each to a multiple of its accumulator's quantum
 * 2^-e, and summed as 64-bit integers from there on (workgroup, 19-way sum, all-reduce): integer addition is associative,
 * so the sums — and with them every transform — do not depend on how the source cloud is cut into workgroups, launches,
 * shards or ranks, nor on the order in which a collective adds.  ANY partition of the comparison cloud over ANY number of
 * contexts / GPUs returns the bits ma_icp returns for the whole cloud on one, PROVIDED the contexts derive the same
 * exponents e, i.e. agree on what bounds a term: the gold map's box (the same everywhere by construction), the gate (a
 * parameter), and three numbers of the comparison cloud —
 *     {n_total: points of the WHOLE cloud, wmax: its largest weight (1 without weights),
 *      rs: a radius of the cloud about the gold map's centring origin — sqrt(3) times its largest |coordinate| there, so that
 *          the largest rs over the shards of a cloud IS the whole cloud's (only ungated runs use it)}.
 * ma_sum_bounds_get returns this context's own three (for its shard); ma_sum_bounds_set installs the whole cloud's:
 * {sum of the n, max of the wmax, max of the rs} over the shards — which is exactly what a single context holding the whole
 * cloud derives for itself.  ma_icp_sharded does this exchange itself — one small all-gather in the first call after a
 * context's source cloud (or communicator) changed, so the ranks of a group must replace their shards TOGETHER: a rank
 * that alone installed a new shard would enter that collective alone.  Callers of the split stepping below do the
 * exchange with their own collective.  The bounds belong to a source cloud: installing a new one
 * (or a new gold map) clears them; NULL clears them explicitly.  Without agreed bounds results are still correct to
 * rounding (1e-12 relative), just not bit-identical across different partitions.
 * ma_sum_scale_get: the exponents e = {W, s_p, s_q, S, E} in force for a run with these parameters (MA_ACC_LEN int64 v_k
 * mean v_k 2^-e[group(k)], groups: k = 0 | 1-3 | 4-6 | 7-15 | 16). */
int ma_sum_bounds_get(ma_ctx* ctx, double out[3]);
int ma_sum_bounds_set(ma_ctx* ctx, const double in[3] /* NULL: clear */);
int ma_sum_scale_get(ma_ctx* ctx, const ma_params* params, int32_t out[5]);

/* Split stepping for the sharded-source multi-GPU mode: every rank holds
 * the whole gold map and a shard of the source cloud; per iteration
 *   ma_step_accumulate -> all-reduce(sum) of MA_ACC_LEN 64-bit INTEGERS -> ma_step_solve.
 * acc_dev is a DEVICE pointer to MA_ACC_LEN int64 owned by the caller (e.g. a torch int64
 * tensor's data_ptr): the fixed-point sums described above, in the ctx's centred frame, which is
 * identical on all ranks because it derives from the gold map alone.  Sum them as integers
 * (any order) and hand the totals to ma_step_solve of every rank.  Integers from different contexts may only be added
 * if all of them were scaled alike, so both calls REQUIRE agreed bounds (ma_sum_bounds_set, above; MA_ERR_STATE without —
 * a lone context installs its own ma_sum_bounds_get).  Nothing here synchronises
 * with the host (an ungated run on a cloud made from a depth frame computes that cloud's box
 * once, in ma_state_set).
 * Call order: ma_set_gold, ma_set_source (this rank's shard), ma_sum_bounds_set, ma_state_set, then the
 * accumulate / solve pairs.  Installing another gold map or source cloud (ma_set_gold,
 * ma_set_source, ma_set_source_from_depth, ma_map_finish(as_gold)) invalidates the stepping
 * state — its buffers are sized for the clouds it was set for — and ma_step_accumulate /
 * ma_step_solve then return MA_ERR_STATE until ma_state_set is called again. */
int ma_state_set(ma_ctx* ctx, const double T[16]);
int ma_state_get(ma_ctx* ctx, double T_out[16], ma_stats* stats);
int ma_step_accumulate(ma_ctx* ctx, const ma_params* params, void* acc_dev);
int ma_step_solve(ma_ctx* ctx, const ma_params* params, const void* acc_dev);

/* The same sharded-source loop with the collective INSIDE the library (SURVEY.md §8e,
 * north_star "RCCL-over-xGMI all-reduce only of the per-pair accumulators"): librccl is
 * loaded with dlopen on first use and, per iteration, the library enqueues on the ctx stream
 *   search+accumulate (local shard) -> 19-way sum -> ncclAllReduce(sum, 17 x int64: exact, order-free) -> solve
 * with the loop in C and no host synchronisation — the loop of ma_icp: the same candidate lists (each rank plans
 * its own shard's; which kernel a rank launches changes no bit of its sums), the same early exit with tol > 0 (decided
 * from the iteration at which the loop stopped, so that every rank issues the same number of collectives), the final
 * state read from pinned host memory.  The early exit needs the pinned slots: runs of 13 ... 4096 iterations have it (a
 * stopped loop enqueues at most 8 launches past its stop, the same number on every rank whatever their list settings);
 * a sharded run of <= 12 or > 4096 iterations enqueues all max_iter iterations — correct (launches after the stop do
 * nothing and their collectives carry zeros), just not short.  Bootstrap: rank 0 obtains an id with
 * ma_comm_unique_id and hands the MA_COMM_ID_BYTES bytes to the other ranks by any channel
 * (torch.distributed broadcast, a file, MPI ...); every rank then calls ma_comm_init
 * (collective).  A 1-rank communicator is valid and makes ma_icp_sharded equal ma_icp bit
 * for bit.  ma_comm_allgather_f64 / ma_comm_allreduce_f64 (op 0 = sum, 1 = max) move small
 * HOST vectors through the same communicator (pair-per-GPU result gather, fleet statistics). */
int ma_comm_unique_id(void* id_out /* MA_COMM_ID_BYTES */);
/* The same sharded loop WITHOUT a library collective on the data path (row e; north_star "all-reduce only of the per-pair
 * 3x3 covariance/centroid accumulators"): the all-reduce of 17 integers is done by the iteration's second kernel itself.
 * Every rank owns a small array of rows in its device memory, mapped into every peer; the wave that has summed a rank's 17
 * sums writes them, as self-validating tagged 8-byte words, straight into its row in EVERY rank's array (over xGMI between
 * GPUs), waits until the rows of all ranks have arrived in its own array, adds them (integers: any order, same bits) and
 * solves — two launches per iteration, as in ma_icp, where the RCCL path has three and a library call.  The wait is
 * bounded (2 s, MAPALIGN_P2P_TIMEOUT_MS at ma_create): a rank whose peers stay silent stops its loop and ma_icp_sharded
 * returns MA_ERR_COMM — never a hang; the exchange must then be set up again.
 * Bootstrap: every rank calls ma_p2p_export (allocates and zeroes its rows, fills MA_P2P_BLOB_BYTES opaque bytes), the
 * blobs travel to all ranks by any channel IN RANK ORDER (torch.distributed all_gather, a file, MPI ...: that exchange is
 * also the barrier after which rows may be written), every rank calls ma_p2p_init with all of them — one export per
 * init: a second ma_p2p_init on rows an earlier set-up has used returns MA_ERR_STATE until they are exported again.  Ranks may be
 * processes (the rows are mapped with hipIpcOpenMemHandle) or contexts of one process (pointers are used directly; on
 * different devices peer access is enabled); at most 16.  An exchange and an RCCL communicator may exist side by side:
 * ma_comm_config chooses which one ma_icp_sharded uses (MA_TRANSPORT_RCCL, the default while both exist, or
 * MA_TRANSPORT_P2P; with only one of them set up, that one).  Results are the same bits either way, and the single GPU's. */
#define MA_P2P_BLOB_BYTES 128
#define MA_TRANSPORT_RCCL 0
#define MA_TRANSPORT_P2P 1
int ma_p2p_export(ma_ctx* ctx, void* blob_out /* MA_P2P_BLOB_BYTES */);
int ma_p2p_init(ma_ctx* ctx, const void* blobs /* n_ranks x MA_P2P_BLOB_BYTES, rank order */, int n_ranks, int rank);
int ma_p2p_destroy(ma_ctx* ctx);
int ma_comm_config(ma_ctx* ctx, int transport);
int ma_comm_init(ma_ctx* ctx, const void* id, int n_ranks, int rank);
int ma_comm_destroy(ma_ctx* ctx);
int ma_icp_sharded(ma_ctx* ctx, const ma_params* params, const double init[16], double T_out[16],
                   ma_stats* stats);
int ma_comm_allgather_f64(ma_ctx* ctx, const double* in, uint64_t count, double* out /* count * n_ranks */);
int ma_comm_allreduce_f64(ma_ctx* ctx, double* inout, uint64_t count, int op);

#ifdef __cplusplus
}
#endif
#endif /* MAPALIGN_H */
