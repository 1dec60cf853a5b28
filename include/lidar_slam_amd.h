/*
 * lidar_slam_amd.h — C-ABI of the MI355X-native scan-matching front end.
 *
 * Drop-in boundary for ONE hot path of FISH1129/Lidar-SLAM: the lidar_localization front_end
 * per-scan pipeline  VoxelFilter downsample -> NDT / ICP scan-to-local-map.  Every entry point
 * below names the reference interface it replaces (paths relative to the reference root):
 *
 *   CloudFilterInterface::Filter        lidar_localization/include/lidar_localization/models/cloud_filter/cloud_filter_interface.hpp:13-18
 *     VoxelFilter (pcl::VoxelGrid)      lidar_localization/src/models/cloud_filter/voxel_filter.cpp:12-41
 *   RegistrationInterface               lidar_localization/include/lidar_localization/models/registration/registration_interface.hpp:14-24
 *     NDTRegistrationManual             lidar_localization/src/models/registration/ndt_registration_manual.cpp:12-78
 *     ICPRegistrationManual             lidar_localization/src/models/registration/icp_registration_manual.cpp:12-179
 *   CloudData::POINT = pcl::PointXYZI   lidar_localization/include/lidar_localization/sensor_data/cloud_data.hpp:33-38
 *
 * The library is plain C at the boundary: opaque handles, raw pointers + sizes, no PCL / Eigen /
 * boost / yaml / torch types.  Host clouds are passed as (pointer, count, stride, intensity offset) so a
 * `cloud->points.data()` of pcl::PointXYZI (stride 32, xyz at 0, intensity at 16) can be handed
 * over as is; 4x4 poses are 16 floats COLUMN-major (Eigen::Matrix4f::data()).
 * `intensity` carries the reference's per-point static weight (front_end.cpp:303-327).
 *
 * Every function returns LSA_OK (0) or a negative lsa_status; nothing throws across the ABI.  There
 * is NO CPU fallback: without a usable gfx950 device every compute entry point fails with
 * LSA_ERR_NO_DEVICE.  The reference's plugins return `true` unconditionally; INTEGRATION.md shows the
 * adaptor mapping non-zero -> false + LOG(ERROR).
 *
 * Threading: like the reference (single-threaded callers, front_end_node.cpp:27-34) a handle is
 * not re-entrant; different handles may be used from different threads.
 */
#ifndef LIDAR_SLAM_AMD_H_
#define LIDAR_SLAM_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSA_VERSION_MAJOR 0
#define LSA_VERSION_MINOR 1

typedef enum {
  LSA_OK = 0,
  LSA_ERR_INVALID = -1,    /* bad argument (null pointer, zero size where one is required, ...) */
  LSA_ERR_NO_DEVICE = -2,  /* no gfx950 device / HIP runtime cannot initialise — there is no CPU path */
  LSA_ERR_HIP = -3,        /* a HIP call failed; lsa_last_error() has the text */
  LSA_ERR_NO_TARGET = -4,  /* ScanMatch / GetFitnessScore before SetInputTarget */
  LSA_ERR_CAPACITY = -5,   /* caller-provided output buffer too small */
  LSA_ERR_COLLECTIVE = -6, /* the all-reduce callback reported failure */
  LSA_ERR_RANGE = -7       /* voxel index would overflow int32 (PCL: "Leaf size is too small") */
} lsa_status;

/* pcl::PointXYZI layout constants (cloud_data.hpp:35) */
#define LSA_POINTXYZI_STRIDE 32
#define LSA_POINTXYZI_W_OFF 16

/* ---------------------------------------------------------------------------------------------
 * Context: one per process per GPU (one process per GPU; see INTEGRATION.md for the multi-GPU layout)
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_context lsa_context;

/* device: HIP ordinal. hip_stream: a hipStream_t to enqueue on, or NULL to create a private stream. */
int lsa_context_create(lsa_context **out, int device, void *hip_stream);
void lsa_context_destroy(lsa_context *ctx);
/* text of the last failure on this context (valid until the next call on it); never NULL */
const char *lsa_last_error(const lsa_context *ctx);
/* text of the last failure of a *_create call that returned no handle (thread-local) */
const char *lsa_last_global_error(void);
void *lsa_context_stream(lsa_context *ctx);
int lsa_context_synchronize(lsa_context *ctx);
int lsa_version(void);

/* ---------------------------------------------------------------------------------------------
 * VoxelFilter  — replaces VoxelFilter::Filter (voxel_filter.cpp:36-41 -> pcl::VoxelGrid::filter)
 * Semantics (PCL 1.7 VoxelGrid, downsample_all_data = true, min_points_per_voxel = 0):
 * one output point per occupied voxel = mean of x,y,z,intensity of its members, outputs ordered by
 * ascending linear voxel index i + j*dx + k*dx*dy.
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_voxel_filter lsa_voxel_filter;

/* VoxelFilter::VoxelFilter(float leaf_size_x, y, z)  (voxel_filter.cpp:20-22) */
int lsa_voxel_filter_create(lsa_voxel_filter **out, lsa_context *ctx, float leaf_x, float leaf_y, float leaf_z);
void lsa_voxel_filter_destroy(lsa_voxel_filter *f);
/* Optional, no reference counterpart (PCL allocates per call): allocate now what filtering clouds of up to max_points needs,
 * so that the first Filter calls neither allocate nor have to grow the occupancy bitmap; async_too: bit 0 also starts the helper
 * thread of lsa_voxel_filter_apply_device_async, bit 1 prepares the host-cloud path (pinned staging rows, second packing thread).
 * Changes no result. */
int lsa_voxel_filter_reserve(lsa_voxel_filter *f, size_t max_points, int async_too);
/* Diagnostics: out[0] synchronous builds, [1] builds repeated because a point fell outside the reused cell box, [2] bitmap growths,
 * [3] host clouds whose rows the device pulled over the link itself while they were being packed (lsa_voxel_filter_apply, from the
 * second scan of a stream on; environment LSA_FILTER_PULL=0 selects the copy-command path). */
int lsa_voxel_filter_counters(lsa_voxel_filter *f, uint64_t out[4]);
/* Diagnostics: builds of a host cloud whose first kernel gave up waiting for the rows the host was packing (2 s; the packing threads
 * were held up that long).  lsa_voxel_filter_apply fails with LSA_ERR_HIP in that case; the asynchronous host-cloud paths
 * (lsa_front_end_update_batch, lsa_pipeline_scan_match_host) send the rows again once they are packed and filter them the ordinary
 * way, so their results never come from a half-delivered cloud.  Counted here either way. */
int lsa_voxel_filter_pull_timeouts(lsa_voxel_filter *f, uint64_t *out);
/* Diagnostics (profiles/scripts/vi_stamps.py): with out == NULL, enable != 0 makes the mark, the accumulate and the scan kernel of every
 * following build note the device's 100 MHz wall clock at their phase boundaries (8 values per workgroup, thread 0; 0 switches it off);
 * with out != NULL the 3 x *per_kernel values of the LAST build are copied out (kernel k's workgroup w at out[k * *per_kernel + 8 w ...]:
 * entry, table cleared, points done, barrier passed, [accumulate: ranks known,] flush issued, flush acknowledged; scan: entry, grid
 * definition read, bits in front of the tile counted, own tile counted, prefix known, prefix written, cells of the ranks issued,
 * acknowledged) and cleared. */
int lsa_voxel_filter_stamps(lsa_voxel_filter *f, int enable, uint64_t *out, size_t cap, size_t *per_kernel);

/* Filter(input_cloud_ptr, filtered_cloud_ptr): host AoS in, host AoS out.
 * out may alias in (viewer.cpp:207 calls Filter with in == out); out_cap in points (n is always
 * enough).  Output points get data[3] = 1.0f like a default-constructed PointXYZI when out_stride >= 16.
 * Returns LSA_ERR_RANGE (and copies the input through, as PCL does) if dx*dy*dz > INT32_MAX. */
int lsa_voxel_filter_apply(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off,
                           void *out_pts, size_t out_cap, size_t out_stride, size_t out_w_off, size_t *n_out);

/* Device-resident variant: d_in / d_out are device pointers to packed float4 {x,y,z,intensity}
 * (16 B / point, 16-B aligned).  d_out must hold n points.  *n_out is written on the host (one sync). */
int lsa_voxel_filter_apply_device(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out);
/* The same call split in two so that a caller can overlap the filter of scan k+1 (on this filter's context / HIP stream)
 * with the registration of scan k (on another context): _async enqueues the whole pipeline and returns at once, _wait
 * blocks until the output is complete and returns the voxel count.  One call in flight per handle; d_in / d_out must stay
 * valid and untouched until _wait returns. */
int lsa_voxel_filter_apply_device_async(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw);
int lsa_voxel_filter_wait(lsa_voxel_filter *f, size_t *n_out);

/* ---------------------------------------------------------------------------------------------
 * BoxFilter — replaces BoxFilter (box_filter.cpp:14-73 -> pcl::CropBox): keeps, in input order, the points with
 * edge_min <= p <= edge_max on every axis, edge = box_filter_size + origin (:62-68).  Not behind the filter string
 * switch in the reference; Matching::ResetLocalMap (matching.cpp:117-131) and the viewer call it directly.
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_box_filter lsa_box_filter;
/* BoxFilter(YAML::Node node): node["box_filter_size"] = [min_x, max_x, min_y, max_y, min_z, max_z] (box_filter.cpp:14-25) */
int lsa_box_filter_create(lsa_box_filter **out, lsa_context *ctx, const float box_filter_size[6]);
void lsa_box_filter_destroy(lsa_box_filter *f);
int lsa_box_filter_set_size(lsa_box_filter *f, const float box_filter_size[6]);   /* SetSize   (:39-54) */
int lsa_box_filter_set_origin(lsa_box_filter *f, const float origin[3]);          /* SetOrigin (:56-60) */
int lsa_box_filter_get_edge(lsa_box_filter *f, float edge[6]);                    /* GetEdge   (:70-73) */
/* Filter(input_cloud_ptr, output_cloud_ptr) (:27-37). Same buffer conventions as lsa_voxel_filter_apply; out_pts may
 * alias in_pts. */
int lsa_box_filter_apply(lsa_box_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off,
                         void *out_pts, size_t out_cap, size_t out_stride, size_t out_w_off, size_t *n_out);
/* device-resident variant: packed float4 {x,y,z,intensity}; d_out must hold n points and must not alias d_in */
int lsa_box_filter_apply_device(lsa_box_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out);

/* ---------------------------------------------------------------------------------------------
 * NDT registration — replaces NDTRegistrationManual (ndt_registration_manual.cpp) and, behind it,
 * lidar_localization::NormalDistributionsTransform / VoxelGrid (ndt_registration_manual/ *.cpp)
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_ndt lsa_ndt;

typedef struct {
  int32_t iterations;       /* Newton iterations (nr_iterations_, NormalDistributionsTransform.cpp:383) */
  int32_t ls_iterations;    /* More-Thuente trial steps taken in this call (:746) */
  int32_t passes;           /* derivative / Hessian passes over the source cloud */
  int32_t converged;        /* hasConverged() */
  double score;             /* last NDT score */
  double trans_probability; /* getTransformationProbability(), score / #source points (:386-388) */
  double pairs_per_pass;    /* mean number of (source point, target voxel) pairs per pass = K-bar * M */
  double final_p[6];        /* final (x,y,z,roll,pitch,yaw) in the reference's Euler chart */
  double gpu_ms;            /* device time of the passes (HIP events on the context's stream) */
} lsa_ndt_stats;

/* NDTRegistrationManual(float res, float step_size, float trans_eps, int max_iter, float outlier_ratio)
 * (ndt_registration_manual.cpp:25-29); YAML defaults config/mapping/front_end.yaml:24-29 */
int lsa_ndt_create(lsa_ndt **out, lsa_context *ctx, float res, float step_size, float trans_eps, int max_iter,
                   float outlier_ratio);
void lsa_ndt_destroy(lsa_ndt *ndt);

/* Which of the reference's two NDT plugins the handle stands in for (call before SetInputTarget):
 *   LSA_NDT_MODE_MANUAL (default): "NDTMANUAL", the in-tree arithmetic with all its quirks (covariance seeded with I,
 *     hand-rolled eigen-solver, static weight, `phi_t -= score`), GetFitnessScore = mean distance to the nearest voxel
 *     centroid (NormalDistributionsTransform.cpp:940-965).
 *   LSA_NDT_MODE_PCL: "NDT" = NDTRegistration(res, step_size, trans_eps, max_iter) (ndt_registration.cpp:18-66) ->
 *     pcl::NormalDistributionsTransform of third-party PCL 1.7 (not in the tree; parity UNPINNED): covariance from zero,
 *     accurate symmetric eigen-solver, rejected leaves keep a zero inverse covariance, no static weight, the
 *     More-Thuente trial loop assigns phi_t, outlier_ratio 0.55 (PCL's default; the plugin never sets it), and
 *     GetFitnessScore = pcl::Registration::getFitnessScore = mean SQUARED distance to the nearest target POINT — the
 *     quantity LoopClosing gates at 0.2 (loop_closing.cpp:253, config/mapping/loop_closing.yaml:11) and Matching logs. */
#define LSA_NDT_MODE_MANUAL 0
#define LSA_NDT_MODE_PCL 1
int lsa_ndt_set_mode(lsa_ndt *ndt, int mode);

/* The derivative pass probes the target through a dense per-cell table (16 B per CELL of the target's bounding box) when
 * that box has at most max_cells cells (default 16 Mi = 256 MiB), otherwise through the occupancy bitmap (0.5 B per cell);
 * same results either way.  0 forces the bitmap.  Call before SetInputTarget. */
int lsa_ndt_set_dense_limit(lsa_ndt *ndt, long long max_cells);

/* SetInputTarget(input_target) (ndt_registration_manual.cpp:48-54 -> NormalDistributionsTransform::setInputTarget
 * -> VoxelGrid::setInput, VoxelGrid.cpp:326-358).  The target is copied to the device; the caller's
 * buffer is not retained. */
int lsa_ndt_set_input_target(lsa_ndt *ndt, const void *pts, size_t n, size_t stride, size_t w_off);
int lsa_ndt_set_input_target_device(lsa_ndt *ndt, const float *d_xyzw, size_t n);

/* ScanMatch(input_source, predict_pose, result_cloud_ptr, result_pose) (ndt_registration_manual.cpp:57-68).
 * result_cloud_ptr is never written by the reference plugin (Registration.cpp:92-95) and no caller reads
 * it, so it has no counterpart here.  stats may be NULL. */
int lsa_ndt_scan_match(lsa_ndt *ndt, const void *src, size_t n, size_t stride, const float predict_pose[16],
                       float result_pose[16], lsa_ndt_stats *stats);
int lsa_ndt_scan_match_device(lsa_ndt *ndt, const float *d_src_xyzw, size_t n, const float predict_pose[16],
                              float result_pose[16], lsa_ndt_stats *stats);

/* GetFitnessScore() (ndt_registration_manual.cpp:71-77 -> NormalDistributionsTransform::getFitnessScore,
 * :940-965): mean distance from each transformed source point of the LAST ScanMatch to the nearest
 * target-voxel centroid, searched the way the reference's octree does. */
int lsa_ndt_get_fitness_score(lsa_ndt *ndt, float *score);

/* ---- introspection used by the parity tests and bench.py ----------------------------------- */

/* info[0..2] = occupied-bounds dims (real_max - real_min + 1), [3..5] = real_min, [6] = occupied cells,
 * [7] = valid cells (>= 6 points and accepted eigenvalues, VoxelGrid.cpp:292-306), [8] = target points */
int lsa_ndt_grid_info(lsa_ndt *ndt, int64_t info[9]);
/* all occupied cells in ascending linear index (x fastest over the occupied bounds):
 * cell_ijk: 3 ints (absolute voxel coords) + npts + valid flag -> 5 ints per cell;
 * rec: centroid(3) icov sym (xx,xy,xz,yy,yz,zz) weight(1) -> 10 doubles per cell. cap in cells. */
int lsa_ndt_grid_dump(lsa_ndt *ndt, int32_t *cell_ints, double *cell_rec, size_t cap, size_t *n_cells);

/* diagnostics of the armed passes: sums over the armed passes since the last reset — out[0] passes, out[1] us spent posting
 * the mail, out[2] us from the mail posted to all rows taken, out[3] us from all rows taken to the next mail's post begun
 * (within one ScanMatch: row sums, Newton step, parameters), out[4] us the kernels waited on the device for their mail
 * (mean over workgroups, summed over passes), out[5] us from the mail taken to the row stored (same).  reset != 0 clears. */
int lsa_ndt_armed_timing(lsa_ndt *ndt, double out[6], int reset);

/* Armed passes (default ON; 0 or the environment variable LSA_NO_ARMED disables them): with the trial loop of the line search
 * dead (every shipped configuration) each Newton iteration is exactly one full derivative pass, so ScanMatch enqueues the
 * kernel of the NEXT pass right behind the current one; it waits on the device for its parameters, which the host posts into
 * a mailbox in device memory once the current pass's sums have produced them (csrc/ndt_derivatives.hip).  Launch and dispatch
 * latency leave the critical path; results are bit-identical.  The counters: passes served that way, armed kernels cancelled
 * (at most one per ScanMatch, at convergence), and passes that had to be redone the ordinary way because the armed kernel had
 * left before its mail came (the calling thread was held up for longer than the kernel's 0.3 ms patience).  Any may be NULL. */
int lsa_ndt_set_armed(lsa_ndt *ndt, int on);
/* At the end of a ScanMatch the armed kernel of the pass that never came is cancelled — no entry point of this library
 * returns with a kernel spinning on the caller's stream.  A caller whose next ScanMatch on the same handle follows at once
 * and who enqueues nothing of its own on the stream in between may OPT IN (on = 1, or the environment variable LSA_CARRY=1)
 * to leaving it PARKED for the first pass of the next ScanMatch (its mail also carries the source pointer, the point count
 * and the grid), which then starts without launch or dispatch latency (~8 us per ScanMatch).  A parked kernel leaves by itself
 * after its patience (0.3 ms) and any other call of this library on the same context releases it first; work the APPLICATION
 * enqueues itself on the stream, or a hipStreamSynchronize of its own, would wait out that patience.
 * lsa_pipeline_scan_match, which knows that scans follow back to back, carries between its scans by itself and cancels
 * behind its last one. */
int lsa_ndt_set_armed_carry(lsa_ndt *ndt, int on);
int lsa_ndt_armed_counters(lsa_ndt *ndt, uint64_t *passes, uint64_t *cancels, uint64_t *fallbacks);
/* Everything the armed path (and the row hand-over of the ordinary passes) can do other than the plain case, counted:
 * nothing about a slow or lost pass is silent. */
typedef struct {
  uint64_t passes;        /* passes served by a kernel enqueued ahead */
  uint64_t cancels;       /* armed kernels released by a cancel mail (ScanMatch end, another call on the context, ...) */
  uint64_t fallbacks;     /* passes redone the ordinary way: the armed kernel (or part of it) had left before its mail came */
  uint64_t late;          /* armed passes whose rows missed the polling limit (2 ms) but were complete once the stream had drained */
  uint64_t rows_late;     /* the same for ordinary passes (limit 20 ms) */
  uint64_t gave_up_seen;  /* times the host, waiting for rows, saw a workgroup report that it had left */
  uint64_t expired;       /* parked kernels found to have left before the next ScanMatch came (cancelled and relaunched) */
  double rows_wait_max_us;/* longest time from a pass's parameters being posted / its kernel launched to its last row taken */
  int32_t parked;         /* 1 while this handle has a kernel parked on its context's stream */
  int32_t pad;
  uint64_t deferred;      /* final passes of a ScanMatch that a batch did not wait for (collected behind the next scan's first mail) */
  double host_gap_max_us; /* host-stall detector: the longest interval between two consecutive looks of the calling thread at the rows
                           * it was waiting for (a look takes < 1 us), or from posting a pass to the first look.  >= 1000 means the OS
                           * held the thread up for longer than that: late / redone passes beside such a value are the box's doing */
  double row_wait_max_us; /* longest wait for ONE workgroup row (first look -> taken or given up): the quantity the 2 ms / 20 ms limits
                           * bound.  rows_wait_max_us above is a whole pass — its rows taken one after the other as they arrive — and
                           * exceeds a limit, with nothing late, when the DEVICE delivers slowly (another process on the GPU, a clock change) */
} lsa_ndt_armed_stats_t;
int lsa_ndt_armed_stats(lsa_ndt *ndt, lsa_ndt_armed_stats_t *out, int reset);
/* diagnostics / tests: how long an armed kernel waits for its mail before it leaves by itself (default 300 us; [1, 1e5]) */
int lsa_ndt_set_armed_patience(lsa_ndt *ndt, double patience_us);

/* ScanMatch can run its whole Newton iteration as ONE persistent launch (csrc/ndt_align.hip: the workgroups stay resident,
 * exchange their partial sums in-launch and each repeats the 6x6 solve) instead of one launch per derivative pass, when the
 * call qualifies: single GPU, the trial loop of the line search dead as in every shipped configuration, at most one
 * workgroup per CU, no trace; otherwise, and whenever the launch gives up (singular system -> SVD on the host, workgroups
 * not all resident in time), the per-pass path runs.  Both give the same bits (tests/test_gpu_align_persistent.py).
 * OFF by default: measured on MI355X it is no faster than the per-pass path alone (the in-launch exchange + the redundant
 * solve cost what the launches did) and slower beside other streams; on = 1 or the environment variable LSA_PERSISTENT=1
 * enables it.  The counters tell how many ScanMatch calls the persistent launch served and how many of its launches gave
 * up and were redone per pass. */
int lsa_ndt_set_persistent(lsa_ndt *ndt, int on);
int lsa_ndt_align_counters(lsa_ndt *ndt, uint64_t *launches, uint64_t *fallbacks);

/* diagnostics: with enable != 0 the next persistent ScanMatch calls record, per derivative pass, six 100 MHz clock values
 * of workgroup 0 (pass body start, end, row published, every flag seen, rows reduced, Newton step done); out (optional)
 * receives those of the last such call, 6 x 40 values, zero where no pass ran. */
int lsa_ndt_align_stamps(lsa_ndt *ndt, int enable, uint64_t *out, size_t cap, size_t *n_out);

/* Self-test of the scalar arithmetic the host driver and the device-side loop share (csrc/gn_math.h): n pseudo-random
 * cases through both; mismatches[0..3] = bitwise differences in {sinf/cosf, sin/cos, Newton step, sqrt/division} (must be
 * 0), mismatches[4] = how many cases were singular systems (taken by both sides alike). */
int lsa_gn_math_selftest(lsa_context *ctx, size_t n, uint64_t seed, uint64_t mismatches[5]);

/* Per-pass trace of the NEXT ScanMatch calls on this handle (rows as lsa_gn_solve_trace writes them); rows == NULL stops
 * it.  The buffers must stay valid while the trace is set. */
int lsa_ndt_set_trace(lsa_ndt *ndt, double *rows, int cap_rows, int *n_rows);

/* One derivative pass of computeDerivatives / computeHessian (NormalDistributionsTransform.cpp:391-445,
 * :901-936) at rigid transform T (f32, column-major) and pose p (f64, for the angle derivatives).
 * kind 0: score + gradient + Hessian, 1: score + gradient, 2: Hessian only.
 * out32 = [score, g(6), H upper triangle row-major (21), #pairs, #points, 0, 0].
 * The source must have been uploaded with lsa_ndt_set_source[_device]. */
int lsa_ndt_set_source(lsa_ndt *ndt, const void *src, size_t n, size_t stride);
int lsa_ndt_set_source_device(lsa_ndt *ndt, const float *d_src_xyzw, size_t n);
int lsa_ndt_derivatives(lsa_ndt *ndt, const float T[16], const double p[6], int kind, double out32[32]);
/* Enqueue `repeat` identical passes back to back and return the mean device time per pass (HIP events
 * on the context's stream): the roofline measurement of bench.py. */
int lsa_ndt_derivatives_timed(lsa_ndt *ndt, const float T[16], const double p[6], int kind, int repeat,
                              double out32[32], double *ms_per_pass);

/* Large sources in arbitrary point order (a merged or shuffled cloud rather than a sensor sweep): the pass hands 14 consecutive
 * points to a wave, and when those are not spatial neighbours every probe and voxel record pulls a cache line of its own (measured:
 * 5 x the algorithmic traffic, twice the time at 2 M points).  For sources of >= 262144 points the library therefore looks at a
 * sample of consecutive pairs and, if most are far apart, walks a spatially ordered copy instead (stable radix sort by 2 m Morton
 * cell: deterministic; the result differs from the unsorted walk by f64 summation order only).  mode 0: never, 1: when the
 * sample says so (default), 2: always.  The environment variable LSA_SOURCE_SORT sets the default. */
int lsa_ndt_set_source_sort(lsa_ndt *ndt, int mode);
int lsa_ndt_source_sort_count(lsa_ndt *ndt, uint64_t *count, int *last_was_sorted);

/* ---- multi-GPU: source cloud sharded over ranks, target replicated ------------------------- */
/* Called once per pass with a DEVICE buffer of n doubles (32) that must be summed in place over all
 * ranks (RCCL all-reduce over xGMI), enqueued on hip_stream; return 0 on success. */
typedef int (*lsa_allreduce_fn)(void *user, void *device_buf, size_t n_doubles, void *hip_stream);
int lsa_ndt_set_allreduce(lsa_ndt *ndt, lsa_allreduce_fn fn, void *user);

/* Convenience for C / C++ hosts: the same exchange step through RCCL called by the library itself (librccl is dlopen()ed
 * on first use — no link-time dependency, none at all for single-GPU use).  Protocol, one process per GPU: rank 0 calls
 * lsa_rccl_get_unique_id and ships the 128 bytes to the other ranks by any means (MPI, a socket, torch.distributed's
 * store); every rank calls lsa_rccl_comm_create (collective), then lsa_ndt_set_allreduce_rccl on its NDT handle. */
typedef struct lsa_rccl_comm lsa_rccl_comm;
int lsa_rccl_get_unique_id(void *id128);
int lsa_rccl_comm_create(lsa_rccl_comm **out, lsa_context *ctx, const void *id128, int rank, int world);
void lsa_rccl_comm_destroy(lsa_rccl_comm *comm);
int lsa_ndt_set_allreduce_rccl(lsa_ndt *ndt, lsa_rccl_comm *comm);   /* comm == NULL clears it */
int lsa_rccl_allreduce_f64(lsa_rccl_comm *comm, void *device_buf, size_t n_doubles);  /* in place, on the context's stream */
/* The ICP plugins shard identically: the neighbour grid is replicated (every rank calls SetInputTarget with the
 * whole target), each rank's ScanMatch gets its contiguous shard of the source, and the 32-double block of every iteration —
 * ICPRegistrationManual's H and B (icp_registration_manual.cpp:161-162), the rigid fit's 17 sums in PCL mode — is summed over the
 * ranks by the same callback (one all-reduce per iteration, on the context's stream); every rank then takes the identical host step.
 * With it installed the iterations run one launch each (no armed kernels: every iteration ends in a collective).
 * GetFitnessScore — of an NDT handle (either mode) or of the "ICP" plugin — sums {distance sum, point count} over the ranks the
 * same way when the handle has an exchange step installed: one all-reduce of 2 doubles. */
typedef struct lsa_icp lsa_icp;   /* (the ICP handle: section below) */
int lsa_icp_set_allreduce(lsa_icp *icp, lsa_allreduce_fn fn, void *user);
int lsa_icp_set_allreduce_rccl(lsa_icp *icp, lsa_rccl_comm *comm);   /* comm == NULL clears it */
/* The host step of ICPMANUAL / the point-to-plane extension with a caller-supplied evaluator of an iteration's sums (what lsa_gn_solve
 * is for NDT): eval fills out32 as the pass kernels do — [1..6] B = -J^T e, [7..27] the upper triangle of H = J^T J row by row,
 * [28] correspondences — already summed over the ranks, for the pose T (column-major 4x4).  No device is touched. */
typedef int (*lsa_icp_eval_fn)(void *user, const float T[16], double out32[32]);
int lsa_icp_gn_solve(lsa_icp_eval_fn eval, void *user, int max_iter, const float predict_pose[16], float result_pose[16]);
int lsa_rccl_comm_count(lsa_rccl_comm *comm, int *count);                              /* ncclCommCount: ranks the communicator spans */

/* ---- host-side Newton / More-Thuente driver with a caller-supplied pass evaluator ---------- */
/* The same driver lsa_ndt_scan_match runs, exposed so the multi-process path can be exercised
 * without a GPU (tests: world_size-2 gloo).  eval must fill out32 as lsa_ndt_derivatives does,
 * already summed over ranks. */
typedef int (*lsa_pass_eval_fn)(void *user, const float T[16], const double p[6], int kind, double out32[32]);
int lsa_gn_solve(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                 float outlier_ratio, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats);
/* same with the NDT mode (LSA_NDT_MODE_*) that selects the line-search bookkeeping; lsa_gn_solve is mode 0 */
int lsa_gn_solve_mode(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                      float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16],
                      lsa_ndt_stats *stats);
/* same, recording one row per derivative pass in the order the passes ran: {p(6), score, step length, pass kind} = 9
 * doubles; *trace_n counts every pass, rows beyond trace_cap are dropped.  trace_rows == NULL: no trace. */
int lsa_gn_solve_trace(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                       float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16],
                       lsa_ndt_stats *stats, double *trace_rows, int trace_cap, int *trace_n);
/* pose helpers of that driver (f32 rigid transform <-> the reference's (t, Euler XYZ) chart) */
void lsa_pose_to_matrix(const double p[6], float T[16]);
void lsa_matrix_to_pose(const float T[16], double p[6]);
/* Self-test hook (pure host arithmetic, no device): the f32 Eigen blocks of csrc/gn_math.h on caller-supplied inputs, so that
 * the CPU tests can hold the PRODUCT's copy to the same reference fixtures as the oracle's (tests/golden/eigen_blocks.json).
 * what 0: Transform<float,3,Affine>::rotation() (NormalDistributionsTransform.cpp:336), in 9 row-major -> out 9;
 *      1: Matrix<float,6,6>::determinant() (icp_registration_manual.cpp:165), in 36 -> out 1;
 *      2: inverse() and inverse() * B (:169), in 36 + 6 -> out 36 + 6;   3: Matrix3f product (:173), in 9 + 9 -> out 9. */
int lsa_eigen_f32_selftest(int what, const float *in, float *out);
void lsa_ndt_gauss_constants(float res, float outlier_ratio, double *d1, double *d2);

/* ---------------------------------------------------------------------------------------------
 * Batch of scans through the per-scan hot path: for k in [0, count): filtered_k = Filter(scan_k) (front_end.cpp:106-107),
 * result_k = ScanMatch(filtered_k, predict_k) (:225) — with the two stages software-pipelined inside the library: the
 * filter of scan k+1 runs on the filter's context (create it on its OWN lsa_context, i.e. its own HIP stream) while the
 * registration of scan k runs on the NDT's.  Every scan is filtered once and registered once; the results are those of
 * the sequential calls.  d_filtered_a / _b: two device buffers of max(n_points) float4 each.  rank / world: (0, 1) on one
 * GPU; with world > 1 the rank registers its contiguous shard of every filtered scan and the NDT handle must carry the
 * all-reduce (lsa_ndt_set_allreduce / _rccl).  stats and n_filtered may be NULL.
 * Inside the batch the library also overlaps ACROSS scans what the reference's own arithmetic leaves independent: with the trial
 * loop of the line search dead, the step length and the convergence test (NormalDistributionsTransform.cpp:379-381) of a Newton
 * iteration are known before the derivatives at the new pose are evaluated (:696-698), so the pose of scan k is final BEFORE its
 * last derivative pass; that pass — it only feeds stats[k].score / trans_probability / pairs_per_pass — is posted, not waited
 * for, and its sums complete stats[k] behind the first pass of scan k + 1 (the last scan's before the call returns).  Every pass
 * runs, every number returned is the sequential call's (tests/test_gpu_parity.py); a third filtered-cloud buffer, owned by the
 * filter handle, keeps the filter of scan k + 2 off the cloud that pass still reads.  LSA_NO_DEFER=1 switches it off.
 * ------------------------------------------------------------------------------------------- */
int lsa_pipeline_scan_match(lsa_voxel_filter *frame_filter, lsa_ndt *ndt, size_t count, const float *const *d_scans_xyzw,
                            const size_t *n_points, const float *predict_poses, float *d_filtered_a, float *d_filtered_b,
                            int rank, int world, float *result_poses, lsa_ndt_stats *stats, size_t *n_filtered);
/* The same batch fed with HOST clouds (`count` clouds of n_points[k] rows of `stride` bytes, intensity at w_off or SIZE_MAX; untouched
 * until the call returns): Filter(host) + ScanMatch per scan as front_end.cpp:106-107,225 calls them, with the next n_filters scans
 * packed and crossing the link while the current one is registered.  filters: 2 .. 8 distinct VoxelFilter handles of the same leaf,
 * each on a context (stream) of its own other than the registration's; their device buffers are used (two suffice: measured, more do
 * not help).  Poses and statistics are those of the sequential host calls. */
int lsa_pipeline_scan_match_host(lsa_voxel_filter *const *filters, int n_filters, lsa_ndt *ndt, size_t count, const void *const *clouds,
                                 const size_t *n_points, size_t stride, size_t w_off, const float *predict_poses, float *result_poses,
                                 lsa_ndt_stats *stats, size_t *n_filtered);

/* ---------------------------------------------------------------------------------------------
 * Position-only initialisation — replaces Matching::generateGauss2DMapCells / resetMapRange / getInitialYawAngle
 * (src/matching/matching.cpp:344-422, :267-307; `init_type: OnlyPosition`, config/matching/matching.yaml:4): a 2-D grid
 * of per-cell height statistics over (local map - origin), and the best of 270 yaw hypotheses of a scan against it.
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_yaw_init lsa_yaw_init;
/* grid_resolution: matching.yaml `grid_resolution` (clamped to >= 0.1 as InitGridMapResolution does, :137-138) */
int lsa_yaw_init_create(lsa_yaw_init **out, lsa_context *ctx, double grid_resolution);
void lsa_yaw_init_destroy(lsa_yaw_init *y);
/* SetInitPose -> generateGauss2DMapCells (:325-342): origin = local_map_origion_ = the position ResetLocalMap was given */
int lsa_yaw_init_set_local_map(lsa_yaw_init *y, const void *pts, size_t n, size_t stride, const float origin[3]);
int lsa_yaw_init_set_local_map_device(lsa_yaw_init *y, const float *d_xyzw, size_t n, const float origin[3]);
/* getInitialYawAngle(current_cloud) (:267-307). probs270 (optional) receives the 270 hypothesis scores. */
int lsa_yaw_init_get_initial_yaw(lsa_yaw_init *y, const void *scan, size_t n, size_t stride, double *yaw,
                                 double *probs270);
int lsa_yaw_init_get_initial_yaw_device(lsa_yaw_init *y, const float *d_scan_xyzw, size_t n, double *yaw,
                                        double *probs270);
/* introspection for the parity tests: local_map_width_/height_, map_min/max_xyz_; cell k = cell_x * height + cell_y */
int lsa_yaw_init_grid_info(lsa_yaw_init *y, int32_t wh[2], float min_xyz[3], float max_xyz[3]);
int lsa_yaw_init_cells(lsa_yaw_init *y, float *mu, float *sigma, int32_t *point_cnt, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * ICP registration.
 *   LSA_ICP_MODE_MANUAL (0) replaces ICPRegistrationManual (icp_registration_manual.cpp:12-179): point-to-point
 *     Gauss-Newton, fixed max_iter iterations, exact 1-NN correspondences gated by squared-distance <=
 *     max_correspond_dis (sic, :147).
 *   LSA_ICP_MODE_PCL (2) replaces ICPRegistration (icp_registration.cpp:20-64) -> pcl::IterativeClosestPoint of
 *     third-party PCL 1.7 (not in the tree; parity UNPINNED): 1-NN correspondences gated at max_correspond_dis,
 *     SVD (Umeyama) increment, DefaultConvergenceCriteria with euclid_eps / trans_eps (lsa_icp_set_criteria; defaults
 *     0.01 / 0.01 = config/mapping/front_end.yaml:31-35), GetFitnessScore = mean squared nearest-point distance.
 *   LSA_ICP_MODE_PLANE (1) is a build EXTENSION, not reference code (BASELINE.json configs[2] asks for point-to-plane ICP;
 *     both plugins of the reference are point-to-point): the ICPMANUAL Gauss-Newton skeleton with the residual
 *     n . (p' - q), pairs gated at distance <= max_correspond_dis, normals = smallest-eigenvalue eigenvector of the
 *     covariance of the target points in the neighbour's voxel (lsa_icp_set_normal_leaf, default 1.0 m, >= 6 points).
 * ------------------------------------------------------------------------------------------- */
#define LSA_ICP_MODE_MANUAL 0
#define LSA_ICP_MODE_PLANE 1
#define LSA_ICP_MODE_PCL 2
/* (typedef struct lsa_icp lsa_icp: declared above, beside the multi-GPU entry points) */
typedef struct {
  int32_t iterations;      /* nr_iterations_ (manual mode: max_iter) */
  int32_t converged;       /* hasConverged() */
  double correspondences;  /* pairs in the last iteration */
  double mse;              /* last mean squared correspondence distance the criteria looked at (PCL mode) */
} lsa_icp_stats;
int lsa_icp_create(lsa_icp **out, lsa_context *ctx, float max_correspond_dis, int max_iter, int mode);
/* ICPRegistration::SetRegistrationParam's setEuclideanFitnessEpsilon / setTransformationEpsilon (icp_registration.cpp:30-33) */
int lsa_icp_set_criteria(lsa_icp *icp, float euclid_eps, float trans_eps);
int lsa_icp_get_stats(lsa_icp *icp, lsa_icp_stats *stats);
/* ICPMANUAL / point-to-plane modes, last ScanMatch: from the second iteration on a source point whose old neighbour is provably
 * still the nearest does not search (csrc/icp.hip; LSA_ICP_NO_SKIP=1 runs every search, same results).  searches / queries:
 * totals over the iterations; per_iteration[i] (up to cap entries, *n_iterations of them exist): source points that searched in
 * iteration i.  Any pointer may be NULL. */
int lsa_icp_search_counts(lsa_icp *icp, uint64_t *searches, uint64_t *queries, uint32_t *per_iteration, size_t cap, size_t *n_iterations);
/* ICPMANUAL / point-to-plane modes launch the kernel of iteration k + 1 while iteration k runs; it takes its pose by mail
 * (csrc/icp.hip, "armed passes"; LSA_ICP_NO_ARMED=1: one launch per iteration, same results).  armed_passes: iterations run that
 * way since the handle was created; fallbacks: iterations whose armed kernel delivered no rows and that were run again the
 * ordinary way — never silent. */
int lsa_icp_armed_counts(lsa_icp *icp, uint64_t *armed_passes, uint64_t *fallbacks);
/* how long a pre-launched iteration kernel waits for its pose before it leaves (default 300 us; tests shorten it to drive the
 * fallback path) */
int lsa_icp_set_armed_patience(lsa_icp *icp, double microseconds);
/* diagnostics: the per-source-point search state of the last ScanMatch, copied to the host — prev_out[i]: position of point i's
 * neighbour in the handle's cell-sorted target (0xffffffff: none); qref_out[4 i .. 4 i + 3]: where the point last searched (xyz,
 * map frame) and the proof radius it left (metres) */
int lsa_icp_query_state(lsa_icp *icp, uint32_t *prev_out, float *qref_out, size_t n);
/* plane mode: edge of the voxels the target normals are estimated in (call before SetInputTarget) */
int lsa_icp_set_normal_leaf(lsa_icp *icp, float leaf);
void lsa_icp_destroy(lsa_icp *icp);
int lsa_icp_set_input_target(lsa_icp *icp, const void *pts, size_t n, size_t stride);
int lsa_icp_set_input_target_device(lsa_icp *icp, const float *d_xyzw, size_t n);
int lsa_icp_scan_match(lsa_icp *icp, const void *src, size_t n, size_t stride, const float predict_pose[16],
                       float result_pose[16]);
int lsa_icp_scan_match_device(lsa_icp *icp, const float *d_src_xyzw, size_t n, const float predict_pose[16],
                              float result_pose[16]);
/* GetFitnessScore(): ICPMANUAL: unimplemented in the reference, returns 0 (:62-67); PCL mode: icp_registration.cpp:60-63 */
int lsa_icp_get_fitness_score(lsa_icp *icp, float *score);

/* ---------------------------------------------------------------------------------------------
 * Front end with a device-resident sliding local map — the caller of the two plugin
 * interfaces above: FrontEnd::Update / UpdateWithNewFrame / SetInitPose
 * (lidar_localization/src/mapping/front_end/front_end.cpp:88-341, :348-424, :343-346), without the detector-box
 * static-weight bookkeeping (:111-161, :247-334).  The key-frame clouds, the concatenated local map, its 0.3 m
 * filtered version and the NDT voxel table all stay in HBM; a key-frame costs no host <-> device transfer.
 * ------------------------------------------------------------------------------------------- */
typedef struct lsa_front_end lsa_front_end;

typedef struct {
  float key_frame_distance;   /* front_end.yaml:11, L1 translation (front_end.cpp:243-245) */
  int32_t local_frame_num;    /* front_end.yaml:12 */
  int32_t use_frame_filter;   /* frame_filter: voxel_filter (1) / no_filter (0) */
  int32_t use_map_filter;     /* local_map_filter */
  float frame_leaf[3];        /* front_end.yaml:46-47 */
  float map_leaf[3];          /* front_end.yaml:44-45 */
  float ndt_res, ndt_step_size, ndt_trans_eps, ndt_outlier_ratio; /* front_end.yaml:24-29 */
  int32_t ndt_max_iter;
  int32_t reset_intensity;    /* 1: like the reference, key-frames after the first are stored with static weight 1.0
                                 (front_end.cpp:304-306); 0: keep the incoming intensities */
} lsa_front_end_config;

typedef struct {
  int32_t key_frame;          /* this scan became a key-frame (local map + target rebuilt) */
  int32_t local_map_frames;
  int64_t filtered_points;    /* source points after the frame filter */
  int64_t local_map_points;   /* concatenated key-frame points */
  int64_t target_points;      /* points handed to SetInputTarget (filtered once >= 10 key-frames) */
  lsa_ndt_stats ndt;          /* of this scan's ScanMatch (zero for the first frame) */
} lsa_front_end_stats;

/* FrontEnd::FrontEnd() + InitWithConfig (front_end.cpp:30-86).  Also allocates the session's device buffers for scans of up to
 * 160,000 points (lsa_front_end_reserve; environment LSA_FE_RESERVE_POINTS = another size, 0 = allocate on the way as scans come). */
int lsa_front_end_create(lsa_front_end **out, lsa_context *ctx, const lsa_front_end_config *cfg);
void lsa_front_end_destroy(lsa_front_end *fe);
/* Allocate now what Update would otherwise allocate under the first scans of a session (key-frame clouds of a whole window, the
 * local-map buffer and tables, filter and registration buffers, the batch entry point's stream and helper thread): ~300 MB at
 * 160 k points.  Larger scans still work (buffers grow); no result depends on it. */
int lsa_front_end_reserve(lsa_front_end *fe, size_t max_scan_points);
/* FrontEnd::SetInitPose */
int lsa_front_end_set_init_pose(lsa_front_end *fe, const float pose[16]);
/* FrontEnd::Update(cloud_data, cloud_pose): one scan in, its pose (map <- lidar, column-major) out */
int lsa_front_end_update(lsa_front_end *fe, const void *cloud, size_t n, size_t stride, size_t w_off, float pose_out[16],
                         lsa_front_end_stats *stats);
/* the same on a cloud already in device memory (packed x, y, z, w rows, produced on the context's stream).  The cloud is the caller's
 * again when the call returns, as with FrontEnd::Update: the reference keeps a key-frame's raw cloud (front_end.cpp:375) and so does
 * this handle — in a copy the frame filter's first kernel leaves in a buffer of the handle — so the local map's update, which runs on
 * a stream of its own after the call has returned, never reads the caller's buffer.  (Without a frame filter, or with the incremental
 * map switched off, the key-frame's cloud is read on / in front of the context's stream: work enqueued on that stream afterwards is
 * ordered behind the read; a writer on another stream, or the host, synchronises the context's stream first.) */
int lsa_front_end_update_device(lsa_front_end *fe, const float *d_xyzw, size_t n, float pose_out[16], lsa_front_end_stats *stats);
/* diagnostics: with out == NULL, enable != 0 makes the three kernels of every following incremental map update note the device's
 * 100 MHz wall clock at their phase boundaries (8 values per workgroup); with out != NULL the 3 x *per_kernel values are copied out
 * (kernel k's workgroup w at out[k * *per_kernel + 8 w ...]) and cleared.  profiles/scripts/lm_stamps.py prints them. */
int lsa_front_end_map_stamps(lsa_front_end *fe, int enable, uint64_t *out, size_t cap, size_t *per_kernel);
/* `count` Updates in one call — a recorded drive being replayed, or a sensor queue that has backed up: the loop
 * front_end_flow.cpp:18-36 runs over its cloud_data_buff_ (while (HasData()) { ValidData(); UpdateLaserOdometry(); }).  The
 * frame filter of scan k + 1 runs on a second stream while scan k is registered and the local map updated (it needs no pose);
 * within the call the next pass's kernel stays parked between two ScanMatches and a ScanMatch's final derivative pass, which
 * only completes its statistics, is collected behind the next scan's first pass (as in lsa_pipeline_scan_match).  Poses,
 * key-frame decisions, local map and statistics are those of `count` calls of lsa_front_end_update_device, bit for bit.
 * d_scans_xyzw[k]: device cloud of n_points[k] points, complete or produced on this context's stream, untouched until the call
 * returns; poses_out: 16 floats per scan; stats: `count` records or NULL.  On an error the scans before it have been applied. */
int lsa_front_end_update_batch_device(lsa_front_end *fe, size_t count, const float *const *d_scans_xyzw, const size_t *n_points, float *poses_out,
                                      lsa_front_end_stats *stats);
/* The same run fed with HOST clouds — what FrontEndFlow::Run's buffered clouds are (front_end_flow.cpp:18-36): `count` clouds of
 * n_points[k] rows of `stride` bytes (x, y, z at 0, the intensity / static weight at w_off, SIZE_MAX = none), untouched until the call
 * returns.  Scan k + 1 is packed and crosses the link (pulled by its frame filter's first kernel) while scan k is registered.  Poses,
 * key-frame decisions, local map and statistics are those of `count` calls of lsa_front_end_update, bit for bit. */
int lsa_front_end_update_batch(lsa_front_end *fe, size_t count, const void *const *clouds, const size_t *n_points, size_t stride, size_t w_off,
                               float *poses_out, lsa_front_end_stats *stats);
/* the NDT handle the front end owns (for GetFitnessScore / introspection); do not destroy it */
lsa_ndt *lsa_front_end_registration(lsa_front_end *fe);
/* ... and its frame filter (NULL without one): diagnostics (lsa_voxel_filter_counters: how many host scans the device pulled itself) */
lsa_voxel_filter *lsa_front_end_frame_filter(lsa_front_end *fe);
/* Incremental local map (default ON; 0 or the environment variable LSA_FE_INCREMENTAL=0 selects the full rebuild).
 * UpdateWithNewFrame (front_end.cpp:348-424) rebuilds filter output and voxel table from all <= 20 key-frames on every
 * key-frame; with the incremental map the per-cell sums of both levels are persistent exact integers, a key-frame adds its own
 * points, subtracts the evicted key-frame's and re-finalises only the cells either touched (csrc/local_map.hip).  Both paths
 * produce the same voxel table bit for bit (tests/test_gpu_front_end.py), so switching is not observable in the poses. */
int lsa_front_end_set_incremental(lsa_front_end *fe, int on);
/* out[0] full target builds, [1] incremental updates, [2] rebuilds of the persistent tables, [3] re-fits of the dense window,
 * [4] 1 while the NDT handle runs on the incremental map's table, [5] 1 if the incremental path switched itself off */
int lsa_front_end_map_counters(lsa_front_end *fe, uint64_t out[6]);
/* tests: rebuilds the target the full way on scratch handles and compares it with the incremental map, cell by cell and bit
 * by bit.  out[0] occupied cells of the full table, [1] cells whose state or record differs in any bit (must be 0),
 * [2] live cells / [3] valid voxels of the incremental map ([2] must equal [0]), [4] points of the full build's filtered map /
 * [5] live cells of the incremental filter level (equal; both 0 while the map is unfiltered). */
int lsa_front_end_map_verify(lsa_front_end *fe, uint64_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* LIDAR_SLAM_AMD_H_ */
