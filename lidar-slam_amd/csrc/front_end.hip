// front_end.hip — lsa_front_end_*: the caller of the hot path, with the sliding local map kept in HBM.
//
// SURVEY.md section 8f row 1.  Mirrors FrontEnd::Update and FrontEnd::UpdateWithNewFrame
// (reference: lidar_localization/src/mapping/front_end/front_end.cpp:88-341 and :348-424) for the scan-to-local-map
// odometry loop, without the detector-box static-weight bookkeeping (:111-161, :247-334, inactive without the bbox
// files the reference loads from an absolute path outside its tree):
//   frame filter -> [first frame: becomes key-frame 0 at the init pose] -> ScanMatch(filtered, predict_pose)
//   -> constant-velocity prediction step = last^-1 * cur, predict = cur * step (:238-240)
//   -> key-frame if the L1 translation from the last key-frame exceeds key_frame_distance (:243-245)
//   -> UpdateWithNewFrame: keep the RAW (unfiltered) cloud of the last `local_frame_num` key-frames, transform each by
//      its pose (pcl::transformPointCloud, f32), concatenate, and SetInputTarget with the raw map while there are fewer
//      than 10 key-frames, else with the map filtered at the local-map leaf (:394-410).
// In the reference every key-frame re-uploads nothing (CPU) but re-transforms 20 x 120 k points and re-filters 2.4 M
// points on one core (~100 ms); here the key-frame clouds stay on the device, so a key-frame costs one D2D copy, 20 small
// transform kernels, one filter and one voxel-table build, with no host <-> device traffic at all.
#include "local_map.h"
#include "ndt_internal.h"

#include <chrono>
#include <cmath>
#include <deque>
#include <vector>

namespace {
constexpr int kBlock = 256;

struct T12 { float v[12]; };

// pcl::transformPointCloud: xyz' = M * xyz + t in f32 (left to right, no FMA); every other field copied
__global__ __launch_bounds__(kBlock) void k_transform(const float4 *__restrict__ in, size_t n, T12 T, float4 *__restrict__ out, int force_w_one) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 s = in[i];
  float4 o;
  o.x = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[0], s.x), lsa_fmul(T.v[1], s.y)), lsa_fmul(T.v[2], s.z)), T.v[3]);
  o.y = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[4], s.x), lsa_fmul(T.v[5], s.y)), lsa_fmul(T.v[6], s.z)), T.v[7]);
  o.z = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[8], s.x), lsa_fmul(T.v[9], s.y)), lsa_fmul(T.v[10], s.z)), T.v[11]);
  o.w = force_w_one ? 1.0f : s.w;
  out[i] = o;
}

// the same into a fixed-size slot of the local-map buffer: points [n, cap) of the slot become NaN, which every kernel
// downstream (filter, voxel table, neighbour grid) skips
__global__ __launch_bounds__(kBlock) void k_transform_slot(const float4 *__restrict__ in, size_t n, size_t cap, T12 T, float4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  float4 o = make_float4(NAN, NAN, NAN, 0.0f);
  if (i < n) {
    float4 s = in[i];
    o.x = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[0], s.x), lsa_fmul(T.v[1], s.y)), lsa_fmul(T.v[2], s.z)), T.v[3]);
    o.y = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[4], s.x), lsa_fmul(T.v[5], s.y)), lsa_fmul(T.v[6], s.z)), T.v[7]);
    o.z = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[8], s.x), lsa_fmul(T.v[9], s.y)), lsa_fmul(T.v[10], s.z)), T.v[11]);
    o.w = s.w;
  }
  out[i] = o;
}

__global__ __launch_bounds__(kBlock) void k_copy_w(const float4 *__restrict__ in, size_t n, float4 *__restrict__ out, int force_w_one) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 s = in[i];
  if (force_w_one) s.w = 1.0f;
  out[i] = s;
}

void mat4_mul(const float a[16], const float b[16], float c[16]) {  // column-major, f32 like Eigen::Matrix4f
  float r[16];
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) {
      float s = 0.f;
      for (int k = 0; k < 4; k++) s += a[k * 4 + row] * b[col * 4 + k];
      r[col * 4 + row] = s;
    }
  memcpy(c, r, sizeof(r));
}

bool mat4_inverse(const float m[16], float inv_out[16]) {  // general 4x4 inverse by cofactors (what Matrix4f::inverse() does), f32
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.f) return false;
  float id = 1.0f / det;
  for (int i = 0; i < 16; i++) inv_out[i] = inv[i] * id;
  return true;
}

void identity16(float m[16]) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}
}  // namespace

struct KeyFrame {
  DevBuf<float4> pts;
  size_t n = 0;
  float pose[16];
  int slot = -1;  // which slot of the local-map buffer holds its transformed points
};

struct lsa_front_end {
  lsa_context *ctx = nullptr;
  lsa_front_end_config cfg{};
  lsa_voxel_filter *frame_filter = nullptr, *map_filter = nullptr;
  lsa_ndt *ndt = nullptr;
  // lsa_front_end_update_batch_device: a second context (stream) with a frame filter of its own, three filtered clouds in rotation
  lsa_context *ctx_b = nullptr;
  lsa_voxel_filter *frame_filter_b = nullptr;
  DevBuf<float4> d_batch[4];   // (the device-cloud run rotates the first three, the host-cloud run all four)
  DevBuf<float4> d_raw[4];     // lsa_front_end_update_batch (host scans): the raw rows of the scans in flight, in rotation like d_batch
  // ... whose ingest is TWO scans deep (a host scan's filter — pack, link, kernels: ~95 us — is longer than a registration): a third
  // context / frame filter takes every other scan
  lsa_context *ctx_c = nullptr;
  lsa_voxel_filter *frame_filter_c = nullptr;
  std::deque<KeyFrame *> frames;
  std::vector<KeyFrame *> pool;
  DevBuf<float4> d_in, d_filtered, d_map, d_map_f;
  PinnedBuf<float> h_stage;
  float init_pose[16], last_pose[16], predict_pose[16], step_pose[16], last_kf_pose[16];
  size_t map_points = 0, target_points = 0;
  size_t slot_cap = 0;  // points per local-map slot
  long frames_seen = 0;
  // incremental local map (local_map.hip): the persistent sums of the window, updated key-frame by key-frame; the NDT handle
  // then runs on its records / dense table (an EXTERNAL table, ndt_internal.h) instead of a table rebuilt per key-frame
  LocalMap lm;
  // The incremental map lives on a context (HIP stream) of its OWN.  A key-frame's update — three dependent launches, ~40 us — then
  // runs beside whatever the registration's stream does next that does not need the map: in a scan-by-scan loop the NEXT scan's frame
  // filter (which a shared stream made wait behind the update: that was the whole difference between a key-frame scan and an ordinary
  // one), in a batch the tail of the deferred final pass and the release of the parked kernel.  The registration's stream is ordered
  // behind the update by an event before anything on it reads the table (fe_map_join); what the update reads — the caller's scan, the
  // slot's old points — is complete by then as far as the HOST knows (the frame filter that read the scan has been waited for).
  // Inside a BATCH (lsa_front_end_update_batch_device) the update stays on the registration's stream: the next scan's frame filter
  // already runs on a stream of its own there, nothing is left to put beside the update, and a cross-stream event in front of the
  // next ScanMatch's first (pre-launched) kernel costs ~10 us per key-frame (measured: the kernel starts 11.7 us after the update's
  // last kernel ends, against 0 on one stream).
  lsa_context *ctx_m = nullptr;
  bool in_batch = false;
  hipEvent_t ev_map = nullptr, ev_in = nullptr, ev_cloud = nullptr;   // ev_cloud: behind the last read of the caller's cloud on the map's stream
  bool map_inflight = false;   // an update has been enqueued on ctx_m's stream that the registration's stream is not yet ordered behind
  DevBuf<float4> d_in_alt;     // lsa_front_end_update (host clouds): the upload alternates between two buffers (see there)
  bool in_alt = false;
  bool cloud_is_callers = false;   // the cloud handed to fe_update_filtered is the caller's own buffer (not a copy this handle owns)
  // how a key-frame's read of a caller-owned device cloud is made safe: 2 (default) the frame filter's first kernel leaves a copy in a
  // buffer of this handle; 1 the context's stream is ordered behind the update's read (what remains without a frame filter);
  // 0 neither — the pre-round-5 contract, the A/B arm of profiles/r05_fe_cloud_ab.log only.  LSA_FE_CLOUD=keep|event|none
  // a single Update enqueues the registration's first pass kernel (parked for its mail) BEHIND the frame filter's kernels before it
  // waits for the filter: launch + dispatch of that kernel pass under the filter instead of between the filter and the first pass
  // (kernel trace of the scan-by-scan loop: 18.5 us of idle GPU there, `profiles/r05_fe_prearm_ab.log`).  LSA_FE_PREARM=0: off.
  bool prearm_single = getenv("LSA_FE_PREARM") ? atoi(getenv("LSA_FE_PREARM")) != 0 : true;
  int cloud_mode = !getenv("LSA_FE_CLOUD") ? 2 : (!strcmp(getenv("LSA_FE_CLOUD"), "none") ? 0 : (!strcmp(getenv("LSA_FE_CLOUD"), "event") ? 1 : 2));
  bool incremental = getenv("LSA_FE_INCREMENTAL") ? atoi(getenv("LSA_FE_INCREMENTAL")) != 0 : true;
  bool lm_off = false;       // fell back to the full build for good (window / table limits, coordinates out of range)
  bool lm_adopted = false;   // the NDT handle currently runs on the local map's table
  // a single Update defers its ScanMatch's final pass over the key-frame's first launches (fe_update_filtered); LSA_FE_DEFER=0: waits for it first
  bool defer_single = getenv("LSA_FE_DEFER") ? atoi(getenv("LSA_FE_DEFER")) != 0 : true;
  unsigned long long scale_n = 0;
  unsigned long long n_full_builds = 0;
};

void lsa_voxel_filter_set_scale_floor(lsa_voxel_filter *f, unsigned long long n);   // voxel_filter.hip
void ndt_set_scale_floor(lsa_ndt *h, unsigned long long n);                         // ndt_grid.hip
bool ndt_carry_scope(lsa_ndt *h, bool on);                                          // ndt_api.hip
void ndt_release_parked(lsa_ndt *h);
void ndt_prearm(lsa_ndt *h);
bool ndt_defer_scope(lsa_ndt *h, bool on);
int ndt_collect_deferred(lsa_ndt *h);
int lsa_voxel_filter_apply_device_async_here(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw);   // voxel_filter.hip
int lsa_voxel_filter_apply_device_keep(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out, float *d_keep);
int lsa_voxel_filter_enqueue_device(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, float *d_keep);
int lsa_voxel_filter_wait_parked(lsa_voxel_filter *f, size_t *n_out);
int lsa_voxel_filter_apply_host_async(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float *d_rows, float *d_out);
int lsa_voxel_filter_apply_host_to_device(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float *d_rows, float *d_out,
                                          size_t *n_out, int *pulled, void (*after_enqueue)(void *), void *after_user);

static int fe_batch_context(lsa_front_end *fe);
static int fe_host_batch_context(lsa_front_end *fe);

namespace {
// The local map buffer is `local_frame_num` slots of slot_cap points; a key-frame's transformed cloud lives in its slot
// until the frame leaves the window (its pose never changes), so a new key-frame costs ONE transform launch — into the
// slot of the frame it evicts — instead of re-transforming all 20.  The frames therefore sit in the buffer in slot order,
// not in age order, with NaN padding behind each: the 0.3 m filter and the voxel table sum in exact fixed point, so their
// results do not depend on the order of the points (and they skip non-finite ones).
// which stream the local-map buffer is written on: the map's own while the incremental map is in use, the registration's otherwise
lsa_context *fe_map_ctx(lsa_front_end *fe) { return (fe->incremental && !fe->lm_off && !fe->in_batch) ? fe->ctx_m : fe->ctx; }
hipStream_t fe_map_stream(lsa_front_end *fe) { return fe_map_ctx(fe)->stream; }

// order the registration's stream behind the map update in flight (before anything on it reads the table, the map buffer or a
// key-frame's cloud); the host itself learns of the update's end from the status block (lm_check)
int fe_map_join(lsa_front_end *fe) {
  if (!fe->map_inflight) return LSA_OK;
  LSA_HIP(fe->ctx, hipStreamWaitEvent(fe->ctx->stream, fe->ev_map, 0));
  fe->map_inflight = false;
  return LSA_OK;
}
// everything the map's stream holds has completed (host wait): the slow paths that read the map buffer on the registration's stream
int fe_map_drain(lsa_front_end *fe) {
  LSA_HIP(fe->ctx, hipStreamSynchronize(fe->ctx_m->stream));
  fe->map_inflight = false;
  return LSA_OK;
}

void fe_transform_into_slot(lsa_front_end *fe, KeyFrame *f) {
  T12 T;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T.v[r * 4 + c] = f->pose[c * 4 + r];
  hipLaunchKernelGGL(k_transform_slot, dim3(lsa_div_up(fe->slot_cap, kBlock)), dim3(kBlock), 0, fe_map_stream(fe), f->pts.p, f->n, fe->slot_cap, T,
                     fe->d_map.p + (size_t)f->slot * fe->slot_cap);
}

// the FULL build of the target from the whole local-map buffer (every key-frame without the incremental map, and whenever
// an entry point needs an ordinary table): 0.3 m filter once there are 10 key-frames (front_end.cpp:394-410), then the
// voxel table
int fe_full_target(lsa_front_end *fe) {
  lsa_context *ctx = fe->ctx;
  int rc;
  // (the map buffer may have been written on the map's own stream: the build below reads it on the registration's)
  if (fe->incremental && (rc = fe_map_drain(fe))) return rc;
  size_t sum = 0;
  for (KeyFrame *f : fe->frames) sum += f->n;
  const size_t total = fe->slot_cap * fe->frames.size();  // slots 0 .. size-1 are the occupied ones (see fe_add_key_frame)
  fe->lm_adopted = false;
  fe->n_full_builds++;
  if ((int)fe->frames.size() < 10 || !fe->cfg.use_map_filter) {  // front_end.cpp:394-396: few key-frames -> unfiltered map
    fe->target_points = sum;
    return lsa_ndt_set_input_target_device(fe->ndt, (const float *)fe->d_map.p, total);
  }
  if ((rc = fe->d_map_f.reserve(ctx, total, 1.25))) return rc;
  size_t V = 0;
  rc = lsa_voxel_filter_apply_device(fe->map_filter, (const float *)fe->d_map.p, total, (float *)fe->d_map_f.p, &V);
  if (rc == LSA_ERR_RANGE) {  // PCL passes the input through
    fe->target_points = sum;
    return lsa_ndt_set_input_target_device(fe->ndt, (const float *)fe->d_map.p, total);
  }
  if (rc) return rc;
  fe->target_points = V;
  return lsa_ndt_set_input_target_device(fe->ndt, (const float *)fe->d_map_f.p, V);
}

int fe_materialize_cb(void *user) { return fe_full_target((lsa_front_end *)user); }

int fe_lm_adopt(lsa_front_end *fe) {
  fe->lm_adopted = true;
  return ndt_adopt_external(fe->ndt, fe->lm.rec.p, fe->lm.dense.p, fe->lm.wmin, fe->lm.wdim, fe->target_points, fe_materialize_cb, fe);
}

// the persistent sums from scratch from the occupied slots; a map the tables / the window cannot hold switches the
// incremental path off for good and builds the target the full way
// in_flight (the map has a stream of its own): the rebuild's last kernels are left running and the registration's stream is ordered
// behind them by the map's event, as after an ordinary update (fe_map_join)
int fe_lm_rebuild(lsa_front_end *fe, bool filtered, bool in_flight = false) {
  const float4 *range = fe->d_map.p;
  const size_t n = fe->slot_cap * fe->frames.size();
  int rc = lm_rebuild(fe->lm, &range, &n, 1, filtered, in_flight);
  if (!rc && in_flight) {
    LSA_HIP(fe->ctx, hipEventRecord(fe->ev_map, fe->ctx_m->stream));
    fe->map_inflight = true;
  }
  if (rc == LSA_ERR_CAPACITY || rc == LSA_ERR_RANGE) {
    fe->lm_off = true;
    return fe_full_target(fe);
  }
  if (rc) return rc;
  if (filtered) fe->target_points = (size_t)(fe->lm.h_status.p->live_f > 0 ? fe->lm.h_status.p->live_f : 0);
  return fe_lm_adopt(fe);
}

// Before a ScanMatch: what the last incremental update reported (its kernels have completed: the frame filter's wait on
// the same stream came after them).  A valid voxel outside the dense window -> the window is re-fitted; tables filling up
// -> rebuilt (larger).
int fe_lm_settle(lsa_front_end *fe) {
  int rc0;
  if ((rc0 = fe_map_join(fe))) return rc0;
  if (!fe->lm.pending) return LSA_OK;
  const int c = lm_check(fe->lm);   // (spins on the status block's epoch: the update's last kernel writes it)
  if (fe->lm.filtered && fe->lm_adopted) fe->target_points = (size_t)(fe->lm.h_status.p->live_f > 0 ? fe->lm.h_status.p->live_f : 0);
  if (c) lsa_ctx_unpark(fe->ctx);   // (inside a batch the next pass's kernel is already parked: the repair goes in front of its successor)
  if (c == 2) return fe_lm_rebuild(fe, fe->lm.filtered);
  if (c == 1) {
    int rc = lm_refit_window(fe->lm);
    if (rc == LSA_ERR_CAPACITY) { fe->lm_off = true; return fe_full_target(fe); }
    if (rc) return rc;
    return fe_lm_adopt(fe);
  }
  return LSA_OK;
}

// `added` is already in fe->frames; its cloud is still the caller's d_cloud (static weights reset to 1 on the way if reset_w)
int fe_rebuild_map(lsa_front_end *fe, KeyFrame *added, bool evicted, const float4 *d_cloud, bool reset_w) {
  lsa_context *ctx = fe->ctx;
  size_t sum = 0, biggest = 0;
  for (KeyFrame *f : fe->frames) { sum += f->n; biggest = std::max(biggest, f->n); }
  int rc;
  const bool relayout = biggest > fe->slot_cap || !fe->d_map.p;
  const bool want_filtered = (int)fe->frames.size() >= 10 && fe->cfg.use_map_filter;
  // the steady state of the incremental map: ONE launch takes the caller's cloud, keeps the key-frame's copy of it, writes its
  // transformed points into the slot and swaps it for the evicted key-frame in the persistent sums
  const bool steady = fe->incremental && !fe->lm_off && !relayout && fe->lm.valid && want_filtered == fe->lm.filtered;
  const bool own_stream = fe_map_ctx(fe) == fe->ctx_m;
  // A single Update may reach this point with its ScanMatch's final derivative pass still in flight (deferred: its sums only complete
  // the statistics, fe_update_filtered).  The steady update takes its rows between its two halves — nothing the pass reads is written
  // before (lm_finish_begin); every other way of building the target starts by taking them.
  if (!steady && (rc = ndt_collect_deferred(fe->ndt))) return rc;
  if (fe->incremental && !fe->lm_off) {
    // an update still in flight on the OTHER stream (the caller alternates between single Updates and batches) comes first
    if (!own_stream && (rc = fe_map_join(fe))) return rc;
    fe->lm.ctx = fe_map_ctx(fe);
  }
  if (own_stream) {
    // the map's stream behind whatever the registration's stream still holds (the caller's own producer of this scan, if it has
    // not been waited for by a frame filter; an update a batch left on the registration's stream; normally nothing: the event is
    // complete when it is recorded)
    lsa_ctx_unpark(ctx);   // (a kernel parked on the registration's stream — a caller who opted into the carry — would hold the event back)
    LSA_HIP(ctx, hipEventRecord(fe->ev_in, ctx->stream));
    LSA_HIP(ctx, hipStreamWaitEvent(fe->ctx_m->stream, fe->ev_in, 0));
  }
  if (!steady && added->n)
    hipLaunchKernelGGL(k_copy_w, dim3(lsa_div_up(added->n, kBlock)), dim3(kBlock), 0, fe_map_stream(fe), d_cloud, added->n, added->pts.p, reset_w ? 1 : 0);
  if (relayout) {  // first frame, or a cloud larger than any before: lay the buffer out anew
    fe->slot_cap = biggest + biggest / 8 + 1;
    // (the reserve may free the buffer an update in flight still writes: drained first — unless the buffer is large enough already, as
    // after lsa_front_end_reserve: the new layout is written on the stream the old one was)
    if (fe->d_map.cap < fe->slot_cap * (size_t)fe->cfg.local_frame_num && (rc = fe_map_drain(fe))) return rc;
    if ((rc = fe->d_map.reserve(ctx, fe->slot_cap * (size_t)fe->cfg.local_frame_num))) return rc;
    int k = 0;
    for (KeyFrame *f : fe->frames) { f->slot = k++; fe_transform_into_slot(fe, f); }
    // The fixed-point scales of every per-cell sum are a constant of the session from here on — sized for a full window of
    // full slots, whatever the clouds at hand hold — so that a full build and the incremental map quantise alike
    unsigned long long cap_pts = 2ull * fe->slot_cap * (unsigned long long)fe->cfg.local_frame_num, s = 1;
    while (s < cap_pts) s <<= 1;
    fe->scale_n = s;
    if (fe->map_filter) lsa_voxel_filter_set_scale_floor(fe->map_filter, s);
    ndt_set_scale_floor(fe->ndt, s);
    fe->lm.valid = false;
    if (fe->incremental && !fe->lm_off && (rc = lm_configure(fe->lm, fe_map_ctx(fe), fe->cfg.map_leaf, fe->cfg.ndt_res, /*pcl*/ 0, s))) return rc;
  }
  LSA_HIP(ctx, hipGetLastError());
  fe->map_points = sum;
  if (!fe->incremental || fe->lm_off) {
    if (!relayout) fe_transform_into_slot(fe, added);
    return fe_full_target(fe);
  }
  if (!steady) {
    if (!relayout) fe_transform_into_slot(fe, added);
    if (!want_filtered) fe->target_points = sum;
    return fe_lm_rebuild(fe, want_filtered, own_stream);
  }
  // the steady state: the evicted key-frame's points (still in the slot) leave the sums, the new key-frame's enter
  float4 *slot = fe->d_map.p + (size_t)added->slot * fe->slot_cap;
  float T[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T[r * 4 + c] = added->pose[c * 4 + r];
  if ((rc = lm_swap(fe->lm, slot, fe->slot_cap, evicted, d_cloud, added->n, T, added->pts.p, reset_w))) return rc;
  if (own_stream && fe->cloud_is_callers && fe->cloud_mode >= 1) {
    // (Only without a frame filter: with one, a device cloud reaches this point as the copy the filter's first kernel left in a
    // buffer of this handle, fe_update_device.)  lm_swap's one kernel is the last reader of the CALLER's cloud (it keeps the
    // key-frame's copy and writes the slot); it runs on the map's stream after this call has returned.  The registration's stream —
    // the caller's — is ordered behind it here, so that whatever the caller enqueues next on its stream (a refill of the same scan
    // buffer) comes after that read, while the rest of the update (dirty cells of both levels) still runs beside it.
    LSA_HIP(ctx, hipEventRecord(fe->ev_cloud, fe->ctx_m->stream));
    LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, fe->ev_cloud, 0));
  }
  if ((rc = lm_finish_begin(fe->lm))) return rc;
  // (the host has been launching for ~10 us: the deferred pass's rows are there, or as good as; should the pass have to be redone, the
  // redo still finds the records and the probe table it was posted for)
  if ((rc = ndt_collect_deferred(fe->ndt))) return rc;
  if ((rc = lm_finish_end(fe->lm))) return rc;
  if (own_stream) {
    LSA_HIP(ctx, hipEventRecord(fe->ev_map, fe->ctx_m->stream));
    fe->map_inflight = true;
  }
  if (!want_filtered) fe->target_points = sum;   // (filtered: the live-cell count arrives with the update's status)
  if (!fe->lm_adopted) return fe_lm_adopt(fe);
  ndt_touch_external(fe->ndt, fe->target_points);
  return LSA_OK;
}

int fe_add_key_frame(lsa_front_end *fe, const float4 *d_cloud, size_t n, const float pose[16], bool reset_w) {
  lsa_context *ctx = fe->ctx;
  KeyFrame *kf;
  bool evicted = false;
  if ((int)fe->frames.size() >= fe->cfg.local_frame_num && !fe->frames.empty()) {  // :364-372: slide the window
    kf = fe->frames.front();
    fe->frames.pop_front();
    evicted = true;   // its transformed points are still in its slot of the map buffer
  } else if (!fe->pool.empty()) {  // (the evicted frame's object is reused and with it its slot of the map buffer)
    kf = fe->pool.back();
    fe->pool.pop_back();
    kf->slot = (int)fe->frames.size();
  } else {
    kf = new KeyFrame();
    kf->slot = (int)fe->frames.size();
  }
  int rc = kf->pts.reserve(ctx, n, 1.1);
  if (rc) { fe->pool.push_back(kf); return rc; }
  kf->n = n;
  memcpy(kf->pose, pose, 16 * sizeof(float));
  fe->frames.push_back(kf);
  while ((int)fe->frames.size() > fe->cfg.local_frame_num) {
    fe->pool.push_back(fe->frames.front());
    fe->frames.pop_front();
  }
  return fe_rebuild_map(fe, kf, evicted, d_cloud, reset_w);
}

// Everything of Update behind the frame filter.  `st` is the CALLER's record and is written in place: inside a batch the
// registration may complete st->ndt after this returns (deferred final pass, ndt_api.hip).
int fe_update_filtered(lsa_front_end *fe, const float4 *d_cloud, size_t n, const float4 *src, size_t n_src, float pose_out[16], lsa_front_end_stats *st,
                       bool batch) {
  int rc;
  memset(st, 0, sizeof(*st));
  st->filtered_points = (int64_t)n_src;
  if (fe->frames.empty()) {  // first frame: key-frame 0 at the init pose, raw intensities kept (:203-218)
    if (batch) lsa_ctx_unpark(fe->ctx);
    if ((rc = fe_add_key_frame(fe, d_cloud, n, fe->init_pose, false))) return rc;
    memcpy(pose_out, fe->init_pose, 16 * sizeof(float));
    memcpy(fe->last_pose, fe->init_pose, sizeof(fe->last_pose));
    memcpy(fe->predict_pose, fe->init_pose, sizeof(fe->predict_pose));
    memcpy(fe->last_kf_pose, fe->init_pose, sizeof(fe->last_kf_pose));
    identity16(fe->step_pose);
    st->key_frame = 1;
  } else {
    float cur[16];
    if (fe->incremental && !fe->lm_off && (rc = fe_lm_settle(fe))) return rc;
    // A single Update defers its ScanMatch's final derivative pass as a batch does (the pose is known before that pass has run: its sums
    // only complete score / probability / pairs of st->ndt) — for the length of this function: if the scan becomes a key-frame, the
    // local-map update's first launches go out while the pass is on its way (fe_rebuild_map takes its rows before anything the pass
    // reads is rewritten); otherwise the rows are taken right below.  Either way st->ndt is complete when this returns.
    const bool single_defer = !batch && fe->defer_single;
    const bool prev_defer = single_defer ? ndt_defer_scope(fe->ndt, true) : false;
    rc = lsa_ndt_scan_match_device(fe->ndt, (const float *)src, n_src, fe->predict_pose, cur, &st->ndt);
    if (single_defer) (void)ndt_defer_scope(fe->ndt, prev_defer);
    if (rc) { if (single_defer) (void)ndt_collect_deferred(fe->ndt); return rc; }
    struct CollectAtExit {   // (every way out of this branch: st is the caller's record, the deferred pass writes into it)
      lsa_ndt *h; bool on;
      ~CollectAtExit() { if (on) (void)ndt_collect_deferred(h); }
    } collect_at_exit{fe->ndt, single_defer};
    memcpy(pose_out, cur, sizeof(cur));
    float inv_last[16];
    if (mat4_inverse(fe->last_pose, inv_last)) mat4_mul(inv_last, cur, fe->step_pose);  // :238
    mat4_mul(cur, fe->step_pose, fe->predict_pose);                                      // :239
    memcpy(fe->last_pose, cur, sizeof(cur));                                             // :240
    float l1 = std::fabs(fe->last_kf_pose[12] - cur[12]) + std::fabs(fe->last_kf_pose[13] - cur[13]) + std::fabs(fe->last_kf_pose[14] - cur[14]);
    if (l1 > fe->cfg.key_frame_distance) {  // :243-245
      // Inside a batch: this ScanMatch's final derivative pass is still in flight (deferred, ndt_api.hip).  Should it have to be redone,
      // the redo must see the table it was posted for, not the one the key-frame is about to change in place: take its rows first
      // (they are as good as there: the pass was posted before the pose came back).  Then the next pass's parked kernel is released:
      // the local-map kernels go in front of its successor.
      if (batch && (rc = ndt_collect_deferred(fe->ndt))) return rc;
      if (batch) lsa_ctx_unpark(fe->ctx);
      // the reference resets every point's static weight to 1.0 before storing a later key-frame (:304-306)
      if ((rc = fe_add_key_frame(fe, d_cloud, n, cur, fe->cfg.reset_intensity != 0))) return rc;
      memcpy(fe->last_kf_pose, cur, sizeof(cur));
      st->key_frame = 1;
      // ... and that successor is enqueued at once, behind the event that ends the update on the map's stream: its launch and dispatch
      // pass while the map is updated.  (A scan-by-scan caller's stream is ordered behind the update later, in front of the next
      // ScanMatch: the next scan's frame filter goes first and runs beside the update.)
      if (batch && (rc = fe_map_join(fe))) return rc;
      if (batch && fe->lm_adopted) ndt_prearm(fe->ndt);
    }
    if (single_defer && (rc = ndt_collect_deferred(fe->ndt))) return rc;   // (not a key-frame: the rows are taken here)
  }
  fe->frames_seen++;
  st->local_map_frames = (int32_t)fe->frames.size();
  st->local_map_points = (int64_t)fe->map_points;
  st->target_points = (int64_t)fe->target_points;
  return LSA_OK;
}

// callers_buffer: d_cloud belongs to the caller (lsa_front_end_update_device), who gets it back when the call returns — as
// FrontEnd::Update(const CloudData &) leaves its argument alone afterwards.  A key-frame's map update reads the raw cloud AFTER the
// call has returned (on the map's stream, beside the next scan's frame filter), so the frame filter's first kernel, which has every
// row in registers anyway, leaves a copy in a buffer of this handle (two, alternating: the update that reads one is seen complete
// by the next call, which fills the other) and everything behind the filter works on that copy: 2 MB of stores per scan inside a
// latency-bound kernel instead of an event that holds the next scan's filter back behind the update's first kernel (measured:
// profiles/r05_fe_cloud_ab.log).  Without a frame filter the event is what orders the caller's stream (fe_rebuild_map).
void fe_prearm_cb(void *user) {
  lsa_front_end *fe = (lsa_front_end *)user;
  if (fe_map_join(fe) == LSA_OK) ndt_prearm(fe->ndt);
}

int fe_update_device(lsa_front_end *fe, const float4 *d_cloud, size_t n, float pose_out[16], lsa_front_end_stats *stats, bool callers_buffer) {
  lsa_context *ctx = fe->ctx;
  int rc;
  // frame filter (:106-107)
  const float4 *src = d_cloud;
  size_t n_src = n;
  fe->cloud_is_callers = callers_buffer;
  if (fe->cfg.use_frame_filter) {
    if ((rc = fe->d_filtered.reserve(ctx, n ? n : 1, 1.25))) return rc;
    size_t V = 0;
    float4 *keep_p = nullptr;
    if (callers_buffer && n && fe->incremental && !fe->lm_off && fe->cloud_mode == 2) {
      DevBuf<float4> &keep = fe->in_alt ? fe->d_in_alt : fe->d_in;
      fe->in_alt = !fe->in_alt;
      if (n > keep.cap) {   // before the reserve: it frees a buffer an earlier copy / update may still read
        LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = fe_map_drain(fe))) return rc;
        if ((rc = keep.reserve(ctx, n, 1.25))) return rc;
      }
      keep_p = keep.p;
    }
    // the filter's launches, then — where a ScanMatch follows — the registration's first pass kernel behind them, then the wait
    if ((rc = lsa_voxel_filter_enqueue_device(fe->frame_filter, (const float *)d_cloud, n, (float *)fe->d_filtered.p, (float *)keep_p))) return rc;
    if (fe->prearm_single && n && !fe->frames.empty()) {
      // (the pass reads the table: the stream goes behind a map update still in flight first, as fe_lm_settle would order it)
      if ((rc = fe_map_join(fe))) return rc;
      ndt_prearm(fe->ndt);
    }
    rc = lsa_voxel_filter_wait_parked(fe->frame_filter, &V);
    if (keep_p && (rc == LSA_OK || rc == LSA_ERR_RANGE)) { d_cloud = keep_p; fe->cloud_is_callers = false; }
    if (rc == LSA_OK) { src = fe->d_filtered.p; n_src = V; }
    else if (rc != LSA_ERR_RANGE) return rc;
    else if (src != d_cloud) src = d_cloud;   // (PCL passes the input through: the copy, if there is one)
  }
  lsa_front_end_stats local;
  return fe_update_filtered(fe, d_cloud, n, src, n_src, pose_out, stats ? stats : &local, false);
}

// A run of scans (replay of a recorded drive, or a sensor queue that has backed up) through Update with the frame filter of
// scan k + 1 running on a second context's stream while scan k is registered and, on key-frames, while the local map is
// updated — the pose-independent part of the next Update overlaps the pose-dependent part of this one.  Inside the run the
// next pass's kernel stays parked between two ScanMatches and the final derivative pass of a ScanMatch, which only completes
// the statistics, is collected behind the next scan's first mail (both as in lsa_pipeline_scan_match).  Poses, key-frame
// decisions, maps and statistics are those of `count` single Updates, bit for bit.
int fe_update_batch(lsa_front_end *fe, size_t count, const float *const *d_scans, const size_t *n_points, float *poses_out, lsa_front_end_stats *stats) {
  lsa_context *ctx = fe->ctx;
  int rc;
  if (!fe->cfg.use_frame_filter) {  // nothing to overlap
    for (size_t k = 0; k < count; k++)
      if ((rc = fe_update_device(fe, (const float4 *)d_scans[k], n_points[k], poses_out + 16 * k, stats ? stats + k : nullptr, true))) return rc;
    return LSA_OK;
  }
  if ((rc = fe_batch_context(fe))) return rc;
  size_t n_max = 1;
  for (size_t k = 0; k < count; k++) n_max = std::max(n_max, n_points[k]);
  // before the reserves (they may free a buffer): neither stream may still read the old ones
  if (fe->d_batch[0].cap < n_max) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_b->stream));
    for (int b = 0; b < 3; b++)
      if ((rc = fe->d_batch[b].reserve(ctx, n_max, 1.25))) return rc;
  }
  lsa_voxel_filter *ff = fe->frame_filter_b;
  // the scans were produced on the caller's stream = this context's (lidar_slam_amd.h): the second stream starts behind it
  LSA_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  LSA_HIP(ctx, hipStreamWaitEvent(fe->ctx_b->stream, ctx->ev1, 0));
  if ((rc = lsa_voxel_filter_apply_device_async_here(ff, d_scans[0], n_points[0], (float *)fe->d_batch[0].p))) return lsa_fail(ctx, rc, "%s", lsa_last_error(fe->ctx_b));
  const bool carry_user = ndt_carry_scope(fe->ndt, true);
  const bool defer_user = ndt_defer_scope(fe->ndt, true);
  fe->in_batch = true;
  struct Restore {
    lsa_front_end *fe; lsa_ndt *h; bool prev, prev_defer;
    ~Restore() { fe->in_batch = false; (void)ndt_collect_deferred(h); (void)ndt_defer_scope(h, prev_defer); (void)ndt_carry_scope(h, prev); if (!prev) ndt_release_parked(h); }
  } restore{fe, fe->ndt, carry_user, defer_user};
  lsa_front_end_stats local;
  for (size_t k = 0; k < count; k++) {
    size_t m = 0;
    rc = lsa_voxel_filter_wait(ff, &m);
    const float4 *src = fe->d_batch[k % 3].p;
    if (rc == LSA_ERR_RANGE) { src = (const float4 *)d_scans[k]; m = n_points[k]; rc = LSA_OK; }   // PCL passes the input through
    if (rc) return lsa_fail(ctx, rc, "%s", lsa_last_error(fe->ctx_b));
    if (k + 1 < count && (rc = lsa_voxel_filter_apply_device_async(ff, d_scans[k + 1], n_points[k + 1], (float *)fe->d_batch[(k + 1) % 3].p)))
      return lsa_fail(ctx, rc, "%s", lsa_last_error(fe->ctx_b));
    rc = fe_update_filtered(fe, (const float4 *)d_scans[k], n_points[k], src, m, poses_out + 16 * k, stats ? stats + k : &local, true);
    if (!stats && (rc == LSA_OK)) rc = ndt_collect_deferred(fe->ndt);   // (`local` is about to be reused)
    if (rc) {  // leave no filter in flight behind an error
      if (k + 1 < count) { size_t dummy; (void)lsa_voxel_filter_wait(ff, &dummy); }
      return rc;
    }
  }
  // the caller's scans are his again when this returns: whatever he enqueues on his stream next comes behind the last map update
  return fe_map_join(fe);
}

// The same run fed with HOST scans (what FrontEndFlow::Run's buffered clouds are, front_end_flow.cpp:18-36): scan k + 1 is packed by the
// second filter's helper threads and pulled over the link by that filter's first kernel (voxel_filter.hip: job kind 5) while scan k is
// registered — the 44 us a 131 k-point scan needs on the link, and the packing, pass under the ~48 us of the registration.  The raw rows
// land in one of three device buffers (a key-frame's cloud is copied from there), the filtered cloud in d_batch as above.
int fe_update_batch_host(lsa_front_end *fe, size_t count, const void *const *h_scans, const size_t *n_points, size_t stride, size_t w_off, float *poses_out,
                         lsa_front_end_stats *stats) {
  lsa_context *ctx = fe->ctx;
  int rc;
  if (!fe->cfg.use_frame_filter) {  // nothing to overlap with: one Update per scan
    for (size_t k = 0; k < count; k++)
      if ((rc = lsa_front_end_update(fe, h_scans[k], n_points[k], stride, w_off, poses_out + 16 * k, stats ? stats + k : nullptr))) return rc;
    return LSA_OK;
  }
  if ((rc = fe_batch_context(fe)) || (rc = fe_host_batch_context(fe))) return rc;
  size_t n_max = 1;
  for (size_t k = 0; k < count; k++) n_max = std::max(n_max, n_points[k]);
  if (fe->d_batch[3].cap < n_max || fe->d_raw[3].cap < n_max) {   // before the reserves (they may free a buffer): no stream may still use the old ones
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_b->stream));
    LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_c->stream));
    if ((rc = fe_map_drain(fe))) return rc;
    for (int b = 0; b < 4; b++)
      if ((rc = fe->d_batch[b].reserve(ctx, n_max, 1.25)) || (rc = fe->d_raw[b].reserve(ctx, n_max, 1.25))) return rc;
  }
  // scan j goes through filter j % 2 into buffers j % 4: k - 1 (its deferred final pass), k (being registered), k + 1 and k + 2 (in
  // flight) never share one
  lsa_voxel_filter *ff[2] = {fe->frame_filter_b, fe->frame_filter_c};
  lsa_context *fc[2] = {fe->ctx_b, fe->ctx_c};
  auto post = [&](size_t j) -> int {
    int r = lsa_voxel_filter_apply_host_async(ff[j & 1], h_scans[j], n_points[j], stride, w_off, (float *)fe->d_raw[j & 3].p, (float *)fe->d_batch[j & 3].p);
    return r ? lsa_fail(ctx, r, "%s", lsa_last_error(fc[j & 1])) : LSA_OK;
  };
  size_t posted = 0;
  auto drain = [&](size_t from) { for (size_t j = from; j < posted; j++) { size_t dummy; (void)lsa_voxel_filter_wait(ff[j & 1], &dummy); } };
  for (; posted < count && posted < 2; posted++)
    if ((rc = post(posted))) { drain(0); return rc; }
  const bool carry_user = ndt_carry_scope(fe->ndt, true);
  const bool defer_user = ndt_defer_scope(fe->ndt, true);
  fe->in_batch = true;
  struct Restore {
    lsa_front_end *fe; lsa_ndt *h; bool prev, prev_defer;
    ~Restore() { fe->in_batch = false; (void)ndt_collect_deferred(h); (void)ndt_defer_scope(h, prev_defer); (void)ndt_carry_scope(h, prev); if (!prev) ndt_release_parked(h); }
  } restore{fe, fe->ndt, carry_user, defer_user};
  lsa_front_end_stats local;
  // diagnostics (LSA_PIPE_TRACE=1, runs of <= 64 scans): when each scan's filter was there and its Update returned, on stderr
  static const bool trace = getenv("LSA_PIPE_TRACE") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  std::vector<double> tr;
  auto stamp = [&] { if (trace && count <= 64) tr.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count()); };
  struct Dump {
    std::vector<double> &t;
    ~Dump() {
      if (t.empty()) return;
      fprintf(stderr, "[lsa fe host run] us since the loop began (wait begins / filter there / next posted / Update back):");
      for (size_t i = 0; i + 3 < t.size(); i += 4) fprintf(stderr, " %.0f/%.0f/%.0f/%.0f", t[i], t[i + 1], t[i + 2], t[i + 3]);
      fprintf(stderr, "\n");
    }
  } dump{tr};
  for (size_t k = 0; k < count; k++) {
    size_t m = 0;
    stamp();
    rc = lsa_voxel_filter_wait(ff[k & 1], &m);
    stamp();
    const float4 *raw = fe->d_raw[k & 3].p;
    const float4 *src = fe->d_batch[k & 3].p;
    if (rc == LSA_ERR_RANGE) { src = raw; m = n_points[k]; rc = LSA_OK; }   // PCL passes the input through
    if (rc) { rc = lsa_fail(ctx, rc, "%s", lsa_last_error(fc[k & 1])); drain(k + 1); return rc; }
    if (posted < count) {
      if ((rc = post(posted))) { drain(k + 1); return rc; }
      posted++;
    }
    stamp();
    rc = fe_update_filtered(fe, raw, n_points[k], src, m, poses_out + 16 * k, stats ? stats + k : &local, true);
    stamp();
    if (!stats && (rc == LSA_OK)) rc = ndt_collect_deferred(fe->ndt);   // (`local` is about to be reused)
    if (rc) { drain(k + 1); return rc; }   // leave no filter in flight behind an error
  }
  return fe_map_join(fe);
}
}  // namespace

extern "C" int lsa_front_end_create(lsa_front_end **out, lsa_context *ctx, const lsa_front_end_config *cfg) {
  if (!out || !ctx || !cfg) { lsa_set_global_error("lsa_front_end_create: null argument"); return LSA_ERR_INVALID; }
  *out = nullptr;
  if (cfg->local_frame_num < 1) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_front_end_create: local_frame_num must be >= 1");
  lsa_front_end *fe = new lsa_front_end();
  fe->ctx = ctx;
  fe->cfg = *cfg;
  int rc = LSA_OK;
  if (cfg->use_frame_filter) rc = lsa_voxel_filter_create(&fe->frame_filter, ctx, cfg->frame_leaf[0], cfg->frame_leaf[1], cfg->frame_leaf[2]);
  if (!rc && cfg->use_map_filter) rc = lsa_voxel_filter_create(&fe->map_filter, ctx, cfg->map_leaf[0], cfg->map_leaf[1], cfg->map_leaf[2]);
  if (!rc) rc = lsa_ndt_create(&fe->ndt, ctx, cfg->ndt_res, cfg->ndt_step_size, cfg->ndt_trans_eps, cfg->ndt_max_iter, cfg->ndt_outlier_ratio);
  // every Update puts the frame filter (and, on key-frames, the local-map work) on this context's stream between two
  // ScanMatches: an armed kernel parked across them would only have to be released again
  if (!rc) rc = lsa_ndt_set_armed_carry(fe->ndt, 0);
  if (!rc && (rc = lsa_context_create(&fe->ctx_m, ctx->device, nullptr))) rc = lsa_fail(ctx, rc, "lsa_front_end_create: the local map's context: %s", lsa_last_global_error());
  if (!rc && (hipEventCreateWithFlags(&fe->ev_map, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&fe->ev_in, hipEventDisableTiming) != hipSuccess ||
              hipEventCreateWithFlags(&fe->ev_cloud, hipEventDisableTiming) != hipSuccess))
    rc = lsa_fail(ctx, LSA_ERR_HIP, "lsa_front_end_create: hipEventCreate failed");
  if (rc) { lsa_front_end_destroy(fe); return rc; }
  identity16(fe->init_pose); identity16(fe->last_pose); identity16(fe->predict_pose); identity16(fe->step_pose); identity16(fe->last_kf_pose);
  // buffers for scans of a 64-beam sensor (<= 160 k points per revolution) are allocated here rather than under the first scans
  // (lsa_front_end_reserve; LSA_FE_RESERVE_POINTS=0 switches it off, another value sizes it; larger scans still just grow them)
  const size_t reserve_points = getenv("LSA_FE_RESERVE_POINTS") ? (size_t)atoll(getenv("LSA_FE_RESERVE_POINTS")) : (size_t)160000;
  if ((rc = lsa_front_end_reserve(fe, reserve_points))) { lsa_front_end_destroy(fe); return rc; }
  *out = fe;
  return LSA_OK;
}

// the second context (stream) and frame filter of lsa_front_end_update_batch_device
static int fe_batch_context(lsa_front_end *fe) {
  if (fe->ctx_b) return LSA_OK;
  int rc;
  if ((rc = lsa_context_create(&fe->ctx_b, fe->ctx->device, nullptr))) return lsa_fail(fe->ctx, rc, "lsa_front_end_update_batch_device: second context: %s", lsa_last_global_error());
  return lsa_voxel_filter_create(&fe->frame_filter_b, fe->ctx_b, fe->cfg.frame_leaf[0], fe->cfg.frame_leaf[1], fe->cfg.frame_leaf[2]);
}

// ... and the third one of lsa_front_end_update_batch (host scans, ingest two deep)
static int fe_host_batch_context(lsa_front_end *fe) {
  if (fe->ctx_c) return LSA_OK;
  int rc;
  if ((rc = lsa_context_create(&fe->ctx_c, fe->ctx->device, nullptr))) return lsa_fail(fe->ctx, rc, "lsa_front_end_update_batch: third context: %s", lsa_last_global_error());
  return lsa_voxel_filter_create(&fe->frame_filter_c, fe->ctx_c, fe->cfg.frame_leaf[0], fe->cfg.frame_leaf[1], fe->cfg.frame_leaf[2]);
}

int lsa_voxel_filter_reserve(lsa_voxel_filter *f, size_t max_points, int async_too);   // voxel_filter.hip

// Allocate now everything Update would otherwise allocate on its way through the first scans of a session: the frame filters'
// index buffers and bitmaps, the filtered-cloud buffers, the key-frame clouds of a whole window, the local-map buffer, the
// incremental map's tables / dirty lists / dense window, the registration's mailbox and row buffers, the batch entry point's second
// stream and helper thread.  Measured on the 30-scan drive of bench.py: scan 0 costs 450 us, scan 1 340 us and the tenth key-frame
// 590 us of which ~150 us each are allocation calls, and every key-frame of the first window pays a 10 us hipMalloc.  None of this
// changes a result: buffers are sized, not laid out (the slot size and the fixed-point scales still come from the first scan).
extern "C" int lsa_front_end_reserve(lsa_front_end *fe, size_t max_scan_points) {
  if (!fe) return LSA_ERR_INVALID;
  lsa_context *ctx = fe->ctx;
  LSA_HIP(ctx, lsa_enter(ctx));
  if (max_scan_points == 0) return LSA_OK;
  const size_t n = max_scan_points;
  int rc;
  if (fe->cfg.use_frame_filter) {
    if ((rc = lsa_voxel_filter_reserve(fe->frame_filter, n, 3))) return rc;   // (lsa_front_end_update's host scans go through its pulled path)
    if ((rc = fe->d_filtered.reserve(ctx, n, 1.25))) return rc;
    // the two buffers a scan's raw rows are kept in (device clouds: the filter's copy; host clouds: where the pulled rows land)
    if ((rc = fe->d_in.reserve(ctx, n, 1.25)) || (rc = fe->d_in_alt.reserve(ctx, n, 1.25))) return rc;
    if ((rc = fe_batch_context(fe))) return rc;
    if ((rc = lsa_voxel_filter_reserve(fe->frame_filter_b, n, 3))) return lsa_fail(ctx, rc, "%s", lsa_last_error(fe->ctx_b));
    if ((rc = fe_host_batch_context(fe))) return rc;
    if ((rc = lsa_voxel_filter_reserve(fe->frame_filter_c, n, 3))) return lsa_fail(ctx, rc, "%s", lsa_last_error(fe->ctx_c));
    if (fe->d_batch[3].cap < n || fe->d_raw[3].cap < n) {
      LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_b->stream));
      for (int b = 0; b < 4; b++)
        if ((rc = fe->d_batch[b].reserve(ctx, n, 1.25)) || (rc = fe->d_raw[b].reserve(ctx, n, 1.25))) return rc;
    }
  }
  // key-frame clouds: one per slot of the window (an evicted key-frame's object and buffer are reused from then on)
  while (fe->frames.size() + fe->pool.size() < (size_t)fe->cfg.local_frame_num) {
    KeyFrame *kf = new KeyFrame();
    if ((rc = kf->pts.reserve(ctx, n, 1.1))) { delete kf; return rc; }
    fe->pool.push_back(kf);
  }
  if (!fe->d_map.p && (rc = fe->d_map.reserve(ctx, (n + n / 8 + 1) * (size_t)fe->cfg.local_frame_num))) return rc;
  if (fe->incremental && !fe->lm_off && (rc = lm_reserve(fe->lm, fe->ctx_m, fe->cfg.use_map_filter != 0))) return rc;
  if ((rc = ndt_reserve(fe->ndt))) return rc;
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_m->stream));
  LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_b ? fe->ctx_b->stream : ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(fe->ctx_c ? fe->ctx_c->stream : ctx->stream));
  return LSA_OK;
}

extern "C" void lsa_front_end_destroy(lsa_front_end *fe) {
  if (!fe) return;
  (void)lsa_enter(fe->ctx);
  (void)hipStreamSynchronize(fe->ctx->stream);
  if (fe->ctx_m) (void)hipStreamSynchronize(fe->ctx_m->stream);
  for (KeyFrame *f : fe->frames) { f->pts.release(); delete f; }
  for (KeyFrame *f : fe->pool) { f->pts.release(); delete f; }
  fe->d_in.release(); fe->d_filtered.release(); fe->d_map.release(); fe->d_map_f.release(); fe->h_stage.release();
  fe->lm.release();
  lsa_voxel_filter_destroy(fe->frame_filter);
  lsa_voxel_filter_destroy(fe->map_filter);
  lsa_ndt_destroy(fe->ndt);
  if (fe->ctx_b) (void)hipStreamSynchronize(fe->ctx_b->stream);
  lsa_voxel_filter_destroy(fe->frame_filter_b);
  for (int b = 0; b < 4; b++) { fe->d_batch[b].release(); fe->d_raw[b].release(); }
  if (fe->ctx_c) (void)hipStreamSynchronize(fe->ctx_c->stream);
  lsa_voxel_filter_destroy(fe->frame_filter_c);
  if (fe->ctx_c) lsa_context_destroy(fe->ctx_c);
  if (fe->ctx_b) lsa_context_destroy(fe->ctx_b);
  if (fe->ev_map) (void)hipEventDestroy(fe->ev_map);
  if (fe->ev_in) (void)hipEventDestroy(fe->ev_in);
  if (fe->ev_cloud) (void)hipEventDestroy(fe->ev_cloud);
  fe->d_in_alt.release();
  if (fe->ctx_m) lsa_context_destroy(fe->ctx_m);
  delete fe;
}

extern "C" int lsa_front_end_set_init_pose(lsa_front_end *fe, const float pose[16]) {  // FrontEnd::SetInitPose (:343-346)
  if (!fe || !pose) return LSA_ERR_INVALID;
  memcpy(fe->init_pose, pose, sizeof(fe->init_pose));
  return LSA_OK;
}

extern "C" int lsa_front_end_update_device(lsa_front_end *fe, const float *d_xyzw, size_t n, float pose_out[16], lsa_front_end_stats *stats) {
  if (!fe || !pose_out || (n && !d_xyzw)) return fe ? lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_device: null argument") : LSA_ERR_INVALID;
  LSA_HIP(fe->ctx, lsa_enter(fe->ctx));
  return fe_update_device(fe, (const float4 *)d_xyzw, n, pose_out, stats, true);
}

extern "C" int lsa_front_end_update_batch(lsa_front_end *fe, size_t count, const void *const *clouds, const size_t *n_points, size_t stride, size_t w_off,
                                          float *poses_out, lsa_front_end_stats *stats) {
  if (!fe || (count && (!clouds || !n_points || !poses_out))) return fe ? lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_batch: null argument") : LSA_ERR_INVALID;
  for (size_t k = 0; k < count; k++)
    if (n_points[k] && !clouds[k]) return lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_batch: scan %zu is null", k);
  if (stride < 12 || (w_off != SIZE_MAX && w_off + 4 > stride)) return lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_batch: point rows of %zu bytes cannot hold xyz + an intensity at offset %zu", stride, w_off);
  if (count == 0) return LSA_OK;
  LSA_HIP(fe->ctx, lsa_enter(fe->ctx));
  return fe_update_batch_host(fe, count, clouds, n_points, stride, w_off, poses_out, stats);
}

extern "C" int lsa_front_end_update_batch_device(lsa_front_end *fe, size_t count, const float *const *d_scans_xyzw, const size_t *n_points, float *poses_out,
                                                 lsa_front_end_stats *stats) {
  if (!fe || (count && (!d_scans_xyzw || !n_points || !poses_out))) return fe ? lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_batch_device: null argument") : LSA_ERR_INVALID;
  for (size_t k = 0; k < count; k++)
    if (n_points[k] && !d_scans_xyzw[k]) return lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update_batch_device: scan %zu is null", k);
  if (count == 0) return LSA_OK;
  LSA_HIP(fe->ctx, lsa_enter(fe->ctx));
  return fe_update_batch(fe, count, d_scans_xyzw, n_points, poses_out, stats);
}

extern "C" int lsa_front_end_update(lsa_front_end *fe, const void *cloud, size_t n, size_t stride, size_t w_off, float pose_out[16],
                                    lsa_front_end_stats *stats) {
  if (!fe || !pose_out || (n && (!cloud || stride < 12))) return fe ? lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_update: bad argument") : LSA_ERR_INVALID;
  lsa_context *ctx = fe->ctx;
  LSA_HIP(ctx, lsa_enter(ctx));
  int rc;
  // The upload alternates between two device buffers: a key-frame's map update reads the scan on the map's own stream after this
  // call has returned, and it is the NEXT call that sees that update complete (fe_lm_settle) — the one after it may overwrite.
  DevBuf<float4> &din = fe->in_alt ? fe->d_in_alt : fe->d_in;
  fe->in_alt = !fe->in_alt;
  if (n > din.cap || (n ? n : 1) * 4 > fe->h_stage.cap) {   // before the reserves: they free a buffer an earlier copy / update may still read
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = fe_map_drain(fe))) return rc;
  }
  if ((rc = din.reserve(ctx, n ? n : 1, 1.25))) return rc;
  // From the second scan on the frame filter's first kernel PULLS the packed rows out of pinned memory while three threads are still
  // packing them (voxel_filter.hip: the host-pointer Filter's path), leaves the raw cloud in `din` and the filtered one in
  // d_filtered: no copy command, the transfer under the packing, the mark under the transfer (238 -> ~150 us per Update of a
  // 131 k-point host scan).
  if (fe->cfg.use_frame_filter && n) {
    if ((rc = fe->d_filtered.reserve(ctx, n, 1.25))) return rc;
    size_t V = 0;
    int pulled = 0;
    // (behind the pulled build's kernels, before this thread joins the packing: the registration's first pass kernel, as in fe_update_device)
    rc = lsa_voxel_filter_apply_host_to_device(fe->frame_filter, cloud, n, stride, w_off, (float *)din.p, (float *)fe->d_filtered.p, &V, &pulled,
                                               (fe->prearm_single && !fe->frames.empty()) ? fe_prearm_cb : nullptr, fe);
    if (pulled) {
      if (rc && rc != LSA_ERR_RANGE) return rc;
      lsa_front_end_stats local;
      if (rc == LSA_ERR_RANGE) return fe_update_filtered(fe, din.p, n, din.p, n, pose_out, stats ? stats : &local, false);   // PCL passes the input through
      return fe_update_filtered(fe, din.p, n, fe->d_filtered.p, V, pose_out, stats ? stats : &local, false);
    }
    if (rc) return rc;
  }
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the staging buffer may still feed an earlier copy)
  if ((rc = fe->h_stage.reserve(ctx, (n ? n : 1) * 4))) return rc;
  if ((rc = lsa_upload_xyzw(ctx, cloud, n, stride, w_off, fe->h_stage.p, din.p))) return rc;
  return fe_update_device(fe, din.p, n, pose_out, stats, false);
}

extern "C" lsa_ndt *lsa_front_end_registration(lsa_front_end *fe) { return fe ? fe->ndt : nullptr; }
extern "C" lsa_voxel_filter *lsa_front_end_frame_filter(lsa_front_end *fe) { return fe ? fe->frame_filter : nullptr; }

extern "C" int lsa_front_end_set_incremental(lsa_front_end *fe, int on) {
  if (!fe) return LSA_ERR_INVALID;
  LSA_HIP(fe->ctx, lsa_enter(fe->ctx));
  const bool want = on != 0;
  if (want == fe->incremental) return LSA_OK;
  int rc0;
  if ((rc0 = fe_map_drain(fe))) return rc0;   // (the two settings write the map buffer on different streams)
  LSA_HIP(fe->ctx, hipStreamSynchronize(fe->ctx->stream));
  fe->incremental = want;
  fe->lm.valid = false;
  fe->lm_off = false;
  // the target as the new setting builds it (same voxels either way: the two paths agree to the bit)
  if (fe->frames.empty()) return LSA_OK;
  if (!want) return fe->lm_adopted ? fe_full_target(fe) : LSA_OK;
  int rc = lm_configure(fe->lm, fe->ctx_m, fe->cfg.map_leaf, fe->cfg.ndt_res, 0, fe->scale_n);
  if (rc) return rc;
  return fe_lm_rebuild(fe, (int)fe->frames.size() >= 10 && fe->cfg.use_map_filter);
}

// Verification for the tests: a FULL build of the target (filter + voxel table) on scratch handles from the very buffer the
// incremental map was fed from, compared cell by cell, bit by bit, with the incremental map's level N.
extern "C" int lsa_front_end_map_verify(lsa_front_end *fe, uint64_t out[6]) {
  if (!fe || !out) return LSA_ERR_INVALID;
  lsa_context *ctx = fe->ctx;
  LSA_HIP(ctx, lsa_enter(ctx));
  memset(out, 0, 6 * sizeof(uint64_t));
  if (!fe->incremental || fe->lm_off || !fe->lm.valid) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_front_end_map_verify: no incremental map to verify");
  int rc;
  if ((rc = fe_lm_settle(fe))) return rc;
  lsa_voxel_filter *sf = nullptr;
  lsa_ndt *sn = nullptr;
  if ((rc = lsa_ndt_create(&sn, ctx, fe->cfg.ndt_res, fe->cfg.ndt_step_size, fe->cfg.ndt_trans_eps, fe->cfg.ndt_max_iter, fe->cfg.ndt_outlier_ratio))) return rc;
  ndt_set_scale_floor(sn, fe->scale_n);
  const size_t total = fe->slot_cap * fe->frames.size();
  DevBuf<float4> tmp;
  if (fe->lm.filtered) {
    rc = lsa_voxel_filter_create(&sf, ctx, fe->cfg.map_leaf[0], fe->cfg.map_leaf[1], fe->cfg.map_leaf[2]);
    if (!rc) lsa_voxel_filter_set_scale_floor(sf, fe->scale_n);
    size_t V = 0;
    if (!rc) rc = tmp.reserve(ctx, total);
    if (!rc) rc = lsa_voxel_filter_apply_device(sf, (const float *)fe->d_map.p, total, (float *)tmp.p, &V);
    if (!rc) rc = lsa_ndt_set_input_target_device(sn, (const float *)tmp.p, V);
    out[4] = V;
    out[5] = (uint64_t)(fe->lm.h_status.p->live_f > 0 ? fe->lm.h_status.p->live_f : 0);
  } else {
    rc = lsa_ndt_set_input_target_device(sn, (const float *)fe->d_map.p, total);
  }
  unsigned long long res[4] = {0, 0, 0, 0};
  if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = LSA_ERR_HIP;   // the scratch build (this stream) before the comparison (the map's)
  if (!rc) rc = lm_verify(fe->lm, sn->vi.cell_of_rank.p, sn->rec.p, sn->state.p, sn->vi.nocc, sn->vi.map, res);
  for (int k = 0; k < 4; k++) out[k] = res[k];
  lsa_ndt_destroy(sn);
  if (sf) lsa_voxel_filter_destroy(sf);
  tmp.release();
  return rc;
}

// diagnostics (profiles/scripts/lm_stamps.py): phase stamps of the local-map update's three kernels
extern "C" int lsa_front_end_map_stamps(lsa_front_end *fe, int enable, uint64_t *out, size_t cap, size_t *per_kernel) {
  if (!fe) return LSA_ERR_INVALID;
  LSA_HIP(fe->ctx, lsa_enter(fe->ctx));
  if (!fe->lm.ctx) return lsa_fail(fe->ctx, LSA_ERR_INVALID, "lsa_front_end_map_stamps: the incremental map is not configured yet");
  int rc;
  if ((rc = fe_map_drain(fe))) return rc;
  if (out) return lm_stamps_fetch(fe->lm, (unsigned long long *)out, cap, per_kernel);
  return lm_stamps_enable(fe->lm, enable != 0);
}

extern "C" int lsa_front_end_map_counters(lsa_front_end *fe, uint64_t out[6]) {
  if (!fe || !out) return LSA_ERR_INVALID;
  out[0] = fe->n_full_builds; out[1] = fe->lm.updates; out[2] = fe->lm.rebuilds; out[3] = fe->lm.refits;
  out[4] = fe->lm_adopted ? 1 : 0; out[5] = fe->lm_off ? 1 : 0;
  return LSA_OK;
}
