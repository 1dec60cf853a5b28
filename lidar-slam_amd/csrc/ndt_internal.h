// ndt_internal.h — internal state of an lsa_ndt handle.
#ifndef LSA_NDT_INTERNAL_H_
#define LSA_NDT_INTERNAL_H_

#include <chrono>
#include <cstdlib>
#include "nn_grid.h"
#include "voxel_index_dev.h"

// per occupied cell: centroid(3), inverse covariance symmetric (xx,xy,xz,yy,yz,zz), mean static weight
constexpr int kRecDoubles = 10;
// per-pass result: [score, g(6), H upper triangle (21), #pairs, #points, 0, 0]
constexpr int kOutDoubles = 32;

struct FitnessState;  // ndt_fitness.hip

struct lsa_ndt {
  lsa_context *ctx = nullptr;
  float resolution = 1.0f;
  float step_size = 0.1f, trans_eps = 0.01f, outlier_ratio = 0.55f;
  int max_iter = 30;
  int mode = 0;  // LSA_NDT_MODE_MANUAL / LSA_NDT_MODE_PCL
  double d1 = 0, d2 = 0;

  // target
  VoxelIndex vi;
  DevBuf<float4> d_target;      // only when set from host
  DevBuf<double> rec;           // [nocc][kRecDoubles]
  DevBuf<int32_t> state;        // [nocc] 0 / 1 / -1
  DevBuf<uint32_t> vbits;       // valid bitmap
  bool vbits_zero = false;      // ... known to be all zero (left so by the last build's k_pack_words) over vbits_zero_cap words
  size_t vbits_zero_cap = 0;
  DevBuf<uint4> words;          // {occ bits, valid bits, prefix, 0} per 32 cells
  DevBuf<float4> dense;         // dense probe table {centroid - cell corner (f32 x3), record rank or -1} per cell (+2 pad)
  bool use_dense = false;       // built when the cell box has <= dense_max_cells cells
  long long dense_max_cells = 16ll << 20;  // lsa_ndt_set_dense_limit
  DevBuf<double> d_small;
  bool has_target = false;
  size_t n_target = 0, n_valid = 0;
  bool n_valid_known = false;    // n_valid is counted on demand (ndt_count_valid)
  DevBuf<int32_t> counts;        // points per occupied voxel (the index's own counts are zeroed again by the build)
  const float4 *d_target_pts = nullptr;  // the cloud the table was built from (ours, or the caller's device buffer)
  FitnessState *fitness = nullptr;       // lazily built leaf table for GetFitnessScore
  NNGrid nn;                             // PCL mode: lazily built point-level neighbour grid for GetFitnessScore

  // EXTERNAL table (the front end's incremental local map, local_map.hip): records and dense probe table live in the
  // owner's persistent structures and are updated in place key-frame by key-frame; vi.map holds the table's window.  The
  // entry points that need more than the pass kernels do (grid_info / grid_dump / GetFitnessScore: ranks, counts, the
  // octree's leaf layout) first ask the owner to materialise an ordinary table (ext_materialize), which leaves this mode.
  const double *rec_ext = nullptr;
  const float4 *dense_ext = nullptr;
  int (*ext_materialize)(void *) = nullptr;
  void *ext_user = nullptr;

  // spatial ordering of large incoherent sources (source_sort.hip): 0 never, 1 when a sample says the cloud is incoherent
  // (default), 2 always (>= 262144 points)
  int source_sort_mode = getenv("LSA_SOURCE_SORT") ? atoi(getenv("LSA_SOURCE_SORT")) : 1;
  bool source_sorted = false;
  unsigned long long n_source_sorts = 0;
  DevBuf<unsigned> sort_keys, sort_idx, sort_small;
  DevBuf<char> sort_tmp;
  DevBuf<float4> d_source_sorted;

  // source
  DevBuf<float4> d_source_own;
  const float4 *d_source = nullptr;
  size_t n_source = 0;
  PinnedBuf<float> h_stage;
  hipEvent_t ev_upload = nullptr;   // behind the kernel that took the last host cloud out of h_stage (upload_points)
  bool upload_pending = false;

  // per-pass buffers
  DevBuf<double> partials;      // [grid][kOutDoubles]
  DevBuf<double> d_out;         // kOutDoubles
  PinnedBuf<double> h_out;
  PinnedBuf<double> h_partials;  // host-mapped: small passes write their workgroup partials here directly
  unsigned long long pass_seq = 0;  // numbers the host-reduced passes (row check words)
  bool no_row_polling = getenv("LSA_NO_ROW_POLLING") != nullptr;  // diagnostics: wait for the stream per pass instead
  unsigned unsynced = 0;            // host-reduced passes enqueued since the stream was last waited for

  // one-launch ScanMatch (ndt_align.hip)
  DevBuf<double> align_rows;        // [2][grid][kOutDoubles]: the workgroups' rows, double-buffered by pass parity
  DevBuf<unsigned> align_flags;     // [num_cus]: per-workgroup epoch flags
  PinnedBuf<double> h_result;       // host-mapped result row
  unsigned align_seq = 64;          // next launch's epoch base: a launch uses epochs base + 1 .. base + max_iter + 2
  // opt-in (lsa_ndt_set_persistent / LSA_PERSISTENT=1): measured on MI355X it does not beat the per-pass path
  bool no_persistent = getenv("LSA_PERSISTENT") == nullptr;
  unsigned long long align_launches = 0, align_fallbacks = 0;
  DevBuf<unsigned long long> align_stamps;  // diagnostics (lsa_ndt_align_stamps)
  bool align_stamps_on = false;

  // armed passes (ndt_derivatives.hip): the next pass's kernel waits on the device for its parameters
  void *mail = nullptr;             // PassMail in fine-grained device memory, written by the host through the BAR
  bool mail_tried = false, mail_is_device = false;
  bool no_armed = getenv("LSA_NO_ARMED") != nullptr;   // lsa_ndt_set_armed
  bool armed_scope = false;         // inside a ScanMatch whose passes are all of kind 0 (line search dead, single GPU)
  bool armed_outstanding = false;   // a kernel for pass `armed_seq` is enqueued and waiting for its mail
  unsigned long long armed_seq = 1;
  unsigned armed_grid = 0;          // launch grid of the outstanding kernel
  unsigned long long armed_gen = 0, target_gen = 0;   // the voxel table it was launched for / the current one
  // leave the last armed kernel parked for the next ScanMatch: OPT-IN for the plain ScanMatch calls (lsa_ndt_set_armed_carry /
  // LSA_CARRY=1; a parked kernel spins on the caller's stream for up to its patience), switched on by the library itself
  // between the scans of a batch (lsa_pipeline_scan_match), where the next ScanMatch is known to follow at once
  bool armed_carry = getenv("LSA_CARRY") != nullptr;
  double armed_patience_us = 300.0;   // how long an armed kernel waits for its mail before it leaves by itself
  // diagnostics of the armed passes (lsa_ndt_armed_timing): sums over passes
  double at_passes = 0, at_post_to_rows_us = 0, at_rows_to_post_us = 0, at_dev_wait_ticks = 0, at_dev_body_ticks = 0, at_post_us = 0;
  std::chrono::steady_clock::time_point at_last_rows;
  bool at_have_last = false;
  DevBuf<float4> src_keep;          // first-tile source points kept by the last armed pass (see k_ndt_pass_armed)
  unsigned long long match_id = 0, keep_match = ~0ull;   // ScanMatch counter; the ScanMatch whose source src_keep holds
  size_t keep_n = 0;
  PinnedBuf<double> h_gave_up;      // [0]: sequence number of the last armed kernel that left because no mail came
  unsigned long long armed_passes = 0, armed_cancels = 0, armed_fallbacks = 0;
  // what used to be silent: rows that missed their polling limit but were there once the stream had drained (armed /
  // ordinary host-reduced passes), armed kernels seen to have given up, and the longest time any pass's rows took
  unsigned long long armed_late = 0, rows_late = 0, armed_gave_up_seen = 0, armed_expired = 0;
  double rows_wait_max_us = 0;
  double host_gap_max_us = 0;   // lsa_host_gap_take_us, folded in after every wait for rows
  double row_wait_max_us = 0;   // lsa_row_wait_take_us, likewise: the longest wait for ONE row

  // deferred final pass of a ScanMatch inside a batch (ndt_api.hip: collect_pending)
  struct Pending {
    bool on = false;
    unsigned long long seq = 0;
    unsigned g = 0;
    float T[16];
    double p[6];
    const float4 *d_source = nullptr;
    size_t n_source = 0;
    lsa_ndt_stats *stats = nullptr;   // the caller's statistics block the pass completes (may be null)
    double pairs_before = 0;
    int passes = 0;
    std::chrono::steady_clock::time_point posted;
  } pending;
  bool defer_final = false;            // set by lsa_pipeline_scan_match for the duration of a batch
  unsigned long long deferred_passes = 0;

  // multi-GPU
  lsa_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;

  // diagnostics: per-pass trace of the Newton driver (lsa_ndt_set_trace)
  double *trace_rows = nullptr;
  int trace_cap = 0, *trace_n = nullptr;

  // last result (for GetFitnessScore)
  float last_T[16];
  bool has_result = false;
  double pass_ms = 0;
};

inline const double *ndt_rec(const lsa_ndt *h) { return h->rec_ext ? h->rec_ext : h->rec.p; }
inline const float4 *ndt_dense(const lsa_ndt *h) { return h->dense_ext ? h->dense_ext : h->dense.p; }
// adopt / leave an external table (ndt_grid.hip); window = {min_b, dim} in cells of the handle's resolution
int ndt_adopt_external(lsa_ndt *h, const double *rec, const float4 *dense, const int min_b[3], const int dim[3], size_t n_target,
                       int (*materialize)(void *), void *user);
int ndt_reserve(lsa_ndt *h);   // ndt_api.hip: set up now what the first ScanMatch would allocate
void ndt_touch_external(lsa_ndt *h, size_t n_target);   // the owner changed the table's contents in place
int ndt_materialize(lsa_ndt *h);                         // no-op unless the table is external
int ndt_source_order(lsa_ndt *h, const float4 *d_src, size_t n, const float4 **d_use);   // source_sort.hip
int ndt_build_grid(lsa_ndt *h, const float4 *d_pts, size_t n);
bool ndt_defer_scope(lsa_ndt *h, bool on);   // ndt_api.hip: the final pass of every ScanMatch of a batch may be deferred
int ndt_collect_deferred(lsa_ndt *h);
bool ndt_carry_scope(lsa_ndt *h, bool on);   // ndt_api.hip: carry between the scans of a batch; returns the previous setting
void ndt_release_parked(lsa_ndt *h);
int ndt_count_valid(lsa_ndt *h);
void ndt_fitness_invalidate(lsa_ndt *h);
void ndt_fitness_release(lsa_ndt *h);
bool ndt_align_persistent_ok(const lsa_ndt *h);
int ndt_align_persistent(lsa_ndt *h, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats, bool *done);

#endif
