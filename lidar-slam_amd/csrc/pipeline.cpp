// pipeline.cpp — lsa_pipeline_scan_match: a batch of scans through frame filter -> NDT ScanMatch with the two stages
// software-pipelined inside the library (the filter of scan k+1 is enqueued on the filter's context before the
// registration of scan k starts on the NDT's context), i.e. the loop a C++ caller such as front_end_node would write
// around lsa_voxel_filter_apply_device_async / _wait and lsa_ndt_scan_match_device, without a foreign-function call per
// stage.  Results are exactly those of the sequential calls: every scan is filtered once and registered once.
#include "lsa_common.h"

#include <chrono>
#include <vector>

bool ndt_carry_scope(lsa_ndt *h, bool on);   // ndt_api.hip
void ndt_release_parked(lsa_ndt *h);
void ndt_prearm(lsa_ndt *h);
bool ndt_defer_scope(lsa_ndt *h, bool on);
int ndt_collect_deferred(lsa_ndt *h);
float *lsa_voxel_filter_pipe_buffer(lsa_voxel_filter *f, size_t n);
void lsa_voxel_filter_retire_begin(lsa_voxel_filter *f);
void lsa_voxel_filter_retire_end(lsa_voxel_filter *f);
int lsa_voxel_filter_apply_device_async_here(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw);   // voxel_filter.hip

extern "C" int lsa_pipeline_scan_match(lsa_voxel_filter *frame_filter, lsa_ndt *ndt, size_t count, const float *const *d_scans_xyzw,
                                       const size_t *n_points, const float *predict_poses, float *d_filtered_a, float *d_filtered_b, int rank,
                                       int world, float *result_poses, lsa_ndt_stats *stats, size_t *n_filtered) {
  if (!frame_filter || !ndt || (count && (!d_scans_xyzw || !n_points || !predict_poses || !result_poses || !d_filtered_a || !d_filtered_b)))
    return LSA_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return LSA_ERR_INVALID;
  if (count == 0) return LSA_OK;
  // THREE filtered-cloud buffers in rotation (the caller's two and one the filter handle owns): the final derivative pass of
  // scan k may still be reading its source when the filter of scan k + 2 is enqueued (deferred final pass, ndt_api.hip) — with
  // two buffers that filter would write the very cloud the pass reads.
  size_t n_max = 0;
  for (size_t k = 0; k < count; k++) n_max = n_points[k] > n_max ? n_points[k] : n_max;
  float *buf[3] = {d_filtered_a, d_filtered_b, lsa_voxel_filter_pipe_buffer(frame_filter, n_max)};
  const size_t nbuf = buf[2] ? 3 : 2;
  int rc = lsa_voxel_filter_apply_device_async_here(frame_filter, d_scans_xyzw[0], n_points[0], buf[0]);
  if (rc) return rc;
  // Between the scans of the batch the next ScanMatch is known to follow at once and nothing else uses the NDT's stream, so
  // the armed kernel of the pass that never came stays parked for the next scan's first pass; behind the LAST scan it is
  // cancelled (unless the caller opted into the carry himself): the batch returns with nothing spinning on the stream.
  ndt_prearm(ndt);   // the first pass's kernel, enqueued while the first filter runs
  const bool carry_user = ndt_carry_scope(ndt, true);
  // ... and the LAST derivative pass of every ScanMatch — whose sums only complete the statistics, the pose is final before it —
  // is posted but not waited for: it runs while the host moves on to the next scan and is collected behind that scan's first
  // mail (behind the last scan: before this call returns).  Needs the third buffer.
  const bool defer_user = ndt_defer_scope(ndt, nbuf == 3);
  struct Restore {
    lsa_ndt *h; bool prev, prev_defer;
    ~Restore() { (void)ndt_collect_deferred(h); (void)ndt_defer_scope(h, prev_defer); (void)ndt_carry_scope(h, prev); if (!prev) ndt_release_parked(h); }
  } restore{ndt, carry_user, defer_user};
  // diagnostics (LSA_PIPE_TRACE=1, batches of <= 64 scans): when each scan's filter was there and its ScanMatch returned, on stderr
  static const bool trace = getenv("LSA_PIPE_TRACE") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  std::vector<double> tr;
  auto stamp = [&] { if (trace && count <= 64) tr.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count()); };
  struct Dump {
    std::vector<double> &t;
    ~Dump() {
      if (t.empty()) return;
      fprintf(stderr, "[lsa pipe] us since the call began (filter there / ScanMatch back):");
      for (size_t i = 0; i + 1 < t.size(); i += 2) fprintf(stderr, " %.0f/%.0f", t[i], t[i + 1]);
      fprintf(stderr, "\n");
    }
  } dump{tr};
  for (size_t k = 0; k < count; k++) {
    size_t m = 0;
    if ((rc = lsa_voxel_filter_wait(frame_filter, &m))) return rc;
    stamp();
    if (k + 1 < count && (rc = lsa_voxel_filter_apply_device_async(frame_filter, d_scans_xyzw[k + 1], n_points[k + 1], buf[(k + 1) % nbuf]))) return rc;
    // multi-GPU: this rank's contiguous shard of the filtered source; the all-reduce callback installed on
    // the NDT handle sums the per-rank normal equations
    if (k + 1 == count) {
      (void)ndt_carry_scope(ndt, carry_user);
      lsa_voxel_filter_retire_begin(frame_filter);   // nothing more to filter: the helper thread retires the filter's stream meanwhile
    }
    size_t lo = (m * (size_t)rank) / (size_t)world, hi = (m * (size_t)(rank + 1)) / (size_t)world;
    rc = lsa_ndt_scan_match_device(ndt, buf[k % nbuf] + 4 * lo, hi - lo, predict_poses + 16 * k, result_poses + 16 * k, stats ? stats + k : nullptr);
    stamp();
    if (k + 1 == count) lsa_voxel_filter_retire_end(frame_filter);
    if (rc) {  // leave no filter in flight behind an error
      if (k + 1 < count) { size_t dummy; (void)lsa_voxel_filter_wait(frame_filter, &dummy); }
      return rc;
    }
    if (n_filtered) n_filtered[k] = m;
  }
  return LSA_OK;
}
