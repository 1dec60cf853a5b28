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

int lsa_voxel_filter_apply_host_async(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float *d_rows, float *d_out);   // voxel_filter.hip
float *lsa_voxel_filter_host_buffer(lsa_voxel_filter *f, size_t n, int which);

// The same batch fed with HOST scans (32-byte PointXYZI rows or any stride): what a replay from host memory, or a sensor queue that
// has backed up, hands over.  The ingest is F scans deep on F >= 2 filter handles (contexts / streams of their own): scans k + 1 ..
// k + F are packed by the handles' helper threads and pulled over the link by their first kernels (voxel_filter.hip, job kind 5)
// while scan k is registered — a 131 k-point scan needs ~44 us on the link and ~35 us of two packing threads, then ~25 us of
// kernels (a handle's turn-around is ~100 us), a registration ~45 us.  Measured: 12.7 k scans/s with two handles, 11.4 / 11.8 k with
// three / four — ~80 us per scan whatever the depth, i.e. not the ingest's latency but what packing (6.3 MB of host memory traffic
// per scan) and the link's reads of the packed rows (2.1 MB) cost the host's memory system.  Buffers are the handles' own (raw rows; two filtered clouds each: scan j's goes to buffer
// (j / F) % 2 of handle j % F, so scan k - 1's — its deferred final pass may still read it —, k's and the F in flight never share
// one).  Results are exactly those of the sequential host calls.
extern "C" int lsa_pipeline_scan_match_host(lsa_voxel_filter *const *filters, int n_filters, lsa_ndt *ndt, size_t count, const void *const *clouds,
                                            const size_t *n_points, size_t stride, size_t w_off, const float *predict_poses, float *result_poses,
                                            lsa_ndt_stats *stats, size_t *n_filtered) {
  constexpr int kMaxFilters = 8;
  if (!filters || n_filters < 2 || n_filters > kMaxFilters || !ndt || (count && (!clouds || !n_points || !predict_poses || !result_poses))) return LSA_ERR_INVALID;
  for (int a = 0; a < n_filters; a++) {
    if (!filters[a]) return LSA_ERR_INVALID;
    for (int b = 0; b < a; b++)
      if (filters[a] == filters[b]) return LSA_ERR_INVALID;
  }
  if (count == 0) return LSA_OK;
  const size_t F = (size_t)n_filters;
  size_t n_max = 1;
  for (size_t k = 0; k < count; k++) n_max = n_points[k] > n_max ? n_points[k] : n_max;
  float *raw[kMaxFilters], *out[kMaxFilters][2];
  for (size_t h = 0; h < F; h++) {
    raw[h] = lsa_voxel_filter_host_buffer(filters[h], n_max, 0);
    out[h][0] = lsa_voxel_filter_host_buffer(filters[h], n_max, 1);
    out[h][1] = lsa_voxel_filter_host_buffer(filters[h], n_max, 2);
    if (!raw[h] || !out[h][0] || !out[h][1]) return LSA_ERR_HIP;
  }
  int rc;
  size_t posted = 0;
  // scan j: handle j % F, filtered buffer (j / F) % 2 of that handle
  auto post = [&](size_t j) { return lsa_voxel_filter_apply_host_async(filters[j % F], clouds[j], n_points[j], stride, w_off, raw[j % F], out[j % F][(j / F) & 1]); };
  auto drain = [&](size_t from) { for (size_t j = from; j < posted; j++) { size_t dummy; (void)lsa_voxel_filter_wait(filters[j % F], &dummy); } };
  for (; posted < count && posted < F; posted++)
    if ((rc = post(posted))) { drain(0); return rc; }
  ndt_prearm(ndt);   // the first pass's kernel, enqueued while the first scan is on its way
  const bool carry_user = ndt_carry_scope(ndt, true);
  const bool defer_user = ndt_defer_scope(ndt, true);
  struct Restore {
    lsa_ndt *h; bool prev, prev_defer;
    ~Restore() { (void)ndt_collect_deferred(h); (void)ndt_defer_scope(h, prev_defer); (void)ndt_carry_scope(h, prev); if (!prev) ndt_release_parked(h); }
  } restore{ndt, carry_user, defer_user};
  for (size_t k = 0; k < count; k++) {
    size_t m = 0;
    rc = lsa_voxel_filter_wait(filters[k % F], &m);
    const float *src = out[k % F][(k / F) & 1];
    if (rc == LSA_ERR_RANGE) { src = raw[k % F]; m = n_points[k]; rc = LSA_OK; }   // PCL passes the input through
    if (rc) { drain(k + 1); return rc; }
    const bool range_src = src == raw[k % F];   // (the registered cloud is then the raw buffer the next post on this handle overwrites: finish with it first)
    if (posted < count && !range_src) {
      if ((rc = post(posted))) { drain(k + 1); return rc; }
      posted++;
    }
    if (k + 1 == count) (void)ndt_carry_scope(ndt, carry_user);
    rc = lsa_ndt_scan_match_device(ndt, src, m, predict_poses + 16 * k, result_poses + 16 * k, stats ? stats + k : nullptr);
    if (rc) { drain(k + 1); return rc; }
    if (range_src) {
      if ((rc = ndt_collect_deferred(ndt))) { drain(k + 1); return rc; }
      if (posted < count) {
        if ((rc = post(posted))) { drain(k + 1); return rc; }
        posted++;
      }
    }
    if (n_filtered) n_filtered[k] = m;
  }
  return LSA_OK;
}
