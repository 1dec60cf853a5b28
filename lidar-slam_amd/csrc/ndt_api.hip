// ndt_api.hip — lsa_ndt_*: the RegistrationInterface replacement for NDT
// (reference: lidar_localization/src/models/registration/ndt_registration_manual.cpp:12-78).
#include "ndt_internal.h"
#include "ndt_pass.h"

#include <chrono>
#include <vector>

int ndt_fitness(lsa_ndt *h, float *score);  // ndt_fitness.hip
int lsa_gn_solve_defer(lsa_pass_eval_fn eval, int (*defer)(void *, const float[16], const double[6], int), void *user, float res, float step_size, float trans_eps,
                       int max_iter, float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats,
                       double *trace_rows, int trace_cap, int *trace_n, double *pairs_before_deferred);   // gn_driver.cpp

namespace {

constexpr size_t kPullRowsMax = 65536;   // lsa_upload_xyzw takes clouds up to this size out of the staging buffer with one kernel

int upload_points(lsa_ndt *h, const void *pts, size_t n, size_t stride, size_t w_off, DevBuf<float4> &dst) {
  lsa_context *ctx = h->ctx;
  int rc;
  const bool grow = n * 4 > h->h_stage.cap || n > dst.cap;
  if (grow || n > kPullRowsMax || !h->ev_upload) {
    // the staging buffer may still be feeding an earlier upload, and growing either buffer frees the old one: wait first
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else if (h->upload_pending) {
    // only the previous upload's kernel reads the staging buffer; everything else the stream holds may go on
    LSA_HIP(ctx, hipEventSynchronize(h->ev_upload));
  }
  h->upload_pending = false;
  if ((rc = h->h_stage.reserve(ctx, n * 4))) return rc;
  if ((rc = dst.reserve(ctx, n, 1.25))) return rc;
  if ((rc = lsa_upload_xyzw(ctx, pts, n, stride, w_off, h->h_stage.p, dst.p))) return rc;
  if (n == 0) return LSA_OK;
  // behind EVERY upload, the chunked one of a large cloud included: its kernels read the staging buffer too, and the next (small)
  // upload repacks it after waiting for this event alone
  if (!h->ev_upload) LSA_HIP(ctx, hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming));
  LSA_HIP(ctx, hipEventRecord(h->ev_upload, ctx->stream));
  h->upload_pending = true;
  return LSA_OK;
}

// the mailbox of the armed passes: fine-grained device memory the host writes through the (large) PCIe BAR; pinned host
// memory where the device has no large BAR (the pollers then read over the link)
int mail_ensure(lsa_ndt *h) {
  if (h->mail || h->mail_tried) return LSA_OK;
  h->mail_tried = true;
  lsa_context *ctx = h->ctx;
  hipDeviceProp_t prop;
  void *m = nullptr;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.isLargeBar &&
      hipExtMallocWithFlags(&m, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
    h->mail = m; h->mail_is_device = true;
  } else if (hipHostMalloc(&m, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    h->mail = m; h->mail_is_device = false;
  } else {
    (void)hipGetLastError();
    return LSA_OK;  // no mailbox: the passes are launched the ordinary way
  }
  LSA_HIP(ctx, hipMemsetAsync(h->mail, 0, 4096, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

void armed_cancel(lsa_ndt *h) {
  if (h->ctx->unpark_arg == h) { h->ctx->unpark = nullptr; h->ctx->unpark_arg = nullptr; }
  if (!h->armed_outstanding) return;
  ndt_mail_post(reinterpret_cast<PassMail *>(h->mail), nullptr, h->armed_seq, true, nullptr, 0, 0);
  h->armed_outstanding = false;
  h->armed_seq++;
  h->armed_cancels++;
}
void armed_cancel_cb(void *user) { armed_cancel((lsa_ndt *)user); }

// A pass whose mail has been posted delivers its rows within microseconds (tens under heavy contention).  Beyond this limit
// the host stops polling, waits for the stream the ordinary way and looks again; both outcomes are COUNTED
// (lsa_ndt_armed_stats), nothing about a slow pass is silent.
constexpr double kArmedRowLimitMs = 2.0;
constexpr double kRowLimitMs = 20.0;   // ordinary (unarmed) passes: up to 190 us of kernel at 2 M points, more beside other streams

// launch grid of an armed kernel for a pass that needs g workgroups: a quarter more, so that the kernel left parked at the
// end of a ScanMatch also fits the next scan's (slightly different) source size
unsigned armed_headroom(const lsa_ndt *h, unsigned g) {
  unsigned cap = (unsigned)h->ctx->num_cus * 3u, want = g + g / 4 + 2;
  return want < cap ? want : (g < cap ? cap : g);
}

int run_pass(lsa_ndt *h, const float T[16], const double p[6], int kind, double out32[32]);
int collect_pending(lsa_ndt *h);

// ---- armed pass, first half: this pass's kernel is already waiting on the device — enqueued behind the previous pass, or left
// parked by the previous ScanMatch — or is launched now; post its parameters and enqueue the kernel of the NEXT pass behind it
struct ArmedPosted {
  unsigned long long seq;
  unsigned g;
  bool kept;
  bool deferred = false;   // taken long after it was posted, by design: its post -> first look interval says nothing about the host
  std::chrono::steady_clock::time_point at0, at1;
};
int armed_post(lsa_ndt *h, const PassParams &P, ArmedPosted *ap) {
  int rc;
  const unsigned g = ndt_pass_grid(h, h->n_source, 0);
  volatile unsigned long long *gave_up = h->h_gave_up.p ? reinterpret_cast<volatile unsigned long long *>(h->h_gave_up.p) : nullptr;
  if (h->armed_outstanding && (h->armed_grid < g || h->armed_gen != h->target_gen || (gave_up && *gave_up == h->armed_seq))) {
    if (gave_up && *gave_up == h->armed_seq) h->armed_expired++;
    armed_cancel(h);   // parked for another table / too small a grid, or it has already left: not usable
  }
  unsigned long long seq = h->armed_seq;
  // A deferred final pass still to be collected sits in the mailbox slot and the row-buffer half of ITS parity.  Normally the
  // next pass has the other one (consecutive numbers); after a cancel in between the numbers skip one and the two would share:
  // take the deferred pass's rows first.
  if (h->pending.on && ((h->pending.seq ^ seq) & 1ull) == 0ull && (rc = collect_pending(h))) return rc;
  if (!h->armed_outstanding && (rc = ndt_launch_armed(h, P, seq, armed_headroom(h, g), h->armed_patience_us))) return rc;
  // (after the first armed pass of this ScanMatch the kernels hold their first tiles from the kept copy)
  static const bool no_keep = getenv("LSA_NO_KEEP") != nullptr;   // A/B switch for measurements (scratch: ab_match.py)
  ap->kept = !no_keep && h->keep_match == h->match_id && h->keep_n == h->n_source;
  ap->at0 = std::chrono::steady_clock::now();
  // host-stall detector: inside a ScanMatch the kernel of this pass has been waiting since the previous pass's rows were taken
  if (h->at_have_last) lsa_host_gap_note_us(std::chrono::duration<double, std::micro>(ap->at0 - h->at_last_rows).count());
  ndt_mail_post(reinterpret_cast<PassMail *>(h->mail), &P, seq, false, h->d_source, h->n_source, g, ap->kept);
  ap->at1 = std::chrono::steady_clock::now();
  h->armed_seq = seq + 1;
  h->armed_outstanding = false;
  if ((rc = ndt_launch_armed(h, P, seq + 1, armed_headroom(h, g), h->armed_patience_us))) return rc;
  h->armed_outstanding = true;
  h->armed_grid = armed_headroom(h, g);
  h->armed_gen = h->target_gen;
  h->armed_passes++;
  ap->seq = seq; ap->g = g;
  return LSA_OK;
}

// ---- second half: take the rows of armed pass `seq` as they arrive.  *redo = true (and LSA_OK): the pass did not run
// (completely); the caller evaluates it the ordinary way.  Every outcome other than the plain one is counted.
int armed_take(lsa_ndt *h, const ArmedPosted &ap, double out32[32], bool *redo) {
  lsa_context *ctx = h->ctx;
  *redo = false;
  const unsigned long long seq = ap.seq;
  const unsigned g = ap.g;
  volatile unsigned long long *gave_up = reinterpret_cast<volatile unsigned long long *>(h->h_gave_up.p);
  const double *rows = ndt_armed_rows(h, seq);
  // host-stall detector: the way from the mail to the first look at the rows (the launch of the next pass's kernel, ~3 us) ...
  if (!ap.deferred) lsa_host_gap_note_us(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ap.at0).count());
  bool ok = !h->no_row_polling && ndt_reduce_partials_host(rows, g, ndt_armed_row_key(seq), kArmedRowLimitMs, out32, gave_up, seq);
  {
    const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ap.at1).count();
    if (w > h->rows_wait_max_us) h->rows_wait_max_us = w;
    const double gp = lsa_host_gap_take_us(true);   // ... and every interval between two looks
    if (gp > h->host_gap_max_us) h->host_gap_max_us = gp;
    const double rw = lsa_row_wait_take_us(true);
    if (rw > h->row_wait_max_us) h->row_wait_max_us = rw;
  }
  if (!ok) {
    // rows missing.  Either a workgroup of the kernel ran out of patience before the mail came (this thread was held up;
    // EVERY workgroup that leaves says so in *gave_up, so the wait above ends at once instead of at its limit), or the rows
    // are slow (limit), or polling is switched off: let everything enqueued drain, look again, and if the pass really did
    // not run (completely) do it the ordinary way, unarmed for the rest of this ScanMatch.
    const bool left = gave_up && *gave_up == seq;
    if (left) h->armed_gave_up_seen++;
    armed_cancel(h);
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
    if (ndt_reduce_partials_host(rows, g, ndt_armed_row_key(seq), 0.0, out32)) {
      // complete after all: the rows were slow (counted as late), or only a workgroup of the launch's head-room — one that
      // has no row to deliver — ran out of patience while the others took the mail (counted above)
      if (!h->no_row_polling && !left) h->armed_late++;
      return LSA_OK;
    }
    h->armed_scope = false;
    h->armed_fallbacks++;
    *redo = true;
    return LSA_OK;
  }
  {  // diagnostics (lsa_ndt_armed_timing)
    const auto at2 = std::chrono::steady_clock::now();
    h->at_passes += 1;
    h->at_post_us += std::chrono::duration<double, std::micro>(ap.at1 - ap.at0).count();
    h->at_post_to_rows_us += std::chrono::duration<double, std::micro>(at2 - ap.at1).count();
    if (h->at_have_last) h->at_rows_to_post_us += std::chrono::duration<double, std::micro>(ap.at0 - h->at_last_rows).count();
    h->at_last_rows = at2; h->at_have_last = true;
    const unsigned long long st = (unsigned long long)out32[30];   // sum over the workgroups of (wait << 20) + body
    h->at_dev_wait_ticks += (double)(st >> 20) / (double)g;
    h->at_dev_body_ticks += (double)(st & 0xfffffull) / (double)g;
  }
  h->unsynced += 2;
  return LSA_OK;
}

// ---- deferred final pass (see gn_driver.cpp: Driver::defer).  Inside a batch (lsa_pipeline_scan_match) the last pass of a
// ScanMatch — whose sums only complete the statistics — is posted like any armed pass but NOT waited for: the call returns
// with the pose, the pass runs while the host moves on, and its rows are taken behind the first mail of the next ScanMatch
// (or at the batch's end), when the statistics of the previous scan are completed.  The mailbox and the row buffer have two
// halves by sequence parity for exactly this overlap (ndt_derivatives.hip).
int collect_pending(lsa_ndt *h) {
  if (!h->pending.on) return LSA_OK;
  lsa_ndt::Pending pd = h->pending;
  h->pending.on = false;
  ArmedPosted ap;
  ap.seq = pd.seq; ap.g = pd.g; ap.kept = true; ap.deferred = true; ap.at0 = ap.at1 = pd.posted;
  double out32[32];
  bool redo = false;
  const bool have_last = h->at_have_last;
  h->at_have_last = false;   // (host-side gaps across a deferred pass mean nothing)
  int rc = armed_take(h, ap, out32, &redo);
  h->at_have_last = have_last && !redo;
  if (rc) return rc;
  if (redo) {  // the ordinary way, on the cloud and at the pose the pass was for
    const float4 *src = h->d_source;
    const size_t n = h->n_source;
    h->d_source = pd.d_source; h->n_source = pd.n_source;
    rc = run_pass(h, pd.T, pd.p, 0, out32);
    h->d_source = src; h->n_source = n;
    if (rc) return rc;
  }
  if (pd.stats) {
    pd.stats->score = out32[0];
    pd.stats->trans_probability = out32[29] > 0 ? out32[0] / out32[29] : 0;
    pd.stats->pairs_per_pass = pd.passes ? (pd.pairs_before + out32[28]) / pd.passes : 0;
  }
  h->deferred_passes++;
  return LSA_OK;
}

int defer_trampoline(void *user, const float T[16], const double p[6], int kind) {
  lsa_ndt *h = (lsa_ndt *)user;
  if (kind != 0 || !h->defer_final || h->pending.on || !(h->armed_scope && h->mail && ndt_pass_host_reduce(h, 0))) return 1;
  PassParams P;
  ndt_make_pass_params(h, T, p, &P);
  ArmedPosted ap;
  if (armed_post(h, P, &ap)) return 1;
  lsa_ndt::Pending &pd = h->pending;
  pd.on = true;
  pd.seq = ap.seq; pd.g = ap.g; pd.posted = ap.at1;
  memcpy(pd.T, T, sizeof(pd.T)); memcpy(pd.p, p, sizeof(pd.p));
  pd.d_source = h->d_source; pd.n_source = h->n_source;
  pd.stats = nullptr; pd.pairs_before = 0; pd.passes = 0;   // completed by scan_match_common once the driver has returned
  return 0;
}

// one pass, synchronous: kernel(s) -> [all-reduce] -> D2H of 32 doubles
int run_pass(lsa_ndt *h, const float T[16], const double p[6], int kind, double out32[32]) {
  lsa_context *ctx = h->ctx;
  PassParams P;
  ndt_make_pass_params(h, T, p, &P);
  int rc;
  if (h->armed_scope && kind == 0 && h->mail && ndt_pass_host_reduce(h, 0)) {
    ArmedPosted ap;
    // The previous ScanMatch's deferred final pass is normally taken BEHIND this pass's mail (its kernel ran in front of this
    // pass's, its rows are being delivered while this one starts).  If its kernel is known to have left before its mail came, the
    // pass will be redone the ordinary way — whose rows go to the shared row half 0 and whose stream wait would sit in front of a
    // pass already posted: redo it first, then post.
    {
      volatile unsigned long long *gave_up = h->h_gave_up.p ? reinterpret_cast<volatile unsigned long long *>(h->h_gave_up.p) : nullptr;
      if (h->pending.on && gave_up && *gave_up == h->pending.seq && (rc = collect_pending(h))) return rc;
    }
    if ((rc = armed_post(h, P, &ap))) return rc;
    if (h->pending.on && (rc = collect_pending(h))) return rc;
    bool redo = false;
    if ((rc = armed_take(h, ap, out32, &redo))) return rc;
    if (redo) return run_pass(h, T, p, kind, out32);
    if (!ap.kept) { h->keep_match = h->match_id; h->keep_n = h->n_source; }   // that pass kept its first tiles
    return LSA_OK;
  }
  if (h->pending.on && (rc = collect_pending(h))) return rc;
  if (ndt_pass_host_reduce(h, kind)) {
    unsigned g = 0;
    unsigned long long key = 0;
    if ((rc = ndt_enqueue_pass_host(h, P, kind, &g, &key))) return rc;
    // The rows are consumed as they arrive in host memory (check word per row): the host does not wait for the stream
    // per pass (measured: 17 us in hipStreamSynchronize around a 9 us kernel), only every so often so that the runtime
    // can retire its commands.  A row that does not arrive in 20 ms falls back to the ordinary wait.
    const auto rt0 = std::chrono::steady_clock::now();
    bool ok = !h->no_row_polling && ndt_reduce_partials_host(h->h_partials.p, g, key, kRowLimitMs, out32);
    {
      const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - rt0).count();
      if (w > h->rows_wait_max_us) h->rows_wait_max_us = w;
      const double gp = lsa_host_gap_take_us(true);
      if (gp > h->host_gap_max_us) h->host_gap_max_us = gp;
      const double rw = lsa_row_wait_take_us(true);
      if (rw > h->row_wait_max_us) h->row_wait_max_us = rw;
    }
    if (!ok) {
      LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      h->unsynced = 0;
      if (!ndt_reduce_partials_host(h->h_partials.p, g, key, 0.0, out32))
        return lsa_fail(ctx, LSA_ERR_HIP, "NDT pass: workgroup rows incomplete after the kernel finished");
      if (!h->no_row_polling) h->rows_late++;
    } else if (++h->unsynced >= 256) {
      LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      h->unsynced = 0;
    }
    return LSA_OK;
  }
  rc = ndt_enqueue_pass(h, P, kind);
  if (rc) return rc;
  if (h->allreduce) {
    int crc = h->allreduce(h->allreduce_user, h->d_out.p, kOutDoubles, (void *)ctx->stream);
    if (crc) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "all-reduce callback returned %d", crc);
  }
  // the block reaches the host as a check-worded row written by a one-wave kernel behind the collective: no D2H copy, no
  // stream wait per pass (a row that does not land within 20 ms falls back to the wait)
  unsigned long long key = 0;
  if ((rc = ndt_publish_out(h, &key))) return rc;
  unsigned long long w[32];
  bool ok = !h->no_row_polling && ndt_fetch_row_host(h->h_out.p, key, kRowLimitMs, w);
  if (!ok) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
    if (!ndt_fetch_row_host(h->h_out.p, key, 0.0, w)) return lsa_fail(ctx, LSA_ERR_HIP, "NDT pass: reduced row incomplete after the stream drained");
    if (!h->no_row_polling) h->rows_late++;
  } else if (++h->unsynced >= 256) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
  }
  memcpy(out32, w, kOutDoubles * sizeof(double));
  return LSA_OK;
}

int eval_trampoline(void *user, const float T[16], const double p[6], int kind, double out32[32]) {
  return run_pass((lsa_ndt *)user, T, p, kind, out32);
}

int check(lsa_ndt *h) {
  if (!h) return LSA_ERR_INVALID;
  hipError_t e = lsa_enter(h->ctx);
  if (e != hipSuccess) return lsa_fail(h->ctx, LSA_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->ctx->device, hipGetErrorString(e));
  return collect_pending(h);   // (a deferred final pass reads the source and the table: nothing may change under it)
}

// entry of the ScanMatch calls: like check(), but a kernel this very handle left parked stays (the call is what it waits for)
int check_keep(lsa_ndt *h) {
  if (!h) return LSA_ERR_INVALID;
  lsa_ctx_unpark(h->ctx, h);
  hipError_t e = hipSetDevice(h->ctx->device);
  if (e != hipSuccess) return lsa_fail(h->ctx, LSA_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->ctx->device, hipGetErrorString(e));
  return LSA_OK;
}

int scan_match_common(lsa_ndt *h, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats) {
  lsa_context *ctx = h->ctx;
  if (!h->has_target) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "ScanMatch before SetInputTarget");
  lsa_ndt_gauss_constants(h->resolution, h->outlier_ratio, &h->d1, &h->d2);
  lsa_ndt_stats st;
  memset(&st, 0, sizeof(st));
  int rc;
  if (ndt_align_persistent_ok(h)) {  // the whole Newton iteration in one launch (ndt_align.hip)
    armed_cancel(h);   // (a kernel an earlier per-pass ScanMatch left parked would hold the stream for its whole patience)
    bool done = false;
    if ((rc = ndt_align_persistent(h, predict_pose, result_pose, &st, &done))) return rc;
    if (done) {
      memcpy(h->last_T, result_pose, sizeof(h->last_T));
      h->has_result = true;
      if (stats) *stats = st;
      return LSA_OK;
    }
    memset(&st, 0, sizeof(st));
  }
  // armed passes: only where every pass of the ScanMatch is a full (kind 0) pass, i.e. the trial loop is dead, on one GPU
  h->match_id++;
  h->at_have_last = false;
  h->armed_scope = !h->no_armed && !h->allreduce && h->n_source > 0 && (double)h->step_size > (double)h->trans_eps / 2;
  if (h->armed_scope && (rc = mail_ensure(h))) { h->armed_scope = false; armed_cancel(h); return rc; }
  if (h->armed_scope && !h->mail) h->armed_scope = false;
  // a kernel this handle left parked (check_keep kept it for this call) that the call will NOT feed — an all-reduce was
  // installed meanwhile, the source is empty, there is no mailbox — would hold the ordinary passes back for its whole patience
  if (!h->armed_scope) armed_cancel(h);
  if (h->pending.on && !h->armed_scope && (rc = collect_pending(h))) return rc;
  double pairs_before = -1.0;
  rc = lsa_gn_solve_defer(eval_trampoline, (h->defer_final && h->armed_scope) ? defer_trampoline : nullptr, h, h->resolution, h->step_size, h->trans_eps, h->max_iter,
                          h->outlier_ratio, h->mode, predict_pose, result_pose, &st, h->trace_rows, h->trace_cap, h->trace_n, &pairs_before);
  if (!rc && h->pending.on && pairs_before >= 0) {   // the last pass was deferred: its sums complete *stats later (collect_pending)
    h->pending.stats = stats;
    h->pending.pairs_before = pairs_before;
    h->pending.passes = st.passes;
  } else if (h->pending.on && (rc || pairs_before < 0)) {
    (void)collect_pending(h);   // (posted by a solve that then failed: do not leave it dangling)
  }
  if (h->armed_scope) {
    h->armed_scope = false;
    // The kernel enqueued for the pass that never came: cancelled — or, when scans follow each other on this handle and
    // nothing else uses the stream in between (lsa_pipeline_scan_match), left PARKED for the first pass of the next
    // ScanMatch (it reads source pointer, size and grid from its mail), whose launch + dispatch latency it then saves.  It
    // leaves by itself after armed_patience_us; any other call on this context releases it first (lsa_enter).  The carry is
    // OPT-IN (lsa_ndt_set_armed_carry) except between the scans of lsa_pipeline_scan_match, which switches it on itself and
    // cancels behind its last scan: no entry point returns with a kernel spinning on the caller's stream unless asked to.
    if (rc || !h->armed_carry || h->unsynced >= 1024) {
      armed_cancel(h);
      if (h->unsynced >= 1024) {  // let the runtime retire its commands now and then (nothing is parked any more)
        (void)hipStreamSynchronize(ctx->stream);
        h->unsynced = 0;
      }
    } else if (h->armed_outstanding) {
      lsa_ctx_unpark(ctx, h);
      ctx->unpark = armed_cancel_cb;
      ctx->unpark_arg = h;
    }
  }
  if (rc) return rc;
  memcpy(h->last_T, result_pose, sizeof(h->last_T));
  h->has_result = true;
  if (stats) *stats = st;
  return LSA_OK;
}

}  // namespace

extern "C" int lsa_ndt_create(lsa_ndt **out, lsa_context *ctx, float res, float step_size, float trans_eps, int max_iter, float outlier_ratio) {
  if (!out || !ctx) { lsa_set_global_error("lsa_ndt_create: null argument"); return LSA_ERR_INVALID; }
  if (!(res > 0)) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_ndt_create: res must be > 0");
  lsa_ndt *h = new lsa_ndt();
  h->ctx = ctx;
  h->resolution = res; h->step_size = step_size; h->trans_eps = trans_eps; h->max_iter = max_iter; h->outlier_ratio = outlier_ratio;
  lsa_ndt_gauss_constants(res, outlier_ratio, &h->d1, &h->d2);
  // sequence numbers (row check words, mail) from a range of this handle's own: a row another handle left in recycled pinned
  // memory can never carry this handle's key
  h->pass_seq = lsa_seq_base();
  h->armed_seq = h->pass_seq + 1;
  *out = h;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_mode(lsa_ndt *h, int mode) {
  if (!h) return LSA_ERR_INVALID;
  if (mode != LSA_NDT_MODE_MANUAL && mode != LSA_NDT_MODE_PCL) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_ndt_set_mode: unknown mode %d", mode);
  lsa_ctx_unpark(h->ctx);
  if (mode != h->mode) { h->has_target = false; h->has_result = false; h->target_gen++; }  // the voxel table depends on the mode
  h->mode = mode;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_dense_limit(lsa_ndt *h, long long max_cells) {
  if (!h) return LSA_ERR_INVALID;
  lsa_ctx_unpark(h->ctx);
  h->dense_max_cells = max_cells < 0 ? 0 : max_cells;
  h->has_target = false;  // the probe structure is chosen when the table is built
  h->target_gen++;
  return LSA_OK;
}

extern "C" void lsa_ndt_destroy(lsa_ndt *h) {
  if (!h) return;
  (void)lsa_enter(h->ctx);
  h->pending.stats = nullptr;
  (void)collect_pending(h);
  armed_cancel(h);
  (void)hipStreamSynchronize(h->ctx->stream);
  ndt_fitness_release(h);
  h->nn.release();
  h->vi.release(); h->d_target.release(); h->rec.release(); h->state.release(); h->vbits.release(); h->words.release(); h->d_small.release(); h->counts.release();
  h->dense.release();
  if (h->mail) { if (h->mail_is_device) (void)hipFree(h->mail); else (void)hipHostFree(h->mail); }
  if (h->ev_upload) (void)hipEventDestroy(h->ev_upload);
  h->align_rows.release(); h->align_flags.release(); h->h_result.release(); h->align_stamps.release(); h->h_gave_up.release(); h->src_keep.release();
  h->sort_keys.release(); h->sort_idx.release(); h->sort_small.release(); h->sort_tmp.release(); h->d_source_sorted.release();
  h->d_source_own.release(); h->h_stage.release(); h->partials.release(); h->d_out.release(); h->h_out.release(); h->h_partials.release();
  delete h;
}

extern "C" int lsa_ndt_set_input_target_device(lsa_ndt *h, const float *d_xyzw, size_t n) {
  int rc = check(h);
  if (rc) return rc;
  if (n == 0) return LSA_OK;  // setInputTarget ignores an empty cloud (NormalDistributionsTransform.cpp:94)
  if (!d_xyzw) return lsa_fail(h->ctx, LSA_ERR_INVALID, "set_input_target: null points");
  h->target_gen++;
  return ndt_build_grid(h, (const float4 *)d_xyzw, n);
}

extern "C" int lsa_ndt_set_input_target(lsa_ndt *h, const void *pts, size_t n, size_t stride, size_t w_off) {
  int rc = check(h);
  if (rc) return rc;
  if (n == 0) return LSA_OK;
  if (!pts || stride < 12) return lsa_fail(h->ctx, LSA_ERR_INVALID, "set_input_target: bad points / stride");
  if ((rc = upload_points(h, pts, n, stride, w_off, h->d_target))) return rc;
  h->target_gen++;
  return ndt_build_grid(h, h->d_target.p, n);
}

extern "C" int lsa_ndt_set_source_device(lsa_ndt *h, const float *d_xyzw, size_t n) {
  int rc = check(h);
  if (rc) return rc;
  if (n && !d_xyzw) return lsa_fail(h->ctx, LSA_ERR_INVALID, "set_source: null points");
  h->n_source = n;
  return ndt_source_order(h, (const float4 *)d_xyzw, n, &h->d_source);
}

extern "C" int lsa_ndt_set_source(lsa_ndt *h, const void *src, size_t n, size_t stride) {
  int rc = check(h);
  if (rc) return rc;
  if (n && (!src || stride < 12)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "set_source: bad points / stride");
  if (n) {
    if ((rc = upload_points(h, src, n, stride, SIZE_MAX, h->d_source_own))) return rc;
  }
  h->n_source = n;
  return ndt_source_order(h, h->d_source_own.p, n, &h->d_source);
}

extern "C" int lsa_ndt_derivatives(lsa_ndt *h, const float T[16], const double p[6], int kind, double out32[32]) {
  int rc = check(h);
  if (rc) return rc;
  if (!T || !p || !out32 || kind < 0 || kind > 2) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_ndt_derivatives: bad argument");
  if (!h->has_target) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "derivatives before SetInputTarget");
  lsa_ndt_gauss_constants(h->resolution, h->outlier_ratio, &h->d1, &h->d2);
  return run_pass(h, T, p, kind, out32);
}

extern "C" int lsa_ndt_derivatives_timed(lsa_ndt *h, const float T[16], const double p[6], int kind, int repeat, double out32[32], double *ms_per_pass) {
  int rc = check(h);
  if (rc) return rc;
  lsa_context *ctx = h->ctx;
  if (!T || !p || !out32 || !ms_per_pass || kind < 0 || kind > 2 || repeat < 1) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_ndt_derivatives_timed: bad argument");
  if (!h->has_target) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "derivatives before SetInputTarget");
  lsa_ndt_gauss_constants(h->resolution, h->outlier_ratio, &h->d1, &h->d2);
  PassParams P;
  ndt_make_pass_params(h, T, p, &P);
  if ((rc = ndt_enqueue_pass(h, P, kind))) return rc;  // warm-up + buffer sizing
  LSA_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  // only the pass kernel itself sits between the two events (the 32-double second-stage reduction is excluded),
  // so ms_per_pass is directly comparable with rocprofv3's average duration of k_ndt_pass
  for (int i = 0; i < repeat; i++)
    if ((rc = ndt_enqueue_pass(h, P, kind, false))) return rc;
  LSA_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  if ((rc = ndt_enqueue_pass(h, P, kind, true))) return rc;
  if ((rc = h->h_out.reserve(ctx, kOutDoubles))) return rc;
  LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, h->d_out.p, kOutDoubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  LSA_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_pass = (double)ms / repeat;
  memcpy(out32, h->h_out.p, kOutDoubles * sizeof(double));
  return LSA_OK;
}

extern "C" int lsa_ndt_scan_match_device(lsa_ndt *h, const float *d_src_xyzw, size_t n, const float predict_pose[16], float result_pose[16],
                                         lsa_ndt_stats *stats) {
  int rc = check_keep(h);  // a kernel this handle left parked for exactly this call stays
  if (rc) return rc;
  if (n && !d_src_xyzw) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ScanMatch: null points");
  h->n_source = n;
  if ((rc = ndt_source_order(h, (const float4 *)d_src_xyzw, n, &h->d_source))) return rc;
  if (!predict_pose || !result_pose) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ScanMatch: null pose");
  return scan_match_common(h, predict_pose, result_pose, stats);
}

extern "C" int lsa_ndt_scan_match(lsa_ndt *h, const void *src, size_t n, size_t stride, const float predict_pose[16], float result_pose[16],
                                  lsa_ndt_stats *stats) {
  int rc = lsa_ndt_set_source(h, src, n, stride);
  if (rc) return rc;
  if (!predict_pose || !result_pose) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ScanMatch: null pose");
  return scan_match_common(h, predict_pose, result_pose, stats);
}

extern "C" int lsa_ndt_get_fitness_score(lsa_ndt *h, float *score) {
  int rc = check(h);
  if (rc) return rc;
  if (!score) return lsa_fail(h->ctx, LSA_ERR_INVALID, "GetFitnessScore: null output");
  if (!h->has_target || !h->has_result) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "GetFitnessScore before SetInputTarget + ScanMatch");
  if ((rc = ndt_materialize(h))) return rc;   // (an incremental local map's table: the octree / point search needs an ordinary one)
  if (h->mode == LSA_NDT_MODE_PCL) {  // pcl::Registration::getFitnessScore over the target POINTS (ndt_registration.cpp:63-66)
    if (!h->nn.built && (rc = nn_grid_build(h->ctx, h->nn, h->d_target_pts, h->n_target, h->resolution, "GetFitnessScore"))) return rc;
    double v = 0;
    if ((rc = nn_fitness(h->ctx, h->nn, h->d_source, h->n_source, h->last_T, INFINITY, &v, h->allreduce, h->allreduce_user))) return rc;
    *score = (float)v;
    return LSA_OK;
  }
  return ndt_fitness(h, score);
}

extern "C" int lsa_ndt_set_armed(lsa_ndt *h, int on) {
  if (!h) return LSA_ERR_INVALID;
  lsa_ctx_unpark(h->ctx);
  h->no_armed = !on;
  return LSA_OK;
}

extern "C" int lsa_ndt_armed_counters(lsa_ndt *h, uint64_t *passes, uint64_t *cancels, uint64_t *fallbacks) {
  if (!h) return LSA_ERR_INVALID;
  if (passes) *passes = h->armed_passes;
  if (cancels) *cancels = h->armed_cancels;
  if (fallbacks) *fallbacks = h->armed_fallbacks;
  return LSA_OK;
}

extern "C" int lsa_ndt_armed_stats(lsa_ndt *h, lsa_ndt_armed_stats_t *out, int reset) {
  if (!h) return LSA_ERR_INVALID;
  if (out) {
    out->passes = h->armed_passes; out->cancels = h->armed_cancels; out->fallbacks = h->armed_fallbacks;
    out->late = h->armed_late; out->rows_late = h->rows_late; out->gave_up_seen = h->armed_gave_up_seen;
    out->expired = h->armed_expired;
    out->rows_wait_max_us = h->rows_wait_max_us;
    out->parked = (h->armed_outstanding && h->ctx->unpark_arg == h) ? 1 : 0;
    out->pad = 0;
    out->deferred = h->deferred_passes;
    out->host_gap_max_us = h->host_gap_max_us;
    out->row_wait_max_us = h->row_wait_max_us;
  }
  if (reset) { h->armed_passes = h->armed_cancels = h->armed_fallbacks = h->armed_late = h->rows_late = h->armed_gave_up_seen = h->armed_expired = h->deferred_passes = 0; h->rows_wait_max_us = 0; h->host_gap_max_us = 0; h->row_wait_max_us = 0; }
  return LSA_OK;
}

extern "C" int lsa_ndt_set_armed_patience(lsa_ndt *h, double patience_us) {
  if (!h) return LSA_ERR_INVALID;
  if (!(patience_us >= 1.0) || patience_us > 1e5) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_ndt_set_armed_patience: %g us is outside [1, 1e5]", patience_us);
  lsa_ctx_unpark(h->ctx);
  h->armed_patience_us = patience_us;
  return LSA_OK;
}

// internal (pipeline.cpp): the carry between the scans of a batch, without releasing what is parked; returns the previous setting
bool ndt_carry_scope(lsa_ndt *h, bool on) {
  const bool prev = h->armed_carry;
  h->armed_carry = on;
  return prev;
}
// internal (pipeline.cpp): a batch is about to call ScanMatch on this handle as soon as its first filter is done — enqueue
// the armed kernel of the first pass NOW (it takes source, size and grid from its mail), so that its launch and dispatch
// latency pass while the filter runs.  Sized for the previous ScanMatch's source (a grid that turns out too small is
// cancelled and relaunched by run_pass like any parked kernel that does not fit).  Nothing happens where the armed passes
// would not be used.
void ndt_prearm(lsa_ndt *h) {
  lsa_context *ctx = h->ctx;
  if (h->no_armed || h->allreduce || !h->has_target || h->armed_outstanding || h->n_source == 0 || !((double)h->step_size > (double)h->trans_eps / 2)) return;
  if (ndt_align_persistent_ok(h) || hipSetDevice(ctx->device) != hipSuccess) return;
  if (mail_ensure(h) != LSA_OK || !h->mail) return;
  lsa_ctx_unpark(ctx, h);
  lsa_ndt_gauss_constants(h->resolution, h->outlier_ratio, &h->d1, &h->d2);
  const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const double p[6] = {0, 0, 0, 0, 0, 0};
  PassParams P;
  ndt_make_pass_params(h, T, p, &P);   // (only the static part is used: the transform and the angle tables come by mail)
  const unsigned grid = armed_headroom(h, ndt_pass_grid(h, h->n_source, 0));
  if (ndt_launch_armed(h, P, h->armed_seq, grid, h->armed_patience_us) != LSA_OK) return;
  h->armed_outstanding = true;
  h->armed_grid = grid;
  h->armed_gen = h->target_gen;
  ctx->unpark = armed_cancel_cb;
  ctx->unpark_arg = h;
}

// internal (pipeline.cpp): inside a batch the final pass of every ScanMatch may be deferred; returns the previous setting
bool ndt_defer_scope(lsa_ndt *h, bool on) {
  const bool prev = h->defer_final;
  h->defer_final = on && getenv("LSA_NO_DEFER") == nullptr;
  return prev;
}
int ndt_collect_deferred(lsa_ndt *h) { return collect_pending(h); }

// internal: release a kernel this handle has left parked (the batch is over and nobody is known to follow)
void ndt_release_parked(lsa_ndt *h) { armed_cancel(h); }

// what the first ScanMatch on a handle otherwise sets up on the way (the mailbox of the armed passes, their host-mapped row
// buffer, the result block): ~250 us of allocation calls the owner of a handle can pay when it creates it
int ndt_reserve(lsa_ndt *h) {
  int rc = mail_ensure(h);
  if (rc) return rc;
  return ndt_reserve_armed(h);
}

extern "C" int lsa_ndt_armed_timing(lsa_ndt *h, double out[6], int reset) {
  if (!h) return LSA_ERR_INVALID;
  if (out) {
    out[0] = h->at_passes; out[1] = h->at_post_us; out[2] = h->at_post_to_rows_us; out[3] = h->at_rows_to_post_us;
    out[4] = h->at_dev_wait_ticks * 0.01; out[5] = h->at_dev_body_ticks * 0.01;
  }
  if (reset) { h->at_passes = h->at_post_us = h->at_post_to_rows_us = h->at_rows_to_post_us = h->at_dev_wait_ticks = h->at_dev_body_ticks = 0; h->at_have_last = false; }
  return LSA_OK;
}

extern "C" int lsa_ndt_set_armed_carry(lsa_ndt *h, int on) {
  if (!h) return LSA_ERR_INVALID;
  lsa_ctx_unpark(h->ctx);
  h->armed_carry = on != 0;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_persistent(lsa_ndt *h, int on) {
  if (!h) return LSA_ERR_INVALID;
  h->no_persistent = !on;
  return LSA_OK;
}

extern "C" int lsa_ndt_align_counters(lsa_ndt *h, uint64_t *launches, uint64_t *fallbacks) {
  if (!h) return LSA_ERR_INVALID;
  if (launches) *launches = h->align_launches;
  if (fallbacks) *fallbacks = h->align_fallbacks;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_trace(lsa_ndt *h, double *rows, int cap_rows, int *n_rows) {
  if (!h) return LSA_ERR_INVALID;
  if (rows && (!n_rows || cap_rows < 0)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_ndt_set_trace: rows without a counter");
  h->trace_rows = rows; h->trace_cap = rows ? cap_rows : 0; h->trace_n = rows ? n_rows : nullptr;
  if (n_rows) *n_rows = 0;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_allreduce(lsa_ndt *h, lsa_allreduce_fn fn, void *user) {
  if (!h) return LSA_ERR_INVALID;
  lsa_ctx_unpark(h->ctx);   // sharded passes are not armed: a parked kernel would only be in their way
  h->allreduce = fn;
  h->allreduce_user = user;
  return LSA_OK;
}

extern "C" int lsa_ndt_grid_info(lsa_ndt *h, int64_t info[9]) {
  int rc = check(h);
  if (rc) return rc;
  if (!info) return lsa_fail(h->ctx, LSA_ERR_INVALID, "grid_info: null output");
  if (!h->has_target) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "grid_info before SetInputTarget");
  if ((rc = ndt_materialize(h))) return rc;
  for (int a = 0; a < 3; a++) { info[a] = h->vi.map.dim[a]; info[3 + a] = h->vi.map.min_b[a]; }
  info[6] = (int64_t)h->vi.nocc;
  if ((rc = ndt_count_valid(h))) return rc;
  info[7] = (int64_t)h->n_valid;
  info[8] = (int64_t)h->n_target;
  return LSA_OK;
}

extern "C" int lsa_ndt_grid_dump(lsa_ndt *h, int32_t *cell_ints, double *cell_rec, size_t cap, size_t *n_cells) {
  int rc = check(h);
  if (rc) return rc;
  lsa_context *ctx = h->ctx;
  if (!n_cells) return lsa_fail(ctx, LSA_ERR_INVALID, "grid_dump: null output");
  if (!h->has_target) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "grid_dump before SetInputTarget");
  if ((rc = ndt_materialize(h))) return rc;
  size_t V = h->vi.nocc;
  *n_cells = V;
  if (!cell_ints && !cell_rec) return LSA_OK;
  if (cap < V) return lsa_fail(ctx, LSA_ERR_CAPACITY, "grid_dump: cap %zu < %zu cells", cap, V);
  std::vector<int32_t> cor(V), cnt(V), st(V);
  LSA_HIP(ctx, hipMemcpyAsync(cor.data(), h->vi.cell_of_rank.p, V * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(cnt.data(), h->counts.p, V * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(st.data(), h->state.p, V * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (cell_rec) LSA_HIP(ctx, hipMemcpyAsync(cell_rec, h->rec.p, V * kRecDoubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (cell_ints) {
    const CellMap &m = h->vi.map;
    for (size_t r = 0; r < V; r++) {
      int cell = cor[r];
      int k = cell / (m.dim[0] * m.dim[1]);
      int rem = cell - k * (m.dim[0] * m.dim[1]);
      int j = rem / m.dim[0], i = rem - j * m.dim[0];
      cell_ints[5 * r] = i + m.min_b[0]; cell_ints[5 * r + 1] = j + m.min_b[1]; cell_ints[5 * r + 2] = k + m.min_b[2];
      cell_ints[5 * r + 3] = cnt[r]; cell_ints[5 * r + 4] = st[r];
    }
  }
  return LSA_OK;
}
