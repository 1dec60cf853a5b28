// voxel_filter.hip — lsa_voxel_filter_*: the CloudFilterInterface::Filter replacement
// (reference: lidar_localization/src/models/cloud_filter/voxel_filter.cpp:36-41 -> pcl::VoxelGrid::filter).
//
// Pipeline on the device (no sort: ranks of the occupancy bitmap are already in PCL's output order):
//   bbox reduce -> grid definition by PCL's min_b / div_b rules (on the device) -> mark bitmap -> single-pass popcount
//   scan -> fixed-point accumulate -> finalize (mean of x,y,z,intensity per voxel; also zeroes what the build touched).
// Algorithmic HBM bytes (SURVEY.md 8d formula A): 16*N read + 16*V written; this implementation re-reads
// the points once per streaming pass (bbox, mark, accumulate: 3 x 16*N + 2 x 4*N cell ids).
#include "voxel_index_dev.h"

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

struct lsa_voxel_filter {
  lsa_context *ctx = nullptr;
  float leaf[3] = {0, 0, 0};
  VoxelIndex vi;
  DevBuf<float4> d_in, d_out;
  DevBuf<float4> d_pipe;   // third output buffer of lsa_pipeline_scan_match (beside the caller's two)
  PinnedBuf<float> h_stage;
  // asynchronous call in flight (lsa_voxel_filter_apply_device_async .. lsa_voxel_filter_wait)
  bool pending = false;
  const float4 *p_in = nullptr;
  float4 *p_out = nullptr;
  size_t p_n = 0;
  // The ~10 kernel launches of one filter cost ~30 us of HOST time; the async entry point hands them to a helper
  // thread so that the caller's thread is free to drive the registration of the previous scan meanwhile.
  // job_state: 0 idle, 1 posted, 3 claimed by the helper, 2 done (job_rc valid), -1 shut down.
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> job_state{0};
  int job_rc = LSA_OK;
  hipError_t job_hip = hipSuccess;   // kind 2: the helper's HIP status (the caller's thread turns it into the error text)
  bool worker_started = false;
  bool retiring = false;   // job kind 3 in flight (lsa_voxel_filter_retire_begin / _end)
  // job kind 1: enqueue the filter of (p_in, p_n, p_out); kind 2: pack + upload the second half of a host cloud on stream2
  int job_kind = 1;
  const void *u_pts = nullptr;
  size_t u_n = 0, u_stride = 0, u_w_off = 0, u_off = 0;
  unsigned long long n_miss = 0, n_calls = 0, n_cap = 0;
  hipStream_t stream2 = nullptr;   // second copy queue of the host path (two SDMA queues move 2 MB in ~50 us, one in ~73)
  hipEvent_t ev2 = nullptr;
  // job kind 4 (the pulled host path, lsa_voxel_filter_apply): both threads claim chunks of the caller's cloud from u_next, pack them
  // into h_stage and publish ready[chunk] = pull_seq; the mark kernel (already enqueued) takes each chunk over the link as it appears
  PinnedBuf<unsigned> h_ready;
  // a second packer (packing 131 k rows is ~70 us of one core's memory bandwidth; the link needs 44): same claim loop, own thread.
  // p2_state: 0 idle, 1 posted, 3 packing, 2 done, -1 shut down
  std::thread packer2;
  std::mutex mu2;
  std::condition_variable cv2;
  std::atomic<int> p2_state{0};
  bool packer2_started = false;
  std::atomic<unsigned> u_next{0};
  unsigned u_chunks = 0, pull_seq = 0;
  bool no_pull = getenv("LSA_FILTER_PULL") ? atoi(getenv("LSA_FILTER_PULL")) == 0 : false;
  unsigned long long n_pulled = 0;
  // a pulled build whose first kernel gave up waiting for the packed rows (PullSource::patience_ticks, 2 s: the packing threads were
  // held up for seconds).  The synchronous call fails with LSA_ERR_HIP; the asynchronous one, whose rows are complete in the staging
  // buffer by the time it is waited for, sends them again and builds the ordinary way.  Counted either way.
  unsigned long long n_pull_timeouts = 0;
  // LSA_PULL_TEST="patience_ticks,delay_us" (tests only): the kernel's patience in 100 MHz ticks and a sleep in front of every packing
  // thread's first chunk, so that the time-out can be met on purpose
  unsigned long long pull_patience = 200000000ull;   // 2 s
  unsigned pull_test_delay_us = 0;
  // LSA_FILTER_STATS: where a pulled call spends its time (sums of microseconds) and how many chunks the caller's thread packed itself
  bool stats_on = getenv("LSA_FILTER_STATS") != nullptr;
  double st_enqueue = 0, st_pack = 0, st_join = 0, st_wait = 0, st_unpack = 0;
  unsigned long long st_main_chunks = 0, st_withdrawn = 0;
};

namespace {
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_filter_finalize(long long *__restrict__ acc, int32_t *__restrict__ cnt,
                                                            const int32_t *__restrict__ cell_of_rank, DevParams *__restrict__ dp,
                                                            float4 *__restrict__ out, DevParams *__restrict__ host_mirror, uint32_t *__restrict__ bits,
                                                            int by_cell /* rows of acc / cnt are keyed by the cell, not by the rank (VoxelIndex::by_cell) */) {
  // the grid definition + voxel count + status go straight into host-mapped pinned memory (no copy kernel): the host
  // reads them after the completion event.  The miss flag of a speculative build is consumed here (no other thread of
  // this kernel reads it), so the definition left on the device is ready for the next build.
  if (blockIdx.x == 0 && threadIdx.x == 0) { *host_mirror = *dp; dp->spec_miss = 0; }
  if (dp->status != VI_OK) return;
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= dp->V) return;  // the grid is sized for the point count; the occupied-cell count only exists on the device
  const CellMap m = dp->map;
  const FixScales fs = dp->fs;
  double o[3];
  const int cell = cell_of_rank[r];
  // one 64-byte row per occupied cell, keyed by the cell or by its rank: four sums and, as the fifth word, the count
  long long *a = acc + (by_cell ? (size_t)cell : r) * kByCellStride;
  lsa_cell_origin(m, cell, o);
  const long long av[4] = {a[0], a[1], a[2], a[3]};
  out[r] = vi_filter_mean(av, (int)a[4], o, fs);
  // leave the index clean for the next scan (was a separate clean-up launch): this cell's sums, its count and its
  // bitmap word (cells sharing the word all store the same zero)
  a[0] = 0; a[1] = 0; a[2] = 0; a[3] = 0; a[4] = 0;
  bits[cell >> 5] = 0u;
}

void filter_pull_join(lsa_voxel_filter *f);

// pull (host path only): the first attempt's mark kernel takes the rows out of pinned memory while both threads are still packing
// them (filter_pull_join: this thread packs too, then waits for the helper); d_in is where the kernel leaves them, so a rebuild
// after a speculative miss finds them resident
// d_keep: the mark kernel of every attempt also copies the rows it reads there (lsa_voxel_filter_apply_device_keep)
// after_enqueue (pulled attempt only): called once the first attempt's kernels are enqueued, before this thread starts packing —
// the caller's own launch behind the filter's (FrontEnd::Update: the registration's first pass kernel, parked for its mail)
int filter_device(lsa_voxel_filter *f, const float4 *d_in, size_t n, float4 *d_out, size_t *n_out, const PullSource *pull = nullptr, float4 *d_keep = nullptr,
                  void (*after_enqueue)(void *) = nullptr, void *after_user = nullptr) {
  lsa_context *ctx = f->ctx;
  *n_out = 0;
  if (n == 0) return LSA_OK;
  // The whole pipeline (bbox -> grid definition -> bitmap -> scan -> sums -> means) is enqueued without a host round
  // trip; the host then waits ONCE for the 200-byte parameter block that carries the voxel count and the status.
  VoxelIndex &vi = f->vi;
  int rc, status = VI_OK;
  f->n_calls++;
  for (int attempt = 0; attempt < 3; attempt++) {
    const bool pulling = pull && attempt == 0;
    const auto tq0 = std::chrono::steady_clock::now();
    rc = vi_enqueue(ctx, vi, d_in, n, /*mode PCL*/ 0, f->leaf, 4, /*allow_spec*/ true, pulling ? pull : nullptr, d_keep);
    if (!rc) {
      hipLaunchKernelGGL(k_filter_finalize, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, vi.acc.p, vi.cnt.p, vi.cell_of_rank.p, vi.dparams.p,
                         d_out, vi.h_params.p, vi.bits.p, vi.by_cell ? 1 : 0);
      if (hipGetLastError() != hipSuccess) rc = lsa_fail(ctx, LSA_ERR_HIP, "voxel filter: launch failed");
    }
    if (!rc) rc = vi_mark_done(ctx, vi);
    if (pulling && f->stats_on) f->st_enqueue += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count();
    if (pulling && !rc && after_enqueue) after_enqueue(after_user);
    if (pulling) filter_pull_join(f);   // (also behind a failed enqueue: the helper thread is packing the caller's cloud and must be done before this returns)
    if (rc) return rc;
    const auto tw0 = std::chrono::steady_clock::now();
    if ((rc = vi_wait(ctx, vi, &status))) return rc;
    if (pulling && f->stats_on) f->st_wait += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw0).count();
    vi.clean = (status == VI_OK || status == VI_SPEC_MISS);  // the finalize kernel zeroed what this build touched
    if (status == VI_SPEC_MISS && vi.last_pull_timeout) {
      f->n_pull_timeouts++;
      return lsa_fail(ctx, LSA_ERR_HIP, "voxel filter: the device waited in vain for the packed rows of the host cloud");
    }
    if (after_enqueue && (status == VI_SPEC_MISS || status == VI_CAPACITY)) lsa_ctx_unpark(ctx);   // (the repeat goes in front of nothing parked)
    if (status == VI_SPEC_MISS) { f->n_miss++; continue; }  // a point fell outside the reused cell box: rebuild from the bounding box
    if (status != VI_CAPACITY) break;
    f->n_cap++;
    if ((rc = vi_grow(ctx, vi))) return rc;  // first call / bigger extent than ever seen: size the bitmap and redo
  }
  if (status == VI_RANGE) return lsa_fail(ctx, LSA_ERR_RANGE, "voxel filter: leaf size too small for the input dataset, integer indices would overflow");
  if (status == VI_CAPACITY) return lsa_fail(ctx, LSA_ERR_HIP, "voxel filter: bitmap capacity could not be established");
  *n_out = (status == VI_OK) ? vi.nocc : 0;
  return LSA_OK;
}

// the enqueue half of filter_device (attempt 0 only); filter_wait() completes it and falls back to the synchronous path
// if the bitmap capacity has to grow
int filter_enqueue(lsa_voxel_filter *f, const float4 *d_in, size_t n, float4 *d_out, const PullSource *pull = nullptr, float4 *d_keep = nullptr) {
  lsa_context *ctx = f->ctx;
  VoxelIndex &vi = f->vi;
  int rc;
  // Same fast variants as the synchronous call (512-point tiles, speculative reuse of the previous grid definition).
  // While the registration still waited for its stream after every pass, a lighter background filter made the overlapped
  // pipeline slower; since the passes hand their rows over without a stream wait it is the other way round (5 A/B pairs:
  // +6 %).  A point outside the reused cell box is caught by filter_wait(), which rebuilds exactly.
  rc = vi_enqueue(ctx, vi, d_in, n, /*mode PCL*/ 0, f->leaf, 4, /*allow_spec*/ true, pull, d_keep);
  if (rc) return rc;
  hipLaunchKernelGGL(k_filter_finalize, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, vi.acc.p, vi.cnt.p, vi.cell_of_rank.p, vi.dparams.p, d_out,
                     vi.h_params.p, vi.bits.p, vi.by_cell ? 1 : 0);
  LSA_HIP(ctx, hipGetLastError());
  return vi_mark_done(ctx, vi);  // filter_wait() marks the index clean once the status is known
}
}  // namespace

namespace {
constexpr unsigned kPullChunkShift = 13;   // 8 192 rows = 128 KB per chunk: ~4 us of packing, ~3 us on the link
// claim chunks until none is left: pack, then publish (rows before the flag; the kernel acquires the flag at system scope)
unsigned filter_pack_chunks(lsa_voxel_filter *f) {
  const size_t chunk = (size_t)1 << kPullChunkShift;
  if (f->pull_test_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(f->pull_test_delay_us));
  for (unsigned done = 0;; done++) {
    const unsigned c = f->u_next.fetch_add(1, std::memory_order_relaxed);
    if (c >= f->u_chunks) return done;
    const size_t at = (size_t)c * chunk, m = f->u_n - at < chunk ? f->u_n - at : chunk;
    lsa_pack_xyzw((const char *)f->u_pts + at * f->u_stride, m, f->u_stride, f->u_w_off, f->h_stage.p + 4 * at);
    _mm_sfence();
    __atomic_store_n(f->h_ready.p + (size_t)c * kPullFlagStride, f->pull_seq, __ATOMIC_RELEASE);
  }
}

// the second packer's thread: spins for 200 us after a job (scans come back to back in a loop), then sleeps in short naps — it never
// touches HIP, the context or anything but the claim counter, the caller's rows, the staging buffer and the flags
void filter_packer2(lsa_voxel_filter *f) {
  for (;;) {
    auto t0 = std::chrono::steady_clock::now();
    int st;
    for (unsigned spin = 0;; spin++) {
      st = f->p2_state.load(std::memory_order_acquire);
      if (st == 1 || st == -1) break;
      __builtin_ia32_pause();
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    if (st != 1 && st != -1) {   // nothing for 200 us: sleep until the next job is posted (an idle handle costs no CPU)
      std::unique_lock<std::mutex> lk(f->mu2);
      f->cv2.wait(lk, [&] { int s = f->p2_state.load(std::memory_order_acquire); return s == 1 || s == -1; });
      st = f->p2_state.load(std::memory_order_acquire);
    }
    if (st == -1) return;
    int expect = 1;
    if (!f->p2_state.compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) continue;   // withdrawn meanwhile
    (void)filter_pack_chunks(f);
    f->p2_state.store(2, std::memory_order_release);
  }
}

void filter_worker(lsa_voxel_filter *f) {
  (void)hipSetDevice(f->ctx->device);
  for (;;) {
    // spin briefly (scans arrive back to back in a pipeline), then sleep on the condition variable
    int st = 0;
    const auto spin_t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
      st = f->job_state.load(std::memory_order_acquire);
      if (st == 1 || st == -1) break;
      __builtin_ia32_pause();
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - spin_t0 > std::chrono::microseconds(100)) break;
    }
    if (st != 1 && st != -1) {
      std::unique_lock<std::mutex> lk(f->mu);
      f->cv.wait(lk, [&] { int s = f->job_state.load(std::memory_order_acquire); return s == 1 || s == -1; });
      st = f->job_state.load(std::memory_order_acquire);
    }
    if (st == -1) return;
    // Claimed (1 -> 3), not just seen, and the kind read only BEHIND the claim: the caller of the pulled path withdraws a job the
    // helper has not picked up by the time every chunk is packed (1 -> 0; a helper that was asleep needs ~50 us to wake — longer
    // than the whole cloud takes to pack) and may post a different one at once; a kind read in front of the claim could belong to
    // the withdrawn job.  From the claim to state 2 (or 0, the wake-up call) the job's fields are this thread's.
    {
      int expect = 1;
      if (!f->job_state.compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) continue;
    }
    const int kind = f->job_kind;
    if (kind == 0) {  // wake-up call only (the first filter of a batch is enqueued by the caller itself): back to spinning
      f->job_state.store(0, std::memory_order_release);
      continue;
    }
    if (kind == 3) {
      // retire the stream's completed commands (the last filter of a batch has delivered; its stream would otherwise cost the
      // caller's next wait on the device ~11 us of bookkeeping) while the caller's thread drives the last registration
      f->job_hip = hipStreamSynchronize(f->ctx->stream);
      f->job_rc = f->job_hip == hipSuccess ? LSA_OK : LSA_ERR_HIP;
      f->job_state.store(2, std::memory_order_release);
      continue;
    }
    if (kind == 4) {
      (void)filter_pack_chunks(f);
      f->job_rc = LSA_OK;
      f->job_state.store(2, std::memory_order_release);
      continue;
    }
    if (kind == 5) {
      // A HOST scan through the filter, asynchronously (lsa_voxel_filter_apply_host_async): with a reusable grid definition the
      // build is enqueued first — its mark kernel pulls the rows over the link as they are published and leaves them in p_in —
      // and this thread and the second packer then pack the caller's rows chunk by chunk; without one, the rows are packed and
      // sent by copy commands in front of the ordinary build.  Either way the caller's thread is free meanwhile.
      const size_t n = f->u_n;
      int rc;
      if (n >= 65536 && !f->no_pull && vi_spec_ready(f->vi, n, 0, f->leaf)) {
        const size_t chunk = (size_t)1 << kPullChunkShift;
        f->u_chunks = (unsigned)((n + chunk - 1) / chunk);
        if (++f->pull_seq == 0) f->pull_seq = 1;
        f->u_next.store(0, std::memory_order_relaxed);
        PullSource pull;
        pull.h_rows = reinterpret_cast<const float4 *>(f->h_stage.p);
        pull.d_copy = const_cast<float4 *>(f->p_in);
        pull.ready = f->h_ready.p;
        pull.seq = f->pull_seq;
        pull.chunk_shift = kPullChunkShift;
        pull.patience_ticks = f->pull_patience;
        if (f->packer2_started) {
          {
            std::lock_guard<std::mutex> lk(f->mu2);
            f->p2_state.store(1, std::memory_order_release);
          }
          f->cv2.notify_one();
        }
        f->n_pulled++;
        rc = filter_enqueue(f, f->p_in, n, f->p_out, &pull);
        (void)filter_pack_chunks(f);   // (also behind a failed enqueue: harmless, and the second packer may be at it)
        for (;;) {   // the second packer: done, or never started on it
          int st2 = f->p2_state.load(std::memory_order_acquire);
          if (st2 == 0) break;
          if (st2 == 2) { f->p2_state.store(0, std::memory_order_release); break; }
          if (st2 == 1 && f->p2_state.compare_exchange_strong(st2, 0, std::memory_order_acq_rel)) break;
          __builtin_ia32_pause();
        }
      } else {
        // (no copy command: the first one on this handle's stream would set up a copy queue — ~7 ms, seen as the first filter of
        // every other fresh front end taking that long — and a kernel reads pinned memory at the link's rate anyway)
        lsa_pack_xyzw(f->u_pts, n, f->u_stride, f->u_w_off, f->h_stage.p);
        hipError_t e = lsa_pull_rows(f->ctx->stream, f->h_stage.p, const_cast<float4 *>(f->p_in), n);
        f->job_hip = e;
        rc = e == hipSuccess ? filter_enqueue(f, f->p_in, n, f->p_out) : LSA_ERR_HIP;
      }
      f->job_rc = rc;
      f->job_state.store(2, std::memory_order_release);
      continue;
    }
    if (kind == 2) {
      // second half of a host cloud down the second copy queue.  The helper touches neither the context nor its error text
      // (the caller's thread may be writing it): stride / offset were validated by the caller, a HIP failure comes back as
      // a code in job_hip and the caller's thread reports it
      const size_t kChunk = lsa_upload_chunk();
      hipError_t e = hipSuccess;
      for (size_t off = 0; off < f->u_n && e == hipSuccess; off += kChunk) {
        const size_t m = f->u_n - off < kChunk ? f->u_n - off : kChunk, at = f->u_off + off;
        lsa_pack_xyzw((const char *)f->u_pts + at * f->u_stride, m, f->u_stride, f->u_w_off, f->h_stage.p + 4 * at);
        e = hipMemcpyAsync(f->d_in.p + at, f->h_stage.p + 4 * at, m * sizeof(float4), hipMemcpyHostToDevice, f->stream2);
      }
      if (e == hipSuccess) e = hipEventRecord(f->ev2, f->stream2);
      f->job_hip = e;
      f->job_rc = e == hipSuccess ? LSA_OK : LSA_ERR_HIP;
    } else {
      f->job_rc = filter_enqueue(f, f->p_in, f->p_n, f->p_out);
    }
    f->job_state.store(2, std::memory_order_release);
  }
}
}  // namespace

namespace {
void filter_post_job(lsa_voxel_filter *f, int kind) {
  if (!f->worker_started) {
    f->worker = std::thread(filter_worker, f);
    f->worker_started = true;
  }
  for (;;) {   // (a wake-up call the helper has not taken yet, or is just taking)
    const int s = f->job_state.load(std::memory_order_acquire);
    if (s != 1 && s != 3) break;
    __builtin_ia32_pause();
  }
  f->job_kind = kind;
  {
    std::lock_guard<std::mutex> lk(f->mu);
    f->job_state.store(1, std::memory_order_release);
  }
  f->cv.notify_one();
}
}  // namespace

namespace {
void filter_pull_join(lsa_voxel_filter *f) {
  const auto t0 = std::chrono::steady_clock::now();
  const unsigned mine = filter_pack_chunks(f);
  const auto t1 = std::chrono::steady_clock::now();
  bool withdrawn = false;
  for (;;) {   // every chunk is claimed; the helper may still be packing its last one — or may never have woken up
    int st = f->job_state.load(std::memory_order_acquire);
    if (st == 2) { f->job_state.store(0, std::memory_order_release); break; }
    if (st == 1 && f->job_state.compare_exchange_strong(st, 0, std::memory_order_acq_rel)) { withdrawn = true; break; }   // withdrawn before it was claimed
    __builtin_ia32_pause();
  }
  for (;;) {   // the same for the second packer
    int st = f->p2_state.load(std::memory_order_acquire);
    if (st == 0) break;
    if (st == 2) { f->p2_state.store(0, std::memory_order_release); break; }
    if (st == 1 && f->p2_state.compare_exchange_strong(st, 0, std::memory_order_acq_rel)) break;
    __builtin_ia32_pause();
  }
  if (f->stats_on) {
    f->st_pack += std::chrono::duration<double, std::micro>(t1 - t0).count();
    f->st_join += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    f->st_main_chunks += mine;
    f->st_withdrawn += withdrawn ? 1 : 0;
  }
}
}  // namespace

extern "C" int lsa_voxel_filter_apply_device_async(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw) {
  if (!f || (n && (!d_in_xyzw || !d_out_xyzw))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device_async: null argument") : LSA_ERR_INVALID;
  if (f->pending) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device_async: a filter is already in flight on this handle");
  lsa_ctx_unpark(f->ctx);  // the helper thread enqueues on this context's stream
  f->p_in = (const float4 *)d_in_xyzw; f->p_out = (float4 *)d_out_xyzw; f->p_n = n;
  f->pending = true;
  if (n == 0) return LSA_OK;
  filter_post_job(f, 1);
  return LSA_OK;
}

// internal (pipeline.cpp): the same call with the launches issued by the CALLER's thread — for the first scan of a batch,
// where nothing is there to overlap with and the helper thread, asleep since the last batch, would take longer to wake up
// (~50 us) than the ~20 us of launch calls cost; the helper is woken meanwhile so that it is spinning when the second scan's
// filter is posted.  Completed by lsa_voxel_filter_wait like any other.
int lsa_voxel_filter_apply_device_async_here(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw) {
  if (!f || (n && (!d_in_xyzw || !d_out_xyzw))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device_async: null argument") : LSA_ERR_INVALID;
  if (f->pending) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device_async: a filter is already in flight on this handle");
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  f->p_in = (const float4 *)d_in_xyzw; f->p_out = (float4 *)d_out_xyzw; f->p_n = n;
  f->pending = true;
  if (n == 0) return LSA_OK;
  filter_post_job(f, 0);
  f->job_rc = filter_enqueue(f, f->p_in, f->p_n, f->p_out);
  while (f->job_state.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();   // the wake-up call has been taken (1 -> 3 -> 0)
  f->job_state.store(2, std::memory_order_release);
  return LSA_OK;
}

// internal (pipeline.cpp): a device buffer of n points owned by this handle (grow-only)
// internal (pipeline.cpp: lsa_pipeline_scan_match_host): the handle's own device buffers for a run of host scans — which = 0 the raw
// rows (d_in), 1 / 2 two filtered clouds (d_out, d_pipe) — sized for n points (grow-only; the caller makes sure nothing is in flight)
float *lsa_voxel_filter_host_buffer(lsa_voxel_filter *f, size_t n, int which) {
  if (!f || which < 0 || which > 2) return nullptr;
  DevBuf<float4> &b = which == 0 ? f->d_in : (which == 1 ? f->d_out : f->d_pipe);
  if ((n ? n : 1) > b.cap) {
    if (hipSetDevice(f->ctx->device) != hipSuccess || hipStreamSynchronize(f->ctx->stream) != hipSuccess) return nullptr;
    if (b.reserve(f->ctx, n ? n : 1, 1.1)) return nullptr;
  }
  return (float *)b.p;
}

float *lsa_voxel_filter_pipe_buffer(lsa_voxel_filter *f, size_t n) {
  if (!f) return nullptr;
  // This is the first thing lsa_pipeline_scan_match does, BEFORE any call that enters the filter's context: a worker thread driving
  // a context on GPU n > 0 still has device 0 current, and a buffer allocated now would land there (the filter of every third scan
  // would write it, the NDT pass read it, from another device).  Select the context's device — nothing else of lsa_enter: a kernel
  // parked on the context's stream has no business being released by an allocation.
  if ((n ? n : 1) > f->d_pipe.cap && hipSetDevice(f->ctx->device) != hipSuccess) return nullptr;
  if (f->d_pipe.reserve(f->ctx, n ? n : 1, 1.1)) return nullptr;
  return (float *)f->d_pipe.p;
}

// internal (front_end.hip): the asynchronous Filter of a HOST scan — raw rows to d_rows, filtered cloud to d_out, both on the device —
// completed by lsa_voxel_filter_wait like lsa_voxel_filter_apply_device_async.  The caller's rows must stay valid until the wait.
int lsa_voxel_filter_apply_host_async(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float *d_rows, float *d_out) {
  if (!f || (n && (!in_pts || !d_rows || !d_out))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_host_async: null argument") : LSA_ERR_INVALID;
  if (f->pending) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_host_async: a filter is already in flight on this handle");
  lsa_context *ctx = f->ctx;
  if (n && (in_stride < 12 || (in_w_off != SIZE_MAX && in_w_off + 4 > in_stride)))
    return lsa_fail(ctx, LSA_ERR_INVALID, "point rows of %zu bytes cannot hold xyz + an intensity at offset %zu (pass SIZE_MAX for clouds without one)", in_stride, in_w_off);
  lsa_ctx_unpark(ctx);  // the helper thread enqueues on this context's stream
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  const size_t chunk = (size_t)1 << kPullChunkShift, chunks = (n + chunk - 1) / chunk;
  if (n * 4 > f->h_stage.cap || chunks * kPullFlagStride > f->h_ready.cap) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (growing frees the old buffers)
  if ((rc = f->h_stage.reserve(ctx, (n ? n : 1) * 4))) return rc;
  if ((rc = f->h_ready.reserve(ctx, (chunks ? chunks : 1) * kPullFlagStride))) return rc;
  if (!f->packer2_started && n >= 65536 && !f->no_pull) {
    f->packer2 = std::thread(filter_packer2, f);
    f->packer2_started = true;
  }
  f->u_pts = in_pts; f->u_stride = in_stride; f->u_w_off = in_w_off; f->u_off = 0; f->u_n = n;
  f->p_in = (const float4 *)d_rows; f->p_out = (float4 *)d_out; f->p_n = n;
  f->pending = true;
  if (n == 0) return LSA_OK;
  filter_post_job(f, 5);
  return LSA_OK;
}

// internal (pipeline.cpp): see job kind 3.  _begin posts it to the helper thread, _end waits for it.
void lsa_voxel_filter_retire_begin(lsa_voxel_filter *f) {
  if (!f || f->pending || !f->worker_started) return;
  filter_post_job(f, 3);
  f->retiring = true;
}
void lsa_voxel_filter_retire_end(lsa_voxel_filter *f) {
  if (!f || !f->retiring) return;
  while (f->job_state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
  f->job_state.store(0, std::memory_order_release);
  f->retiring = false;
}

// internal (front_end.hip): lsa_voxel_filter_apply_device[_keep] in two halves on the CALLER's thread — the launches now, the wait in
// lsa_voxel_filter_wait_parked — so that the caller can enqueue something of its own behind the filter's kernels in between (the
// registration's first pass kernel, parked for its mail: FrontEnd::Update).  d_keep may be null.
int lsa_voxel_filter_enqueue_device(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, float *d_keep) {
  if (!f || (n && (!d_in_xyzw || !d_out_xyzw))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_enqueue_device: null argument") : LSA_ERR_INVALID;
  if (f->pending) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_enqueue_device: a filter is already in flight on this handle");
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  f->p_in = (const float4 *)d_in_xyzw; f->p_out = (float4 *)d_out_xyzw; f->p_n = n;
  f->pending = true;
  if (n == 0) return LSA_OK;
  f->job_rc = filter_enqueue(f, f->p_in, n, f->p_out, nullptr, (float4 *)d_keep);
  f->job_state.store(2, std::memory_order_release);
  return LSA_OK;
}

static int filter_wait_impl(lsa_voxel_filter *f, size_t *n_out, bool keep_parked);

// ... the second half: like lsa_voxel_filter_wait, but a kernel parked on this context's stream behind the filter's stays parked
// unless the build has to be repeated (a point outside the reused cell box, a bitmap to grow: then it is released first — the
// repeat's kernels would queue behind it for its whole patience)
int lsa_voxel_filter_wait_parked(lsa_voxel_filter *f, size_t *n_out) { return filter_wait_impl(f, n_out, true); }

extern "C" int lsa_voxel_filter_wait(lsa_voxel_filter *f, size_t *n_out) { return filter_wait_impl(f, n_out, false); }

static int filter_wait_impl(lsa_voxel_filter *f, size_t *n_out, bool keep_parked) {
  if (!f || !n_out) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_wait: null argument") : LSA_ERR_INVALID;
  if (!f->pending) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_wait: nothing in flight");
  f->pending = false;
  *n_out = 0;
  if (f->p_n == 0) return LSA_OK;
  if (keep_parked) LSA_HIP(f->ctx, hipSetDevice(f->ctx->device));
  else LSA_HIP(f->ctx, lsa_enter(f->ctx));
  while (f->job_state.load(std::memory_order_acquire) != 2) {}  // the helper is still issuing the launches (<= ~30 us)
  f->job_state.store(0, std::memory_order_release);
  if (f->job_rc) return f->job_rc;
  int status = VI_OK;
  int rc = vi_wait(f->ctx, f->vi, &status);
  if (rc) return rc;
  f->vi.clean = (status == VI_OK || status == VI_SPEC_MISS);  // the finalize kernel zeroed what this build touched
  if (keep_parked) {
    if (status == VI_OK || status == VI_RANGE) f->n_calls++;   // (a repeat counts itself, in filter_device)
    if (status == VI_SPEC_MISS && !f->vi.last_pull_timeout) f->n_miss++;
    if (status == VI_CAPACITY) f->n_cap++;
    if (status == VI_SPEC_MISS || status == VI_CAPACITY) lsa_ctx_unpark(f->ctx);   // (the repeat goes in front of nothing parked)
  }
  if (status == VI_SPEC_MISS && f->vi.last_pull_timeout) {
    // Not a point outside the reused box: the pulling kernel's tiles gave up waiting for their chunks and never wrote those rows
    // into p_in — a rebuild from p_in would filter whatever the buffer held before.  The rows themselves are complete by now (the
    // helper reports the job done only behind the last packed chunk, both packers joined) and still in the staging buffer (the
    // asynchronous paths deliver into device memory, not into it): send them again and build the ordinary way.
    f->n_pull_timeouts++;
    hipError_t e = lsa_pull_rows(f->ctx->stream, f->h_stage.p, const_cast<float4 *>(f->p_in), f->p_n);
    if (e != hipSuccess) return lsa_fail(f->ctx, LSA_ERR_HIP, "voxel filter: re-sending the rows of a host cloud failed: %s", hipGetErrorString(e));
    return filter_device(f, f->p_in, f->p_n, f->p_out, n_out);
  }
  if (status == VI_SPEC_MISS) return filter_device(f, f->p_in, f->p_n, f->p_out, n_out);  // rare: rebuild exactly
  if (status == VI_CAPACITY) {  // rare: extent larger than ever seen; size the bitmap and run the synchronous path
    if ((rc = vi_grow(f->ctx, f->vi))) return rc;
    return filter_device(f, f->p_in, f->p_n, f->p_out, n_out);
  }
  if (status == VI_RANGE) return lsa_fail(f->ctx, LSA_ERR_RANGE, "voxel filter: leaf size too small for the input dataset, integer indices would overflow");
  *n_out = (status == VI_OK) ? f->vi.nocc : 0;
  return LSA_OK;
}

// internal (front_end.hip): VoxelIndex::scale_n_min of this filter's index
void lsa_voxel_filter_set_scale_floor(lsa_voxel_filter *f, unsigned long long n) {
  if (f && f->vi.scale_n_min != n) { f->vi.scale_n_min = n; f->vi.spec_valid = false; }
}

extern "C" int lsa_voxel_filter_create(lsa_voxel_filter **out, lsa_context *ctx, float lx, float ly, float lz) {
  if (!out || !ctx) { lsa_set_global_error("lsa_voxel_filter_create: null argument"); return LSA_ERR_INVALID; }
  if (!(lx > 0) || !(ly > 0) || !(lz > 0)) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_voxel_filter_create: leaf sizes must be > 0");
  lsa_voxel_filter *f = new lsa_voxel_filter();
  f->ctx = ctx;
  f->leaf[0] = lx; f->leaf[1] = ly; f->leaf[2] = lz;
  f->vi.allow_by_cell = true;   // k_filter_finalize reads and clears the rows by their key
  if (const char *t = getenv("LSA_PULL_TEST")) {
    unsigned long long ticks = 0;
    unsigned delay = 0;
    if (sscanf(t, "%llu,%u", &ticks, &delay) >= 1 && ticks) { f->pull_patience = ticks; f->pull_test_delay_us = delay; }
  }
  *out = f;
  return LSA_OK;
}

// Allocate ahead what filtering clouds of up to max_points will need (index buffers, a bitmap of 2^17 words = 4.2 M cells: a
// 200 m x 200 m x 25 m box at a 0.5 m leaf with the speculative margin; the scan kernel's grid covers the capacity, so a much
// larger one costs every build: 2^20 words made k_scan 9.3 instead of 7.2 us; a larger box still grows it once), and start the helper thread of the asynchronous entry
// point if `async_too`: after this a Filter call over such a cloud allocates nothing and finds the bitmap large enough on its
// first build.  Optional — every call sizes what it needs — and without effect on any result.  async_too: bit 0 the helper thread of the
// asynchronous entry point, bit 1 what the host-cloud entry points use (pinned staging rows, chunk flags, the second packing thread).
extern "C" int lsa_voxel_filter_reserve(lsa_voxel_filter *f, size_t max_points, int async_too) {
  if (!f) return LSA_ERR_INVALID;
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  int rc = vi_prepare(f->ctx, f->vi, max_points ? max_points : 1, 4, (size_t)1 << 17);
  if (rc) return rc;
  // the rows keyed by the cell (one 64-byte line per cell of the bitmap's capacity: 272 MB here) and, for clouds of half a million
  // points and more, the byte map of the large path: now, not under the first call
  if ((rc = vi_prepare_large(f->ctx, f->vi, 4, max_points >= 512 * 1024))) return rc;
  if (!f->vi.clean) {   // the bitmap and the accumulator rows are cleared here instead of under the first Filter call
    VoxelIndex &vi = f->vi;
    LSA_HIP(f->ctx, hipMemsetAsync(vi.bits.p, 0, vi.W_cap * sizeof(uint32_t), f->ctx->stream));
    LSA_HIP(f->ctx, hipMemsetAsync(vi.acc.p, 0, vi.acc.cap * sizeof(long long), f->ctx->stream));
    LSA_HIP(f->ctx, hipMemsetAsync(vi.cnt.p, 0, vi.cnt.cap * sizeof(int32_t), f->ctx->stream));
    vi.acc_ns = 4;
    vi.clean = true;
  }
  if ((async_too & 1) && !f->worker_started) {
    f->worker = std::thread(filter_worker, f);
    f->worker_started = true;
  }
  if (async_too & 2) {   // the host-cloud path: pinned staging rows, chunk flags, the second packing thread
    const size_t n = max_points ? max_points : 1, chunk = (size_t)1 << kPullChunkShift;
    if ((rc = f->h_stage.reserve(f->ctx, n * 4))) return rc;
    if ((rc = f->h_ready.reserve(f->ctx, ((n + chunk - 1) / chunk) * kPullFlagStride))) return rc;
    if (!f->packer2_started) {
      f->packer2 = std::thread(filter_packer2, f);
      f->packer2_started = true;
    }
  }
  return LSA_OK;
}

// diagnostics: out[0] synchronous builds, [1] speculative misses (built twice), [2] bitmap growths, [3] host clouds whose rows the
// mark kernel pulled over the link itself (lsa_voxel_filter_apply's steady state)
extern "C" int lsa_voxel_filter_counters(lsa_voxel_filter *f, uint64_t out[4]) {
  if (!f || !out) return LSA_ERR_INVALID;
  out[0] = f->n_calls; out[1] = f->n_miss; out[2] = f->n_cap; out[3] = f->n_pulled;
  return LSA_OK;
}

// diagnostics: see lidar_slam_amd.h
extern "C" int lsa_voxel_filter_stamps(lsa_voxel_filter *f, int enable, uint64_t *out, size_t cap, size_t *per_kernel) {
  if (!f) return LSA_ERR_INVALID;
  lsa_context *ctx = f->ctx;
  LSA_HIP(ctx, lsa_enter(ctx));
  const size_t n = 3 * kViStampWgs * 8;
  if (per_kernel) *per_kernel = kViStampWgs * 8;
  if (!out) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!enable) { f->vi.stamps.release(); return LSA_OK; }
    int rc = f->vi.stamps.reserve(ctx, n);
    if (rc) return rc;
    LSA_HIP(ctx, hipMemsetAsync(f->vi.stamps.p, 0, n * sizeof(unsigned long long), ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LSA_OK;
  }
  if (!f->vi.stamps.p) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_voxel_filter_stamps: not enabled");
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LSA_HIP(ctx, hipMemcpy(out, f->vi.stamps.p, (n < cap ? n : cap) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  LSA_HIP(ctx, hipMemsetAsync(f->vi.stamps.p, 0, n * sizeof(unsigned long long), ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

extern "C" int lsa_voxel_filter_pull_timeouts(lsa_voxel_filter *f, uint64_t *out) {
  if (!f || !out) return LSA_ERR_INVALID;
  *out = f->n_pull_timeouts;
  return LSA_OK;
}

extern "C" void lsa_voxel_filter_destroy(lsa_voxel_filter *f) {
  if (!f) return;
  if (getenv("LSA_FILTER_STATS")) fprintf(stderr, "voxel filter: %llu sync calls, %llu speculative misses, %llu capacity growths, W_cap %zu, last W %zu dims %d %d %d\n", f->n_calls, f->n_miss, f->n_cap, f->vi.W_cap, f->vi.W, f->vi.map.dim[0], f->vi.map.dim[1], f->vi.map.dim[2]);
  if (f->stats_on && f->n_pulled) {
    const double k = 1.0 / (double)f->n_pulled;
    fprintf(stderr, "voxel filter: %llu pulled host clouds: enqueue %.1f us, this thread packing %.1f us (%.1f chunks), waiting for the helper %.1f us (withdrawn %llu), waiting for the build %.1f us\n",
            f->n_pulled, f->st_enqueue * k, f->st_pack * k, (double)f->st_main_chunks * k, f->st_join * k, f->st_withdrawn, f->st_wait * k);
  }
  if (f->worker_started) {
    {
      std::lock_guard<std::mutex> lk(f->mu);
      f->job_state.store(-1, std::memory_order_release);
    }
    f->cv.notify_one();
    f->worker.join();
  }
  if (f->packer2_started) {
    {
      std::lock_guard<std::mutex> lk(f->mu2);
      f->p2_state.store(-1, std::memory_order_release);
    }
    f->cv2.notify_one();
    f->packer2.join();
  }
  (void)lsa_enter(f->ctx);
  (void)hipStreamSynchronize(f->ctx->stream);
  if (f->stream2) { (void)hipStreamSynchronize(f->stream2); (void)hipStreamDestroy(f->stream2); (void)hipEventDestroy(f->ev2); }
  f->vi.release(); f->d_in.release(); f->d_out.release(); f->d_pipe.release(); f->h_stage.release(); f->h_ready.release();
  delete f;
}

extern "C" int lsa_voxel_filter_apply_device(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out) {
  if (!f || !n_out || (n && (!d_in_xyzw || !d_out_xyzw))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device: null argument") : LSA_ERR_INVALID;
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  // filter_device returns once the output is complete (it waits on the pipeline's completion event); the bitmap
  // clean-up kernel queued behind that event keeps running and is ordered before anything enqueued later
  return filter_device(f, (const float4 *)d_in_xyzw, n, (float4 *)d_out_xyzw, n_out);
}

// internal (front_end.hip): lsa_voxel_filter_apply_device that also leaves a copy of the input rows in d_keep (n rows) — written by
// the build's first kernel from the rows it has in registers anyway.  FrontEnd::Update keeps a key-frame's raw cloud
// (front_end.cpp:375) and reads it after the call has returned; with the copy the CALLER's buffer is not read any more once this
// returns.  The copy is complete whenever the call returns LSA_OK or LSA_ERR_RANGE.
int lsa_voxel_filter_apply_device_keep(lsa_voxel_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out, float *d_keep) {
  if (!f || !n_out || (n && (!d_in_xyzw || !d_out_xyzw || !d_keep))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply_device_keep: null argument") : LSA_ERR_INVALID;
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  return filter_device(f, (const float4 *)d_in_xyzw, n, (float4 *)d_out_xyzw, n_out, nullptr, (float4 *)d_keep);
}

namespace {
// The steady state of a stream of host scans (the previous scan left a grid definition this one fits): nothing is copied by a copy
// command.  filter_pull_begin starts the helper threads packing the caller's rows into the staging buffer at once; the caller then
// enqueues the build (filter_device with the PullSource) — whose first kernel pulls the rows as they are published and leaves them in
// d_rows — and packs too (filter_pull_join).
bool filter_can_pull(lsa_voxel_filter *f, size_t n) {
  return n >= 65536 && !f->pending && !f->no_pull && vi_spec_ready(f->vi, n, 0, f->leaf);
}
int filter_pull_begin(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float4 *d_rows, PullSource *pull) {
  lsa_context *ctx = f->ctx;
  int rc;
  const size_t chunk = (size_t)1 << kPullChunkShift;
  f->u_chunks = (unsigned)((n + chunk - 1) / chunk);
  if ((rc = f->h_ready.reserve(ctx, (size_t)f->u_chunks * kPullFlagStride))) return rc;
  if (++f->pull_seq == 0) f->pull_seq = 1;   // (the flags start zeroed and are never reset: a sequence number per call)
  f->u_pts = in_pts; f->u_stride = in_stride; f->u_w_off = in_w_off; f->u_off = 0; f->u_n = n;
  f->u_next.store(0, std::memory_order_relaxed);
  pull->h_rows = reinterpret_cast<const float4 *>(f->h_stage.p);
  pull->d_copy = d_rows;
  pull->ready = f->h_ready.p;
  pull->seq = f->pull_seq;
  pull->chunk_shift = kPullChunkShift;
  pull->patience_ticks = f->pull_patience;
  filter_post_job(f, 4);
  if (!f->packer2_started) {
    f->packer2 = std::thread(filter_packer2, f);
    f->packer2_started = true;
  }
  {
    std::lock_guard<std::mutex> lk(f->mu2);
    f->p2_state.store(1, std::memory_order_release);
  }
  f->cv2.notify_one();
  f->n_pulled++;
  return LSA_OK;
}
}  // namespace

// internal (front_end.hip): a HOST scan through the filter with BOTH results on the device — the packed rows in d_rows (the front
// end keeps a key-frame's raw cloud) and the filtered cloud in d_out — by the pulled path above.  *pulled = 0: its preconditions do
// not hold (first scan of a stream, a small cloud, LSA_FILTER_PULL=0): nothing was done, the caller uploads and filters the ordinary way.
int lsa_voxel_filter_apply_host_to_device(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, float *d_rows, float *d_out,
                                          size_t *n_out, int *pulled, void (*after_enqueue)(void *), void *after_user) {
  *pulled = 0;
  *n_out = 0;
  if (!f || !in_pts || !d_rows || !d_out) return LSA_ERR_INVALID;
  lsa_context *ctx = f->ctx;
  if (in_stride < 12 || (in_w_off != SIZE_MAX && in_w_off + 4 > in_stride)) return LSA_OK;   // (the ordinary path reports it)
  if (!filter_can_pull(f, n)) return LSA_OK;
  int rc;
  if (n * 4 > f->h_stage.cap) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (growing frees the old staging buffer)
  if ((rc = f->h_stage.reserve(ctx, n * 4))) return rc;
  PullSource pull;
  if ((rc = filter_pull_begin(f, in_pts, n, in_stride, in_w_off, reinterpret_cast<float4 *>(d_rows), &pull))) return rc;
  *pulled = 1;
  return filter_device(f, reinterpret_cast<const float4 *>(d_rows), n, reinterpret_cast<float4 *>(d_out), n_out, &pull, nullptr, after_enqueue, after_user);
}

extern "C" int lsa_voxel_filter_apply(lsa_voxel_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, void *out_pts,
                                      size_t out_cap, size_t out_stride, size_t out_w_off, size_t *n_out) {
  if (!f || !n_out || (n && (!in_pts || !out_pts))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply: null argument") : LSA_ERR_INVALID;
  lsa_context *ctx = f->ctx;
  *n_out = 0;
  if (n == 0) return LSA_OK;
  if (in_stride < 12 || out_stride < 12) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_voxel_filter_apply: stride < 12");
  // validated HERE, once, before any thread starts packing (lsa_upload_xyzw checks again for its other callers)
  if (in_w_off != SIZE_MAX && in_w_off + 4 > in_stride)
    return lsa_fail(ctx, LSA_ERR_INVALID, "point rows of %zu bytes cannot hold xyz + an intensity at offset %zu (pass SIZE_MAX for clouds without one)", in_stride, in_w_off);
  LSA_HIP(ctx, lsa_enter(ctx));
  int rc;
  // growing a buffer frees the old one, which an earlier command might still use: wait first.  (Not otherwise: every earlier call
  // on this handle returned behind its build's completion event, i.e. behind every copy out of / kernel store into the staging
  // buffer, and nothing else touches it — the wait on an idle stream cost 3-5 us per scan.)
  if (n * 4 > f->h_stage.cap || n > f->d_in.cap) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = f->h_stage.reserve(ctx, n * 4))) return rc;
  if ((rc = f->d_in.reserve(ctx, n, 1.25))) return rc;
  // Host cloud -> device: packing 131 k rows of 32 B costs ~65 us on one core and the four 512 KB copies behind it another
  // ~35 us on one copy queue.  Large clouds are therefore split in two: the helper thread packs the second half and sends
  // it down a second copy queue while this thread does the same with the first half (measured: the data is resident
  // after ~85 us instead of ~100: the two packers share the memory system and the runtime's locks).
  bool pulled = false;
  PullSource pull;
  if (filter_can_pull(f, n)) {
    if ((rc = filter_pull_begin(f, in_pts, n, in_stride, in_w_off, f->d_in.p, &pull))) return rc;
    pulled = true;
  } else if (n >= 65536 && !f->pending) {
    if (!f->stream2) {
      LSA_HIP(ctx, hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking));
      LSA_HIP(ctx, hipEventCreateWithFlags(&f->ev2, hipEventDisableTiming));
    }
    const size_t half = (n / 2 + 32767) / 32768 * 32768;   // whole upload chunks in the first half
    f->u_pts = in_pts; f->u_stride = in_stride; f->u_w_off = in_w_off; f->u_off = half; f->u_n = n - half;
    filter_post_job(f, 2);
    rc = lsa_upload_xyzw(ctx, in_pts, half, in_stride, in_w_off, f->h_stage.p, f->d_in.p);
    while (f->job_state.load(std::memory_order_acquire) != 2) {}
    f->job_state.store(0, std::memory_order_release);
    if (rc || f->job_rc) {
      // leave no copy of either half in flight into d_in / out of h_stage behind an error
      (void)hipStreamSynchronize(f->stream2);
      (void)hipStreamSynchronize(ctx->stream);
      if (rc) return rc;
      return lsa_fail(ctx, LSA_ERR_HIP, "lsa_voxel_filter_apply: upload of the second half failed: %s", hipGetErrorString(f->job_hip));
    }
    LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, f->ev2, 0));
  } else if ((rc = lsa_upload_xyzw(ctx, in_pts, n, in_stride, in_w_off, f->h_stage.p, f->d_in.p))) {
    return rc;
  }
  size_t V = 0;
  // The filtered points go STRAIGHT into the pinned staging buffer (device-visible host memory: the finalize kernel's 16-byte
  // stores cross the link as they are issued, ~120 KB for a KITTI scan) instead of into device memory and back through a copy
  // command and a second wait (13 us).  The staging buffer is free by then — its uploads were ordered in front of the kernels —
  // and visible to the host once the build's completion event has been waited for (vi_wait); a build that ends in the range
  // error writes nothing, so the packed input the pass-through below reads is still there.
  // (pulled: the staging buffer is input and output at once here too — an attempt's finalize kernel runs behind its mark kernel,
  // i.e. behind the last read of the packed rows, and a rebuild after a miss reads the rows from d_in, where the pull left them)
  rc = filter_device(f, f->d_in.p, n, reinterpret_cast<float4 *>(f->h_stage.p), &V, pulled ? &pull : nullptr);
  if (rc == LSA_ERR_RANGE) {
    // PCL: warns and sets output = *input_
    if (out_pts != in_pts) {
      if (out_cap < n) return lsa_fail(ctx, LSA_ERR_CAPACITY, "lsa_voxel_filter_apply: out_cap %zu < %zu", out_cap, n);
      for (size_t i = 0; i < n; i++) {
        const float *p = f->h_stage.p + 4 * i;
        char *o = (char *)out_pts + i * out_stride;
        memcpy(o, p, 12);
        if (out_stride >= 16 && out_w_off != 12) ((float *)o)[3] = 1.0f;
        if (out_w_off != SIZE_MAX && out_w_off + 4 <= out_stride) memcpy(o + out_w_off, p + 3, 4);
      }
    }
    *n_out = n;
    return LSA_ERR_RANGE;
  }
  if (rc) return rc;
  if (V > out_cap) return lsa_fail(ctx, LSA_ERR_CAPACITY, "lsa_voxel_filter_apply: out_cap %zu < %zu voxels", out_cap, V);
  for (size_t i = 0; i < V; i++) {
    const float *p = f->h_stage.p + 4 * i;
    char *o = (char *)out_pts + i * out_stride;
    memcpy(o, p, 12);
    if (out_stride >= 16 && out_w_off != 12) ((float *)o)[3] = 1.0f;
    if (out_w_off != SIZE_MAX && out_w_off + 4 <= out_stride) memcpy(o + out_w_off, p + 3, 4);
  }
  *n_out = V;
  return LSA_OK;
}
