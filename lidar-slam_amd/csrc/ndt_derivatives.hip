// ndt_derivatives.hip — the hot loop: one NDT derivative pass over the source cloud.
//
// Replaces NormalDistributionsTransform::computeDerivatives / computeHessian and what they call per point
// (reference: lidar_localization/src/models/registration/ndt_registration_manual/NormalDistributionsTransform.cpp
//  :391-445 computeDerivatives, :448-482 computePointDerivatives, :485-520 updateDerivatives, :875-899 updateHessian,
//  :901-936 computeHessian; VoxelGrid.cpp:432-480 radiusSearch; pcl::transformPointCloud, f32).
//
// Per 14-point tile, owned by ONE wave: f32 rigid transform (bit-identical to PCL's, no FMA contraction), row probes of the
// <= 3x3x3 cell neighbourhood (dense per-cell table, or the {occupancy, valid, prefix} bitmap words for huge cell boxes),
// scan-based compaction of the hits into an LDS pair queue, then dense f64 pair math on the gathered 80-byte voxel
// records exploiting the symmetry of C^-1 (21 unique Hessian entries + 6 gradient + score = 28 values), LDS transpose
// reduction -> one 32-double partial per workgroup, then a fixed-order second-stage reduction (no atomics: bitwise
// reproducible run to run).  No MFMA: the largest contraction is 6x6.
//
// Algorithmic bytes per pass (SURVEY.md 8d formula C, with this implementation's sizes):
//   16*M (points) + 27*16*M (dense probe entries, worst case) + 80*Kbar*M (records); the probe table and the records are
//   L2 / Infinity-Cache resident at every BASELINE.json size (PMC: HBM traffic = the source cloud), so the pass is
//   latency bound at KITTI size and VALU-issue bound at 2 M points, not HBM bound.
#include <chrono>
#include <cstring>
#include <immintrin.h>
#include "gn_math.h"
#include "ndt_pass_dev.h"

namespace {

template <int KIND, bool DENSE>
__global__ __launch_bounds__(kBlockD, pass_waves(KIND)) void k_ndt_pass(const float4 *__restrict__ src, size_t n, PassParams P, const uint4 *__restrict__ words,
                                                      const float4 *__restrict__ dense, const double *__restrict__ rec, double *__restrict__ partials,
                                                      unsigned long long key) {
  constexpr int NB = pass_bufs(KIND);
  __shared__ __align__(16) unsigned char lds[lds_bytes(NB)];
  double *jh = reinterpret_cast<double *>(lds + lds_const(NB));
  float *Tl = reinterpret_cast<float *>(lds + lds_tl(NB));
  if (threadIdx.x < 69) jh[threadIdx.x] = (&P.ja[0][0])[threadIdx.x];
  else if (threadIdx.x < 69 + 12) Tl[threadIdx.x - 69] = P.T[threadIdx.x - 69];
  __syncthreads();
  const PassScalars Q = pass_scalars(P);
  PassPre pre;
  pass_pre_init<DENSE>(pre);
  const double v = ndt_pass_body<KIND, DENSE>(src, n, Q, Tl, jh, words, dense, rec, lds, gridDim.x, pre);
  // one 256-byte row per workgroup, written once (check-worded when it goes straight to the host)
  if (threadIdx.x < kOutDoubles) lsa_store_row32(partials + (size_t)blockIdx.x * kOutDoubles, (int)threadIdx.x, v, key);
}

}  // namespace

constexpr unsigned kHostReduceMaxBlocks = 1024;
constexpr size_t kHostRowsDoubles = (size_t)2 * kHostReduceMaxBlocks * kOutDoubles;   // two halves (ndt_armed_rows)

// ---- armed pass: the NEXT pass's kernel is launched before its parameters exist -------------------------------------
// A pass kernel at KITTI size is ~5 us of work behind ~3 us of host launch call and ~4.5 us until its first wave runs, and
// the host can only launch it once the previous pass's sums have produced the next transform.  With the line search dead
// (one full pass per Newton iteration, see ndt_align.hip) the kind of the next kernel is known in advance, so it is
// enqueued right behind the current one and its workgroups wait — resident, for the ~3 us the host needs to take the
// rows, solve and post — for the dynamic part of the parameters (rigid transform + the 69 angle-derivative constants,
// 600 B) in a MAILBOX in fine-grained device memory that the host writes through the PCIe BAR (measured: host store ->
// visible to a polling wave -> echo in host memory 1.95 us round trip).  The pollers read device memory, not the link.
// The mail carries the pass's sequence number and a check word over the payload (the host's write-combined stores may
// land in any order; a torn read is simply repeated); a CANCEL mail (converged: the armed kernel is not needed) or the end of
// its patience (0.3 ms by default, lsa_ndt_set_armed_patience) makes the kernel exit without writing its row.  Same body, same grid, same rows: the results are the bits
// of the unarmed path.
struct PassMail {
  unsigned long long words[80];   // [0..5] the 12 floats of T, [6..74] jh (ja then ha), [75] virtual grid, [76] source pointer, [77] #points,
                                  // [78] 1: the kept copy of the first tiles (see k_ndt_pass_armed) is this pass's source
  unsigned long long check;       // key(seq) ^ XOR_c rotl(word_c, c % 31 + 1)
  unsigned long long seq;         // sequence number of the pass these parameters are for; | kMailCancel: that pass is cancelled
};
constexpr unsigned long long kMailCancel = 1ull << 63;
__host__ __device__ inline unsigned long long mail_key(unsigned long long seq) { return (seq + 1) * 0x9E3779B97F4A7C15ull; }

namespace {
template <bool DENSE>
__global__ __launch_bounds__(kBlockD, pass_waves(0)) void k_ndt_pass_armed(PassParams P, const uint4 *__restrict__ words, const float4 *__restrict__ dense,
                                                                         const double *__restrict__ rec, double *__restrict__ partials, const PassMail *mail,
                                                                         unsigned long long seq, unsigned long long patience_ticks,
                                                                         unsigned long long *gave_up /* host-mapped */, float4 *__restrict__ keep,
                                                                         unsigned keep_cap) {
  constexpr int NB = 2;
  __shared__ __align__(16) unsigned char lds[lds_bytes(NB)];
  __shared__ int s_go;
  __shared__ unsigned long long s_dyn[4];   // virtual grid, source pointer, #points, "source = the kept copy"
  double *jh = reinterpret_cast<double *>(lds + lds_const(NB));
  float *Tl = reinterpret_cast<float *>(lds + lds_tl(NB));
  const int tid = (int)threadIdx.x, lane = tid & 63;
  PassScalars Q = pass_scalars(P);   // every kernel argument the pass needs, fetched now, while there is nothing else to do
  pass_scalars_pin<DENSE>(Q);
  PassPre pre;   // zeroed accumulators + probe index fields: also ready before the wait
  pass_pre_init<DENSE>(pre);
  pass_pre_pin(pre);
  unsigned long long t_mail = 0, t_start = 0;   // (wave 0) 100 MHz clock: kernel start, mail taken — diagnostics in row column 30
  // Every pass of a ScanMatch reads the same source cloud and a wave's first tile is a function of its position in the grid
  // only, so each pass KEEPS its first tiles' points in memory the handle owns and the next one loads them from there right
  // now, before it knows anything else: that takes the source load (a cold-cache miss at the head of the dependency chain)
  // off the path between the mail's arrival and the rows.  The mail says whether the copy is this pass's source (not for
  // the first pass of a ScanMatch: the caller's cloud is only ever read once the mail names it).
  float4 s_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const size_t i0 = ((size_t)blockIdx.x * kWaves + (size_t)(tid >> 6)) * kTilePts + (size_t)(lane >> 2);
    if ((lane >> 2) < kTilePts && i0 < keep_cap) s_pre = keep[i0];
  }
  if (tid < 64) {  // wave 0: wait for this pass's mail, take the payload, verify it
    const unsigned long long *mw = reinterpret_cast<const unsigned long long *>(mail);
    const unsigned long long t0 = wall_clock64();
    t_start = t0;
    int go = 0;
    for (;;) {
      // ONE round trip per look: sequence number, payload and check word are fetched together and the payload is taken if it
      // is whole (the check word is keyed with the sequence number, so words of two different posts cannot pass) — not
      // "see the number, then fetch the payload", which put a second uncached read between the mail's arrival and the pass
      const unsigned long long sq = __hip_atomic_load(&mail->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long w0 = __hip_atomic_load(mw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long w1 = lane < 16 ? __hip_atomic_load(mw + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
      const unsigned long long chk = __hip_atomic_load(&mail->check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((sq & ~kMailCancel) > seq || sq == (seq | kMailCancel)) break;  // cancelled, or the host has already moved on to a later pass
      if (sq == seq) {
        unsigned long long x = lsa_row_mix(w0, lane % 31) ^ (lane < 16 ? lsa_row_mix(w1, (64 + lane) % 31) : 0ull);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
        if ((x ^ mail_key(seq)) == chk) {
          unsigned long long *dstw = reinterpret_cast<unsigned long long *>(lds + lds_const(NB));   // jh[0..68]
          if (lane >= 6) dstw[lane - 6] = w0;                                       // jh[0..57]
          else reinterpret_cast<unsigned long long *>(Tl)[lane] = w0;               // T: 6 words = 12 floats
          if (lane < 11) dstw[58 + lane] = w1;                                      // jh[58..68]
          else if (lane < 15) s_dyn[lane - 11] = w1;                                // words 75, 76, 77, 78
          go = 1;
          t_mail = wall_clock64();
          break;
        }
      }
      if (wall_clock64() - t0 > patience_ticks) {
        // nobody posted in time: leave, and tell the host so that it does not wait for rows.  EVERY workgroup that leaves says
        // so (they all store the same number): the patience clocks are per workgroup, so a mail that lands inside their skew is
        // taken by some workgroups and not by others — the host must learn of one missing row at once, whichever it is
        if (lane == 0) __hip_atomic_store(gave_up, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) s_go = go;
  }
  __syncthreads();
  if (!s_go) return;
  // workgroup-uniform values out of LDS: into scalar registers (the pass body has no vector register to spare)
  auto uniform64 = [](unsigned long long v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const unsigned vgrid = (unsigned)uniform64(s_dyn[0]);
  if (blockIdx.x >= vgrid) return;   // launched with head-room for a larger source than this pass has
  const float4 *src = reinterpret_cast<const float4 *>(uniform64(s_dyn[1]));
  const size_t n = (size_t)uniform64(s_dyn[2]);
  const bool kept = __builtin_amdgcn_readfirstlane((int)(unsigned)s_dyn[3]) != 0 && n <= (size_t)keep_cap;
  const double v = ndt_pass_body<0, DENSE>(src, n, Q, Tl, jh, words, dense, rec, lds, vgrid, pre, kept, s_pre, (!kept && n <= (size_t)keep_cap) ? keep : nullptr);
  // column 30 (unused by the sums' consumers): ticks this workgroup waited for its mail x 2^20 + ticks from the mail to here
  const double stamp = (double)(((t_mail - t_start) << 20) + ((wall_clock64() - t_mail) & 0xfffffull));
  if (tid < kOutDoubles) lsa_store_row32(partials + (size_t)blockIdx.x * kOutDoubles, tid, tid == 30 ? stamp : v, mail_key(seq) | 1ull);
}
}  // namespace

// host: post the dynamic parameters of pass `seq` (or its cancellation) into the mailbox.  The mailbox has TWO slots, taken by
// the parity of the sequence number: the final pass of one ScanMatch may still be on its way when the first mail of the next is
// posted (deferred final pass, ndt_api.hip), and a mail must never be overwritten before every workgroup it is for has taken it.
void ndt_mail_post(PassMail *mail_base, const PassParams *P, unsigned long long seq, bool cancel, const void *d_src, size_t n, unsigned vgrid, bool kept) {
  PassMail *mail = mail_base + (seq & 1ull);
  if (cancel) {
    __atomic_store_n(&mail->seq, seq | kMailCancel, __ATOMIC_RELEASE);
    _mm_sfence();
    return;
  }
  unsigned long long w[80];
  memset(w, 0, sizeof(w));
  memcpy(&w[0], P->T, 12 * sizeof(float));
  memcpy(&w[6], &P->ja[0][0], 24 * sizeof(double));
  memcpy(&w[30], &P->ha[0][0], 45 * sizeof(double));
  w[75] = vgrid;
  w[76] = (unsigned long long)(uintptr_t)d_src;
  w[77] = (unsigned long long)n;
  w[78] = kept ? 1ull : 0ull;
  unsigned long long x = 0;
  for (int c = 0; c < 80; c++) x ^= lsa_row_mix(w[c], c % 31);
  for (int c = 0; c < 80; c++) mail->words[c] = w[c];
  mail->check = x ^ mail_key(seq);
  _mm_sfence();   // payload before the sequence number (the BAR mapping may be write-combining)
  __atomic_store_n(&mail->seq, seq, __ATOMIC_RELEASE);
  _mm_sfence();
}

// enqueue the kernel of pass `seq` with `grid` workgroups (>= the grid the pass will turn out to need: the mail names the
// number that share the tiles, the others leave); it waits at most patience_us for its mail
// the host-mapped row buffer, the give-up word and the kept first-tile points of the armed passes (sized once, never moved)
int ndt_reserve_armed(lsa_ndt *h) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->h_partials.reserve(ctx, kHostRowsDoubles))) return rc;
  if (!h->h_gave_up.p) {
    if ((rc = h->h_gave_up.reserve(ctx, 8))) return rc;
    memset(h->h_gave_up.p, 0, 8 * sizeof(double));
  }
  // first-tile points kept from pass to pass (k_ndt_pass_armed): sized once for the largest armed grid, never moved
  if (!h->src_keep.p && (rc = h->src_keep.reserve(ctx, (size_t)kHostReduceMaxBlocks * kWaves * kTilePts))) return rc;
  return h->h_out.reserve(ctx, kOutDoubles);
}

int ndt_launch_armed(lsa_ndt *h, const PassParams &P, unsigned long long seq, unsigned grid, double patience_us) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = ndt_reserve_armed(h))) return rc;
  const PassMail *mail = reinterpret_cast<const PassMail *>(h->mail) + (seq & 1ull);   // the slot of this sequence number's parity
  double *rows = ndt_armed_rows(h, seq);                                                   // ... and its half of the row buffer
  const unsigned long long ticks = (unsigned long long)(patience_us * 100.0);
  unsigned long long *gu = reinterpret_cast<unsigned long long *>(h->h_gave_up.p);
  if (h->use_dense)
    hipLaunchKernelGGL((k_ndt_pass_armed<true>), dim3(grid), dim3(kBlockD), 0, ctx->stream, P, h->words.p, ndt_dense(h), ndt_rec(h), rows, mail, seq, ticks, gu, h->src_keep.p, (unsigned)h->src_keep.cap);
  else
    hipLaunchKernelGGL((k_ndt_pass_armed<false>), dim3(grid), dim3(kBlockD), 0, ctx->stream, P, h->words.p, ndt_dense(h), ndt_rec(h), rows, mail, seq, ticks, gu, h->src_keep.p, (unsigned)h->src_keep.cap);
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

unsigned long long ndt_armed_row_key(unsigned long long seq) { return mail_key(seq) | 1ull; }
// rows of the armed pass `seq`: the half of the host-mapped row buffer that belongs to its parity (two consecutive passes never
// share rows: the deferred final pass of a ScanMatch and the first pass of the next are in flight together)
double *ndt_armed_rows(lsa_ndt *h, unsigned long long seq) { return h->h_partials.p + (size_t)(seq & 1ull) * kHostReduceMaxBlocks * kOutDoubles; }

namespace {

// second stage: out[c] = sum_b partials[b][c], fixed order -> deterministic. 1024 threads = 32 columns x 32 row lanes.
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *__restrict__ partials, int nb, double *__restrict__ out) {
  int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  double v = 0.0;
  for (int b = rl; b < nb; b += 32) v += partials[(size_t)b * kOutDoubles + col];
  __shared__ double sm[32][33];
  sm[rl][col] = v;
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (rl < st) sm[rl][col] += sm[rl + st][col];
    __syncthreads();
  }
  if (rl == 0) out[col] = sm[0][col];
}

}  // namespace

static void ndt_angle_derivatives_libm(const double p[6], double j[8][3], double q[15][3]) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  j[0][0] = -sx * sz + cx * sy * cz; j[0][1] = -sx * cz - cx * sy * sz; j[0][2] = -cx * cy;
  j[1][0] = cx * sz + sx * sy * cz;  j[1][1] = cx * cz - sx * sy * sz;  j[1][2] = -sx * cy;
  j[2][0] = -sy * cz;                j[2][1] = sy * sz;                 j[2][2] = cy;
  j[3][0] = sx * cy * cz;            j[3][1] = -sx * cy * sz;           j[3][2] = sx * sy;
  j[4][0] = -cx * cy * cz;           j[4][1] = cx * cy * sz;            j[4][2] = -cx * sy;
  j[5][0] = -cy * sz;                j[5][1] = -cy * cz;                j[5][2] = 0;
  j[6][0] = cx * cz - sx * sy * sz;  j[6][1] = -cx * sz - sx * sy * cz; j[6][2] = 0;
  j[7][0] = sx * cz + cx * sy * sz;  j[7][1] = cx * sy * cz - sx * sz;  j[7][2] = 0;
  q[0][0] = -cx * sz - sx * sy * cz;  q[0][1] = -cx * cz + sx * sy * sz;  q[0][2] = sx * cy;
  q[1][0] = -sx * sz + cx * sy * cz;  q[1][1] = -cx * sy * sz - sx * cz;  q[1][2] = -cx * cy;
  q[2][0] = cx * cy * cz;             q[2][1] = -cx * cy * sz;            q[2][2] = cx * sy;
  q[3][0] = sx * cy * cz;             q[3][1] = -sx * cy * sz;            q[3][2] = sx * sy;
  q[4][0] = -sx * cz - cx * sy * sz;  q[4][1] = sx * sz - cx * sy * cz;   q[4][2] = 0;
  q[5][0] = cx * cz - sx * sy * sz;   q[5][1] = -sx * sy * cz - cx * sz;  q[5][2] = 0;
  q[6][0] = -cy * cz;                 q[6][1] = cy * sz;                  q[6][2] = sy;
  q[7][0] = -sx * sy * cz;            q[7][1] = sx * sy * sz;             q[7][2] = sx * cy;
  q[8][0] = cx * sy * cz;             q[8][1] = -cx * sy * sz;            q[8][2] = -cx * cy;
  q[9][0] = sy * sz;                  q[9][1] = sy * cz;                  q[9][2] = 0;
  q[10][0] = -sx * cy * sz;           q[10][1] = -sx * cy * cz;           q[10][2] = 0;
  q[11][0] = cx * cy * sz;            q[11][1] = cx * cy * cz;            q[11][2] = 0;
  q[12][0] = -cy * cz;                q[12][1] = cy * sz;                 q[12][2] = 0;
  q[13][0] = -cx * sz - sx * sy * cz; q[13][1] = -cx * cz + sx * sy * sz; q[13][2] = 0;
  q[14][0] = -sx * sz + cx * sy * cz; q[14][1] = -cx * sy * sz - sx * cz; q[14][2] = 0;
}

void ndt_make_pass_params(const lsa_ndt *h, const float T[16], const double p[6], PassParams *P) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) P->T[r * 4 + c] = T[c * 4 + r];
  // computeAngleDerivatives (:523-645), including the |angle| < 10e-5 snap: gn_math.h, the arithmetic the device-side
  // Newton loop runs too (its sin / cos are within 1 ulp of libm's; out of its range fall back to libm)
  if (!gm::angle_derivatives(p, P->ja, P->ha)) ndt_angle_derivatives_libm(p, P->ja, P->ha);
  P->d1 = h->d1; P->d2 = h->d2;
  P->res = h->resolution;
  {
    // dense-table pre-filter: accept when the f32 squared distance <= r^2 + margin.  The margin covers the f32 rounding
    // of the cell corner ((float)(index) * res), of the stored centroid offset and of the subtraction chain: a few ulps
    // of the largest coordinate of the cell box times 2 r, doubled; false positives are re-tested exactly in f64
    double r = (double)h->resolution, cmax = 0.0;
    for (int a = 0; a < 3; a++) {
      cmax = std::fmax(cmax, std::fabs((double)h->vi.map.min_b[a]) * r);
      cmax = std::fmax(cmax, std::fabs((double)(h->vi.map.min_b[a] + h->vi.map.dim[a])) * r);
    }
    // The margin is a function of a coordinate BOUND, not of this target's box, wherever the box lies within +-4096 m: the
    // pre-filter's survivors (false positives included) fill the pair queue, i.e. they decide which lane adds which pair, so
    // two tables with the same voxels but different boxes (a full rebuild's bounding box, the incremental local map's
    // window) only give the same BITS if they pre-filter alike.  (4 mm of slack on a 1 m radius: ~1 % more candidates.)
    cmax = std::fmax(cmax, 4096.0);
    double margin = 16.0 * r * (cmax + 4.0 * r) * 1.1920929e-07 + 1e-6;
    P->r2f_pre = (float)(r * r + margin);
  }
  {
    int e = 0;
    P->res_pow2 = (std::frexp(h->resolution, &e) == 0.5f) ? 1 : 0;  // mantissa 0.5: an exact power of two
    P->inv_res = 1.0f / h->resolution;
  }
  {  // exact squared-distance threshold: the largest double x with sqrt(x) < radius (sqrt is monotone)
    double r = (double)h->resolution, t = r * r;
    while (std::sqrt(t) >= r) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) < r) t = std::nextafter(t, INFINITY);
    P->r2max = t;
  }
  for (int a = 0; a < 3; a++) { P->min_b[a] = h->vi.map.min_b[a]; P->dim[a] = h->vi.map.dim[a]; }
}

unsigned ndt_pass_grid(const lsa_ndt *h, size_t n, int kind) {
  size_t blocks = (n + kWaves * kTilePts - 1) / (kWaves * kTilePts);  // one tile per wave until the chip is full, then grid-stride
  size_t cap = (size_t)h->ctx->num_cus * pass_waves(kind);  // the resident set: every workgroup runs at once
  if (blocks > cap) blocks = cap;
  static const size_t forced = getenv("LSA_PASS_GRID_CAP") ? (size_t)atoi(getenv("LSA_PASS_GRID_CAP")) : 0;   // experiments: fewer, longer-running workgroups
  if (forced && blocks > forced) blocks = forced;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

namespace {
template <bool DENSE>
void launch_pass_t(lsa_ndt *h, const PassParams &P, int kind, unsigned g, const uint4 *words, const double *rec, double *dst, unsigned long long key) {
  hipStream_t st = h->ctx->stream;
  const float4 *dense = ndt_dense(h);
  if (kind == 0) hipLaunchKernelGGL((k_ndt_pass<0, DENSE>), dim3(g), dim3(kBlockD), 0, st, h->d_source, h->n_source, P, words, dense, rec, dst, key);
  else if (kind == 1) hipLaunchKernelGGL((k_ndt_pass<1, DENSE>), dim3(g), dim3(kBlockD), 0, st, h->d_source, h->n_source, P, words, dense, rec, dst, key);
  else hipLaunchKernelGGL((k_ndt_pass<2, DENSE>), dim3(g), dim3(kBlockD), 0, st, h->d_source, h->n_source, P, words, dense, rec, dst, key);
}
void launch_pass(lsa_ndt *h, const PassParams &P, int kind, unsigned g, const uint4 *words, const double *rec, double *dst, unsigned long long key) {
  if (h->use_dense) launch_pass_t<true>(h, P, kind, g, words, rec, dst, key);
  else launch_pass_t<false>(h, P, kind, g, words, rec, dst, key);
}
}  // namespace

namespace {
// the reduced (and, multi-GPU, all-reduced) 32-double block -> host-mapped memory as one check-worded row: the host takes it
// as it lands instead of paying a D2H copy + stream wait per pass (~10 us of a sharded pass)
__global__ __launch_bounds__(64) void k_row_to_host(const double *__restrict__ d_out, double *__restrict__ h_row, unsigned long long key) {
  if (threadIdx.x < kOutDoubles) lsa_store_row32(h_row, (int)threadIdx.x, d_out[threadIdx.x], key);
}
}  // namespace

int ndt_publish_out(lsa_ndt *h, unsigned long long *key_out) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->h_out.reserve(ctx, kOutDoubles))) return rc;
  unsigned long long key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  if (!key) key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  hipLaunchKernelGGL(k_row_to_host, dim3(1), dim3(64), 0, ctx->stream, h->d_out.p, h->h_out.p, key);
  LSA_HIP(ctx, hipGetLastError());
  *key_out = key;
  return LSA_OK;
}

// enqueue one pass: partial kernel + second-stage reduction into h->d_out (device). No sync.
int ndt_enqueue_pass(lsa_ndt *h, const PassParams &P, int kind, bool reduce) {
  lsa_context *ctx = h->ctx;
  unsigned g = ndt_pass_grid(h, h->n_source, kind);
  int rc;
  if ((rc = h->partials.reserve(ctx, (size_t)g * kOutDoubles, 1.0))) return rc;
  if ((rc = h->d_out.reserve(ctx, kOutDoubles))) return rc;
  const uint4 *words = h->words.p;
  const double *rec = ndt_rec(h);
  launch_pass(h, P, kind, g, words, rec, h->partials.p, 0ull);
  if (reduce) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, h->partials.p, (int)g, h->d_out.p);
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

// Single-GPU passes: the workgroup partials are written straight into host-mapped pinned memory and the fixed-order
// second stage runs on the host as the rows arrive, which removes a kernel launch, a copy and the stream wait from every
// pass (cfg2: ~10 us of ~25; cfg4, 768 rows: 44 -> 35 us per pass).  The summation order is the one k_reduce_partials
// uses, so both paths give the same bits.  1024 rows covers every grid ndt_pass_grid() produces (<= 256 CUs x 4).
bool ndt_pass_host_reduce(const lsa_ndt *h, int kind) { return !h->allreduce && ndt_pass_grid(h, h->n_source, kind) <= kHostReduceMaxBlocks; }

int ndt_enqueue_pass_host(lsa_ndt *h, const PassParams &P, int kind, unsigned *grid_out, unsigned long long *key_out) {
  lsa_context *ctx = h->ctx;
  unsigned g = ndt_pass_grid(h, h->n_source, kind);
  int rc;
  if ((rc = h->h_partials.reserve(ctx, kHostRowsDoubles))) return rc;
  const uint4 *words = h->words.p;
  const double *rec = ndt_rec(h);
  double *dst = h->h_partials.p;  // pinned + mapped: the same pointer is valid on the device
  // a fresh non-zero key per pass (odd multiplier: a bijection of the counter, so consecutive keys share no pattern).
  // EVEN: the armed passes write their rows into the same host buffer under ODD keys (ndt_armed_row_key) drawn from a counter
  // of their own, and a row one kind left behind must never pass for the other's (found by the short-patience stress test:
  // passes redone the ordinary way between armed ones took stale armed rows whose key happened to coincide)
  unsigned long long key = ((++h->pass_seq) * 0x9E3779B97F4A7C15ull) & ~1ull;
  if (!key) key = ((++h->pass_seq) * 0x9E3779B97F4A7C15ull) & ~1ull;
  launch_pass(h, P, kind, g, words, rec, dst, key);
  LSA_HIP(ctx, hipGetLastError());
  *grid_out = g;
  *key_out = key;
  return LSA_OK;
}

// Host second stage.  sm[b % 32][col] accumulates rows b, b + 32, ... in ascending b (the order k_reduce_partials uses);
// the rows are read front to back once, because they sit in device-written pinned memory that is cold in the CPU
// caches.  key != 0: the rows are taken as they ARRIVE — each is accepted once its check word matches (the kernel may
// still be running, nothing waits for the stream) — and a row that does not show up within the time limit makes the call
// return false (the caller then waits for the stream the ordinary way and calls again).
// One check-worded row out of host-mapped memory, taken as it ARRIVES: accepted once its check word matches (the kernel
// may still be running, nothing waits for the stream).  false: the row did not show up within limit_ms.
namespace {
thread_local unsigned long long t_gap_max_tsc = 0;
thread_local double t_gap_max_noted_us = 0.0;
thread_local unsigned long long t_row_max_tsc = 0;   // the longest wait for ONE row (first look -> accepted / given up): what limit_ms bounds
thread_local unsigned long long t_last_tsc = 0;      // the last time stamp taken while rows were being taken (0: none since the last reset)
// ticks of the time-stamp counter per microsecond, measured once (200 us against the steady clock)
double tsc_per_us() {
  static const double r = [] {
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long c0 = __builtin_ia32_rdtsc();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200.0) __builtin_ia32_pause();
    const unsigned long long c1 = __builtin_ia32_rdtsc();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return (double)(c1 - c0) / us;
  }();
  return r;
}
}  // namespace

double lsa_host_gap_take_us(bool reset) {
  double us = t_gap_max_tsc ? (double)t_gap_max_tsc / tsc_per_us() : 0.0;
  if (t_gap_max_noted_us > us) us = t_gap_max_noted_us;
  if (reset) { t_gap_max_tsc = 0; t_gap_max_noted_us = 0.0; t_last_tsc = 0; }   // (t_last_tsc: intervals are measured inside one wait for rows, not across calls)
  return us;
}
double lsa_row_wait_take_us(bool reset) {
  const double us = t_row_max_tsc ? (double)t_row_max_tsc / tsc_per_us() : 0.0;
  if (reset) t_row_max_tsc = 0;
  return us;
}
void lsa_host_gap_note_us(double us) {
  if (us > t_gap_max_noted_us) t_gap_max_noted_us = us;
  t_last_tsc = __builtin_ia32_rdtsc();   // (called where a wait for rows begins: the first checkpoint measures from here)
}

namespace {
// the detector's state while ONE caller takes rows: kept on the caller's stack and folded into the thread's maxima once, at the end
// (a thread-local access per row is a call into the dynamic linker's TLS lookup in a shared library: 135 of them per pass)
struct RowClock {
  unsigned long long last = 0;      // the last time stamp taken (0: none yet)
  unsigned long long gap_max = 0;   // longest interval between two stamps
  unsigned long long row_max = 0;   // longest wait for one row
  unsigned fast_rows = 0;
};
void row_clock_fold(const RowClock &c) {
  if (c.gap_max > t_gap_max_tsc) t_gap_max_tsc = c.gap_max;
  if (c.row_max > t_row_max_tsc) t_row_max_tsc = c.row_max;
  t_last_tsc = c.last;
}

// Host-stall detector (ndt_pass.h).  A row that is already there — 130 of a pass's 135, normally — costs NO time-stamp read of its
// own: two per row were measured as +0.3..0.4 us per pass (+3..5 % on the pipeline); such rows are covered by a checkpoint every 16th
// of them (last stamp -> now: pure host time, since none of them was waited for).  A row that has to be waited for is timed look by look.
inline bool fetch_row(const double *row, unsigned long long key, double limit_ms, unsigned long long out32[32], const volatile unsigned long long *abort_word,
                      unsigned long long abort_value, RowClock &c) {
  const volatile unsigned long long *wr = reinterpret_cast<const volatile unsigned long long *>(row);
  {
    unsigned long long w[32], x = 0ull;
    for (int col = 0; col < 32; col++) w[col] = wr[col];
    for (int col = 0; col < 31; col++) x ^= lsa_row_mix(w[col], col);
    if ((x ^ key) == w[31]) {
      memcpy(out32, w, sizeof(w));
      out32[31] = 0ull;
      if ((++c.fast_rows & 15u) == 0u) {
        const unsigned long long now = __builtin_ia32_rdtsc();
        if (c.last && now - c.last > c.gap_max) c.gap_max = now - c.last;
        c.last = now;
      }
      return true;
    }
  }
  std::chrono::steady_clock::time_point t0;
  bool timing = false;
  unsigned spins = 0;
  unsigned long long last = __builtin_ia32_rdtsc();
  if (c.last && last - c.last > c.gap_max) c.gap_max = last - c.last;
  const unsigned long long first = last;
  for (;;) {
    __builtin_ia32_pause();
    unsigned long long w[32], x = 0ull;
    for (int col = 0; col < 32; col++) w[col] = wr[col];
    for (int col = 0; col < 31; col++) x ^= lsa_row_mix(w[col], col);
    const unsigned long long now = __builtin_ia32_rdtsc();   // the longest interval between two looks
    if (now - last > c.gap_max) c.gap_max = now - last;
    last = now;
    const bool match = (x ^ key) == w[31];
    bool out = false;
    if (!match && (++spins & 63u) == 0u) {
      if (abort_word && *abort_word == abort_value) out = true;   // the producer has said that it will not deliver
      else if (!timing) { t0 = std::chrono::steady_clock::now(); timing = true; }
      else if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit_ms) out = true;
    }
    if (match || out) {
      if (now - first > c.row_max) c.row_max = now - first;
      c.last = now;
      if (match) {
        memcpy(out32, w, sizeof(w));
        out32[31] = 0ull;
      }
      return match;
    }
  }
}
}  // namespace

bool ndt_fetch_row_host(const double *row, unsigned long long key, double limit_ms, unsigned long long out32[32], const volatile unsigned long long *abort_word,
                        unsigned long long abort_value) {
  RowClock c;
  c.last = t_last_tsc;
  const bool ok = fetch_row(row, key, limit_ms, out32, abort_word, abort_value, c);
  row_clock_fold(c);
  return ok;
}

bool lsa_rows_fast_available();   // host_pack.cpp
unsigned lsa_rows_take_fast(const double *partials, unsigned b, unsigned nb, unsigned long long key, double (*sm)[32]);

bool ndt_reduce_partials_host(const double *partials, unsigned nb, unsigned long long key, double limit_ms, double out32[32],
                              const volatile unsigned long long *abort_word, unsigned long long abort_value) {
  RowClock clk;
  clk.last = t_last_tsc;
  struct Fold { RowClock &c; ~Fold() { row_clock_fold(c); } } fold{clk};
  double sm[32][32];
  for (int rl = 0; rl < 32; rl++)
    for (int col = 0; col < 32; col++) sm[rl][col] = 0.0;
  const bool fast = key != 0 && lsa_rows_fast_available();
  for (unsigned b = 0; b < nb; b++) {
    if (fast) {   // the rows that are there already, four columns at a time (host_pack.cpp); comes back at the first that is not
      b = lsa_rows_take_fast(partials, b, nb, key, sm);
      const unsigned long long now = __builtin_ia32_rdtsc();   // the stall detector's checkpoint: pure host time since the last stamp
      if (clk.last && now - clk.last > clk.gap_max) clk.gap_max = now - clk.last;
      clk.last = now;
      if (b == nb) break;
    }
    const double *row = partials + (size_t)b * kOutDoubles;
    double v[32];
    if (key) {
      unsigned long long w[32];
      if (!fetch_row(row, key, limit_ms, w, abort_word, abort_value, clk)) return false;  // the limit applies per row: a stalled kernel stalls at its first missing one
      memcpy(v, w, sizeof(v));
    } else {
      memcpy(v, row, sizeof(v));
    }
    double *dst = sm[b & 31];
    for (int col = 0; col < 32; col++) dst[col] += v[col];
  }
  for (int st = 16; st > 0; st >>= 1)
    for (int rl = 0; rl < st; rl++)
      for (int col = 0; col < 32; col++) sm[rl][col] += sm[rl + st][col];
  for (int col = 0; col < 32; col++) out32[col] = sm[0][col];
  return true;
}
