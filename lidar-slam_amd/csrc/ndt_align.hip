// ndt_align.hip — one launch per ScanMatch: the Newton iteration of the NDT registration on the device.
//
// Replaces, for the common case, the host loop of gn_driver.cpp (reference: NormalDistributionsTransform::
// computeTransformation, lidar_localization/src/models/registration/ndt_registration_manual/
// NormalDistributionsTransform.cpp:346-384, with computeStepLengthMT :648-756).  Because `interval_converged` starts true
// whenever step_max > step_min (:683) the More-Thuente trial loop is dead for every configuration the reference ships
// (step_size 0.1, trans_eps 0.01): an iteration is exactly ONE full derivative pass at x + clamp(|dx|) d.  At KITTI size
// such a pass is a ~7 us kernel wrapped in ~8 us of launch + first-wave + row-landing latency, four times per scan; here
// the workgroups stay resident and hand their 32-double rows to EACH OTHER:
//
//   pass body (ndt_pass_dev.h, the very code k_ndt_pass runs, same grid, same tiles, same per-workgroup row)
//   -> lanes 0..31 of wave 0 store the row write-through (agent-scope relaxed stores = `sc1`), the wave drains them
//      (s_waitcnt vmcnt(0)), lane 0 stores the workgroup's epoch flag (`sc1`)
//   -> wave 0 of every workgroup polls the G flags with `sc1` loads (one 4-byte word per workgroup), a workgroup barrier
//      follows, then every thread loads its share of the G rows with `sc1` loads   [MI355X_MICROARCH.md, "Valid forms",
//      first row of the hand-off table: one signalling lane per storing workgroup, poll -> barrier -> loads, all sc1]
//   -> the rows are added in the fixed order of k_reduce_partials / ndt_reduce_partials_host, EVERY workgroup computes the
//      same 32 sums, solves the same 6x6 system and builds the same next transform (gn_math.h: the arithmetic the host
//      driver runs, bit for bit) — an all-gather instead of fan-in + broadcast: one hop per iteration, no "last arriver",
//      no release fence (the in-launch "last block reduces" variant with an agent release cost 10-25 us).
//
// Rows are double-buffered by pass parity (a workgroup can be at most one pass ahead of the slowest one, because it needs
// that one's row to go on); flags carry epoch0 + pass + 1 with epoch0 advancing by max_iter + 8 per launch, compared as a signed
// difference, so nothing has to be zeroed between launches.  Every wait is bounded (wall clock): a launch whose workgroups
// are not all resident gives up with a status word and the host falls back to the per-pass path — as it does when the
// 6x6 system needs the SVD's rank rule, when the line search is live (step_max <= step_min), with an all-reduce
// (multi-GPU), for grids beyond one workgroup per CU and when a trace is requested.
#include <chrono>
#include <vector>

#include "gn_math.h"
#include "ndt_pass_dev.h"

namespace {

enum { ALIGN_OK = 0, ALIGN_NEEDS_HOST = 1, ALIGN_TIMEOUT = 2 };

struct AlignArgs {
  double p0[6];                      // pose of pass 0 (matrix_to_pose of the guess)
  double step_max, step_min, trans_eps;
  int max_iter;
  unsigned epoch0;                   // flags of pass k carry epoch0 + k + 1
  unsigned long long key;            // check-word key of the result row
  unsigned long long timeout_ticks;  // per wait, in wall_clock64 ticks (100 MHz)
  unsigned long long *stamps;        // optional (lsa_ndt_align_stamps): workgroup 0 records kStampsPerPass clock values per pass
};
constexpr int kStampsPerPass = 6, kStampPasses = 40;
#define ALIGN_STAMP(k) do { if (A.stamps && blockIdx.x == 0 && tid == 0 && pass < (unsigned)kStampPasses) A.stamps[pass * kStampsPerPass + (k)] = wall_clock64(); } while (0)

// solver state in LDS (doubles), after the workgroup's reduced 32 sums
constexpr int kStP = 0, kStXt = 6, kStAt = 12, kStScore = 13, kStNpts = 14, kStPairs = 15, kStT = 16 /* 16 floats */, kStInts = 24;
enum { kIIter = 0, kIPasses = 1, kIConverged = 2, kIStatus = 3, kIRun = 4 };
constexpr int kAlignExtra = (32 + 32) * 8;   // out32 + state
constexpr int kDepthBatch = 6;               // row loads in flight per thread and round: 4 x kDepthBatch

// a double of lane `src` (wave-uniform lane index) in every lane: two v_readlane_b32
__device__ __forceinline__ double lane_bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcastf(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// The Newton step of gm::newton_step, spread over the lanes of ONE wave (a single lane issues an f64 instruction at the
// price of 64, and the serial version cost 5.5 us per pass): lane i < 6 holds row i of the augmented system [H | -g] in
// seven registers, pivot rows travel by v_readlane, every arithmetic operation on every element is the one
// gm::lu6_solve / gm::newton_step performs (same order, FMA contraction off), so the result has the same bits —
// lsa_gn_math_selftest runs both on the same systems.  Returns a gm::StepAction (wave-uniform); on STEP_PASS lanes 0..5 have
// written x_t, lane 0 the step length.
__device__ __forceinline__ int wave_newton_step(int lane, const double *out32, const double *p_lds, double step_max, double step_min, double *xt_lds,
                                                double *at_lds, bool *nan_norm) {
#pragma clang fp contract(off)
  const int i = lane < 6 ? lane : 5;  // lanes >= 6 shadow row 5; nothing reads them
  double A[7];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int r = i < j ? i : j, c = i < j ? j : i;
    A[j] = out32[7 + r * 6 - (r * (r - 1)) / 2 + (c - r)];  // H upper triangle, row-major
  }
  const double g_i = out32[1 + i];
  A[6] = -g_i;
  double pmax = 0.0, pmin = DBL_MAX, invs[6];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    int r = c;
    double best = lane_bcast(A[c], c);
    best = best < 0 ? -best : best;
#pragma unroll
    for (int k = c + 1; k < 6; k++) {
      double v = lane_bcast(A[c], k);
      v = v < 0 ? -v : v;
      if (v > best) { best = v; r = k; }
    }
    if (!(best > 0.0) || !(best <= DBL_MAX)) ok = false;
    r = __builtin_amdgcn_readfirstlane(r);
    double Pr[7];
#pragma unroll
    for (int j = c; j < 7; j++) {
      Pr[j] = lane_bcast(A[j], r);
      const double Cr = lane_bcast(A[j], c);
      if (lane == r) A[j] = Cr;
      if (lane == c) A[j] = Pr[j];
    }
    pmax = best > pmax ? best : pmax;
    pmin = best < pmin ? best : pmin;
    const double inv = 1.0 / Pr[c];
    invs[c] = inv;
    if (lane > c) {
      const double f = A[c] * inv;
      if (f != 0.0) {
#pragma unroll
        for (int j = c + 1; j < 7; j++) A[j] = A[j] - f * Pr[j];
      }
    }
  }
  if (!ok || !(pmin > 1e-10 * pmax)) return gm::STEP_NEEDS_HOST;
  double v = A[6], x_mine = 0.0, x[6];
#pragma unroll
  for (int k = 5; k >= 0; k--) {
    const double xk = lane_bcast(v, k) * invs[k];
    x[k] = xk;
    if (!(xk >= -DBL_MAX && xk <= DBL_MAX)) ok = false;
    if (lane == k) x_mine = xk;
    v = v - A[k] * xk;  // meaningful in lanes < k: the next unknowns' rows
  }
  if (!ok) return gm::STEP_NEEDS_HOST;
  double nrm2 = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) nrm2 += x[k] * x[k];
  const double nrm = __builtin_sqrt(nrm2);
  *nan_norm = nrm != nrm;
  if (nrm == 0 || nrm != nrm) return gm::STEP_DONE;
  double d_i = x_mine / nrm;
  double d_phi_0 = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) d_phi_0 += lane_bcast(g_i, k) * lane_bcast(d_i, k);
  d_phi_0 = -d_phi_0;
  const double p_i = p_lds[i];
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) {
      if (lane < 6) xt_lds[lane] = p_i;
      if (lane == 0) *at_lds = 0.0;
      return gm::STEP_NO_PASS;
    }
    d_i = d_i * -1;
  }
  double a_t = nrm < step_max ? nrm : step_max;
  a_t = a_t > step_min ? a_t : step_min;
  if (lane < 6) xt_lds[lane] = p_i + d_i * a_t;
  if (lane == 0) *at_lds = a_t;
  return gm::STEP_PASS;
}

// f32 half of the next pass's parameters (one wave): sinf / cosf of the three Euler angles on six lanes, then the matrix
__device__ __forceinline__ bool wave_pose_to_matrix(int lane, const double *xt_lds, float *Tc /* column-major 16 */, float *Tl /* row-major 3x4 */) {
  bool okf = true;
  const int k = lane % 3;
  const float a = (float)xt_lds[3 + k];
  const float r = gm::sincosf_one(a, lane >= 3 && lane < 6, &okf);
  if (__any(!okf && lane < 6)) return false;
  const float s3[3] = {lane_bcastf(r, 0), lane_bcastf(r, 1), lane_bcastf(r, 2)}, c3[3] = {lane_bcastf(r, 3), lane_bcastf(r, 4), lane_bcastf(r, 5)};
  double p[6];
#pragma unroll
  for (int q = 0; q < 6; q++) p[q] = xt_lds[q];
  float T[16];
  gm::pose_to_matrix_sc(p, s3, c3, T);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 16; q++) Tc[q] = T[q];
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int cc = 0; cc < 4; cc++) Tl[rr * 4 + cc] = T[cc * 4 + rr];
  }
  return true;
}

// f64 half (another wave, concurrently): snapped sin / cos of the three angles on three lanes, then the 69 table entries
__device__ __forceinline__ bool wave_angle_tables(int lane, const double *xt_lds, double *jh) {
  bool okd = true;
  double s, c;
  gm::snapped_sincos(xt_lds[3 + (lane % 3)], &s, &c, &okd);
  if (__any(!okd && lane < 3)) return false;
  const double sx = lane_bcast(s, 0), sy = lane_bcast(s, 1), sz = lane_bcast(s, 2), cx = lane_bcast(c, 0), cy = lane_bcast(c, 1), cz = lane_bcast(c, 2);
  if (lane == 0) gm::angle_tables(sx, cx, sy, cy, sz, cz, reinterpret_cast<double (*)[3]>(jh), reinterpret_cast<double (*)[3]>(jh + 24));
  return true;
}

// kIRun values: 0 = leave the loop, 1 = run the next pass, 2 = a step was taken, its pass parameters are still to be built
// Wave 0, after the sums of a pass have been reduced: bookkeeping of the step that led to this pass, then the next step.
__device__ __forceinline__ void align_step_wave(const AlignArgs &A, unsigned pass, int lane, const double *out32, double *st, int *sti) {
  int it = sti[kIIter];
  const int passes = sti[kIPasses] + 1;
  const double pairs = st[kStPairs] + out32[28];
  bool finished = false, conv = false;
  if (pass > 0) {
    // the step that led to this pass is complete: p += delta (:375), convergence test (:379-381), ++iterations
    const double a_t = st[kStAt];
    conv = it > A.max_iter || (it && ((a_t < 0 ? -a_t : a_t) < A.trans_eps));
    it = it + 1;
    finished = conv;
  }
  // all lanes have read what they need from the state before lane 0 rewrites it
  wave_sync();
  if (lane < 6 && pass > 0) st[kStP + lane] = st[kStXt + lane];
  if (lane == 0) {
    sti[kIPasses] = passes;
    st[kStScore] = out32[0];
    st[kStNpts] = out32[29];
    st[kStPairs] = pairs;
  }
  wave_sync();
  int run = 0, status = ALIGN_OK;
  if (!finished) {
    bool nan_norm = false;
    const int act = wave_newton_step(lane, out32, st + kStP, A.step_max, A.step_min, st + kStXt, st + kStAt, &nan_norm);
    if (act == gm::STEP_DONE) { conv = !nan_norm; }                       // :359-363
    else if (act == gm::STEP_NEEDS_HOST) { status = ALIGN_NEEDS_HOST; }
    else if (act == gm::STEP_NO_PASS) {
      // step length 0 without an evaluation (:664): p, g and H stay, so every further iteration repeats it until the
      // iteration-count or step-size exit fires
      for (;;) {
        conv = it > A.max_iter || (it && (0.0 < A.trans_eps));
        it = it + 1;
        if (conv) break;
      }
    } else {
      run = 2;
    }
  }
  if (lane == 0) {
    sti[kIIter] = it;
    sti[kIConverged] = conv ? 1 : 0;
    sti[kIStatus] = status;
    sti[kIRun] = run;
  }
}

template <bool DENSE>
__global__ __launch_bounds__(kBlockD, 1) void k_ndt_align(const float4 *__restrict__ src, size_t n, PassParams P, AlignArgs A,
                                                                     const uint4 *__restrict__ words, const float4 *__restrict__ dense,
                                                                     const double *__restrict__ rec, unsigned long long *rows /* [2][G][32] */,
                                                                     unsigned *flags /* [G] */, double *h_result /* host-mapped [32] */) {
  constexpr int NB = 2;
  __shared__ __align__(16) unsigned char lds[lds_bytes(NB) + kAlignExtra];
  double *jh = reinterpret_cast<double *>(lds + lds_const(NB));
  float *Tl = reinterpret_cast<float *>(lds + lds_tl(NB));
  double *out32 = reinterpret_cast<double *>(lds + lds_bytes(NB));
  double *st = out32 + 32;
  int *sti = reinterpret_cast<int *>(st + kStInts);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const unsigned G = gridDim.x;
  const PassScalars Q = pass_scalars(P);
  if (tid < 69) jh[tid] = (&P.ja[0][0])[tid];
  else if (tid < 69 + 12) Tl[tid - 69] = P.T[tid - 69];
  if (tid == 128) {
    for (int i = 0; i < 6; i++) { st[kStP + i] = A.p0[i]; st[kStXt + i] = A.p0[i]; }
    st[kStAt] = 0.0; st[kStScore] = 0.0; st[kStNpts] = 0.0; st[kStPairs] = 0.0;
    float *Tc = reinterpret_cast<float *>(st + kStT);  // final_transformation_ = guess (Registration.cpp:80)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) Tc[c * 4 + r] = P.T[r * 4 + c];
    Tc[3] = Tc[7] = Tc[11] = 0.0f; Tc[15] = 1.0f;
    sti[kIIter] = 0; sti[kIPasses] = 0; sti[kIConverged] = 0; sti[kIStatus] = ALIGN_OK; sti[kIRun] = 1;
  }
  __syncthreads();
  for (unsigned pass = 0;; pass++) {
    ALIGN_STAMP(0);
    PassPre pre;
    pass_pre_init<DENSE>(pre);
    const double v = ndt_pass_body<0, DENSE>(src, n, Q, Tl, jh, words, dense, rec, lds, gridDim.x, pre);
    ALIGN_STAMP(1);
    // ---- publish this workgroup's row, then wait for everybody's
    const unsigned epoch = A.epoch0 + pass + 1u;
    unsigned long long *buf = rows + (size_t)(pass & 1u) * G * kOutDoubles;
    if (tid < kOutDoubles)
      __hip_atomic_store(buf + (size_t)blockIdx.x * kOutDoubles + tid, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave drains its row before it signals
      if (tid == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ALIGN_STAMP(2);
      const unsigned long long t0 = wall_clock64();
      bool timed_out = false;
      for (;;) {
        bool ok = true;
        for (unsigned b = (unsigned)lane; b < G; b += 64u) {
          unsigned f = __hip_atomic_load(flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((int)(f - epoch) >= 0);
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > A.timeout_ticks) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (tid == 0 && timed_out) { sti[kIStatus] = ALIGN_TIMEOUT; sti[kIRun] = 0; }
    }
    __syncthreads();
    ALIGN_STAMP(3);
    if (sti[kIStatus] != ALIGN_OK) break;
    // ---- fixed-order reduction of the G rows (the order of k_reduce_partials: sm[b % 32] += row b in ascending b, then
    // the tree over the 32 slots with strides 16, 8, 4, 2, 1), computed redundantly by every workgroup
    {
      const int col = tid & 31, rg = tid >> 5;  // this thread owns slots rg, rg + 8, rg + 16, rg + 24 of column col
      double a4[4] = {0.0, 0.0, 0.0, 0.0};
      for (unsigned d0 = 0; d0 * 32u < G; d0 += kDepthBatch) {
        unsigned long long w[kDepthBatch][4];
#pragma unroll
        for (int dd = 0; dd < kDepthBatch; dd++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned b = (d0 + dd) * 32u + (unsigned)(rg + 8 * j);
            w[dd][j] = b < G ? __hip_atomic_load(buf + (size_t)b * kOutDoubles + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
          }
#pragma unroll
        for (int dd = 0; dd < kDepthBatch; dd++)
#pragma unroll
          for (int j = 0; j < 4; j++) a4[j] += __longlong_as_double((long long)w[dd][j]);  // + 0.0 past the last row: no-op
      }
      // strides 16 and 8 stay inside the thread: slot rl + 16 = rg + 8 (j + 2), slot rl + 8 = rg + 8 (j + 1)
      a4[0] += a4[2]; a4[1] += a4[3];
      a4[0] += a4[1];
      double *sm = reinterpret_cast<double *>(lds);  // aliases the tile slices, which the pass body has left
      sm[rg * 32 + col] = a4[0];
      __syncthreads();
      if (tid < 32) {  // strides 4, 2, 1
        double s0 = sm[col] + sm[4 * 32 + col], s1 = sm[32 + col] + sm[5 * 32 + col];
        double s2 = sm[2 * 32 + col] + sm[6 * 32 + col], s3 = sm[3 * 32 + col] + sm[7 * 32 + col];
        s0 += s2; s1 += s3;
        out32[col] = s0 + s1;
      }
      __syncthreads();
    }
    ALIGN_STAMP(4);
    if (tid < 64) align_step_wave(A, pass, lane, out32, st, sti);
    __syncthreads();
    if (sti[kIRun] == 2) {  // the next pass's parameters: the f32 matrix on wave 0, the f64 tables on wave 1, side by side
      bool ok = true;
      if (tid < 64) ok = wave_pose_to_matrix(lane, st + kStXt, reinterpret_cast<float *>(st + kStT), Tl);
      else if (tid < 128) ok = wave_angle_tables(lane, st + kStXt, jh);
      if (!ok && lane == 0) sti[kIStatus] = ALIGN_NEEDS_HOST;  // an angle outside the fast trig range
    }
    ALIGN_STAMP(5);
    __syncthreads();
    if (sti[kIStatus] != ALIGN_OK || sti[kIRun] == 0) break;
  }
  // ---- result row to the host (workgroup 0): T (16 floats in 8 words), p, score, #points, sum of pairs, counters
  if (blockIdx.x == 0 && tid < 32) {
    const unsigned long long *Tw = reinterpret_cast<const unsigned long long *>(st + kStT);
    unsigned long long w = 0ull;
    if (tid < 8) w = Tw[tid];
    else if (tid < 14) w = (unsigned long long)__double_as_longlong(st[kStP + tid - 8]);
    else if (tid == 14) w = (unsigned long long)__double_as_longlong(st[kStScore]);
    else if (tid == 15) w = (unsigned long long)__double_as_longlong(st[kStNpts]);
    else if (tid == 16) w = (unsigned long long)__double_as_longlong(st[kStPairs]);
    else if (tid == 17) w = (unsigned long long)(unsigned)sti[kIIter] | ((unsigned long long)(unsigned)sti[kIPasses] << 32);
    else if (tid == 18) w = (unsigned long long)(unsigned)sti[kIConverged] | ((unsigned long long)(unsigned)sti[kIStatus] << 32);
    lsa_store_row32(h_result, tid, __longlong_as_double((long long)w), A.key);
  }
}

// ---- self-test of gn_math.h: the same pseudo-random inputs through the serial functions on the host and through the
// wave-parallel forms (what k_ndt_align runs) on the device
struct SelfOut {
  uint32_t sf, cf;        // sinf / cosf bits
  uint64_t sd, cd;        // sin / cos (f64) bits
  int32_t act;            // Newton step action
  uint64_t xt[6], at;     // its outputs
  uint32_t T[16];
  uint64_t jh;            // XOR of the 69 angle-derivative table entries of x_t
  uint64_t sq, dv;        // sqrt, division
};
struct SelfIn {
  float af;
  double ad, H[6][6], g[6], p[6], a, b;
};

GM_HD inline uint64_t self_rng(uint64_t &s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return s >> 11;
}
GM_HD inline double self_u(uint64_t &s) { return (double)self_rng(s) * (1.0 / 9007199254740992.0); }  // [0, 1)

GM_HD inline void self_inputs(uint64_t seed, uint64_t i, SelfIn *in) {
#pragma clang fp contract(off)
  uint64_t s = seed ^ (i * 0x9E3779B97F4A7C15ull);
  self_rng(s);
  const double span = (i % 3 == 0) ? 0.4 : (i % 3 == 1 ? 6.6 : 0.002);
  in->af = (float)((self_u(s) - 0.5) * span);
  in->ad = (self_u(s) - 0.5) * (i % 2 ? 7.0 : 0.3);
  // a negative definite H = -(B B^T + I) scaled like an NDT Hessian, a gradient, a pose
  double B[6][6];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) B[r][c] = (self_u(s) - 0.5) * (r < 3 ? 300.0 : 3000.0);
  for (int r = 0; r < 6; r++)
    for (int c = r; c < 6; c++) {
      double v = (r == c) ? 1.0 : 0.0;
      for (int k = 0; k < 6; k++) v += B[r][k] * B[c][k];
      in->H[r][c] = -v; in->H[c][r] = -v;
    }
  if (i % 17 == 0)  // a singular system now and then: the device must ask for the host path exactly when the host would
    for (int c = 0; c < 6; c++) { in->H[5][c] = in->H[4][c]; in->H[c][5] = in->H[c][4]; }
  if (i % 17 == 0) in->H[5][5] = in->H[4][4];
  for (int r = 0; r < 6; r++) in->g[r] = (self_u(s) - 0.5) * (i % 5 == 0 ? 1.0 : 2.0e4);
  in->p[0] = (self_u(s) - 0.5) * 100.0; in->p[1] = (self_u(s) - 0.5) * 100.0; in->p[2] = (self_u(s) - 0.5) * 4.0;
  in->p[3] = (i % 4 == 0) ? -3.14159 + self_u(s) * 1e-3 : (self_u(s) - 0.5) * 0.1;
  in->p[4] = (self_u(s) - 0.5) * 0.1; in->p[5] = (self_u(s) - 0.5) * 6.0;
  in->a = self_u(s) * 1.0e6 + 1.0e-9; in->b = self_u(s) * 3.0 + 1.0e-3;
}

GM_HD inline void self_scalars(const SelfIn &in, SelfOut *o) {
#pragma clang fp contract(off)
  bool ok = true;
  o->sf = __builtin_bit_cast(uint32_t, gm::sincosf_one(in.af, false, &ok));
  o->cf = __builtin_bit_cast(uint32_t, gm::sincosf_one(in.af, true, &ok));
  double sd, cd;
  gm::sincos_d(in.ad, &sd, &cd, &ok);
  o->sd = __builtin_bit_cast(uint64_t, sd);
  o->cd = __builtin_bit_cast(uint64_t, cd);
  o->sq = __builtin_bit_cast(uint64_t, __builtin_sqrt(in.a));
  o->dv = __builtin_bit_cast(uint64_t, in.a / in.b);
}

// host reference: the serial functions
void self_case_host(uint64_t seed, uint64_t i, SelfOut *o) {
  SelfIn in;
  self_inputs(seed, i, &in);
  memset(o, 0, sizeof(*o));
  self_scalars(in, o);
  gm::StepOut so;
  memset(&so, 0, sizeof(so));
  o->act = gm::newton_step(in.p, in.g, in.H, 0.1, 0.005, &so);
  if (o->act == gm::STEP_PASS || o->act == gm::STEP_NO_PASS) {
    for (int k = 0; k < 6; k++) o->xt[k] = __builtin_bit_cast(uint64_t, so.x_t[k]);
    o->at = __builtin_bit_cast(uint64_t, so.a_t);
  }
  if (o->act == gm::STEP_PASS) {
    for (int k = 0; k < 16; k++) o->T[k] = __builtin_bit_cast(uint32_t, so.T[k]);
    double ja[8][3], ha[15][3];
    gm::angle_derivatives(so.x_t, ja, ha);
    uint64_t h = 0;
    for (int k = 0; k < 24; k++) h ^= lsa_row_mix(__builtin_bit_cast(uint64_t, (&ja[0][0])[k]), k % 31);
    for (int k = 0; k < 45; k++) h ^= lsa_row_mix(__builtin_bit_cast(uint64_t, (&ha[0][0])[k]), (k + 7) % 31);
    o->jh = h;
  }
}

// device: one case per 64-lane workgroup, through the wave-parallel forms
__global__ __launch_bounds__(64) void k_gn_selftest(uint64_t seed, size_t n, SelfOut *out) {
  __shared__ double out32[32], st[32], jh[72];
  __shared__ float Tc[16], Tl[12];
  const int lane = (int)threadIdx.x;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  SelfIn in;
  self_inputs(seed, i, &in);
  if (lane == 0) {
    for (int k = 0; k < 32; k++) out32[k] = 0.0;
    for (int k = 0; k < 6; k++) { out32[1 + k] = in.g[k]; st[k] = in.p[k]; st[6 + k] = 0.0; }
    int k = 7;
    for (int r = 0; r < 6; r++)
      for (int c = r; c < 6; c++) out32[k++] = in.H[r][c];
    st[12] = 0.0;
    for (int q = 0; q < 72; q++) jh[q] = 0.0;
    for (int q = 0; q < 16; q++) Tc[q] = 0.0f;
  }
  __syncthreads();
  bool nan_norm = false;
  const int act = wave_newton_step(lane, out32, st, 0.1, 0.005, st + 6, st + 12, &nan_norm);
  __syncthreads();
  if (act == gm::STEP_PASS) {
    wave_pose_to_matrix(lane, st + 6, Tc, Tl);
    wave_angle_tables(lane, st + 6, jh);
  }
  __syncthreads();
  if (lane == 0) {
    SelfOut o;
    memset(&o, 0, sizeof(o));
    self_scalars(in, &o);
    o.act = act;
    if (act == gm::STEP_PASS || act == gm::STEP_NO_PASS) {
      for (int k = 0; k < 6; k++) o.xt[k] = __builtin_bit_cast(uint64_t, st[6 + k]);
      o.at = __builtin_bit_cast(uint64_t, st[12]);
    }
    if (act == gm::STEP_PASS) {
      for (int k = 0; k < 16; k++) o.T[k] = __builtin_bit_cast(uint32_t, Tc[k]);
      uint64_t h = 0;
      for (int k = 0; k < 24; k++) h ^= lsa_row_mix(__builtin_bit_cast(uint64_t, jh[k]), k % 31);
      for (int k = 0; k < 45; k++) h ^= lsa_row_mix(__builtin_bit_cast(uint64_t, jh[24 + k]), (k + 7) % 31);
      o.jh = h;
    }
    out[i] = o;
  }
}

}  // namespace

extern "C" int lsa_gn_math_selftest(lsa_context *ctx, size_t n, uint64_t seed, uint64_t mismatches[5]) {
  if (!ctx || !mismatches || n == 0 || n > (1u << 20)) return ctx ? lsa_fail(ctx, LSA_ERR_INVALID, "lsa_gn_math_selftest: bad argument") : LSA_ERR_INVALID;
  LSA_HIP(ctx, lsa_enter(ctx));
  SelfOut *d = nullptr;
  LSA_HIP(ctx, hipMalloc((void **)&d, n * sizeof(SelfOut)));
  hipLaunchKernelGGL(k_gn_selftest, dim3((unsigned)n), dim3(64), 0, ctx->stream, seed, n, d);
  std::vector<SelfOut> h(n);
  hipError_t e = hipMemcpyAsync(h.data(), d, n * sizeof(SelfOut), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return lsa_fail(ctx, LSA_ERR_HIP, "lsa_gn_math_selftest: %s", hipGetErrorString(e));
  for (int k = 0; k < 5; k++) mismatches[k] = 0;
  for (size_t i = 0; i < n; i++) {
    SelfOut r;
    self_case_host(seed, i, &r);
    const SelfOut &g = h[i];
    mismatches[0] += (r.sf != g.sf) + (r.cf != g.cf);
    mismatches[1] += (r.sd != g.sd) + (r.cd != g.cd);
    bool same = r.act == g.act && r.at == g.at && r.jh == g.jh;
    for (int k = 0; k < 6; k++) same = same && r.xt[k] == g.xt[k];
    for (int k = 0; k < 16; k++) same = same && r.T[k] == g.T[k];
    mismatches[2] += !same;
    mismatches[3] += (r.sq != g.sq) + (r.dv != g.dv);
    mismatches[4] += (r.act == gm::STEP_NEEDS_HOST);  // how many cases took the singular branch (not a mismatch count)
  }
  return LSA_OK;
}

// diagnostics: clock values (100 MHz) workgroup 0 took during the last persistent ScanMatch, kStampsPerPass per pass:
// pass body start / end, row + flag published, all flags seen, rows reduced, Newton step done
extern "C" int lsa_ndt_align_stamps(lsa_ndt *h, int enable, uint64_t *out, size_t cap, size_t *n_out) {
  if (!h) return LSA_ERR_INVALID;
  lsa_context *ctx = h->ctx;
  h->align_stamps_on = enable != 0;
  if (n_out) *n_out = 0;
  if (!out || !h->align_stamps.p) return LSA_OK;
  LSA_HIP(ctx, lsa_enter(ctx));   // (the copy below goes onto the stream: release anything parked on it first)
  size_t n = (size_t)kStampsPerPass * kStampPasses;
  if (n > cap) n = cap;
  LSA_HIP(ctx, hipMemcpyAsync(out, h->align_stamps.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_out) *n_out = n;
  return LSA_OK;
}

// Whether this ScanMatch can run as one persistent launch (see the header comment for the cases that cannot).
bool ndt_align_persistent_ok(const lsa_ndt *h) {
  if (h->no_persistent || h->allreduce || h->trace_rows || h->n_source == 0) return false;
  if (!((double)h->step_size > (double)h->trans_eps / 2)) return false;  // the trial loop would be live
  if (h->max_iter < 0 || h->max_iter > 100000) return false;
  // one workgroup per CU at most: every workgroup of the launch must be resident for the in-launch hand-off
  return ndt_pass_grid(h, h->n_source, 0) <= (unsigned)h->ctx->num_cus;
}

// Returns LSA_OK with *done = true when the device-side loop delivered the result; *done = false (and LSA_OK) when the
// caller has to run the per-pass driver instead.
int ndt_align_persistent(lsa_ndt *h, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats, bool *done) {
  lsa_context *ctx = h->ctx;
  *done = false;
  const unsigned G = ndt_pass_grid(h, h->n_source, 0);
  int rc;
  if (h->align_rows.cap < (size_t)2 * G * kOutDoubles || !h->align_flags.p) {
    // the previous launch's result row comes from workgroup 0; others may still be leaving: drain before the buffers move
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = h->align_rows.reserve(ctx, (size_t)2 * G * kOutDoubles, 1.5))) return rc;
    if ((rc = h->align_flags.reserve(ctx, (size_t)ctx->num_cus))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(h->align_flags.p, 0, (size_t)ctx->num_cus * sizeof(unsigned), ctx->stream));
    h->align_seq = 64;
  }
  if ((rc = h->h_result.reserve(ctx, kOutDoubles))) return rc;
  AlignArgs A;
  float T0[16];
  memcpy(T0, predict_pose, sizeof(T0));
  lsa_matrix_to_pose(T0, A.p0);
  A.step_max = (double)h->step_size;           // float -> double like the plugin (ndt_registration_manual.cpp:14-22)
  A.step_min = (double)h->trans_eps / 2;
  A.trans_eps = (double)h->trans_eps;
  A.max_iter = h->max_iter;
  // this launch uses epochs epoch0 + 1 .. epoch0 + max_iter + 2; the next one starts beyond them
  A.epoch0 = h->align_seq;
  h->align_seq += (unsigned)h->max_iter + 8u;
  unsigned long long key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  if (!key) key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  A.key = key;
  A.timeout_ticks = 2000000ull;                // 20 ms per wait
  A.stamps = nullptr;
  if (h->align_stamps_on) {
    if ((rc = h->align_stamps.reserve(ctx, (size_t)kStampsPerPass * kStampPasses))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(h->align_stamps.p, 0, (size_t)kStampsPerPass * kStampPasses * sizeof(unsigned long long), ctx->stream));
    A.stamps = h->align_stamps.p;
  }
  PassParams P;
  ndt_make_pass_params(h, T0, A.p0, &P);
  unsigned long long *rows = reinterpret_cast<unsigned long long *>(h->align_rows.p);
  if (h->use_dense)
    hipLaunchKernelGGL((k_ndt_align<true>), dim3(G), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, A, h->words.p, ndt_dense(h), ndt_rec(h), rows,
                       h->align_flags.p, h->h_result.p);
  else
    hipLaunchKernelGGL((k_ndt_align<false>), dim3(G), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, A, h->words.p, ndt_dense(h), ndt_rec(h), rows,
                       h->align_flags.p, h->h_result.p);
  LSA_HIP(ctx, hipGetLastError());
  unsigned long long w[32];
  double r[32];
  bool ok = !h->no_row_polling && ndt_fetch_row_host(h->h_result.p, key, 200.0, w);
  if (!ok) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
    if (!ndt_fetch_row_host(h->h_result.p, key, 0.0, w)) return lsa_fail(ctx, LSA_ERR_HIP, "NDT align: result row incomplete after the kernel finished");
  } else if (++h->unsynced >= 256) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
  }
  memcpy(r, w, sizeof(r));
  const int status = (int)(w[18] >> 32);
  if (status != ALIGN_OK) {
    // the launch gave up (SVD needed, or its workgroups were not all resident in time): let it drain, then the caller
    // runs the per-pass driver from the start
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
    h->align_fallbacks++;
    return LSA_OK;
  }
  memcpy(result_pose, &w[0], 16 * sizeof(float));
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    const double score = r[14], npts = r[15], pairs = r[16];
    stats->iterations = (int32_t)(w[17] & 0xffffffffu);
    stats->passes = (int32_t)(w[17] >> 32);
    stats->ls_iterations = 0;
    stats->converged = (int32_t)(w[18] & 0xffffffffu);
    stats->score = score;
    stats->trans_probability = npts > 0 ? score / npts : 0;
    stats->pairs_per_pass = stats->passes ? pairs / stats->passes : 0;
    memcpy(stats->final_p, &r[8], 6 * sizeof(double));
    stats->gpu_ms = 0;
  }
  h->align_launches++;
  *done = true;
  return LSA_OK;
}
