// icp.hip — lsa_icp_*: the RegistrationInterface replacement for ICPRegistrationManual
// (reference: lidar_localization/src/models/registration/icp_registration_manual.cpp:12-179).
//
// Algorithm kept from the reference (calculateTrans, :125-178): for a FIXED number of iterations
//   transform the source with the current f32 pose (:132) -> exact 1-NN in the target (pcl::KdTreeFLANN, squared f32
//   distance) (:146) -> drop the pair if sqdist > max_correspond_distance_ (squared vs un-squared, sic, :147) ->
//   e = p' - q, J = [I3 | -R hat(p)] (:156-159) -> H += J^T J, B -= J^T e (:161-162) -> skip the update if det H == 0
//   (:165) -> delta = H^-1 B in f32 (:169) -> t += delta[0:3], R *= SO3::exp(delta[3:6]) (:171-176).
// The kd-tree becomes the uniform grid of nn_grid.h (cell = sqrt(gate) so the gated neighbour lies within +-1 cell):
// target points are counting-sorted by cell rank once per SetInputTarget, and one lane per source point scans the 27
// neighbouring cells.  J^T J / J^T e are formed from f32 J, e exactly as the
// reference forms them, but ACCUMULATED in f64 with a fixed reduction tree (the reference's running f32 sums depend on
// point order); the 6x6 solve and the pose update stay in f32 on the host.
//
// mode 2 (LSA_ICP_MODE_PCL) is the "ICP" plugin instead (reference: lidar_localization/src/models/registration/
// icp_registration.cpp:20-64 -> pcl::IterativeClosestPoint, PCL 1.7, third-party and absent from the tree: UNPINNED):
//   correspondences = exact 1-NN gated at max_correspond_dis (CorrespondenceEstimation::determineCorrespondences),
//   increment = Umeyama / SVD rigid fit of the pairs (TransformationEstimationSVD), the working cloud is transformed IN
//   PLACE by each increment in f32 as PCL does, final = increment * final, and DefaultConvergenceCriteria stops the loop
//   (max_iter; translation^2 <= trans_eps and cos(angle) >= 1 - trans_eps; relative MSE change < euclid_eps; absolute
//   MSE change < 1e-12).  One kernel per iteration applies the pending increment, finds the neighbour and reduces the 17
//   sums the fit needs (n, sum p, sum q, sum q p^T, sum d^2) in f64; the 3x3 SVD runs on the host.
//   GetFitnessScore = pcl::Registration::getFitnessScore: mean squared distance to the nearest target point.
//
// mode 1 (LSA_ICP_MODE_PLANE) is a BUILD EXTENSION, not reference code: BASELINE.json configs[2] asks for point-to-plane
// ICP while both plugins of the reference are point-to-point.  It keeps the ICPRegistrationManual skeleton (fixed number
// of Gauss-Newton iterations, f32 pose, t += d[0:3], R *= exp(d[3:6]), skipped when det H == 0) with the scalar residual
// r = n . (p' - q) and the row J = n^T [I | -R hat(p)]; pairs are gated at squared distance <= max_correspond_dis^2; the
// normal of a target point is the smallest-eigenvalue eigenvector of the covariance of the target points in its voxel
// (leaf `normal_leaf`, >= 6 points, the NDT grid's machinery), looked up once per SetInputTarget for every cell-sorted
// target point.  Its oracle is oracle/icp_oracle.c orc_icp_plane; parity is definable only between the two.
#include "eig3_dev.h"
#include "nn_grid_dev.h"
#include "ndt_pass.h"
#include "gn_math.h"

#include <immintrin.h>

#include <cmath>
#include <vector>

namespace {
constexpr int kBlock = 256;
constexpr int kOut = 32;
// The neighbour grid's cell is HALF the gate distance and a query may walk kGateRings = 3 rings: a neighbour within the
// gate lies at most ceil(gate / cell) = 2 rings out (+1 ring of slack for the f32 cell rounding), so the gated answer is
// exact, while the search usually stops after the query's own cell and a few face neighbours because the best distance is
// already proven (exact slab bounds, nn_grid_dev.h).  Measured per-iteration kernel time on the cfg2 pair, gate / cell =
// 1, 1.5, 2, 2.5, 3: 57, 45, 43, 54, 67 us (bigger cells: more points per visited cell; smaller: more rings for the
// queries that have no neighbour).
constexpr int kGateRings = 3;
constexpr int kSub = 4;  // lanes per query (nn_nearest_sub): the per-lane chain of dependent cell visits is the bottleneck
constexpr float kCellPerGate = 1.0f / 2.0f;

struct IcpParams {
  float T[12];   // row-major 3x4 current pose
  float gate;    // max_correspond_distance_ compared with the SQUARED distance (sic)
  int use_coarse;  // iterations after the first, when the coarse grid exists: far queries search it
  int allow_skip;  // iterations after the first: a query whose old neighbour is provably still the answer does not search
};

// original index -> slot of the fine grid's cell-sorted array
__global__ __launch_bounds__(kBlock) void k_slot_of_id(const float4 *__restrict__ sorted, size_t n_sorted, unsigned *__restrict__ slot_of_id) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_sorted) slot_of_id[__float_as_uint(sorted[i].w)] = (unsigned)i;
}

// The neighbour search of one ICP iteration for one query (kSub adjacent lanes share it).  A query whose neighbour of the previous
// iteration is farther than a fine cell — or that had none — searches the COARSE grid (one phased step covers the gate); the
// others the fine one.  ONE call site: the grid is selected per lane (a second inlined instance of the ring search next to the 27
// f64 accumulators spills).
//
// SKIPPED searches.  ICPRegistrationManual runs a fixed 30 iterations and the pose stops moving long before the last one; a
// query then asks for the neighbour it already has.  Every search therefore leaves, beside the neighbour, a PROOF RADIUS: the
// query's position q_ref and a lower bound D on the distance from q_ref to every OTHER target point (the runner-up among the
// points it visited, or the radius of the sphere it covered — the search looks kSkipWiden farther than the answer alone needs,
// nn_nearest_sub<.., TRACK>).  One iteration later the query sits at q, m = |q - q_ref| away; every other point is at least D - m
// from it, so when the old neighbour is still closer than that — by a margin 40 times the f32 rounding of the distances — it
// IS the exact nearest neighbour (strictly: no tie can arise) and no cell is looked at.  Likewise a query that had nothing
// within the gate has nothing within it now while D - m stays beyond the gate.  The answer is the one the full search returns;
// `LSA_ICP_NO_SKIP=1` runs every search for the A/B (tests/test_gpu_icp_fitness.py).
constexpr float kSkipWiden = 0.10f;
struct IcpQueryState {
  unsigned prev;   // cell-sorted slot of the neighbour found last (0xffffffff: none)
  float4 ref;      // xyz: the query's position when it last SEARCHED, w: the proof radius D
};
__device__ __forceinline__ NNHit icp_search(const NNView &Gf, const NNView &Gc, const unsigned *__restrict__ slot_of_id, int use_coarse, int allow_skip,
                                            float qx, float qy, float qz, int sub, float gate, IcpQueryState &st, bool *searched) {
  NNHit seed;
  seed.d2 = INFINITY; seed.id = 0xffffffffu; seed.x = seed.y = seed.z = 0.0f; seed.slot = 0u;
  const unsigned seed_slot = st.prev;
  if (seed_slot < Gf.n_sorted) {
    const float4 t = Gf.sorted[seed_slot];
    const float ex = lsa_fsub(qx, t.x), ey = lsa_fsub(qy, t.y), ez = lsa_fsub(qz, t.z);
    seed.d2 = lsa_fadd(lsa_fadd(lsa_fmul(ex, ex), lsa_fmul(ey, ey)), lsa_fmul(ez, ez));
    seed.id = __float_as_uint(t.w); seed.x = t.x; seed.y = t.y; seed.z = t.z; seed.slot = seed_slot;
  }
  *searched = false;
  if (allow_skip) {
    const float mx = qx - st.ref.x, my = qy - st.ref.y, mz = qz - st.ref.z;
    // The old bound carried to the new position.  Every quantity here is a DISTANCE computed from f32 differences, each
    // correctly rounded relative to itself (not to the coordinates' magnitude), so the computed values are off by a few 2^-24 of
    // themselves; the margin is 1e-5 of them (+ 1 um) — 40 times that.  (The bounds that come from cell geometry carry the
    // grid's own 1 mm slack already, nn_grid_dev.h.)
    const float mm = sqrtf(mx * mx + my * my + mz * mz);
    // the very position the last search was run from (the pose has stopped moving in f32): the same query has the same answer,
    // ties included — the one case the proof below cannot make, and the one query in thousands that would search for ever
    if (mx == 0.0f && my == 0.0f && mz == 0.0f) return seed;
    const float lb = st.ref.w - mm - 1.0e-5f * (st.ref.w + mm) - 1.0e-6f;
    if (seed.id != 0xffffffffu && sqrtf(seed.d2) * (1.0f + 1.0e-5f) < lb) return seed;   // still the nearest, and alone at that distance
    if ((seed.id == 0xffffffffu || seed.d2 > gate) && lb > 0.0f && lb * lb > gate * (1.0f + 4.0e-5f)) return seed;   // nothing within the gate (the caller drops the pair)
  }
  *searched = true;
  const bool far = use_coarse && !(seed.d2 <= Gf.cell * Gf.cell);   // (no seed: d2 = inf -> far)
  NNView G = Gf;
  if (far) G = Gc;
  float others = 0.0f;
  NNHit hit = nn_nearest_sub<kSub, true>(G, qx, qy, qz, far ? 2 : kGateRings, sub, gate, 0xffffffffu, &seed, kSkipWiden, &others);
  if (far && hit.id != 0xffffffffu && hit.id != seed.id) hit.slot = slot_of_id[hit.id];
  st.prev = hit.id == 0xffffffffu ? 0xffffffffu : hit.slot;
  st.ref = make_float4(qx, qy, qz, others);
  return hit;
}

// The 32 columns of a workgroup row ([0] unused, [1..6] B, [7..27] H upper triangle, [28] #pairs, [29] #finite source points,
// [30] #queries that ran a search)
// are accumulated by the kSub = 4 lanes that share a query: column c lives in lane (c & 3) of the group, accumulator c >> 2 —
// eight f64 registers per lane instead of 27 + 2 in the group's first lane, and a wave reduction of 4 steps x 8 values instead
// of 6 x 29 (3.3 us at the tail of every pass).  Per column the values meet in the order they met before (the group's own
// running sum, then the groups pairwise at distances 32, 16, 8, 4): same bits.
__device__ __forceinline__ void icp_accumulate(double (&acc)[8], const double (&term)[32], int sub) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const double a = (sub & 1) ? term[4 * j + 1] : term[4 * j], b = (sub & 1) ? term[4 * j + 3] : term[4 * j + 2];
    acc[j] += (sub & 2) ? b : a;
  }
}

// one column's term: the owning lane adds it, the others add +0.0 (which leaves a sum that started at +0.0 as it is)
__device__ __forceinline__ void icp_add_column(double (&acc)[8], int col /* compile-time */, int sub, double v) {
  acc[col >> 2] += (sub == (col & 3)) ? v : 0.0;
}

// wave reduction of the per-lane columns, block sum, check-worded row (tail of k_gn_pass / k_gn_pass_armed)
__device__ __forceinline__ void icp_store_row(double (&acc)[8], double *__restrict__ partials, unsigned long long key) {
  __shared__ double sm[kBlock / 64][kOut];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off >= kSub; off >>= 1) {
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = __shfl_down(acc[j], off, 64);
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] += t[j];
  }
  if (lane < kSub) {
#pragma unroll
    for (int j = 0; j < 8; j++) sm[wave][4 * j + lane] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; w++) v += sm[w][threadIdx.x];
    lsa_store_row32(partials + (size_t)blockIdx.x * kOut, (int)threadIdx.x, v, key);
  }
}

// One Gauss-Newton pass of ICPMANUAL (PLANE = false) or of the point-to-plane extension (PLANE = true) over this workgroup's
// share of the source points: the columns of its row, spread over the lanes as icp_accumulate describes.
template <bool PLANE>
__device__ __forceinline__ void icp_gn_body(const float4 *__restrict__ src, size_t n, const IcpParams &P, const NNView &G, const NNView &Gc,
                                            const unsigned *__restrict__ slot_of_id, unsigned *__restrict__ prev, float4 *__restrict__ qref,
                                            const float4 *__restrict__ slot_normals, double (&acc)[8]) {
  static_assert(kSub == 4, "the column-to-lane mapping of icp_accumulate assumes four lanes per query");
  const int sub = threadIdx.x & (kSub - 1);  // kSub adjacent lanes share one source point's neighbour search
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / kSub; i < n; i += (size_t)gridDim.x * blockDim.x / kSub) {
    float4 s = src[i];
    if (!lsa_finite3(s.x, s.y, s.z)) continue;  // pcl::isFinite(ori_point) (:141)
    if (sub == 1) acc[7] += 1.0;                // column 29: finite source points
    float qx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[0], s.x), lsa_fmul(P.T[1], s.y)), lsa_fmul(P.T[2], s.z)), P.T[3]);
    float qy = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[4], s.x), lsa_fmul(P.T[5], s.y)), lsa_fmul(P.T[6], s.z)), P.T[7]);
    float qz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[8], s.x), lsa_fmul(P.T[9], s.y)), lsa_fmul(P.T[10], s.z)), P.T[11]);
    if (!lsa_finite3(qx, qy, qz)) continue;
    IcpQueryState st;
    st.prev = prev[i];
    st.ref = P.allow_skip ? qref[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // (plane) the old neighbour's normal travels with the old neighbour: one round trip instead of two when the search is skipped
    const unsigned seed_slot = st.prev;
    float4 nv_seed = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PLANE && seed_slot < G.n_sorted) nv_seed = slot_normals[seed_slot];
    bool searched;
    NNHit hit = icp_search(G, Gc, slot_of_id, P.use_coarse, P.allow_skip, qx, qy, qz, sub, P.gate, st, &searched);
    if (sub == 0 && searched) { prev[i] = st.prev; qref[i] = st.ref; }  // next iteration's seed and proof (all kSub lanes have read theirs)
    if (sub == 2 && searched) acc[7] += 1.0;    // column 30: queries that searched (lsa_icp_search_counts)
    // manual: sqdist > max_correspond_distance_ (:147); plane: P.gate = max_correspond_dis^2.  The same decision in all kSub lanes.
    if (hit.id == 0xffffffffu || hit.d2 > P.gate) continue;
    if (PLANE) {
      float4 nv = nv_seed;
      if (hit.slot != seed_slot) nv = slot_normals[hit.slot];
      if (nv.w == 0.0f) continue;
      float e0 = lsa_fsub(qx, hit.x), e1 = lsa_fsub(qy, hit.y), e2 = lsa_fsub(qz, hit.z);
      float r = lsa_fadd(lsa_fadd(lsa_fmul(nv.x, e0), lsa_fmul(nv.y, e1)), lsa_fmul(nv.z, e2));
      float J[6] = {nv.x, nv.y, nv.z, 0.f, 0.f, 0.f};
      // A = -R * hat(p) in f32 as the manual plugin forms it; J[3 + c] = n . A[:, c]
      float A[3][3];
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        float nr0 = -P.T[rr * 4 + 0], nr1 = -P.T[rr * 4 + 1], nr2 = -P.T[rr * 4 + 2];
        // row rr of (-R) times hat(p) = [[0,-z,y],[z,0,-x],[-y,x,0]], summed left to right like the oracle
        A[rr][0] = lsa_fadd(lsa_fadd(lsa_fmul(nr0, 0.0f), lsa_fmul(nr1, s.z)), lsa_fmul(nr2, -s.y));
        A[rr][1] = lsa_fadd(lsa_fadd(lsa_fmul(nr0, -s.z), lsa_fmul(nr1, 0.0f)), lsa_fmul(nr2, s.x));
        A[rr][2] = lsa_fadd(lsa_fadd(lsa_fmul(nr0, s.y), lsa_fmul(nr1, -s.x)), lsa_fmul(nr2, 0.0f));
      }
#pragma unroll
      for (int c = 0; c < 3; c++) J[3 + c] = lsa_fadd(lsa_fadd(lsa_fmul(nv.x, A[0][c]), lsa_fmul(nv.y, A[1][c])), lsa_fmul(nv.z, A[2][c]));
      // out32: [1..6] B = -sum J r, [7..27] H = sum J^T J upper triangle
      const double Jd[6] = {J[0], J[1], J[2], J[3], J[4], J[5]}, rd = r;
#pragma unroll
      for (int a = 0; a < 6; a++) {
        icp_add_column(acc, 1 + a, sub, -(Jd[a] * rd));
#pragma unroll
        for (int b = a; b < 6; b++) icp_add_column(acc, 7 + a * 6 - (a * (a - 1)) / 2 + (b - a), sub, Jd[a] * Jd[b]);
      }
      icp_add_column(acc, 28, sub, 1.0);  // a pair
    } else {
      float e0 = lsa_fsub(qx, hit.x), e1 = lsa_fsub(qy, hit.y), e2 = lsa_fsub(qz, hit.z);
      // J = [I | A], A = -R * hat(p) in f32 (:156-159); hat(p) = [[0,-z,y],[z,0,-x],[-y,x,0]]
      float A[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        float nr0 = -P.T[r * 4 + 0], nr1 = -P.T[r * 4 + 1], nr2 = -P.T[r * 4 + 2];
        A[r][0] = lsa_fadd(lsa_fmul(nr1, s.z), lsa_fmul(nr2, -s.y));
        A[r][1] = lsa_fadd(lsa_fmul(nr0, -s.z), lsa_fmul(nr2, s.x));
        A[r][2] = lsa_fadd(lsa_fmul(nr0, s.y), lsa_fmul(nr1, -s.x));
      }
      double ed[3] = {e0, e1, e2};
      double term[32];
      term[0] = 0.0;
      // B = -J^T e
      term[1] = -ed[0]; term[2] = -ed[1]; term[3] = -ed[2];
#pragma unroll
      for (int a = 0; a < 3; a++) term[4 + a] = -((double)A[0][a] * ed[0] + (double)A[1][a] * ed[1] + (double)A[2][a] * ed[2]);
      // H = J^T J upper triangle: [I, A; A^T, A^T A]
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) {
          double v;
          if (a < 3 && b < 3) v = (a == b) ? 1.0 : 0.0;
          else if (a < 3) v = (double)A[a][b - 3];
          else v = (double)A[0][a - 3] * A[0][b - 3] + (double)A[1][a - 3] * A[1][b - 3] + (double)A[2][a - 3] * A[2][b - 3];
          term[7 + a * 6 - (a * (a - 1)) / 2 + (b - a)] = v;
        }
      term[28] = 1.0;  // a pair
      term[29] = 0.0; term[30] = 0.0; term[31] = 0.0;
      icp_accumulate(acc, term, sub);
    }
  }
}

template <bool PLANE>
__global__ __launch_bounds__(kBlock) void k_gn_pass(const float4 *__restrict__ src, size_t n, IcpParams P, NNView G, NNView Gc, const unsigned *__restrict__ slot_of_id,
                                                    unsigned *__restrict__ prev, float4 *__restrict__ qref, const float4 *__restrict__ slot_normals,
                                                    double *__restrict__ partials, unsigned long long key) {
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0.0;
  icp_gn_body<PLANE>(src, n, P, G, Gc, slot_of_id, prev, qref, slot_normals, acc);
  icp_store_row(acc, partials, key);
}

// ARMED passes (the mechanism of the NDT passes, ndt_derivatives.hip, in its plainest form: ICPMANUAL runs a FIXED number of
// iterations, so every armed kernel gets its mail and nothing is ever cancelled or left parked).  The kernel of iteration k + 1
// is enqueued while iteration k runs; it starts when k's kernel ends and its first wave waits, resident, for the pose — 12 floats
// + the two flags, posted by the host into fine-grained device memory once it has solved iteration k's normal equations.
// Launch call and launch-to-first-wave latency (~8 us of a ~20 us iteration) leave the critical path.
struct IcpMail {
  unsigned long long words[7];   // [0..5] T (12 floats), [6] bit 0 use_coarse, bit 1 allow_skip
  unsigned long long check;      // key ^ XOR_c rotl-mix(word_c)
  unsigned long long seq;        // the pass these words are for
  unsigned long long pad[7];
};
static_assert(sizeof(IcpMail) == 128, "one mail slot = one 128-byte line");

template <bool PLANE>
__global__ __launch_bounds__(kBlock) void k_gn_pass_armed(const float4 *__restrict__ src, size_t n, float gate, NNView G, NNView Gc,
                                                          const unsigned *__restrict__ slot_of_id, unsigned *__restrict__ prev, float4 *__restrict__ qref,
                                                          const float4 *__restrict__ slot_normals, double *__restrict__ partials, unsigned long long key,
                                                          const IcpMail *mail, unsigned long long seq, unsigned long long patience_ticks,
                                                          unsigned long long *gave_up /* host-mapped */) {
  __shared__ unsigned long long s_mail[8];
  __shared__ int s_go;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long *mw = reinterpret_cast<const unsigned long long *>(mail);
    const unsigned long long t0 = wall_clock64();
    int go = 0;
    for (;;) {
      // one round trip per look: words, check word and sequence number together; the payload is taken if it is whole
      const unsigned long long w = lane < 9 ? __hip_atomic_load(mw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
      unsigned long long x = lane < 7 ? lsa_row_mix(w, lane) : 0ull;
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
      const unsigned long long chk = __shfl(w, 7, 64), sq = __shfl(w, 8, 64), x0 = __shfl(x, 0, 64);
      if (sq == seq && (x0 ^ key) == chk) {
        if (lane < 7) s_mail[lane] = w;
        go = 1;
        break;
      }
      if (wall_clock64() - t0 > patience_ticks) {   // nobody posted: leave and say so (the host then runs the pass the ordinary way)
        if (lane == 0) __hip_atomic_store(gave_up, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) s_go = go;
  }
  __syncthreads();
  if (!s_go) return;
  IcpParams P;
  {
    const float *tf = reinterpret_cast<const float *>(s_mail);
#pragma unroll
    for (int k = 0; k < 12; k++) P.T[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tf[k])));
    const int flags = __builtin_amdgcn_readfirstlane((int)(unsigned)s_mail[6]);
    P.use_coarse = flags & 1;
    P.allow_skip = (flags >> 1) & 1;
    P.gate = gate;
  }
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0.0;
  icp_gn_body<PLANE>(src, n, P, G, Gc, slot_of_id, prev, qref, slot_normals, acc);
  icp_store_row(acc, partials, key);
}

// ---- point-to-plane extension --------------------------------------------------------------------------------------
// per occupied voxel of the normal grid: population covariance from the corner-relative fixed-point sums -> Jacobi ->
// eigenvector of the smallest eigenvalue; w = 1 marks a usable normal (>= 6 points)
__global__ __launch_bounds__(kBlock) void k_plane_normals(const long long *__restrict__ acc, const int32_t *__restrict__ cnt, size_t nocc, FixScales fs,
                                                          float4 *__restrict__ normals) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nocc) return;
  const long long *a = acc + r * vi_row_stride(10);   // ten sums and, behind them, the count (voxel_index.h)
  int ni = (int)a[10];
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ni >= 6) {
    double n = (double)ni;
    double md[3] = {((double)a[0] * fs.inv_s1) / n, ((double)a[1] * fs.inv_s1) / n, ((double)a[2] * fs.inv_s1) / n};
    double c[6];
    c[0] = (double)a[3] * fs.inv_s2 / n - md[0] * md[0];
    c[1] = (double)a[4] * fs.inv_s2 / n - md[0] * md[1];
    c[2] = (double)a[5] * fs.inv_s2 / n - md[0] * md[2];
    c[3] = (double)a[6] * fs.inv_s2 / n - md[1] * md[1];
    c[4] = (double)a[7] * fs.inv_s2 / n - md[1] * md[2];
    c[5] = (double)a[8] * fs.inv_s2 / n - md[2] * md[2];
    double ev[3], V[3][3];
    eig3_jacobi(c, ev, V);
    out = make_float4((float)V[0][0], (float)V[1][0], (float)V[2][0], 1.0f);
  }
  normals[r] = out;
}

// one normal per cell-sorted target point: the voxel of the point in the normal grid -> rank -> normal
__global__ __launch_bounds__(kBlock) void k_slot_normals(const float4 *__restrict__ sorted, size_t n, CellMap nm, const uint32_t *__restrict__ nbits,
                                                         const uint32_t *__restrict__ nprefix, const float4 *__restrict__ normals, float4 *__restrict__ slot_normals) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = sorted[i];
  int ijk[3];
  lsa_cell_ijk(nm, p.x, p.y, p.z, ijk);
  int cell = lsa_cell_linear(nm, ijk);
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cell >= 0) {
    uint32_t w = nbits[cell >> 5], bit = 1u << (cell & 31);
    if (w & bit) out = normals[nprefix[cell >> 5] + __popc(w & (bit - 1u))];
  }
  slot_normals[i] = out;
}

struct T12 { float v[12]; };

// PCL-mode iteration: work[i] = T * in[i] (the pending increment, f32 like IterativeClosestPoint::transformCloud), then
// the gated nearest neighbour and the sums of the rigid fit.
// out32: [0] sum d2, [1..3] sum p (source side), [4..6] sum q (target side), [7..15] sum q_r * p_c, [28] #pairs, [29] #finite
__global__ __launch_bounds__(kBlock) void k_pcl_pass(const float4 *__restrict__ in, float4 *__restrict__ work, size_t n, T12 T, double gate2, NNView G,
                                                     unsigned *__restrict__ prev, double *__restrict__ partials, unsigned long long key) {
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) acc[k] = 0.0;
  unsigned ncorr = 0, npts = 0;
  const int sub = threadIdx.x & (kSub - 1);  // kSub adjacent lanes share one source point's neighbour search
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / kSub; i < n; i += (size_t)gridDim.x * blockDim.x / kSub) {
    float4 s = in[i];
    if (!lsa_finite3(s.x, s.y, s.z)) { if (sub == 0) work[i] = s; continue; }
    float px = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[0], s.x), lsa_fmul(T.v[1], s.y)), lsa_fmul(T.v[2], s.z)), T.v[3]);
    float py = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[4], s.x), lsa_fmul(T.v[5], s.y)), lsa_fmul(T.v[6], s.z)), T.v[7]);
    float pz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[8], s.x), lsa_fmul(T.v[9], s.y)), lsa_fmul(T.v[10], s.z)), T.v[11]);
    if (sub == 0) work[i] = make_float4(px, py, pz, s.w);
    if (!lsa_finite3(px, py, pz)) continue;
    if (sub == 0) npts++;
    NNHit hit = nn_nearest_sub<kSub>(G, px, py, pz, kGateRings, sub, (float)gate2, prev[i]);
    if (sub == 0) prev[i] = hit.id == 0xffffffffu ? 0xffffffffu : hit.slot;
    if (sub != 0) continue;  // lane 0 of the group carries the point into the sums
    if (hit.id == 0xffffffffu || (double)hit.d2 > gate2) continue;  // distance[0] > max_dist_sqr
    ncorr++;
    acc[0] += (double)hit.d2;
    double p[3] = {px, py, pz}, q[3] = {hit.x, hit.y, hit.z};
#pragma unroll
    for (int a = 0; a < 3; a++) { acc[1 + a] += p[a]; acc[4 + a] += q[a]; }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) acc[7 + r * 3 + c] += q[r] * p[c];
  }
  __shared__ double sm[kBlock / 64][kOut];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < kOut) for (int w = 0; w < kBlock / 64; w++) sm[w][threadIdx.x] = 0.0;
  __syncthreads();
  {
    double red[18];
#pragma unroll
    for (int k = 0; k < 16; k++) red[k] = acc[k];
    red[16] = (double)ncorr; red[17] = (double)npts;
    lsa_wave_sum_n<18>(red);   // (see lsa_wave_sum_n)
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) sm[wave][k] = red[k];
      sm[wave][28] = red[16]; sm[wave][29] = red[17];
    }
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; w++) v += sm[w][threadIdx.x];
    lsa_store_row32(partials + (size_t)blockIdx.x * kOut, (int)threadIdx.x, v, key);
  }
}

__global__ __launch_bounds__(1024) void k_reduce32(const double *__restrict__ partials, int nb, double *__restrict__ out) {
  int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  double v = 0.0;
  for (int b = rl; b < nb; b += 32) v += partials[(size_t)b * kOut + col];
  __shared__ double sm[32][33];
  sm[rl][col] = v;
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (rl < st) sm[rl][col] += sm[rl + st][col];
    __syncthreads();
  }
  if (rl == 0) out[col] = sm[0][col];
}

// Sophus::SO3f::exp(omega).matrix(): unit quaternion, then Eigen's Quaternion::toRotationMatrix, all f32
void so3_exp(const float w[3], float R[9]) {
  float theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  float imag, real;
  if (theta_sq < 1e-10f) {
    float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
  } else {
    float theta = std::sqrt(theta_sq), half = 0.5f * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}
}  // namespace

struct lsa_icp {
  lsa_context *ctx = nullptr;
  float max_corr = 2.0f;
  int max_iter = 30;
  int mode = 0;
  float euclid_eps = 0.01f, trans_eps = 0.01f;  // PCL mode (config/mapping/front_end.yaml:31-35)
  NNGrid grid;
  // A second, COARSE grid over the same points (cells of the full gate, twice the fine ones) for the queries whose nearest
  // target point is far: rings 0 + 1 of it cover the gate, so such a query ends after one phased step instead of walking
  // rings 2 and 3 of the fine grid row by row (21 of ~40 us per iteration: the launch waits for those 2 % of the queries).
  // The exact answer is the same point either way (ties -> lowest original index in both); slot_of_id maps it back to the
  // fine grid's slot (seeds and per-slot normals are indexed by that).
  NNGrid grid_c;
  DevBuf<unsigned> slot_of_id;
  bool no_coarse = getenv("LSA_ICP_NO_COARSE") != nullptr;
  bool no_skip = getenv("LSA_ICP_NO_SKIP") != nullptr;
  DevBuf<float4> qref;    // per source point: where it last searched and how far every other target point was (icp_search)
  // armed passes (k_gn_pass_armed): two mail slots in fine-grained device memory (host-coherent memory where there is no large
  // BAR), the word a kernel that ran out of patience reports to, counters
  void *mail = nullptr;
  bool mail_tried = false, mail_is_device = false;
  bool no_armed = getenv("LSA_ICP_NO_ARMED") != nullptr || getenv("LSA_NO_ARMED") != nullptr;
  PinnedBuf<unsigned long long> h_gave_up;
  uint64_t armed_passes = 0, armed_fallbacks = 0;
  double armed_patience_us = 300.0;   // how long a pre-launched kernel waits for its pose before it leaves (lsa_icp_set_armed_patience)
  std::vector<uint32_t> searches_per_iter;   // last ScanMatch (manual / plane modes): queries that searched, per iteration
  uint64_t n_searches = 0, n_queries = 0;
  VoxelIndex vi_n;                                  // plane mode: voxel grid the normals come from
  DevBuf<float4> cell_normals, slot_normals;        // per occupied voxel / per cell-sorted target point
  float normal_leaf = 1.0f;
  DevBuf<float4> d_target, d_source_own, d_work;
  DevBuf<double> partials, d_out;
  PinnedBuf<double> h_out, h_partials;
  // multi-GPU ("ICP shards identically"): the source sharded over the ranks, the neighbour grid replicated, the
  // 32-double block of every iteration summed in place over the ranks by this callback (one RCCL all-reduce per iteration)
  lsa_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  unsigned long long pass_seq = 0, row_key = 0;  // check word of the rows of the pass in flight (0: plain rows)
  unsigned unsynced = 0;
  DevBuf<unsigned> prev;  // per source point: cell-sorted slot of the neighbour found one iteration ago (0xffffffff: none)
  PinnedBuf<float> h_stage;
  bool has_target = false;
  size_t n_target = 0;
  // last ScanMatch (PCL mode: GetFitnessScore re-reads the source)
  const float4 *d_last_src = nullptr;
  size_t n_last_src = 0;
  float last_T[16];
  bool has_result = false;
  lsa_icp_stats stats{};
};

namespace {
int icp_check(lsa_icp *h) {
  if (!h) return LSA_ERR_INVALID;
  hipError_t e = lsa_enter(h->ctx);
  if (e != hipSuccess) return lsa_fail(h->ctx, LSA_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->ctx->device, hipGetErrorString(e));
  return LSA_OK;
}

int icp_upload(lsa_icp *h, const void *pts, size_t n, size_t stride, DevBuf<float4> &dst) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->h_stage.reserve(ctx, n * 4))) return rc;
  if ((rc = dst.reserve(ctx, n, 1.25))) return rc;
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return lsa_upload_xyzw(ctx, pts, n, stride, SIZE_MAX, h->h_stage.p, dst.p);
}

int icp_build(lsa_icp *h, const float4 *d_pts, size_t n) {
  h->has_target = false;
  h->has_result = false;
  // manual mode gates the SQUARED distance at max_corr (sic) -> cell sqrt(max_corr); PCL mode gates the distance
  float cell = ((h->mode == LSA_ICP_MODE_MANUAL) ? std::sqrt(h->max_corr) : h->max_corr) * kCellPerGate;
  int rc = nn_grid_build(h->ctx, h->grid, d_pts, n, cell, "ICP SetInputTarget");
  if (rc) return rc;
  h->grid_c.built = false;
  if (!h->no_coarse && h->mode != LSA_ICP_MODE_PCL && h->grid.use_dense && n >= 4096) {
    if ((rc = nn_grid_build(h->ctx, h->grid_c, d_pts, n, cell / kCellPerGate, "ICP SetInputTarget (coarse grid)"))) return rc;
    if (h->grid_c.use_dense) {
      if ((rc = h->slot_of_id.reserve(h->ctx, n, 1.25))) return rc;
      hipLaunchKernelGGL(k_slot_of_id, dim3(lsa_div_up(h->grid.n_sorted, kBlock)), dim3(kBlock), 0, h->ctx->stream, h->grid.sorted.p, h->grid.n_sorted, h->slot_of_id.p);
      LSA_HIP(h->ctx, hipGetLastError());
    } else {
      h->grid_c.built = false;
    }
  }
  if (h->mode == LSA_ICP_MODE_PLANE) {
    lsa_context *ctx = h->ctx;
    VoxelIndex &vi = h->vi_n;
    const float leaf[3] = {h->normal_leaf, h->normal_leaf, h->normal_leaf};
    int status = VI_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
      if ((rc = vi_enqueue(ctx, vi, d_pts, n, /*cells floor(x / leaf)*/ 1, leaf, 10))) return rc;
      if ((rc = vi_fetch(ctx, vi))) return rc;
      if ((rc = vi_wait(ctx, vi, &status))) return rc;
      if (status != VI_CAPACITY) break;
      if ((rc = vi_grow(ctx, vi))) return rc;
    }
    if (status != VI_OK) return lsa_fail(ctx, LSA_ERR_RANGE, "ICP SetInputTarget: normal grid status %d", status);
    size_t V = vi.nocc;
    if ((rc = h->cell_normals.reserve(ctx, V + 1, 1.25))) return rc;
    if ((rc = h->slot_normals.reserve(ctx, n, 1.25))) return rc;
    hipLaunchKernelGGL(k_plane_normals, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, vi.acc.p, vi.cnt.p, V, vi.fs, h->cell_normals.p);
    // non-finite target points are absent from the sorted array; its first start[V_nn] entries are the finite ones
    hipLaunchKernelGGL(k_slot_normals, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, h->grid.sorted.p, h->grid.n_sorted, vi.map, vi.bits.p,
                       vi.prefix.p, h->cell_normals.p, h->slot_normals.p);
    LSA_HIP(ctx, hipGetLastError());
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  h->n_target = n;
  h->has_target = true;
  return LSA_OK;
}

unsigned icp_grid_size(const lsa_icp *h, size_t n) {
  unsigned g = (unsigned)std::min<size_t>((n * kSub + kBlock - 1) / kBlock, (size_t)h->ctx->num_cus * 8);
  return g < 1 ? 1 : g;
}

// Where the pass kernels write their workgroup rows: up to kHostRows rows go straight into host-mapped pinned memory and
// the host adds them (fixed order) — no reduction launch, no copy; larger grids use the device reduction.
constexpr unsigned kHostRows = 512;
// (with an exchange step installed the rows stay on the device: second stage -> all-reduce -> one copy)
bool icp_host_rows(const lsa_icp *h, unsigned g) { return !h->allreduce && g <= kHostRows; }
double *icp_partials_dst(lsa_icp *h, unsigned g) { return icp_host_rows(h, g) ? h->h_partials.p : h->partials.p; }
// host-bound rows carry a check word (lsa_store_row32) so that the host can add them as they arrive; 0 = plain rows
unsigned long long icp_next_key(lsa_icp *h, unsigned g) {
  if (!icp_host_rows(h, g)) return h->row_key = 0ull;
  unsigned long long key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  if (!key) key = (++h->pass_seq) * 0x9E3779B97F4A7C15ull;
  return h->row_key = key;
}

// brings the 32 sums of `g` partial rows to h->h_out
int icp_fetch_sums(lsa_icp *h, unsigned g, bool may_sync = true) {
  lsa_context *ctx = h->ctx;
  LSA_HIP(ctx, hipGetLastError());
  if (icp_host_rows(h, g)) {
    // rows b, b + 32, ... accumulate in ascending b, then a 32-way tree (the order k_reduce32 uses), each row as soon as
    // its check word matches; a row that is 20 ms late is waited for the ordinary way
    if (!ndt_reduce_partials_host(h->h_partials.p, g, h->row_key, 20.0, h->h_out.p)) {
      LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      h->unsynced = 0;
      if (!ndt_reduce_partials_host(h->h_partials.p, g, h->row_key, 0.0, h->h_out.p))
        return lsa_fail(ctx, LSA_ERR_HIP, "ICP pass: workgroup rows incomplete after the kernel finished");
    } else if (++h->unsynced >= 256 && may_sync) {  // let the runtime retire its commands now and then (not with an armed kernel waiting on the stream)
      LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
      h->unsynced = 0;
    }
    return LSA_OK;
  }
  hipLaunchKernelGGL(k_reduce32, dim3(1), dim3(1024), 0, ctx->stream, h->partials.p, (int)g, h->d_out.p);
  LSA_HIP(ctx, hipGetLastError());
  if (h->allreduce) {  // the exchange step: this rank's 32 sums + every other rank's, in place, on the stream
    int crc = h->allreduce(h->allreduce_user, h->d_out.p, kOut, (void *)ctx->stream);
    if (crc) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "ICP: all-reduce callback returned %d", crc);
  }
  LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, h->d_out.p, kOut * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

int icp_mail_ensure(lsa_icp *h) {
  if (h->mail || h->mail_tried) return LSA_OK;
  h->mail_tried = true;
  lsa_context *ctx = h->ctx;
  hipDeviceProp_t prop;
  void *m = nullptr;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.isLargeBar && hipExtMallocWithFlags(&m, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
    h->mail = m; h->mail_is_device = true;
  } else if (hipHostMalloc(&m, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    h->mail = m; h->mail_is_device = false;
  } else {
    (void)hipGetLastError();
    return LSA_OK;  // no mailbox: one launch per iteration, the ordinary way
  }
  int rc = h->h_gave_up.reserve(ctx, 8);
  if (rc) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->mail, 0, 4096, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

void icp_mail_post(IcpMail *base, unsigned long long seq, unsigned long long key, const IcpParams &P) {
  IcpMail *m = base + (seq & 1ull);   // two slots by parity: the mail of pass k stays whole while the kernel of pass k + 1 is already resident
  unsigned long long w[7];
  memcpy(w, P.T, 12 * sizeof(float));
  w[6] = (P.use_coarse ? 1ull : 0ull) | (P.allow_skip ? 2ull : 0ull);
  unsigned long long x = 0;
  for (int c = 0; c < 7; c++) x ^= lsa_row_mix(w[c], c);
  for (int c = 0; c < 7; c++) m->words[c] = w[c];
  m->check = x ^ key;
  _mm_sfence();   // payload before the sequence number (the BAR mapping may be write-combining)
  __atomic_store_n(&m->seq, seq, __ATOMIC_RELEASE);
  _mm_sfence();
}

constexpr double kIcpArmedRowLimitMs = 2.0;

// The part of one ICPMANUAL / point-to-plane iteration that runs on the host (icp_registration_manual.cpp:165-176) from the 32 sums
// of the iteration ([1..6] B, [7..27] the upper triangle of H): Hessian.determinant() == 0 -> the pose stays; else
// delta_x = Hessian.inverse() * B, translation += delta[0:3], rotation_matrix_ *= exp(delta[3:6]) — Eigen's own f32 operation order
// (gn_math.h; the checker's copy is pinned against the reference's vendored Eigen).  Every rank of a sharded ScanMatch runs it on the
// same reduced sums and arrives at the same bits: no broadcast.
void icp_host_step(const double *o, float R[9], float t[3]) {
  float H[36], B[6];
  int k = 7;
  for (int a = 0; a < 6; a++) B[a] = (float)o[1 + a];
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) { H[a * 6 + b] = H[b * 6 + a] = (float)o[k]; k++; }
  float d[6];
  if (gm::solve6f_eigen(H, B, d) == 0.0f) return;  // :165
  t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
  float dR[9];
  so3_exp(d + 3, dR);
  gm::mul3f_eigen(R, dR, R);
}

int icp_match_manual(lsa_icp *h, const float4 *d_src, size_t n, const float predict[16], float result[16]) {
  lsa_context *ctx = h->ctx;
  float T[16];
  memcpy(T, predict, sizeof(T));
  float R[9], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r];
    t[r] = T[12 + r];
  }
  unsigned g = icp_grid_size(h, n);
  int rc;
  if ((rc = h->partials.reserve(ctx, (size_t)g * kOut, 1.0))) return rc;
  if ((rc = h->d_out.reserve(ctx, kOut))) return rc;
  if ((rc = h->h_out.reserve(ctx, kOut))) return rc;
  if ((rc = h->h_partials.reserve(ctx, (size_t)kHostRows * kOut))) return rc;
  if ((rc = h->prev.reserve(ctx, n ? n : 1, 1.25))) return rc;
  if ((rc = h->qref.reserve(ctx, n ? n : 1, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->prev.p, 0xff, (n ? n : 1) * sizeof(unsigned), ctx->stream));  // no seeds: the first iteration searches cold
  NNView G = nn_view(h->grid);
  const bool coarse = h->grid_c.built && h->grid_c.use_dense && h->slot_of_id.p;
  NNView Gc = coarse ? nn_view(h->grid_c) : G;
  h->searches_per_iter.clear();
  h->n_searches = h->n_queries = 0;
  const bool plane = h->mode == LSA_ICP_MODE_PLANE;
  // armed passes: iteration 0 is launched with its parameters; from then on the kernel of iteration k + 1 is enqueued while k
  // runs and takes its pose by mail
  bool armed = !h->no_armed && icp_host_rows(h, g) && h->max_iter > 1;   // (not with an exchange step: every iteration ends in a collective)
  if (armed && (rc = icp_mail_ensure(h))) return rc;
  armed = armed && h->mail;
  const float gate = plane ? h->max_corr * h->max_corr : h->max_corr;
  unsigned long long seq_next = 0, key_next = 0;   // the armed kernel in flight, if any
  auto launch_plain = [&](const IcpParams &P) {
    if (plane) hipLaunchKernelGGL(k_gn_pass<true>, dim3(g), dim3(kBlock), 0, ctx->stream, d_src, n, P, G, Gc, h->slot_of_id.p, h->prev.p, h->qref.p, h->slot_normals.p, icp_partials_dst(h, g), icp_next_key(h, g));
    else hipLaunchKernelGGL(k_gn_pass<false>, dim3(g), dim3(kBlock), 0, ctx->stream, d_src, n, P, G, Gc, h->slot_of_id.p, h->prev.p, h->qref.p, (const float4 *)nullptr, icp_partials_dst(h, g), icp_next_key(h, g));
  };
  auto launch_armed = [&]() {
    const unsigned long long current = h->row_key;   // (the key of the pass whose rows are being waited for stays)
    (void)icp_next_key(h, g);
    seq_next = h->pass_seq; key_next = h->row_key;
    h->row_key = current;
    const IcpMail *slot = reinterpret_cast<const IcpMail *>(h->mail) + (seq_next & 1ull);
    const unsigned long long ticks = (unsigned long long)(h->armed_patience_us * 100.0);   // wall_clock64: 100 MHz
    if (plane) hipLaunchKernelGGL(k_gn_pass_armed<true>, dim3(g), dim3(kBlock), 0, ctx->stream, d_src, n, gate, G, Gc, h->slot_of_id.p, h->prev.p, h->qref.p, h->slot_normals.p, h->h_partials.p, key_next, slot, seq_next, ticks, h->h_gave_up.p);
    else hipLaunchKernelGGL(k_gn_pass_armed<false>, dim3(g), dim3(kBlock), 0, ctx->stream, d_src, n, gate, G, Gc, h->slot_of_id.p, h->prev.p, h->qref.p, (const float4 *)nullptr, h->h_partials.p, key_next, slot, seq_next, ticks, h->h_gave_up.p);
  };
  for (int it = 0; it < h->max_iter; it++) {
    IcpParams P;
    P.use_coarse = (coarse && it > 0) ? 1 : 0;   // the first iteration searches cold: no seed says who is far
    P.allow_skip = (!h->no_skip && it > 0) ? 1 : 0;   // ... and leaves every query's proof radius
    P.gate = gate;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) P.T[r * 4 + c] = R[r * 3 + c];
      P.T[r * 4 + 3] = t[r];
    }
    bool fetched = false;
    if (armed && it > 0) {
      // this iteration's kernel is resident (or about to be): hand it the pose, enqueue its successor, take its rows
      const unsigned long long seq = seq_next, key = key_next;
      icp_mail_post(reinterpret_cast<IcpMail *>(h->mail), seq, key, P);
      if (it + 1 < h->max_iter) launch_armed();
      LSA_HIP(ctx, hipGetLastError());
      h->armed_passes++;
      if (ndt_reduce_partials_host(h->h_partials.p, g, key, kIcpArmedRowLimitMs, h->h_out.p, h->h_gave_up.p, seq)) {
        fetched = true;
        if (++h->unsynced >= 256) {  // let the runtime retire its commands now and then: behind the LAST kernel only, nothing may be parked
          if (it + 1 == h->max_iter) { LSA_HIP(ctx, hipStreamSynchronize(ctx->stream)); h->unsynced = 0; }
        }
      } else {
        // its rows did not come (it ran out of patience, or something held the stream): drain the stream — the successor, which
        // will get no mail, leaves after its patience — look once more, and go on one launch per iteration
        LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        h->unsynced = 0;
        h->armed_fallbacks++;
        armed = false;
        fetched = ndt_reduce_partials_host(h->h_partials.p, g, key, 0.0, h->h_out.p);
      }
    }
    if (!fetched) {
      launch_plain(P);
      if (armed && it + 1 < h->max_iter) launch_armed();   // (iteration 0)
      if ((rc = icp_fetch_sums(h, g, /*may_sync=*/!armed))) return rc;
    }
    const double *o = h->h_out.p;
    h->searches_per_iter.push_back((uint32_t)o[30]);
    h->n_searches += (uint64_t)o[30]; h->n_queries += (uint64_t)o[29];
    icp_host_step(o, R, t);
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
    T[12 + r] = t[r];
  }
  memcpy(result, T, sizeof(T));
  h->stats.iterations = h->max_iter; h->stats.converged = 1; h->stats.correspondences = h->max_iter ? h->h_out.p[28] : 0; h->stats.mse = 0;
  return LSA_OK;
}

// ---- PCL mode host side: 3x3 SVD (one-sided Jacobi, f64) and Eigen::umeyama(src, dst, false) from the 17 sums ----
void svd3(const double M[9], double U[9], double S[3], double V[9]) {
  double A[3][3], W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = M[i * 3 + j];
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; k++) { al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q]; }
        double lim = std::sqrt(al * be);
        if (ga == 0.0 || std::fabs(ga) <= 1e-18 * lim) continue;
        off = std::fmax(off, std::fabs(ga) / (lim > 0 ? lim : 1));
        double zeta = (be - al) / (2.0 * ga);
        double t = ((zeta >= 0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
          double wp = W[k][p], wq = W[k][q];
          W[k][p] = c * wp - s * wq; W[k][q] = s * wp + c * wq;
        }
      }
    if (off < 1e-16) break;
  }
  int ord[3] = {0, 1, 2};
  double sv[3];
  for (int j = 0; j < 3; j++) sv[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++)
      if (sv[ord[j]] > sv[ord[i]]) std::swap(ord[i], ord[j]);
  for (int c = 0; c < 3; c++) {
    int j = ord[c];
    S[c] = sv[j];
    for (int r = 0; r < 3; r++) { V[r * 3 + c] = W[r][j]; U[r * 3 + c] = (sv[j] > 0) ? A[r][j] / sv[j] : 0.0; }
  }
  if (S[2] == 0 || S[2] <= 1e-300 * (S[0] > 0 ? S[0] : 1)) {  // rank deficient: complete U to an orthonormal basis
    if (S[1] == 0) {
      double u0[3] = {U[0], U[3], U[6]};
      if (S[0] == 0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
      double e[3] = {0, 0, 0};
      e[(std::fabs(u0[0]) < 0.9) ? 0 : 1] = 1;
      double u1[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
      double nn = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      for (int r = 0; r < 3; r++) U[r * 3 + 1] = u1[r] / nn;
    }
    double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

double det3(const double m[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// sums: n pairs, sp = sum p (source), sq = sum q (target), sqp[r][c] = sum q_r p_c.  T: column-major increment.
void umeyama_from_sums(double n, const double sp[3], const double sq[3], const double sqp[9], float T[16]) {
  double pm[3], qm[3], M[9], U[9], S[3], V[9];
  for (int a = 0; a < 3; a++) { pm[a] = sp[a] / n; qm[a] = sq[a] / n; }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) M[r * 3 + c] = sqp[r * 3 + c] / n - qm[r] * pm[c];  // sigma = 1/n sum (q - qm)(p - pm)^T
  svd3(M, U, S, V);
  double Sd[3] = {1, 1, 1};
  if (det3(M) < 0) Sd[2] = -1;
  int rank = 0;
  for (int i = 0; i < 3; i++)
    if (!(std::fabs(S[i]) <= std::fabs(S[0]) * 1.1920929e-07)) rank++;  // !isMuchSmallerThan(d(i), d(0)) at f32 precision
  if (rank == 2) Sd[2] = (det3(U) * det3(V) > 0) ? 1 : -1;
  float Rf[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      Rf[r * 3 + c] = (float)(U[r * 3 + 0] * Sd[0] * V[c * 3 + 0] + U[r * 3 + 1] * Sd[1] * V[c * 3 + 1] + U[r * 3 + 2] * Sd[2] * V[c * 3 + 2]);
  float pmf[3] = {(float)pm[0], (float)pm[1], (float)pm[2]}, qmf[3] = {(float)qm[0], (float)qm[1], (float)qm[2]};
  for (int k = 0; k < 16; k++) T[k] = 0.0f;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[c * 4 + r] = Rf[r * 3 + c];
    T[12 + r] = qmf[r] - ((Rf[r * 3 + 0] * pmf[0] + Rf[r * 3 + 1] * pmf[1]) + Rf[r * 3 + 2] * pmf[2]);
  }
  T[15] = 1.0f;
}

void mat4_mul_f32(const float A[16], const float B[16], float C[16]) {  // column-major, C = A * B (Eigen::Matrix4f product)
  float o[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) o[c * 4 + r] = ((A[0 * 4 + r] * B[c * 4 + 0] + A[1 * 4 + r] * B[c * 4 + 1]) + A[2 * 4 + r] * B[c * 4 + 2]) + A[3 * 4 + r] * B[c * 4 + 3];
  memcpy(C, o, sizeof(o));
}

T12 to_t12(const float T[16]) {
  T12 o;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) o.v[r * 4 + c] = T[c * 4 + r];
  return o;
}

int icp_match_pcl(lsa_icp *h, const float4 *d_src, size_t n, const float predict[16], float result[16]) {
  lsa_context *ctx = h->ctx;
  unsigned g = icp_grid_size(h, n);
  int rc;
  if ((rc = h->partials.reserve(ctx, (size_t)g * kOut, 1.0))) return rc;
  if ((rc = h->d_out.reserve(ctx, kOut))) return rc;
  if ((rc = h->h_out.reserve(ctx, kOut))) return rc;
  if ((rc = h->h_partials.reserve(ctx, (size_t)kHostRows * kOut))) return rc;
  if ((rc = h->d_work.reserve(ctx, n ? n : 1, 1.25))) return rc;
  if ((rc = h->prev.reserve(ctx, n ? n : 1, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->prev.p, 0xff, (n ? n : 1) * sizeof(unsigned), ctx->stream));
  NNView G = nn_view(h->grid);
  float final_T[16], incr[16];
  memcpy(final_T, predict, sizeof(final_T));
  memcpy(incr, predict, sizeof(incr));  // the first pass applies the guess to the input
  const double gate2 = (double)h->max_corr * (double)h->max_corr;
  const double rot_thr = 1.0 - (double)h->trans_eps, tr_thr = (double)h->trans_eps, mse_rel = (double)h->euclid_eps, mse_abs = 1e-12;
  double prev_mse = 1.7976931348623157e308, cur_mse = 0, cnt = 0;
  int iterations = 0, converged = 0;
  const float4 *in = d_src;
  for (;;) {
    hipLaunchKernelGGL(k_pcl_pass, dim3(g), dim3(kBlock), 0, ctx->stream, in, h->d_work.p, n, to_t12(incr), gate2, G, h->prev.p, icp_partials_dst(h, g), icp_next_key(h, g));
    if ((rc = icp_fetch_sums(h, g))) return rc;
    in = h->d_work.p;
    const double *o = h->h_out.p;
    cnt = o[28];
    if (cnt < 3) { converged = 0; break; }  // min_number_correspondences_: "Not enough correspondences found"
    umeyama_from_sums(cnt, o + 1, o + 4, o + 7, incr);
    mat4_mul_f32(incr, final_T, final_T);
    ++iterations;
    // DefaultConvergenceCriteria::hasConverged
    if (iterations >= h->max_iter) { converged = 1; break; }
    double cos_angle = 0.5 * ((double)incr[0] + (double)incr[5] + (double)incr[10] - 1.0);
    double tsq = (double)incr[12] * incr[12] + (double)incr[13] * incr[13] + (double)incr[14] * incr[14];
    if (cos_angle >= rot_thr && tsq <= tr_thr) { converged = 1; break; }
    cur_mse = o[0] / cnt;
    if (std::fabs(cur_mse - prev_mse) < mse_abs) { converged = 1; break; }
    if (std::fabs(cur_mse - prev_mse) / prev_mse < mse_rel) { converged = 1; break; }
    prev_mse = cur_mse;
  }
  memcpy(result, final_T, sizeof(final_T));
  h->stats.iterations = iterations; h->stats.converged = converged; h->stats.correspondences = cnt; h->stats.mse = cur_mse;
  return LSA_OK;
}

int icp_match(lsa_icp *h, const float4 *d_src, size_t n, const float predict[16], float result[16]) {
  if (!h->has_target) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "ICP ScanMatch before SetInputTarget");
  int rc = (h->mode == LSA_ICP_MODE_PCL) ? icp_match_pcl(h, d_src, n, predict, result) : icp_match_manual(h, d_src, n, predict, result);  // manual + plane share the GN loop
  if (rc) return rc;
  h->d_last_src = d_src; h->n_last_src = n;
  memcpy(h->last_T, result, sizeof(h->last_T));
  h->has_result = true;
  return LSA_OK;
}
}  // namespace

extern "C" int lsa_icp_create(lsa_icp **out, lsa_context *ctx, float max_correspond_dis, int max_iter, int mode) {
  if (!out || !ctx) { lsa_set_global_error("lsa_icp_create: null argument"); return LSA_ERR_INVALID; }
  *out = nullptr;
  if (mode != LSA_ICP_MODE_MANUAL && mode != LSA_ICP_MODE_PCL && mode != LSA_ICP_MODE_PLANE)
    return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_create: unknown mode %d (0 = ICPMANUAL, 1 = point-to-plane extension, 2 = ICP / PCL semantics)", mode);
  if (!(max_correspond_dis > 0) || max_iter < 0) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_create: bad parameters");
  lsa_icp *h = new lsa_icp();
  h->ctx = ctx; h->max_corr = max_correspond_dis; h->max_iter = max_iter; h->mode = mode;
  h->pass_seq = lsa_seq_base();   // row check words from a range of this handle's own (lsa_common.h)
  *out = h;
  return LSA_OK;
}

extern "C" int lsa_icp_set_allreduce(lsa_icp *h, lsa_allreduce_fn fn, void *user) {
  if (!h) return LSA_ERR_INVALID;
  h->allreduce = fn;
  h->allreduce_user = user;
  return LSA_OK;
}

// The host driver of ICPMANUAL / the point-to-plane extension with a caller-supplied evaluator of the per-iteration sums (what
// lsa_gn_solve is for NDT): the multi-process path without a GPU — tests/test_dist_cpu.py evaluates each rank's shard with the CPU
// oracle and sums over gloo.  eval fills out32 as the pass kernels do ([1..6] B, [7..27] H upper triangle row by row, [28] pairs),
// already summed over the ranks, for the pose T (column-major 4x4).
extern "C" int lsa_icp_gn_solve(lsa_icp_eval_fn eval, void *user, int max_iter, const float predict_pose[16], float result_pose[16]) {
  if (!eval || !predict_pose || !result_pose || max_iter < 0) return LSA_ERR_INVALID;
  float T[16];
  memcpy(T, predict_pose, sizeof(T));
  float R[9], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r];
    t[r] = T[12 + r];
  }
  for (int it = 0; it < max_iter; it++) {
    double o[32];
    for (int c = 0; c < 32; c++) o[c] = 0.0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
      T[12 + r] = t[r];
    }
    int rc = eval(user, T, o);
    if (rc) return LSA_ERR_COLLECTIVE;
    icp_host_step(o, R, t);
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
    T[12 + r] = t[r];
  }
  memcpy(result_pose, T, sizeof(T));
  return LSA_OK;
}

extern "C" int lsa_icp_set_criteria(lsa_icp *h, float euclid_eps, float trans_eps) {
  if (!h) return LSA_ERR_INVALID;
  h->euclid_eps = euclid_eps; h->trans_eps = trans_eps;
  return LSA_OK;
}

extern "C" int lsa_icp_set_normal_leaf(lsa_icp *h, float leaf) {
  if (!h) return LSA_ERR_INVALID;
  if (!(leaf > 0)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_icp_set_normal_leaf: leaf must be > 0");
  h->normal_leaf = leaf;
  h->has_target = false;  // the normals belong to the old leaf
  return LSA_OK;
}

extern "C" int lsa_icp_get_stats(lsa_icp *h, lsa_icp_stats *st) {
  if (!h || !st) return LSA_ERR_INVALID;
  *st = h->stats;
  return LSA_OK;
}

extern "C" void lsa_icp_destroy(lsa_icp *h) {
  if (!h) return;
  (void)lsa_enter(h->ctx);
  (void)hipStreamSynchronize(h->ctx->stream);
  h->grid.release(); h->grid_c.release(); h->slot_of_id.release(); h->vi_n.release(); h->cell_normals.release(); h->slot_normals.release(); h->d_target.release(); h->d_source_own.release(); h->d_work.release();
  h->partials.release(); h->d_out.release(); h->h_out.release(); h->h_partials.release(); h->h_stage.release(); h->prev.release(); h->qref.release(); h->h_gave_up.release();
  if (h->mail) { if (h->mail_is_device) (void)hipFree(h->mail); else (void)hipHostFree(h->mail); }
  delete h;
}

extern "C" int lsa_icp_set_input_target_device(lsa_icp *h, const float *d_xyzw, size_t n) {
  int rc = icp_check(h);
  if (rc) return rc;
  if (!n || !d_xyzw) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ICP SetInputTarget: empty target");
  return icp_build(h, (const float4 *)d_xyzw, n);
}

extern "C" int lsa_icp_set_input_target(lsa_icp *h, const void *pts, size_t n, size_t stride) {
  int rc = icp_check(h);
  if (rc) return rc;
  if (!n || !pts || stride < 12) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ICP SetInputTarget: bad points / stride");
  if ((rc = icp_upload(h, pts, n, stride, h->d_target))) return rc;
  return icp_build(h, h->d_target.p, n);
}

extern "C" int lsa_icp_search_counts(lsa_icp *h, uint64_t *searches, uint64_t *queries, uint32_t *per_iteration, size_t cap, size_t *n_iterations) {
  if (!h) return LSA_ERR_INVALID;
  if (searches) *searches = h->n_searches;
  if (queries) *queries = h->n_queries;
  if (n_iterations) *n_iterations = h->searches_per_iter.size();
  if (per_iteration)
    for (size_t i = 0; i < cap && i < h->searches_per_iter.size(); i++) per_iteration[i] = h->searches_per_iter[i];
  return LSA_OK;
}

extern "C" int lsa_icp_set_armed_patience(lsa_icp *h, double microseconds) {
  if (!h) return LSA_ERR_INVALID;
  if (!(microseconds >= 0.0) || microseconds > 1.0e6) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_icp_set_armed_patience: %g us (0 .. 1e6)", microseconds);
  h->armed_patience_us = microseconds;
  return LSA_OK;
}

extern "C" int lsa_icp_armed_counts(lsa_icp *h, uint64_t *armed_passes, uint64_t *fallbacks) {
  if (!h) return LSA_ERR_INVALID;
  if (armed_passes) *armed_passes = h->armed_passes;
  if (fallbacks) *fallbacks = h->armed_fallbacks;
  return LSA_OK;
}

// diagnostics (tests, profiles/scripts): the per-query search state the last ScanMatch left on the device, copied to the host —
// prev[i]: cell-sorted slot of source point i's neighbour (0xffffffff: none), qref[4 i ..]: where it last searched + proof radius
extern "C" int lsa_icp_query_state(lsa_icp *h, uint32_t *prev_out, float *qref_out, size_t n) {
  if (!h || !prev_out || !qref_out) return LSA_ERR_INVALID;
  lsa_context *ctx = h->ctx;
  LSA_HIP(ctx, lsa_enter(ctx));
  if (n > h->prev.cap || n > h->qref.cap) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_query_state: %zu points asked for, the last ScanMatch had fewer", n);
  LSA_HIP(ctx, hipMemcpyAsync(prev_out, h->prev.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(qref_out, h->qref.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

extern "C" int lsa_icp_scan_match_device(lsa_icp *h, const float *d_src_xyzw, size_t n, const float predict_pose[16], float result_pose[16]) {
  int rc = icp_check(h);
  if (rc) return rc;
  if (!predict_pose || !result_pose || (n && !d_src_xyzw)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ICP ScanMatch: null argument");
  return icp_match(h, (const float4 *)d_src_xyzw, n, predict_pose, result_pose);
}

extern "C" int lsa_icp_scan_match(lsa_icp *h, const void *src, size_t n, size_t stride, const float predict_pose[16], float result_pose[16]) {
  int rc = icp_check(h);
  if (rc) return rc;
  if (!predict_pose || !result_pose || (n && (!src || stride < 12))) return lsa_fail(h->ctx, LSA_ERR_INVALID, "ICP ScanMatch: bad argument");
  if (n && (rc = icp_upload(h, src, n, stride, h->d_source_own))) return rc;
  return icp_match(h, h->d_source_own.p, n, predict_pose, result_pose);
}

extern "C" int lsa_icp_get_fitness_score(lsa_icp *h, float *score) {
  if (!h || !score) return LSA_ERR_INVALID;
  *score = 0.0f;  // ICPMANUAL: "not implemented" in the reference, prints a notice and returns 0 (:62-67)
  if (h->mode != LSA_ICP_MODE_PCL) return LSA_OK;
  int rc = icp_check(h);
  if (rc) return rc;
  if (!h->has_target || !h->has_result) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "GetFitnessScore before SetInputTarget + ScanMatch");
  double v = 0;
  if ((rc = nn_fitness(h->ctx, h->grid, h->d_last_src, h->n_last_src, h->last_T, INFINITY, &v, h->allreduce, h->allreduce_user))) return rc;
  *score = (float)v;  // icp_registration.cpp:60-63: float GetFitnessScore() { return icp_ptr_->getFitnessScore(); }
  return LSA_OK;
}
