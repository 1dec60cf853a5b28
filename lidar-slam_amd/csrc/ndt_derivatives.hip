// ndt_derivatives.hip — the hot loop: one NDT derivative pass over the source cloud.
//
// Replaces NormalDistributionsTransform::computeDerivatives / computeHessian and what they call per point
// (reference: lidar_localization/src/models/registration/ndt_registration_manual/NormalDistributionsTransform.cpp
//  :391-445 computeDerivatives, :448-482 computePointDerivatives, :485-520 updateDerivatives, :875-899 updateHessian,
//  :901-936 computeHessian; VoxelGrid.cpp:432-480 radiusSearch; pcl::transformPointCloud, f32).
//
// Per 64-point tile: f32 rigid transform (bit-identical to PCL's, no FMA contraction), block-cooperative probe of the
// <= 3x3x3 cell neighbourhood through the {occupancy, valid, prefix} words, compaction of the hits into an LDS pair
// queue, then dense f64 pair math on the gathered 80-byte voxel records exploiting the symmetry of C^-1 (21 unique
// Hessian entries + 6 gradient + score = 28 values), wave shuffle reduction -> LDS -> one 32-double partial per
// workgroup, then a fixed-order second-stage reduction (no atomics: bitwise reproducible run to run).
// No MFMA: the largest contraction is 6x6.
//
// Algorithmic bytes per pass (SURVEY.md 8d formula C, with this implementation's record size):
//   16*M (points) + 27*16*M (cell words, worst case) + 80*Kbar*M (records); the word table and the records are
//   L2 / Infinity-Cache resident at every BASELINE.json size, so the pass is latency / f64-issue bound, not HBM bound.
#include "ndt_internal.h"
#include "ndt_pass.h"

namespace {

constexpr int kBlockD = 256;
#ifndef LSA_PASS_WAVES
#define LSA_PASS_WAVES 2  // min waves per SIMD requested from the register allocator (2 x 256-thread workgroups per CU)
#endif

template <int KIND>
__device__ __forceinline__ void pair_update(const PassParams &P, const double *__restrict__ R, float tx, float ty, float tz,
                                            const double pg[8], const double ph[15], double acc[28]) {
  // R: centroid(3), C = icov (xx,xy,xz,yy,yz,zz), w.  The radius test (VoxelGrid.cpp:470-472) was done by the probe phase.
  double d0 = (double)tx - R[0], d1 = (double)ty - R[1], d2 = (double)tz - R[2];  // x_trans - centroid (:434)
  double C00 = R[3], C01 = R[4], C02 = R[5], C11 = R[6], C12 = R[7], C22 = R[8];
  double u0 = C00 * d0 + C01 * d1 + C02 * d2;
  double u1 = C01 * d0 + C11 * d1 + C12 * d2;
  double u2 = C02 * d0 + C12 * d1 + C22 * d2;
  double q = d0 * u0 + d1 * u1 + d2 * u2;
  double e = exp(-P.d2 * q / 2);  // :494
  double f = P.d2 * e;
  if (f > 1 || f < 0 || f != f) return;  // :499-501 / :884-886
  if (KIND != 2) acc[0] += R[9] * (-P.d1 * e);  // static weight multiplies the score only (:440-441)
  f *= P.d1;
  // J columns 3..5: J3 = (0, pg0, pg1), J4 = (pg2, pg3, pg4), J5 = (pg5, pg6, pg7)   (:453-460)
  double uJ3 = u1 * pg[0] + u2 * pg[1];
  double uJ4 = u0 * pg[2] + u1 * pg[3] + u2 * pg[4];
  double uJ5 = u0 * pg[5] + u1 * pg[6] + u2 * pg[7];
  if (KIND != 2) {
    acc[1] += f * u0; acc[2] += f * u1; acc[3] += f * u2;
    acc[4] += f * uJ3; acc[5] += f * uJ4; acc[6] += f * uJ5;
  }
  if (KIND != 1) {
    double nd2 = -P.d2;
    // C * J_j for j = 3..5
    double a3x = C01 * pg[0] + C02 * pg[1], a3y = C11 * pg[0] + C12 * pg[1], a3z = C12 * pg[0] + C22 * pg[1];
    double a4x = C00 * pg[2] + C01 * pg[3] + C02 * pg[4], a4y = C01 * pg[2] + C11 * pg[3] + C12 * pg[4], a4z = C02 * pg[2] + C12 * pg[3] + C22 * pg[4];
    double a5x = C00 * pg[5] + C01 * pg[6] + C02 * pg[7], a5y = C01 * pg[5] + C11 * pg[6] + C12 * pg[7], a5z = C02 * pg[5] + C12 * pg[6] + C22 * pg[7];
    // rows 0..2 (translation)
    acc[7] += f * (nd2 * u0 * u0 + C00);
    acc[8] += f * (nd2 * u0 * u1 + C01);
    acc[9] += f * (nd2 * u0 * u2 + C02);
    acc[10] += f * (nd2 * u0 * uJ3 + a3x);
    acc[11] += f * (nd2 * u0 * uJ4 + a4x);
    acc[12] += f * (nd2 * u0 * uJ5 + a5x);
    acc[13] += f * (nd2 * u1 * u1 + C11);
    acc[14] += f * (nd2 * u1 * u2 + C12);
    acc[15] += f * (nd2 * u1 * uJ3 + a3y);
    acc[16] += f * (nd2 * u1 * uJ4 + a4y);
    acc[17] += f * (nd2 * u1 * uJ5 + a5y);
    acc[18] += f * (nd2 * u2 * u2 + C22);
    acc[19] += f * (nd2 * u2 * uJ3 + a3z);
    acc[20] += f * (nd2 * u2 * uJ4 + a4z);
    acc[21] += f * (nd2 * u2 * uJ5 + a5z);
    // rows 3..5 (rotation): u . H_E block + J_j . (C J_i)   (:465-480, :512-514)
    // blocks: (3,3)=a=(0,ph0,ph1) (3,4)=b=(0,ph2,ph3) (3,5)=c=(0,ph4,ph5) (4,4)=d=(ph6..8) (4,5)=e=(ph9..11) (5,5)=f=(ph12..14)
    acc[22] += f * (nd2 * uJ3 * uJ3 + (u1 * ph[0] + u2 * ph[1]) + (pg[0] * a3y + pg[1] * a3z));
    acc[23] += f * (nd2 * uJ3 * uJ4 + (u1 * ph[2] + u2 * ph[3]) + (pg[2] * a3x + pg[3] * a3y + pg[4] * a3z));
    acc[24] += f * (nd2 * uJ3 * uJ5 + (u1 * ph[4] + u2 * ph[5]) + (pg[5] * a3x + pg[6] * a3y + pg[7] * a3z));
    acc[25] += f * (nd2 * uJ4 * uJ4 + (u0 * ph[6] + u1 * ph[7] + u2 * ph[8]) + (pg[2] * a4x + pg[3] * a4y + pg[4] * a4z));
    acc[26] += f * (nd2 * uJ4 * uJ5 + (u0 * ph[9] + u1 * ph[10] + u2 * ph[11]) + (pg[5] * a4x + pg[6] * a4y + pg[7] * a4z));
    acc[27] += f * (nd2 * uJ5 * uJ5 + (u0 * ph[12] + u1 * ph[13] + u2 * ph[14]) + (pg[5] * a5x + pg[6] * a5y + pg[7] * a5z));
  }
}

// One workgroup (256 lanes = 4 waves) walks tiles of 64 source points in three phases per tile:
//   0. every wave loads the tile's 64 points; wave 0 transforms them (f32, PCL order) and publishes the clipped 3x3x3
//      cell box, and each wave computes its quarter of the 23 point-derivative dot products (8 Jacobian + 15 Hessian
//      terms, computePointDerivatives) into LDS -- once per POINT, not once per pair as the reference does;
//   A. all 256 lanes PROBE: 64 points x 27 cells = 1728 probes, 7 per lane, each = one 16-B {occ,valid,prefix} word from
//      L2, a popcount rank, the 24-B centroid and the radius test (on the squared distance against a host-computed
//      exact threshold: no f64 sqrt per probe).  Hits are compacted into an LDS pair queue at offsets from a block-wide
//      exclusive scan (not atomics), so the queue order -- and with it every f64 sum -- is reproducible run to run;
//   B. all lanes drain the queue: one (point, voxel) pair per lane per step, dense f64 pair math into 28 registers.
// Only Kbar ~ 2.5-5 of the 27 probed cells hold a usable voxel; separating the cheap probe from the expensive pair
// math keeps the f64 pipes full instead of leaving ~85 % of the lanes idle in a per-point loop, and spreads a
// KITTI-sized pass (7-25 k points) over >= 118 workgroups.  The 69 angle-derivative constants are touched only in
// phase 0 (a quarter per wave), which keeps them out of the pair loop's scalar registers.
constexpr int kTilePts = 64;
constexpr int kProbesPerPoint = 27;
constexpr int kProbesPerLane = (kTilePts * kProbesPerPoint + kBlockD - 1) / kBlockD;  // 7
constexpr int kQueueCap = kTilePts * kProbesPerPoint;                                 // 1728
constexpr int kPointTerms = 23;   // pg[8] + ph[15]
constexpr int kTermStride = 24;
constexpr int kRedStride = 29;    // 28 accumulators per lane, padded: conflict-free LDS transpose

struct TilePoint {
  float sx, sy, sz;  // source point (untransformed)
  float tx, ty, tz;  // transformed point
  int cell0;         // linear index of the box's low corner
  int nxyz;          // box extent packed nx | ny << 8 | nz << 16 (0 = point contributes nothing)
};

constexpr int kLdsTp = 0;
constexpr int kLdsTerms = kLdsTp + kTilePts * (int)sizeof(TilePoint);                    // 2048
constexpr int kLdsQueue = kLdsTerms + kTilePts * kTermStride * 8;                        // + 12288
constexpr int kLdsWaveTot = kLdsQueue + kQueueCap * 4;                                   // + 6912
constexpr int kLdsTileEnd = kLdsWaveTot + 16;
constexpr int kLdsRed = kBlockD * kRedStride * 8;                                        // 59392 (aliases the tile area after the loop)
constexpr int kLdsBytes = (kLdsRed > kLdsTileEnd ? kLdsRed : kLdsTileEnd) + 8 * 32 * 8;  // + second-stage [8][32]

template <int KIND>
__global__ __launch_bounds__(kBlockD, LSA_PASS_WAVES) void k_ndt_pass(const float4 *__restrict__ src, size_t n, PassParams P, const uint4 *__restrict__ words,
                                                      const double *__restrict__ rec, double *__restrict__ partials) {
  __shared__ __align__(16) unsigned char lds[kLdsBytes];
  TilePoint *tp = reinterpret_cast<TilePoint *>(lds + kLdsTp);
  double *terms = reinterpret_cast<double *>(lds + kLdsTerms);
  unsigned *queue = reinterpret_cast<unsigned *>(lds + kLdsQueue);
  unsigned *wave_tot = reinterpret_cast<unsigned *>(lds + kLdsWaveTot);
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; k++) acc[k] = 0.0;
  unsigned long long pairs = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dimx = P.dim[0], dimxy = P.dim[0] * P.dim[1];
  const size_t ntiles = (n + kTilePts - 1) / kTilePts;
  const double *jh = &P.ja[0][0];  // ja[8][3] followed by ha[15][3]: 23 constant 3-vectors
  // software prefetch of the next tile's points (every wave holds its own copy of the 64 points)
  float4 s_next = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    size_t i0 = (size_t)blockIdx.x * kTilePts + lane;
    if (i0 < n) s_next = src[i0];
  }
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- phase 0
    {
      size_t i = tile * kTilePts + lane;
      float4 s = s_next;
      size_t inext = (tile + gridDim.x) * kTilePts + lane;
      if (inext < n) s_next = src[inext];
      if (wave == 0) {
        TilePoint t;
        t.nxyz = 0; t.cell0 = 0; t.sx = s.x; t.sy = s.y; t.sz = s.z; t.tx = t.ty = t.tz = 0.f;
        if (i < n) {
          // pcl::transformPointCloud: x' = m00*x + m01*y + m02*z + m03, f32, left to right, no FMA
          float tx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[0], s.x), lsa_fmul(P.T[1], s.y)), lsa_fmul(P.T[2], s.z)), P.T[3]);
          float ty = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[4], s.x), lsa_fmul(P.T[5], s.y)), lsa_fmul(P.T[6], s.z)), P.T[7]);
          float tz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[8], s.x), lsa_fmul(P.T[9], s.y)), lsa_fmul(P.T[10], s.z)), P.T[11]);
          t.tx = tx; t.ty = ty; t.tz = tz;
          if (lsa_finite3(tx, ty, tz)) {
            // radiusSearch cell box (VoxelGrid.cpp:438-455), relative to real_min, clipped to the occupied bounds
            float r = P.res;
            int hx = (int)floorf(lsa_fdiv(lsa_fadd(tx, r), r)) - P.min_b[0], lx = (int)floorf(lsa_fdiv(lsa_fsub(tx, r), r)) - P.min_b[0];
            int hy = (int)floorf(lsa_fdiv(lsa_fadd(ty, r), r)) - P.min_b[1], ly = (int)floorf(lsa_fdiv(lsa_fsub(ty, r), r)) - P.min_b[1];
            int hz = (int)floorf(lsa_fdiv(lsa_fadd(tz, r), r)) - P.min_b[2], lz = (int)floorf(lsa_fdiv(lsa_fsub(tz, r), r)) - P.min_b[2];
            hx = min(hx, P.dim[0] - 1); hy = min(hy, P.dim[1] - 1); hz = min(hz, P.dim[2] - 1);
            lx = max(lx, 0); ly = max(ly, 0); lz = max(lz, 0);
            // with radius == leaf the box is at most 3 cells wide per axis
            int nx = min(hx - lx + 1, 3), ny = min(hy - ly + 1, 3), nz = min(hz - lz + 1, 3);
            if (nx > 0 && ny > 0 && nz > 0) {
              t.cell0 = lx + ly * dimx + lz * dimxy;
              t.nxyz = nx | (ny << 8) | (nz << 16);
            }
          }
        }
        tp[lane] = t;
      }
      // computePointDerivatives (:448-482): dot products of the source point with the angle-derivative vectors;
      // wave w owns terms 6w .. 6w+5 (KIND 1 needs only the 8 Jacobian terms)
      double x = s.x, y = s.y, z = s.z;
      const int k0 = wave * 6, k1 = min(k0 + 6, (KIND == 1) ? 8 : kPointTerms);
      for (int k = k0; k < k1; k++) terms[lane * kTermStride + k] = x * jh[k * 3] + y * jh[k * 3 + 1] + z * jh[k * 3 + 2];
    }
    __syncthreads();
    // ---- phase A: probe
    // The 7 probes of a lane are independent; they are staged (all word loads, then all centroid loads) so the L2 round
    // trips overlap instead of forming 14 dependent hops.  Out-of-box probes read word 0 / record 0 and are masked out.
    unsigned hit_rank[kProbesPerLane];
    unsigned hit_mask = 0;
    int p_pl[kProbesPerLane], p_cell[kProbesPerLane];
    uint4 p_w[kProbesPerLane];
#pragma unroll
    for (int j = 0; j < kProbesPerLane; j++) {
      int q = threadIdx.x + j * kBlockD;
      int pl = (q < kQueueCap) ? q / kProbesPerPoint : 0;
      int o = q - pl * kProbesPerPoint;
      int dz = o / 9, dy = (o - dz * 9) / 3, dx = o - dz * 9 - dy * 3;
      int nxyz = tp[pl].nxyz;
      bool in_box = q < kQueueCap && dx < (nxyz & 0xff) && dy < ((nxyz >> 8) & 0xff) && dz < (nxyz >> 16);
      p_pl[j] = pl;
      p_cell[j] = in_box ? tp[pl].cell0 + dx + dy * dimx + dz * dimxy : -1;
    }
#pragma unroll
    for (int j = 0; j < kProbesPerLane; j++) p_w[j] = words[(p_cell[j] < 0 ? 0 : p_cell[j]) >> 5];
    double p_c[kProbesPerLane][3];
#pragma unroll
    for (int j = 0; j < kProbesPerLane; j++) {
      uint32_t bit = 1u << (p_cell[j] & 31);
      // points_per_voxel >= min_points_per_voxel_ and accepted eigenvalues (VoxelGrid.cpp:465)
      bool ok = p_cell[j] >= 0 && (p_w[j].y & bit);
      unsigned rank = ok ? p_w[j].z + __popc(p_w[j].x & (bit - 1u)) : 0u;
      hit_rank[j] = ((unsigned)p_pl[j]) | (rank << 6);
      if (ok) hit_mask |= 1u << j;
      const double *R = rec + (size_t)rank * kRecDoubles;
      p_c[j][0] = R[0]; p_c[j][1] = R[1]; p_c[j][2] = R[2];
    }
#pragma unroll
    for (int j = 0; j < kProbesPerLane; j++) {
      double cx = p_c[j][0] - (double)tp[p_pl[j]].tx, cy = p_c[j][1] - (double)tp[p_pl[j]].ty, cz = p_c[j][2] - (double)tp[p_pl[j]].tz;
      double d2 = cx * cx + cy * cy + cz * cz;
      if (!(d2 <= P.r2max)) hit_mask &= ~(1u << j);  // sqrt(d2) < radius (VoxelGrid.cpp:470-472)
    }
    // deterministic compaction: exclusive scan of the per-lane hit counts over the block
    unsigned cnt = __popc(hit_mask), incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlockD / 64; w++) {
      unsigned t = wave_tot[w];
      if (w < wave) base += t;
      total += t;
    }
    unsigned pos = base + incl - cnt;
#pragma unroll
    for (int j = 0; j < kProbesPerLane; j++)
      if (hit_mask & (1u << j)) queue[pos++] = hit_rank[j];
    __syncthreads();
    // ---- phase B: dense pair math
    for (unsigned qi = threadIdx.x; qi < total; qi += kBlockD) {
      unsigned e = queue[qi];
      const unsigned pl = e & 63u;
      const double *R = rec + (size_t)(e >> 6) * kRecDoubles;
      double pg[8], ph[15];
      const double *tm = terms + pl * kTermStride;
#pragma unroll
      for (int k = 0; k < 8; k++) pg[k] = tm[k];
      if (KIND != 1) {
#pragma unroll
        for (int k = 0; k < 15; k++) ph[k] = tm[8 + k];
      }
      pair_update<KIND>(P, R, tp[pl].tx, tp[pl].ty, tp[pl].tz, pg, ph, acc);
    }
    if (threadIdx.x == 0) pairs += total;
    __syncthreads();  // tp / terms / queue are rewritten by the next tile
  }
  // block reduction through LDS in a fixed order (bitwise reproducible): transpose the 28 per-lane accumulators,
  // 8 x 32 lanes add 32 rows each, then 32 lanes add the 8 partial rows
  double *red = reinterpret_cast<double *>(lds);
  double *red2 = reinterpret_cast<double *>(lds + (kLdsRed > kLdsTileEnd ? kLdsRed : kLdsTileEnd));
#pragma unroll
  for (int k = 0; k < 28; k++) red[threadIdx.x * kRedStride + k] = acc[k];
  __syncthreads();
  {
    int k = threadIdx.x & 31, part = threadIdx.x >> 5;
    double v = 0.0;
    if (k < 28) {
      const double *r = red + (size_t)(part * 32) * kRedStride + k;
#pragma unroll 8
      for (int c = 0; c < 32; c++) v += r[c * kRedStride];
    }
    red2[part * 32 + k] = v;
  }
  // slot 28 of every partial row carries the pair count (thread 0 holds it); rows 28..31 of red2 part 0 are unused sums
  __syncthreads();
  if (threadIdx.x == 0) red2[28] = (double)pairs;
  __syncthreads();
  if (threadIdx.x < kOutDoubles) {
    double v = 0.0;
    if (threadIdx.x < 28) {
#pragma unroll
      for (int part = 0; part < 8; part++) v += red2[part * 32 + threadIdx.x];
    } else if (threadIdx.x == 28) {
      v = red2[28];
    } else if (threadIdx.x == 29) {
      size_t mine = 0;  // points owned by this workgroup (tiles blockIdx.x, + gridDim.x, ...)
      for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) mine += min((size_t)kTilePts, n - tile * kTilePts);
      v = (double)mine;
    }
    partials[(size_t)blockIdx.x * kOutDoubles + threadIdx.x] = v;  // one 256-byte row per workgroup, written once
  }
}

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

void ndt_make_pass_params(const lsa_ndt *h, const float T[16], const double p[6], PassParams *P) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) P->T[r * 4 + c] = T[c * 4 + r];
  // computeAngleDerivatives (:523-645), including the |angle| < 10e-5 snap
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  double (*j)[3] = P->ja;
  j[0][0] = -sx * sz + cx * sy * cz; j[0][1] = -sx * cz - cx * sy * sz; j[0][2] = -cx * cy;
  j[1][0] = cx * sz + sx * sy * cz;  j[1][1] = cx * cz - sx * sy * sz;  j[1][2] = -sx * cy;
  j[2][0] = -sy * cz;                j[2][1] = sy * sz;                 j[2][2] = cy;
  j[3][0] = sx * cy * cz;            j[3][1] = -sx * cy * sz;           j[3][2] = sx * sy;
  j[4][0] = -cx * cy * cz;           j[4][1] = cx * cy * sz;            j[4][2] = -cx * sy;
  j[5][0] = -cy * sz;                j[5][1] = -cy * cz;                j[5][2] = 0;
  j[6][0] = cx * cz - sx * sy * sz;  j[6][1] = -cx * sz - sx * sy * cz; j[6][2] = 0;
  j[7][0] = sx * cz + cx * sy * sz;  j[7][1] = cx * sy * cz - sx * sz;  j[7][2] = 0;
  double (*q)[3] = P->ha;
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
  P->d1 = h->d1; P->d2 = h->d2;
  P->res = h->resolution;
  {  // exact squared-distance threshold: the largest double x with sqrt(x) < radius (sqrt is monotone)
    double r = (double)h->resolution, t = r * r;
    while (std::sqrt(t) >= r) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) < r) t = std::nextafter(t, INFINITY);
    P->r2max = t;
  }
  for (int a = 0; a < 3; a++) { P->min_b[a] = h->vi.map.min_b[a]; P->dim[a] = h->vi.map.dim[a]; }
}

unsigned ndt_pass_grid(const lsa_ndt *h, size_t n) {
  size_t blocks = (n + 63) / 64;  // one 64-point tile per workgroup until the chip is full, then grid-stride
  size_t cap = (size_t)h->ctx->num_cus * 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// enqueue one pass: partial kernel + second-stage reduction into h->d_out (device). No sync.
int ndt_enqueue_pass(lsa_ndt *h, const PassParams &P, int kind, bool reduce) {
  lsa_context *ctx = h->ctx;
  unsigned g = ndt_pass_grid(h, h->n_source);
  int rc;
  if ((rc = h->partials.reserve(ctx, (size_t)g * kOutDoubles, 1.0))) return rc;
  if ((rc = h->d_out.reserve(ctx, kOutDoubles))) return rc;
  const uint4 *words = h->words.p;
  const double *rec = h->rec.p;
  if (kind == 0) hipLaunchKernelGGL(k_ndt_pass<0>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, h->partials.p);
  else if (kind == 1) hipLaunchKernelGGL(k_ndt_pass<1>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, h->partials.p);
  else hipLaunchKernelGGL(k_ndt_pass<2>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, h->partials.p);
  if (reduce) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, h->partials.p, (int)g, h->d_out.p);
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

// Single-GPU, KITTI-sized passes (<= kHostReduceMaxBlocks workgroups): the workgroup partials are written straight into
// host-mapped pinned memory and the fixed-order second stage runs on the host, which removes a kernel launch and a copy
// kernel (~8 us) from every pass.  The summation order is the one k_reduce_partials uses, so both paths give the same bits.
constexpr unsigned kHostReduceMaxBlocks = 512;

bool ndt_pass_host_reduce(const lsa_ndt *h) { return !h->allreduce && ndt_pass_grid(h, h->n_source) <= kHostReduceMaxBlocks; }

int ndt_enqueue_pass_host(lsa_ndt *h, const PassParams &P, int kind, unsigned *grid_out) {
  lsa_context *ctx = h->ctx;
  unsigned g = ndt_pass_grid(h, h->n_source);
  int rc;
  if ((rc = h->h_partials.reserve(ctx, (size_t)kHostReduceMaxBlocks * kOutDoubles))) return rc;
  const uint4 *words = h->words.p;
  const double *rec = h->rec.p;
  double *dst = h->h_partials.p;  // pinned + mapped: the same pointer is valid on the device
  if (kind == 0) hipLaunchKernelGGL(k_ndt_pass<0>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, dst);
  else if (kind == 1) hipLaunchKernelGGL(k_ndt_pass<1>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, dst);
  else hipLaunchKernelGGL(k_ndt_pass<2>, dim3(g), dim3(kBlockD), 0, ctx->stream, h->d_source, h->n_source, P, words, rec, dst);
  LSA_HIP(ctx, hipGetLastError());
  *grid_out = g;
  return LSA_OK;
}

void ndt_reduce_partials_host(const double *partials, unsigned nb, double out32[32]) {
  // sm[b % 32][col] accumulates rows b, b + 32, ... in ascending b (the order k_reduce_partials uses); the rows are
  // read front to back once, because they sit in device-written pinned memory that is cold in the CPU caches
  double sm[32][32];
  for (int rl = 0; rl < 32; rl++)
    for (int col = 0; col < 32; col++) sm[rl][col] = 0.0;
  for (unsigned b = 0; b < nb; b++) {
    const double *row = partials + (size_t)b * kOutDoubles;
    double *dst = sm[b & 31];
    for (int col = 0; col < 32; col++) dst[col] += row[col];
  }
  for (int st = 16; st > 0; st >>= 1)
    for (int rl = 0; rl < st; rl++)
      for (int col = 0; col < 32; col++) sm[rl][col] += sm[rl + st][col];
  for (int col = 0; col < 32; col++) out32[col] = sm[0][col];
}
