// ndt_pass_dev.h — internal, device only: the body of one NDT derivative pass, shared by the one-pass kernel
// (ndt_derivatives.hip: k_ndt_pass) and the persistent whole-ScanMatch kernel (ndt_align.hip: k_ndt_align).
// See ndt_derivatives.hip for what it replaces in the reference and for the design notes.
#ifndef LSA_NDT_PASS_DEV_H_
#define LSA_NDT_PASS_DEV_H_

#include "ndt_internal.h"
#include "ndt_pass.h"

namespace {

constexpr int kBlockD = 256;
// Waves per SIMD requested from the register allocator = 256-lane workgroups resident per CU.  The pass is latency bound
// (three dependent L2 round trips per tile), so occupancy is throughput: 2 -> 3 waves bought ~20 %.  The full pass
// (28 f64 accumulators = 56 VGPRs live throughout) fits 3 waves (<= 168 VGPRs) without spilling, the gradient-only pass
// of the line search fits 4 (<= 128).
#ifndef LSA_PASS_WAVES
#define LSA_PASS_WAVES 3
#endif
#ifndef LSA_PASS_WAVES_GRAD
#define LSA_PASS_WAVES_GRAD 4
#endif
constexpr int pass_waves(int kind) { return kind == 1 ? LSA_PASS_WAVES_GRAD : LSA_PASS_WAVES; }
// Issue priority of a wave while it is on its way to its row loads (phase 0 and the probe of a tile); its pair math runs at 0.
// The waves of a SIMD all walk the same phases, and left to round-robin arbitration they drift into the same one — all probing
// (latency bound, the VALUs idle) or all in pair math (the next tile's loads not even issued).  Letting the wave that is about to
// wait for memory go first keeps them apart: 190.5 -> 178.8 us per pass at 2 M points, 184 -> 171 us on the wide map, nothing at
// KITTI size (one tile per wave).  Levels 1, 2, 3 measure the same; raising it in the pair math instead, or dropping it as soon as the
// loads are issued, gets half of it; a fixed priority per wave slot none (`profiles/r04_pass_setprio_ab.log`).  0 = off.
#ifndef LSA_PASS_SETPRIO
#define LSA_PASS_SETPRIO 1
#endif

// The scalar part of PassParams a pass works with.  A kernel's arguments are fetched by scalar loads at the point of first use,
// and right after a launch the scalar cache is cold: each such fetch is a trip to memory.  The pass kernels therefore copy
// these values into registers up front — the armed kernel BEFORE it starts waiting for its mail (pass_scalars_pin), so that
// nothing between the mail's arrival and the rows waits for a kernel argument (measured: 0.8 us in front of the transform).
struct PassScalars {
  double d1, d2, r2max;
  float res, inv_res, r2f_pre;
  int res_pow2;
  int min_b0, min_b1, min_b2;
  int dim0, dim1, dim2;
};
__device__ __forceinline__ PassScalars pass_scalars(const PassParams &P) {
  PassScalars q;
  q.d1 = P.d1; q.d2 = P.d2; q.r2max = P.r2max;
  q.res = P.res; q.inv_res = P.inv_res; q.r2f_pre = P.r2f_pre; q.res_pow2 = P.res_pow2;
  q.min_b0 = P.min_b[0]; q.min_b1 = P.min_b[1]; q.min_b2 = P.min_b[2];
  q.dim0 = P.dim[0]; q.dim1 = P.dim[1]; q.dim2 = P.dim[2];
  return q;
}
// forces every value into a scalar register HERE (and keeps the compiler from fetching it again later)
template <bool DENSE>
__device__ __forceinline__ void pass_scalars_pin(PassScalars &q) {
  if (!DENSE) return;   // (the bitmap-probe variant is short of scalar registers: pinned values would spill)
  asm volatile("" : "+s"(q.d1), "+s"(q.d2), "+s"(q.r2max), "+s"(q.res), "+s"(q.inv_res), "+s"(q.res_pow2), "+s"(q.r2f_pre));
  asm volatile("" : "+s"(q.min_b0), "+s"(q.min_b1), "+s"(q.min_b2), "+s"(q.dim0), "+s"(q.dim1), "+s"(q.dim2));
}

// exp(x) as the device library computes it — the same range reduction, the same eleven Horner steps with the same constants, hence the
// same bits (tests/test_gpu_parity.py pins the pass against the oracle's libm to 1e-11 either way) — with each coefficient brought
// into a SCALAR register pair right where it is used.  The library's form keeps every step as `coefficient += r * p`, which needs
// the coefficient in a fresh vector register per step and call: nine 64-bit moves per pair round of the pass, 35 vector instructions
// per exp in all; a three-address FMA reads the constant from scalar registers (two s_mov on the scalar unit, which has nothing else
// to do here) and needs none of the moves.  The range checks of the general function go too: ldexp saturates to 0 / inf by itself,
// and the argument here is -d2 q / 2 of a finite pair.  Same box, alternating processes (`profiles/r05_pass_exp_ab.log`): 2 M-point
// pass 172.9 -> 170.2 us, wide map 171 -> 162 us, armed KITTI-sized body 4.39 -> 4.30 us; every sum keeps its bits.
#ifndef LSA_PASS_EXP_SCALAR
#define LSA_PASS_EXP_SCALAR 1
#endif
__device__ __forceinline__ double lsa_sconst(unsigned long long bits) {
  double c = __longlong_as_double((long long)bits);
  // (without this the compiler goes back to the two-address form and its moves; with it every step is s_mov_b64 + v_fma_f64)
  asm volatile("" : "+s"(c));
  return c;
}
__device__ __forceinline__ double lsa_pass_exp(double x) {
#if LSA_PASS_EXP_SCALAR
#pragma clang fp contract(off)
  const double n = __builtin_rint(x * lsa_sconst(0x3ff71547652b82feull));
  double r = __builtin_fma(lsa_sconst(0xbfe62e42fefa39efull), n, x);
  r = __builtin_fma(lsa_sconst(0xbc7abc9e3b39803full), n, r);
  double p = __builtin_fma(lsa_sconst(0x3e5ade156a5dcb37ull), r, lsa_sconst(0x3e928af3fca7ab0cull));
  p = __builtin_fma(r, p, lsa_sconst(0x3ec71dee623fde64ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3efa01997c89e6b0ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3f2a01a014761f6eull));
  p = __builtin_fma(r, p, lsa_sconst(0x3f56c16c1852b7b0ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3f81111111122322ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3fa55555555502a1ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3fc5555555555511ull));
  p = __builtin_fma(r, p, lsa_sconst(0x3fe000000000000bull));
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  return ldexp(p, (int)n);
#else
  return exp(x);
#endif
}

template <int KIND>
__device__ __forceinline__ void pair_update(const PassScalars P, const double *__restrict__ R, float tx, float ty, float tz,
                                            const double pg[8], const double *ph /* LDS, read at the point of use */, double acc[28],
                                            bool live, unsigned &accepted /* wave-uniform */) {
  // R: centroid(3), C = icov (xx,xy,xz,yy,yz,zz), w.
  double d0 = (double)tx - R[0], d1 = (double)ty - R[1], d2 = (double)tz - R[2];  // x_trans - centroid (:434)
  // the exact radius test of radiusSearch (VoxelGrid.cpp:470-472), on the squared f64 distance against the host-computed
  // threshold.  The dense-table probe only pre-filters in f32 with a safety margin; the bitmap probe has already applied
  // this very test, for which it is a no-op.
  // (`live`: the lane holds a pair at all — the callers run every lane so that the count below is one scalar per wave)
  const bool in = live && (d0 * d0 + d1 * d1 + d2 * d2 <= P.r2max);
  accepted += (unsigned)__popcll(__ballot(in));
  if (!in) return;
  double C00 = R[3], C01 = R[4], C02 = R[5], C11 = R[6], C12 = R[7], C22 = R[8];
  double u0 = C00 * d0 + C01 * d1 + C02 * d2;
  double u1 = C01 * d0 + C11 * d1 + C12 * d2;
  double u2 = C02 * d0 + C12 * d1 + C22 * d2;
  double q = d0 * u0 + d1 * u1 + d2 * u2;
  double e = lsa_pass_exp(-P.d2 * q / 2);  // :494
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
    const double nd2 = -P.d2;
    // H_ij += f * (-d2 (u.J_i)(u.J_j) + u.H_E(i,j) + J_j^T C J_i)   (:512-514).  Written column group by column group so
    // that each C J_j (3 values) dies as soon as its column is done: the pair math is register-bound (28 f64
    // accumulators stay live), and every register saved here buys occupancy.
    // translation x translation: J_j^T C J_i = C_ij
    acc[7] += f * (nd2 * u0 * u0 + C00);
    acc[8] += f * (nd2 * u0 * u1 + C01);
    acc[9] += f * (nd2 * u0 * u2 + C02);
    acc[13] += f * (nd2 * u1 * u1 + C11);
    acc[14] += f * (nd2 * u1 * u2 + C12);
    acc[18] += f * (nd2 * u2 * u2 + C22);
    // H_E blocks (:465-480): (3,3)=(0,ph0,ph1) (3,4)=(0,ph2,ph3) (3,5)=(0,ph4,ph5) (4,4)=(ph6..8) (4,5)=(ph9..11) (5,5)=(ph12..14)
    {  // column 3: a = C J_3, J_3 = (0, pg0, pg1)
      double ax = C01 * pg[0] + C02 * pg[1], ay = C11 * pg[0] + C12 * pg[1], az = C12 * pg[0] + C22 * pg[1];
      acc[10] += f * (nd2 * u0 * uJ3 + ax);
      acc[15] += f * (nd2 * u1 * uJ3 + ay);
      acc[19] += f * (nd2 * u2 * uJ3 + az);
      acc[22] += f * (nd2 * uJ3 * uJ3 + (u1 * ph[0] + u2 * ph[1]) + (pg[0] * ay + pg[1] * az));
      acc[23] += f * (nd2 * uJ3 * uJ4 + (u1 * ph[2] + u2 * ph[3]) + (pg[2] * ax + pg[3] * ay + pg[4] * az));
      acc[24] += f * (nd2 * uJ3 * uJ5 + (u1 * ph[4] + u2 * ph[5]) + (pg[5] * ax + pg[6] * ay + pg[7] * az));
    }
    {  // column 4: a = C J_4, J_4 = (pg2, pg3, pg4)
      double ax = C00 * pg[2] + C01 * pg[3] + C02 * pg[4], ay = C01 * pg[2] + C11 * pg[3] + C12 * pg[4], az = C02 * pg[2] + C12 * pg[3] + C22 * pg[4];
      acc[11] += f * (nd2 * u0 * uJ4 + ax);
      acc[16] += f * (nd2 * u1 * uJ4 + ay);
      acc[20] += f * (nd2 * u2 * uJ4 + az);
      acc[25] += f * (nd2 * uJ4 * uJ4 + (u0 * ph[6] + u1 * ph[7] + u2 * ph[8]) + (pg[2] * ax + pg[3] * ay + pg[4] * az));
      acc[26] += f * (nd2 * uJ4 * uJ5 + (u0 * ph[9] + u1 * ph[10] + u2 * ph[11]) + (pg[5] * ax + pg[6] * ay + pg[7] * az));
    }
    {  // column 5: a = C J_5, J_5 = (pg5, pg6, pg7)
      double ax = C00 * pg[5] + C01 * pg[6] + C02 * pg[7], ay = C01 * pg[5] + C11 * pg[6] + C12 * pg[7], az = C02 * pg[5] + C12 * pg[6] + C22 * pg[7];
      acc[12] += f * (nd2 * u0 * uJ5 + ax);
      acc[17] += f * (nd2 * u1 * uJ5 + ay);
      acc[21] += f * (nd2 * u2 * uJ5 + az);
      acc[27] += f * (nd2 * uJ5 * uJ5 + (u0 * ph[12] + u1 * ph[13] + u2 * ph[14]) + (pg[5] * ax + pg[6] * ay + pg[7] * az));
    }
  }
}

// Each WAVE of a 256-lane workgroup walks its own tiles of 14 source points, in four phases per tile, without ever
// waiting for the other three waves (the only workgroup barrier is the one before the final reduction):
//   0. 4 lanes per point: one transforms it (f32, PCL order) and publishes the clipped 3x3x3 cell box, and each of the
//      four computes its share of the 23 point-derivative dot products (8 Jacobian + 15 Hessian terms,
//      computePointDerivatives) into LDS -- once per POINT, not once per pair as the reference does;
//   A1. ROW probes: the <= 3 cells of one (y, z) row of a box are consecutive linear indices, i.e. adjacent bits of one
//      (rarely two) 16-B {occ, valid, prefix} words.  14 points x 9 rows = 126 row probes (<= 2 per lane) read the word
//      from L2, mask the valid bits and emit one CANDIDATE {point, record rank} per valid cell -- integer work only;
//   A2. candidates (valid voxels inside the box, ~10 per point) load the 24-B centroid and take the radius test on the
//      squared distance against a host-computed exact threshold (no f64 sqrt); survivors (Kbar ~ 2.5-5 per point) become
//      PAIRS.  Both compactions place their output at offsets from wave-wide exclusive scans (no atomics), so the queue
//      order -- and with it every f64 sum -- is reproducible run to run;
//   B. the wave takes FULL rounds of 64 pairs off the queue (what is left joins the next tile's pairs): one (point,
//      voxel) pair per lane per step, dense f64 pair math into 28 registers.
// Tile size: 14 points = two probe rounds (16 points needed a third, three quarters empty) and 50 pairs per tile on
// average at cfg5, which the carry-over packs into full rounds: 244 -> 216 us per pass against 16-point tiles drained
// tile by tile.
// Separating the cheap integer probe from the centroid test and that from the expensive pair math keeps the f64 pipes on
// useful work (the reference computes a distance for all 27 cells of every point).  All hand-offs inside a tile go
// through the wave's private LDS slice and need only wave-level ordering (LDS is in order per wave), so the 8 waves
// resident on a CU drift apart and hide each other's L2 latency; a workgroup-wide formulation of the same phases (6
// s_barrier per 64-point tile) left the VALUs idle half of the time.  A KITTI-sized pass (7-25 k points) spreads over
// >= 470 waves.  The 69 angle-derivative constants are touched only in phase 0, which keeps them out of the pair loop's
// scalar registers.
constexpr int kWaves = kBlockD / 64;
constexpr int kTilePts = 14;                                             // points per wave tile, 4 lanes per point in phase 0 (56 lanes)
constexpr int kTileSlots = 16;                                           // LDS slots per buffer half (power of two)
constexpr int kRowsPerPoint = 9;
constexpr int kRowProbes = kTilePts * kRowsPerPoint;                     // 126
constexpr int kRowRounds = (kRowProbes + 63) / 64;                       // 3
constexpr int kQueueCap = kTilePts * 27;                                 // 378 candidates / pairs at most
constexpr int kCandPerLane = 4;                                          // A2 candidates per lane per round
constexpr int kPointTerms = 23;   // pg[8] + ph[15]
constexpr int kTermStride = 24;
constexpr int kRedStride = 29;    // 28 accumulators per lane, padded: conflict-free LDS transpose
constexpr int kRedLanes = 128;    // the transpose runs in two rounds of 128 lanes to halve its LDS footprint

struct TilePoint {
  float tx, ty, tz;  // transformed point
  int cell0;         // linear index of the box's low corner
  int nxyz;          // per axis one byte, x | y << 8 | z << 16: box extent (bits 0-1) and cells clipped off in front (bits 2-3); 0 = point contributes nothing
  float rx, ry, rz;  // transformed point relative to the low corner of the box's first cell (f32; dense-table pre-filter)
};

// per-wave LDS slice
// The point data of a tile (TilePoint + its 23 terms) is double-buffered (nb = 2) in the passes that carry pairs over
// from one tile to the next (see phase B): the pairs a tile leaves after its last FULL round of 64 join the next tile's
// first round, whose phase 0 writes the other buffer.  The gradient-only pass of the line search (KIND 1) keeps one buffer:
// its pair math is cheap and the 14 KB it saves per workgroup is what lets a fourth workgroup share the CU.
constexpr int kLdsTp = 0;
constexpr int kQueueSlots = kQueueCap + 64;                                              // a tile's pairs behind < 64 carried ones
constexpr int lds_terms(int nb) { return nb * 512; }                                     // after nb x 16 * 32 B of TilePoint
constexpr int lds_cand(int nb) { return lds_terms(nb) + nb * kTileSlots * kTermStride * 8; }  // + nb x 3072
constexpr int lds_queue(int nb) { return lds_cand(nb) + kQueueCap * 4; }                 // + 1728
constexpr int lds_slice(int nb) { return ((lds_queue(nb) + (kQueueSlots + 1) * 4 + 255) / 256) * 256; }  // 7424 / 11008 (queue + one dump slot)
constexpr int lds_tiles(int nb) { return lds_slice(nb) * kWaves; }                       // 29696 / 44032
constexpr int kLdsRed = kRedLanes * kRedStride * 8;                                      // 29696 (aliases the slices after the loops)
constexpr int lds_red2(int nb) { return kLdsRed > lds_tiles(nb) ? kLdsRed : lds_tiles(nb); }
constexpr int lds_const(int nb) { return lds_red2(nb) + (8 * 32 + 2 * kWaves) * 8; }     // + second-stage [8][32] + per-wave counters
constexpr int lds_tl(int nb) { return lds_const(nb) + 72 * 8; }                          // + the 69 angle-derivative constants
constexpr int lds_bytes(int nb) { return lds_tl(nb) + 16 * 4; }                          // + the 3x4 rigid transform
constexpr int pass_bufs(int kind) { return kind == 1 ? 1 : 2; }
static_assert(kTilePts <= kTileSlots && kTileSlots * (int)sizeof(TilePoint) <= 512, "tile points overflow their LDS slot");

// wave-level ordering point for LDS hand-offs between lanes of ONE wave: LDS operations of a wave execute in order, so it
// is enough to stop the compiler from moving accesses across it (this is what a workgroup barrier of a single-wave
// workgroup is lowered to)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive scan of one count per lane over the wave; returns this lane's offset, *total = wave sum
__device__ __forceinline__ unsigned wave_excl_scan(unsigned cnt, int lane, unsigned *total) {
  unsigned incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  *total = __shfl(incl, 63, 64);
  return incl - cnt;
}

// What a pass can prepare before it knows its transform: the zeroed accumulators and, for the dense probe, which (point, dy, dz)
// a lane probes in each round — {point slot : 6, -, dy : 2, dz : 2, -, live : 1}, 16 bits per round, a function of the lane
// only.  The armed kernel builds it BEFORE it starts waiting for its mail (pass_pre_pin keeps the compiler from sinking the
// work behind the wait): ~90 register moves and a few integer divisions less between the mail and the rows.
struct PassPre {
  double acc[28];
  unsigned probe_lut;
};
template <bool DENSE>
__device__ __forceinline__ void pass_pre_init(PassPre &pre) {
#pragma unroll
  for (int k = 0; k < 28; k++) pre.acc[k] = 0.0;
  const int lane = threadIdx.x & 63;
  unsigned probe_lut = 0;
  if (DENSE) {
#pragma unroll
    for (int j = 0; j < kRowRounds; j++) {
      const int q = lane + j * 64;
      if (q < kRowProbes) {  // (the last round has a few idle lanes: all-zero entry, live bit clear)
#ifdef LSA_PROBE_POINT_MAJOR
        const int pl = q / kRowsPerPoint;
        const int o = q - pl * kRowsPerPoint;
#else
        // ROW-major: adjacent lanes probe the same (dy, dz) row of CONSECUTIVE points.  Consecutive points of a spatially coherent
        // source share their cell box more often than not, so the lanes of a quad ask for the same 16 bytes and the texture addresser
        // sends the L1 one access instead of four — and their survivors sit in adjacent queue slots, so the pair step's record loads
        // coalesce the same way (point-major, a point's nine rows side by side, every lane of a load went to a line of its own:
        // 46 L1 accesses per load instruction at 2 M points, `profiles/r04_cfg5_pass_l1_pmc.csv`)
        const int o = q / kTilePts;
        const int pl = q - o * kTilePts;
#endif
        const int dz = o / 3, dy = o - dz * 3;
        probe_lut |= ((unsigned)pl | ((unsigned)dy << 8) | ((unsigned)dz << 10) | 0x8000u) << (16 * j);
      }
    }
  }
  pre.probe_lut = probe_lut;
}
__device__ __forceinline__ void pass_pre_pin(PassPre &pre) {
#pragma unroll
  for (int k = 0; k < 28; k++) asm volatile("" : "+v"(pre.acc[k]));
  asm volatile("" : "+v"(pre.probe_lut));
}

// The three cells of one loaded row: pre-filter on the f32 distance, survivors to the pair queue in (cell, lane) order — one wave
// ballot per cell gives every keeping lane its slot (count of keeping lanes below it) and the wave the running total as a SCALAR.
// SHIFTED: some lane's row lost cells in front to the table's box, so entry ii of its load is cell ii + (cells lost) of the cube.
template <bool SHIFTED>
__device__ __forceinline__ void probe_cells(const float4 e0, const float4 e1, const float4 e2, float rx, float ry, float rz, int nx /* extent | lost << 2 */,
                                            unsigned slot, float res, float r2f_pre, unsigned *queue, unsigned &total) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int ii = SHIFTED ? i - (nx >> 2) : i;
    const float4 en = SHIFTED ? (ii == 0 ? e0 : (ii == 1 ? e1 : e2)) : (i == 0 ? e0 : (i == 1 ? e1 : e2));
    float dx = (rx - (float)i * res) - en.x, dy = ry - en.y, dz = rz - en.z;
    float d2 = dx * dx + dy * dy + dz * dz;
    const int rank = __float_as_int(en.w);  // -1: unoccupied, too few points or rejected eigenvalues (VoxelGrid.cpp:465)
    const bool ok = (unsigned)ii < (unsigned)(nx & 3) && rank >= 0 && d2 <= r2f_pre;
    const unsigned long long m = __ballot(ok);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (ok) queue[total + below] = slot | ((unsigned)rank << 6);  // point slot incl. its buffer half
    total += (unsigned)__popcll(m);
  }
}

// DENSE: the probe reads a dense per-cell table {centroid - cell corner (f32 x 3), record rank or -1} (16 B per cell,
// built when the cell box is small enough, ndt_grid.hip) instead of the bitmap words: one 48-B row load answers "valid?",
// "which record?" and "roughly how far?" for 3 cells, so the candidate stage (A2: a dependent centroid load from the
// record table + a second compaction) disappears; the f32 distance is only a pre-filter with a safety margin, the exact
// f64 test runs in the pair step on the record's centroid it loads anyway.  ~20 % fewer instructions per tile and one
// L2 round trip less on the critical path.  The bitmap path stays for cell boxes too large for a dense table.
template <int KIND, bool DENSE>
__device__ __forceinline__ double ndt_pass_body(const float4 *__restrict__ src, size_t n, const PassScalars P, const float *Tl /* LDS: 3x4 row-major */,
                                                const double *jh /* LDS: ja[8][3] then ha[15][3] */, const uint4 *__restrict__ words,
                                                const float4 *__restrict__ dense, const double *__restrict__ rec, unsigned char *lds,
                                                unsigned vgrid /* workgroups sharing the tiles: gridDim.x, or fewer for an armed launch */,
                                                PassPre &pre, bool have_first = false /* wave-uniform */,
                                                float4 s_first = make_float4(0.f, 0.f, 0.f, 0.f), float4 *__restrict__ keep = nullptr) {
  constexpr int NB = pass_bufs(KIND);
  constexpr int kLdsSlice = lds_slice(NB), kLdsTerms = lds_terms(NB), kLdsCand = lds_cand(NB), kLdsQueue = lds_queue(NB), kLdsRed2 = lds_red2(NB);
  // (the wave index is uniform across the wave: as a scalar, everything derived from it — the LDS slice, the tile walk — stays
  // out of the vector registers)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char *slice = lds + wave * kLdsSlice;
  TilePoint *tp = reinterpret_cast<TilePoint *>(slice + kLdsTp);
  double *terms = reinterpret_cast<double *>(slice + kLdsTerms);
  unsigned *cand = reinterpret_cast<unsigned *>(slice + kLdsCand);
  unsigned *queue = reinterpret_cast<unsigned *>(slice + kLdsQueue);
  double (&acc)[28] = pre.acc;
  unsigned accepted = 0;        // (point, voxel) pairs this WAVE accepted (wave-uniform: a scalar register)
  // phase 0: the 4 lanes of a QUAD share a point (quad 14 and 15 = lanes 56..63 are spare: they work on tile slots 14 / 15,
  // which exist in LDS and which nothing reads)
  const int part4 = lane & 3, ppl = lane >> 2;
  const int dimx = P.dim0, dimxy = P.dim0 * P.dim1;
  const size_t ntiles = (n + kTilePts - 1) / kTilePts;
  const size_t tile0 = (size_t)blockIdx.x * kWaves + wave, tstride = (size_t)vgrid * kWaves;
  // jh = ja[8][3] followed by ha[15][3]: 23 constant 3-vectors.  Phase 0 indexes them per LANE (each of the 4 lanes of a
  // point owns 6 of the 23 terms), which kernel arguments cannot do without turning into per-lane global loads: the
  // caller has staged them (and the rigid transform) in LDS.
  // software prefetch of the next tile's points (each of the 4 lanes of a point holds its own copy)
  float4 s_next = make_float4(0.f, 0.f, 0.f, 0.f);
  // (an armed launch may already hold its first tile's points: it took them from the copy the previous pass of the same
  // ScanMatch kept, before the parameters of this pass had even arrived)
  if (have_first) {
    s_next = s_first;
  } else {
    size_t i0 = tile0 * kTilePts + ppl;
    if (ppl < kTilePts && i0 < n) s_next = src[i0];
  }
  const unsigned probe_lut = pre.probe_lut;
  unsigned carried = 0;  // pairs at the front of the queue left over by the previous tile (< 64, wave-uniform)
  int pbase = 0;         // which half of tp / terms this tile uses: 0 or kTileSlots
  for (size_t tile = tile0; tile < ntiles; tile += tstride, pbase ^= (NB == 2 ? kTileSlots : 0)) {
    // ---- phase 0
    if (LSA_PASS_SETPRIO) __builtin_amdgcn_s_setprio(LSA_PASS_SETPRIO);
    {
      size_t i = tile * kTilePts + ppl;
      float4 s = s_next;
      size_t inext = (tile + tstride) * kTilePts + ppl;
      if (ppl < kTilePts && inext < n) s_next = src[inext];
      if (keep != nullptr && tile == tile0 && part4 == 3 && ppl < kTilePts && i < n) keep[i] = s;  // for the next pass's early start
      {
        // One AXIS per lane: lane group g < 3 of a point transforms coordinate g (f32, PCL order) and clips that axis of the
        // radiusSearch cell box (VoxelGrid.cpp:438-455, relative to real_min, clipped to the occupied bounds); the three
        // axes are independent until the very end, where the quad's first lane gathers the box from its neighbours (DPP
        // quad broadcasts, no LDS).  (The whole transform + box ran in 14 of the wave's 64 lanes before: ~100 issue slots
        // per tile, now ~45.)
        int a = part4 < 3 ? part4 : 0;
        asm volatile("" : "+v"(a));  // lane-constant: do not hoist its selects out of the tile loop (registers)
        const int mb = a == 0 ? P.min_b0 : (a == 1 ? P.min_b1 : P.min_b2);
        const int dmax = (a == 0 ? P.dim0 : (a == 1 ? P.dim1 : P.dim2)) - 1;
        const int stride = a == 0 ? 1 : (a == 1 ? dimx : dimxy);
        const float *Ta = Tl + 4 * a;
        // pcl::transformPointCloud: x' = m00*x + m01*y + m02*z + m03, f32, left to right, no FMA
        const float t = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(Ta[0], s.x), lsa_fmul(Ta[1], s.y)), lsa_fmul(Ta[2], s.z)), Ta[3]);
        const float r = P.res;
        int h, l;
        if (P.res_pow2) {
          // a power-of-two leaf (the YAML default 1.0): x / r == x * (1 / r) exactly, so the correctly rounded f32
          // divisions (~12 instructions each) collapse to multiplications without changing a single bit
          const float ir = P.inv_res;
          h = (int)floorf(lsa_fmul(lsa_fadd(t, r), ir)) - mb; l = (int)floorf(lsa_fmul(lsa_fsub(t, r), ir)) - mb;
        } else {
          h = (int)floorf(lsa_fdiv(lsa_fadd(t, r), r)) - mb; l = (int)floorf(lsa_fdiv(lsa_fsub(t, r), r)) - mb;
        }
        h = min(h, dmax);
        const int l0 = l;                  // the cube's first cell on this axis BEFORE it is clipped to the table's box
        l = max(l, 0);
        const int na = min(h - l + 1, 3);  // with radius == leaf the box is at most 3 cells wide per axis
        const bool ok = ppl < kTilePts && i < n && part4 < 3 && (__float_as_uint(t) & 0x7f800000u) != 0x7f800000u && na > 0;
        // extent of this axis (2 bits) + how many cells the clip cut off in front (2 bits: <= 2 where any cell is left) + "axis fine" bit
        unsigned w = ok ? ((((unsigned)na) | ((unsigned)(l - l0) << 2)) << (8 * a)) | (1u << (24 + a)) : 0u;
        int lin = ok ? l * stride : 0;
        // Relative to the low corner of the UNCLIPPED cube's first cell (dense pre-filter): the f32 distances of the pre-filter —
        // and with them which borderline candidates take a slot of the pair queue, i.e. which lane adds which pair — must not depend
        // on where the table's box happens to end (a full build's bounding box and the incremental map's window differ there; the
        // same cell measured from two different corners rounds differently once in ~10^7 candidates).
        const float rr = t - (float)(l0 + mb) * r;
        // quad_perm broadcasts of lane 1 (0x55) and lane 2 (0xaa) of each quad
        w += (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0x55, 0xf, 0xf, true) + (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xaa, 0xf, 0xf, true);
        lin += __builtin_amdgcn_mov_dpp(lin, 0x55, 0xf, 0xf, true) + __builtin_amdgcn_mov_dpp(lin, 0xaa, 0xf, 0xf, true);
        float *tpw = reinterpret_cast<float *>(tp + pbase + ppl);
        if (part4 < 3) {
          tpw[a] = t;        // TilePoint::tx / ty / tz
          tpw[5 + a] = rr;   // TilePoint::rx / ry / rz (read only behind a valid box)
        }
        if (part4 == 0) {
          const bool valid = (w >> 24) == 7u;  // all three coordinates finite, box not empty on any axis
          tp[pbase + ppl].cell0 = valid ? lin : 0;
          tp[pbase + ppl].nxyz = valid ? (int)(w & 0xffffffu) : 0;
        }
      }
      // computePointDerivatives (:448-482): dot products of the source point with the angle-derivative vectors;
      // lane group g of a point owns terms 6g .. 6g+5 (KIND 1 needs only the 8 Jacobian terms)
      double x = s.x, y = s.y, z = s.z;
      // (a fixed trip count: the six terms' 18 constants are fetched from LDS together instead of term by term)
      const int k0 = part4 * 6;
      constexpr int kLast = (KIND == 1) ? 8 : kPointTerms;
      double tv[6];
#pragma unroll
      for (int u = 0; u < 6; u++) {
        const int k = min(k0 + u, kLast - 1);
        tv[u] = x * jh[k * 3] + y * jh[k * 3 + 1] + z * jh[k * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 6; u++)
        if (k0 + u < kLast) terms[(pbase + ppl) * kTermStride + k0 + u] = tv[u];
    }
    wave_sync();
    unsigned total = carried;  // the queue already holds the previous tile's left-over pairs
    if constexpr (DENSE) {
      // ---- phase A (dense table): row probes -> pairs, pre-filtered on the f32 distance
      // which (point, dy, dz) a lane probes in each round depends on the lane only: packed once per kernel into ONE register
      // (probe_lut: 8 bits per field, 16 bits per round) instead of ~12 integer instructions per round and tile — and instead
      // of the ~30 registers the compiler would keep live through the pair math if it hoisted the unpacked values
      unsigned lut_v = probe_lut;  // opaque copy: the unpacked fields must not be hoisted out of the tile loop (registers)
      asm volatile("" : "+v"(lut_v));
      int r_cell[kRowRounds], r_nx[kRowRounds], r_pl[kRowRounds];
      float r_y[kRowRounds], r_z[kRowRounds];
      float4 e0[kRowRounds], e1[kRowRounds], e2[kRowRounds];
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        const unsigned lut = lut_v >> (16 * j);
        const int pl = lut & 0x3f, dy = (lut >> 8) & 3, dz = (lut >> 10) & 3;
        const bool live = (lut >> 15) & 1;
        // (dy, dz) — and i below — count from the UNCLIPPED cube's first cell: which lane probes which row, and in which step a
        // cell's survivors enter the queue, must not depend on where the table's box ends either (the queue order decides which
        // lane adds which pair, i.e. the last bits of every sum).  The row is loaded from the clipped box.
        int nxyz = tp[pbase + pl].nxyz;   // per axis: extent | cells clipped off in front << 2
        const int iy = dy - ((nxyz >> 10) & 3), iz = dz - ((nxyz >> 18) & 3);   // the row's place inside the clipped box
        bool in_box = live && (unsigned)iy < (unsigned)((nxyz >> 8) & 3) && (unsigned)iz < (unsigned)((nxyz >> 16) & 3);
        r_pl[j] = pl;
        r_nx[j] = in_box ? (nxyz & 0xf) : 0;
        r_cell[j] = in_box ? tp[pbase + pl].cell0 + __mul24(iy, dimx) + ((iz & 1) ? dimxy : 0) + ((iz & 2) ? 2 * dimxy : 0) : 0;
        r_y[j] = tp[pbase + pl].ry - (float)dy * P.res;
        r_z[j] = tp[pbase + pl].rz - (float)dz * P.res;
      }
      // all row loads first so the L2 round trips overlap (the table is padded by two cells past the last row)
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        const float4 *row = dense + r_cell[j];
        e0[j] = row[0]; e1[j] = row[1]; e2[j] = row[2];
      }
      // The survivors go to the queue in (round, cell, lane) order: one wave ballot per (round, cell) gives every keeping
      // lane its slot (count of keeping lanes below it) and the wave the running total as a SCALAR — no cross-lane scan,
      // no stores for rejected entries.  The order is a function of the data only, so every f64 sum stays reproducible.
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        const float rx = tp[pbase + r_pl[j]].rx;
        // Entry ii of the loaded (clipped) row is cell i = ii + (cells clipped off in front) of the unclipped cube.  Nothing is clipped
        // unless the point sits within a cell of the table's low faces — for a whole wave almost never: then entry i IS cell i and the
        // eight selects per cell that pick the entry are skipped (a wave-uniform branch; 48 of the ~480 VALU instructions of a tile).
        const bool clipped = __ballot((r_nx[j] >> 2) != 0) != 0ull;
        if (clipped) probe_cells<true>(e0[j], e1[j], e2[j], rx, r_y[j], r_z[j], r_nx[j], (unsigned)(pbase + r_pl[j]), P.res, P.r2f_pre, queue, total);
        else probe_cells<false>(e0[j], e1[j], e2[j], rx, r_y[j], r_z[j], r_nx[j], (unsigned)(pbase + r_pl[j]), P.res, P.r2f_pre, queue, total);
      }
      wave_sync();
    } else {
    // ---- phase A1: row probes -> candidates (valid voxels inside the box)
    unsigned c_ent[kRowRounds * 3];
    unsigned c_mask = 0;
    // The probe -> (point, row) mapping depends on the lane only, so the compiler would hoist ~30 registers of index
    // arithmetic out of the tile loop and keep them live through the pair math, which is register-bound; an opaque copy
    // of the lane id makes it recompute them per tile (a few dozen integer instructions) instead.
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    {
      int r_cell[kRowRounds], r_nx[kRowRounds], r_pl[kRowRounds];
      uint4 w0[kRowRounds], w1[kRowRounds];
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        int q = lane_v + j * 64;
        int pl = (q < kRowProbes) ? q / kRowsPerPoint : 0;
        int o = q - pl * kRowsPerPoint;
        int dz = o / 3, dy = o - dz * 3;
        int nxyz = tp[pbase + pl].nxyz;
        bool in_box = q < kRowProbes && dy < ((nxyz >> 8) & 3) && dz < ((nxyz >> 16) & 3);
        r_pl[j] = pl;
        r_nx[j] = in_box ? (nxyz & 3) : 0;
        r_cell[j] = in_box ? tp[pbase + pl].cell0 + dy * dimx + dz * dimxy : 0;
      }
      // all word loads first so the L2 round trips overlap; the second word is needed only when the row straddles a
      // 32-cell boundary (out-of-box probes read word 0 and are masked by nx == 0)
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        int wi = r_cell[j] >> 5;
        w0[j] = words[wi];
        w1[j] = ((r_cell[j] & 31) + r_nx[j] > 32) ? words[wi + 1] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int j = 0; j < kRowRounds; j++) {
        // the row's bits as one 64-bit window over {word, next word}: prefix(next) = prefix(word) + popc(occ(word)), so
        // rank(cell) = prefix(word) + popc(occ64 below the cell) holds on either side of the 32-cell boundary
        const int sh = r_cell[j] & 31;
        const unsigned long long occ = ((unsigned long long)w1[j].x << 32) | w0[j].x;
        const unsigned long long val = (((unsigned long long)w1[j].y << 32) | w0[j].y) >> sh;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          // points_per_voxel >= min_points_per_voxel_ and accepted eigenvalues (VoxelGrid.cpp:465)
          bool ok = i < r_nx[j] && ((val >> i) & 1ull);
          unsigned rank = w0[j].z + (unsigned)__popcll(occ & ((1ull << (sh + i)) - 1ull));
          c_ent[j * 3 + i] = ((unsigned)(pbase + r_pl[j])) | (rank << 6);
          if (ok) c_mask |= 1u << (j * 3 + i);
        }
      }
    }
    unsigned total_c;
    {
      unsigned pos = wave_excl_scan(__popc(c_mask), lane, &total_c);
#pragma unroll
      for (int k = 0; k < kRowRounds * 3; k++)
        if (c_mask & (1u << k)) cand[pos++] = c_ent[k];
    }
    wave_sync();
    // ---- phase A2: candidates -> pairs (radius test on the centroid distance).  A lane owns a contiguous chunk of the
    // candidate list, so one scan per round keeps the order; a round covers 64 x 4 candidates (a tile has ~160, the worst
    // case 432 takes two rounds)
    for (unsigned cbase = 0; cbase < total_c; cbase += 64 * kCandPerLane) {
      const unsigned left = min(total_c - cbase, (unsigned)(64 * kCandPerLane));
      const unsigned per = (left + 63) / 64;  // <= kCandPerLane
      const unsigned c0 = cbase + lane * per, c_end = cbase + left;
      unsigned p_ent[kCandPerLane];
      unsigned p_mask = 0;
      double pc[kCandPerLane][3];
#pragma unroll
      for (int k = 0; k < kCandPerLane; k++) {
        unsigned ci = c0 + k;
        bool live = (unsigned)k < per && ci < c_end;
        unsigned e = live ? cand[ci] : 0u;
        p_ent[k] = e;
        if (live) p_mask |= 1u << k;
        const double *R = reinterpret_cast<const double *>(reinterpret_cast<const char *>(rec) + (e >> 6) * (unsigned)(kRecDoubles * sizeof(double)));
        if (live) { pc[k][0] = R[0]; pc[k][1] = R[1]; pc[k][2] = R[2]; }
        else { pc[k][0] = pc[k][1] = pc[k][2] = 0.0; }
      }
#pragma unroll
      for (int k = 0; k < kCandPerLane; k++) {
        const TilePoint &t = tp[p_ent[k] & 63u];
        double cx = pc[k][0] - (double)t.tx, cy = pc[k][1] - (double)t.ty, cz = pc[k][2] - (double)t.tz;
        double d2 = cx * cx + cy * cy + cz * cz;
        if (!(d2 <= P.r2max)) p_mask &= ~(1u << k);  // sqrt(d2) < radius (VoxelGrid.cpp:470-472)
      }
      unsigned round_total;
      unsigned pos = total + wave_excl_scan(__popc(p_mask), lane, &round_total);
#pragma unroll
      for (int k = 0; k < kCandPerLane; k++)
        if (p_mask & (1u << k)) queue[pos++] = p_ent[k];
      total += round_total;
    }
    wave_sync();
    }
    // ---- phase B: dense pair math, in FULL rounds of 64 pairs.  A tile leaves 57 pairs on average (cfg5), so one round
    // per tile ran at 76 % lane use and every fifth tile paid a second, nearly empty one; instead the pairs beyond the last
    // full round stay at the front of the queue and join the next tile's (their point data sits in the other buffer
    // half, which that tile does not touch).  Pairs are never carried twice: the previous tile's are at the front and go
    // out with the first round, and if even carried + fresh stays below 64 while something was carried, the round is run
    // partly filled.
    {
      if (LSA_PASS_SETPRIO) __builtin_amdgcn_s_setprio(0);
      total = (unsigned)__builtin_amdgcn_readfirstlane((int)total);  // wave-uniform by construction: keep the round bookkeeping scalar
      const unsigned full = total >> 6;
      unsigned left = total & 63u;
      // (single-buffered pass: nothing may be carried, every tile is drained)
      const bool drain = NB == 1 ? left != 0 : (full == 0 && carried != 0);  // < 64 in all, part of it a tile old: run it now
      const unsigned rounds = drain ? full + 1u : full;
      for (unsigned r = 0; r < rounds; r++) {
        const unsigned qi = r * 64 + lane;
        const bool live = qi < total;          // (only the last round of a drained tile has idle lanes: they read entry 0)
        const unsigned e = live ? queue[qi] : 0u;
        const unsigned pl = e & 63u;
        const double *R = reinterpret_cast<const double *>(reinterpret_cast<const char *>(rec) + (e >> 6) * (unsigned)(kRecDoubles * sizeof(double)));
        double pg[8];
        const double *tm = terms + pl * kTermStride;
#pragma unroll
        for (int k = 0; k < 8; k++) pg[k] = tm[k];
        // the 15 Hessian terms stay in LDS and are read where they are used: 30 fewer live registers in the pair math
        pair_update<KIND>(P, R, tp[pl].tx, tp[pl].ty, tp[pl].tz, pg, tm + 8, acc, live, accepted);
      }
      if (drain) left = 0;
      if (full != 0 && left != 0) {  // move the left-over pairs (all of THIS tile) to the front of the queue
        unsigned e = lane < left ? queue[full * 64 + lane] : 0u;
        wave_sync();
        if (lane < left) queue[lane] = e;
      }
      carried = left;
    }
    wave_sync();  // the other half of tp / terms and the queue behind the carried pairs are rewritten by the next tile
  }
  // the last tile's left-over pairs
  if (carried) {
    const bool live = (unsigned)lane < carried;
    const unsigned e = live ? queue[lane] : 0u;
    const unsigned pl = e & 63u;
    const double *R = reinterpret_cast<const double *>(reinterpret_cast<const char *>(rec) + (e >> 6) * (unsigned)(kRecDoubles * sizeof(double)));
    double pg[8];
    const double *tm = terms + pl * kTermStride;
#pragma unroll
    for (int k = 0; k < 8; k++) pg[k] = tm[k];
    pair_update<KIND>(P, R, tp[pl].tx, tp[pl].ty, tp[pl].tz, pg, tm + 8, acc, live, accepted);
  }
  // block reduction through LDS in a fixed order (bitwise reproducible): transpose the 28 per-lane accumulators (two
  // rounds of 128 lanes), 8 x 32 lanes add 32 rows each, then 32 lanes add the 8 partial rows
  __syncthreads();  // the transposition area aliases the other waves' tile slices: every wave must have left its loop
  double *red = reinterpret_cast<double *>(lds);
  double *red2 = reinterpret_cast<double *>(lds + kLdsRed2);
  double *wcnt = red2 + 8 * 32;  // [kWaves][2]: pairs, points per wave
  {
    const unsigned wave_pairs = accepted;
    // source points this wave walked: its tiles are tile0, tile0 + tstride, ...; all full but possibly the cloud's last one
    unsigned long long npts = 0;
    if (tile0 < ntiles) {
      const size_t last = ntiles - 1 - tile0;
      npts = (unsigned long long)(last / tstride + 1) * kTilePts;
      if (last % tstride == 0) npts -= (unsigned long long)(ntiles * kTilePts - n);
    }
    if (lane == 0) { wcnt[wave * 2] = (double)wave_pairs; wcnt[wave * 2 + 1] = (double)npts; }
  }
#pragma unroll
  for (int round = 0; round < kBlockD / kRedLanes; round++) {
    if ((int)(threadIdx.x / kRedLanes) == round) {
#pragma unroll
      for (int k = 0; k < 28; k++) red[(threadIdx.x % kRedLanes) * kRedStride + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < kRedLanes) {
      int k = threadIdx.x & 31, part = threadIdx.x >> 5;  // 4 parts of 32 rows per round
      double v = 0.0;
      if (k < 28) {
        // all 32 rows are fetched before the first addition (the accumulators are dead by now: registers are free), and added as
        // four interleaved chains of eight (rows c, c + 4, c + 8, ...) joined pairwise: a FIXED order like the single chain it
        // replaces, at a quarter of its dependent-addition latency (the pass is latency bound at KITTI size: 0.72 -> 0.5 us a round)
        const double *r = red + (size_t)(part * 32) * kRedStride + k;
        double x[32];
#pragma unroll
        for (int c = 0; c < 32; c++) x[c] = r[c * kRedStride];
        double c0 = x[0], c1 = x[1], c2 = x[2], c3 = x[3];
#pragma unroll
        for (int c = 4; c < 32; c += 4) { c0 += x[c]; c1 += x[c + 1]; c2 += x[c + 2]; c3 += x[c + 3]; }
        v = (c0 + c1) + (c2 + c3);
      }
      red2[(round * (kRedLanes / 32) + part) * 32 + k] = v;
    }
    __syncthreads();
  }
  double v = 0.0;
  if (threadIdx.x < kOutDoubles) {
    if (threadIdx.x < 28) {
      double y[8];
#pragma unroll
      for (int part = 0; part < 8; part++) y[part] = red2[part * 32 + threadIdx.x];
      v = ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));   // fixed pairwise order
    } else if (threadIdx.x == 28) {
      for (int w = 0; w < kWaves; w++) v += wcnt[w * 2];       // (point, voxel) pairs of this workgroup
    } else if (threadIdx.x == 29) {
      for (int w = 0; w < kWaves; w++) v += wcnt[w * 2 + 1];   // source points of this workgroup
    }
  }
  return v;  // threads 0..31: column threadIdx.x of this workgroup's 32-double row
}

}  // namespace

#endif
