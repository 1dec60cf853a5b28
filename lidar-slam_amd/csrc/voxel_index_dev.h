// voxel_index_dev.h — internal device helpers shared by the .hip translation units.
#ifndef LSA_VOXEL_INDEX_DEV_H_
#define LSA_VOXEL_INDEX_DEV_H_

#include "voxel_index.h"

// f32 primitives that must NOT be contracted into FMAs: the reference (PCL / the in-tree VoxelGrid) is built for
// baseline x86-64 and rounds every product and sum separately.  HIP's __fmul_rn / __fadd_rn are plain `x * y` /
// `x + y` (clang __clang_hip_math.h) and DO get fused after inlining under the default -ffp-contract, so each
// operation is fenced with the pragma instead.
__device__ __forceinline__ float lsa_fmul(float a, float b) {
#pragma clang fp contract(off)
  float r = a * b;
  return r;
}
__device__ __forceinline__ float lsa_fadd(float a, float b) {
#pragma clang fp contract(off)
  float r = a + b;
  return r;
}
__device__ __forceinline__ float lsa_fsub(float a, float b) {
#pragma clang fp contract(off)
  float r = a - b;
  return r;
}
__device__ __forceinline__ float lsa_fdiv(float a, float b) {
#pragma clang fp contract(off)
  float r = a / b;  // correctly rounded: built with -fhip-fp32-correctly-rounded-divide-sqrt
  return r;
}

__device__ __forceinline__ bool lsa_finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

__host__ __device__ inline int lsa_ceil_log2(double v) {
  if (!(v > 0)) return 0;
  int e;
  double f = frexp(v, &e);
  return (f == 0.5) ? e - 1 : e;
}

// Fixed-point scales of the per-cell sums for up to `n` summands of |d| <= leaf (+ slop) < 2 * leaf and |intensity| < 1024:
// no int64 sum can overflow.  *n_cap: the largest point count the scales are safe for.  One function for the device-side
// grid definition (voxel_index.hip) and the host (the incremental local map, whose persistent sums must be quantised
// exactly as a full rebuild quantises them).
__host__ __device__ inline FixScales vi_make_scales(double n, double leaf_max, unsigned long long *n_cap) {
  int bits_n = lsa_ceil_log2(n + 1.0), bits_d = lsa_ceil_log2(2.0 * leaf_max);
  int k1 = 62 - bits_n - bits_d, k2 = 62 - bits_n - 2 * bits_d, kw = 62 - bits_n - 10;
  k1 = k1 < 44 ? k1 : 44; k2 = k2 < 44 ? k2 : 44; kw = kw < 40 ? kw : 40;
  FixScales fs;
  fs.s1 = ldexp(1.0, k1); fs.inv_s1 = ldexp(1.0, -k1);
  fs.s2 = ldexp(1.0, k2); fs.inv_s2 = ldexp(1.0, -k2);
  fs.sw = ldexp(1.0, kw); fs.inv_sw = ldexp(1.0, -kw);
  // |d| < 2^bits_d, |d d| < 2^(2 bits_d), |w| < 2^10: below 2^51 the terms can be rounded with the magic-number add
  fs.fast_round = (bits_d + k1 <= 50 && 2 * bits_d + k2 <= 50 && 10 + kw <= 50) ? 1 : 0;
  fs.pad = 0;
  if (n_cap) *n_cap = (bits_n >= 62) ? ~0ull : ((1ull << bits_n) - 1ull);
  return fs;
}

// integer cell coordinates (relative to min_b) of a point.
//  mode 0: PCL voxel_grid.hpp: ijk = (int)(floor(x * inverse_leaf) - (float)min_b)
//  mode 1: reference VoxelGrid.cpp:218-222: floor(p.x / voxel_x_) - min_b   (f32 division, correctly rounded)
__device__ __forceinline__ void lsa_cell_ijk(const CellMap &m, float x, float y, float z, int ijk[3]) {
  if (m.mode == 0) {
    ijk[0] = (int)(floorf(lsa_fmul(x, m.inv[0])) - (float)m.min_b[0]);
    ijk[1] = (int)(floorf(lsa_fmul(y, m.inv[1])) - (float)m.min_b[1]);
    ijk[2] = (int)(floorf(lsa_fmul(z, m.inv[2])) - (float)m.min_b[2]);
  } else {
    ijk[0] = (int)floorf(lsa_fdiv(x, m.leaf[0])) - m.min_b[0];
    ijk[1] = (int)floorf(lsa_fdiv(y, m.leaf[1])) - m.min_b[1];
    ijk[2] = (int)floorf(lsa_fdiv(z, m.leaf[2])) - m.min_b[2];
  }
}

__device__ __forceinline__ int lsa_cell_linear(const CellMap &m, const int ijk[3]) {
  if ((unsigned)ijk[0] >= (unsigned)m.dim[0] || (unsigned)ijk[1] >= (unsigned)m.dim[1] || (unsigned)ijk[2] >= (unsigned)m.dim[2]) return -1;
  return ijk[0] + ijk[1] * m.dim[0] + ijk[2] * m.dim[0] * m.dim[1];
}

// reference corner of a cell, the offset all fixed-point sums are taken against (any fixed value works;
// the corner keeps |d| <= ~leaf so the sums stay far from int64 overflow)
// (The three helpers below round every product and sum separately — contraction off — so that the full index build and the
// incremental local map, which inline them into different kernels, cannot end up with differently fused arithmetic: their
// per-cell results must agree to the bit.)
__device__ __forceinline__ void lsa_abs_origin(const int ijk_abs[3], const float leaf[3], double o[3]) {
#pragma clang fp contract(off)
  o[0] = (double)ijk_abs[0] * (double)leaf[0];
  o[1] = (double)ijk_abs[1] * (double)leaf[1];
  o[2] = (double)ijk_abs[2] * (double)leaf[2];
}
__device__ __forceinline__ void lsa_cell_origin(const CellMap &m, int cell, double o[3]) {
  int k = cell / (m.dim[0] * m.dim[1]);
  int r = cell - k * (m.dim[0] * m.dim[1]);
  int j = r / m.dim[0];
  int i = r - j * m.dim[0];
  const int ijk[3] = {i + m.min_b[0], j + m.min_b[1], k + m.min_b[2]};
  lsa_abs_origin(ijk, m.leaf, o);
}

// fixed-point terms of one point against its cell's low corner `o`: first moments (3), [second moments (6)], intensity
// x rounded to the nearest integer, ties to even, as a 64-bit integer (what __double2ll_rn gives).  FAST (|x| < 2^51, which the
// scales guarantee when FixScales::fast_round is set): x + 1.5 * 2^52 lands in [2^52, 2^53), where doubles are the integers, so the
// add itself rounds x to the nearest-even integer and the sum's low mantissa bits are that integer in offset form: one f64 add and
// one 64-bit integer subtraction instead of rint + the ~10-instruction f64 -> i64 sequence (gfx950 has no f64 -> i64 conversion), on a
// path that runs once per point and term (4 or 10 per point in every voxel-index build and in the incremental local map).
__device__ __forceinline__ unsigned long long vi_round_ll(double x, bool fast) {
#pragma clang fp contract(off)
  if (fast) {
    const double magic = 6755399441055744.0;   // 1.5 * 2^52
    return (unsigned long long)(__double_as_longlong(x + magic) - __double_as_longlong(magic));
  }
  return (unsigned long long)__double2ll_rn(x);
}
// The intensity term.  Unlike the coordinate terms (a point lies in its cell: |d| <= leaf by construction) nothing bounds an
// intensity: the scales assume |w| < 1024 (KITTI reflectance and the reference's static weights are in [0, 1]), and only the
// coordinates are tested for finiteness.  The magic-number rounding turns anything at or beyond 2^51 — a 16-bit reflectance at
// kw = 40, an infinity, a NaN — into arbitrary integer bits, which the incremental local map would then keep in its persistent
// sums until the key-frame is evicted.  So: a non-finite intensity counts as 0, a finite one beyond the fast range takes the exact
// conversion (clamped to +-2^62: a cell's sum of such terms can still overflow, the documented limit is |w| < 1024).  The branch is
// never taken on in-range data (one compare per point).
__device__ __forceinline__ unsigned long long vi_weight_term(float w, const FixScales &fs) {
#pragma clang fp contract(off)
  const double x = (double)w * fs.sw;
  if (__builtin_expect(fs.fast_round != 0 && fabs(x) < 2251799813685248.0 /* 2^51 */, 1)) return vi_round_ll(x, true);
  if (!isfinite(w)) return 0ull;
  const double lim = 4611686018427387904.0;   // 2^62
  return (unsigned long long)__double2ll_rn(fmin(fmax(x, -lim), lim));
}
template <int NS>
__device__ __forceinline__ void vi_point_terms(const double o[3], const FixScales &fs, const float4 &p, unsigned long long v[NS]) {
#pragma clang fp contract(off)
  const bool fast = fs.fast_round != 0;
  double dx = (double)p.x - o[0], dy = (double)p.y - o[1], dz = (double)p.z - o[2];
  v[0] = vi_round_ll(dx * fs.s1, fast);
  v[1] = vi_round_ll(dy * fs.s1, fast);
  v[2] = vi_round_ll(dz * fs.s1, fast);
  if (NS == 10) {
    v[3] = vi_round_ll(dx * dx * fs.s2, fast);
    v[4] = vi_round_ll(dx * dy * fs.s2, fast);
    v[5] = vi_round_ll(dx * dz * fs.s2, fast);
    v[6] = vi_round_ll(dy * dy * fs.s2, fast);
    v[7] = vi_round_ll(dy * dz * fs.s2, fast);
    v[8] = vi_round_ll(dz * dz * fs.s2, fast);
  }
  v[NS - 1] = vi_weight_term(p.w, fs);
}

// the VoxelFilter's output point of a cell: the f32 mean of x, y, z and the intensity from the cell's four sums
__device__ __forceinline__ float4 vi_filter_mean(const long long a[4], int cnt, const double o[3], const FixScales &fs) {
#pragma clang fp contract(off)
  const double n = (double)cnt;
  float4 v;
  v.x = (float)(o[0] + ((double)a[0] * fs.inv_s1) / n);
  v.y = (float)(o[1] + ((double)a[1] * fs.inv_s1) / n);
  v.z = (float)(o[2] + ((double)a[2] * fs.inv_s1) / n);
  v.w = (float)(((double)a[3] * fs.inv_sw) / n);
  return v;
}

__device__ __forceinline__ void lsa_cell_unravel(const CellMap &m, int cell, int ijk_abs[3]) {
  int k = cell / (m.dim[0] * m.dim[1]);
  int r = cell - k * (m.dim[0] * m.dim[1]);
  int j = r / m.dim[0];
  int i = r - j * m.dim[0];
  ijk_abs[0] = i + m.min_b[0]; ijk_abs[1] = j + m.min_b[1]; ijk_abs[2] = k + m.min_b[2];
}

// Lanes 0..31 of a wave write one 32-double workgroup row.  key != 0: the row goes to host-mapped memory and the host
// picks it up without waiting for the stream, so column 31 carries key ^ XOR_c rotl(word_c, c + 1) over the other 31 words
// (it must hold no payload): the host can tell a complete row of THIS pass from a stale or half-arrived one whatever order
// the stores land in.  The rotation by the column index keeps equal or swapped stale columns from cancelling.
__host__ __device__ __forceinline__ unsigned long long lsa_row_mix(unsigned long long w, int col) {
  const int r = col + 1;  // 1..31
  return (w << r) | (w >> (64 - r));
}
__device__ __forceinline__ void lsa_store_row32(double *__restrict__ row, int col, double v, unsigned long long key) {
  if (!key) { row[col] = v; return; }
  unsigned long long w = (unsigned long long)__double_as_longlong(v);
  unsigned long long x = col == 31 ? 0ull : lsa_row_mix(w, col);
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) x ^= __shfl_xor(x, off, 64);
  if (col == 31) w = x ^ key;
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(row) + col, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// N wave sums at once, STEP-major: the N exchanges of one step are issued together and waited for once.  Calling lsa_wave_sum
// per value serialises N x 6 cross-lane exchanges (each a ~100-cycle LDS-crossbar round trip the compiler does not overlap:
// 27 sums = 162 dependent waits, ~9 us at the tail of an ICP pass); the association order per value is the same, so lane 0
// ends up with the same bits.
template <int N>
__device__ __forceinline__ void lsa_wave_sum_n(double (&v)[N]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = __shfl_down(v[k], off, 64);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += t[k];
  }
}

__device__ __forceinline__ double lsa_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

#endif
