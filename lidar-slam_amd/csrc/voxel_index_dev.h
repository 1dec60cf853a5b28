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
__device__ __forceinline__ void lsa_cell_origin(const CellMap &m, int cell, double o[3]) {
  int k = cell / (m.dim[0] * m.dim[1]);
  int r = cell - k * (m.dim[0] * m.dim[1]);
  int j = r / m.dim[0];
  int i = r - j * m.dim[0];
  o[0] = (double)(i + m.min_b[0]) * (double)m.leaf[0];
  o[1] = (double)(j + m.min_b[1]) * (double)m.leaf[1];
  o[2] = (double)(k + m.min_b[2]) * (double)m.leaf[2];
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
