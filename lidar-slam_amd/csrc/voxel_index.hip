// voxel_index.hip — sparse voxel index: bbox reduce, occupancy bitmap, popcount prefix scan,
// fixed-point per-cell accumulation.  HBM-bound streaming kernels over packed float4 points
// (16 B/point, one global_load_dwordx4 per lane, grid-stride, grid sized >> 256 CUs).
#include "voxel_index_dev.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kScanTile = 2048;  // words per scan block: 256 threads x 8 words

__global__ __launch_bounds__(kBlock) void k_bbox_partial(const float4 *__restrict__ pts, size_t n, float *__restrict__ partial) {
  float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
  unsigned cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (lsa_finite3(p.x, p.y, p.z)) {
      mn0 = fminf(mn0, p.x); mn1 = fminf(mn1, p.y); mn2 = fminf(mn2, p.z);
      mx0 = fmaxf(mx0, p.x); mx1 = fmaxf(mx1, p.y); mx2 = fmaxf(mx2, p.z);
      cnt++;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn0 = fminf(mn0, __shfl_down(mn0, off, 64)); mn1 = fminf(mn1, __shfl_down(mn1, off, 64)); mn2 = fminf(mn2, __shfl_down(mn2, off, 64));
    mx0 = fmaxf(mx0, __shfl_down(mx0, off, 64)); mx1 = fmaxf(mx1, __shfl_down(mx1, off, 64)); mx2 = fmaxf(mx2, __shfl_down(mx2, off, 64));
    cnt += __shfl_down(cnt, off, 64);
  }
  __shared__ float s[kBlock / 64][8];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    s[wave][0] = mn0; s[wave][1] = mn1; s[wave][2] = mn2; s[wave][3] = mx0; s[wave][4] = mx1; s[wave][5] = mx2;
    s[wave][6] = __uint_as_float(cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned c = 0;
    for (int w = 0; w < kBlock / 64; w++) {
      mn0 = fminf(mn0, s[w][0]); mn1 = fminf(mn1, s[w][1]); mn2 = fminf(mn2, s[w][2]);
      mx0 = fmaxf(mx0, s[w][3]); mx1 = fmaxf(mx1, s[w][4]); mx2 = fmaxf(mx2, s[w][5]);
      c += __float_as_uint(s[w][6]);
    }
    float *o = partial + (size_t)blockIdx.x * 8;
    o[0] = mn0; o[1] = mn1; o[2] = mn2; o[3] = mx0; o[4] = mx1; o[5] = mx2; o[6] = __uint_as_float(c); o[7] = 0.f;
  }
}

__global__ __launch_bounds__(kBlock) void k_bbox_final(const float *__restrict__ partial, int nb, float *__restrict__ out) {
  float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
  unsigned long long cnt = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    const float *p = partial + (size_t)i * 8;
    mn0 = fminf(mn0, p[0]); mn1 = fminf(mn1, p[1]); mn2 = fminf(mn2, p[2]);
    mx0 = fmaxf(mx0, p[3]); mx1 = fmaxf(mx1, p[4]); mx2 = fmaxf(mx2, p[5]);
    cnt += __float_as_uint(p[6]);
  }
  __shared__ float s[kBlock][6];
  __shared__ unsigned long long sc[kBlock];
  s[threadIdx.x][0] = mn0; s[threadIdx.x][1] = mn1; s[threadIdx.x][2] = mn2;
  s[threadIdx.x][3] = mx0; s[threadIdx.x][4] = mx1; s[threadIdx.x][5] = mx2;
  sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      for (int a = 0; a < 3; a++) s[threadIdx.x][a] = fminf(s[threadIdx.x][a], s[threadIdx.x + st][a]);
      for (int a = 3; a < 6; a++) s[threadIdx.x][a] = fmaxf(s[threadIdx.x][a], s[threadIdx.x + st][a]);
      sc[threadIdx.x] += sc[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int a = 0; a < 6; a++) out[a] = s[0][a];
    unsigned long long c = sc[0];
    out[6] = __uint_as_float((unsigned)(c & 0xffffffffu));
    out[7] = __uint_as_float((unsigned)(c >> 32));
  }
}

// ---- LDS tile hashing ---------------------------------------------------------------------------------------
// Consecutive lidar returns hit the same few hundred voxels, so a tile of kTile consecutive points is first
// de-duplicated / pre-aggregated in an LDS open-addressing table keyed by the cell index; only one global atomic per
// DISTINCT cell of the tile leaves the CU.  (Global atomics execute at the memory side at ~1e10/s and serialise on a
// shared word; LDS atomics are two orders of magnitude cheaper.)  A point that cannot be placed because the table is
// full falls back to direct global atomics, so correctness never depends on the tile's diversity.
constexpr int kTile = 2048;      // points per workgroup
constexpr int kHashCap = 1024;   // LDS slots (power of two)

__device__ __forceinline__ unsigned lds_hash(int cell) { return ((unsigned)cell * 2654435761u) >> 22; }  // top 10 bits

// returns the slot of `cell` in keys[] (inserting it if new), or -1 if the table is full
__device__ __forceinline__ int lds_insert(int *keys, int cell) {
  unsigned h = lds_hash(cell) & (kHashCap - 1);
  for (int probe = 0; probe < kHashCap; probe++) {
    int old = keys[h];
    if (old == cell) return (int)h;
    if (old == -1) {
      old = atomicCAS(&keys[h], -1, cell);
      if (old == -1 || old == cell) return (int)h;
    }
    h = (h + 1) & (kHashCap - 1);
  }
  return -1;
}

__global__ __launch_bounds__(kBlock) void k_mark(const float4 *__restrict__ pts, size_t n, CellMap m, uint32_t *__restrict__ bits,
                                                 int32_t *__restrict__ cell_idx) {
  __shared__ int keys[kHashCap];
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) keys[k] = -1;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * kTile;
#pragma unroll 2
  for (int k = 0; k < kTile / kBlock; k++) {
    size_t i = base + (size_t)k * kBlock + threadIdx.x;
    if (i >= n) break;
    float4 p = pts[i];
    int cell = -1;
    if (lsa_finite3(p.x, p.y, p.z)) {
      int ijk[3];
      lsa_cell_ijk(m, p.x, p.y, p.z, ijk);
      cell = lsa_cell_linear(m, ijk);
    }
    cell_idx[i] = cell;
    if (cell >= 0 && lds_insert(keys, cell) < 0) atomicOr(bits + (cell >> 5), 1u << (cell & 31));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) {
    int cell = keys[k];
    if (cell >= 0) {
      uint32_t bit = 1u << (cell & 31);
      uint32_t *w = bits + (cell >> 5);
      if (!(*w & bit)) atomicOr(w, bit);  // bits only go 0 -> 1 here: a stale 0 merely costs the atomic
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_scan_block_sums(const uint32_t *__restrict__ bits, size_t W, uint32_t *__restrict__ blocksums) {
  size_t base = (size_t)blockIdx.x * kScanTile;
  unsigned s = 0;
  for (int k = 0; k < kScanTile / kBlock; k++) {
    size_t i = base + (size_t)k * kBlock + threadIdx.x;
    if (i < W) s += __popc(bits[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ unsigned sw[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kBlock / 64; w++) t += sw[w];
    blocksums[blockIdx.x] = t;
  }
}

// single block: exclusive scan of blocksums[nb] in place, total -> blocksums[nb]
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *__restrict__ blocksums, int nb) {
  __shared__ unsigned s[1024];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + threadIdx.x;
    unsigned v = (i < nb) ? blocksums[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    unsigned incl = s[threadIdx.x];
    unsigned c = carry;
    if (i < nb) blocksums[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) blocksums[nb] = carry;
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(const uint32_t *__restrict__ bits, size_t W, const uint32_t *__restrict__ blocksums,
                                                       uint32_t *__restrict__ prefix, int32_t *__restrict__ cell_of_rank) {
  // thread t owns 8 consecutive words of the tile so the in-block scan is one value per thread
  size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * 8;
  uint32_t wv[8];
  unsigned local = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    size_t i = base + k;
    wv[k] = (i < W) ? bits[i] : 0u;
    local += __popc(wv[k]);
  }
  __shared__ unsigned s[kBlock];
  s[threadIdx.x] = local;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = blocksums[blockIdx.x] + s[threadIdx.x] - local;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    size_t i = base + k;
    if (i < W) {
      prefix[i] = run;
      uint32_t b = wv[k];
      while (b) {
        int bit = __ffs(b) - 1;
        b &= b - 1;
        cell_of_rank[run++] = (int32_t)(i * 32 + bit);
      }
    }
  }
}

template <int NS>
__device__ __forceinline__ void point_terms(const CellMap &m, const FixScales &fs, int cell, const float4 &p, unsigned long long v[NS]) {
  double o[3];
  lsa_cell_origin(m, cell, o);
  double dx = (double)p.x - o[0], dy = (double)p.y - o[1], dz = (double)p.z - o[2];
  v[0] = (unsigned long long)__double2ll_rn(dx * fs.s1);
  v[1] = (unsigned long long)__double2ll_rn(dy * fs.s1);
  v[2] = (unsigned long long)__double2ll_rn(dz * fs.s1);
  if (NS == 10) {
    v[3] = (unsigned long long)__double2ll_rn(dx * dx * fs.s2);
    v[4] = (unsigned long long)__double2ll_rn(dx * dy * fs.s2);
    v[5] = (unsigned long long)__double2ll_rn(dx * dz * fs.s2);
    v[6] = (unsigned long long)__double2ll_rn(dy * dy * fs.s2);
    v[7] = (unsigned long long)__double2ll_rn(dy * dz * fs.s2);
    v[8] = (unsigned long long)__double2ll_rn(dz * dz * fs.s2);
  }
  v[NS - 1] = (unsigned long long)__double2ll_rn((double)p.w * fs.sw);
}

__device__ __forceinline__ unsigned cell_rank(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ prefix, int cell) {
  return prefix[cell >> 5] + __popc(bits[cell >> 5] & ((1u << (cell & 31)) - 1u));
}

// per-cell fixed-point sums: LDS pre-aggregation per tile, then one set of global atomics per distinct cell
template <int NS>
__global__ __launch_bounds__(kBlock) void k_accum(const float4 *__restrict__ pts, const int32_t *__restrict__ cell_idx, size_t n, CellMap m,
                                                  FixScales fs, const uint32_t *__restrict__ bits, const uint32_t *__restrict__ prefix,
                                                  long long *__restrict__ acc, int32_t *__restrict__ cnt) {
  __shared__ int keys[kHashCap];
  __shared__ int counts[kHashCap];
  __shared__ unsigned long long sums[kHashCap][NS];
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) {
    keys[k] = -1;
    counts[k] = 0;
#pragma unroll
    for (int j = 0; j < NS; j++) sums[k][j] = 0ull;
  }
  __syncthreads();
  size_t base = (size_t)blockIdx.x * kTile;
  for (int k = 0; k < kTile / kBlock; k++) {
    size_t i = base + (size_t)k * kBlock + threadIdx.x;
    if (i >= n) break;
    int cell = cell_idx[i];
    if (cell < 0) continue;
    float4 p = pts[i];
    unsigned long long v[NS];
    point_terms<NS>(m, fs, cell, p, v);
    int slot = lds_insert(keys, cell);
    if (slot >= 0) {
#pragma unroll
      for (int j = 0; j < NS; j++) atomicAdd(&sums[slot][j], v[j]);
      atomicAdd(&counts[slot], 1);
    } else {  // table full: straight to memory
      unsigned rank = cell_rank(bits, prefix, cell);
      unsigned long long *a = (unsigned long long *)(acc + (size_t)rank * NS);
#pragma unroll
      for (int j = 0; j < NS; j++) atomicAdd(a + j, v[j]);
      atomicAdd(cnt + rank, 1);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) {
    int cell = keys[k];
    if (cell < 0) continue;
    unsigned rank = cell_rank(bits, prefix, cell);
    unsigned long long *a = (unsigned long long *)(acc + (size_t)rank * NS);
#pragma unroll
    for (int j = 0; j < NS; j++) atomicAdd(a + j, sums[k][j]);
    atomicAdd(cnt + rank, counts[k]);
  }
}

inline unsigned stream_grid(lsa_context *ctx, size_t n) {
  size_t blocks = (n + kBlock - 1) / kBlock;
  size_t cap = (size_t)ctx->num_cus * 8;  // >> 256 workgroups, grid-stride beyond (guide: ~2048 blocks)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

int ceil_log2(double v) {
  if (!(v > 0)) return 0;
  int e;
  double f = frexp(v, &e);  // v = f * 2^e, f in [0.5,1)
  return (f == 0.5) ? e - 1 : e;
}

}  // namespace

int vi_bbox(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, float mn[3], float mx[3], size_t *n_finite) {
  unsigned nb = stream_grid(ctx, n);
  int rc = vi.scratch.reserve(ctx, (size_t)nb * 8 + 8);
  if (rc) return rc;
  rc = vi.h_small.reserve(ctx, 16);
  if (rc) return rc;
  float *partial = vi.scratch.p;
  float *out = vi.scratch.p + (size_t)nb * 8;
  hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(kBlock), 0, ctx->stream, d_pts, n, partial);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(kBlock), 0, ctx->stream, partial, (int)nb, out);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipMemcpyAsync(vi.h_small.p, out, 8 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int a = 0; a < 3; a++) { mn[a] = vi.h_small.p[a]; mx[a] = vi.h_small.p[3 + a]; }
  uint32_t lo, hi;
  memcpy(&lo, &vi.h_small.p[6], 4);
  memcpy(&hi, &vi.h_small.p[7], 4);
  *n_finite = (size_t)(((unsigned long long)hi << 32) | lo);
  return LSA_OK;
}

int vi_make_map_pcl(const float mn[3], const float mx[3], const float leaf[3], CellMap *m) {
  // PCL 1.7 filters/impl/voxel_grid.hpp applyFilter: overflow check, min_b/max_b/div_b
  m->mode = 0;
  long long d[3];
  for (int a = 0; a < 3; a++) {
    m->leaf[a] = leaf[a];
    m->inv[a] = 1.0f / leaf[a];
    d[a] = (long long)((mx[a] - mn[a]) * m->inv[a]) + 1;
  }
  // (PCL multiplies in int64, which itself overflows past ~9.2e18 cells; compare in double so that case is caught too)
  if ((double)d[0] * (double)d[1] * (double)d[2] > (double)INT32_MAX) return LSA_ERR_RANGE;
  for (int a = 0; a < 3; a++) {
    m->min_b[a] = (int)floorf(mn[a] * m->inv[a]);
    int max_b = (int)floorf(mx[a] * m->inv[a]);
    m->dim[a] = max_b - m->min_b[a] + 1;
  }
  m->ncells = (long long)m->dim[0] * m->dim[1] * m->dim[2];
  if (m->ncells > (long long)INT32_MAX) return LSA_ERR_RANGE;
  return LSA_OK;
}

int vi_make_map_ndt(const float mn[3], const float mx[3], float leaf, CellMap *m) {
  // reference VoxelGrid::findBoundaries (VoxelGrid.cpp:366-372): real_min/real_max = floor(min|max / leaf).
  // The reference additionally pads its DENSE allocation to multiples of 16x16x8 cells (:379-392); every lookup
  // is clipped to the real bounds (:449-455), so the padding is an allocation detail with no observable effect.
  m->mode = 1;
  for (int a = 0; a < 3; a++) {
    m->leaf[a] = leaf;
    m->inv[a] = 1.0f / leaf;
    m->min_b[a] = (int)floorf(mn[a] / leaf);
    int max_b = (int)floorf(mx[a] / leaf);
    m->dim[a] = max_b - m->min_b[a] + 1;
  }
  m->ncells = (long long)m->dim[0] * m->dim[1] * m->dim[2];
  if (m->ncells > (long long)INT32_MAX || m->dim[0] <= 0 || m->dim[1] <= 0 || m->dim[2] <= 0) return LSA_ERR_RANGE;
  return LSA_OK;
}

void vi_make_scales(const CellMap &m, size_t n, FixScales *fs) {
  double leaf_max = fmax(fmax(m.leaf[0], m.leaf[1]), m.leaf[2]);
  int bits_n = ceil_log2((double)n + 1.0);
  int bits_d = ceil_log2(2.0 * leaf_max);  // |d| <= leaf (+ rounding slop) < 2 * leaf
  int k1 = 62 - bits_n - bits_d;
  int k2 = 62 - bits_n - 2 * bits_d;
  int kw = 62 - bits_n - 10;  // |intensity| < 1024
  if (k1 > 44) k1 = 44;
  if (k2 > 44) k2 = 44;
  if (kw > 40) kw = 40;
  fs->s1 = ldexp(1.0, k1); fs->inv_s1 = ldexp(1.0, -k1);
  fs->s2 = ldexp(1.0, k2); fs->inv_s2 = ldexp(1.0, -k2);
  fs->sw = ldexp(1.0, kw); fs->inv_sw = ldexp(1.0, -kw);
}

int vi_build(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n) {
  const CellMap &m = vi.map;
  vi.W = (size_t)((m.ncells + 31) / 32);
  unsigned nsb = lsa_div_up(vi.W, kScanTile);
  int rc;
  if ((rc = vi.bits.reserve(ctx, vi.W, 1.25))) return rc;
  if ((rc = vi.prefix.reserve(ctx, vi.W, 1.25))) return rc;
  if ((rc = vi.blocksums.reserve(ctx, (size_t)nsb + 1, 1.25))) return rc;
  if ((rc = vi.cell_idx.reserve(ctx, n, 1.25))) return rc;
  if ((rc = vi.cell_of_rank.reserve(ctx, n, 1.25))) return rc;  // #occupied <= #points
  if ((rc = vi.h_small.reserve(ctx, 16))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(vi.bits.p, 0, vi.W * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_mark, dim3(lsa_div_up(n, kTile)), dim3(kBlock), 0, ctx->stream, d_pts, n, m, vi.bits.p, vi.cell_idx.p);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nsb), dim3(kBlock), 0, ctx->stream, vi.bits.p, vi.W, vi.blocksums.p);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, ctx->stream, vi.blocksums.p, (int)nsb);
  hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(kBlock), 0, ctx->stream, vi.bits.p, vi.W, vi.blocksums.p, vi.prefix.p, vi.cell_of_rank.p);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipMemcpyAsync(vi.h_small.p, vi.blocksums.p + nsb, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t tot;
  memcpy(&tot, vi.h_small.p, 4);
  vi.nocc = tot;
  return LSA_OK;
}

int vi_accumulate(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, int ns) {
  int rc;
  if ((rc = vi.acc.reserve(ctx, vi.nocc * (size_t)ns + 1, 1.25))) return rc;
  if ((rc = vi.cnt.reserve(ctx, vi.nocc + 1, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(vi.acc.p, 0, (vi.nocc * (size_t)ns + 1) * sizeof(long long), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(vi.cnt.p, 0, (vi.nocc + 1) * sizeof(int32_t), ctx->stream));
  unsigned g = lsa_div_up(n, kTile);
  if (ns == 4)
    hipLaunchKernelGGL(k_accum<4>, dim3(g), dim3(kBlock), 0, ctx->stream, d_pts, vi.cell_idx.p, n, vi.map, vi.fs, vi.bits.p, vi.prefix.p, vi.acc.p, vi.cnt.p);
  else
    hipLaunchKernelGGL(k_accum<10>, dim3(g), dim3(kBlock), 0, ctx->stream, d_pts, vi.cell_idx.p, n, vi.map, vi.fs, vi.bits.p, vi.prefix.p, vi.acc.p, vi.cnt.p);
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}
