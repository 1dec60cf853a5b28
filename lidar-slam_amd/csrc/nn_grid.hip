// nn_grid.hip — build of the exact nearest-neighbour grid (see nn_grid.h): sparse voxel index over the points, then a
// counting sort of the points by occupied-cell rank.
#include <cstdlib>
#include "nn_grid_dev.h"

namespace {
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_count(const int32_t *__restrict__ cell_idx, size_t n, const uint32_t *__restrict__ bits,
                                                  const uint32_t *__restrict__ prefix, uint32_t *__restrict__ cnt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cell = cell_idx[i];
  if (cell < 0) return;
  unsigned rank = prefix[cell >> 5] + __popc(bits[cell >> 5] & ((1u << (cell & 31)) - 1u));
  atomicAdd(cnt + rank, 1u);
}

// single block: start[i] = exclusive prefix of cnt[0..n), start[n] = total
__global__ __launch_bounds__(1024) void k_excl_scan(const uint32_t *__restrict__ cnt, size_t n, uint32_t *__restrict__ start) {
  __shared__ unsigned s[1024];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n; base += 1024) {
    size_t i = base + threadIdx.x;
    unsigned v = (i < n) ? cnt[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    unsigned incl = s[threadIdx.x], c = carry;
    if (i < n) start[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) start[n] = carry;
}

__global__ __launch_bounds__(kBlock) void k_scatter(const float4 *__restrict__ pts, const int32_t *__restrict__ cell_idx, size_t n,
                                                    const uint32_t *__restrict__ bits, const uint32_t *__restrict__ prefix,
                                                    const uint32_t *__restrict__ start, uint32_t *__restrict__ fill, float4 *__restrict__ sorted) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cell = cell_idx[i];
  if (cell < 0) return;
  unsigned rank = prefix[cell >> 5] + __popc(bits[cell >> 5] & ((1u << (cell & 31)) - 1u));
  unsigned slot = start[rank] + atomicAdd(fill + rank, 1u);
  float4 p = pts[i];
  p.w = __uint_as_float((unsigned)i);  // original index: deterministic tie-break, and what the kd-tree would return
  sorted[slot] = p;
}
// dense[cell] = {start[rank], #points} for every occupied cell (the table was zeroed before)
__global__ __launch_bounds__(kBlock) void k_nn_dense_fill(const int32_t *__restrict__ cell_of_rank, const uint32_t *__restrict__ start, size_t V,
                                                          uint2 *__restrict__ dense) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  unsigned a = start[r], b = start[r + 1];
  dense[cell_of_rank[r]] = make_uint2(a, b - a);
}
struct T12 { float v[12]; };

// four adjacent lanes per source point share its search (nn_nearest_sub); the unbounded ring search ends as soon as the
// best distance is proven minimal, which for a scan registered to its map is the query's own cell or ring 1
constexpr int kFitSub = 4;
__global__ __launch_bounds__(kBlock) void k_nn_fitness(const float4 *__restrict__ src, size_t n, T12 T, float max_range, NNView G,
                                                       double *__restrict__ partials) {
  double sum = 0.0, cnt = 0.0;
  const int sub = threadIdx.x & (kFitSub - 1);
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / kFitSub; i < n; i += (size_t)gridDim.x * blockDim.x / kFitSub) {
    float4 s = src[i];
    float qx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[0], s.x), lsa_fmul(T.v[1], s.y)), lsa_fmul(T.v[2], s.z)), T.v[3]);
    float qy = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[4], s.x), lsa_fmul(T.v[5], s.y)), lsa_fmul(T.v[6], s.z)), T.v[7]);
    float qz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T.v[8], s.x), lsa_fmul(T.v[9], s.y)), lsa_fmul(T.v[10], s.z)), T.v[11]);
    if (!lsa_finite3(qx, qy, qz)) continue;
    NNHit h = nn_nearest_sub<kFitSub>(G, qx, qy, qz, 0x7fffffff, sub);
    if (sub == 0 && h.id != 0xffffffffu && h.d2 <= max_range) { sum += (double)h.d2; cnt += 1.0; }
  }
  __shared__ double sm[kBlock / 64][2];
  sum = lsa_wave_sum(sum);
  cnt = lsa_wave_sum(cnt);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave][0] = sum; sm[wave][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; w++) v += sm[w][threadIdx.x];
    partials[(size_t)blockIdx.x * 2 + threadIdx.x] = v;
  }
}
}  // namespace

int nn_fitness(lsa_context *ctx, NNGrid &g, const float4 *d_src, size_t n, const float T[16], float max_range, double *out, lsa_allreduce_fn allreduce,
               void *allreduce_user) {
  if (!g.built) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "fitness: no neighbour grid");
  *out = 1.7976931348623157e308;
  if (!n && !allreduce) return LSA_OK;   // (a rank with an empty shard still takes part in the collective)
  unsigned nb = (unsigned)std::min<size_t>(lsa_div_up(n * kFitSub, kBlock), (size_t)ctx->num_cus * 8);
  int rc;
  if ((rc = g.fit_partials.reserve(ctx, (size_t)(nb ? nb : 1) * 2, 1.0))) return rc;
  if ((rc = g.fit_host.reserve(ctx, (size_t)(nb ? nb : 1) * 2))) return rc;
  T12 t;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) t.v[r * 4 + c] = T[c * 4 + r];
  if (n) {
    hipLaunchKernelGGL(k_nn_fitness, dim3(nb), dim3(kBlock), 0, ctx->stream, d_src, n, t, max_range, nn_view(g), g.fit_partials.p);
    LSA_HIP(ctx, hipGetLastError());
    LSA_HIP(ctx, hipMemcpyAsync(g.fit_host.p, g.fit_partials.p, (size_t)nb * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  double sum = 0.0, cnt = 0.0;
  for (unsigned b = 0; b < nb && n; b++) { sum += g.fit_host.p[2 * b]; cnt += g.fit_host.p[2 * b + 1]; }  // fixed order
  if (allreduce) {
    // source sharded over the ranks ("one extra all-reduce of 2 scalars for fitness"): {sum, count} of this shard
    // go back to the device, are summed over the ranks in place and come back
    g.fit_host.p[0] = sum; g.fit_host.p[1] = cnt;
    LSA_HIP(ctx, hipMemcpyAsync(g.fit_partials.p, g.fit_host.p, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int crc = allreduce(allreduce_user, g.fit_partials.p, 2, (void *)ctx->stream);
    if (crc) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "fitness: all-reduce callback returned %d", crc);
    LSA_HIP(ctx, hipMemcpyAsync(g.fit_host.p, g.fit_partials.p, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sum = g.fit_host.p[0]; cnt = g.fit_host.p[1];
  }
  if (cnt > 0) *out = sum / cnt;
  return LSA_OK;
}

int nn_grid_build(lsa_context *ctx, NNGrid &g, const float4 *d_pts, size_t n, float cell, const char *who) {
  VoxelIndex &vi = g.vi;
  vi.want_cell_idx = true;   // k_count / k_scatter below walk the points by their cell
  g.built = false;
  if (!(cell > 0.05f)) cell = 0.05f;
  const float leaf[3] = {cell, cell, cell};
  int rc, status = VI_OK;
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = vi_enqueue(ctx, vi, d_pts, n, /*mode: floor(x / cell)*/ 1, leaf, 4))) return rc;
    if ((rc = vi_fetch(ctx, vi))) return rc;
    if ((rc = vi_wait(ctx, vi, &status))) return rc;
    if (status != VI_CAPACITY) break;
    if ((rc = vi_grow(ctx, vi))) return rc;
  }
  if (status == VI_EMPTY) return lsa_fail(ctx, LSA_ERR_INVALID, "%s: no finite points", who);
  if (status != VI_OK) return lsa_fail(ctx, LSA_ERR_RANGE, "%s: neighbour grid overflows int32", who);
  size_t V = vi.nocc;
  if ((rc = g.cnt.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = g.start.reserve(ctx, V + 2, 1.25))) return rc;
  if ((rc = g.fill.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = g.sorted.reserve(ctx, n, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(g.cnt.p, 0, (V + 1) * sizeof(uint32_t), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(g.fill.p, 0, (V + 1) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_count, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, vi.cell_idx.p, n, vi.bits.p, vi.prefix.p, g.cnt.p);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, g.cnt.p, V, g.start.p);
  hipLaunchKernelGGL(k_scatter, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, vi.cell_idx.p, n, vi.bits.p, vi.prefix.p, g.start.p,
                     g.fill.p, g.sorted.p);
  {
    const CellMap &m = vi.map;
    const long long ncell = (long long)m.dim[0] * m.dim[1] * m.dim[2];
    const char *lim = getenv("LSA_NN_DENSE_MAX_CELLS");  // diagnostics / tests: 0 forces the bitmap walk
    g.use_dense = ncell > 0 && ncell <= (lim ? atoll(lim) : kNNDenseMaxCells);
    if (g.use_dense) {
      if ((rc = g.dense.reserve(ctx, (size_t)ncell, 1.25))) return rc;
      LSA_HIP(ctx, hipMemsetAsync(g.dense.p, 0, (size_t)ncell * sizeof(uint2), ctx->stream));
      hipLaunchKernelGGL(k_nn_dense_fill, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, vi.cell_of_rank.p, g.start.p, V, g.dense.p);
    }
  }
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g.n = n;
  g.n_sorted = (size_t)vi.h_params.p[0].n_finite;
  g.cell = cell;
  g.built = true;
  return LSA_OK;
}
