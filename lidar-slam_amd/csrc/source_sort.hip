// source_sort.hip — spatial ordering of a large, incoherent source cloud before the NDT passes.
//
// The derivative pass (ndt_derivatives.hip) hands 14 CONSECUTIVE source points to a wave; their probes of the dense per-cell
// table and their voxel records share cache lines when the points are spatial neighbours — what a sweeping sensor delivers.
// A 2 M-point cloud in arbitrary order (a merged map, a shuffled export) defeats that: measured on the wide map, every
// 48-byte probe row and 80-byte record pulls 128-byte lines nobody else in the wave shares, 2.9 GB of fabric traffic per
// launch against 0.58 GB algorithmic, 448 us instead of 228 (`wide_map`).  The reference has no counterpart
// (its per-point loop, NormalDistributionsTransform.cpp:391-445, is order-blind on a CPU cache hierarchy of a different
// shape); this is an MI355X-side measure.
//
//   1. a sample of consecutive point pairs decides whether the cloud is incoherent (most pairs further apart than 2 m);
//   2. if so: key = Morton code of the point's 2 m cell (x, y: 11 bits each) over 6 bits of z, STABLE radix sort of
//      (key, original index) pairs (rocPRIM through hipCUB), gather into a buffer the handle owns.
// The order inside a cell is the input order and the sort is deterministic, so every f64 sum stays reproducible run to run;
// against the unsorted cloud the result differs by summation order only (1e-11 relative, tests/test_gpu_edge_and_scale.py).
#include <hipcub/hipcub.hpp>

#include "ndt_internal.h"

#include <climits>

namespace {
constexpr int kBlock = 256;
constexpr float kCell = 2.0f;

__global__ __launch_bounds__(kBlock) void k_src_coherence(const float4 *__restrict__ src, size_t n, size_t stride, unsigned *__restrict__ out /* [0] far, [1] looked at */) {
  // pair (i, i + 1) for i = t * stride: 2^16 samples spread over the cloud
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t i = t * stride;
  unsigned far = 0, seen = 0;
  if (i + 1 < n) {
    float4 a = src[i], b = src[i + 1];
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float d2 = dx * dx + dy * dy + dz * dz;
    if (d2 == d2) { seen = 1; far = d2 > 4.0f ? 1u : 0u; }   // (pairs with a non-finite point do not vote)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { far += __shfl_down(far, off, 64); seen += __shfl_down(seen, off, 64); }
  if ((threadIdx.x & 63) == 0) { if (far) atomicAdd(out, far); if (seen) atomicAdd(out + 1, seen); }
}

__device__ __forceinline__ unsigned spread11(unsigned v) {  // 11 bits -> every second bit of 22
  v &= 0x7ffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ __launch_bounds__(kBlock) void k_src_keys(const float4 *__restrict__ src, size_t n, unsigned *__restrict__ keys, unsigned *__restrict__ idx) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = src[i];
  unsigned key = 0xffffffffu;   // non-finite points go last (the pass skips them wherever they are)
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    int ix = (int)floorf(p.x / kCell) + 1024, iy = (int)floorf(p.y / kCell) + 1024, iz = (int)floorf(p.z / kCell) + 32;
    key = ((spread11((unsigned)ix) | (spread11((unsigned)iy) << 1)) << 6) | ((unsigned)iz & 63u);   // (wraps beyond +-2 km: only the locality suffers)
  }
  keys[i] = key;
  idx[i] = (unsigned)i;
}

__global__ __launch_bounds__(kBlock) void k_src_gather(const float4 *__restrict__ src, const unsigned *__restrict__ idx, size_t n, float4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}
}  // namespace

// Decide (mode 1) or skip deciding (mode 2) and, if the cloud is incoherent, order it: on LSA_OK *d_use points to the cloud the
// passes should walk (the caller's, or the handle's sorted copy).  One host wait per call for clouds of >= kSortMinPoints points.
int ndt_source_order(lsa_ndt *h, const float4 *d_src, size_t n, const float4 **d_use) {
  lsa_context *ctx = h->ctx;
  *d_use = d_src;
  h->source_sorted = false;
  constexpr size_t kSortMinPoints = 262144;
  // (hipcub's segment-free radix sort takes an int count: clouds beyond INT_MAX points are walked as they come)
  if (h->source_sort_mode == 0 || n < kSortMinPoints || n > (size_t)INT_MAX) return LSA_OK;
  int rc;
  // A deferred final pass of the previous ScanMatch (ndt_api.hip) may still have to be REDONE on the host's say-so, on the cloud it
  // was posted for — which may be the sorted copy this call is about to overwrite: take its rows first.
  if ((rc = ndt_collect_deferred(h))) return rc;
  lsa_ctx_unpark(ctx);   // kernels are about to go onto the stream: nothing may be parked on it
  if ((rc = h->sort_small.reserve(ctx, 4))) return rc;
  if (h->source_sort_mode == 1) {
    const size_t samples = 65536, stride = n / samples;
    LSA_HIP(ctx, hipMemsetAsync(h->sort_small.p, 0, 2 * sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(k_src_coherence, dim3(lsa_div_up(samples, kBlock)), dim3(kBlock), 0, ctx->stream, d_src, n, stride, h->sort_small.p);
    unsigned res[2] = {0, 0};
    LSA_HIP(ctx, hipMemcpyAsync(res, h->sort_small.p, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h->unsynced = 0;
    if (res[1] == 0 || 2ull * res[0] <= res[1]) return LSA_OK;   // at least half of the consecutive pairs are neighbours: leave it
  }
  if ((rc = h->sort_keys.reserve(ctx, 2 * n, 1.1)) || (rc = h->sort_idx.reserve(ctx, 2 * n, 1.1)) || (rc = h->d_source_sorted.reserve(ctx, n, 1.1))) return rc;
  unsigned *keys_in = h->sort_keys.p, *keys_out = h->sort_keys.p + n, *idx_in = h->sort_idx.p, *idx_out = h->sort_idx.p + n;
  hipLaunchKernelGGL(k_src_keys, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_src, n, keys_in, idx_in);
  size_t tmp_bytes = 0;
  LSA_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0, 28, ctx->stream));
  if ((rc = h->sort_tmp.reserve(ctx, tmp_bytes + 256, 1.1))) return rc;
  LSA_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(h->sort_tmp.p, tmp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0, 28, ctx->stream));
  hipLaunchKernelGGL(k_src_gather, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_src, idx_out, n, h->d_source_sorted.p);
  LSA_HIP(ctx, hipGetLastError());
  *d_use = h->d_source_sorted.p;
  h->source_sorted = true;
  h->n_source_sorts++;
  return LSA_OK;
}

extern "C" int lsa_ndt_set_source_sort(lsa_ndt *h, int mode) {
  if (!h) return LSA_ERR_INVALID;
  if (mode < 0 || mode > 2) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_ndt_set_source_sort: mode %d (0 never, 1 when incoherent, 2 always)", mode);
  h->source_sort_mode = mode;
  return LSA_OK;
}

extern "C" int lsa_ndt_source_sort_count(lsa_ndt *h, uint64_t *count, int *last_was_sorted) {
  if (!h) return LSA_ERR_INVALID;
  if (count) *count = h->n_source_sorts;
  if (last_was_sorted) *last_was_sorted = h->source_sorted ? 1 : 0;
  return LSA_OK;
}
