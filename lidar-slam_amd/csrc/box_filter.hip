// box_filter.hip — lsa_box_filter_*: the CloudFilterInterface replacement for BoxFilter
// (reference: lidar_localization/src/models/cloud_filter/box_filter.cpp:14-73 -> pcl::CropBox<PointXYZI>, PCL 1.7
// filters/impl/crop_box.hpp, third-party: UNPINNED).  Callers: Matching::ResetLocalMap crops the global map around the
// vehicle (src/matching/matching.cpp:117-131), viewer.cpp crops the display map.
//
// Semantics kept: edge = size + origin per axis (box_filter.cpp:62-68); a point is kept when
// min <= p <= max on all three axes (inclusive both sides, as CropBox's `<` / `>` rejections leave it), non-finite points
// are dropped (CropBox skips them when the cloud is not dense), every field is copied, and the output keeps the INPUT
// ORDER — so this is an order-preserving stream compaction: per-workgroup counts -> one-workgroup exclusive scan ->
// ranked scatter, 16 B read + 16 B written per kept point, no atomics.
#include "voxel_index_dev.h"

namespace {
constexpr int kBlock = 256;
constexpr int kItems = 8;  // points per thread, consecutive -> a workgroup owns 2048 consecutive points

struct Box { float mn[3], mx[3]; };

__device__ __forceinline__ bool box_keep(const Box &b, float4 p) {
  if (!lsa_finite3(p.x, p.y, p.z)) return false;
  return !((p.x < b.mn[0] || p.y < b.mn[1] || p.z < b.mn[2]) || (p.x > b.mx[0] || p.y > b.mx[1] || p.z > b.mx[2]));
}

__global__ __launch_bounds__(kBlock) void k_box_count(const float4 *__restrict__ in, size_t n, Box b, uint32_t *__restrict__ block_cnt) {
  size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
  unsigned c = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++)
    if (base + k < n && box_keep(b, in[base + k])) c++;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  __shared__ unsigned sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// one workgroup: exclusive scan of the block counts (in place) and the total (also mirrored to host-mapped memory)
__global__ __launch_bounds__(1024) void k_box_scan(uint32_t *__restrict__ block_cnt, unsigned nb, unsigned long long *__restrict__ total,
                                                   unsigned long long *__restrict__ total_host) {
  __shared__ unsigned s[1024];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < nb; base += 1024) {
    unsigned i = base + threadIdx.x;
    unsigned v = (i < nb) ? block_cnt[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    unsigned incl = s[threadIdx.x], c = carry;
    if (i < nb) block_cnt[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { *total = carry; *total_host = carry; }
}

__global__ __launch_bounds__(kBlock) void k_box_scatter(const float4 *__restrict__ in, size_t n, Box b, const uint32_t *__restrict__ block_off,
                                                        float4 *__restrict__ out) {
  size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
  float4 p[kItems];
  unsigned keep = 0, c = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    if (base + k < n) {
      p[k] = in[base + k];
      if (box_keep(b, p[k])) { keep |= 1u << k; c++; }
    }
  }
  // exclusive scan of the per-thread counts over the workgroup: wave scan + wave offsets
  unsigned incl = c;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __shared__ unsigned wsum[kBlock / 64];
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned woff = 0;
  for (int w = 0; w < wave; w++) woff += wsum[w];
  size_t dst = (size_t)block_off[blockIdx.x] + woff + (incl - c);
#pragma unroll
  for (int k = 0; k < kItems; k++)
    if (keep & (1u << k)) out[dst++] = p[k];
}
}  // namespace

struct lsa_box_filter {
  lsa_context *ctx = nullptr;
  float size[6] = {0, 0, 0, 0, 0, 0};
  float origin[3] = {0, 0, 0};
  float edge[6] = {0, 0, 0, 0, 0, 0};
  DevBuf<float4> d_in, d_out;
  DevBuf<uint32_t> block_cnt;
  DevBuf<unsigned long long> d_total;
  PinnedBuf<unsigned long long> h_total;
  PinnedBuf<float> h_stage;
};

namespace {
void calc_edge(lsa_box_filter *f) {  // box_filter.cpp:62-68
  for (int i = 0; i < 3; i++) {
    f->edge[2 * i] = f->size[2 * i] + f->origin[i];
    f->edge[2 * i + 1] = f->size[2 * i + 1] + f->origin[i];
  }
}

int box_device(lsa_box_filter *f, const float4 *d_in, size_t n, float4 *d_out, size_t *n_out) {
  lsa_context *ctx = f->ctx;
  *n_out = 0;
  if (!n) return LSA_OK;
  unsigned nb = lsa_div_up(n, (size_t)kBlock * kItems);
  int rc;
  if ((rc = f->block_cnt.reserve(ctx, nb, 1.25))) return rc;
  if ((rc = f->d_total.reserve(ctx, 1))) return rc;
  if ((rc = f->h_total.reserve(ctx, 1))) return rc;
  Box b;
  for (int a = 0; a < 3; a++) { b.mn[a] = f->edge[2 * a]; b.mx[a] = f->edge[2 * a + 1]; }
  hipLaunchKernelGGL(k_box_count, dim3(nb), dim3(kBlock), 0, ctx->stream, d_in, n, b, f->block_cnt.p);
  hipLaunchKernelGGL(k_box_scan, dim3(1), dim3(1024), 0, ctx->stream, f->block_cnt.p, nb, f->d_total.p, f->h_total.p);
  hipLaunchKernelGGL(k_box_scatter, dim3(nb), dim3(kBlock), 0, ctx->stream, d_in, n, b, f->block_cnt.p, d_out);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_out = (size_t)f->h_total.p[0];
  return LSA_OK;
}
}  // namespace

extern "C" int lsa_box_filter_create(lsa_box_filter **out, lsa_context *ctx, const float box_filter_size[6]) {
  if (!out || !ctx || !box_filter_size) { lsa_set_global_error("lsa_box_filter_create: null argument"); return LSA_ERR_INVALID; }
  lsa_box_filter *f = new lsa_box_filter();
  f->ctx = ctx;
  memcpy(f->size, box_filter_size, sizeof(f->size));
  calc_edge(f);
  *out = f;
  return LSA_OK;
}

extern "C" void lsa_box_filter_destroy(lsa_box_filter *f) {
  if (!f) return;
  (void)lsa_enter(f->ctx);
  (void)hipStreamSynchronize(f->ctx->stream);
  f->d_in.release(); f->d_out.release(); f->block_cnt.release(); f->d_total.release(); f->h_total.release(); f->h_stage.release();
  delete f;
}

extern "C" int lsa_box_filter_set_size(lsa_box_filter *f, const float box_filter_size[6]) {
  if (!f || !box_filter_size) return LSA_ERR_INVALID;
  memcpy(f->size, box_filter_size, sizeof(f->size));
  calc_edge(f);
  return LSA_OK;
}

extern "C" int lsa_box_filter_set_origin(lsa_box_filter *f, const float origin[3]) {
  if (!f || !origin) return LSA_ERR_INVALID;
  memcpy(f->origin, origin, sizeof(f->origin));
  calc_edge(f);
  return LSA_OK;
}

extern "C" int lsa_box_filter_get_edge(lsa_box_filter *f, float edge[6]) {
  if (!f || !edge) return LSA_ERR_INVALID;
  memcpy(edge, f->edge, sizeof(f->edge));
  return LSA_OK;
}

extern "C" int lsa_box_filter_apply_device(lsa_box_filter *f, const float *d_in_xyzw, size_t n, float *d_out_xyzw, size_t *n_out) {
  if (!f || !n_out || (n && (!d_in_xyzw || !d_out_xyzw))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_box_filter_apply_device: null argument") : LSA_ERR_INVALID;
  if (n && d_in_xyzw == d_out_xyzw) return lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_box_filter_apply_device: in-place not supported on device buffers");
  LSA_HIP(f->ctx, lsa_enter(f->ctx));
  return box_device(f, (const float4 *)d_in_xyzw, n, (float4 *)d_out_xyzw, n_out);
}

extern "C" int lsa_box_filter_apply(lsa_box_filter *f, const void *in_pts, size_t n, size_t in_stride, size_t in_w_off, void *out_pts, size_t out_cap,
                                    size_t out_stride, size_t out_w_off, size_t *n_out) {
  if (!f || !n_out || (n && (!in_pts || !out_pts))) return f ? lsa_fail(f->ctx, LSA_ERR_INVALID, "lsa_box_filter_apply: null argument") : LSA_ERR_INVALID;
  lsa_context *ctx = f->ctx;
  *n_out = 0;
  if (n == 0) return LSA_OK;
  if (in_stride < 12 || out_stride < 12) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_box_filter_apply: stride < 12");
  LSA_HIP(ctx, lsa_enter(ctx));
  int rc;
  // the staging buffer may still be feeding an earlier async copy, and growing a buffer frees the old one: wait first
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = f->h_stage.reserve(ctx, n * 4))) return rc;
  if ((rc = f->d_in.reserve(ctx, n, 1.25))) return rc;
  if ((rc = f->d_out.reserve(ctx, n, 1.25))) return rc;
  if ((rc = lsa_upload_xyzw(ctx, in_pts, n, in_stride, in_w_off, f->h_stage.p, f->d_in.p))) return rc;
  size_t V = 0;
  if ((rc = box_device(f, f->d_in.p, n, f->d_out.p, &V))) return rc;
  if (V > out_cap) return lsa_fail(ctx, LSA_ERR_CAPACITY, "lsa_box_filter_apply: out_cap %zu < %zu points", out_cap, V);
  if (V) LSA_HIP(ctx, hipMemcpyAsync(f->h_stage.p, f->d_out.p, V * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < V; i++) {  // the input was consumed into the staging buffer first, so out_pts == in_pts is fine
    const float *p = f->h_stage.p + 4 * i;
    char *o = (char *)out_pts + i * out_stride;
    memcpy(o, p, 12);
    if (out_stride >= 16 && out_w_off != 12) ((float *)o)[3] = 1.0f;
    if (out_w_off != SIZE_MAX && out_w_off + 4 <= out_stride) memcpy(o + out_w_off, p + 3, 4);
  }
  *n_out = V;
  return LSA_OK;
}
