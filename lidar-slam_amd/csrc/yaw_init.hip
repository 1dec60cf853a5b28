// yaw_init.hip — lsa_yaw_init_*: Matching's position-only initialisation on the device
// (reference: lidar_localization/src/matching/matching.cpp — generateGauss2DMapCells :344-394, resetMapRange :396-422,
//  getInitialYawAngle :267-307; `init_type: OnlyPosition` is the default of config/matching/matching.yaml:4).
//
// The local map (centred on the GNSS position) is binned into a 2-D grid of `grid_resolution` cells, each holding a
// running mean / "variance" of the heights that fell into it; the current scan is rotated through 270 yaw hypotheses
// and each is scored by sum exp(-(z - mu)^2 / (2 sigma)) over the cells its points land in.
//
// What has to be kept from the text to reproduce its numbers:
//   * the per-cell statistics are a SEQUENTIAL mixed float / double recurrence in input order (:375-391), not a closed
//     form — so the points of a cell are brought into input order first (count -> scan -> scatter -> rank-by-index), then
//     one thread per cell replays the recurrence;
//   * cell index = round((float difference) / (double resolution)), width / height likewise (:286-287, :357-358);
//   * the 270 rotations are Eigen::AngleAxisf(float(2 pi / 270) * i, UnitZ) applied by pcl::transformPointCloud in f32;
//     the host builds that table with the same libm sinf / cosf the reference calls;
//   * scores accumulate in double; the arg-max compares them with a FLOAT running maximum (:271, :302-305).
// The 270 x N scoring is the parallel part: one workgroup per (rotation, chunk of points), fixed-order reductions.
#include "voxel_index_dev.h"

#include <cfloat>
#include <cmath>
#include <vector>

namespace {
constexpr int kBlock = 256;
constexpr int kAngles = 270;
constexpr int kChunk = kBlock * 8;  // points per scoring workgroup

struct Origin { float v[3]; };

__global__ __launch_bounds__(kBlock) void k_yaw_bbox(const float4 *__restrict__ pts, size_t n, Origin o, float *__restrict__ partial) {
  float mn0 = FLT_MAX, mn1 = FLT_MAX, mn2 = FLT_MAX, mx0 = -FLT_MAX, mx1 = -FLT_MAX, mx2 = -FLT_MAX;
  unsigned cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (!lsa_finite3(p.x, p.y, p.z)) continue;
    float cx = lsa_fsub(p.x, o.v[0]), cy = lsa_fsub(p.y, o.v[1]), cz = lsa_fsub(p.z, o.v[2]);
    // resetMapRange: strict comparisons against running extrema (:398-421) == min / max
    mn0 = fminf(mn0, cx); mn1 = fminf(mn1, cy); mn2 = fminf(mn2, cz);
    mx0 = fmaxf(mx0, cx); mx1 = fmaxf(mx1, cy); mx2 = fmaxf(mx2, cz);
    cnt++;
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
    float *out = partial + (size_t)blockIdx.x * 8;
    out[0] = mn0; out[1] = mn1; out[2] = mn2; out[3] = mx0; out[4] = mx1; out[5] = mx2; out[6] = __uint_as_float(c); out[7] = 0.f;
  }
}

struct Grid2D {
  float mn0, mn1;
  double res;
  int width, height;
};

__device__ __forceinline__ int yaw_cell(const Grid2D &g, float x, float y) {
  // int cell_x = std::round((point.x - map_min_xyz_(0)) / grid_map_resolution_)   (float difference, double division)
  int cx = (int)round((double)lsa_fsub(x, g.mn0) / g.res);
  int cy = (int)round((double)lsa_fsub(y, g.mn1) / g.res);
  if (cx < 0 || cy < 0 || cx >= g.width || cy >= g.height) return -1;
  return cx * g.height + cy;
}

__global__ __launch_bounds__(kBlock) void k_yaw_count(const float4 *__restrict__ pts, size_t n, Origin o, Grid2D g, int32_t *__restrict__ cell_of, uint32_t *__restrict__ count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  int c = -1;
  if (lsa_finite3(p.x, p.y, p.z)) c = yaw_cell(g, lsa_fsub(p.x, o.v[0]), lsa_fsub(p.y, o.v[1]));
  cell_of[i] = c;
  if (c >= 0) atomicAdd(count + c, 1u);
}

// single workgroup: start[i] = exclusive prefix of count[0..n), start[n] = total
__global__ __launch_bounds__(1024) void k_yaw_scan(const uint32_t *__restrict__ count, size_t n, uint32_t *__restrict__ start) {
  __shared__ unsigned s[1024];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n; base += 1024) {
    size_t i = base + threadIdx.x;
    unsigned v = (i < n) ? count[i] : 0;
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

__global__ __launch_bounds__(kBlock) void k_yaw_scatter(const float4 *__restrict__ pts, const int32_t *__restrict__ cell_of, size_t n, Origin o,
                                                        const uint32_t *__restrict__ start, uint32_t *__restrict__ fill, float2 *__restrict__ slots) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = cell_of[i];
  if (c < 0) return;
  unsigned s = start[c] + atomicAdd(fill + c, 1u);
  slots[s] = make_float2(lsa_fsub(pts[i].z, o.v[2]), __uint_as_float((unsigned)i));  // centred height, original index
}

// bring every cell's points into INPUT order: position = number of points of the same cell with a smaller index
__global__ __launch_bounds__(kBlock) void k_yaw_order(const float2 *__restrict__ slots, const int32_t *__restrict__ cell_of, size_t n,
                                                      const uint32_t *__restrict__ start, float *__restrict__ ordered_z) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = cell_of[i];
  if (c < 0) return;
  unsigned a = start[c], b = start[c + 1];
  unsigned pos = 0;
  float z = 0.f;
  for (unsigned k = a; k < b; k++) {
    float2 e = slots[k];
    unsigned idx = __float_as_uint(e.y);
    pos += idx < (unsigned)i ? 1u : 0u;
    if (idx == (unsigned)i) z = e.x;
  }
  ordered_z[a + pos] = z;
}

// one thread per cell replays the reference's running statistics over the cell's points in input order (:375-391)
__global__ __launch_bounds__(kBlock) void k_yaw_cells(const float *__restrict__ ordered_z, const uint32_t *__restrict__ start, size_t ncells,
                                                      float *__restrict__ mu_out, float *__restrict__ sigma_out, int32_t *__restrict__ cnt_out) {
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  unsigned a = start[c], b = start[c + 1];
  float mu = 0.0f, sigma = 0.0f;
  int point_cnt = 0;
  for (unsigned k = a; k < b; k++) {
    float z = ordered_z[k];
    if (point_cnt == 0) {
      mu = z; sigma = 0.0f; point_cnt = 1;
    } else {
      float new_mu = lsa_fdiv(lsa_fadd(lsa_fmul((float)point_cnt, mu), z), (float)(point_cnt + 1));
      // (cnt - 1) * sigma [float] + pow(z - mu, 2) [double] + (cnt + 1) * pow(new_mu - mu, 2) [double]
      //   + 2 * (new_mu - mu) * (z - new_mu) [float], summed left to right in double, stored to float, then /= cnt
      float t0 = lsa_fmul((float)(point_cnt - 1), sigma);
      float d1 = lsa_fsub(z, mu), d2 = lsa_fsub(new_mu, mu);
      double acc = __dadd_rn((double)t0, __dmul_rn((double)d1, (double)d1));
      acc = __dadd_rn(acc, __dmul_rn((double)(point_cnt + 1), __dmul_rn((double)d2, (double)d2)));
      float t3 = lsa_fmul(lsa_fmul(2.0f, d2), lsa_fsub(z, new_mu));
      acc = __dadd_rn(acc, (double)t3);
      sigma = lsa_fdiv((float)acc, (float)point_cnt);
      mu = new_mu;
      point_cnt++;
    }
  }
  mu_out[c] = mu; sigma_out[c] = sigma; cnt_out[c] = point_cnt;
}

// score: workgroup (chunk, rotation) -> partial[rotation][chunk]; fixed-order reductions
__global__ __launch_bounds__(kBlock) void k_yaw_score(const float4 *__restrict__ scan, size_t n, const float4 *__restrict__ rot /* {c, s, d, 0} */, Grid2D g,
                                                      const float *__restrict__ mu, const float *__restrict__ sigma, const int32_t *__restrict__ cnt,
                                                      double *__restrict__ partial, unsigned nchunks) {
  const float4 r = rot[blockIdx.y];
  const float c = r.x, s = r.y, d = r.z, ns = -r.y;
  double sum = 0.0;
  size_t base = (size_t)blockIdx.x * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; k++) {
    size_t j = base + (size_t)k * kBlock + threadIdx.x;
    if (j >= n) break;
    float4 p = scan[j];
    // pcl::transformPointCloud with [[c, -s, 0, 0], [s, c, 0, 0], [0, 0, d, 0]]: ((m0 x + m1 y) + m2 z) + m3, f32, no FMA
    float qx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(c, p.x), lsa_fmul(ns, p.y)), lsa_fmul(0.0f, p.z)), 0.0f);
    float qy = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(s, p.x), lsa_fmul(c, p.y)), lsa_fmul(0.0f, p.z)), 0.0f);
    float qz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(0.0f, p.x), lsa_fmul(0.0f, p.y)), lsa_fmul(d, p.z)), 0.0f);
    if (!lsa_finite3(qx, qy, qz)) continue;
    int cell = yaw_cell(g, qx, qy);
    if (cell < 0) continue;
    if (cnt[cell] == 0) continue;
    float dz = lsa_fsub(qz, mu[cell]);
    float two_sigma = lsa_fmul(2.0f, sigma[cell]);
    sum += exp(-((double)dz * (double)dz) / (double)two_sigma);  // exp(-pow(z - mu, 2) / (2 * sigma)) (:296)
  }
  __shared__ double sm[kBlock];
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * nchunks + blockIdx.x] = sm[0];
}
}  // namespace

struct lsa_yaw_init {
  lsa_context *ctx = nullptr;
  double res = 0.8;
  Grid2D grid{};
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool built = false;
  DevBuf<float4> d_pts, d_rot;
  DevBuf<float> scratch, ordered_z, mu, sigma;
  DevBuf<int32_t> cell_of, cnt;
  DevBuf<uint32_t> count, start, fill;
  DevBuf<float2> slots;
  DevBuf<double> partial;
  PinnedBuf<float> h_stage;
  bool rot_ready = false;
};

namespace {
int yaw_upload(lsa_yaw_init *h, const void *pts, size_t n, size_t stride) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->h_stage.reserve(ctx, (n ? n : 1) * 4))) return rc;
  if ((rc = h->d_pts.reserve(ctx, n ? n : 1, 1.25))) return rc;
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return lsa_upload_xyzw(ctx, pts, n, stride, SIZE_MAX, h->h_stage.p, h->d_pts.p);
}

int yaw_build(lsa_yaw_init *h, const float4 *d_map, size_t n, const float origin[3]) {
  lsa_context *ctx = h->ctx;
  h->built = false;
  Origin o{{origin[0], origin[1], origin[2]}};
  int rc;
  unsigned nb = (unsigned)std::min<size_t>(lsa_div_up(n ? n : 1, kBlock), (size_t)ctx->num_cus * 8);
  if ((rc = h->scratch.reserve(ctx, (size_t)nb * 8))) return rc;
  hipLaunchKernelGGL(k_yaw_bbox, dim3(nb), dim3(kBlock), 0, ctx->stream, d_map, n, o, h->scratch.p);
  LSA_HIP(ctx, hipGetLastError());
  std::vector<float> part((size_t)nb * 8);
  LSA_HIP(ctx, hipMemcpyAsync(part.data(), h->scratch.p, part.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // SetInitPose resets the range to (+max, -max) before every rebuild (:331-334)
  for (int a = 0; a < 3; a++) { h->mn[a] = FLT_MAX; h->mx[a] = -FLT_MAX; }
  unsigned long long nfinite = 0;
  for (unsigned b = 0; b < nb; b++) {
    const float *p = part.data() + (size_t)b * 8;
    for (int a = 0; a < 3; a++) { h->mn[a] = std::fmin(h->mn[a], p[a]); h->mx[a] = std::fmax(h->mx[a], p[3 + a]); }
    unsigned c;
    memcpy(&c, p + 6, 4);
    nfinite += c;
  }
  Grid2D g;
  g.mn0 = h->mn[0]; g.mn1 = h->mn[1]; g.res = h->res;
  g.width = g.height = 0;
  if (nfinite) {
    g.width = (int)std::round((double)(h->mx[0] - h->mn[0]) / h->res);   // :357-358
    g.height = (int)std::round((double)(h->mx[1] - h->mn[1]) / h->res);
    if (g.width < 0 || g.height < 0) g.width = g.height = 0;
  }
  if ((double)g.width * (double)g.height > 2.0e9) return lsa_fail(ctx, LSA_ERR_RANGE, "yaw init: %d x %d cells overflow int32", g.width, g.height);
  h->grid = g;
  size_t nc = (size_t)g.width * (size_t)g.height;
  if ((rc = h->mu.reserve(ctx, nc + 1, 1.25))) return rc;
  if ((rc = h->sigma.reserve(ctx, nc + 1, 1.25))) return rc;
  if ((rc = h->cnt.reserve(ctx, nc + 1, 1.25))) return rc;
  if (nc == 0 || n == 0) { h->built = true; return LSA_OK; }
  if ((rc = h->count.reserve(ctx, nc + 1, 1.25))) return rc;
  if ((rc = h->start.reserve(ctx, nc + 2, 1.25))) return rc;
  if ((rc = h->fill.reserve(ctx, nc + 1, 1.25))) return rc;
  if ((rc = h->cell_of.reserve(ctx, n, 1.25))) return rc;
  if ((rc = h->slots.reserve(ctx, n, 1.25))) return rc;
  if ((rc = h->ordered_z.reserve(ctx, n, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->count.p, 0, (nc + 1) * sizeof(uint32_t), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(h->fill.p, 0, (nc + 1) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_yaw_count, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_map, n, o, g, h->cell_of.p, h->count.p);
  hipLaunchKernelGGL(k_yaw_scan, dim3(1), dim3(1024), 0, ctx->stream, h->count.p, nc, h->start.p);
  hipLaunchKernelGGL(k_yaw_scatter, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_map, h->cell_of.p, n, o, h->start.p, h->fill.p, h->slots.p);
  hipLaunchKernelGGL(k_yaw_order, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, h->slots.p, h->cell_of.p, n, h->start.p, h->ordered_z.p);
  hipLaunchKernelGGL(k_yaw_cells, dim3(lsa_div_up(nc, kBlock)), dim3(kBlock), 0, ctx->stream, h->ordered_z.p, h->start.p, nc, h->mu.p, h->sigma.p, h->cnt.p);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  h->built = true;
  return LSA_OK;
}

int yaw_rotations(lsa_yaw_init *h) {
  if (h->rot_ready) return LSA_OK;
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->d_rot.reserve(ctx, kAngles))) return rc;
  std::vector<float> r((size_t)kAngles * 4);
  const float delta_angle = (float)(2 * M_PI / kAngles);  // float delta_angle = 2 * M_PI / angle_size (:270)
  for (int i = 0; i < kAngles; i++) {
    // Eigen::AngleAxisf(delta_angle * i, UnitZ).matrix() (Eigen 3.2.92 AngleAxis.h): diag (c, c, (1 - c) * 1 + c), off-diag -s / s
    float angle = delta_angle * (float)i;
    float s = std::sin(angle), c = std::cos(angle);
    r[4 * i] = c; r[4 * i + 1] = s; r[4 * i + 2] = (1.0f - c) * 1.0f + c; r[4 * i + 3] = 0.f;
  }
  LSA_HIP(ctx, hipMemcpyAsync(h->d_rot.p, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  h->rot_ready = true;
  return LSA_OK;
}

int yaw_score(lsa_yaw_init *h, const float4 *d_scan, size_t n, double *yaw, double *probs270) {
  lsa_context *ctx = h->ctx;
  if (!h->built) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "GetInitialYawAngle before the local map was set");
  int rc;
  if ((rc = yaw_rotations(h))) return rc;
  double probs[kAngles];
  for (int i = 0; i < kAngles; i++) probs[i] = 0.0;
  size_t nc = (size_t)h->grid.width * (size_t)h->grid.height;
  if (n && nc) {
    unsigned nchunks = lsa_div_up(n, kChunk);
    if ((rc = h->partial.reserve(ctx, (size_t)kAngles * nchunks))) return rc;
    hipLaunchKernelGGL(k_yaw_score, dim3(nchunks, kAngles), dim3(kBlock), 0, ctx->stream, d_scan, n, h->d_rot.p, h->grid, h->mu.p, h->sigma.p, h->cnt.p,
                       h->partial.p, nchunks);
    LSA_HIP(ctx, hipGetLastError());
    std::vector<double> part((size_t)kAngles * nchunks);
    LSA_HIP(ctx, hipMemcpyAsync(part.data(), h->partial.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < kAngles; i++) {
      double s = 0.0;
      for (unsigned c = 0; c < nchunks; c++) s += part[(size_t)i * nchunks + c];  // fixed order
      probs[i] = s;
    }
  }
  // :299-306: float running maximum, strict '>'
  const float delta_angle = (float)(2 * M_PI / kAngles);
  float max_prob = -FLT_MAX;
  double max_prob_angle = 0;
  for (int it = 0; it < kAngles; it++) {
    if (probs[it] > (double)max_prob) {
      max_prob = (float)probs[it];
      max_prob_angle = (double)((float)it * delta_angle);
    }
  }
  if (yaw) *yaw = max_prob_angle;
  if (probs270) memcpy(probs270, probs, sizeof(probs));
  return LSA_OK;
}

int yaw_check(lsa_yaw_init *h) {
  if (!h) return LSA_ERR_INVALID;
  hipError_t e = lsa_enter(h->ctx);
  if (e != hipSuccess) return lsa_fail(h->ctx, LSA_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->ctx->device, hipGetErrorString(e));
  return LSA_OK;
}
}  // namespace

extern "C" int lsa_yaw_init_create(lsa_yaw_init **out, lsa_context *ctx, double grid_resolution) {
  if (!out || !ctx) { lsa_set_global_error("lsa_yaw_init_create: null argument"); return LSA_ERR_INVALID; }
  lsa_yaw_init *h = new lsa_yaw_init();
  h->ctx = ctx;
  h->res = std::max(0.1, grid_resolution);  // InitGridMapResolution (:137-138)
  *out = h;
  return LSA_OK;
}

extern "C" void lsa_yaw_init_destroy(lsa_yaw_init *h) {
  if (!h) return;
  (void)lsa_enter(h->ctx);
  (void)hipStreamSynchronize(h->ctx->stream);
  h->d_pts.release(); h->d_rot.release(); h->scratch.release(); h->ordered_z.release(); h->mu.release(); h->sigma.release();
  h->cell_of.release(); h->cnt.release(); h->count.release(); h->start.release(); h->fill.release(); h->slots.release();
  h->partial.release(); h->h_stage.release();
  delete h;
}

extern "C" int lsa_yaw_init_set_local_map_device(lsa_yaw_init *h, const float *d_xyzw, size_t n, const float origin[3]) {
  int rc = yaw_check(h);
  if (rc) return rc;
  if (!origin || (n && !d_xyzw)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_yaw_init_set_local_map: null argument");
  return yaw_build(h, (const float4 *)d_xyzw, n, origin);
}

extern "C" int lsa_yaw_init_set_local_map(lsa_yaw_init *h, const void *pts, size_t n, size_t stride, const float origin[3]) {
  int rc = yaw_check(h);
  if (rc) return rc;
  if (!origin || (n && (!pts || stride < 12))) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_yaw_init_set_local_map: bad argument");
  if ((rc = yaw_upload(h, pts, n, stride))) return rc;
  return yaw_build(h, h->d_pts.p, n, origin);
}

extern "C" int lsa_yaw_init_get_initial_yaw_device(lsa_yaw_init *h, const float *d_scan_xyzw, size_t n, double *yaw, double *probs270) {
  int rc = yaw_check(h);
  if (rc) return rc;
  if (n && !d_scan_xyzw) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_yaw_init_get_initial_yaw: null scan");
  return yaw_score(h, (const float4 *)d_scan_xyzw, n, yaw, probs270);
}

extern "C" int lsa_yaw_init_get_initial_yaw(lsa_yaw_init *h, const void *scan, size_t n, size_t stride, double *yaw, double *probs270) {
  int rc = yaw_check(h);
  if (rc) return rc;
  if (n && (!scan || stride < 12)) return lsa_fail(h->ctx, LSA_ERR_INVALID, "lsa_yaw_init_get_initial_yaw: bad argument");
  if (!h->built) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "GetInitialYawAngle before the local map was set");
  // the scan is staged behind the map's device copy only when that copy is the caller's (device path); here the map
  // statistics are already reduced to the cell tables, so the staging buffer can be reused
  if ((rc = yaw_upload(h, scan, n, stride))) return rc;
  return yaw_score(h, h->d_pts.p, n, yaw, probs270);
}

extern "C" int lsa_yaw_init_grid_info(lsa_yaw_init *h, int32_t wh[2], float min_xyz[3], float max_xyz[3]) {
  if (!h || !wh) return LSA_ERR_INVALID;
  if (!h->built) return lsa_fail(h->ctx, LSA_ERR_NO_TARGET, "grid_info before the local map was set");
  wh[0] = h->grid.width; wh[1] = h->grid.height;
  if (min_xyz) memcpy(min_xyz, h->mn, sizeof(h->mn));
  if (max_xyz) memcpy(max_xyz, h->mx, sizeof(h->mx));
  return LSA_OK;
}

extern "C" int lsa_yaw_init_cells(lsa_yaw_init *h, float *mu, float *sigma, int32_t *cnt, size_t cap) {
  int rc = yaw_check(h);
  if (rc) return rc;
  lsa_context *ctx = h->ctx;
  if (!h->built) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "cells before the local map was set");
  size_t nc = (size_t)h->grid.width * (size_t)h->grid.height;
  if (cap < nc) return lsa_fail(ctx, LSA_ERR_CAPACITY, "lsa_yaw_init_cells: cap %zu < %zu cells", cap, nc);
  if (!nc) return LSA_OK;
  if (mu) LSA_HIP(ctx, hipMemcpyAsync(mu, h->mu.p, nc * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (sigma) LSA_HIP(ctx, hipMemcpyAsync(sigma, h->sigma.p, nc * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (cnt) LSA_HIP(ctx, hipMemcpyAsync(cnt, h->cnt.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}
