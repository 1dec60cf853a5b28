// icp.hip — lsa_icp_*: the RegistrationInterface replacement for ICPRegistrationManual
// (reference: lidar_localization/src/models/registration/icp_registration_manual.cpp:12-179).
//
// Algorithm kept from the reference (calculateTrans, :125-178): for a FIXED number of iterations
//   transform the source with the current f32 pose (:132) -> exact 1-NN in the target (pcl::KdTreeFLANN, squared f32
//   distance) (:146) -> drop the pair if sqdist > max_correspond_distance_ (squared vs un-squared, sic, :147) ->
//   e = p' - q, J = [I3 | -R hat(p)] (:156-159) -> H += J^T J, B -= J^T e (:161-162) -> skip the update if det H == 0
//   (:165) -> delta = H^-1 B in f32 (:169) -> t += delta[0:3], R *= SO3::exp(delta[3:6]) (:171-176).
// The kd-tree becomes a uniform grid (cell = sqrt(gate) so the gated neighbour lies within +-1 cell) over the same
// sparse voxel index as the NDT table: target points are counting-sorted by cell rank once per SetInputTarget, and one
// lane per source point scans the 27 neighbouring cells.  J^T J / J^T e are formed from f32 J, e exactly as the
// reference forms them, but ACCUMULATED in f64 with a fixed reduction tree (the reference's running f32 sums depend on
// point order); the 6x6 solve and the pose update stay in f32 on the host.
#include "voxel_index_dev.h"

#include <cmath>

namespace {
constexpr int kBlock = 256;
constexpr int kOut = 32;

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

struct IcpParams {
  float T[12];   // row-major 3x4 current pose
  float gate;    // max_correspond_distance_ compared with the SQUARED distance (sic)
  CellMap map;
};

__global__ __launch_bounds__(kBlock) void k_icp_pass(const float4 *__restrict__ src, size_t n, IcpParams P, const uint32_t *__restrict__ bits,
                                                     const uint32_t *__restrict__ prefix, const uint32_t *__restrict__ start,
                                                     const float4 *__restrict__ sorted, double *__restrict__ partials) {
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = 0.0;
  unsigned ncorr = 0, npts = 0;
  const CellMap &m = P.map;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 s = src[i];
    if (!lsa_finite3(s.x, s.y, s.z)) continue;  // pcl::isFinite(ori_point) (:141)
    npts++;
    float qx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[0], s.x), lsa_fmul(P.T[1], s.y)), lsa_fmul(P.T[2], s.z)), P.T[3]);
    float qy = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[4], s.x), lsa_fmul(P.T[5], s.y)), lsa_fmul(P.T[6], s.z)), P.T[7]);
    float qz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(P.T[8], s.x), lsa_fmul(P.T[9], s.y)), lsa_fmul(P.T[10], s.z)), P.T[11]);
    if (!lsa_finite3(qx, qy, qz)) continue;
    int c[3];
    lsa_cell_ijk(m, qx, qy, qz, c);
    float best = INFINITY;
    unsigned best_id = 0xffffffffu;
    float bx = 0, by = 0, bz = 0;
    for (int dz = -1; dz <= 1; dz++) {
      int iz = c[2] + dz;
      if ((unsigned)iz >= (unsigned)m.dim[2]) continue;
      for (int dy = -1; dy <= 1; dy++) {
        int iy = c[1] + dy;
        if ((unsigned)iy >= (unsigned)m.dim[1]) continue;
        for (int dx = -1; dx <= 1; dx++) {
          int ix = c[0] + dx;
          if ((unsigned)ix >= (unsigned)m.dim[0]) continue;
          int cell = ix + iy * m.dim[0] + iz * m.dim[0] * m.dim[1];
          uint32_t w = bits[cell >> 5], bit = 1u << (cell & 31);
          if (!(w & bit)) continue;
          unsigned rank = prefix[cell >> 5] + __popc(w & (bit - 1u));
          unsigned a = start[rank], b = start[rank + 1];
          for (unsigned k = a; k < b; k++) {
            float4 t = sorted[k];
            float ex = lsa_fsub(qx, t.x), ey = lsa_fsub(qy, t.y), ez = lsa_fsub(qz, t.z);
            float d = lsa_fadd(lsa_fadd(lsa_fmul(ex, ex), lsa_fmul(ey, ey)), lsa_fmul(ez, ez));  // FLANN L2_Simple, f32
            unsigned id = __float_as_uint(t.w);
            if (d < best || (d == best && id < best_id)) { best = d; best_id = id; bx = t.x; by = t.y; bz = t.z; }
          }
        }
      }
    }
    if (best_id == 0xffffffffu || best > P.gate) continue;  // :147
    ncorr++;
    float e0 = lsa_fsub(qx, bx), e1 = lsa_fsub(qy, by), e2 = lsa_fsub(qz, bz);
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
    // B = -J^T e
    acc[0] -= ed[0]; acc[1] -= ed[1]; acc[2] -= ed[2];
#pragma unroll
    for (int a = 0; a < 3; a++) acc[3 + a] -= (double)A[0][a] * ed[0] + (double)A[1][a] * ed[1] + (double)A[2][a] * ed[2];
    // H = J^T J upper triangle: [I, A; A^T, A^T A]
    int k = 6;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) {
        double v;
        if (a < 3 && b < 3) v = (a == b) ? 1.0 : 0.0;
        else if (a < 3) v = (double)A[a][b - 3];
        else v = (double)A[0][a - 3] * A[0][b - 3] + (double)A[1][a - 3] * A[1][b - 3] + (double)A[2][a - 3] * A[2][b - 3];
        acc[k++] += v;
      }
  }
  __shared__ double sm[kBlock / 64][kOut];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; k++) {
    double v = lsa_wave_sum(acc[k]);
    if (lane == 0) sm[wave][1 + k] = v;
  }
  {
    double vc = lsa_wave_sum((double)ncorr), vn = lsa_wave_sum((double)npts);
    if (lane == 0) { sm[wave][0] = 0.0; sm[wave][28] = vc; sm[wave][29] = vn; sm[wave][30] = 0.0; sm[wave][31] = 0.0; }
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
    for (int w = 0; w < kBlock / 64; w++) v += sm[w][threadIdx.x];
    partials[(size_t)blockIdx.x * kOut + threadIdx.x] = v;
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

// f32 partial-pivot LU: determinant and inverse of the 6x6 (Eigen's Matrix<float,6,6>::determinant()/inverse() both go
// through PartialPivLU for sizes > 4)
float lu6_inverse(const float A[36], float inv[36]) {
  float M[6][12];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) { M[r][c] = A[r * 6 + c]; M[r][6 + c] = (r == c) ? 1.0f : 0.0f; }
  float det = 1.0f;
  for (int k = 0; k < 6; k++) {
    int piv = k;
    float mv = std::fabs(M[k][k]);
    for (int r = k + 1; r < 6; r++)
      if (std::fabs(M[r][k]) > mv) { mv = std::fabs(M[r][k]); piv = r; }
    if (mv == 0.0f) return 0.0f;
    if (piv != k) {
      for (int c = 0; c < 12; c++) std::swap(M[k][c], M[piv][c]);
      det = -det;
    }
    det *= M[k][k];
    float ip = 1.0f / M[k][k];
    for (int c = 0; c < 12; c++) M[k][c] *= ip;
    for (int r = 0; r < 6; r++)
      if (r != k) {
        float f = M[r][k];
        if (f != 0.0f)
          for (int c = 0; c < 12; c++) M[r][c] -= f * M[k][c];
      }
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) inv[r * 6 + c] = M[r][6 + c];
  return det;
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
  VoxelIndex vi;
  DevBuf<float4> d_target, d_sorted, d_source_own;
  DevBuf<uint32_t> cnt, start, fill;
  DevBuf<double> partials, d_out;
  PinnedBuf<double> h_out;
  PinnedBuf<float> h_stage;
  bool has_target = false;
  size_t n_target = 0;
};

namespace {
int icp_check(lsa_icp *h) {
  if (!h) return LSA_ERR_INVALID;
  hipError_t e = hipSetDevice(h->ctx->device);
  if (e != hipSuccess) return lsa_fail(h->ctx, LSA_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->ctx->device, hipGetErrorString(e));
  return LSA_OK;
}

int icp_upload(lsa_icp *h, const void *pts, size_t n, size_t stride, DevBuf<float4> &dst) {
  lsa_context *ctx = h->ctx;
  int rc;
  if ((rc = h->h_stage.reserve(ctx, n * 4))) return rc;
  if ((rc = dst.reserve(ctx, n, 1.25))) return rc;
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  lsa_pack_xyzw(pts, n, stride, SIZE_MAX, h->h_stage.p);
  LSA_HIP(ctx, hipMemcpyAsync(dst.p, h->h_stage.p, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  return LSA_OK;
}

int icp_build(lsa_icp *h, const float4 *d_pts, size_t n) {
  lsa_context *ctx = h->ctx;
  VoxelIndex &vi = h->vi;
  h->has_target = false;
  float cell = std::sqrt(h->max_corr);
  if (!(cell > 0.05f)) cell = 0.05f;
  const float leaf[3] = {cell, cell, cell};
  int rc, status = VI_OK;
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = vi_enqueue(ctx, vi, d_pts, n, /*mode NDT-style floor(x / cell)*/ 1, leaf, 4))) return rc;
    if ((rc = vi_fetch(ctx, vi))) return rc;
    if ((rc = vi_wait(ctx, vi, &status))) return rc;
    if (status != VI_CAPACITY) break;
    if ((rc = vi_grow(ctx, vi))) return rc;
  }
  if (status == VI_EMPTY) return lsa_fail(ctx, LSA_ERR_INVALID, "ICP SetInputTarget: no finite points");
  if (status != VI_OK) return lsa_fail(ctx, LSA_ERR_RANGE, "ICP SetInputTarget: neighbour grid overflows int32");
  size_t V = vi.nocc;
  if ((rc = h->cnt.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = h->start.reserve(ctx, V + 2, 1.25))) return rc;
  if ((rc = h->fill.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = h->d_sorted.reserve(ctx, n, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->cnt.p, 0, (V + 1) * sizeof(uint32_t), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(h->fill.p, 0, (V + 1) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_count, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, vi.cell_idx.p, n, vi.bits.p, vi.prefix.p, h->cnt.p);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, h->cnt.p, V, h->start.p);
  hipLaunchKernelGGL(k_scatter, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, d_pts, vi.cell_idx.p, n, vi.bits.p, vi.prefix.p, h->start.p,
                     h->fill.p, h->d_sorted.p);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  h->n_target = n;
  h->has_target = true;
  return LSA_OK;
}

int icp_match(lsa_icp *h, const float4 *d_src, size_t n, const float predict[16], float result[16]) {
  lsa_context *ctx = h->ctx;
  if (!h->has_target) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "ICP ScanMatch before SetInputTarget");
  float T[16];
  memcpy(T, predict, sizeof(T));
  float R[9], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r];
    t[r] = T[12 + r];
  }
  unsigned g = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 8);
  if (g < 1) g = 1;
  int rc;
  if ((rc = h->partials.reserve(ctx, (size_t)g * kOut, 1.0))) return rc;
  if ((rc = h->d_out.reserve(ctx, kOut))) return rc;
  if ((rc = h->h_out.reserve(ctx, kOut))) return rc;
  for (int it = 0; it < h->max_iter; it++) {
    IcpParams P;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) P.T[r * 4 + c] = R[r * 3 + c];
      P.T[r * 4 + 3] = t[r];
    }
    P.gate = h->max_corr;
    P.map = h->vi.map;
    hipLaunchKernelGGL(k_icp_pass, dim3(g), dim3(kBlock), 0, ctx->stream, d_src, n, P, h->vi.bits.p, h->vi.prefix.p, h->start.p, h->d_sorted.p, h->partials.p);
    hipLaunchKernelGGL(k_reduce32, dim3(1), dim3(1024), 0, ctx->stream, h->partials.p, (int)g, h->d_out.p);
    LSA_HIP(ctx, hipGetLastError());
    LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, h->d_out.p, kOut * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double *o = h->h_out.p;
    float H[36], B[6];
    int k = 7;
    for (int a = 0; a < 6; a++) B[a] = (float)o[1 + a];
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) { H[a * 6 + b] = H[b * 6 + a] = (float)o[k]; k++; }
    float Hi[36];
    float det = lu6_inverse(H, Hi);
    if (det == 0.0f) continue;  // :165
    float d[6];
    for (int a = 0; a < 6; a++) {
      float s = 0;
      for (int b = 0; b < 6; b++) s += Hi[a * 6 + b] * B[b];
      d[a] = s;
    }
    t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
    float dR[9], Rn[9];
    so3_exp(d + 3, dR);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Rn[r * 3 + c] = (R[r * 3 + 0] * dR[0 * 3 + c] + R[r * 3 + 1] * dR[1 * 3 + c]) + R[r * 3 + 2] * dR[2 * 3 + c];
    memcpy(R, Rn, sizeof(R));
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
    T[12 + r] = t[r];
  }
  memcpy(result, T, sizeof(T));
  return LSA_OK;
}
}  // namespace

extern "C" int lsa_icp_create(lsa_icp **out, lsa_context *ctx, float max_correspond_dis, int max_iter, int mode) {
  if (!out || !ctx) { lsa_set_global_error("lsa_icp_create: null argument"); return LSA_ERR_INVALID; }
  *out = nullptr;
  if (mode != 0) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_create: mode %d (point-to-plane) is not available in this build", mode);
  if (!(max_correspond_dis > 0) || max_iter < 0) return lsa_fail(ctx, LSA_ERR_INVALID, "lsa_icp_create: bad parameters");
  lsa_icp *h = new lsa_icp();
  h->ctx = ctx; h->max_corr = max_correspond_dis; h->max_iter = max_iter; h->mode = mode;
  *out = h;
  return LSA_OK;
}

extern "C" void lsa_icp_destroy(lsa_icp *h) {
  if (!h) return;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  h->vi.release(); h->d_target.release(); h->d_sorted.release(); h->d_source_own.release(); h->cnt.release(); h->start.release(); h->fill.release();
  h->partials.release(); h->d_out.release(); h->h_out.release(); h->h_stage.release();
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
  *score = 0.0f;  // "not implemented" in the reference: prints a notice and returns 0 (:62-67)
  return LSA_OK;
}
