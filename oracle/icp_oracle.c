/*
 * icp_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lsa_oracle.h header).
 *
 * Restates ICPRegistrationManual (point-to-point Gauss-Newton, all f32):
 *   S/models/registration/icp_registration_manual.cpp:45-60 (ScanMatch), :125-178 (calculateTrans)
 * Third-party pieces it calls and that are absent from /root/reference (pins: CMake only,
 * lidar_localization/cmake/PCL.cmake:1; sophus via "sophus/se3.hpp", no version):
 *   pcl::KdTreeFLANN::nearestKSearch(k=1)  -> exact 1-NN, SQUARED L2 distance in f32
 *   Sophus::SO3f::hat / exp                -> textbook hat; exp via unit quaternion
 * In the tree and PINNED bit for bit (vendored Eigen 3.2.92 compiled in place, oracle/_ref; eigen_f32_oracle.c):
 *   Eigen 6x6 f32 determinant / inverse / inverse * B, Matrix3f products, the per-point J / J^T J / J^T e expressions
 * Quirk kept (:147): the SQUARED NN distance is compared with the un-squared threshold.
 * PARITY STATUS: the Eigen arithmetic is pinned; the loop around it, the kd-tree's tie order and SO3::exp are restated
 * from text and UNPINNED (tests compare with a float tolerance, see tests/test_gpu_icp_fitness.py).
 */
#include "lsa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float cell;
  int lo[3], dim[3];
  int *start; /* dim0*dim1*dim2 + 1 */
  int *idx;   /* point ids sorted by cell */
  const void *pts; size_t stride; size_t n;
} nn_grid;

static inline const float *P(const void *pts, size_t stride, size_t i) { return (const float *)((const char *)pts + i * stride); }

static nn_grid *nn_build(const void *pts, size_t n, size_t stride, float cell) {
  nn_grid *g = (nn_grid *)calloc(1, sizeof(nn_grid));
  g->cell = cell; g->pts = pts; g->stride = stride; g->n = n;
  int mn[3] = {2147483647, 2147483647, 2147483647}, mx[3] = {-2147483647, -2147483647, -2147483647};
  for (size_t i = 0; i < n; i++) {
    const float *p = P(pts, stride, i);
    for (int a = 0; a < 3; a++) { int c = (int)floorf(p[a] / cell); if (c < mn[a]) mn[a] = c; if (c > mx[a]) mx[a] = c; }
  }
  for (int a = 0; a < 3; a++) { g->lo[a] = mn[a]; g->dim[a] = mx[a] - mn[a] + 1; }
  size_t nc = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
  g->start = (int *)calloc(nc + 1, sizeof(int));
  g->idx = (int *)malloc(n * sizeof(int));
  int *cid = (int *)malloc(n * sizeof(int));
  for (size_t i = 0; i < n; i++) {
    const float *p = P(pts, stride, i);
    int c[3];
    for (int a = 0; a < 3; a++) c[a] = (int)floorf(p[a] / cell) - g->lo[a];
    cid[i] = c[0] + c[1] * g->dim[0] + c[2] * g->dim[0] * g->dim[1];
    g->start[cid[i] + 1]++;
  }
  for (size_t c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
  int *fill = (int *)calloc(nc, sizeof(int));
  for (size_t i = 0; i < n; i++) g->idx[g->start[cid[i]] + fill[cid[i]]++] = (int)i;
  free(fill); free(cid);
  return g;
}

static void nn_free(nn_grid *g) { if (g) { free(g->start); free(g->idx); free(g); } }

/* exact 1-NN restricted to squared distance <= gate (cell >= sqrt(gate)); returns -1 if none */
static int nn_query(const nn_grid *g, const float q[3], float gate, float *sqd_out) {
  int c[3];
  for (int a = 0; a < 3; a++) c[a] = (int)floorf(q[a] / g->cell) - g->lo[a];
  int best = -1; float bd = INFINITY;
  for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
    int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
    if (x < 0 || y < 0 || z < 0 || x >= g->dim[0] || y >= g->dim[1] || z >= g->dim[2]) continue;
    int ci = x + y * g->dim[0] + z * g->dim[0] * g->dim[1];
    for (int k = g->start[ci]; k < g->start[ci + 1]; k++) {
      int id = g->idx[k];
      const float *p = P(g->pts, g->stride, (size_t)id);
      float ex = q[0] - p[0], ey = q[1] - p[1], ez = q[2] - p[2];
      float d = ex * ex + ey * ey + ez * ez;
      if (d < bd || (d == bd && id < best)) { bd = d; best = id; }
    }
  }
  if (best < 0 || bd > gate) return -1;
  *sqd_out = bd;
  return best;
}

/* Eigen's Matrix<float,6,6>::determinant() == 0 test, inverse() and inverse() * B (:165-169): eigen_f32_oracle.c, pinned.
 * Returns the determinant; d = H^-1 B when it is not zero. */
static float solve6f(const float H[36], const float B[6], float d[6]) {
  float det = orc_det6f(H);
  if (det == 0.0f) return det;
  float Hi[36];
  orc_inv6f(H, Hi);
  orc_mul66f_vec(Hi, B, d);
  return det;
}

static void so3_exp_f32(const float w[3], float R[9]) {
  /* Sophus SO3::exp via unit quaternion, then Eigen Quaternion::toRotationMatrix */
  float theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  float imag, real;
  if (theta_sq < 1e-10f) {
    float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_po4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_po4;
  } else {
    float theta = sqrtf(theta_sq), half = 0.5f * theta;
    imag = sinf(half) / theta; real = cosf(half);
  }
  float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
}

int orc_icp_manual(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                   float max_correspond_dis, int max_iter, const float guess[16], float out_pose[16]) {
  float T[16];
  memcpy(T, guess, sizeof(T));
  float R[9], t[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r]; t[r] = T[12 + r]; }
  float cell = sqrtf(max_correspond_dis);
  if (!(cell > 0.05f)) cell = 0.05f;
  nn_grid *g = nt ? nn_build(tgt, nt, tstride, cell) : NULL;
  for (int it = 0; it < max_iter; it++) {
    float H[36], B[6];
    memset(H, 0, sizeof(H)); memset(B, 0, sizeof(B));
    for (size_t i = 0; i < ns; i++) {
      const float *p = P(src, sstride, i);
      if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
      float q[3];
      orc_transform_point(T, p, q); /* pcl::transformPointCloud, :132 */
      float sqd;
      int id = g ? nn_query(g, q, max_correspond_dis, &sqd) : -1;
      if (id < 0) continue; /* distances[0] > max_correspond_distance_ (:147, squared vs un-squared, sic) */
      const float *c = P(tgt, tstride, (size_t)id);
      float e[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
      orc_icp_acc(R, p, e, H, B); /* :156-162, Eigen's product association (pinned) */
    }
    float d[6];
    if (solve6f(H, B, d) == 0.0f) continue; /* :165 */
    t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
    float dR[9];
    so3_exp_f32(d + 3, dR);
    orc_mul3f(R, dR, R); /* rotation_matrix_ *= delta_rotation.matrix() (:173) */
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c]; T[12 + r] = t[r]; }
  }
  nn_free(g);
  memcpy(out_pose, T, sizeof(T));
  return 0;
}

/* One ICPRegistrationManual iteration's sums over `src` at pose T (column-major): out32[1..6] = -J^T e, [7..27] the upper triangle of
 * J^T J row by row, [28] correspondences, [29] finite source points — the layout of csrc/icp.hip's workgroup rows, accumulated in f64 as
 * the kernels do.  What a rank of a sharded ScanMatch contributes to the exchange step (tests/test_dist_cpu.py evaluates each rank's
 * shard with this and sums over gloo).  J and e are formed in f32 exactly as orc_icp_manual forms them (:132-159). */
int orc_icp_sums(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride, float max_correspond_dis,
                 const float T[16], double out32[32]) {
  float R[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r];
  for (int k = 0; k < 32; k++) out32[k] = 0.0;
  float cell = sqrtf(max_correspond_dis);
  if (!(cell > 0.05f)) cell = 0.05f;
  nn_grid *g = nt ? nn_build(tgt, nt, tstride, cell) : NULL;
  for (size_t i = 0; i < ns; i++) {
    const float *p = P(src, sstride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    out32[29] += 1.0;
    float q[3];
    orc_transform_point(T, p, q);
    float sqd;
    int id = g ? nn_query(g, q, max_correspond_dis, &sqd) : -1;
    if (id < 0) continue;
    const float *c = P(tgt, tstride, (size_t)id);
    float e[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
    float H1[36], B1[6];
    memset(H1, 0, sizeof(H1)); memset(B1, 0, sizeof(B1));
    orc_icp_acc(R, p, e, H1, B1);   /* this point's J^T J and -J^T e (f32 J; every product of two f32 is exact in f64) */
    float J[3][6];
    memset(J, 0, sizeof(J));
    J[0][0] = J[1][1] = J[2][2] = 1.0f;
    const float hat[9] = {0.0f, -p[2], p[1], p[2], 0.0f, -p[0], -p[1], p[0], 0.0f};
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) J[r][3 + cc] = (-R[r * 3]) * hat[cc] + ((-R[r * 3 + 1]) * hat[3 + cc] + (-R[r * 3 + 2]) * hat[6 + cc]);
    int k = 7;
    for (int a = 0; a < 6; a++) {
      out32[1 + a] += -((double)J[0][a] * (double)e[0] + (double)J[1][a] * (double)e[1] + (double)J[2][a] * (double)e[2]);
      for (int b = a; b < 6; b++) { out32[k] += (double)J[0][a] * J[0][b] + (double)J[1][a] * J[1][b] + (double)J[2][a] * J[2][b]; k++; }
    }
    out32[28] += 1.0;
  }
  nn_free(g);
  return 0;
}

/* ======================================================================================================================
 * Point-to-plane ICP — a BUILD EXTENSION, not reference code: BASELINE.json configs[2] asks for "point-to-plane ICP" but
 * both ICP plugins of the reference are point-to-point (SURVEY.md 8a note).  It reuses the ICPRegistrationManual skeleton
 * above (fixed max_iter Gauss-Newton iterations, f32 pose, t += d[0:3], R *= exp(d[3:6]), skip when det H == 0) with the
 * scalar residual r = n . (p' - q) and the row J = n^T [I | -R hat(p)].  Defined here and in csrc/icp.hip identically:
 *   - correspondence: exact 1-NN in the target, kept when the squared distance <= max_correspond_dis^2;
 *   - normal of a target point = eigenvector of the smallest eigenvalue of the population covariance of the target
 *     points in its voxel (cells floor(x / normal_leaf)), voxels with fewer than 6 points give no normal (pair dropped);
 *   - J and r are formed in f32, the normal equations are accumulated in f64, the 6x6 solve is the manual plugin's f32 LU.
 * Parity for this mode is only definable between these two restatements.
 * ==================================================================================================================== */
void orc_eig3_jacobi(const double a[9], double evals[3], double evecs[9]);

typedef struct { int lo[3], dim[3]; float leaf; float *normal; /* 4 per cell: nx, ny, nz, valid */ } normal_grid;

static int ng_cell(const normal_grid *g, const float *p) {
  int c[3];
  for (int a = 0; a < 3; a++) {
    c[a] = (int)floorf(p[a] / g->leaf) - g->lo[a];
    if (c[a] < 0 || c[a] >= g->dim[a]) return -1;
  }
  return c[0] + c[1] * g->dim[0] + c[2] * g->dim[0] * g->dim[1];
}

static normal_grid *ng_build(const void *pts, size_t n, size_t stride, float leaf) {
  normal_grid *g = (normal_grid *)calloc(1, sizeof(normal_grid));
  g->leaf = leaf;
  int mn[3] = {2147483647, 2147483647, 2147483647}, mx[3] = {-2147483647, -2147483647, -2147483647};
  size_t nf = 0;
  for (size_t i = 0; i < n; i++) {
    const float *p = P(pts, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    nf++;
    for (int a = 0; a < 3; a++) { int c = (int)floorf(p[a] / leaf); if (c < mn[a]) mn[a] = c; if (c > mx[a]) mx[a] = c; }
  }
  if (!nf) { g->dim[0] = g->dim[1] = g->dim[2] = 0; return g; }
  for (int a = 0; a < 3; a++) { g->lo[a] = mn[a]; g->dim[a] = mx[a] - mn[a] + 1; }
  size_t nc = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
  double *sum = (double *)calloc(nc * 9, sizeof(double)); /* sx sy sz sxx sxy sxz syy syz szz, about the cell corner */
  int *cnt = (int *)calloc(nc, sizeof(int));
  g->normal = (float *)calloc(nc * 4, sizeof(float));
  for (size_t i = 0; i < n; i++) {
    const float *p = P(pts, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int c = ng_cell(g, p);
    int ci[3] = {c % g->dim[0], (c / g->dim[0]) % g->dim[1], c / (g->dim[0] * g->dim[1])};
    double d[3];
    for (int a = 0; a < 3; a++) d[a] = (double)p[a] - (double)(ci[a] + g->lo[a]) * (double)leaf;
    double *s = sum + (size_t)c * 9;
    s[0] += d[0]; s[1] += d[1]; s[2] += d[2];
    s[3] += d[0] * d[0]; s[4] += d[0] * d[1]; s[5] += d[0] * d[2]; s[6] += d[1] * d[1]; s[7] += d[1] * d[2]; s[8] += d[2] * d[2];
    cnt[c]++;
  }
  for (size_t c = 0; c < nc; c++) {
    if (cnt[c] < 6) continue;
    double m = (double)cnt[c];
    const double *s = sum + c * 9;
    double mu[3] = {s[0] / m, s[1] / m, s[2] / m};
    double C[9] = {s[3] / m - mu[0] * mu[0], s[4] / m - mu[0] * mu[1], s[5] / m - mu[0] * mu[2],
                   0, s[6] / m - mu[1] * mu[1], s[7] / m - mu[1] * mu[2], 0, 0, s[8] / m - mu[2] * mu[2]};
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    double ev[3], V[9];
    orc_eig3_jacobi(C, ev, V);
    g->normal[c * 4] = (float)V[0]; g->normal[c * 4 + 1] = (float)V[3]; g->normal[c * 4 + 2] = (float)V[6]; /* column 0 */
    g->normal[c * 4 + 3] = 1.0f;
  }
  free(sum); free(cnt);
  return g;
}

static void ng_free(normal_grid *g) { if (g) { free(g->normal); free(g); } }

int orc_icp_plane(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                  float max_correspond_dis, int max_iter, float normal_leaf, const float guess[16], float out_pose[16], double stats[2]) {
  float T[16];
  memcpy(T, guess, sizeof(T));
  float R[9], t[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = T[c * 4 + r]; t[r] = T[12 + r]; }
  float cell = max_correspond_dis > 0.05f ? max_correspond_dis : 0.05f;
  nn_grid *g = nt ? nn_build(tgt, nt, tstride, cell) : NULL;
  normal_grid *ng = ng_build(tgt, nt, tstride, normal_leaf);
  float gate = max_correspond_dis * max_correspond_dis;
  double ncorr = 0;
  for (int it = 0; it < max_iter; it++) {
    double Hd[36], Bd[6];
    memset(Hd, 0, sizeof(Hd)); memset(Bd, 0, sizeof(Bd));
    ncorr = 0;
    for (size_t i = 0; i < ns; i++) {
      const float *p = P(src, sstride, i);
      if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
      float q[3];
      orc_transform_point(T, p, q);
      if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
      float sqd;
      int id = g ? nn_query(g, q, gate, &sqd) : -1;
      if (id < 0) continue;
      const float *c = P(tgt, tstride, (size_t)id);
      int nc = ng_cell(ng, c);
      if (nc < 0 || ng->normal[(size_t)nc * 4 + 3] == 0.0f) continue;
      const float *nv = ng->normal + (size_t)nc * 4;
      float e[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
      float r = (nv[0] * e[0] + nv[1] * e[1]) + nv[2] * e[2];
      float hat[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
      float A[3][3];
      for (int rr = 0; rr < 3; rr++)
        for (int cc = 0; cc < 3; cc++)
          A[rr][cc] = ((-R[rr * 3 + 0]) * hat[0 * 3 + cc] + (-R[rr * 3 + 1]) * hat[1 * 3 + cc]) + (-R[rr * 3 + 2]) * hat[2 * 3 + cc];
      float J[6] = {nv[0], nv[1], nv[2], 0, 0, 0};
      for (int cc = 0; cc < 3; cc++) J[3 + cc] = (nv[0] * A[0][cc] + nv[1] * A[1][cc]) + nv[2] * A[2][cc];
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) Hd[a * 6 + b] += (double)J[a] * (double)J[b];
        Bd[a] -= (double)J[a] * (double)r;
      }
      ncorr += 1;
    }
    float H[36], B[6];
    for (int k = 0; k < 36; k++) H[k] = (float)Hd[k];
    for (int k = 0; k < 6; k++) B[k] = (float)Bd[k];
    float d[6];
    if (solve6f(H, B, d) == 0.0f) continue;
    t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
    float dR[9];
    so3_exp_f32(d + 3, dR);
    orc_mul3f(R, dR, R);
    for (int r2 = 0; r2 < 3; r2++) { for (int c2 = 0; c2 < 3; c2++) T[c2 * 4 + r2] = R[r2 * 3 + c2]; T[12 + r2] = t[r2]; }
  }
  nn_free(g); ng_free(ng);
  memcpy(out_pose, T, sizeof(T));
  if (stats) { stats[0] = max_iter; stats[1] = ncorr; }
  return 0;
}
