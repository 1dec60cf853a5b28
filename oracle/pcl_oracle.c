/*
 * pcl_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY): the pieces of the hot path whose arithmetic lives in
 * third-party PCL ("find_package(PCL 1.7 REQUIRED)", lidar_localization/cmake/PCL.cmake:1; no lockfile, not vendored)
 * rather than in the reference tree:
 *   - the "NDT" plugin's voxel statistics (S/models/registration/ndt_registration.cpp:46-66 -> pcl::NormalDistributions
 *     Transform -> pcl::VoxelGridCovariance): only the eigen-solver lives here, the mode switch is in ndt_oracle.c;
 *   - pcl::Registration::getFitnessScore, the quantity LoopClosing gates on (S/mapping/loop_closing/loop_closing.cpp:253);
 *   - the "ICP" plugin (S/models/registration/icp_registration.cpp:20-64 -> pcl::IterativeClosestPoint).
 * PARITY STATUS: UNPINNED. PCL is absent from this image and the reference holds no tests or golden vectors at these
 * call sites, so this file restates PCL 1.7's published algorithm and is anchored on the reference's call
 * sites and parameters only.
 */
#include "lsa_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline const float *pxyz(const void *pts, size_t stride, size_t i) {
  return (const float *)((const char *)pts + i * stride);
}

/* ---- Eigen::SelfAdjointEigenSolver<Matrix3d> stand-in: cyclic Jacobi, ascending eigenvalues ---------------- */
void orc_eig3_jacobi(const double a[9], double evals[3], double evecs[9]) {
  double A[3][3] = {{a[0], a[1], a[2]}, {a[1], a[4], a[5]}, {a[2], a[5], a[8]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p][q];
        if (apq == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { /* A <- A J */
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) { /* A <- J^T A */
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int ord[3] = {0, 1, 2};
  double d[3] = {A[0][0], A[1][1], A[2][2]};
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++)
      if (d[ord[j]] < d[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int c = 0; c < 3; c++) {
    evals[c] = d[ord[c]];
    for (int r = 0; r < 3; r++) evecs[r * 3 + c] = V[r][ord[c]];
  }
}

/* ---- exact 1-NN over the target points (stands in for pcl::search::KdTree / FLANN; f32 L2_Simple distances) ---- */
typedef struct {
  float cell, mn[3];
  int dim[3];
  int *start;     /* dim0*dim1*dim2 + 1 */
  float *pts;     /* sorted xyz + original index */
  int *idx;
  size_t n;
} nn_grid;

static void nn_free(nn_grid *g) {
  if (!g) return;
  free(g->start); free(g->pts); free(g->idx); free(g);
}

static inline int nn_cell(const nn_grid *g, float v, int a) { return (int)floorf((v - g->mn[a]) / g->cell); }

static nn_grid *nn_build(const void *pts, size_t n, size_t stride, float cell) {
  nn_grid *g = (nn_grid *)calloc(1, sizeof(nn_grid));
  float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  g->cell = cell;
  g->mn[0] = g->mn[1] = g->mn[2] = FLT_MAX;
  size_t nf = 0;
  for (size_t i = 0; i < n; i++) {
    const float *p = pxyz(pts, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    nf++;
    for (int a = 0; a < 3; a++) { if (p[a] < g->mn[a]) g->mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
  }
  if (!nf) { free(g); return NULL; }
  for (int a = 0; a < 3; a++) g->dim[a] = nn_cell(g, mx[a], a) + 1;
  size_t nc = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
  g->start = (int *)calloc(nc + 1, sizeof(int));
  g->pts = (float *)malloc(nf * 3 * sizeof(float));
  g->idx = (int *)malloc(nf * sizeof(int));
  g->n = nf;
  for (size_t i = 0; i < n; i++) {
    const float *p = pxyz(pts, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    size_t c = (size_t)nn_cell(g, p[0], 0) + (size_t)g->dim[0] * ((size_t)nn_cell(g, p[1], 1) + (size_t)g->dim[1] * nn_cell(g, p[2], 2));
    g->start[c + 1]++;
  }
  for (size_t c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
  int *fill = (int *)calloc(nc, sizeof(int));
  for (size_t i = 0; i < n; i++) { /* ascending original index inside a cell */
    const float *p = pxyz(pts, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    size_t c = (size_t)nn_cell(g, p[0], 0) + (size_t)g->dim[0] * ((size_t)nn_cell(g, p[1], 1) + (size_t)g->dim[1] * nn_cell(g, p[2], 2));
    int s = g->start[c] + fill[c]++;
    memcpy(g->pts + (size_t)s * 3, p, 3 * sizeof(float));
    g->idx[s] = (int)i;
  }
  free(fill);
  return g;
}

static inline void nn_visit(const nn_grid *g, int ix, int iy, int iz, const float q[3], float *best, int *best_id, float nn[3]) {
  size_t cc = (size_t)ix + (size_t)g->dim[0] * ((size_t)iy + (size_t)g->dim[1] * iz);
  for (int k = g->start[cc]; k < g->start[cc + 1]; k++) {
    const float *t = g->pts + (size_t)k * 3;
    float ex = q[0] - t[0], ey = q[1] - t[1], ez = q[2] - t[2];
    float d = ex * ex + ey * ey + ez * ez;
    if (d < *best || (d == *best && g->idx[k] < *best_id)) { *best = d; *best_id = g->idx[k]; nn[0] = t[0]; nn[1] = t[1]; nn[2] = t[2]; }
  }
}

/* returns the original index of the nearest point (ties: lowest index), *d2 = f32 squared distance.
 * Chebyshev rings of cells around the query's cell, clipped to the grid, until the best distance is provably minimal. */
static int nn_query(const nn_grid *g, const float q[3], float *d2, float nn[3]) {
  int c[3];
  for (int a = 0; a < 3; a++) {
    double v = floor(((double)q[a] - g->mn[a]) / g->cell);
    c[a] = (v < -1e9) ? -1000000000 : (v > 1e9) ? 1000000000 : (int)v;
  }
  int rho0 = 0, rho_end = 0;
  for (int a = 0; a < 3; a++) {
    int out = (c[a] < 0) ? -c[a] : (c[a] > g->dim[a] - 1 ? c[a] - (g->dim[a] - 1) : 0);
    if (out > rho0) rho0 = out;
    int far = (abs(c[a]) > abs(g->dim[a] - 1 - c[a])) ? abs(c[a]) : abs(g->dim[a] - 1 - c[a]);
    if (far > rho_end) rho_end = far;
  }
  float best = INFINITY;
  int best_id = -1;
  for (int rho = rho0; rho <= rho_end; rho++) {
    int z0 = c[2] - rho, z1 = c[2] + rho, y0 = c[1] - rho, y1 = c[1] + rho, x0 = c[0] - rho, x1 = c[0] + rho;
    int zc0 = z0 < 0 ? 0 : z0, zc1 = z1 > g->dim[2] - 1 ? g->dim[2] - 1 : z1;
    int yc0 = y0 < 0 ? 0 : y0, yc1 = y1 > g->dim[1] - 1 ? g->dim[1] - 1 : y1;
    int xc0 = x0 < 0 ? 0 : x0, xc1 = x1 > g->dim[0] - 1 ? g->dim[0] - 1 : x1;
    for (int iz = zc0; iz <= zc1; iz++)
      for (int iy = yc0; iy <= yc1; iy++) {
        if (iz == z0 || iz == z1 || iy == y0 || iy == y1) {
          for (int ix = xc0; ix <= xc1; ix++) nn_visit(g, ix, iy, iz, q, &best, &best_id, nn);
        } else {
          if (x0 >= 0 && x0 < g->dim[0]) nn_visit(g, x0, iy, iz, q, &best, &best_id, nn);
          if (x1 != x0 && x1 >= 0 && x1 < g->dim[0]) nn_visit(g, x1, iy, iz, q, &best, &best_id, nn);
        }
      }
    /* every cell not yet visited is at least rho * cell away (minus rounding slack) */
    float lb = (float)rho * g->cell - 1e-3f;
    if (best_id >= 0 && lb > 0 && best <= lb * lb) break;
  }
  *d2 = best;
  return best_id;
}

double orc_fitness_pcl(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                       const float T[16], double max_range) {
  /* PCL 1.7 registration/impl/registration.hpp getFitnessScore (double max_range) */
  nn_grid *g = nn_build(tgt, nt, tstride, 1.0f);
  if (!g) return DBL_MAX;
  double fitness = 0.0;
  long nr = 0;
  for (size_t i = 0; i < ns; i++) {
    float q[3], d2, nn[3];
    orc_transform_point(T, pxyz(src, sstride, i), q);
    if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
    if (nn_query(g, q, &d2, nn) < 0) continue;
    if ((double)d2 <= max_range) { fitness += (double)d2; nr++; }
  }
  nn_free(g);
  return nr > 0 ? fitness / (double)nr : DBL_MAX;
}

/* ---- 3x3 SVD (one-sided Jacobi, f64) for Umeyama ------------------------------------------------------------- */
static void svd3(const double M[9], double U[9], double S[3], double V[9]) {
  double A[3][3], W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = M[i * 3 + j];
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; k++) { al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q]; }
        double lim = sqrt(al * be);
        if (ga == 0.0 || fabs(ga) <= 1e-18 * lim) continue;
        off = fmax(off, fabs(ga) / (lim > 0 ? lim : 1));
        double zeta = (be - al) / (2.0 * ga);
        double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
          double wp = W[k][p], wq = W[k][q];
          W[k][p] = c * wp - s * wq; W[k][q] = s * wp + c * wq;
        }
      }
    if (off < 1e-16) break;
  }
  int ord[3] = {0, 1, 2};
  double sv[3];
  for (int j = 0; j < 3; j++) sv[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++)
      if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int c = 0; c < 3; c++) {
    int j = ord[c];
    S[c] = sv[j];
    for (int r = 0; r < 3; r++) { V[r * 3 + c] = W[r][j]; U[r * 3 + c] = (sv[j] > 0) ? A[r][j] / sv[j] : 0.0; }
  }
  /* complete U to an orthonormal basis when the matrix is rank deficient */
  if (S[2] <= 1e-300 * (S[0] > 0 ? S[0] : 1) || S[2] == 0) {
    if (S[1] == 0) { /* rank <= 1: any orthonormal completion */
      double u0[3] = {U[0], U[3], U[6]};
      if (S[0] == 0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
      double e[3] = {0, 0, 0};
      e[(fabs(u0[0]) < 0.9) ? 0 : 1] = 1;
      double u1[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
      double nn = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      for (int r = 0; r < 3; r++) U[r * 3 + 1] = u1[r] / nn;
    }
    double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

static double det3(const double m[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* Eigen::umeyama(src, dst, with_scaling = false) for Scalar = float (Eigen 3.2 Geometry/Umeyama.h:95-170), which
 * pcl::registration::TransformationEstimationSVD<PointXYZI, PointXYZI, float> calls. Means / demeaned products are
 * formed in f32 left to right as the scalar code path would; the 3x3 SVD runs in f64 and is rounded. */
static void umeyama_f32(const float *src, const float *dst, int n, float T[16]) {
  float inv_n = 1.0f / (float)n;
  float sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) { sm[a] += src[i * 3 + a]; dm[a] += dst[i * 3 + a]; }
  for (int a = 0; a < 3; a++) { sm[a] *= inv_n; dm[a] *= inv_n; }
  float sig[9] = {0};
  for (int i = 0; i < n; i++)
    for (int r = 0; r < 3; r++) {
      float dr = dst[i * 3 + r] - dm[r];
      for (int c = 0; c < 3; c++) sig[r * 3 + c] += dr * (src[i * 3 + c] - sm[c]);
    }
  double M[9], U[9], S[3], V[9];
  for (int k = 0; k < 9; k++) { sig[k] *= inv_n; M[k] = sig[k]; }
  svd3(M, U, S, V);
  double Sd[3] = {1, 1, 1};
  if (det3(M) < 0) Sd[2] = -1;
  int rank = 0;
  for (int i = 0; i < 3; i++)
    if (!(fabs(S[i]) <= fabs(S[0]) * (double)FLT_EPSILON)) rank++; /* !isMuchSmallerThan(d(i), d(0)) in f32 precision */
  if (rank == 2) {
    if (det3(U) * det3(V) > 0) { Sd[2] = 1; }
    else { Sd[2] = -1; }
  }
  double R[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r * 3 + c] = U[r * 3 + 0] * Sd[0] * V[c * 3 + 0] + U[r * 3 + 1] * Sd[1] * V[c * 3 + 1] + U[r * 3 + 2] * Sd[2] * V[c * 3 + 2];
  for (int k = 0; k < 16; k++) T[k] = 0.0f;
  float Rf[9];
  for (int k = 0; k < 9; k++) Rf[k] = (float)R[k];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[c * 4 + r] = Rf[r * 3 + c];
    T[12 + r] = dm[r] - ((Rf[r * 3 + 0] * sm[0] + Rf[r * 3 + 1] * sm[1]) + Rf[r * 3 + 2] * sm[2]);
  }
  T[15] = 1.0f;
}

static void mat4_mul_f32(const float A[16], const float B[16], float C[16]) { /* column-major, C = A * B */
  float o[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) o[c * 4 + r] = ((A[0 * 4 + r] * B[c * 4 + 0] + A[1 * 4 + r] * B[c * 4 + 1]) + A[2 * 4 + r] * B[c * 4 + 2]) + A[3 * 4 + r] * B[c * 4 + 3];
  memcpy(C, o, sizeof(o));
}

int orc_icp_pcl(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                float euclid_eps, float max_correspond_dis, float trans_eps, int max_iter, const float guess[16],
                float out_pose[16], double stats[4]) {
  /* PCL 1.7 registration/impl/icp.hpp computeTransformation + correspondence_estimation.hpp determineCorrespondences
   * + default_convergence_criteria.hpp hasConverged */
  nn_grid *g = nn_build(tgt, nt, tstride, max_correspond_dis > 0.05f ? max_correspond_dis : 0.05f);
  if (!g) return -1;
  float *work = (float *)malloc((ns ? ns : 1) * 3 * sizeof(float));
  float *cs = (float *)malloc((ns ? ns : 1) * 3 * sizeof(float)), *ct = (float *)malloc((ns ? ns : 1) * 3 * sizeof(float));
  float final_T[16], incr[16];
  memcpy(final_T, guess, sizeof(final_T));
  for (size_t i = 0; i < ns; i++) orc_transform_point(guess, pxyz(src, sstride, i), work + i * 3); /* identity: exact copy */
  int nr_iterations = 0, converged = 0;
  double max_dist_sqr = (double)max_correspond_dis * (double)max_correspond_dis;
  double prev_mse = DBL_MAX, cur_mse = 0;
  double rot_thr = 1.0 - (double)trans_eps, tr_thr = (double)trans_eps, mse_rel = (double)euclid_eps, mse_abs = 1e-12;
  size_t cnt = 0;
  do {
    cnt = 0;
    double mse = 0;
    for (size_t i = 0; i < ns; i++) {
      const float *q = work + i * 3;
      if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
      float d2, nn[3];
      if (nn_query(g, q, &d2, nn) < 0) continue;
      if ((double)d2 > max_dist_sqr) continue;
      memcpy(cs + cnt * 3, q, 3 * sizeof(float));
      memcpy(ct + cnt * 3, nn, 3 * sizeof(float));
      mse += (double)d2;
      cnt++;
    }
    if (cnt < 3) { converged = 0; break; } /* min_number_correspondences_ */
    umeyama_f32(cs, ct, (int)cnt, incr);
    for (size_t i = 0; i < ns; i++) {
      float *q = work + i * 3, o[3];
      if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
      orc_transform_point(incr, q, o);
      memcpy(q, o, sizeof(o));
    }
    mat4_mul_f32(incr, final_T, final_T);
    ++nr_iterations;
    /* DefaultConvergenceCriteria::hasConverged */
    if (nr_iterations >= max_iter) { converged = 1; break; }
    double cos_angle = 0.5 * ((double)incr[0] + (double)incr[5] + (double)incr[10] - 1.0);
    double tsq = (double)incr[12] * incr[12] + (double)incr[13] * incr[13] + (double)incr[14] * incr[14];
    if (cos_angle >= rot_thr && tsq <= tr_thr) { converged = 1; break; }
    cur_mse = mse / (double)cnt;
    if (fabs(cur_mse - prev_mse) < mse_abs) { converged = 1; break; }
    if (fabs(cur_mse - prev_mse) / prev_mse < mse_rel) { converged = 1; break; }
    prev_mse = cur_mse;
  } while (!converged);
  memcpy(out_pose, final_T, sizeof(final_T));
  if (stats) { stats[0] = nr_iterations; stats[1] = converged; stats[2] = (double)cnt; stats[3] = cur_mse; }
  free(work); free(cs); free(ct);
  nn_free(g);
  return 0;
}
