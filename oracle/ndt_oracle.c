/*
 * ndt_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lsa_oracle.h header).
 *
 * Restates, in plain C with the reference's evaluation order and precisions
 * (f32 points / f32 rigid transform, f64 everything else):
 *   NS/VoxelGrid.cpp, NI/SymmetricEigenSolver.h, NS/Registration.cpp,
 *   NS/NormalDistributionsTransform.cpp
 * plus the vendored-Eigen calls that path makes.  PARITY STATUS: the loops in this
 * file are UNPINNED (no reference fixtures exist, reference unbuildable here); the
 * numeric building blocks orc_eig3 / orc_inv3 / orc_svd6_solve / orc_pose_to_matrix /
 * orc_matrix_to_pose (incl. the f32 polar step of Transform::rotation()) are pinned against real reference code by oracle/_ref.
 *
 * Build with -ffp-contract=off: the reference is built for baseline x86-64 (no FMA).
 */
#include "lsa_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* point access: PCL PointXYZI is 32 B, xyz at 0, intensity at 16            */
/* ------------------------------------------------------------------------ */
static inline const float *pt_xyz(const void *pts, size_t stride, size_t i) {
  return (const float *)((const char *)pts + i * stride);
}
static inline float pt_w(const void *pts, size_t stride, size_t w_off, size_t i) {
  return *(const float *)((const char *)pts + i * stride + w_off);
}

/* ======================================================================== */
/* building blocks                                                           */
/* ======================================================================== */

static void cross3(const double u[3], const double v[3], double out[3]) {
  /* NI/SymmetricEigenSolver.h:242-251 */
  out[0] = u[1] * v[2] - u[2] * v[1];
  out[1] = u[2] * v[0] - u[0] * v[2];
  out[2] = u[0] * v[1] - u[1] * v[0];
}

void orc_eig3(const double a[9], double evals[3], double evecs[9]) {
  /* NI/SymmetricEigenSolver.h:55-136 (compute) */
  double a00 = a[0], a01 = a[1], a02 = a[2], a11 = a[4], a12 = a[5], a22 = a[8];
  memset(evecs, 0, 9 * sizeof(double));
  evals[0] = evals[1] = evals[2] = 0.0;

  double max0 = (fabs(a00) > fabs(a01)) ? fabs(a00) : fabs(a01);
  double max1 = (fabs(a02) > fabs(a11)) ? fabs(a02) : fabs(a11);
  double max2 = (fabs(a12) > fabs(a22)) ? fabs(a12) : fabs(a22);
  double maxAbs = (max0 > max1) ? max0 : max1;
  maxAbs = (maxAbs > max2) ? maxAbs : max2;
  if (maxAbs == 0.0) {
    evecs[0] = evecs[4] = evecs[8] = 1.0;
    return;
  }
  double inv = 1.0 / maxAbs;
  a00 *= inv; a01 *= inv; a02 *= inv; a11 *= inv; a12 *= inv; a22 *= inv;

  double norm = a01 * a01 + a02 * a02 + a12 * a12;
  if (norm > 0.0) {
    double traceDiv3 = (a00 + a11 + a22) / 3.0;
    double b00 = a00 - traceDiv3, b11 = a11 - traceDiv3, b22 = a22 - traceDiv3;
    double denom = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
    double c00 = b11 * b22 - a12 * a12;
    double c01 = a01 * b22 - a12 * a02;
    double c02 = a01 * a12 - b11 * a02;
    double det = (b00 * c00 - a01 * c01 + a02 * c02) / (denom * denom * denom);
    double halfDet = det * 0.5;
    halfDet = (halfDet > -1.0) ? halfDet : -1.0;
    halfDet = (halfDet < 1.0) ? halfDet : 1.0;

    double angle = acos(halfDet) / 3.0;
    double beta2 = cos(angle) * 2.0;
    double beta0 = cos(angle + M_PI * 2.0 / 3.0) * 2.0;
    double beta1 = -(beta0 + beta2);
    evals[0] = traceDiv3 + denom * beta0;
    evals[1] = traceDiv3 + denom * beta1;
    evals[2] = traceDiv3 + denom * beta2;

    int i0, i2, i1 = 1;
    if (halfDet >= 0.0) { i0 = 2; i2 = 0; } else { i0 = 0; i2 = 2; }

    /* computeEigenvector0, :149-183 */
    {
      double e0 = evals[i0];
      double r0[3] = {a00 - e0, a01, a02};
      double r1[3] = {a01, a11 - e0, a12};
      double r2[3] = {a02, a12, a22 - e0};
      double rxr[3][3];
      cross3(r0, r1, rxr[0]);
      cross3(r0, r2, rxr[1]);
      cross3(r1, r2, rxr[2]);
      /* quirk (:172-174): x^2 + y^2 * z^2, not x^2 + y^2 + z^2 */
      double d0 = rxr[0][0] * rxr[0][0] + rxr[0][1] * rxr[0][1] * rxr[0][2] * rxr[0][2];
      double d1 = rxr[1][0] * rxr[1][0] + rxr[1][1] * rxr[1][1] * rxr[1][2] * rxr[1][2];
      double d2 = rxr[2][0] * rxr[2][0] + rxr[2][1] * rxr[2][1] * rxr[2][2] * rxr[2][2];
      double dmax = (d0 > d1) ? d0 : d1;
      int imax = (d0 > d1) ? 0 : 1;
      /* quirk (:179-180): dmax is overwritten first, so imax can never become 2 */
      dmax = (d2 > dmax) ? d2 : dmax;
      imax = (d2 > dmax) ? 2 : imax;
      double s = sqrt(dmax);
      for (int r = 0; r < 3; r++) evecs[r * 3 + i0] = rxr[imax][r] / s;
    }
    /* computeEigenvector1, :185-239 */
    {
      double w[3] = {evecs[0 * 3 + i0], evecs[1 * 3 + i0], evecs[2 * 3 + i0]};
      double u[3], v[3];
      /* computeOrthogonalComplement, :253-264 */
      int c = (fabs(w[0]) > fabs(w[1]));
      double inv_length = c ? (1.0 / sqrt(w[0] * w[0] + w[2] * w[2])) : (1.0 / sqrt(w[1] * w[1] + w[2] * w[2]));
      u[0] = c ? -w[2] * inv_length : 0.0;
      u[1] = c ? 0.0 : w[2] * inv_length;
      u[2] = c ? w[0] * inv_length : -w[1] * inv_length;
      cross3(w, u, v);

      double e1 = evals[i1];
      double au[3], av[3], t0, t1, t2;
      t0 = u[0]; t1 = u[1]; t2 = u[2];
      au[0] = (a00 - e1) * t0 + a01 * t1 + a02 * t2;
      au[1] = a01 * t0 + (a11 - e1) * t1 + a12 * t2;
      au[2] = a02 * t0 + a12 * t1 + (a22 - e1) * t2;
      t0 = v[0]; t1 = v[1]; t2 = v[2];
      av[0] = (a00 - e1) * t0 + a01 * t1 + a02 * t2;
      av[1] = a01 * t0 + (a11 - e1) * t1 + a12 * t2;
      av[2] = a02 * t0 + a12 * t1 + (a22 - e1) * t2;

      double m00 = u[0] * au[0] + u[1] * au[1] + u[2] * au[2];
      double m01 = u[0] * av[0] + u[1] * av[1] + u[2] * av[2];
      double m11 = v[0] * av[0] + v[1] * av[1] + v[2] * av[2];
      double am00 = fabs(m00), am01 = fabs(m01), am11 = fabs(m11);
      if (am00 > 0 || am01 > 0 || am11 > 0) {
        double u_mult = (am00 >= am11) ? m01 : m11;
        double v_mult = (am00 >= am11) ? m00 : m01;
        int res = fabs(u_mult) >= fabs(v_mult);
        double *large = res ? &u_mult : &v_mult;
        double *small = res ? &v_mult : &u_mult;
        *small /= (*large);
        *large = 1.0 / sqrt(1.0 + (*small) * (*small));
        *small *= (*large);
        for (int r = 0; r < 3; r++) evecs[r * 3 + i1] = u[r] * u_mult - v[r] * v_mult;
      } else {
        for (int r = 0; r < 3; r++) evecs[r * 3 + i1] = u[r];
      }
    }
    /* computeEigenvector2, :266-273 */
    {
      double e0[3] = {evecs[0 * 3 + i0], evecs[1 * 3 + i0], evecs[2 * 3 + i0]};
      double e1[3] = {evecs[0 * 3 + i1], evecs[1 * 3 + i1], evecs[2 * 3 + i1]};
      double e2[3];
      cross3(e0, e1, e2);
      for (int r = 0; r < 3; r++) evecs[r * 3 + i2] = e2[r];
    }
  } else {
    evals[0] = a00; evals[1] = a11; evals[2] = a22;
    evecs[0] = evecs[4] = evecs[8] = 1.0;
  }
  evals[0] *= maxAbs; evals[1] *= maxAbs; evals[2] *= maxAbs;
}

static inline double cof3(const double *m, int i, int j) {
  /* Eigen LU/InverseImpl.h:124-136 cofactor_3x3<i,j> */
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}

void orc_inv3(const double m[9], double out[9]) {
  /* Eigen LU/InverseImpl.h:140-169 */
  double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  double det = c0 * m[0] + (c1 * m[3] + c2 * m[6]); /* Eigen redux of 3: a0 + (a1 + a2) */
  double invdet = 1.0 / det;
  out[0] = c0 * invdet; out[1] = c1 * invdet; out[2] = c2 * invdet;
  out[3] = cof3(m, 0, 1) * invdet; out[4] = cof3(m, 1, 1) * invdet; out[5] = cof3(m, 2, 1) * invdet;
  out[6] = cof3(m, 0, 2) * invdet; out[7] = cof3(m, 1, 2) * invdet; out[8] = cof3(m, 2, 2) * invdet;
}

int orc_svd6_solve(const double H[36], const double b[6], double x[6]) {
  /* One-sided (Hestenes) Jacobi SVD in f64: A = U S V^T. Eigen's JacobiSVD is two-sided; both
   * are backward stable, the pseudo-inverse solve agrees to rounding (pinned by oracle/_ref). */
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { A[i][j] = H[i * 6 + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 6; k++) { alpha += A[k][p] * A[k][p]; beta += A[k][q] * A[k][q]; gamma += A[k][p] * A[k][q]; }
        if (gamma == 0.0) continue;
        double lim = sqrt(alpha * beta);
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= DBL_EPSILON * 1e-3 * lim) continue;
        off = fmax(off, fabs(gamma) / (lim > 0 ? lim : 1));
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; k++) {
          double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
          double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[6]; int ord[6];
  for (int j = 0; j < 6; j++) {
    double s2 = 0; for (int k = 0; k < 6; k++) s2 += A[k][j] * A[k][j];
    sv[j] = sqrt(s2); ord[j] = j;
  }
  for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  /* SVDBase.h:130-140 rank(), :181-186 threshold() = diagSize * eps */
  double thr = fmax(sv[ord[0]] * 6.0 * DBL_EPSILON, DBL_MIN);
  int nonzero = 0; for (int i = 0; i < 6; i++) if (sv[ord[i]] != 0.0) nonzero++;
  int rank = nonzero; while (rank > 0 && sv[ord[rank - 1]] < thr) rank--;
  for (int i = 0; i < 6; i++) x[i] = 0.0;
  for (int r = 0; r < rank; r++) {
    int j = ord[r];
    double ub = 0; for (int k = 0; k < 6; k++) ub += (A[k][j] / sv[j]) * b[k];
    double coef = ub / sv[j];
    for (int k = 0; k < 6; k++) x[k] += V[k][j] * coef;
  }
  return rank;
}

static void rot_axis_f32(float angle, int axis, float R[9]) {
  /* Eigen Geometry/AngleAxis.h toRotationMatrix() specialised to a unit axis, f32 */
  float s = sinf(angle), c = cosf(angle);
  float d = (1.0f - c) * 1.0f + c; /* (cos1_axis . axis) + c on the rotation axis */
  for (int i = 0; i < 9; i++) R[i] = 0.0f;
  if (axis == 0) { R[0] = d; R[4] = c; R[5] = -s; R[7] = s; R[8] = c; }
  else if (axis == 1) { R[0] = c; R[2] = s; R[4] = d; R[6] = -s; R[8] = c; }
  else { R[0] = c; R[1] = -s; R[3] = s; R[4] = c; R[8] = d; }
}

static void mat3_mul_f32(const float a[9], const float b[9], float out[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      out[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}

void orc_pose_to_matrix(const double p[6], float T[16]) {
  /* NS/NormalDistributionsTransform.cpp:691-694 (and :370-373, :717-720) */
  float Rx[9], Ry[9], Rz[9], Rxy[9], R[9];
  rot_axis_f32((float)p[3], 0, Rx);
  rot_axis_f32((float)p[4], 1, Ry);
  rot_axis_f32((float)p[5], 2, Rz);
  mat3_mul_f32(Rx, Ry, Rxy);
  mat3_mul_f32(Rxy, Rz, R);
  for (int i = 0; i < 16; i++) T[i] = 0.0f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
  T[12] = (float)p[0]; T[13] = (float)p[1]; T[14] = (float)p[2]; T[15] = 1.0f;
}

void orc_matrix_to_pose(const float T[16], double p[6]) {
  /* NS/NormalDistributionsTransform.cpp:331-337.  Transform::rotation() is an f32 two-sided Jacobi SVD polar factor of
   * the linear part (orc_rotation_f32, eigen_f32_oracle.c: pinned bit for bit); the Euler angles are taken from ITS result.
   * Euler extraction: Eigen 3.2.92 Geometry/EulerAngles.h:35-101 with (a0,a1,a2)=(0,1,2):
   * odd=0, i=0, j=1, k=2, first angle folded into [0,pi] before the final negation. */
  float L[9], Rm[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) L[r * 3 + c] = T[c * 4 + r];
  orc_rotation_f32(L, Rm);
#define M(r, c) Rm[(r) * 3 + (c)]
  float res0 = atan2f(M(1, 2), M(2, 2));
  float c2 = sqrtf(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
  float res1;
  if (res0 > 0.0f) {
    res0 = res0 - (float)M_PI;
    res1 = atan2f(-M(0, 2), -c2);
  } else {
    res1 = atan2f(-M(0, 2), c2);
  }
  float s1 = sinf(res0), c1 = cosf(res0);
  float res2 = atan2f(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
#undef M
  p[0] = T[12]; p[1] = T[13]; p[2] = T[14];
  p[3] = -res0; p[4] = -res1; p[5] = -res2;
}

void orc_transform_point(const float T[16], const float in[3], float out[3]) {
  /* PCL 1.7 common/impl/transforms.hpp: x' = m00*x + m01*y + m02*z + m03 in f32, left to right */
  float x = in[0], y = in[1], z = in[2];
  out[0] = T[0] * x + T[4] * y + T[8] * z + T[12];
  out[1] = T[1] * x + T[5] * y + T[9] * z + T[13];
  out[2] = T[2] * x + T[6] * y + T[10] * z + T[14];
}

/* ======================================================================== */
/* voxel grid                                                                */
/* ======================================================================== */

struct orc_grid {
  int mode;             /* 0: in-tree NDTMANUAL VoxelGrid; 1: PCL 1.7 VoxelGridCovariance semantics (unpinned) */
  float leaf;
  float max_x, max_y, max_z, min_x, min_y, min_z;
  int max_b[3], min_b[3], vgrid[3], real_max[3], real_min[3];
  int voxel_num;
  int *npts;            /* points_id_[i].size() */
  int *ppv;             /* points_per_voxel_ (may be -1) */
  double *centroid;     /* 3 */
  double *icov;         /* 9 row-major */
  double *sval;
  double *tmp_centroid; /* 3 */
  double *tmp_cov;      /* 9 */
  double *tmp_sval;
  void *octree;         /* fitness_oracle.c */
};

void *orc_octree_build(const void *pts, size_t n, size_t stride, float leaf);
void orc_octree_free(void *o);
void orc_octree_nearest(void *o, const float q[3], float bounds[6]);

static int round_up(int input, int factor) { /* NS/VoxelGrid.cpp:56-59 */
  return (input < 0) ? -((-input) / factor) * factor : ((input + factor - 1) / factor) * factor;
}
static int round_down(int input, int factor) { /* :62-65 */
  return (input < 0) ? -((-input + factor - 1) / factor) * factor : (input / factor) * factor;
}

static inline int vid3(const orc_grid *g, int ix, int iy, int iz) { /* :237-242 */
  return (ix - g->min_b[0]) + (iy - g->min_b[1]) * g->vgrid[0] + (iz - g->min_b[2]) * g->vgrid[0] * g->vgrid[1];
}

void orc_cov_expr(const double tmp_cov[9], const double pt_sum[3], double pn, double cov[9]) {
  /* NS/VoxelGrid.cpp:286 (centroid), :293-294 */
  double c[3] = {pt_sum[0] / pn, pt_sum[1] / pn, pt_sum[2] / pn};
  double f = (pn - 1.0) / pn;
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) {
      double v = (tmp_cov[r * 3 + k] - 2.0 * (pt_sum[r] * c[k])) / pn + c[r] * c[k];
      cov[r * 3 + k] = v * f;
    }
}

void orc_cov_floor_rebuild(const double evecs[9], const double evals[3], double cov[9]) {
  /* NS/VoxelGrid.cpp:317: covariance = evecs * evals * evecs.inverse(), evals a dense Matrix3d */
  double ve[9], vinv[9];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) ve[r * 3 + k] = evecs[r * 3 + k] * evals[k];
  orc_inv3(evecs, vinv);
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++)
      cov[r * 3 + k] = ve[r * 3 + 0] * vinv[0 * 3 + k] + (ve[r * 3 + 1] * vinv[1 * 3 + k] + ve[r * 3 + 2] * vinv[2 * 3 + k]);
}

void orc_grid_free(orc_grid *g) {
  if (!g) return;
  free(g->npts); free(g->ppv); free(g->centroid); free(g->icov); free(g->sval);
  free(g->tmp_centroid); free(g->tmp_cov); free(g->tmp_sval);
  if (g->octree) orc_octree_free(g->octree);
  free(g);
}

void orc_eig3_jacobi(const double a[9], double evals[3], double evecs[9]); /* pcl_oracle.c */

orc_grid *orc_grid_build(const void *pts, size_t n, size_t stride, size_t w_off, float leaf) {
  return orc_grid_build_mode(pts, n, stride, w_off, leaf, 0);
}

/* mode 1 = the deltas of pcl::VoxelGridCovariance<PointXYZI>::applyFilter (PCL 1.7 filters/impl/
 * voxel_grid_covariance.hpp) against the in-tree grid, as SURVEY.md 8c lists them: sum p p^T starts at 0 (not I),
 * Eigen::SelfAdjointEigenSolver (restated as cyclic Jacobi: same eigen-pairs to rounding), a leaf whose eigenvalues
 * are rejected keeps its zero icov_ and STAYS in the centroid kd-tree, +-inf in icov_ drops the leaf, and there is
 * no static weight (w == 1). Cell geometry and the radius search stay as in mode 0 (for res = 1 PCL's
 * floor(x * inv_leaf) and the in-tree floor(x / leaf) coincide). */
orc_grid *orc_grid_build_mode(const void *pts, size_t n, size_t stride, size_t w_off, float leaf, int mode) {
  if (n == 0) return NULL;
  orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
  g->leaf = leaf;
  g->mode = mode;
  /* findBoundaries, NS/VoxelGrid.cpp:402-430 */
  g->max_x = g->max_y = g->max_z = -FLT_MAX;
  g->min_x = g->min_y = g->min_z = FLT_MAX;
  for (size_t i = 0; i < n; i++) {
    const float *p = pt_xyz(pts, stride, i);
    g->max_x = (g->max_x < p[0]) ? p[0] : g->max_x;
    g->max_y = (g->max_y < p[1]) ? p[1] : g->max_y;
    g->max_z = (g->max_z < p[2]) ? p[2] : g->max_z;
    g->min_x = (p[0] < g->min_x) ? p[0] : g->min_x;
    g->min_y = (p[1] < g->min_y) ? p[1] : g->min_y;
    g->min_z = (p[2] < g->min_z) ? p[2] : g->min_z;
  }
  /* :366-399 */
  g->real_max[0] = g->max_b[0] = (int)floorf(g->max_x / leaf);
  g->real_max[1] = g->max_b[1] = (int)floorf(g->max_y / leaf);
  g->real_max[2] = g->max_b[2] = (int)floorf(g->max_z / leaf);
  g->real_min[0] = g->min_b[0] = (int)floorf(g->min_x / leaf);
  g->real_min[1] = g->min_b[1] = (int)floorf(g->min_y / leaf);
  g->real_min[2] = g->min_b[2] = (int)floorf(g->min_z / leaf);
  g->max_b[0] = round_up(g->max_b[0], 16); g->max_b[1] = round_up(g->max_b[1], 16); g->max_b[2] = round_up(g->max_b[2], 8);
  g->min_b[0] = round_down(g->min_b[0], 16); g->min_b[1] = round_down(g->min_b[1], 16); g->min_b[2] = round_down(g->min_b[2], 8);
  for (int a = 0; a < 3; a++) g->vgrid[a] = g->max_b[a] - g->min_b[a] + 1;
  g->voxel_num = (g->vgrid[0] > 0 && g->vgrid[1] > 0 && g->vgrid[2] > 0) ? g->vgrid[0] * g->vgrid[1] * g->vgrid[2] : 0;
  size_t vn = (size_t)g->voxel_num;
  /* initialize(), :73-98 */
  g->npts = (int *)calloc(vn, sizeof(int));
  g->ppv = (int *)calloc(vn, sizeof(int));
  g->centroid = (double *)calloc(vn * 3, sizeof(double));
  g->icov = (double *)calloc(vn * 9, sizeof(double));
  g->sval = (double *)calloc(vn, sizeof(double));
  g->tmp_centroid = (double *)calloc(vn * 3, sizeof(double));
  g->tmp_cov = (double *)calloc(vn * 9, sizeof(double));
  g->tmp_sval = (double *)calloc(vn, sizeof(double));

  /* octree over the same voxel ids (:336-349) — used only by the fitness score */
  g->octree = orc_octree_build(pts, n, stride, leaf);

  /* scatterPointsToVoxelGrid, :244-270 */
  for (size_t pid = 0; pid < n; pid++) {
    const float *p = pt_xyz(pts, stride, pid);
    int ix = (int)floorf(p[0] / leaf), iy = (int)floorf(p[1] / leaf), iz = (int)floorf(p[2] / leaf);
    int vid = vid3(g, ix, iy, iz);
    double x = p[0], y = p[1], z = p[2];
    if (g->npts[vid] == 0) {
      double *tc = g->tmp_cov + (size_t)vid * 9;
      memset(tc, 0, 9 * sizeof(double));
      if (mode == 0) tc[0] = tc[4] = tc[8] = 1.0; /* seeded with identity (:257); PCL starts from zero */
      g->ppv[vid] = 0;
    }
    double *ts = g->tmp_centroid + (size_t)vid * 3;
    ts[0] += x; ts[1] += y; ts[2] += z;
    double *tc = g->tmp_cov + (size_t)vid * 9;
    tc[0] += x * x; tc[1] += x * y; tc[2] += x * z;
    tc[3] += y * x; tc[4] += y * y; tc[5] += y * z;
    tc[6] += z * x; tc[7] += z * y; tc[8] += z * z;
    g->npts[vid]++;
    g->ppv[vid]++;
    g->tmp_sval[vid] += (mode == 0) ? pt_w(pts, stride, w_off, pid) : 1.0f;
  }

  /* computeCentroidAndCovariance, :273-323 */
  for (int ix = g->real_min[0]; ix <= g->real_max[0]; ix++)
    for (int iy = g->real_min[1]; iy <= g->real_max[1]; iy++)
      for (int iz = g->real_min[2]; iz <= g->real_max[2]; iz++) {
        int i = vid3(g, ix, iy, iz);
        int ipn = g->npts[i];
        double pn = (double)ipn;
        const double *ps = g->tmp_centroid + (size_t)i * 3;
        double *c = g->centroid + (size_t)i * 3;
        if (ipn > 0) {
          c[0] = ps[0] / pn; c[1] = ps[1] / pn; c[2] = ps[2] / pn;
          g->sval[i] = g->tmp_sval[i] / pn;
        }
        if (ipn >= 6) { /* min_points_per_voxel_, :37 */
          double cov[9];
          orc_cov_expr(g->tmp_cov + (size_t)i * 9, ps, pn, cov);
          double evals[3], evecs[9];
          if (mode == 0) orc_eig3(cov, evals, evecs);
          else orc_eig3_jacobi(cov, evals, evecs);
          if (evals[0] < 0 || evals[1] < 0 || evals[2] <= 0) { /* :303-306 */
            if (mode == 0) g->ppv[i] = -1;
            /* PCL: nr_points = -1 but the centroid was already pushed into the kd-tree cloud, and ndt.hpp never
             * looks at nr_points: the leaf keeps contributing with icov_ == 0 (score only). Restated by leaving
             * ppv >= 6 and the calloc'ed zero icov. */
            continue;
          }
          double min_ev = evals[2] * 0.01;
          if (evals[0] < min_ev) { /* :310-318 */
            evals[0] = min_ev;
            if (evals[1] < min_ev) evals[1] = min_ev;
            orc_cov_floor_rebuild(evecs, evals, cov);
          }
          orc_inv3(cov, g->icov + (size_t)i * 9); /* :320 */
          if (mode == 1) { /* icov_.maxCoeff() == inf || icov_.minCoeff() == -inf -> leaf dropped */
            const double *ic = g->icov + (size_t)i * 9;
            for (int k = 0; k < 9; k++)
              if (isinf(ic[k])) { g->ppv[i] = -1; break; }
          }
        }
      }
  return g;
}

void orc_grid_info(const orc_grid *g, int info[13]) {
  for (int a = 0; a < 3; a++) {
    info[a] = g->vgrid[a]; info[3 + a] = g->min_b[a]; info[6 + a] = g->real_min[a]; info[9 + a] = g->real_max[a];
  }
  info[12] = g->voxel_num;
}

void orc_grid_counts(const orc_grid *g, int *occupied, int *valid) {
  int o = 0, v = 0;
  for (int i = 0; i < g->voxel_num; i++) { if (g->npts[i] > 0) o++; if (g->ppv[i] >= 6) v++; }
  if (occupied) *occupied = o;
  if (valid) *valid = v;
}

int orc_grid_cell(const orc_grid *g, int vid, int *npts, int *ppv, double centroid[3], double icov[9], double *weight) {
  if (vid < 0 || vid >= g->voxel_num) return -1;
  if (npts) *npts = g->npts[vid];
  if (ppv) *ppv = g->ppv[vid];
  if (centroid) memcpy(centroid, g->centroid + (size_t)vid * 3, 3 * sizeof(double));
  if (icov) memcpy(icov, g->icov + (size_t)vid * 9, 9 * sizeof(double));
  if (weight) *weight = g->sval[vid];
  return 0;
}

int orc_grid_dump(const orc_grid *g, int *ints, double *dbls, int cap) {
  int k = 0;
  for (int iz = g->real_min[2]; iz <= g->real_max[2]; iz++)
    for (int iy = g->real_min[1]; iy <= g->real_max[1]; iy++)
      for (int ix = g->real_min[0]; ix <= g->real_max[0]; ix++) {
        int i = vid3(g, ix, iy, iz);
        if (g->npts[i] <= 0) continue;
        if (k < cap) {
          if (ints) { ints[5 * k] = ix; ints[5 * k + 1] = iy; ints[5 * k + 2] = iz; ints[5 * k + 3] = g->npts[i]; ints[5 * k + 4] = g->ppv[i]; }
          if (dbls) {
            memcpy(dbls + 13 * (size_t)k, g->centroid + (size_t)i * 3, 3 * sizeof(double));
            memcpy(dbls + 13 * (size_t)k + 3, g->icov + (size_t)i * 9, 9 * sizeof(double));
            dbls[13 * (size_t)k + 12] = g->sval[i];
          }
        }
        k++;
      }
  return k;
}

int orc_grid_radius_search(const orc_grid *g, const float p[3], float radius, int *ids, int cap) {
  /* NS/VoxelGrid.cpp:432-480, max_nn = INT_MAX */
  float tx = p[0], ty = p[1], tz = p[2], leaf = g->leaf;
  int max_x = (int)floorf((tx + radius) / leaf), max_y = (int)floorf((ty + radius) / leaf), max_z = (int)floorf((tz + radius) / leaf);
  int min_x = (int)floorf((tx - radius) / leaf), min_y = (int)floorf((ty - radius) / leaf), min_z = (int)floorf((tz - radius) / leaf);
  max_x = (max_x > g->real_max[0]) ? g->real_max[0] : max_x;
  max_y = (max_y > g->real_max[1]) ? g->real_max[1] : max_y;
  max_z = (max_z > g->real_max[2]) ? g->real_max[2] : max_z;
  min_x = (min_x < g->real_min[0]) ? g->real_min[0] : min_x;
  min_y = (min_y < g->real_min[1]) ? g->real_min[1] : min_y;
  min_z = (min_z < g->real_min[2]) ? g->real_min[2] : min_z;
  int nn = 0;
  for (int ix = min_x; ix <= max_x; ix++)
    for (int iy = min_y; iy <= max_y; iy++)
      for (int iz = min_z; iz <= max_z; iz++) {
        int vid = vid3(g, ix, iy, iz);
        if (g->ppv[vid] >= 6) {
          const double *c = g->centroid + (size_t)vid * 3;
          double cx = c[0] - (double)tx, cy = c[1] - (double)ty, cz = c[2] - (double)tz;
          double distance = sqrt(cx * cx + cy * cy + cz * cz);
          if (distance < radius) {
            if (nn < cap) ids[nn] = vid;
            nn++;
          }
        }
      }
  return nn;
}

/* nearestVoxel + nearestNeighborDistance, NS/VoxelGrid.cpp:483-543 */
static double grid_nn_distance(const orc_grid *g, const float q[3], float max_range) {
  float b[6];
  orc_octree_nearest(g->octree, q, b);
  float leaf = g->leaf;
  int lx = (int)floorf(b[0] / leaf), ly = (int)floorf(b[1] / leaf), lz = (int)floorf(b[2] / leaf);
  int ux = (int)floorf(b[3] / leaf), uy = (int)floorf(b[4] / leaf), uz = (int)floorf(b[5] / leaf);
  double min_dist = DBL_MAX;
  int nn_vid = -1;
  float qx = q[0], qy = q[1], qz = q[2];
  for (int i = lx; i <= ux; i++)
    for (int j = ly; j <= uy; j++)
      for (int k = lz; k <= uz; k++) {
        int vid = vid3(g, i, j, k);
        const double *c = g->centroid + (size_t)vid * 3;
        if (g->npts[vid] > 0) {
          double d = sqrt((qx - c[0]) * (qx - c[0]) + (qy - c[1]) * (qy - c[1]) + (qz - c[2]) * (qz - c[2]));
          if (d < min_dist) { min_dist = d; nn_vid = vid; }
        }
      }
  if (nn_vid < 0) return DBL_MAX; /* the reference would index centroid_[-1]; unreachable for a non-empty node */
  const double *c = g->centroid + (size_t)nn_vid * 3;
  double md = sqrt((qx - c[0]) * (qx - c[0]) + (qy - c[1]) * (qy - c[1]) + (qz - c[2]) * (qz - c[2]));
  if (md >= max_range) return DBL_MAX;
  return md;
}

/* ======================================================================== */
/* NDT                                                                       */
/* ======================================================================== */

struct orc_ndt {
  float resolution;
  double step_size, trans_eps, outlier_ratio;
  int max_iter;
  double d1, d2;
  orc_grid *grid;
  /* angle-derivative vectors, NI/NormalDistributionsTransform.h:150-158 */
  double j_ang[8][3];
  double h_ang[15][3];
  int real_iterations;
  int mode; /* 0 NDTMANUAL (in-tree), 1 PCL-semantics (the "NDT" plugin, unpinned) */
  /* trace */
  double *trace; int trace_cap; int *trace_n;
  /* per-align bookkeeping */
  long pairs_total; int passes;
};

orc_ndt *orc_ndt_create(float res, float step_size, float trans_eps, int max_iter, float outlier_ratio) {
  orc_ndt *h = (orc_ndt *)calloc(1, sizeof(orc_ndt));
  h->resolution = res;
  h->step_size = step_size;     /* float -> double, as setStepSize(double) receives it */
  h->trans_eps = trans_eps;
  h->max_iter = max_iter;
  h->outlier_ratio = outlier_ratio;
  return h;
}

void orc_ndt_free(orc_ndt *h) {
  if (!h) return;
  orc_grid_free(h->grid);
  free(h);
}

int orc_ndt_set_target(orc_ndt *h, const void *pts, size_t n, size_t stride, size_t w_off) {
  /* NS/NormalDistributionsTransform.cpp:89-98 */
  if (n == 0) return 0;
  orc_grid_free(h->grid);
  h->grid = orc_grid_build_mode(pts, n, stride, w_off, h->resolution, h->mode);
  return h->grid ? 0 : -1;
}

void orc_ndt_set_mode(orc_ndt *h, int mode) { h->mode = mode; }

const orc_grid *orc_ndt_grid(const orc_ndt *h) { return h->grid; }

static void ndt_gauss(orc_ndt *h) {
  /* NS/NormalDistributionsTransform.cpp:315-321 */
  double c1 = 10 * (1 - h->outlier_ratio);
  double c2 = h->outlier_ratio / pow(h->resolution, 3);
  double d3 = -log(c2);
  h->d1 = -log(c1 + c2) - d3;
  h->d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / h->d1);
}

void orc_ndt_gauss(const orc_ndt *hc, double *d1, double *d2) {
  orc_ndt tmp = *hc;
  ndt_gauss(&tmp);
  *d1 = tmp.d1; *d2 = tmp.d2;
}

static void angle_derivatives(orc_ndt *h, const double pose[6]) {
  /* NS/NormalDistributionsTransform.cpp:523-645 (compute_hessian defaults to true) */
  double cx, cy, cz, sx, sy, sz;
  if (fabs(pose[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(pose[3]); sx = sin(pose[3]); }
  if (fabs(pose[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(pose[4]); sy = sin(pose[4]); }
  if (fabs(pose[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(pose[5]); sz = sin(pose[5]); }
  double (*j)[3] = h->j_ang;
  j[0][0] = -sx * sz + cx * sy * cz; j[0][1] = -sx * cz - cx * sy * sz; j[0][2] = -cx * cy;          /* a */
  j[1][0] = cx * sz + sx * sy * cz;  j[1][1] = cx * cz - sx * sy * sz;  j[1][2] = -sx * cy;          /* b */
  j[2][0] = -sy * cz;                j[2][1] = sy * sz;                 j[2][2] = cy;                /* c */
  j[3][0] = sx * cy * cz;            j[3][1] = -sx * cy * sz;           j[3][2] = sx * sy;           /* d */
  j[4][0] = -cx * cy * cz;           j[4][1] = cx * cy * sz;            j[4][2] = -cx * sy;          /* e */
  j[5][0] = -cy * sz;                j[5][1] = -cy * cz;                j[5][2] = 0;                 /* f */
  j[6][0] = cx * cz - sx * sy * sz;  j[6][1] = -cx * sz - sx * sy * cz; j[6][2] = 0;                 /* g */
  j[7][0] = sx * cz + cx * sy * sz;  j[7][1] = cx * sy * cz - sx * sz;  j[7][2] = 0;                 /* h */
  double (*q)[3] = h->h_ang;
  q[0][0] = -cx * sz - sx * sy * cz; q[0][1] = -cx * cz + sx * sy * sz; q[0][2] = sx * cy;           /* a2 */
  q[1][0] = -sx * sz + cx * sy * cz; q[1][1] = -cx * sy * sz - sx * cz; q[1][2] = -cx * cy;          /* a3 */
  q[2][0] = cx * cy * cz;            q[2][1] = -cx * cy * sz;           q[2][2] = cx * sy;           /* b2 */
  q[3][0] = sx * cy * cz;            q[3][1] = -sx * cy * sz;           q[3][2] = sx * sy;           /* b3 */
  q[4][0] = -sx * cz - cx * sy * sz; q[4][1] = sx * sz - cx * sy * cz;  q[4][2] = 0;                 /* c2 */
  q[5][0] = cx * cz - sx * sy * sz;  q[5][1] = -sx * sy * cz - cx * sz; q[5][2] = 0;                 /* c3 */
  q[6][0] = -cy * cz;                q[6][1] = cy * sz;                 q[6][2] = sy;                /* d1 */
  q[7][0] = -sx * sy * cz;           q[7][1] = sx * sy * sz;            q[7][2] = sx * cy;           /* d2 */
  q[8][0] = cx * sy * cz;            q[8][1] = -cx * sy * sz;           q[8][2] = -cx * cy;          /* d3 */
  q[9][0] = sy * sz;                 q[9][1] = sy * cz;                 q[9][2] = 0;                 /* e1 */
  q[10][0] = -sx * cy * sz;          q[10][1] = -sx * cy * cz;          q[10][2] = 0;                /* e2 */
  q[11][0] = cx * cy * sz;           q[11][1] = cx * cy * cz;           q[11][2] = 0;                /* e3 */
  q[12][0] = -cy * cz;               q[12][1] = cy * sz;                q[12][2] = 0;                /* f1 */
  q[13][0] = -cx * sz - sx * sy * cz; q[13][1] = -cx * cz + sx * sy * sz; q[13][2] = 0;              /* f2 */
  q[14][0] = -sx * sz + cx * sy * cz; q[14][1] = -cx * sy * sz - sx * cz; q[14][2] = 0;              /* f3 */
}

/* Eigen's fixed-size-3 reduction associates as a0 + (a1 + a2) (Core/Redux.h unroller halves the
 * range; confirmed bit-for-bit against the vendored Eigen by oracle/_ref: inv3/dot3/mv3 fixtures) */
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
static inline void mv3(const double m[9], const double v[3], double out[3]) {
  out[0] = m[0] * v[0] + (m[1] * v[1] + m[2] * v[2]);
  out[1] = m[3] * v[0] + (m[4] * v[1] + m[5] * v[2]);
  out[2] = m[6] * v[0] + (m[7] * v[1] + m[8] * v[2]);
}
double orc_dot3(const double a[3], const double b[3]) { return dot3(a, b); }
void orc_mv3(const double m[9], const double v[3], double out[3]) { mv3(m, v, out); }

/* point_gradient as 6 columns of 3; point_hessian as 18x6 -> ph[row][col] */
static void point_derivatives(const orc_ndt *h, const double x[3], double pg[6][3], double ph[18][6], int compute_hessian) {
  /* NS/NormalDistributionsTransform.cpp:448-482 */
  pg[3][1] = dot3(x, h->j_ang[0]); pg[3][2] = dot3(x, h->j_ang[1]);
  pg[4][0] = dot3(x, h->j_ang[2]); pg[4][1] = dot3(x, h->j_ang[3]); pg[4][2] = dot3(x, h->j_ang[4]);
  pg[5][0] = dot3(x, h->j_ang[5]); pg[5][1] = dot3(x, h->j_ang[6]); pg[5][2] = dot3(x, h->j_ang[7]);
  if (compute_hessian) {
    double a[3] = {0, dot3(x, h->h_ang[0]), dot3(x, h->h_ang[1])};
    double b[3] = {0, dot3(x, h->h_ang[2]), dot3(x, h->h_ang[3])};
    double c[3] = {0, dot3(x, h->h_ang[4]), dot3(x, h->h_ang[5])};
    double d[3] = {dot3(x, h->h_ang[6]), dot3(x, h->h_ang[7]), dot3(x, h->h_ang[8])};
    double e[3] = {dot3(x, h->h_ang[9]), dot3(x, h->h_ang[10]), dot3(x, h->h_ang[11])};
    double f[3] = {dot3(x, h->h_ang[12]), dot3(x, h->h_ang[13]), dot3(x, h->h_ang[14])};
    for (int r = 0; r < 3; r++) {
      ph[9 + r][3] = a[r];  ph[12 + r][3] = b[r]; ph[15 + r][3] = c[r];
      ph[9 + r][4] = b[r];  ph[12 + r][4] = d[r]; ph[15 + r][4] = e[r];
      ph[9 + r][5] = c[r];  ph[12 + r][5] = e[r]; ph[15 + r][5] = f[r];
    }
  }
}

static double update_derivatives(const orc_ndt *h, double g[6], double H[36], double pg[6][3], double ph[18][6],
                                 const double xt[3], const double cinv[9], int compute_hessian) {
  /* NS/NormalDistributionsTransform.cpp:485-520 */
  double cx[3];
  mv3(cinv, xt, cx);
  double e = exp(-h->d2 * dot3(xt, cx) / 2);
  double score_inc = -h->d1 * e;
  e = h->d2 * e;
  if (e > 1 || e < 0 || e != e) return 0.0;
  e *= h->d1;
  for (int i = 0; i < 6; i++) {
    double cov_dxd_pi[3];
    mv3(cinv, pg[i], cov_dxd_pi);
    g[i] += dot3(xt, cov_dxd_pi) * e;
    if (compute_hessian) {
      for (int j = 0; j < 6; j++) {
        double cj[3], chij[3], phij[3] = {ph[3 * i][j], ph[3 * i + 1][j], ph[3 * i + 2][j]};
        mv3(cinv, pg[j], cj);
        mv3(cinv, phij, chij);
        H[i * 6 + j] += e * (-h->d2 * dot3(xt, cov_dxd_pi) * dot3(xt, cj) + dot3(xt, chij) + dot3(pg[j], cov_dxd_pi));
      }
    }
  }
  return score_inc;
}

static void update_hessian(const orc_ndt *h, double H[36], double pg[6][3], double ph[18][6], const double xt[3], const double cinv[9]) {
  /* NS/NormalDistributionsTransform.cpp:875-899 */
  double cx[3];
  mv3(cinv, xt, cx);
  double e = h->d2 * exp(-h->d2 * dot3(xt, cx) / 2);
  if (e > 1 || e < 0 || e != e) return;
  e *= h->d1;
  for (int i = 0; i < 6; i++) {
    double cov_dxd_pi[3];
    mv3(cinv, pg[i], cov_dxd_pi);
    for (int j = 0; j < 6; j++) {
      double cj[3], chij[3], phij[3] = {ph[3 * i][j], ph[3 * i + 1][j], ph[3 * i + 2][j]};
      mv3(cinv, pg[j], cj);
      mv3(cinv, phij, chij);
      H[i * 6 + j] += e * (-h->d2 * dot3(xt, cov_dxd_pi) * dot3(xt, cj) + dot3(xt, chij) + dot3(pg[j], cov_dxd_pi));
    }
  }
}

static void init_point_mats(double pg[6][3], double ph[18][6]) {
  /* NS/NormalDistributionsTransform.cpp:414-416 */
  memset(pg, 0, sizeof(double) * 18);
  pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
  memset(ph, 0, sizeof(double) * 18 * 6);
}

double orc_ndt_derivatives(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16],
                           const double p[6], int compute_hessian, double g[6], double H[36], long *pairs) {
  /* NS/NormalDistributionsTransform.cpp:391-445 */
  if (h->d1 == 0 && h->d2 == 0) ndt_gauss(h);
  memset(g, 0, 6 * sizeof(double));
  memset(H, 0, 36 * sizeof(double));
  angle_derivatives(h, p);
  double pg[6][3], ph[18][6];
  init_point_mats(pg, ph);
  double score = 0;
  long np = 0;
  int ids[64];
  for (size_t idx = 0; idx < n; idx++) {
    const float *xs = pt_xyz(src, stride, idx);
    float xt_f[3];
    orc_transform_point(T, xs, xt_f);
    int k = orc_grid_radius_search(h->grid, xt_f, h->resolution, ids, 64);
    for (int i = 0; i < k; i++) {
      int vid = ids[i];
      double x[3] = {xs[0], xs[1], xs[2]};
      const double *c = h->grid->centroid + (size_t)vid * 3;
      double xt[3] = {(double)xt_f[0] - c[0], (double)xt_f[1] - c[1], (double)xt_f[2] - c[2]};
      const double *cinv = h->grid->icov + (size_t)vid * 9;
      double sv = h->grid->sval[vid];
      point_derivatives(h, x, pg, ph, compute_hessian);
      score += sv * update_derivatives(h, g, H, pg, ph, xt, cinv, compute_hessian); /* weight on score only, :440-441 */
      np++;
    }
  }
  if (pairs) *pairs = np;
  h->pairs_total += np; h->passes++;
  return score;
}

void orc_ndt_hessian(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16], const double p[6], double H[36]) {
  /* NS/NormalDistributionsTransform.cpp:901-936.
   * NOTE: the reference leaves point_gradient / point_hessian UNINITIALISED here (:911-912, undefined
   * behaviour: the identity block and the zero entries are stack garbage). Restated with the
   * initialisation computeDerivatives uses (:414-416), i.e. the evidently intended values. The angle
   * vectors are whatever the last computeDerivatives call left (p is unused in the reference too). */
  (void)p;
  memset(H, 0, 36 * sizeof(double));
  double pg[6][3], ph[18][6];
  init_point_mats(pg, ph);
  int ids[64];
  long np = 0;
  for (size_t idx = 0; idx < n; idx++) {
    const float *xs = pt_xyz(src, stride, idx);
    float xt_f[3];
    orc_transform_point(T, xs, xt_f);
    int k = orc_grid_radius_search(h->grid, xt_f, h->resolution, ids, 64);
    for (int i = 0; i < k; i++) {
      int vid = ids[i];
      double x[3] = {xs[0], xs[1], xs[2]};
      const double *c = h->grid->centroid + (size_t)vid * 3;
      double xt[3] = {(double)xt_f[0] - c[0], (double)xt_f[1] - c[1], (double)xt_f[2] - c[2]};
      point_derivatives(h, x, pg, ph, 1);
      update_hessian(h, H, pg, ph, xt, h->grid->icov + (size_t)vid * 9);
      np++;
    }
  }
  h->pairs_total += np; h->passes++;
}

/* ---- More-Thuente (NS/NormalDistributionsTransform.cpp:69-75, 760-873) -- */
static double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
static double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

static double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_t_next = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_t_next);
    return fmax(a_t + 0.66 * (a_u - a_t), a_t_next);
  } else {
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    double w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

static int update_interval_mt(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u, double *g_u, double a_t, double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  else if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  else if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  return 1;
}

static void trace_row(orc_ndt *h, const double p[6], double score, double step, int kind) {
  if (!h->trace || !h->trace_n) return;
  if (*h->trace_n < h->trace_cap) {
    double *r = h->trace + (size_t)(*h->trace_n) * 9;
    memcpy(r, p, 6 * sizeof(double)); r[6] = score; r[7] = step; r[8] = kind;
  }
  (*h->trace_n)++;
}

void orc_ndt_set_trace(orc_ndt *h, double *buf, int cap_rows, int *n_rows) {
  h->trace = buf; h->trace_cap = cap_rows; h->trace_n = n_rows;
  if (n_rows) *n_rows = 0;
}

static double step_length_mt(orc_ndt *h, const void *src, size_t n, size_t stride, const double x[6], double step_dir[6],
                             double step_init, double step_max, double step_min, double *score, double g[6], double H[36],
                             float final_T[16], int *ls_iters) {
  /* NS/NormalDistributionsTransform.cpp:648-756 */
  double phi_0 = -(*score);
  double d_phi_0 = 0;
  for (int i = 0; i < 6; i++) d_phi_0 += g[i] * step_dir[i];
  d_phi_0 = -d_phi_0;
  double x_t[6];
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 *= -1;
    for (int i = 0; i < 6; i++) step_dir[i] *= -1;
  }
  int max_step_iterations = 10, step_iterations = 0;
  double mu = 1.e-4, nu = 0.9, a_l = 0, a_u = 0;
  double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
  double g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
  double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
  double g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
  int interval_converged = (step_max - step_min) > 0, open_interval = 1; /* sic (:683): true when max > min */
  double a_t = step_init;
  a_t = fmin(a_t, step_max);
  a_t = fmax(a_t, step_min);
  for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
  orc_pose_to_matrix(x_t, final_T);
  *score = orc_ndt_derivatives(h, src, n, stride, final_T, x_t, 1, g, H, NULL);
  trace_row(h, x_t, *score, a_t, 0);
  double phi_t = -(*score);
  double d_phi_t = 0;
  for (int i = 0; i < 6; i++) d_phi_t += g[i] * step_dir[i];
  d_phi_t = -d_phi_t;
  double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
  double d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval) a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = (a_t < step_max) ? a_t : step_max;
    a_t = (a_t > step_min) ? a_t : step_min;
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
    orc_pose_to_matrix(x_t, final_T);
    *score = orc_ndt_derivatives(h, src, n, stride, final_T, x_t, 0, g, H, NULL);
    trace_row(h, x_t, *score, a_t, 1);
    {
      double dd = 0;
      for (int i = 0; i < 6; i++) dd += g[i] * step_dir[i];
      if (h->mode == 0) {
        phi_t -= *score; /* sic (:726): accumulates instead of assigning */
        d_phi_t -= dd;   /* sic (:727) */
      } else {           /* PCL ndt.hpp assigns */
        phi_t = -(*score);
        d_phi_t = -dd;
      }
    }
    psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = 0;
      f_l += phi_0 - mu * d_phi_0 * a_l; g_l += mu * d_phi_0;
      f_u += phi_0 - mu * d_phi_0 * a_u; g_u += mu * d_phi_0;
    }
    if (open_interval) interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
    else interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) {
    orc_ndt_hessian(h, src, n, stride, final_T, x_t, H);
    trace_row(h, x_t, *score, a_t, 2);
  }
  h->real_iterations += step_iterations;
  *ls_iters += step_iterations;
  return a_t;
}

int orc_ndt_align(orc_ndt *h, const void *src, size_t n, size_t stride, const float guess[16], float out_pose[16], orc_ndt_stats *stats) {
  /* NS/Registration.cpp:76-95 + NS/NormalDistributionsTransform.cpp:310-389 */
  if (!h->grid) return -1;
  int nr_iterations = 0, converged = 0, ls_iters = 0;
  h->pairs_total = 0; h->passes = 0;
  ndt_gauss(h);
  float final_T[16];
  /* final_transformation_ = Identity (Registration.cpp:80); replaced by guess if guess != I (:323-328).
   * Either way the first pass sees source transformed by `guess` (identity transform is exact). */
  memcpy(final_T, guess, sizeof(final_T));
  double p[6], delta_p[6], g[6], H[36];
  orc_matrix_to_pose(final_T, p);
  double score = orc_ndt_derivatives(h, src, n, stride, final_T, p, 1, g, H, NULL);
  trace_row(h, p, score, 0.0, 0);
  double trans_probability = 0;
  double delta_p_norm = 0;
  while (!converged) {
    double negg[6];
    for (int i = 0; i < 6; i++) negg[i] = -g[i];
    orc_svd6_solve(H, negg, delta_p); /* :353-355 */
    double s2 = 0;
    for (int i = 0; i < 6; i++) s2 += delta_p[i] * delta_p[i];
    delta_p_norm = sqrt(s2);
    if (delta_p_norm == 0 || delta_p_norm != delta_p_norm) { /* :359-363 */
      trans_probability = score / (double)(int)n;
      converged = (delta_p_norm == delta_p_norm);
      goto done;
    }
    for (int i = 0; i < 6; i++) delta_p[i] /= delta_p_norm;
    delta_p_norm = step_length_mt(h, src, n, stride, p, delta_p, delta_p_norm, h->step_size, h->trans_eps / 2, &score, g, H, final_T, &ls_iters);
    for (int i = 0; i < 6; i++) { delta_p[i] *= delta_p_norm; p[i] = p[i] + delta_p[i]; }
    if (nr_iterations > h->max_iter || (nr_iterations && (fabs(delta_p_norm) < h->trans_eps))) converged = 1; /* :379-381 */
    nr_iterations++;
  }
  if (n) trans_probability = score / (double)n;
done:
  memcpy(out_pose, final_T, sizeof(final_T));
  if (stats) {
    stats->iterations = nr_iterations; stats->ls_iterations = ls_iters; stats->passes = h->passes; stats->converged = converged;
    stats->score = score; stats->trans_probability = trans_probability;
    stats->pairs_per_pass = h->passes ? (double)h->pairs_total / h->passes : 0;
    memcpy(stats->final_p, p, sizeof(p));
  }
  return 0;
}

double orc_ndt_fitness(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16]) {
  /* NS/NormalDistributionsTransform.cpp:940-965, max_range = DBL_MAX -> (float)DBL_MAX = +inf */
  double fitness = 0.0;
  int nr = 0;
  float max_range_f = (float)INFINITY;
  for (size_t i = 0; i < n; i++) {
    float q[3];
    orc_transform_point(T, pt_xyz(src, stride, i), q);
    double d = grid_nn_distance(h->grid, q, max_range_f);
    if (d < DBL_MAX) { fitness += d; nr++; }
  }
  if (nr > 0) return fitness / nr;
  return DBL_MAX;
}

/* test hook: overwrite one cell's finalised record (absolute voxel coordinates), so the derivative loops can be
 * compared on IDENTICAL voxel tables (isolates the pass arithmetic from table round-off) */
int orc_grid_set_cell(orc_grid *g, int ix, int iy, int iz, const double centroid[3], const double icov[9], double weight) {
  if (ix < g->real_min[0] || ix > g->real_max[0] || iy < g->real_min[1] || iy > g->real_max[1] || iz < g->real_min[2] || iz > g->real_max[2]) return -1;
  int vid = vid3(g, ix, iy, iz);
  memcpy(g->centroid + (size_t)vid * 3, centroid, 3 * sizeof(double));
  memcpy(g->icov + (size_t)vid * 9, icov, 9 * sizeof(double));
  g->sval[vid] = weight;
  return 0;
}
