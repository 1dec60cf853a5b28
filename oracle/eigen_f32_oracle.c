/*
 * eigen_f32_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lsa_oracle.h header).
 *
 * The f32 Eigen calls on the path, restated operation by operation from the reference's vendored Eigen 3.2.92
 * (lidar_localization/third_party/eigen3/Eigen/src/...) and PINNED bit for bit against that very code compiled in place
 * (oracle/ref_blocks.cpp -> tests/golden/eigen_blocks.json: rot3f, det6f, inv6f, mul3f, icpacc):
 *
 *   orc_rotation_f32   Transform<float,3,Affine>::rotation() — NS/NormalDistributionsTransform.cpp:336 calls it in front of
 *                      eulerAngles(0,1,2).  Geometry/Transform.h:1059-1074 (computeRotationScaling): a two-sided
 *                      JacobiSVD<Matrix3f> (SVD/JacobiSVD.h:676-780, 2x2 kernel :408-435, Jacobi/Jacobi.h:80-113 makeJacobi,
 *                      :298-430 apply_rotation_in_the_plane, non-vectorised for 3x3), x = det(U V^T), R = (U with column 0
 *                      divided by x) V^T.
 *   orc_det6f          Matrix<float,6,6>::determinant() (S/models/registration/icp_registration_manual.cpp:165):
 *                      LU/Determinant.h -> PartialPivLU (LU/PartialPivLU.h:331-385 unblocked_lu), sign * diagonal().prod()
 *                      with Eigen's unrolled reduction tree (Core/Redux.h: halves, recursively).
 *   orc_inv6f          Matrix<float,6,6>::inverse() (:169): PartialPivLU::solve(Identity) (LU/PartialPivLU.h:191-209):
 *                      permute, then the two in-place triangular solves of Core/products/TriangularSolverMatrix.h:95-118 —
 *                      column-oriented, b = (x_i *= 1 / u_ii) and r -= b * l for the rows still open.
 *   orc_mul66f_vec     Hessian.inverse() * B: one coefficient per row, six products summed in the reduction tree.
 *   orc_mul3f          Matrix3f * Matrix3f (:173, and every 3-term f32 dot of the path): a0 b0 + (a1 b1 + a2 b2).
 *   orc_icp_acc        J = [I | -R hat(p)], Hessian += J^T J, B += -J^T e (:156-162); hat(p) is written out as data
 *                      (Sophus, which provides SO3f::hat in the reference, is absent from the tree).
 *
 * No FMA contraction (oracle/Makefile: -ffp-contract=off); x86-64 SSE arithmetic is plain IEEE f32.
 */
#include "lsa_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* Eigen's unrolled 3-term reduction: t0 + (t1 + t2) (Core/Redux.h redux_novec_unroller: Length 3 -> halves 1 | 2) */
static inline float dot3f(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

/* x' = c x + s y, y' = -s x + c y (Jacobi/Jacobi.h:418-426, the non-vectorised path) */
static inline void rot_pair(float *x, float *y, float c, float s) {
  float xi = *x, yi = *y;
  *x = c * xi + s * yi;
  *y = -s * xi + c * yi;
}

/* JacobiRotation::makeJacobi(x, y, z) (Jacobi/Jacobi.h:80-113), real f32 */
static void make_jacobi_f32(float x, float y, float z, float *c, float *s) {
  if (y == 0.0f) { *c = 1.0f; *s = 0.0f; return; }
  float tau = (x - z) / (2.0f * fabsf(y));
  float w = sqrtf(tau * tau + 1.0f);
  float t = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
  float sign_t = t > 0.0f ? 1.0f : -1.0f;
  float n = 1.0f / sqrtf(t * t + 1.0f);
  *s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
  *c = n;
}

void orc_rotation_f32(const float L[9], float R[9]) {
  /* JacobiSVD<Matrix3f>::compute (SVD/JacobiSVD.h:676-780), ComputeFullU | ComputeFullV; W row-major here */
  float W[3][3], U[3][3], V[3][3];
  float scale = 0.0f;
  for (int k = 0; k < 9; k++) { float a = fabsf(L[k]); if (a > scale) scale = a; }   /* cwiseAbs().maxCoeff() */
  if (scale == 0.0f) scale = 1.0f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) { W[r][c] = L[r * 3 + c] / scale; U[r][c] = V[r][c] = (r == c) ? 1.0f : 0.0f; }
  const float precision = 2.0f * FLT_EPSILON, consider_as_zero = 2.0f * 0x1p-149f;   /* 2 * denorm_min */
  int finished = 0;
  while (!finished) {
    finished = 1;
    for (int p = 1; p < 3; p++)
      for (int q = 0; q < p; q++) {
        const float ap = fabsf(W[p][p]), aq = fabsf(W[q][q]);
        const float mx = ap < aq ? aq : ap;                                   /* numext::maxi = std::max */
        float thr = precision * mx;
        thr = consider_as_zero < thr ? thr : consider_as_zero;
        if (fabsf(W[p][q]) > thr || fabsf(W[q][p]) > thr) {
          finished = 0;
          /* real_2x2_jacobi_svd (:408-435) on m = [W(p,p) W(p,q); W(q,p) W(q,q)] */
          float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
          float c1, s1;
          float t = m00 + m11, d = m10 - m01;
          if (d == 0.0f) { s1 = 0.0f; c1 = 1.0f; }
          else {
            float u = t / d;
            float tmp = sqrtf(1.0f + u * u);
            s1 = 1.0f / tmp;
            c1 = u / tmp;
          }
          if (!(c1 == 1.0f && s1 == 0.0f)) { rot_pair(&m00, &m10, c1, s1); rot_pair(&m01, &m11, c1, s1); }   /* m.applyOnTheLeft(0,1,rot1) */
          float cr, sr;
          make_jacobi_f32(m00, m01, m11, &cr, &sr);                                                        /* j_right->makeJacobi(m,0,1) */
          /* *j_left = rot1 * j_right->transpose()  (Jacobi.h:50-56 with the transpose (c, -s)) */
          const float c2 = cr, s2 = -sr;
          const float cl = c1 * c2 - s1 * s2, sl = c1 * s2 + s1 * c2;
          if (!(cl == 1.0f && sl == 0.0f)) {
            for (int i = 0; i < 3; i++) rot_pair(&W[p][i], &W[q][i], cl, sl);   /* m_workMatrix.applyOnTheLeft(p,q,j_left) */
            for (int i = 0; i < 3; i++) rot_pair(&U[i][p], &U[i][q], cl, sl);   /* m_matrixU.applyOnTheRight(p,q,j_left.transpose()) */
          }
          if (!(cr == 1.0f && s2 == 0.0f)) {
            for (int i = 0; i < 3; i++) rot_pair(&W[i][p], &W[i][q], cr, s2);   /* m_workMatrix.applyOnTheRight(p,q,j_right) */
            for (int i = 0; i < 3; i++) rot_pair(&V[i][p], &V[i][q], cr, s2);   /* m_matrixV.applyOnTheRight(p,q,j_right) */
          }
        }
      }
  }
  float sv[3];
  for (int i = 0; i < 3; i++) {   /* step 3: positive singular values */
    float a = fabsf(W[i][i]);
    sv[i] = a;
    if (a != 0.0f) { float f = W[i][i] / a; for (int r = 0; r < 3; r++) U[r][i] *= f; }
  }
  for (int i = 0; i < 3; i++) sv[i] *= scale;
  for (int i = 0; i < 3; i++) {   /* step 4: descending order, first maximum wins */
    int pos = 0;
    float best = sv[i];
    for (int k = i + 1; k < 3; k++) if (sv[k] > best) { best = sv[k]; pos = k - i; }
    if (best == 0.0f) break;
    if (pos) {
      pos += i;
      float ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
      for (int r = 0; r < 3; r++) { float tu = U[r][i]; U[r][i] = U[r][pos]; U[r][pos] = tu; float tv = V[r][i]; V[r][i] = V[r][pos]; V[r][pos] = tv; }
    }
  }
  /* Transform.h:1063: x = (U * V^T).determinant(); LU/Determinant.h:18-24,55-64 (bruteforce_det3_helper) */
  float M[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i][j] = dot3f(U[i][0], V[j][0], U[i][1], V[j][1], U[i][2], V[j][2]);
#define DET3H(a, b, c) (M[0][a] * (M[1][b] * M[2][c] - M[1][c] * M[2][b]))
  const float x = DET3H(0, 1, 2) - DET3H(1, 0, 2) + DET3H(2, 0, 1);
#undef DET3H
  for (int r = 0; r < 3; r++) U[r][0] /= x;   /* m.col(0) /= x */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = dot3f(U[i][0], V[j][0], U[i][1], V[j][1], U[i][2], V[j][2]);
}

/* PartialPivLU of a 6x6 f32 (LU/PartialPivLU.h:331-385).  lu row-major; returns the number of transpositions */
static int lu6f(const float A[36], float lu[6][6], int trans[6]) {
  int nt = 0;
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) lu[r][c] = A[r * 6 + c];
  for (int k = 0; k < 6; k++) {
    int piv = k;
    float big = fabsf(lu[k][k]);
    for (int r = k + 1; r < 6; r++) if (fabsf(lu[r][k]) > big) { big = fabsf(lu[r][k]); piv = r; }   /* first maximum */
    trans[k] = piv;
    if (big != 0.0f) {
      if (piv != k) { for (int c = 0; c < 6; c++) { float t = lu[k][c]; lu[k][c] = lu[piv][c]; lu[piv][c] = t; } nt++; }
      for (int r = k + 1; r < 6; r++) lu[r][k] /= lu[k][k];
    }
    for (int r = k + 1; r < 6; r++)
      for (int c = k + 1; c < 6; c++) lu[r][c] -= lu[r][k] * lu[k][c];
  }
  return nt;
}

float orc_det6f(const float H[36]) {
  float lu[6][6];
  int trans[6];
  const int nt = lu6f(H, lu, trans);
  /* diagonal().prod(): (d0 (d1 d2)) (d3 (d4 d5)) */
  const float p = (lu[0][0] * (lu[1][1] * lu[2][2])) * (lu[3][3] * (lu[4][4] * lu[5][5]));
  return ((nt & 1) ? -1.0f : 1.0f) * p;
}

void orc_inv6f(const float H[36], float Hi[36]) {
  float lu[6][6], X[6][6];
  int trans[6];
  (void)lu6f(H, lu, trans);
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) X[r][c] = (r == c) ? 1.0f : 0.0f;
  for (int k = 0; k < 6; k++)   /* dst = permutationP() * Identity */
    if (trans[k] != k) for (int c = 0; c < 6; c++) { float t = X[k][c]; X[k][c] = X[trans[k]][c]; X[trans[k]][c] = t; }
  for (int i = 0; i < 6; i++)   /* UnitLower */
    for (int j = 0; j < 6; j++) {
      const float b = X[i][j];
      for (int s = i + 1; s < 6; s++) X[s][j] -= b * lu[s][i];
    }
  for (int i = 5; i >= 0; i--) {   /* Upper */
    const float a = 1.0f / lu[i][i];
    for (int j = 0; j < 6; j++) {
      const float b = (X[i][j] *= a);
      for (int s = 0; s < i; s++) X[s][j] -= b * lu[s][i];
    }
  }
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hi[r * 6 + c] = X[r][c];
}

void orc_mul66f_vec(const float Hi[36], const float B[6], float out[6]) {
  for (int r = 0; r < 6; r++) {
    const float *h = Hi + r * 6;
    out[r] = (h[0] * B[0] + (h[1] * B[1] + h[2] * B[2])) + (h[3] * B[3] + (h[4] * B[4] + h[5] * B[5]));
  }
}

void orc_mul3f(const float A[9], const float B[9], float out[9]) {
  float r[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[i * 3 + j] = dot3f(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
  memcpy(out, r, sizeof(r));
}

void orc_icp_acc(const float R[9], const float p[3], const float e[3], float H[36], float B[6]) {
  const float hat[9] = {0.0f, -p[2], p[1], p[2], 0.0f, -p[0], -p[1], p[0], 0.0f};
  float J[3][6];
  memset(J, 0, sizeof(J));
  J[0][0] = J[1][1] = J[2][2] = 1.0f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) J[r][3 + c] = dot3f(-R[r * 3], hat[c], -R[r * 3 + 1], hat[3 + c], -R[r * 3 + 2], hat[6 + c]);
  for (int a = 0; a < 6; a++) {
    for (int b = 0; b < 6; b++) H[a * 6 + b] += dot3f(J[0][a], J[0][b], J[1][a], J[1][b], J[2][a], J[2][b]);
    B[a] += dot3f(-J[0][a], e[0], -J[1][a], e[1], -J[2][a], e[2]);
  }
}
