// eig3_dev.h — internal: symmetric 3x3 eigen-decomposition on the device (cyclic Jacobi, f64).
#ifndef LSA_EIG3_DEV_H_
#define LSA_EIG3_DEV_H_

#include <hip/hip_runtime.h>

// Eigen::SelfAdjointEigenSolver<Matrix3d> stand-in for the PCL-semantics NDT table, and the plane normals of the
// point-to-plane ICP extension (cyclic Jacobi; same eigen-pairs up to rounding and sign,
// and V L V^-1 does not depend on the sign).  Eigenvalues ascending, eigenvectors in the columns of V.
__device__ inline void eig3_jacobi(const double a_in[6] /*xx,xy,xz,yy,yz,zz*/, double evals[3], double V[3][3]) {
  double A[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
  double W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
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
        for (int k = 0; k < 3; k++) { double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 3; k++) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; k++) { double wkp = W[k][p], wkq = W[k][q]; W[k][p] = c * wkp - s * wkq; W[k][q] = s * wkp + c * wkq; }
      }
  }
  int o0 = 0, o1 = 1, o2 = 2;
  double d0 = A[0][0], d1 = A[1][1], d2 = A[2][2];
  if (d1 < d0) { double t = d0; d0 = d1; d1 = t; int ti = o0; o0 = o1; o1 = ti; }
  if (d2 < d0) { double t = d0; d0 = d2; d2 = t; int ti = o0; o0 = o2; o2 = ti; }
  if (d2 < d1) { double t = d1; d1 = d2; d2 = t; int ti = o1; o1 = o2; o2 = ti; }
  evals[0] = d0; evals[1] = d1; evals[2] = d2;
  for (int r = 0; r < 3; r++) { V[r][0] = W[r][o0]; V[r][1] = W[r][o1]; V[r][2] = W[r][o2]; }
}

#endif
