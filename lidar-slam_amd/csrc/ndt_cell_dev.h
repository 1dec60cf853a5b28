// ndt_cell_dev.h — internal, device only: one NDT voxel record from its fixed-point sums (computeCentroidAndCovariance,
// reference: lidar_localization/src/models/registration/ndt_registration_manual/VoxelGrid.cpp:273-323, with the hand-rolled
// eigen-solver of include/.../ndt_registration_manual/SymmetricEigenSolver.h:55-273 and its two quirks).  Shared by the
// full table build (ndt_grid.hip) and the incremental local map (local_map.hip): both must give the same bits per cell.
#ifndef LSA_NDT_CELL_DEV_H_
#define LSA_NDT_CELL_DEV_H_

#include "eig3_dev.h"
#include "voxel_index_dev.h"

// ---- 3x3 symmetric eigen-solver exactly as the reference's hand-rolled one, including its two quirks
// (reference: include/.../ndt_registration_manual/SymmetricEigenSolver.h:55-273; quirks :172-174 and :179-180).
// They matter: with the mis-normalised first eigenvector the "eigenvectors" are an orthogonal but rotated frame,
// and the regularised covariance V L' V^-1 (VoxelGrid.cpp:317) inherits that rotation.
__device__ inline void cross3(const double u[3], const double v[3], double o[3]) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}

__device__ inline void eig3_reference(const double a_in[6] /*xx,xy,xz,yy,yz,zz*/, double evals[3], double V[3][3] /*columns*/) {
  double a00 = a_in[0], a01 = a_in[1], a02 = a_in[2], a11 = a_in[3], a12 = a_in[4], a22 = a_in[5];
  for (int i = 0; i < 3; i++) { evals[i] = 0; for (int j = 0; j < 3; j++) V[i][j] = 0; }
  double max0 = fmax(fabs(a00), fabs(a01)), max1 = fmax(fabs(a02), fabs(a11)), max2 = fmax(fabs(a12), fabs(a22));
  double maxAbs = fmax(fmax(max0, max1), max2);
  if (maxAbs == 0.0) { V[0][0] = V[1][1] = V[2][2] = 1.0; return; }
  double inv = 1.0 / maxAbs;
  a00 *= inv; a01 *= inv; a02 *= inv; a11 *= inv; a12 *= inv; a22 *= inv;
  double norm = a01 * a01 + a02 * a02 + a12 * a12;
  if (norm > 0.0) {
    double tr3 = (a00 + a11 + a22) / 3.0;
    double b00 = a00 - tr3, b11 = a11 - tr3, b22 = a22 - tr3;
    double denom = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
    double c00 = b11 * b22 - a12 * a12, c01 = a01 * b22 - a12 * a02, c02 = a01 * a12 - b11 * a02;
    double det = (b00 * c00 - a01 * c01 + a02 * c02) / (denom * denom * denom);
    double halfDet = fmin(fmax(det * 0.5, -1.0), 1.0);
    double angle = acos(halfDet) / 3.0;
    double beta2 = cos(angle) * 2.0;
    double beta0 = cos(angle + 3.14159265358979323846 * 2.0 / 3.0) * 2.0;
    double beta1 = -(beta0 + beta2);
    evals[0] = tr3 + denom * beta0; evals[1] = tr3 + denom * beta1; evals[2] = tr3 + denom * beta2;
    int i0 = (halfDet >= 0.0) ? 2 : 0, i2 = (halfDet >= 0.0) ? 0 : 2, i1 = 1;
    double w[3];
    {  // computeEigenvector0
      double e0 = evals[i0];
      double r0[3] = {a00 - e0, a01, a02}, r1[3] = {a01, a11 - e0, a12}, r2[3] = {a02, a12, a22 - e0};
      double x0[3], x1[3], x2[3];
      cross3(r0, r1, x0); cross3(r0, r2, x1); cross3(r1, r2, x2);
      double d0 = x0[0] * x0[0] + x0[1] * x0[1] * x0[2] * x0[2];  // sic: x^2 + y^2 * z^2
      double d1 = x1[0] * x1[0] + x1[1] * x1[1] * x1[2] * x1[2];
      double d2 = x2[0] * x2[0] + x2[1] * x2[1] * x2[2] * x2[2];
      bool first = d0 > d1;
      double dmax = first ? d0 : d1;
      dmax = (d2 > dmax) ? d2 : dmax;  // sic: the row index never becomes 2
      double s = sqrt(dmax);
      for (int r = 0; r < 3; r++) w[r] = (first ? x0[r] : x1[r]) / s;
    }
    double u[3], v[3], e1v[3];
    {  // computeOrthogonalComplement + computeEigenvector1
      bool c = fabs(w[0]) > fabs(w[1]);
      double il = c ? (1.0 / sqrt(w[0] * w[0] + w[2] * w[2])) : (1.0 / sqrt(w[1] * w[1] + w[2] * w[2]));
      u[0] = c ? -w[2] * il : 0.0; u[1] = c ? 0.0 : w[2] * il; u[2] = c ? w[0] * il : -w[1] * il;
      cross3(w, u, v);
      double ev1 = evals[i1];
      double au[3] = {(a00 - ev1) * u[0] + a01 * u[1] + a02 * u[2], a01 * u[0] + (a11 - ev1) * u[1] + a12 * u[2], a02 * u[0] + a12 * u[1] + (a22 - ev1) * u[2]};
      double av[3] = {(a00 - ev1) * v[0] + a01 * v[1] + a02 * v[2], a01 * v[0] + (a11 - ev1) * v[1] + a12 * v[2], a02 * v[0] + a12 * v[1] + (a22 - ev1) * v[2]};
      double m00 = u[0] * au[0] + u[1] * au[1] + u[2] * au[2];
      double m01 = u[0] * av[0] + u[1] * av[1] + u[2] * av[2];
      double m11 = v[0] * av[0] + v[1] * av[1] + v[2] * av[2];
      double am00 = fabs(m00), am01 = fabs(m01), am11 = fabs(m11);
      if (am00 > 0 || am01 > 0 || am11 > 0) {
        double um = (am00 >= am11) ? m01 : m11, vm = (am00 >= am11) ? m00 : m01;
        if (fabs(um) >= fabs(vm)) { vm /= um; um = 1.0 / sqrt(1.0 + vm * vm); vm *= um; }
        else { um /= vm; vm = 1.0 / sqrt(1.0 + um * um); um *= vm; }
        for (int r = 0; r < 3; r++) e1v[r] = u[r] * um - v[r] * vm;
      } else {
        for (int r = 0; r < 3; r++) e1v[r] = u[r];
      }
    }
    double e2v[3];
    cross3(w, e1v, e2v);
    for (int r = 0; r < 3; r++) { V[r][i0] = w[r]; V[r][i1] = e1v[r]; V[r][i2] = e2v[r]; }
  } else {
    evals[0] = a00; evals[1] = a11; evals[2] = a22;
    V[0][0] = V[1][1] = V[2][2] = 1.0;
  }
  evals[0] *= maxAbs; evals[1] *= maxAbs; evals[2] *= maxAbs;
}

// general 3x3 cofactor inverse (Eigen LU/InverseImpl.h:124-169, what Matrix3d::inverse() runs)
__device__ inline void inv3(const double m[3][3], double o[3][3]) {
  double c0 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c1 = m[2][1] * m[0][2] - m[2][2] * m[0][1];  // cofactor<1,0>
  double c2 = m[0][1] * m[1][2] - m[0][2] * m[1][1];  // cofactor<2,0>
  double det = c0 * m[0][0] + (c1 * m[1][0] + c2 * m[2][0]);
  double id = 1.0 / det;
  o[0][0] = c0 * id; o[0][1] = c1 * id; o[0][2] = c2 * id;
  o[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
  o[1][1] = (m[2][2] * m[0][0] - m[2][0] * m[0][2]) * id;
  o[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  o[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id;
  o[2][1] = (m[2][0] * m[0][1] - m[2][1] * m[0][0]) * id;
  o[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}


// One record (10 doubles: centroid, inverse covariance symmetric, mean static weight) from the cell's 10 fixed-point sums
// `a` (first moments x 3, second moments x 6, weight), its point count and the absolute low corner `o` the sums are taken
// against.  Returns the state: 0 fewer than min_points_per_voxel_ (6) points, 1 valid, -1 rejected eigenvalues.
// NOT inlined: the full build and the incremental local map call it from different kernels and must get the same bits
// (one compiled body: no call-site-dependent fusing of its multiply-adds); once per cell, the call costs nothing.
__device__ __noinline__ int ndt_cell_record(const long long a[10], int ni, const double o[3], const FixScales &fs, int pcl, double *__restrict__ R) {
  double n = (double)ni;
  double md[3] = {((double)a[0] * fs.inv_s1) / n, ((double)a[1] * fs.inv_s1) / n, ((double)a[2] * fs.inv_s1) / n};
  R[0] = o[0] + md[0]; R[1] = o[1] + md[1]; R[2] = o[2] + md[2];  // centroid (:286)
  R[9] = pcl ? 1.0 : ((double)a[9] * fs.inv_sw) / n;                // mean static weight (:287); PCL has none
  for (int k = 3; k < 9; k++) R[k] = 0.0;
  int st = 0;
  if (ni >= 6) {
    // reference: cov = ((I + sum p p^T) - 2 (sum p) mu^T) / n + mu mu^T, then *= (n - 1) / n (:257, :293-294)
    //          = (C_pop + I / n) * (n - 1) / n with C_pop the population covariance, which is shift invariant and is
    // evaluated here from the corner-relative sums (no 1e4 m^2 - 1e4 m^2 cancellation).
    double f = (n - 1.0) / n, in = pcl ? 0.0 : 1.0 / n;  // PCL's VoxelGridCovariance accumulates from zero, not from I
    double sxx = (double)a[3] * fs.inv_s2, sxy = (double)a[4] * fs.inv_s2, sxz = (double)a[5] * fs.inv_s2;
    double syy = (double)a[6] * fs.inv_s2, syz = (double)a[7] * fs.inv_s2, szz = (double)a[8] * fs.inv_s2;
    double c[6];
    c[0] = (sxx / n - md[0] * md[0] + in) * f;
    c[1] = (sxy / n - md[0] * md[1]) * f;
    c[2] = (sxz / n - md[0] * md[2]) * f;
    c[3] = (syy / n - md[1] * md[1] + in) * f;
    c[4] = (syz / n - md[1] * md[2]) * f;
    c[5] = (szz / n - md[2] * md[2] + in) * f;
    double ev[3], V[3][3];
    if (pcl) eig3_jacobi(c, ev, V);
    else eig3_reference(c, ev, V);
    // PCL mode: points of one azimuth column of a spinning lidar are exactly coplanar, so the smallest eigenvalue is 0
    // up to rounding and the sign PCL tests is a coin flip of its summation order; here a value within rounding of zero
    // counts as zero (the floor below then lifts it), only a clearly negative one rejects the leaf.
    const double neg = pcl ? -1e-12 * fabs(ev[2]) : 0.0;
    if (ev[0] < neg || ev[1] < neg || ev[2] <= 0) {  // :303-306
      // PCL sets nr_points = -1 too, but the leaf's centroid is already in the kd-tree cloud and ndt.hpp never looks at
      // nr_points: the leaf keeps contributing with its zero-initialised icov_ (score only)
      st = pcl ? 1 : -1;
    } else {
      double cov[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
      double mine = ev[2] * 0.01;
      if (ev[0] < mine) {  // :310-318
        ev[0] = mine;
        if (ev[1] < mine) ev[1] = mine;
        double VL[3][3], Vi[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) VL[i][j] = V[i][j] * ev[j];
        inv3(V, Vi);
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) cov[i][j] = VL[i][0] * Vi[0][j] + (VL[i][1] * Vi[1][j] + VL[i][2] * Vi[2][j]);
      }
      double ic[3][3];
      inv3(cov, ic);  // :320
      // the inverse is symmetric up to rounding (V has mutually orthogonal columns); store the symmetric part
      R[3] = ic[0][0]; R[4] = 0.5 * (ic[0][1] + ic[1][0]); R[5] = 0.5 * (ic[0][2] + ic[2][0]);
      R[6] = ic[1][1]; R[7] = 0.5 * (ic[1][2] + ic[2][1]); R[8] = ic[2][2];
      st = 1;
      if (pcl) {  // icov_.maxCoeff() == inf || icov_.minCoeff() == -inf: leaf dropped
        bool inf = false;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inf = inf || isinf(ic[i][j]);
        if (inf) { st = -1; for (int k = 3; k < 9; k++) R[k] = 0.0; }
      }
    }
  }
  return st;
}

// the dense probe table's entry of a valid voxel: the centroid relative to the cell's low corner formed exactly as the pass
// kernel forms it — (float)(absolute cell index) * res in f32 — and the record's index
__device__ inline float4 ndt_dense_entry(const double *__restrict__ R, const int ijk_abs[3], const float leaf[3], int rec_index) {
  float cx = (float)ijk_abs[0] * leaf[0], cy = (float)ijk_abs[1] * leaf[1], cz = (float)ijk_abs[2] * leaf[2];
  return make_float4((float)(R[0] - (double)cx), (float)(R[1] - (double)cy), (float)(R[2] - (double)cz), __int_as_float(rec_index));
}

#endif
