// gn_math.h — the scalar arithmetic of the Newton driver, written ONCE for host and device.
//
// The Newton iteration of NormalDistributionsTransform::computeTransformation (reference:
// lidar_localization/src/models/registration/ndt_registration_manual/NormalDistributionsTransform.cpp:346-384) runs
// either on the host between derivative-pass launches (gn_driver.cpp) or inside ONE persistent launch
// (ndt_align.hip).  Both must produce the same bits, so everything between "32 sums in" and "next rigid transform +
// angle-derivative vectors out" lives here as plain IEEE arithmetic with FMA contraction switched off: +, -, *, /, sqrt and
// float <-> double conversions are correctly rounded on x86-64 and on gfx950 alike (checked on the GPU by
// tests/test_gpu_align_persistent.py::test_gn_math_device_equals_host).  No libm call is made on either side:
//   * sinf / cosf of the f32 Euler angles (Eigen's AngleAxisf in :691-694 calls std::sin / std::cos on floats) restate the
//     algorithm glibc >= 2.28 uses (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h: double-precision polynomials
//     of Arm's optimized-routines, |x| < 120), so the product builds the same f32 matrix as the CPU path it replaces;
//     checked bit for bit against this image's libm (glibc 2.35) on 6e6 evaluations per run of tests/cpp/test_sanitize.cpp
//     (1.2e8 once, while writing it: no difference).
//   * sin / cos of the f64 angles (computeAngleDerivatives, :523-645) use fdlibm's kernels behind a Cody-Waite reduction:
//     within 1 ulp of libm, and these values only enter f64 sums.
#ifndef LSA_GN_MATH_H_
#define LSA_GN_MATH_H_

#include <cfloat>
#include <cmath>
#include <cstdint>

// always_inline: on the device a real call would pass the small arrays through scratch memory
#if defined(__HIPCC__) || defined(__HIP__)
#define GM_HD __host__ __device__ __attribute__((always_inline))
#else
#define GM_HD __attribute__((always_inline))
#endif

namespace gm {

// ------------------------------------------------------------------------------------------------ f32 sin / cos
struct SinCosF {
  double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};

GM_HD inline float poly_sc(double x, double x2, int n, bool neg) {
#pragma clang fp contract(off)
  // the two coefficient sets of __sincosf_table: [1] is [0] with the cosine coefficients negated
  const double sg = neg ? -1.0 : 1.0;
  const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10,
               c4 = sg * 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double t1 = s2 + x2 * s3;
    double x7 = x3 * x2;
    double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  }
  double x4 = x2 * x2;
  double t2 = c3 + x2 * c4;
  double t1 = c0 + x2 * c1;
  double x6 = x4 * x2;
  double c = t1 + x4 * c2;
  return (float)(c + x6 * t2);
}

GM_HD inline uint32_t abstop12(float x) { return (__builtin_bit_cast(uint32_t, x) >> 20) & 0x7ffu; }

// sin (cosine == false) or cos of a float; *ok = false for |x| >= 120 or non-finite (the caller then leaves the fast path)
GM_HD inline float sincosf_one(float y, bool cosine, bool *ok) {
#pragma clang fp contract(off)
  double x = (double)y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {  // |y| < pi / 4
    double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return cosine ? 1.0f : y;
    return poly_sc(x, s, cosine ? 1 : 0, false);
  }
  if (abstop12(y) < abstop12(120.0f)) {
    // reduce_fast: the quadrant from a scaled float -> int conversion (hpi_inv = 2 / pi * 2^24)
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // {1, -1, -1, 1}[n & 3]
    return poly_sc(x * sign, x * x, cosine ? (n ^ 1) : n, (n & 2) != 0);
  }
  *ok = false;
  return 0.0f;
}

// ------------------------------------------------------------------------------------------------ f64 sin / cos
GM_HD inline double ksin(double x, double y) {
#pragma clang fp contract(off)
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x, v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
GM_HD inline double kcos(double x, double y) {
#pragma clang fp contract(off)
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}
// sin and cos of a double with |x| < 1e5 (Euler angles stay within a few multiples of pi); *ok = false otherwise
GM_HD inline void sincos_d(double x, double *s, double *c, bool *ok) {
#pragma clang fp contract(off)
  double ax = x < 0 ? -x : x;
  if (!(ax < 1.0e5)) { *ok = false; *s = 0.0; *c = 1.0; return; }
  double y0 = x, y1 = 0.0;
  int n = 0;
  if (ax > 0.78539816339744830962) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    double fn = __builtin_rint(x * invpio2);
    n = (int)fn;
    double r = x - fn * pio2_1;  // exact: pio2_1 has 33 significant bits, |fn| < 2^17
    double w = fn * pio2_1t;
    double t = r;
    w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    y0 = r - w;
    y1 = (r - y0) - w;
  }
  double sn = ksin(y0, y1), cs = kcos(y0, y1);
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

// ------------------------------------------------------------------------------------------------ 6x6 solve
// Gaussian elimination with partial pivoting; false when the system is not comfortably regular (pivot ratio below 1e-10,
// a non-finite value): the caller then takes the Jacobi-SVD path (host only), which alone implements Eigen's rank rule.
GM_HD inline bool lu6_solve(const double H[6][6], const double b[6], double x[6]) {
#pragma clang fp contract(off)
  // every loop is fully unrolled and every index static (the row swap is a chain of guarded swaps, not A[r][j]): on the
  // device the 42 values then live in registers instead of scratch memory
  double A[6][7];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = H[i][j];
    A[i][6] = b[i];
  }
  double pmax = 0.0, pmin = DBL_MAX, invs[6];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    int r = c;
    double best = A[c][c] < 0 ? -A[c][c] : A[c][c];
#pragma unroll
    for (int i = c + 1; i < 6; i++) {
      double v = A[i][c] < 0 ? -A[i][c] : A[i][c];
      if (v > best) { best = v; r = i; }
    }
    if (!(best > 0.0) || !(best <= DBL_MAX)) ok = false;
#pragma unroll
    for (int i = c + 1; i < 6; i++)
      if (r == i) {
#pragma unroll
        for (int j = c; j < 7; j++) { double t = A[c][j]; A[c][j] = A[i][j]; A[i][j] = t; }
      }
    pmax = best > pmax ? best : pmax;
    pmin = best < pmin ? best : pmin;
    const double inv = 1.0 / A[c][c];
    invs[c] = inv;
#pragma unroll
    for (int i = c + 1; i < 6; i++) {
      const double f = A[i][c] * inv;
      if (f != 0.0) {
#pragma unroll
        for (int j = c + 1; j < 7; j++) A[i][j] = A[i][j] - f * A[c][j];
      }
    }
  }
  if (!ok || !(pmin > 1e-10 * pmax)) return false;
  // back substitution: the known unknowns are subtracted in DESCENDING order and the pivot reciprocals of the
  // elimination are reused (x_i = v_i * (1 / a_ii)) — the order in which the wave-parallel solver of ndt_align.hip learns
  // them (lane i holds row i), so that both give the same bits
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = A[i][6];
#pragma unroll
    for (int j = 5; j > i; j--) v = v - A[i][j] * x[j];
    x[i] = v * invs[i];
    if (!(x[i] >= -DBL_MAX && x[i] <= DBL_MAX)) ok = false;
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------ pose -> matrix
// (Translation3f(x) * AngleAxisf(roll, X) * AngleAxisf(pitch, Y) * AngleAxisf(yaw, Z)).matrix()  (:691-694), column-major.
// Eigen AngleAxis<float>::toRotationMatrix for a coordinate axis (Geometry/AngleAxis.h): the diagonal entry on the axis
// is (1 - c) * 1 + c evaluated in f32, not the literal 1.
GM_HD inline void axis_rotation(float s, float c, int axis, float R[3][3]) {
#pragma clang fp contract(off)
  float d = (1.0f - c) * 1.0f + c;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = 0.0f;
  int a = axis, b = (axis + 1) % 3, e = (axis + 2) % 3;
  R[a][a] = d; R[b][b] = c; R[e][e] = c; R[b][e] = -s; R[e][b] = s;
}
GM_HD inline void mul33(const float A[3][3], const float B[3][3], float C[3][3]) {
#pragma clang fp contract(off)
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}
// s[k], c[k]: sinf / cosf of (float)p[3 + k]
GM_HD inline void pose_to_matrix_sc(const double p[6], const float s[3], const float c[3], float T[16]) {
  float Rx[3][3], Ry[3][3], Rz[3][3], Rxy[3][3], R[3][3];
  axis_rotation(s[0], c[0], 0, Rx);
  axis_rotation(s[1], c[1], 1, Ry);
  axis_rotation(s[2], c[2], 2, Rz);
  mul33(Rx, Ry, Rxy);
  mul33(Rxy, Rz, R);
  for (int i = 0; i < 16; i++) T[i] = 0.0f;
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) T[cc * 4 + r] = R[r][cc];
  T[12] = (float)p[0]; T[13] = (float)p[1]; T[14] = (float)p[2]; T[15] = 1.0f;
}
GM_HD inline bool pose_to_matrix(const double p[6], float T[16]) {
  bool ok = true;
  float s[3], c[3];
  for (int k = 0; k < 3; k++) {
    float a = (float)p[3 + k];
    s[k] = sincosf_one(a, false, &ok);
    c[k] = sincosf_one(a, true, &ok);
  }
  pose_to_matrix_sc(p, s, c, T);
  return ok;
}

// ------------------------------------------------------------------------------------------------ matrix -> pose, f32 Eigen blocks
// The f32 Eigen calls the reference makes around the hot loops, operation by operation in Eigen 3.2.92's order (the
// reference vendors it: lidar_localization/third_party/eigen3); oracle/eigen_f32_oracle.c holds the checker's copy of the
// same restatement, and that one is pinned bit for bit against the vendored Eigen compiled in place (oracle/ref_blocks.cpp,
// tests/golden/eigen_blocks.json); tests/test_cabi_cpu.py holds THESE to the same fixtures through lsa_matrix_to_pose /
// lsa_eigen_f32_selftest.
//
// Eigen reduces a fixed-size 3-term sum as t0 + (t1 + t2) (Core/Redux.h, unrolled by halves) — not left to right.
GM_HD inline float dot3_eigen(float a0, float b0, float a1, float b1, float a2, float b2) {
#pragma clang fp contract(off)
  return a0 * b0 + (a1 * b1 + a2 * b2);
}
GM_HD inline void rot_pair(float *x, float *y, float c, float s) {   // Jacobi/Jacobi.h:418-426
#pragma clang fp contract(off)
  const float xi = *x, yi = *y;
  *x = c * xi + s * yi;
  *y = -s * xi + c * yi;
}
// Transform<float,3,Affine>::rotation() (NormalDistributionsTransform.cpp:336; Eigen Geometry/Transform.h:1059-1074): two-sided
// JacobiSVD<Matrix3f> of the linear part (SVD/JacobiSVD.h:676-780), x = det(U V^T), R = (U with column 0 / x) V^T.  Row-major.
GM_HD inline void rotation_polar_f32(const float L[9], float R[9]) {
#pragma clang fp contract(off)
  float W[3][3], U[3][3], V[3][3];
  float scale = 0.0f;
  for (int k = 0; k < 9; k++) { const float a = __builtin_fabsf(L[k]); if (a > scale) scale = a; }
  if (scale == 0.0f) scale = 1.0f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) { W[r][c] = L[r * 3 + c] / scale; U[r][c] = V[r][c] = (r == c) ? 1.0f : 0.0f; }
  const float precision = 2.0f * FLT_EPSILON, consider_as_zero = 2.0f * 0x1p-149f;
  bool finished = false;
  for (int sweep = 0; !finished && sweep < 64; sweep++) {   // (Eigen loops until a sweep changes nothing: 2-5 sweeps; NaN input ends it at once)
    finished = true;
    for (int p = 1; p < 3; p++)
      for (int q = 0; q < p; q++) {
        const float ap = __builtin_fabsf(W[p][p]), aq = __builtin_fabsf(W[q][q]);
        float thr = precision * (ap < aq ? aq : ap);
        thr = consider_as_zero < thr ? thr : consider_as_zero;
        if (!(__builtin_fabsf(W[p][q]) > thr || __builtin_fabsf(W[q][p]) > thr)) continue;
        finished = false;
        // real_2x2_jacobi_svd (:408-435)
        float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
        float c1 = 1.0f, s1 = 0.0f;
        const float t = m00 + m11, d = m10 - m01;
        if (d != 0.0f) {
          const float u = t / d;
          const float tmp = __builtin_sqrtf(1.0f + u * u);
          s1 = 1.0f / tmp;
          c1 = u / tmp;
        }
        if (!(c1 == 1.0f && s1 == 0.0f)) { rot_pair(&m00, &m10, c1, s1); rot_pair(&m01, &m11, c1, s1); }
        // JacobiRotation::makeJacobi(m00, m01, m11) (Jacobi/Jacobi.h:80-113)
        float cr = 1.0f, sr = 0.0f;
        if (m01 != 0.0f) {
          const float tau = (m00 - m11) / (2.0f * __builtin_fabsf(m01));
          const float w = __builtin_sqrtf(tau * tau + 1.0f);
          const float tt = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
          const float sign_t = tt > 0.0f ? 1.0f : -1.0f;
          const float n = 1.0f / __builtin_sqrtf(tt * tt + 1.0f);
          sr = -sign_t * (m01 / __builtin_fabsf(m01)) * __builtin_fabsf(tt) * n;
          cr = n;
        }
        const float s2 = -sr;                                             // j_right.transpose()
        const float cl = c1 * cr - s1 * s2, sl = c1 * s2 + s1 * cr;       // j_left = rot1 * j_right.transpose()
        if (!(cl == 1.0f && sl == 0.0f)) {
          for (int i = 0; i < 3; i++) rot_pair(&W[p][i], &W[q][i], cl, sl);
          for (int i = 0; i < 3; i++) rot_pair(&U[i][p], &U[i][q], cl, sl);
        }
        if (!(cr == 1.0f && s2 == 0.0f)) {
          for (int i = 0; i < 3; i++) rot_pair(&W[i][p], &W[i][q], cr, s2);
          for (int i = 0; i < 3; i++) rot_pair(&V[i][p], &V[i][q], cr, s2);
        }
      }
  }
  float sv[3];
  for (int i = 0; i < 3; i++) {
    const float a = __builtin_fabsf(W[i][i]);
    sv[i] = a;
    if (a != 0.0f) { const float f = W[i][i] / a; for (int r = 0; r < 3; r++) U[r][i] *= f; }
  }
  for (int i = 0; i < 3; i++) sv[i] *= scale;
  for (int i = 0; i < 3; i++) {   // descending order, the first maximum wins
    int pos = i;
    for (int k = i + 1; k < 3; k++) if (sv[k] > sv[pos]) pos = k;
    if (sv[pos] == 0.0f) break;
    if (pos != i) {
      const float ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
      for (int r = 0; r < 3; r++) { const float tu = U[r][i]; U[r][i] = U[r][pos]; U[r][pos] = tu; const float tv = V[r][i]; V[r][i] = V[r][pos]; V[r][pos] = tv; }
    }
  }
  float M[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i][j] = dot3_eigen(U[i][0], V[j][0], U[i][1], V[j][1], U[i][2], V[j][2]);
  // LU/Determinant.h: bruteforce_det3_helper(0,1,2) - helper(1,0,2) + helper(2,0,1)
  const float x = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                  M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  for (int r = 0; r < 3; r++) U[r][0] /= x;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = dot3_eigen(U[i][0], V[j][0], U[i][1], V[j][1], U[i][2], V[j][2]);
}

// Matrix<float,6,6>::determinant() / inverse() / inverse() * B as ICPRegistrationManual calls them
// (lidar_localization/src/models/registration/icp_registration_manual.cpp:165-169): PartialPivLU (LU/PartialPivLU.h:331-385, unblocked
// for sizes <= 16), determinant = sign * diagonal().prod() reduced by halves, inverse = solve(Identity) by the column-oriented
// triangular solves of Core/products/TriangularSolverMatrix.h:95-118 (b = (x_i *= 1 / u_ii); r -= b * l).  Row-major.
struct Lu6f {
  float lu[6][6];
  int trans[6];
  int nt;
};
GM_HD inline void lu6f_factor(const float A[36], Lu6f *f) {
#pragma clang fp contract(off)
  f->nt = 0;
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) f->lu[r][c] = A[r * 6 + c];
  for (int k = 0; k < 6; k++) {
    int piv = k;
    float big = __builtin_fabsf(f->lu[k][k]);
    for (int r = k + 1; r < 6; r++) if (__builtin_fabsf(f->lu[r][k]) > big) { big = __builtin_fabsf(f->lu[r][k]); piv = r; }
    f->trans[k] = piv;
    if (big != 0.0f) {
      if (piv != k) { for (int c = 0; c < 6; c++) { const float t = f->lu[k][c]; f->lu[k][c] = f->lu[piv][c]; f->lu[piv][c] = t; } f->nt++; }
      for (int r = k + 1; r < 6; r++) f->lu[r][k] /= f->lu[k][k];
    }
    for (int r = k + 1; r < 6; r++)
      for (int c = k + 1; c < 6; c++) f->lu[r][c] -= f->lu[r][k] * f->lu[k][c];
  }
}
GM_HD inline float lu6f_det(const Lu6f &f) {
#pragma clang fp contract(off)
  const float p = (f.lu[0][0] * (f.lu[1][1] * f.lu[2][2])) * (f.lu[3][3] * (f.lu[4][4] * f.lu[5][5]));
  return ((f.nt & 1) ? -1.0f : 1.0f) * p;
}
GM_HD inline void lu6f_inverse(const Lu6f &f, float Hi[36]) {
#pragma clang fp contract(off)
  float X[6][6];
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) X[r][c] = (r == c) ? 1.0f : 0.0f;
  for (int k = 0; k < 6; k++)
    if (f.trans[k] != k) for (int c = 0; c < 6; c++) { const float t = X[k][c]; X[k][c] = X[f.trans[k]][c]; X[f.trans[k]][c] = t; }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      const float b = X[i][j];
      for (int s = i + 1; s < 6; s++) X[s][j] -= b * f.lu[s][i];
    }
  for (int i = 5; i >= 0; i--) {
    const float a = 1.0f / f.lu[i][i];
    for (int j = 0; j < 6; j++) {
      const float b = (X[i][j] *= a);
      for (int s = 0; s < i; s++) X[s][j] -= b * f.lu[s][i];
    }
  }
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hi[r * 6 + c] = X[r][c];
}
GM_HD inline void mul66f_vec(const float Hi[36], const float B[6], float out[6]) {
#pragma clang fp contract(off)
  for (int r = 0; r < 6; r++) {
    const float *h = Hi + r * 6;
    out[r] = (h[0] * B[0] + (h[1] * B[1] + h[2] * B[2])) + (h[3] * B[3] + (h[4] * B[4] + h[5] * B[5]));
  }
}
// delta_x = Hessian.inverse() * B unless Hessian.determinant() == 0 (:165-169); returns the determinant
GM_HD inline float solve6f_eigen(const float H[36], const float B[6], float d[6]) {
  Lu6f f;
  lu6f_factor(H, &f);
  const float det = lu6f_det(f);
  if (det == 0.0f) return det;
  float Hi[36];
  lu6f_inverse(f, Hi);
  mul66f_vec(Hi, B, d);
  return det;
}
// Matrix3f * Matrix3f (rotation_matrix_ *= delta_rotation.matrix(), :173), row-major; out may alias A
GM_HD inline void mul3f_eigen(const float A[9], const float B[9], float out[9]) {
  float r[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[i * 3 + j] = dot3_eigen(A[i * 3], B[j], A[i * 3 + 1], B[3 + j], A[i * 3 + 2], B[6 + j]);
  for (int k = 0; k < 9; k++) out[k] = r[k];
}

// ------------------------------------------------------------------------------------------------ angle derivatives
// computeAngleDerivatives (:523-645), including the |angle| < 10e-5 snap; ja = j_ang_a..h (:551-581), ha = h_ang_a2, a3,
// b2, b3, c2, c3, d1, d2, d3, e1, e2, e3, f1, f2, f3 (:584-642)
// sin / cos of one Euler angle with the reference's small-angle snap (:527-549)
GM_HD inline void snapped_sincos(double a, double *s, double *c, bool *ok) {
  if ((a < 0 ? -a : a) < 10e-5) { *c = 1.0; *s = 0.0; } else sincos_d(a, s, c, ok);
}
GM_HD inline void angle_tables(double sx, double cx, double sy, double cy, double sz, double cz, double ja[8][3], double ha[15][3]);
GM_HD inline bool angle_derivatives(const double p[6], double ja[8][3], double ha[15][3]) {
  bool ok = true;
  double cx, cy, cz, sx, sy, sz;
  snapped_sincos(p[3], &sx, &cx, &ok);
  snapped_sincos(p[4], &sy, &cy, &ok);
  snapped_sincos(p[5], &sz, &cz, &ok);
  angle_tables(sx, cx, sy, cy, sz, cz, ja, ha);
  return ok;
}
GM_HD inline void angle_tables(double sx, double cx, double sy, double cy, double sz, double cz, double ja[8][3], double ha[15][3]) {
#pragma clang fp contract(off)
  double (*j)[3] = ja;
  j[0][0] = -sx * sz + cx * sy * cz; j[0][1] = -sx * cz - cx * sy * sz; j[0][2] = -cx * cy;
  j[1][0] = cx * sz + sx * sy * cz;  j[1][1] = cx * cz - sx * sy * sz;  j[1][2] = -sx * cy;
  j[2][0] = -sy * cz;                j[2][1] = sy * sz;                 j[2][2] = cy;
  j[3][0] = sx * cy * cz;            j[3][1] = -sx * cy * sz;           j[3][2] = sx * sy;
  j[4][0] = -cx * cy * cz;           j[4][1] = cx * cy * sz;            j[4][2] = -cx * sy;
  j[5][0] = -cy * sz;                j[5][1] = -cy * cz;                j[5][2] = 0;
  j[6][0] = cx * cz - sx * sy * sz;  j[6][1] = -cx * sz - sx * sy * cz; j[6][2] = 0;
  j[7][0] = sx * cz + cx * sy * sz;  j[7][1] = cx * sy * cz - sx * sz;  j[7][2] = 0;
  double (*q)[3] = ha;
  q[0][0] = -cx * sz - sx * sy * cz;  q[0][1] = -cx * cz + sx * sy * sz;  q[0][2] = sx * cy;
  q[1][0] = -sx * sz + cx * sy * cz;  q[1][1] = -cx * sy * sz - sx * cz;  q[1][2] = -cx * cy;
  q[2][0] = cx * cy * cz;             q[2][1] = -cx * cy * sz;            q[2][2] = cx * sy;
  q[3][0] = sx * cy * cz;             q[3][1] = -sx * cy * sz;            q[3][2] = sx * sy;
  q[4][0] = -sx * cz - cx * sy * sz;  q[4][1] = sx * sz - cx * sy * cz;   q[4][2] = 0;
  q[5][0] = cx * cz - sx * sy * sz;   q[5][1] = -sx * sy * cz - cx * sz;  q[5][2] = 0;
  q[6][0] = -cy * cz;                 q[6][1] = cy * sz;                  q[6][2] = sy;
  q[7][0] = -sx * sy * cz;            q[7][1] = sx * sy * sz;             q[7][2] = sx * cy;
  q[8][0] = cx * sy * cz;             q[8][1] = -cx * sy * sz;            q[8][2] = -cx * cy;
  q[9][0] = sy * sz;                  q[9][1] = sy * cz;                  q[9][2] = 0;
  q[10][0] = -sx * cy * sz;           q[10][1] = -sx * cy * cz;           q[10][2] = 0;
  q[11][0] = cx * cy * sz;            q[11][1] = cx * cy * cz;            q[11][2] = 0;
  q[12][0] = -cy * cz;                q[12][1] = cy * sz;                 q[12][2] = 0;
  q[13][0] = -cx * sz - sx * sy * cz; q[13][1] = -cx * cz + sx * sy * sz; q[13][2] = 0;
  q[14][0] = -sx * sz + cx * sy * cz; q[14][1] = -cx * sy * sz - sx * cz; q[14][2] = 0;
}

// ------------------------------------------------------------------------------------------------ one Newton step
// The part of computeTransformation's loop body (:346-384) + computeStepLengthMT (:648-756) that runs between two
// derivative passes WHEN THE TRIAL LOOP IS DEAD: `interval_converged` starts true for step_max > step_min (:683), so the
// step length is the Newton step's norm clamped into [step_min, step_max] and exactly one full pass follows at x + a_t d.
enum StepAction {
  STEP_PASS = 0,       // evaluate the next pass at x_t (T and p_next filled in)
  STEP_NO_PASS = 1,    // d_phi_0 == 0: step length 0, no pass (:664), p unchanged
  STEP_DONE = 2,       // |delta| == 0 or NaN (:359-363): leave the loop now
  STEP_NEEDS_HOST = 3  // singular / ill-conditioned system (SVD rank rule) or an angle outside the fast trig range
};

struct StepOut {
  double x_t[6];   // pose of the next pass (== p + delta for the bookkeeping after it)
  double a_t;      // step length
  float T[16];     // rigid transform of the next pass
  bool nan_norm;   // STEP_DONE: the norm was NaN (converged_ = false) rather than 0
};

GM_HD inline int newton_step(const double p[6], const double g[6], const double H[6][6], double step_max, double step_min, StepOut *o) {
#pragma clang fp contract(off)
  double negg[6], delta[6];
  for (int i = 0; i < 6; i++) negg[i] = -g[i];
  if (!lu6_solve(H, negg, delta)) return STEP_NEEDS_HOST;
  double nrm2 = 0;
  for (int i = 0; i < 6; i++) nrm2 += delta[i] * delta[i];
  double nrm = __builtin_sqrt(nrm2);
  o->nan_norm = nrm != nrm;
  if (nrm == 0 || nrm != nrm) return STEP_DONE;
  for (int i = 0; i < 6; i++) delta[i] /= nrm;
  // computeStepLengthMT up to the first evaluation
  double d_phi_0 = 0;
  for (int i = 0; i < 6; i++) d_phi_0 += g[i] * delta[i];
  d_phi_0 = -d_phi_0;
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) {
      o->a_t = 0;
      for (int i = 0; i < 6; i++) o->x_t[i] = p[i];
      return STEP_NO_PASS;
    }
    for (int i = 0; i < 6; i++) delta[i] *= -1;
  }
  double a_t = nrm < step_max ? nrm : step_max;  // std::min(a_t, step_max)
  a_t = a_t > step_min ? a_t : step_min;         // std::max(a_t, step_min)
  o->a_t = a_t;
  for (int i = 0; i < 6; i++) o->x_t[i] = p[i] + delta[i] * a_t;
  if (!pose_to_matrix(o->x_t, o->T)) return STEP_NEEDS_HOST;
  return STEP_PASS;
}

}  // namespace gm

#endif
