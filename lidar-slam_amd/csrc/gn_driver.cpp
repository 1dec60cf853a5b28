// gn_driver.cpp — host side of the NDT registration: Newton iteration with More-Thuente line search.
//
// Follows lidar_localization::NormalDistributionsTransform::computeTransformation and
// computeStepLengthMT (reference: lidar_localization/src/models/registration/ndt_registration_manual/
// NormalDistributionsTransform.cpp:310-389, :648-756, :760-873).  The control flow is serial and
// data-dependent (<= 10 extra passes per Newton iteration), so it stays on one host thread; only the
// derivative / Hessian PASSES are parallel, delivered by the evaluator as 32 doubles already summed
// over GPUs.  All arithmetic is f64 except the rigid transform, which the reference builds in f32.
//
// Deliberately kept quirks of the reference:
//   * every step length is clamped into [trans_eps / 2, step_size] (:686-687, :712-713)
//   * `interval_converged` starts TRUE whenever step_max > step_min (:683), so the trial loop (:705) is
//     entered only if that test is false;
//   * inside the loop phi_t / d_phi_t ACCUMULATE (`-=`) instead of being assigned (:726-727);
//   * convergence: iteration count > max_iter, or (not the first iteration and step < trans_eps) (:379).
#include <cfloat>
#include <cmath>
#include <cstring>

#include "gn_math.h"
#include "lidar_slam_amd.h"

int lsa_gn_solve_defer(lsa_pass_eval_fn eval, int (*defer)(void *, const float[16], const double[6], int), void *user, float res, float step_size, float trans_eps,
                       int max_iter, float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats,
                       double *trace_rows, int trace_cap, int *trace_n, double *pairs_before_deferred);

namespace {

inline void unpack_h(const double *out32, double H[6][6]) {
  int k = 7;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) { H[i][j] = out32[k]; H[j][i] = out32[k]; k++; }
}

// x = pinv(H) b through a one-sided Jacobi SVD with Eigen's rank rule (threshold = 6 * eps * sigma_max,
// Eigen SVD/SVDBase.h:130-140,181-186): what JacobiSVD<Matrix6d>(H, FullU|FullV).solve(b) returns (:353-355).
void svd6_solve(const double H[6][6], const double b[6], double x[6]) {
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { A[i][j] = H[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 6; k++) { alpha += A[k][p] * A[k][p]; beta += A[k][q] * A[k][q]; gamma += A[k][p] * A[k][q]; }
        double lim = std::sqrt(alpha * beta);
        if (gamma == 0.0 || std::fabs(gamma) <= DBL_EPSILON * 1e-3 * lim) continue;
        off = std::fmax(off, std::fabs(gamma) / (lim > 0 ? lim : 1));
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = ((zeta >= 0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; k++) {
          double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq; A[k][q] = s * ap + c * aq;
          double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[6];
  int ord[6];
  for (int j = 0; j < 6; j++) {
    double s2 = 0;
    for (int k = 0; k < 6; k++) s2 += A[k][j] * A[k][j];
    sv[j] = std::sqrt(s2);
    ord[j] = j;
  }
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++)
      if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double thr = std::fmax(sv[ord[0]] * 6.0 * DBL_EPSILON, DBL_MIN);
  int rank = 0;
  for (int i = 0; i < 6; i++) if (sv[ord[i]] != 0.0) rank++;
  while (rank > 0 && sv[ord[rank - 1]] < thr) rank--;
  for (int i = 0; i < 6; i++) x[i] = 0.0;
  for (int r = 0; r < rank; r++) {
    int j = ord[r];
    double ub = 0;
    for (int k = 0; k < 6; k++) ub += (A[k][j] / sv[j]) * b[k];
    double coef = ub / sv[j];
    for (int k = 0; k < 6; k++) x[k] += V[k][j] * coef;
  }
}

// The Newton system is solved 4 times per scan on the critical path between two passes, and the Jacobi SVD above costs
// 2.5-5 us.  For a well-conditioned H (every scan that has planes in three directions) pinv(H) b == H^-1 b, so Gaussian
// elimination with partial pivoting (gm::lu6_solve, shared with the device-side loop of ndt_align.hip) gives the same step
// to ~cond * eps (1e-11 relative here; the pose leaves this function through an f32 matrix anyway) in ~0.2 us.  Anything
// else — a pivot ratio below 1e-10, a non-finite value — makes it return false and the SVD path above is taken, which
// alone implements Eigen's rank rule.

// libm fall-back of gm::pose_to_matrix for angles outside its range (|angle| >= 120 rad: never an Euler angle)
void axis_rotation_libm(float angle, int axis, float R[3][3]) { gm::axis_rotation(sinf(angle), cosf(angle), axis, R); }

// More-Thuente helpers (:69-75, :760-873, "Copied from ndt.hpp")
inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {  // case 1
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {  // case 2
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = std::sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double nxt = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    return (a_t > a_l) ? std::fmin(a_t + 0.66 * (a_u - a_t), nxt) : std::fmax(a_t + 0.66 * (a_u - a_t), nxt);
  }
  double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;  // case 4
  double w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

bool update_interval(double &a_l, double &f_l, double &g_l, double &a_u, double &f_u, double &g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

struct Driver {
  lsa_pass_eval_fn eval;
  // Optional (library-internal callers): enqueue a pass WITHOUT waiting for its sums; non-zero = cannot, evaluate it the ordinary
  // way.  Used for the one pass per ScanMatch whose sums the iteration never reads: with the trial loop of the line search
  // dead, the step length and with it the convergence test (:379-381) are known BEFORE the derivatives at the new pose are
  // (computeStepLengthMT evaluates them at :696-698 all the same); when that test ends the iteration, the pass only feeds
  // the statistics (score, transformation probability, pair count), which the caller completes when the sums arrive.
  int (*defer)(void *user, const float T[16], const double p[6], int kind) = nullptr;
  bool deferred = false;
  int iterations_so_far = 0;
  void *user;
  double step_size, trans_eps;
  int max_iter;
  int mode = 0;  // LSA_NDT_MODE_*
  double score = 0, g[6] = {0}, H[6][6] = {{0}};
  float T[16];
  int passes = 0, ls_iters = 0;
  double pairs_total = 0, npoints = 0;
  int rc = 0;
  double *trace = nullptr;  // optional per-pass rows {p(6), score, step length, kind} (lsa_gn_solve_trace)
  int trace_cap = 0, *trace_n = nullptr;

  void trace_row(const double p[6], double step, int kind) {
    if (!trace || !trace_n) return;
    if (*trace_n < trace_cap) {
      double *r = trace + (size_t)(*trace_n) * 9;
      memcpy(r, p, 6 * sizeof(double));
      r[6] = score; r[7] = step; r[8] = kind;
    }
    (*trace_n)++;
  }

  bool pass(const double p[6], int kind) {
    double out[32];
    rc = eval(user, T, p, kind, out);
    if (rc) return false;
    passes++;
    if (kind != 2) {
      score = out[0];
      for (int i = 0; i < 6; i++) g[i] = out[1 + i];
    }
    if (kind != 1) unpack_h(out, H);
    else memset(H, 0, sizeof(H));  // computeDerivatives zeroes the Hessian even when it does not fill it (:403)
    pairs_total += out[28];
    npoints = out[29];
    return true;
  }

  // computeStepLengthMT (:648-756); returns the step length, or NaN on evaluator failure
  double step_length(const double x[6], double dir[6], double step_init, double step_max, double step_min) {
    double phi_0 = -score, d_phi_0 = 0;
    for (int i = 0; i < 6; i++) d_phi_0 += g[i] * dir[i];
    d_phi_0 = -d_phi_0;
    if (d_phi_0 >= 0) {
      if (d_phi_0 == 0) return 0;
      d_phi_0 *= -1;
      for (int i = 0; i < 6; i++) dir[i] *= -1;
    }
    const int max_step_iterations = 10;
    int step_iterations = 0;
    const double mu = 1.e-4, nu = 0.9;
    double a_l = 0, a_u = 0;
    double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
    double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
    bool interval_converged = (step_max - step_min) > 0, open_interval = true;
    double a_t = std::fmax(std::fmin(step_init, step_max), step_min);
    double x_t[6];
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
    lsa_pose_to_matrix(x_t, T);
    // (interval_converged: the loop below will not run, a_t is this call's result whatever the pass says)
    if (defer && interval_converged && (iterations_so_far > max_iter || (iterations_so_far && std::fabs(a_t) < trans_eps)) &&
        defer(user, T, x_t, 0) == 0) {
      passes++;
      deferred = true;
      return a_t;
    }
    if (!pass(x_t, 0)) return NAN;
    trace_row(x_t, a_t, 0);
    double phi_t = -score, d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t += g[i] * dir[i];
    d_phi_t = -d_phi_t;
    double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
      a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                          : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
      a_t = (a_t < step_max) ? a_t : step_max;
      a_t = (a_t > step_min) ? a_t : step_min;
      for (int i = 0; i < 6; i++) x_t[i] = x[i] + dir[i] * a_t;
      lsa_pose_to_matrix(x_t, T);
      if (!pass(x_t, 1)) return NAN;
      trace_row(x_t, a_t, 1);
      double dd = 0;
      for (int i = 0; i < 6; i++) dd += g[i] * dir[i];
      if (mode == LSA_NDT_MODE_PCL) {  // pcl ndt.hpp assigns
        phi_t = -score;
        d_phi_t = -dd;
      } else {
        phi_t -= score;  // sic (:726)
        d_phi_t -= dd;   // sic (:727)
      }
      psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
      d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
      if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
        open_interval = false;
        f_l += phi_0 - mu * d_phi_0 * a_l; g_l += mu * d_phi_0;
        f_u += phi_0 - mu * d_phi_0 * a_u; g_u += mu * d_phi_0;
      }
      interval_converged = open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                         : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
      step_iterations++;
    }
    if (step_iterations) {
      if (!pass(x_t, 2)) return NAN;  // computeHessian (:749-751)
      trace_row(x_t, a_t, 2);
    }
    ls_iters += step_iterations;
    return a_t;
  }
};

}  // namespace

extern "C" void lsa_pose_to_matrix(const double p[6], float T[16]) {
  // (Translation3f(x) * AngleAxisf(roll, X) * AngleAxisf(pitch, Y) * AngleAxisf(yaw, Z)).matrix()  (:691-694); the same
  // arithmetic the device-side loop runs (gn_math.h: sinf / cosf restated from glibc, bit-identical to this image's libm)
  if (gm::pose_to_matrix(p, T)) return;
  float Rx[3][3], Ry[3][3], Rz[3][3], Rxy[3][3], R[3][3];
  axis_rotation_libm((float)p[3], 0, Rx);
  axis_rotation_libm((float)p[4], 1, Ry);
  axis_rotation_libm((float)p[5], 2, Rz);
  gm::mul33(Rx, Ry, Rxy);
  gm::mul33(Rxy, Rz, R);
  memset(T, 0, 16 * sizeof(float));
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r][c];
  T[12] = (float)p[0]; T[13] = (float)p[1]; T[14] = (float)p[2]; T[15] = 1.0f;
}

extern "C" void lsa_matrix_to_pose(const float T[16], double p[6]) {
  // p = (translation, rotation().eulerAngles(0, 1, 2)) (:331-337).  Transform::rotation() is an f32 Jacobi-SVD polar factor of
  // the linear part (gm::rotation_polar_f32: Eigen's own sequence of operations; it moves a pose that has been through f32
  // products by an ulp or two, which the first pass's angle tables then see); the Euler angles are taken from ITS result.
  // Eigen 3.2.92's eulerAngles folds the first angle into [0, pi] before negating, so a slightly negative roll comes back
  // in the flipped chart (roll + pi, pi - pitch, yaw +- pi); the chart map is affine and norm preserving, the Newton
  // iteration is unaffected.
  float L[9], Rm[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) L[r * 3 + c] = T[c * 4 + r];
  gm::rotation_polar_f32(L, Rm);
  auto M = [&](int r, int c) { return Rm[r * 3 + c]; };
  float r0 = atan2f(M(1, 2), M(2, 2));
  float c2 = sqrtf(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
  float r1;
  if (r0 > 0.0f) {
    r0 -= (float)M_PI;
    r1 = atan2f(-M(0, 2), -c2);
  } else {
    r1 = atan2f(-M(0, 2), c2);
  }
  float s1 = sinf(r0), c1 = cosf(r0);
  float r2 = atan2f(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
  p[0] = T[12]; p[1] = T[13]; p[2] = T[14];
  p[3] = -r0; p[4] = -r1; p[5] = -r2;
}

// tests (CPU): the f32 Eigen blocks of gn_math.h on caller-supplied inputs.  what: 0 rotation_polar_f32 (in 9 -> out 9), 1 determinant
// (in 36 -> out 1), 2 inverse and inverse * B (in 36 + 6 -> out 36 + 6), 3 Matrix3f product (in 9 + 9 -> out 9).  Pure host arithmetic.
extern "C" int lsa_eigen_f32_selftest(int what, const float *in, float *out) {
  if (!in || !out) return LSA_ERR_INVALID;
  if (what == 0) { gm::rotation_polar_f32(in, out); return LSA_OK; }
  if (what == 1) { gm::Lu6f f; gm::lu6f_factor(in, &f); out[0] = gm::lu6f_det(f); return LSA_OK; }
  if (what == 2) { gm::Lu6f f; gm::lu6f_factor(in, &f); gm::lu6f_inverse(f, out); gm::mul66f_vec(out, in + 36, out + 36); return LSA_OK; }
  if (what == 3) { gm::mul3f_eigen(in, in + 9, out); return LSA_OK; }
  return LSA_ERR_INVALID;
}

extern "C" void lsa_ndt_gauss_constants(float res, float outlier_ratio_f, double *d1, double *d2) {
  // (:315-321), eq. 6.8 of Magnusson 2009; outlier_ratio arrives as float (ndt_registration_manual.cpp:20)
  double outlier_ratio = outlier_ratio_f;
  double c1 = 10 * (1 - outlier_ratio);
  double c2 = outlier_ratio / std::pow(res, 3);
  double d3 = -std::log(c2);
  *d1 = -std::log(c1 + c2) - d3;
  *d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / *d1);
}

extern "C" int lsa_gn_solve(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                            float outlier_ratio, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats) {
  return lsa_gn_solve_mode(eval, user, res, step_size, trans_eps, max_iter, outlier_ratio, LSA_NDT_MODE_MANUAL, predict_pose, result_pose, stats);
}

extern "C" int lsa_gn_solve_mode(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                                 float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats) {
  return lsa_gn_solve_trace(eval, user, res, step_size, trans_eps, max_iter, outlier_ratio, mode, predict_pose, result_pose, stats, nullptr, 0, nullptr);
}

extern "C" int lsa_gn_solve_trace(lsa_pass_eval_fn eval, void *user, float res, float step_size, float trans_eps, int max_iter,
                                  float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats,
                                  double *trace_rows, int trace_cap, int *trace_n) {
  return lsa_gn_solve_defer(eval, nullptr, user, res, step_size, trans_eps, max_iter, outlier_ratio, mode, predict_pose, result_pose, stats, trace_rows, trace_cap,
                            trace_n, nullptr);
}

// library-internal form (ndt_api.hip): `defer` as documented in Driver; *pairs_before_deferred >= 0 on return iff the last pass was
// deferred — then stats->score / trans_probability / pairs_per_pass are NOT final: the caller completes them from the pass's 32
// sums (score = out[0], trans_probability = out[0] / out[29], pairs_per_pass = (*pairs_before_deferred + out[28]) / passes)
int lsa_gn_solve_defer(lsa_pass_eval_fn eval, int (*defer)(void *, const float[16], const double[6], int), void *user, float res, float step_size, float trans_eps,
                       int max_iter, float outlier_ratio, int mode, const float predict_pose[16], float result_pose[16], lsa_ndt_stats *stats,
                       double *trace_rows, int trace_cap, int *trace_n, double *pairs_before_deferred) {
  (void)res; (void)outlier_ratio;  // the evaluator owns d1/d2; kept in the signature to mirror the plugin ctor
  if (pairs_before_deferred) *pairs_before_deferred = -1.0;
  if (!eval || !predict_pose || !result_pose) return LSA_ERR_INVALID;
  Driver d;
  d.eval = eval; d.user = user; d.mode = mode;
  d.defer = (trace_rows || !pairs_before_deferred) ? nullptr : defer;   // (a trace wants every pass's score as it happens)
  d.trace = trace_rows; d.trace_cap = trace_cap; d.trace_n = trace_n;
  if (trace_n) *trace_n = 0;
  d.step_size = step_size; d.trans_eps = trans_eps; d.max_iter = max_iter;  // float -> double like the plugin (:14-22)
  memcpy(d.T, predict_pose, sizeof(d.T));  // final_transformation_ = guess (Registration.cpp:80, Normal...:323-328)
  double p[6], delta[6];
  lsa_matrix_to_pose(d.T, p);
  int iterations = 0;
  bool converged = false;
  double trans_probability = 0;
  if (!d.pass(p, 0)) return d.rc;  // :346
  d.trace_row(p, 0.0, 0);
  while (!converged) {
    double negg[6];
    for (int i = 0; i < 6; i++) negg[i] = -d.g[i];
    if (!gm::lu6_solve(d.H, negg, delta)) svd6_solve(d.H, negg, delta);
    double nrm2 = 0;
    for (int i = 0; i < 6; i++) nrm2 += delta[i] * delta[i];
    double nrm = std::sqrt(nrm2);
    if (nrm == 0 || nrm != nrm) {  // :359-363
      trans_probability = d.npoints > 0 ? d.score / d.npoints : 0;
      converged = (nrm == nrm);
      goto done;
    }
    for (int i = 0; i < 6; i++) delta[i] /= nrm;
    d.iterations_so_far = iterations;
    nrm = d.step_length(p, delta, nrm, d.step_size, d.trans_eps / 2);
    if (nrm != nrm) return d.rc ? d.rc : LSA_ERR_HIP;
    for (int i = 0; i < 6; i++) { delta[i] *= nrm; p[i] = p[i] + delta[i]; }
    if (iterations > d.max_iter || (iterations && (std::fabs(nrm) < d.trans_eps))) converged = true;  // :379-381
    iterations++;
  }
  if (d.npoints > 0) trans_probability = d.score / d.npoints;
done:
  memcpy(result_pose, d.T, sizeof(d.T));
  if (stats) {
    stats->iterations = iterations;
    stats->ls_iterations = d.ls_iters;
    stats->passes = d.passes;
    stats->converged = converged ? 1 : 0;
    stats->score = d.score;
    stats->trans_probability = trans_probability;
    stats->pairs_per_pass = d.passes ? d.pairs_total / d.passes : 0;
    memcpy(stats->final_p, p, sizeof(p));
    stats->gpu_ms = 0;
  }
  if (d.deferred && pairs_before_deferred) *pairs_before_deferred = d.pairs_total;
  return LSA_OK;
}
