// ref_blocks.cpp — harness around REAL reference code (TEST INFRASTRUCTURE ONLY).
//
// Built by oracle/Makefile into oracle/_ref/ref_blocks ONLY when /root/reference is present.
// It compiles, where they lie and unmodified:
//   * the reference header NI/SymmetricEigenSolver.h   (header-only, needs only Eigen)
//   * the reference's vendored Eigen 3.2.92             (lidar_localization/third_party/eigen3)
// No stand-in headers are used: every other reference source on this path needs PCL / boost /
// ROS headers the image lacks and is therefore treated as unbuildable ("Oracle").
//
// Protocol: one request per stdin line, one reply line of %.17g numbers on stdout.
//   eig3  a00 a01 a02 a11 a12 a22          -> evals(3) evecs row-major(9)
//   inv3  m(9 row-major)                   -> inverse(9 row-major)
//   svd6  H(36 row-major) b(6)             -> x(6) = JacobiSVD(H, FullU|FullV).solve(b)
//   p2m   p(6)                             -> T(16 col-major) = Translation*AngleAxis(X)*AngleAxis(Y)*AngleAxis(Z) in f32
//   m2p   T(16 col-major)                  -> p(6): translation, rotation().eulerAngles(0,1,2) in f32
//   mv3   m(9 row-major) v(3)              -> Matrix3d * Vector3d (3)
//   dot3  a(3) b(3)                        -> a.dot(b) (1)
//   vlv   evecs(9 row-major) evals(3)      -> evecs * evals.asDiagonal() (as Matrix3d) * evecs.inverse()   (NS/VoxelGrid.cpp:300,317)
//   covx  tmp_cov(9 row-major) pt_sum(3) n -> the covariance expression of NS/VoxelGrid.cpp:286,293-294
// f32 blocks (round 4): what NormalDistributionsTransform.cpp:331-337 and icp_registration_manual.cpp:156-176 ask of Eigen
//   rot3f  L(9 row-major)                  -> Transform<float,3,Affine>::rotation() of a transform whose linear part is L (9 row-major):
//                                             the f32 JacobiSVD polar step in front of eulerAngles (NS/...:336)
//   det6f  H(36 row-major)                 -> Matrix<float,6,6>::determinant() (1)                       (icp_registration_manual.cpp:165)
//   inv6f  H(36 row-major) B(6)            -> Hessian.inverse() (36 row-major), then Hessian.inverse() * B (6)   (:169)
//   mul3f  A(9 row-major) B(9 row-major)   -> A *= B (9 row-major), Matrix3f                             (:173)
//   icpacc R(9 row-major) p(3) e(3)        -> J = [I | -R * hat(p)] (the hat matrix written out as data, Sophus is absent),
//                                             then Hessian += J^T J (36 row-major) and B += -J^T e (6) from zero     (:156-162)
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <eigen3/Eigen/Dense>
#include <eigen3/Eigen/Geometry>
#include "lidar_localization/models/registration/ndt_registration_manual/SymmetricEigenSolver.h"

static void emit(const std::vector<double> &v) {
  for (size_t i = 0; i < v.size(); i++) std::printf(i ? " %.17g" : "%.17g", v[i]);
  std::printf("\n");
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string op;
    is >> op;
    std::vector<double> in;
    double v;
    while (is >> v) in.push_back(v);
    std::vector<double> out;
    if (op == "eig3" && in.size() == 6) {
      Eigen::Matrix3d m;
      m << in[0], in[1], in[2], in[1], in[3], in[4], in[2], in[4], in[5];
      lidar_localization::SymmetricEigensolver3x3 sv(m);
      sv.compute();
      Eigen::Vector3d ev = sv.eigenvalues();
      Eigen::Matrix3d ec = sv.eigenvectors();
      for (int i = 0; i < 3; i++) out.push_back(ev(i));
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(ec(r, c));
    } else if (op == "inv3" && in.size() == 9) {
      Eigen::Matrix3d m;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = in[r * 3 + c];
      Eigen::Matrix3d mi = m.inverse();
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(mi(r, c));
    } else if (op == "svd6" && in.size() == 42) {
      Eigen::Matrix<double, 6, 6> H;
      Eigen::Matrix<double, 6, 1> b;
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) H(r, c) = in[r * 6 + c];
      for (int r = 0; r < 6; r++) b(r) = in[36 + r];
      Eigen::JacobiSVD<Eigen::Matrix<double, 6, 6> > sv(H, Eigen::ComputeFullU | Eigen::ComputeFullV);
      Eigen::Matrix<double, 6, 1> x = sv.solve(b);
      for (int r = 0; r < 6; r++) out.push_back(x(r));
    } else if (op == "p2m" && in.size() == 6) {
      Eigen::Matrix4f T = (Eigen::Translation<float, 3>(static_cast<float>(in[0]), static_cast<float>(in[1]), static_cast<float>(in[2])) *
                           Eigen::AngleAxis<float>(static_cast<float>(in[3]), Eigen::Vector3f::UnitX()) *
                           Eigen::AngleAxis<float>(static_cast<float>(in[4]), Eigen::Vector3f::UnitY()) *
                           Eigen::AngleAxis<float>(static_cast<float>(in[5]), Eigen::Vector3f::UnitZ())).matrix();
      for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) out.push_back(T(r, c));
    } else if (op == "m2p" && in.size() == 16) {
      Eigen::Matrix4f T;
      for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) T(r, c) = static_cast<float>(in[c * 4 + r]);
      Eigen::Transform<float, 3, Eigen::Affine, Eigen::ColMajor> tr;
      tr.matrix() = T;
      Eigen::Vector3f t = tr.translation();
      Eigen::Vector3f e = tr.rotation().eulerAngles(0, 1, 2);
      Eigen::Matrix<double, 6, 1> p;
      p << t(0), t(1), t(2), e(0), e(1), e(2);
      for (int r = 0; r < 6; r++) out.push_back(p(r));
    } else if (op == "mv3" && in.size() == 12) {
      Eigen::Matrix3d m;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = in[r * 3 + c];
      Eigen::Vector3d x(in[9], in[10], in[11]);
      Eigen::Vector3d y = m * x;
      for (int r = 0; r < 3; r++) out.push_back(y(r));
    } else if (op == "dot3" && in.size() == 6) {
      Eigen::Vector3d a(in[0], in[1], in[2]), b(in[3], in[4], in[5]);
      out.push_back(a.dot(b));
    } else if (op == "vlv" && in.size() == 12) {
      Eigen::Matrix3d evecs;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) evecs(r, c) = in[r * 3 + c];
      Eigen::Vector3d ev(in[9], in[10], in[11]);
      Eigen::Matrix3d evals = ev.asDiagonal();
      Eigen::Matrix3d covariance = evecs * evals * evecs.inverse();
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(covariance(r, c));
    } else if (op == "covx" && in.size() == 13) {
      Eigen::Matrix3d tmp_cov;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tmp_cov(r, c) = in[r * 3 + c];
      Eigen::Vector3d pt_sum(in[9], in[10], in[11]);
      double point_num = in[12];
      Eigen::Vector3d centroid = pt_sum / point_num;
      Eigen::Matrix3d covariance;
      covariance = (tmp_cov - 2.0 * (pt_sum * centroid.transpose())) / point_num + centroid * centroid.transpose();
      covariance *= (point_num - 1.0) / point_num;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(covariance(r, c));
    } else if (op == "rot3f" && in.size() == 9) {
      Eigen::Transform<float, 3, Eigen::Affine, Eigen::ColMajor> tr;
      tr.setIdentity();
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tr.matrix()(r, c) = static_cast<float>(in[r * 3 + c]);
      Eigen::Matrix3f R = tr.rotation();
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(R(r, c));
    } else if (op == "det6f" && in.size() == 36) {
      Eigen::Matrix<float, 6, 6> Hessian;
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hessian(r, c) = static_cast<float>(in[r * 6 + c]);
      out.push_back(Hessian.determinant());
    } else if (op == "inv6f" && in.size() == 42) {
      Eigen::Matrix<float, 6, 6> Hessian;
      Eigen::Matrix<float, 6, 1> B;
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Hessian(r, c) = static_cast<float>(in[r * 6 + c]);
      for (int r = 0; r < 6; r++) B(r) = static_cast<float>(in[36 + r]);
      Eigen::Matrix<float, 6, 6> Hi = Hessian.inverse();
      Eigen::Matrix<float, 6, 1> delta_x = Hessian.inverse() * B;
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) out.push_back(Hi(r, c));
      for (int r = 0; r < 6; r++) out.push_back(delta_x(r));
    } else if (op == "mul3f" && in.size() == 18) {
      Eigen::Matrix3f A, Bm;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { A(r, c) = static_cast<float>(in[r * 3 + c]); Bm(r, c) = static_cast<float>(in[9 + r * 3 + c]); }
      A *= Bm;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.push_back(A(r, c));
    } else if (op == "icpacc" && in.size() == 15) {
      Eigen::Matrix3f rotation_matrix_, hat;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rotation_matrix_(r, c) = static_cast<float>(in[r * 3 + c]);
      const float px = static_cast<float>(in[9]), py = static_cast<float>(in[10]), pz = static_cast<float>(in[11]);
      hat << 0.f, -pz, py, pz, 0.f, -px, -py, px, 0.f;
      Eigen::Vector3f err_dis(static_cast<float>(in[12]), static_cast<float>(in[13]), static_cast<float>(in[14]));
      Eigen::Matrix<float, 6, 6> Hessian;
      Eigen::Matrix<float, 6, 1> B;
      Hessian.setZero();
      B.setZero();
      Eigen::Matrix<float, 3, 6> Jacobian(Eigen::Matrix<float, 3, 6>::Zero());
      Jacobian.leftCols<3>() = Eigen::Matrix3f::Identity();
      Jacobian.rightCols<3>() = -rotation_matrix_ * hat;
      Hessian += Jacobian.transpose() * Jacobian;
      B += -Jacobian.transpose() * err_dis;
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) out.push_back(Hessian(r, c));
      for (int r = 0; r < 6; r++) out.push_back(B(r));
    } else {
      std::printf("ERR\n");
      continue;
    }
    emit(out);
  }
  return 0;
}
