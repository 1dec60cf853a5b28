// lsa_plugins.hpp — header-only C++ host side above the C-ABI, mirroring the reference's plugin layer.
//
// Same class shapes, method names, argument meaning and error behaviour as
//   lidar_localization::CloudFilterInterface  (include/lidar_localization/models/cloud_filter/cloud_filter_interface.hpp:13-18)
//   lidar_localization::VoxelFilter           (src/models/cloud_filter/voxel_filter.cpp:12-41)
//   lidar_localization::NoFilter              (src/models/cloud_filter/no_filter.cpp:13-16)
//   lidar_localization::RegistrationInterface (include/lidar_localization/models/registration/registration_interface.hpp:14-24)
//   lidar_localization::NDTRegistrationManual (src/models/registration/ndt_registration_manual.cpp:12-78)
//   lidar_localization::ICPRegistrationManual (src/models/registration/icp_registration_manual.cpp:12-67)
// but free of PCL / Eigen / yaml-cpp so it builds in this image: the cloud and pose types below are layout-compatible
// with pcl::PointCloud<pcl::PointXYZI>::points and Eigen::Matrix4f.  INTEGRATION.md shows the same adaptors written
// against the reference's own types (CloudData::CLOUD_PTR, Eigen::Matrix4f, YAML::Node) for its catkin tree.
//
// Like the reference, every method returns true and reports problems on stderr (the reference: LOG(ERROR)); unlike it,
// a failed call leaves the outputs untouched and last_status() holds the lsa_status.
#ifndef LSA_PLUGINS_HPP_
#define LSA_PLUGINS_HPP_

#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "lidar_slam_amd.h"

namespace lsa {

struct alignas(16) PointXYZI {  // == pcl::PointXYZI (cloud_data.hpp:35): 32 bytes
  float x = 0, y = 0, z = 0, data3 = 1.0f;
  float intensity = 0, pad0 = 0, pad1 = 0, pad2 = 0;
};
static_assert(sizeof(PointXYZI) == LSA_POINTXYZI_STRIDE, "PointXYZI must be 32 bytes");

struct Cloud {
  std::vector<PointXYZI> points;
  size_t size() const { return points.size(); }
};

struct CloudData {  // cloud_data.hpp:33-48
  using POINT = PointXYZI;
  using CLOUD = Cloud;
  using CLOUD_PTR = std::shared_ptr<Cloud>;
  double time = 0.0;
  CLOUD_PTR cloud_ptr = std::make_shared<Cloud>();
};

struct Matrix4f {  // column-major like Eigen::Matrix4f
  float m[16];
  static Matrix4f Identity() {
    Matrix4f r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
  }
  float &operator()(int row, int col) { return m[col * 4 + row]; }
  float operator()(int row, int col) const { return m[col * 4 + row]; }
  const float *data() const { return m; }
  float *data() { return m; }
};

// one per process per GPU; owns the lsa_context
class Context {
 public:
  explicit Context(int device = 0, void *hip_stream = nullptr) {
    int rc = lsa_context_create(&ctx_, device, hip_stream);
    if (rc != LSA_OK) throw std::runtime_error(std::string("lsa_context_create: ") + lsa_last_global_error());
  }
  ~Context() { lsa_context_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  lsa_context *get() const { return ctx_; }

 private:
  lsa_context *ctx_ = nullptr;
};

inline bool report(const char *who, lsa_context *ctx, int rc) {
  if (rc != LSA_OK) std::fprintf(stderr, "[lsa] %s failed (%d): %s\n", who, rc, lsa_last_error(ctx));
  return true;  // the reference's plugins return true unconditionally and callers ignore the value
}

// ---------------------------------------------------------------------------------------------- cloud_filter
class CloudFilterInterface {
 public:
  virtual ~CloudFilterInterface() = default;
  virtual bool Filter(const CloudData::CLOUD_PTR &input_cloud_ptr, CloudData::CLOUD_PTR &filtered_cloud_ptr) = 0;
};

class NoFilter : public CloudFilterInterface {
 public:
  bool Filter(const CloudData::CLOUD_PTR &in, CloudData::CLOUD_PTR &out) override {
    if (out.get() != in.get()) out->points = in->points;  // no_filter.cpp:14: deep copy
    return true;
  }
};

class VoxelFilterHip : public CloudFilterInterface {
 public:
  VoxelFilterHip(const std::shared_ptr<Context> &ctx, float leaf_size_x, float leaf_size_y, float leaf_size_z) : ctx_(ctx) {
    report("lsa_voxel_filter_create", ctx_->get(), lsa_voxel_filter_create(&f_, ctx_->get(), leaf_size_x, leaf_size_y, leaf_size_z));
  }
  ~VoxelFilterHip() override { lsa_voxel_filter_destroy(f_); }
  bool Filter(const CloudData::CLOUD_PTR &in, CloudData::CLOUD_PTR &out) override {
    // `out` may be `in` (viewer.cpp:207) or an empty pre-allocated cloud (front_end.cpp:106-107)
    std::vector<PointXYZI> tmp(in->points.size());
    size_t n_out = 0;
    status_ = lsa_voxel_filter_apply(f_, in->points.data(), in->points.size(), sizeof(PointXYZI), LSA_POINTXYZI_W_OFF, tmp.data(), tmp.size(),
                                     sizeof(PointXYZI), LSA_POINTXYZI_W_OFF, &n_out);
    if (status_ == LSA_OK || status_ == LSA_ERR_RANGE) {  // PCL passes the input through when the leaf is too small
      tmp.resize(n_out);
      out->points.swap(tmp);
      return true;
    }
    return report("VoxelFilterHip::Filter", ctx_->get(), status_);
  }
  int last_status() const { return status_; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_voxel_filter *f_ = nullptr;
  int status_ = LSA_OK;
};

// BoxFilter(YAML::Node node) with node["box_filter_size"] = [min_x, max_x, min_y, max_y, min_z, max_z] (box_filter.cpp:14-73)
class BoxFilterHip : public CloudFilterInterface {
 public:
  BoxFilterHip(const std::shared_ptr<Context> &ctx, const std::vector<float> &box_filter_size) : ctx_(ctx) {
    if (box_filter_size.size() != 6) throw std::invalid_argument("box_filter_size needs 6 values");
    report("lsa_box_filter_create", ctx_->get(), lsa_box_filter_create(&f_, ctx_->get(), box_filter_size.data()));
  }
  ~BoxFilterHip() override { lsa_box_filter_destroy(f_); }
  bool Filter(const CloudData::CLOUD_PTR &in, CloudData::CLOUD_PTR &out) override {  // :27-37 (clears, then fills the output)
    std::vector<PointXYZI> tmp(in->points.size());
    size_t n_out = 0;
    status_ = lsa_box_filter_apply(f_, in->points.data(), in->points.size(), sizeof(PointXYZI), LSA_POINTXYZI_W_OFF, tmp.data(), tmp.size(),
                                   sizeof(PointXYZI), LSA_POINTXYZI_W_OFF, &n_out);
    if (status_ != LSA_OK) return report("BoxFilterHip::Filter", ctx_->get(), status_);
    tmp.resize(n_out);
    out->points.swap(tmp);
    return true;
  }
  void SetSize(const std::vector<float> &size) { if (size.size() == 6) status_ = lsa_box_filter_set_size(f_, size.data()); }      // :39-54
  void SetOrigin(const std::vector<float> &origin) { if (origin.size() == 3) status_ = lsa_box_filter_set_origin(f_, origin.data()); }  // :56-60
  std::vector<float> GetEdge() {                                                                                                  // :70-73
    std::vector<float> e(6);
    status_ = lsa_box_filter_get_edge(f_, e.data());
    return e;
  }
  int last_status() const { return status_; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_box_filter *f_ = nullptr;
  int status_ = LSA_OK;
};

// Matching's position-only initialisation (matching.cpp:267-307, 344-422): generateGauss2DMapCells + getInitialYawAngle
class YawInitHip {
 public:
  YawInitHip(const std::shared_ptr<Context> &ctx, double grid_resolution = 0.8) : ctx_(ctx) {
    report("lsa_yaw_init_create", ctx_->get(), lsa_yaw_init_create(&y_, ctx_->get(), grid_resolution));
  }
  ~YawInitHip() { lsa_yaw_init_destroy(y_); }
  bool SetLocalMap(const CloudData::CLOUD_PTR &local_map, float x, float y, float z) {   // ResetLocalMap(x, y, z) + SetInitPose
    const float origin[3] = {x, y, z};
    status_ = lsa_yaw_init_set_local_map(y_, local_map->points.data(), local_map->points.size(), sizeof(PointXYZI), origin);
    return report("YawInitHip::SetLocalMap", ctx_->get(), status_);
  }
  double getInitialYawAngle(const CloudData::CLOUD_PTR &current_cloud) {
    double yaw = 0.0;
    status_ = lsa_yaw_init_get_initial_yaw(y_, current_cloud->points.data(), current_cloud->points.size(), sizeof(PointXYZI), &yaw, nullptr);
    report("YawInitHip::getInitialYawAngle", ctx_->get(), status_);
    return yaw;
  }
  int last_status() const { return status_; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_yaw_init *y_ = nullptr;
  int status_ = LSA_OK;
};

// ---------------------------------------------------------------------------------------------- registration
class RegistrationInterface {
 public:
  virtual ~RegistrationInterface() = default;
  virtual bool SetInputTarget(const CloudData::CLOUD_PTR &input_target) = 0;
  virtual bool ScanMatch(const CloudData::CLOUD_PTR &input_source, const Matrix4f &predict_pose, CloudData::CLOUD_PTR &result_cloud_ptr,
                         Matrix4f &result_pose) = 0;
  virtual float GetFitnessScore() = 0;
};

// pcl::transformPointCloud(src, out, T): what the PCL-backed plugins leave in result_cloud_ptr
inline void transform_into(const CloudData::CLOUD &src, const Matrix4f &T, CloudData::CLOUD &out) {
  out.points = src.points;
  for (auto &p : out.points) {
    float x = p.x, y = p.y, z = p.z;
    p.x = T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3);
    p.y = T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3);
    p.z = T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3);
  }
}

class NDTRegistrationHip : public RegistrationInterface {
 public:
  NDTRegistrationHip(const std::shared_ptr<Context> &ctx, float res, float step_size, float trans_eps, int max_iter, float outlier_ratio)
      : ctx_(ctx) {
    report("lsa_ndt_create", ctx_->get(), lsa_ndt_create(&ndt_, ctx_->get(), res, step_size, trans_eps, max_iter, outlier_ratio));
    std::printf("NDT HIP params: res %g, step_size %g, trans_eps %g, max_iter %d, outlier_ratio %g\n", res, step_size, trans_eps, max_iter,
                outlier_ratio);  // the reference echoes its parameters (ndt_registration_manual.cpp:37-44)
  }
  // NDTRegistration(float res, float step_size, float trans_eps, int max_iter) (ndt_registration.cpp:18-22): the "NDT"
  // plugin = pcl::NormalDistributionsTransform semantics (LSA_NDT_MODE_PCL); PCL's default outlier ratio 0.55.
  NDTRegistrationHip(const std::shared_ptr<Context> &ctx, float res, float step_size, float trans_eps, int max_iter) : ctx_(ctx) {
    report("lsa_ndt_create", ctx_->get(), lsa_ndt_create(&ndt_, ctx_->get(), res, step_size, trans_eps, max_iter, 0.55f));
    report("lsa_ndt_set_mode", ctx_->get(), lsa_ndt_set_mode(ndt_, LSA_NDT_MODE_PCL));
    pcl_mode_ = true;
    std::printf("NDT HIP (PCL semantics) params: res %g, step_size %g, trans_eps %g, max_iter %d\n", res, step_size, trans_eps, max_iter);
  }
  ~NDTRegistrationHip() override { lsa_ndt_destroy(ndt_); }
  bool SetInputTarget(const CloudData::CLOUD_PTR &t) override {
    status_ = lsa_ndt_set_input_target(ndt_, t->points.data(), t->points.size(), sizeof(PointXYZI), LSA_POINTXYZI_W_OFF);
    return report("NDTRegistrationHip::SetInputTarget", ctx_->get(), status_);
  }
  bool ScanMatch(const CloudData::CLOUD_PTR &src, const Matrix4f &predict_pose, CloudData::CLOUD_PTR &result_cloud_ptr, Matrix4f &result_pose) override {
    // never written by NDTRegistrationManual (Registration.cpp:92-95), read by no caller; pcl's align() fills it
    Matrix4f out = predict_pose;
    status_ = lsa_ndt_scan_match(ndt_, src->points.data(), src->points.size(), sizeof(PointXYZI), predict_pose.data(), out.data(), &stats_);
    if (status_ == LSA_OK) {
      result_pose = out;
      if (pcl_mode_ && result_cloud_ptr) transform_into(*src, out, *result_cloud_ptr);
    }
    return report("NDTRegistrationHip::ScanMatch", ctx_->get(), status_);
  }
  float GetFitnessScore() override {
    float s = 0.f;
    status_ = lsa_ndt_get_fitness_score(ndt_, &s);
    report("NDTRegistrationHip::GetFitnessScore", ctx_->get(), status_);
    return s;
  }
  const lsa_ndt_stats &stats() const { return stats_; }
  int last_status() const { return status_; }
  lsa_ndt *handle() const { return ndt_; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_ndt *ndt_ = nullptr;
  lsa_ndt_stats stats_{};
  int status_ = LSA_OK;
  bool pcl_mode_ = false;
};

class ICPRegistrationHip : public RegistrationInterface {
 public:
  // ICPRegistrationManual(max_correspond_dis, max_iter) (icp_registration_manual.cpp:29-33): point-to-point GN (:125-178)
  ICPRegistrationHip(const std::shared_ptr<Context> &ctx, float max_correspond_dis, int max_iter, int mode = LSA_ICP_MODE_MANUAL) : ctx_(ctx), mode_(mode) {
    report("lsa_icp_create", ctx_->get(), lsa_icp_create(&icp_, ctx_->get(), max_correspond_dis, max_iter, mode));
  }
  // ICPRegistration(euclid_eps, max_correspond_dis, trans_eps, max_iter) (icp_registration.cpp:20-33): the "ICP" plugin =
  // pcl::IterativeClosestPoint semantics (LSA_ICP_MODE_PCL)
  ICPRegistrationHip(const std::shared_ptr<Context> &ctx, float euclid_eps, float max_correspond_dis, float trans_eps, int max_iter)
      : ctx_(ctx), mode_(LSA_ICP_MODE_PCL) {
    report("lsa_icp_create", ctx_->get(), lsa_icp_create(&icp_, ctx_->get(), max_correspond_dis, max_iter, LSA_ICP_MODE_PCL));
    report("lsa_icp_set_criteria", ctx_->get(), lsa_icp_set_criteria(icp_, euclid_eps, trans_eps));
  }
  ~ICPRegistrationHip() override { lsa_icp_destroy(icp_); }
  bool SetInputTarget(const CloudData::CLOUD_PTR &t) override {
    status_ = lsa_icp_set_input_target(icp_, t->points.data(), t->points.size(), sizeof(PointXYZI));
    return report("ICPRegistrationHip::SetInputTarget", ctx_->get(), status_);
  }
  bool ScanMatch(const CloudData::CLOUD_PTR &src, const Matrix4f &predict_pose, CloudData::CLOUD_PTR &result_cloud_ptr, Matrix4f &result_pose) override {
    Matrix4f out = predict_pose;
    status_ = lsa_icp_scan_match(icp_, src->points.data(), src->points.size(), sizeof(PointXYZI), predict_pose.data(), out.data());
    if (status_ == LSA_OK) {
      result_pose = out;
      // both ICP plugins fill result_cloud_ptr with the aligned source (icp_registration_manual.cpp:57, icp_registration.cpp:55)
      if (result_cloud_ptr) transform_into(*src, out, *result_cloud_ptr);
    }
    return report("ICPRegistrationHip::ScanMatch", ctx_->get(), status_);
  }
  float GetFitnessScore() override {  // ICPMANUAL: unimplemented in the reference, returns 0 (:62-67); "ICP": pcl's getFitnessScore
    float s = 0.f;
    status_ = lsa_icp_get_fitness_score(icp_, &s);
    report("ICPRegistrationHip::GetFitnessScore", ctx_->get(), status_);
    return s;
  }
  int last_status() const { return status_; }
  lsa_icp_stats stats() const { lsa_icp_stats st{}; lsa_icp_get_stats(icp_, &st); return st; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_icp *icp_ = nullptr;
  int status_ = LSA_OK;
  int mode_ = LSA_ICP_MODE_MANUAL;
};

// ---------------------------------------------------------------------------------------------- front end
// FrontEnd (include/lidar_localization/mapping/front_end/front_end.hpp; front_end.cpp:30-424) with the sliding local map kept on the
// device: SetInitPose, Update(cloud_data, cloud_pose) — one host cloud in, its pose out — and, for a caller that holds a queue of
// clouds (front_end_flow.cpp:18-36), UpdateRun: the whole queue in one call with the next scans' packing / PCIe / frame filter under
// the current scan's registration.  Constructed with the YAML defaults of config/mapping/front_end.yaml.
class FrontEndHip {
 public:
  static lsa_front_end_config DefaultConfig() {
    lsa_front_end_config c{};
    c.key_frame_distance = 2.0f; c.local_frame_num = 20; c.use_frame_filter = 1; c.use_map_filter = 1;
    for (int a = 0; a < 3; a++) { c.frame_leaf[a] = 0.5f; c.map_leaf[a] = 0.3f; }
    c.ndt_res = 1.0f; c.ndt_step_size = 0.1f; c.ndt_trans_eps = 0.01f; c.ndt_outlier_ratio = 0.55f; c.ndt_max_iter = 30;
    c.reset_intensity = 1;
    return c;
  }
  explicit FrontEndHip(std::shared_ptr<Context> ctx, const lsa_front_end_config &cfg = DefaultConfig()) : ctx_(std::move(ctx)) {
    int rc = lsa_front_end_create(&fe_, ctx_->get(), &cfg);
    if (rc != LSA_OK) throw std::runtime_error(std::string("lsa_front_end_create: ") + lsa_last_error(ctx_->get()));
  }
  ~FrontEndHip() { lsa_front_end_destroy(fe_); }
  FrontEndHip(const FrontEndHip &) = delete;
  FrontEndHip &operator=(const FrontEndHip &) = delete;
  bool SetInitPose(const Matrix4f &init_pose) { status_ = lsa_front_end_set_init_pose(fe_, init_pose.data()); return report("FrontEnd::SetInitPose", ctx_->get(), status_); }
  bool Update(const CloudData::CLOUD_PTR &cloud, Matrix4f &cloud_pose) {   // front_end.cpp:88-341
    status_ = lsa_front_end_update(fe_, cloud->points.data(), cloud->size(), sizeof(PointXYZI), offsetof(PointXYZI, intensity), cloud_pose.data(), &last_);
    return report("FrontEnd::Update", ctx_->get(), status_);
  }
  bool UpdateRun(const std::vector<CloudData::CLOUD_PTR> &clouds, std::vector<Matrix4f> &poses) {   // front_end_flow.cpp:18-36 over a queue
    std::vector<const void *> ptrs(clouds.size());
    std::vector<size_t> n(clouds.size());
    for (size_t k = 0; k < clouds.size(); k++) { ptrs[k] = clouds[k]->points.data(); n[k] = clouds[k]->size(); }
    poses.assign(clouds.size(), Matrix4f::Identity());
    static_assert(sizeof(Matrix4f) == 16 * sizeof(float), "poses are handed over as 16 floats each");
    status_ = lsa_front_end_update_batch(fe_, clouds.size(), ptrs.data(), n.data(), sizeof(PointXYZI), offsetof(PointXYZI, intensity),
                                         clouds.empty() ? nullptr : poses[0].data(), nullptr);
    return report("FrontEnd::UpdateRun", ctx_->get(), status_);
  }
  const lsa_front_end_stats &last_stats() const { return last_; }
  int last_status() const { return status_; }
  lsa_front_end *get() const { return fe_; }

 private:
  std::shared_ptr<Context> ctx_;
  lsa_front_end *fe_ = nullptr;
  lsa_front_end_stats last_{};
  int status_ = LSA_OK;
};

}  // namespace lsa
#endif
