// test_host_mirror.cpp — drives the C++ host mirror (lidar-slam_amd/host/lsa_plugins.hpp) the way
// FrontEnd::Update does (front_end.cpp:88-341: Filter -> SetInputTarget -> ScanMatch -> GetFitnessScore) and checks every
// result against the CPU oracle (oracle/lsa_oracle.h, plain C).  Built and run by tests/test_gpu_cpp_host.py on the GPU box:
//   g++ -std=c++17 -O1 -Iinclude -Ilidar-slam_amd/host -Ioracle tests/cpp/test_host_mirror.cpp
//       -Llidar-slam_amd -llidar_slam_amd -Loracle -llsa_oracle -Wl,-rpath,...
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>

#include "lsa_oracle.h"
#include "lsa_plugins.hpp"

using namespace lsa;

static int g_fail = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__);    \
      std::printf(__VA_ARGS__);                           \
      std::printf("\n");                                  \
      g_fail++;                                           \
    }                                                     \
  } while (0)

// a small street: ground, two walls, a few boxes; sampled from a moving viewpoint
static CloudData::CLOUD_PTR make_cloud(unsigned seed, float ox, float yaw, size_t n) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::normal_distribution<float> N(0.f, 0.02f);
  auto c = std::make_shared<Cloud>();
  c->points.reserve(n);
  float cy = std::cos(-yaw), sy = std::sin(-yaw);
  for (size_t i = 0; i < n; i++) {
    float x, y, z;
    float pick = U(rng);
    if (pick < 0.5f) { x = -30.f + 60.f * U(rng); y = -11.f + 22.f * U(rng); z = -1.73f; }                 // ground
    else if (pick < 0.7f) { x = -30.f + 60.f * U(rng); y = 11.f + 0.8f * std::floor(x / 9.f) * 0.5f; z = -1.73f + 5.f * U(rng); }   // wall with setbacks
    else if (pick < 0.9f) { x = -30.f + 60.f * U(rng); y = -11.f - 0.6f * std::floor(x / 7.f) * 0.5f; z = -1.73f + 5.f * U(rng); }
    else { int b = (int)(U(rng) * 6); x = -25.f + 10.f * b + 2.f * U(rng); y = (b % 2 ? 6.f : -6.f) + 1.5f * U(rng); z = -1.73f + 1.5f * U(rng); }  // boxes
    x += N(rng); y += N(rng); z += N(rng);
    // world -> sensor at (ox, 0, 0) with heading yaw
    float dx = x - ox, dy = y;
    PointXYZI p;
    p.x = cy * dx - sy * dy; p.y = sy * dx + cy * dy; p.z = z; p.intensity = 0.2f + 0.8f * U(rng);
    c->points.push_back(p);
  }
  return c;
}

static void pose_err(const Matrix4f &a, const float b[16], double *dt, double *dr) {
  double t = 0, r = 0;
  for (int i = 0; i < 3; i++) t += (a(i, 3) - b[12 + i]) * (a(i, 3) - b[12 + i]);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r += (a(i, j) - b[j * 4 + i]) * (a(i, j) - b[j * 4 + i]);
  *dt = std::sqrt(t); *dr = std::sqrt(r);
}

int main() {
  std::shared_ptr<Context> ctx;
  try { ctx = std::make_shared<Context>(0); } catch (const std::exception &e) { std::printf("FAIL: %s\n", e.what()); return 2; }

  auto scan0 = make_cloud(1, 0.f, 0.f, 60000);
  auto scan1 = make_cloud(2, 0.6f, 0.01f, 60000);

  // ---- CloudFilterInterface: VoxelFilterHip / NoFilter, including in == out (viewer.cpp:207)
  std::shared_ptr<CloudFilterInterface> frame_filter = std::make_shared<VoxelFilterHip>(ctx, 0.5f, 0.5f, 0.5f);
  CloudData::CLOUD_PTR filtered(new Cloud());
  CHECK(frame_filter->Filter(scan1, filtered), "Filter returned false");
  std::vector<float> ref(scan1->size() * 4);
  size_t n_ref = 0;
  int rc = orc_voxel_filter(scan1->points.data(), scan1->size(), 32, 16, 0.5f, 0.5f, 0.5f, ref.data(), scan1->size(), &n_ref);
  CHECK(rc == 0 && n_ref == filtered->size(), "voxel count %zu vs oracle %zu", filtered->size(), n_ref);
  double worst = 0;
  for (size_t i = 0; i < n_ref && i < filtered->size(); i++) {
    const PointXYZI &p = filtered->points[i];
    worst = std::fmax(worst, std::fabs(p.x - ref[4 * i]));
    worst = std::fmax(worst, std::fabs(p.y - ref[4 * i + 1]));
    worst = std::fmax(worst, std::fabs(p.z - ref[4 * i + 2]));
    worst = std::fmax(worst, std::fabs(p.intensity - ref[4 * i + 3]));
  }
  CHECK(worst < 2e-5, "filter values differ by %g", worst);
  CloudData::CLOUD_PTR alias(new Cloud(*scan1));
  frame_filter->Filter(alias, alias);
  CHECK(alias->size() == filtered->size(), "in==out filter size %zu vs %zu", alias->size(), filtered->size());
  NoFilter nf;
  CloudData::CLOUD_PTR copy(new Cloud());
  nf.Filter(scan1, copy);
  CHECK(copy->size() == scan1->size() && copy->points[7].x == scan1->points[7].x, "NoFilter is not a deep copy");
  std::printf("filter: %zu -> %zu points, max |diff| vs oracle %.2e\n", scan1->size(), filtered->size(), worst);

  // ---- RegistrationInterface: NDT
  std::shared_ptr<RegistrationInterface> reg = std::make_shared<NDTRegistrationHip>(ctx, 1.0f, 0.1f, 0.01f, 30, 0.55f);
  CHECK(reg->SetInputTarget(scan0), "SetInputTarget returned false");
  Matrix4f guess = Matrix4f::Identity(), pose = Matrix4f::Identity();
  CloudData::CLOUD_PTR result_cloud(new Cloud());
  CHECK(reg->ScanMatch(filtered, guess, result_cloud, pose), "ScanMatch returned false");
  orc_ndt *o = orc_ndt_create(1.0f, 0.1f, 0.01f, 30, 0.55f);
  orc_ndt_set_target(o, scan0->points.data(), scan0->size(), 32, 16);
  float opose[16];
  orc_ndt_stats ost;
  orc_ndt_align(o, filtered->points.data(), filtered->size(), 32, guess.data(), opose, &ost);
  double dt, dr;
  pose_err(pose, opose, &dt, &dr);
  auto *ndt = static_cast<NDTRegistrationHip *>(reg.get());
  CHECK(dt < 1e-4 && dr < 1e-5, "NDT pose differs from oracle: dt %g dR %g", dt, dr);
  CHECK(ndt->stats().iterations == ost.iterations && ndt->stats().passes == ost.passes, "NDT iterations %d/%d passes %d/%d", ndt->stats().iterations,
        ost.iterations, ndt->stats().passes, ost.passes);
  CHECK(std::fabs(pose(0, 3) - 0.6f) < 0.08f, "NDT did not recover the 0.6 m motion: %g", pose(0, 3));
  float fit = reg->GetFitnessScore();
  double ofit = orc_ndt_fitness(o, filtered->points.data(), filtered->size(), 32, opose);
  CHECK(std::fabs(fit - ofit) < 1e-5 * ofit, "fitness %g vs oracle %g", fit, ofit);
  std::printf("ndt: %d iterations, x = %.4f, |dt| %.2e |dR| %.2e vs oracle, fitness %.6f (oracle %.6f)\n", ost.iterations, pose(0, 3), dt, dr, fit, ofit);
  orc_ndt_free(o);

  // ---- RegistrationInterface: ICP (ICPRegistrationManual semantics)
  std::shared_ptr<CloudFilterInterface> map_filter = std::make_shared<VoxelFilterHip>(ctx, 0.3f, 0.3f, 0.3f);
  CloudData::CLOUD_PTR map(new Cloud());
  map_filter->Filter(scan0, map);
  std::shared_ptr<RegistrationInterface> icp = std::make_shared<ICPRegistrationHip>(ctx, 2.0f, 30);
  icp->SetInputTarget(map);
  Matrix4f ipose = Matrix4f::Identity();
  icp->ScanMatch(filtered, guess, result_cloud, ipose);
  float oipose[16];
  orc_icp_manual(map->points.data(), map->size(), 32, filtered->points.data(), filtered->size(), 32, 2.0f, 30, guess.data(), oipose);
  pose_err(ipose, oipose, &dt, &dr);
  CHECK(dt < 3e-4 && dr < 3e-5, "ICP pose differs from oracle: dt %g dR %g", dt, dr);
  CHECK(result_cloud->size() == filtered->size(), "ICP result cloud not filled");
  CHECK(icp->GetFitnessScore() == 0.f, "ICP fitness must be 0 like the reference");
  std::printf("icp: x = %.4f, |dt| %.2e |dR| %.2e vs oracle\n", ipose(0, 3), dt, dr);

  // ---- the PCL-backed plugins: "NDT" (ndt_registration.cpp) and "ICP" (icp_registration.cpp); oracle = pcl_oracle.c (unpinned)
  {
    std::shared_ptr<RegistrationInterface> pndt = std::make_shared<NDTRegistrationHip>(ctx, 1.0f, 0.1f, 0.01f, 30);
    pndt->SetInputTarget(map);
    Matrix4f ppose = Matrix4f::Identity();
    CloudData::CLOUD_PTR aligned(new Cloud());
    pndt->ScanMatch(filtered, guess, aligned, ppose);
    orc_ndt *po = orc_ndt_create(1.0f, 0.1f, 0.01f, 30, 0.55f);
    orc_ndt_set_mode(po, 1);
    orc_ndt_set_target(po, map->points.data(), map->size(), 32, 16);
    float popose[16];
    orc_ndt_align(po, filtered->points.data(), filtered->size(), 32, guess.data(), popose, &ost);
    pose_err(ppose, popose, &dt, &dr);
    CHECK(dt < 1e-4 && dr < 1e-5, "NDT (PCL mode) pose differs from oracle: dt %g dR %g", dt, dr);
    CHECK(aligned->size() == filtered->size(), "NDT (PCL mode) must fill result_cloud_ptr like pcl's align()");
    float pfit = pndt->GetFitnessScore();
    double pofit = orc_fitness_pcl(map->points.data(), map->size(), 32, filtered->points.data(), filtered->size(), 32, ppose.data(), INFINITY);
    CHECK(std::fabs(pfit - pofit) < 1e-5 * pofit, "PCL fitness %g vs oracle %g", pfit, pofit);
    std::printf("ndt(pcl): x = %.4f, |dt| %.2e |dR| %.2e vs oracle, fitness %.6f (oracle %.6f)\n", ppose(0, 3), dt, dr, pfit, pofit);
    orc_ndt_free(po);

    std::shared_ptr<RegistrationInterface> picp = std::make_shared<ICPRegistrationHip>(ctx, 0.01f, 2.0f, 0.01f, 30);
    picp->SetInputTarget(map);
    Matrix4f pip = Matrix4f::Identity();
    picp->ScanMatch(filtered, guess, aligned, pip);
    float opip[16];
    double ist[4];
    orc_icp_pcl(map->points.data(), map->size(), 32, filtered->points.data(), filtered->size(), 32, 0.01f, 2.0f, 0.01f, 30, guess.data(), opip, ist);
    pose_err(pip, opip, &dt, &dr);
    auto st = static_cast<ICPRegistrationHip *>(picp.get())->stats();
    CHECK(dt < 3e-4 && dr < 3e-5, "ICP (PCL mode) pose differs from oracle: dt %g dR %g", dt, dr);
    CHECK(st.iterations == (int)ist[0] && st.converged == (int)ist[1], "ICP (PCL mode) iterations %d/%d", st.iterations, (int)ist[0]);
    float ifit = picp->GetFitnessScore();
    double oifit = orc_fitness_pcl(map->points.data(), map->size(), 32, filtered->points.data(), filtered->size(), 32, pip.data(), INFINITY);
    CHECK(std::fabs(ifit - oifit) < 1e-5 * oifit, "ICP PCL fitness %g vs oracle %g", ifit, oifit);
    std::printf("icp(pcl): %d iterations, x = %.4f, |dt| %.2e |dR| %.2e vs oracle, fitness %.6f\n", st.iterations, pip(0, 3), dt, dr, ifit);
  }

  // ---- BoxFilter + position-only initialisation (Matching's helpers) against their oracles
  {
    std::vector<float> size = {-30.f, 30.f, -10.f, 12.f, -2.f, 1.f}, origin = {2.5f, -1.f, 0.25f};
    BoxFilterHip box(ctx, size);
    box.SetOrigin(origin);
    std::vector<float> edge = box.GetEdge();
    CHECK(edge[0] == size[0] + origin[0] && edge[5] == size[5] + origin[2], "BoxFilter edge = size + origin");
    CloudData::CLOUD_PTR cropped(new Cloud());
    box.Filter(scan0, cropped);
    std::vector<float> oc(scan0->size() * 4);
    size_t on = 0;
    orc_box_filter(scan0->points.data(), scan0->size(), 32, 16, size.data(), origin.data(), oc.data(), scan0->size(), &on);
    CHECK(cropped->size() == on && on > 0 && on < scan0->size(), "BoxFilter kept %zu, oracle %zu", cropped->size(), on);
    bool same = cropped->size() == on;
    for (size_t i = 0; same && i < on; i++) same = cropped->points[i].x == oc[4 * i] && cropped->points[i].y == oc[4 * i + 1] && cropped->points[i].z == oc[4 * i + 2];
    CHECK(same, "BoxFilter output differs from the oracle");

    YawInitHip yi(ctx, 0.8);
    yi.SetLocalMap(map, 0.f, 0.f, 0.f);
    double yaw = yi.getInitialYawAngle(scan1);
    const float zero[3] = {0.f, 0.f, 0.f};
    orc_yaw *oy = orc_yaw_build(map->points.data(), map->size(), 32, zero, 0.8);
    double oyaw = orc_yaw_angle(oy, scan1->points.data(), scan1->size(), 32, nullptr);
    orc_yaw_free(oy);
    CHECK(yaw == oyaw, "initial yaw %g vs oracle %g", yaw, oyaw);
    std::printf("box filter: %zu -> %zu points; initial yaw %.4f rad (oracle %.4f)\n", scan0->size(), cropped->size(), yaw, oyaw);
  }

  // ---- FrontEnd mirror: one Update per host cloud against the same clouds as ONE run (bit-identical), along a short drive
  {
    std::vector<CloudData::CLOUD_PTR> drive;
    for (int i = 0; i < 10; i++) drive.push_back(make_cloud(100 + i, 0.5f * i, 0.004f * i, 70000));
    FrontEndHip one(ctx), run(ctx);
    std::vector<Matrix4f> single(drive.size(), Matrix4f::Identity()), batch;
    int key_frames = 0;
    for (size_t i = 0; i < drive.size(); i++) {
      CHECK(one.Update(drive[i], single[i]) && one.last_status() == LSA_OK, "FrontEnd::Update %zu: status %d", i, one.last_status());
      key_frames += one.last_stats().key_frame;
    }
    CHECK(run.UpdateRun(drive, batch) && run.last_status() == LSA_OK, "FrontEnd::UpdateRun: status %d", run.last_status());
    bool same = batch.size() == single.size();
    for (size_t i = 0; same && i < single.size(); i++)
      for (int e = 0; e < 16; e++) same = same && batch[i].data()[e] == single[i].data()[e];
    CHECK(same, "FrontEnd: the run and the single Updates disagree");
    const float moved = single.back()(0, 3);
    CHECK(key_frames >= 2 && std::fabs(moved - 4.5f) < 0.3f, "FrontEnd: %d key-frames, x = %.3f after 4.5 m", key_frames, moved);
    std::printf("front end: 10 host clouds, %d key-frames, x = %.3f m (truth 4.50), run == single Updates\n", key_frames, moved);
  }

  std::printf(g_fail ? "FAILED (%d)\n" : "ALL OK\n", g_fail);
  return g_fail ? 1 : 0;
}
