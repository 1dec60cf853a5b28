// test_sanitize.cpp — the host-only code of the product (Newton / More-Thuente driver, shared scalar math, point packing,
// file readers) and the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only —
// GPU sanitizers are not available on the pool).  Built by tests/cpp/Makefile (`make asan`), run by
// tests/test_sanitizers_cpu.py.  The evaluator behind lsa_gn_solve is the oracle's derivative pass, i.e. the wiring
// tests/test_dist_cpu.py uses, so the driver walks the same control flow it walks on the GPU box.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "gn_math.h"
#include "lidar_slam_amd.h"
#include "lsa_io.hpp"
extern "C" {
#include "lsa_oracle.h"
}

void lsa_pack_xyzw(const void *pts, size_t n, size_t stride, size_t w_off, float *out4);

static int g_fail = 0;
#define CHECK(cond, ...)                                                                   \
  do {                                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); g_fail++; } \
  } while (0)

struct P8 { float v[8]; };

static std::vector<P8> street(unsigned seed, float ox, float yaw, size_t n) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::normal_distribution<float> N(0.f, 0.02f);
  std::vector<P8> c(n);
  float cy = std::cos(-yaw), sy = std::sin(-yaw);
  for (size_t i = 0; i < n; i++) {
    float x, y, z, pick = U(rng);
    if (pick < 0.5f) { x = -30.f + 60.f * U(rng); y = -11.f + 22.f * U(rng); z = -1.73f; }
    else if (pick < 0.7f) { x = -30.f + 60.f * U(rng); y = 11.f + 0.4f * std::floor(x / 9.f); z = -1.73f + 5.f * U(rng); }
    else if (pick < 0.9f) { x = -30.f + 60.f * U(rng); y = -11.f - 0.3f * std::floor(x / 7.f); z = -1.73f + 5.f * U(rng); }
    else { int b = (int)(U(rng) * 6); x = -25.f + 10.f * b + 2.f * U(rng); y = (b % 2 ? 6.f : -6.f) + 1.5f * U(rng); z = -1.73f + 1.5f * U(rng); }
    x += N(rng); y += N(rng); z += N(rng);
    float dx = x - ox, dy = y;
    P8 p{};
    p.v[0] = cy * dx - sy * dy; p.v[1] = sy * dx + cy * dy; p.v[2] = z; p.v[3] = 1.f; p.v[4] = 0.2f + 0.8f * U(rng);
    c[i] = p;
  }
  return c;
}

struct EvalCtx { orc_ndt *o; const std::vector<P8> *src; int calls; };

static int eval_cb(void *user, const float T[16], const double p[6], int kind, double out32[32]) {
  EvalCtx *e = (EvalCtx *)user;
  e->calls++;
  double g[6], H[36];
  long pairs = 0;
  memset(out32, 0, 32 * sizeof(double));
  if (kind == 2) {
    orc_ndt_hessian(e->o, e->src->data(), e->src->size(), sizeof(P8), T, p, H);
  } else {
    out32[0] = orc_ndt_derivatives(e->o, e->src->data(), e->src->size(), sizeof(P8), T, p, kind == 0, g, H, &pairs);
    for (int i = 0; i < 6; i++) out32[1 + i] = g[i];
  }
  if (kind != 1) {
    int k = 7;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) out32[k++] = H[i * 6 + j];
  }
  out32[28] = (double)pairs; out32[29] = (double)e->src->size();
  return 0;
}

static void test_math() {
  // sinf / cosf of gn_math.h restate glibc's algorithm: bit-identical to this libm
  unsigned long long st = 7;
  long bad = 0;
  for (long i = 0; i < 3000000; i++) {
    st = st * 6364136223846793005ULL + 1442695040888963407ULL;
    double u = (double)(st >> 11) / 9007199254740992.0;
    float x = (float)((u - 0.5) * ((i % 3 == 0) ? 0.4 : (i % 3 == 1) ? 6.6 : 0.002));
    bool ok = true;
    bad += gm::sincosf_one(x, false, &ok) != sinf(x);
    bad += gm::sincosf_one(x, true, &ok) != cosf(x);
    bad += !ok;
  }
  CHECK(bad == 0, "sinf/cosf differ from libm on %ld of 6e6 evaluations", bad);
  bool ok = true;
  (void)gm::sincosf_one(1000.f, false, &ok);
  CHECK(!ok, "range guard of sincosf_one");
  ok = true;
  (void)gm::sincosf_one(NAN, true, &ok);
  CHECK(!ok, "NaN guard of sincosf_one");
  // f64 sin / cos: within 2 ulp of libm over the Euler-angle range
  double worst = 0;
  for (long i = 0; i < 1000000; i++) {
    st = st * 6364136223846793005ULL + 1442695040888963407ULL;
    double x = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * ((i & 1) ? 13.0 : 0.5);
    double s, c;
    ok = true;
    gm::sincos_d(x, &s, &c, &ok);
    double es = std::fabs(s - std::sin(x)) / std::fmax(std::fabs(std::sin(x)) * DBL_EPSILON, DBL_MIN);
    double ec = std::fabs(c - std::cos(x)) / std::fmax(std::fabs(std::cos(x)) * DBL_EPSILON, DBL_MIN);
    worst = std::fmax(worst, std::fmax(es, ec));
  }
  CHECK(worst <= 2.0, "sincos_d is %.2f ulp from libm", worst);
  // lu6_solve against the residual, and its refusal of a singular system
  std::mt19937 rng(3);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int t = 0; t < 2000; t++) {
    double B[6][6], H[6][6], b[6], x[6];
    for (auto &r : B) for (double &v : r) v = U(rng) * 100.0;
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) { double v = (r == c); for (int k = 0; k < 6; k++) v += B[r][k] * B[c][k]; H[r][c] = -v; }
    for (double &v : b) v = U(rng) * 1e4;
    CHECK(gm::lu6_solve(H, b, x), "regular system refused");
    double res = 0, nb = 0;
    for (int r = 0; r < 6; r++) { double v = -b[r]; for (int c = 0; c < 6; c++) v += H[r][c] * x[c]; res = std::fmax(res, std::fabs(v)); nb = std::fmax(nb, std::fabs(b[r])); }
    CHECK(res <= 1e-6 * nb, "lu6 residual %.3g", res / nb);
    for (int c = 0; c < 6; c++) { H[5][c] = H[4][c]; H[c][5] = H[c][4]; }
    H[5][5] = H[4][4];
    CHECK(!gm::lu6_solve(H, b, x), "singular system accepted");
  }
}

// the f32 Eigen blocks (gn_math.h) against the oracle's copy (eigen_f32_oracle.c, which the golden fixtures pin to the
// reference's vendored Eigen): same bits on random, degenerate and non-finite inputs, and no UB on any of them
static void test_eigen_f32_blocks() {
  std::mt19937 rng(11);
  std::normal_distribution<float> N(0.f, 1.f);
  long bad_rot = 0, bad_lu = 0, bad_mul = 0;
  for (int t = 0; t < 20000; t++) {
    float L[9], Ra[9], Rb[9];
    // near-rotations (what the path feeds), general matrices, rank-deficient ones, zeros
    float a = N(rng) * (t % 3 ? 0.2f : 2.f), b = N(rng) * 0.2f, c = N(rng) * (t % 5 ? 0.2f : 3.f);
    float Rx[9] = {1, 0, 0, 0, std::cos(a), -std::sin(a), 0, std::sin(a), std::cos(a)};
    float Ry[9] = {std::cos(b), 0, std::sin(b), 0, 1, 0, -std::sin(b), 0, std::cos(b)};
    float Rz[9] = {std::cos(c), -std::sin(c), 0, std::sin(c), std::cos(c), 0, 0, 0, 1};
    float Rxy[9];
    gm::mul3f_eigen(Rx, Ry, Rxy);
    gm::mul3f_eigen(Rxy, Rz, L);
    if (t % 7 == 3) for (float &v : L) v += N(rng) * 1e-3f;
    if (t % 97 == 5) for (float &v : L) v = N(rng);
    if (t % 97 == 6) { for (float &v : L) v = 0.f; }
    if (t % 97 == 7) { L[3] = L[0]; L[4] = L[1]; L[5] = L[2]; }
    if (t % 997 == 8) L[4] = NAN;
    gm::rotation_polar_f32(L, Ra);
    orc_rotation_f32(L, Rb);
    bad_rot += std::memcmp(Ra, Rb, sizeof(Ra)) != 0;
    float H[36], B[6], da[6] = {0}, db[6] = {0}, Hi[36];
    for (int r = 0; r < 6; r++) for (int k = 0; k < 6; k++) H[r * 6 + k] = 0.f;
    int n = 6 + (t % 50);
    for (int i = 0; i < n; i++) {
      float J[6];
      for (int k = 0; k < 6; k++) J[k] = N(rng) * (k < 3 ? 1.f : 20.f);
      if (t % 11 == 4) J[4] = 0.f;   // a direction no correspondence constrains: exactly singular
      for (int r = 0; r < 6; r++) for (int k = 0; k < 6; k++) H[r * 6 + k] += J[r] * J[k];
    }
    for (float &v : B) v = N(rng) * 50.f;
    const float deta = gm::solve6f_eigen(H, B, da);
    const float detb = orc_det6f(H);
    if (detb != 0.0f) { orc_inv6f(H, Hi); orc_mul66f_vec(Hi, B, db); }
    bad_lu += std::memcmp(&deta, &detb, sizeof(float)) != 0 || (detb != 0.0f && std::memcmp(da, db, sizeof(da)) != 0);
    float A3[9], B3[9], Ma[9], Mb[9];
    for (float &v : A3) v = N(rng);
    for (float &v : B3) v = N(rng);
    gm::mul3f_eigen(A3, B3, Ma);
    orc_mul3f(A3, B3, Mb);
    bad_mul += std::memcmp(Ma, Mb, sizeof(Ma)) != 0;
  }
  CHECK(bad_rot == 0, "rotation_polar_f32 differs from the oracle's on %ld inputs", bad_rot);
  CHECK(bad_lu == 0, "6x6 f32 determinant / solve differ from the oracle's on %ld inputs", bad_lu);
  CHECK(bad_mul == 0, "Matrix3f product differs from the oracle's on %ld inputs", bad_mul);
  // lsa_matrix_to_pose == orc_matrix_to_pose on poses that have been through f32 products
  long bad_pose = 0;
  for (int t = 0; t < 5000; t++) {
    double p[6] = {N(rng) * 30., N(rng) * 30., N(rng), N(rng) * (t % 4 ? 0.05 : 1.5), N(rng) * 0.05, N(rng) * (t % 3 ? 0.3 : 3.0)};
    float T[16];
    lsa_pose_to_matrix(p, T);
    double pa[6], pb[6];
    lsa_matrix_to_pose(T, pa);
    orc_matrix_to_pose(T, pb);
    bad_pose += std::memcmp(pa, pb, sizeof(pa)) != 0;
  }
  CHECK(bad_pose == 0, "lsa_matrix_to_pose differs from the oracle's on %ld poses", bad_pose);
}

static void test_driver_and_oracle() {
  auto tgt = street(1, 0.f, 0.f, 60000), raw = street(2, 0.6f, 0.01f, 40000);
  // oracle voxel filter (also exercises in-place / capacity handling)
  std::vector<float> f4(raw.size() * 4);
  size_t nf = 0;
  CHECK(orc_voxel_filter(raw.data(), raw.size(), sizeof(P8), 16, 0.5f, 0.5f, 0.5f, f4.data(), raw.size(), &nf) == 0 && nf > 1000, "oracle voxel filter");
  std::vector<P8> src(nf);
  for (size_t i = 0; i < nf; i++) { P8 p{}; memcpy(p.v, &f4[4 * i], 12); p.v[3] = 1.f; p.v[4] = f4[4 * i + 3]; src[i] = p; }
  float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.3f, 0.05f, 0, 1};
  struct Cfg { float step, eps; int iters; int mode; } cfgs[] = {{0.1f, 0.01f, 30, 0}, {1.5f, 3.0f, 4, 0}, {0.1f, 0.01f, 30, 1}, {1.5f, 3.0f, 4, 1}};
  for (const Cfg &c : cfgs) {
    orc_ndt *o = orc_ndt_create(1.0f, c.step, c.eps, c.iters, 0.55f);
    orc_ndt_set_mode(o, c.mode);
    CHECK(orc_ndt_set_target(o, tgt.data(), tgt.size(), sizeof(P8), 16) == 0, "oracle set_target");
    float Tr[16], T[16];
    orc_ndt_stats sr;
    CHECK(orc_ndt_align(o, src.data(), src.size(), sizeof(P8), guess, Tr, &sr) == 0, "oracle align");
    EvalCtx e{o, &src, 0};
    lsa_ndt_stats st;
    std::vector<double> trace(9 * 64);
    int ntr = 0;
    int rc = lsa_gn_solve_trace(eval_cb, &e, 1.0f, c.step, c.eps, c.iters, 0.55f, c.mode, guess, T, &st, trace.data(), 64, &ntr);
    CHECK(rc == 0, "lsa_gn_solve_trace rc %d", rc);
    CHECK(st.iterations == sr.iterations && st.passes == sr.passes && st.ls_iterations == sr.ls_iterations && st.converged == sr.converged,
          "driver vs oracle counts: it %d/%d passes %d/%d ls %d/%d", st.iterations, sr.iterations, st.passes, sr.passes, st.ls_iterations, sr.ls_iterations);
    CHECK(ntr == st.passes && e.calls == st.passes, "trace rows %d, evaluator calls %d, passes %d", ntr, e.calls, st.passes);
    double d = 0;
    for (int i = 0; i < 16; i++) d = std::fmax(d, std::fabs((double)T[i] - Tr[i]));
    CHECK(d <= 1e-5, "driver pose differs from the oracle's by %.3g (step %.2f mode %d)", d, c.step, c.mode);
    CHECK(std::isfinite(orc_ndt_fitness(o, src.data(), src.size(), sizeof(P8), Tr)), "oracle fitness");
    orc_ndt_free(o);
  }
  // a source far from the map: zero gradient and Hessian -> singular system -> the SVD path, zero step, early exit
  {
    orc_ndt *o = orc_ndt_create(1.0f, 0.1f, 0.01f, 30, 0.55f);
    orc_ndt_set_target(o, tgt.data(), tgt.size(), sizeof(P8), 16);
    std::vector<P8> far(src.begin(), src.begin() + 200);
    for (P8 &p : far) p.v[0] += 1000.f;
    EvalCtx e{o, &far, 0};
    lsa_ndt_stats st;
    float T[16];
    CHECK(lsa_gn_solve(eval_cb, &e, 1.0f, 0.1f, 0.01f, 30, 0.55f, guess, T, &st) == 0 && st.passes == 1 && st.iterations == 0, "far source");
    CHECK(memcmp(T, guess, sizeof(T)) == 0, "far source leaves the guess");
    orc_ndt_free(o);
  }
  // ICP / fitness / yaw / box oracles on small clouds
  {
    std::vector<P8> a(tgt.begin(), tgt.begin() + 4000), b(src.begin(), src.begin() + 1500);
    float T[16];
    double s2[2], s4[4];
    CHECK(orc_icp_manual(a.data(), a.size(), sizeof(P8), b.data(), b.size(), sizeof(P8), 2.0f, 5, guess, T) == 0, "oracle icp_manual");
    CHECK(orc_icp_plane(a.data(), a.size(), sizeof(P8), b.data(), b.size(), sizeof(P8), 2.0f, 5, 1.0f, guess, T, s2) == 0, "oracle icp_plane");
    CHECK(orc_icp_pcl(a.data(), a.size(), sizeof(P8), b.data(), b.size(), sizeof(P8), 0.01f, 2.0f, 0.01f, 10, guess, T, s4) == 0, "oracle icp_pcl");
    CHECK(std::isfinite(orc_fitness_pcl(a.data(), a.size(), sizeof(P8), b.data(), b.size(), sizeof(P8), T, INFINITY)), "oracle fitness_pcl");
    float origin[3] = {0, 0, 0}, size[6] = {-20, 20, -10, 10, -3, 3};
    orc_yaw *y = orc_yaw_build(a.data(), a.size(), sizeof(P8), origin, 0.8);
    double probs[270];
    (void)orc_yaw_angle(y, b.data(), b.size(), sizeof(P8), probs);
    orc_yaw_free(y);
    std::vector<float> out(a.size() * 4);
    size_t nb = 0;
    CHECK(orc_box_filter(a.data(), a.size(), sizeof(P8), 16, size, origin, out.data(), a.size(), &nb) == 0 && nb <= a.size(), "oracle box filter");
  }
}

static void test_pack_and_io() {
  std::vector<P8> c = street(9, 0.f, 0.f, 1000);
  std::vector<float> out(4 * c.size());
  lsa_pack_xyzw(c.data(), c.size(), sizeof(P8), 16, out.data());
  CHECK(out[4 * 999 + 3] == c[999].v[4] && out[4 * 999] == c[999].v[0], "pack stride 32");
  lsa_pack_xyzw(c.data(), c.size(), sizeof(P8), SIZE_MAX, out.data());
  CHECK(out[4 * 999 + 3] == 1.0f, "pack without intensity");
  std::vector<float> xyz(3 * 1000), xyzw(4 * 1000);
  for (size_t i = 0; i < 1000; i++) { memcpy(&xyz[3 * i], c[i].v, 12); memcpy(&xyzw[4 * i], c[i].v, 12); xyzw[4 * i + 3] = c[i].v[4]; }
  lsa_pack_xyzw(xyz.data(), 1000, 12, SIZE_MAX, out.data());     // exactly-sized (n, 3) rows: no byte past the buffer
  CHECK(out[4 * 999 + 2] == c[999].v[2], "pack stride 12");
  lsa_pack_xyzw(xyzw.data(), 1000, 16, 12, out.data());
  CHECK(out[4 * 999 + 3] == c[999].v[4], "pack stride 16");
  lsa_pack_xyzw(nullptr, 0, 32, 16, out.data());
  // KITTI .bin and PCD (ascii / binary) round trips through the header-only readers
  std::string dir = "/tmp/lsa_sanitize_io";
  std::string cmd = "mkdir -p " + dir;
  CHECK(std::system(cmd.c_str()) == 0, "mkdir");
  {
    FILE *f = std::fopen((dir + "/a.bin").c_str(), "wb");
    std::fwrite(xyzw.data(), 4, xyzw.size(), f);
    std::fclose(f);
    lsa::Cloud cl;
    CHECK(lsa::ReadKittiBin(dir + "/a.bin", cl) && cl.points.size() == 1000 && cl.points[7].intensity == xyzw[31], "ReadKittiBin");
    f = std::fopen((dir + "/trunc.bin").c_str(), "wb");
    std::fwrite(xyzw.data(), 1, 30, f);   // not a multiple of 16 bytes
    std::fclose(f);
    CHECK(!lsa::ReadKittiBin(dir + "/trunc.bin", cl), "truncated .bin accepted");
    CHECK(!lsa::ReadKittiBin(dir + "/missing.bin", cl), "missing .bin accepted");
  }
  {
    FILE *f = std::fopen((dir + "/a.pcd").c_str(), "wb");
    std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 3\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA ascii\n");
    std::fprintf(f, "1 2 3 0.5\nnan 0 0 1\n-4.5 5.25 6 0.25\n");
    std::fclose(f);
    lsa::Cloud cl, cl2;
    CHECK(lsa::ReadPcd(dir + "/a.pcd", cl) && cl.points.size() == 3 && cl.points[2].y == 5.25f, "ReadPcd ascii");
    std::vector<int> idx;
    lsa::RemoveNaN(cl, cl2, idx);
    CHECK(cl2.points.size() == 2 && idx.size() == 2 && idx[1] == 2, "RemoveNaN");
    lsa::RemoveNaN(cl, cl, idx);   // in place
    CHECK(cl.points.size() == 2, "RemoveNaN in place");
    f = std::fopen((dir + "/b.pcd").c_str(), "wb");
    std::fprintf(f, "VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 1000\nHEIGHT 1\nPOINTS 1000\nDATA binary\n");
    std::fwrite(xyzw.data(), 4, xyzw.size(), f);
    std::fclose(f);
    CHECK(lsa::ReadPcd(dir + "/b.pcd", cl) && cl.points.size() == 1000 && cl.points[999].z == xyzw[4 * 999 + 2], "ReadPcd binary");
    f = std::fopen((dir + "/short.pcd").c_str(), "wb");
    std::fprintf(f, "VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 1000\nHEIGHT 1\nPOINTS 1000\nDATA binary\n");
    std::fwrite(xyzw.data(), 4, 100, f);   // body shorter than the header promises
    std::fclose(f);
    CHECK(!lsa::ReadPcd(dir + "/short.pcd", cl), "short binary PCD accepted");
  }
}

// The vectorised taker of check-worded workgroup rows (host_pack.cpp) against the scalar rule it restates (voxel_index_dev.h:
// column 31 = key ^ XOR_c rotl(word_c, c + 1); ndt_reduce_partials_host: sm[b % 32][col] += row_b[col] in ascending b).
bool lsa_rows_fast_available();
unsigned lsa_rows_take_fast(const double *partials, unsigned b, unsigned nb, unsigned long long key, double (*sm)[32]);
static void test_rows_fast() {
  if (!lsa_rows_fast_available()) { std::printf("rows: no AVX2 on this host, fast path not exercised\n"); return; }
  std::mt19937_64 rng(11);
  const unsigned nb = 135;
  const unsigned long long key = 0x9e3779b97f4a7c15ull;
  std::vector<double> rows((size_t)nb * 32 + 4);   // (+4: the last row ends inside the allocation whatever the loads' width)
  std::uniform_real_distribution<double> U(-1e6, 1e6);
  auto seal = [&](unsigned b) {
    unsigned long long x = 0;
    for (int c = 0; c < 31; c++) {
      unsigned long long w; memcpy(&w, &rows[(size_t)b * 32 + c], 8);
      const int r = c + 1;
      x ^= (w << r) | (w >> (64 - r));
    }
    x ^= key;
    memcpy(&rows[(size_t)b * 32 + 31], &x, 8);
  };
  for (unsigned b = 0; b < nb; b++) {
    for (int c = 0; c < 31; c++) rows[(size_t)b * 32 + c] = (c % 7 == 3) ? 0.0 : U(rng);
    seal(b);
  }
  double ref[32][32] = {}, got[32][32] = {};
  for (unsigned b = 0; b < nb; b++)
    for (int c = 0; c < 31; c++) ref[b & 31][c] += rows[(size_t)b * 32 + c];
  CHECK(lsa_rows_take_fast(rows.data(), 0, nb, key, got) == nb, "rows: a complete pass not taken whole");
  CHECK(memcmp(ref, got, sizeof(ref)) == 0, "rows: vector sums differ from the scalar ones");
  // a stale row (sealed under the previous pass's key), a half-arrived one (one payload word still old) and a wrong key: taken up to there
  unsigned long long stale = key + 2; memcpy(&stale, &rows[(size_t)40 * 32 + 31], 8); stale ^= key ^ (key - 2); memcpy(&rows[(size_t)40 * 32 + 31], &stale, 8);
  rows[(size_t)90 * 32 + 17] += 1.0;
  double part[32][32] = {};
  CHECK(lsa_rows_take_fast(rows.data(), 0, nb, key, part) == 40, "rows: a row of another pass accepted");
  CHECK(lsa_rows_take_fast(rows.data(), 41, nb, key, part) == 90, "rows: a half-arrived row accepted");
  CHECK(lsa_rows_take_fast(rows.data(), 91, nb, key, part) == nb && lsa_rows_take_fast(rows.data(), 0, nb, key + 1, part) == 0, "rows: tail / wrong key");
  for (int c = 0; c < 31; c++) { part[40 & 31][c] += rows[(size_t)40 * 32 + c]; part[90 & 31][c] += rows[(size_t)90 * 32 + c] - (c == 17 ? 1.0 : 0.0); }
  bool close = true;   // (the two rows were added out of order here, so only to rounding)
  for (int r = 0; r < 32; r++) for (int c = 0; c < 31; c++) close = close && std::fabs(part[r][c] - ref[r][c]) <= 1e-6;
  CHECK(close, "rows: resumed sums wrong");
}

int main() {
  test_rows_fast();
  test_math();
  test_eigen_f32_blocks();
  test_pack_and_io();
  test_driver_and_oracle();
  if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
  std::printf("sanitize ok\n");
  return 0;
}
