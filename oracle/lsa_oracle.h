/*
 * lsa_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference's scan-matching hot path
 * (VoxelFilter -> NDT / ICP scan-to-local-map).  Nothing under oracle/ is part
 * of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker / the timed CPU
 * baseline.  The product library (lidar-slam_amd/csrc) never links, loads or
 * calls anything in this directory.
 *
 * PARITY STATUS: "parity partially pinned".
 *   - The reference holds NO tests, golden vectors or fixtures for this path
 *     (SURVEY.md section 4 / 8c), and its in-tree NDT sources need PCL / boost /
 *     ROS headers that this image lacks, so they are unbuildable here under the
 *     no-stand-in rule (DESIGN.md section "Oracle").
 *   - What IS pinned against real reference code: the 3x3 symmetric
 *     eigen-solver (reference header SymmetricEigenSolver.h, compiled in place)
 *     and the vendored-Eigen building blocks the path calls (Matrix3d::inverse,
 *     JacobiSVD<6x6>::solve, Transform::rotation() + eulerAngles(0,1,2),
 *     Translation*AngleAxis^3, and ICPMANUAL's f32 determinant / inverse /
 *     products) — see oracle/_ref (built by oracle/Makefile) and
 *     tests/golden/eigen_blocks.json.
 *   - Everything else (grid build loops, radius search, derivative loops,
 *     More-Thuente control flow, PCL VoxelGrid, ICP) is restated from the
 *     reference text / published PCL 1.7 algorithm and is UNPINNED.
 *
 * Citations use SURVEY.md's abbreviations:
 *   NS/ = lidar_localization/src/models/registration/ndt_registration_manual/
 *   NI/ = lidar_localization/include/lidar_localization/models/registration/ndt_registration_manual/
 *   S/  = lidar_localization/src/
 */
#ifndef LSA_ORACLE_H_
#define LSA_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- small building blocks (pinned against oracle/_ref) ---------------- */

/* NI/SymmetricEigenSolver.h:55-273 incl. its two transcription quirks.
 * a: 3x3 row-major (only the upper triangle is read). evecs: row-major, eigenvectors in COLUMNS. */
void orc_eig3(const double a[9], double evals[3], double evecs[9]);
/* Eigen 3.2.92 LU/InverseImpl.h:124-169 cofactor inverse, row-major in/out. */
void orc_inv3(const double m[9], double out[9]);
/* x = V S^-1 U^T b with Eigen's rank rule (SVD/SVDBase.h:130-140,260-272); H row-major. returns rank. */
int orc_svd6_solve(const double H[36], const double b[6], double x[6]);
/* Translation(p0..2) * AngleAxis(p3,X) * AngleAxis(p4,Y) * AngleAxis(p5,Z) in f32
 * (NS/NormalDistributionsTransform.cpp:691-694). T: column-major 4x4. */
void orc_pose_to_matrix(const double p[6], float T[16]);
/* p = (translation, rotation.eulerAngles(0,1,2)) in f32 then widened (NS/...:331-337). */
void orc_matrix_to_pose(const float T[16], double p[6]);
/* The f32 Eigen calls of the path (eigen_f32_oracle.c; each pinned bit for bit by oracle/_ref, row-major 3x3 / 6x6):
 * Transform<float,3,Affine>::rotation() (NS/...:336), Matrix<float,6,6>::determinant() / inverse() / inverse() * B,
 * Matrix3f * Matrix3f and the per-point accumulation of S/models/registration/icp_registration_manual.cpp:156-176. */
void orc_rotation_f32(const float L[9], float R[9]);
float orc_det6f(const float H[36]);
void orc_inv6f(const float H[36], float Hi[36]);
void orc_mul66f_vec(const float Hi[36], const float B[6], float out[6]);
void orc_mul3f(const float A[9], const float B[9], float out[9]);
void orc_icp_acc(const float R[9], const float p[3], const float e[3], float H[36], float B[6]);
/* pcl::transformPointCloud f32 semantics (PCL 1.7 common/impl/transforms.hpp). xyz only. */
void orc_transform_point(const float T[16], const float in[3], float out[3]);

/* Eigen fixed-size-3 dot / mat-vec association (a0 + (a1 + a2)), pinned by oracle/_ref */
double orc_dot3(const double a[3], const double b[3]);
void orc_mv3(const double m[9], const double v[3], double out[3]);
/* NS/VoxelGrid.cpp:286,293-294 covariance expression and :317 eigen-floor rebuild */
void orc_cov_expr(const double tmp_cov[9], const double pt_sum[3], double n, double cov[9]);
void orc_cov_floor_rebuild(const double evecs[9], const double evals[3], double cov[9]);

/* ---- NDT voxel grid (NS/VoxelGrid.cpp) --------------------------------- */

typedef struct orc_grid orc_grid;

orc_grid *orc_grid_build(const void *pts, size_t n, size_t stride, size_t w_off, float leaf);
/* mode 0 = orc_grid_build; mode 1 = PCL 1.7 VoxelGridCovariance semantics (the "NDT" plugin's table; UNPINNED) */
orc_grid *orc_grid_build_mode(const void *pts, size_t n, size_t stride, size_t w_off, float leaf, int mode);
void orc_grid_free(orc_grid *g);
/* info[0..2]=vgrid dims, [3..5]=min_b, [6..8]=real_min, [9..11]=real_max, [12]=voxel_num */
void orc_grid_info(const orc_grid *g, int info[13]);
/* number of cells with >=1 point, and with points_per_voxel >= 6 (valid for NDT) */
void orc_grid_counts(const orc_grid *g, int *occupied, int *valid);
/* dense accessors by reference voxel id (idx-min_b + ...), VoxelGrid.cpp:237-242 */
int orc_grid_cell(const orc_grid *g, int vid, int *npts, int *ppv, double centroid[3], double icov[9], double *weight);
/* VoxelGrid.cpp:432-480; returns count, ids in the reference's x->y->z order */
int orc_grid_radius_search(const orc_grid *g, const float p[3], float radius, int *ids, int cap);
/* dump of all occupied cells in ascending (ix + iy*dx + iz*dx*dy over REAL bounds) order:
 * for each: ix,iy,iz (absolute voxel coords), npts, ppv -> ints[5*k..]; centroid(3) icov(9) weight(1) -> dbl[13*k..] */
int orc_grid_dump(const orc_grid *g, int *ints, double *dbls, int cap);
/* test hook: overwrite one cell's finalised record (absolute voxel coords) */
int orc_grid_set_cell(orc_grid *g, int ix, int iy, int iz, const double centroid[3], const double icov[9], double weight);

/* ---- NDT registration (NS/NormalDistributionsTransform.cpp) ------------ */

typedef struct orc_ndt orc_ndt;

typedef struct {
  int iterations;      /* nr_iterations_ (NS/...:383) */
  int ls_iterations;   /* real_iterations_ increment for this align (:753) */
  int passes;          /* derivative / hessian passes over the source */
  int converged;
  double score;        /* last score */
  double trans_probability;
  double pairs_per_pass; /* mean (point,voxel) pairs = K-bar * M */
  double final_p[6];
} orc_ndt_stats;

/* parameters arrive as float exactly as NDTRegistrationManual's ctor passes them
 * (S/models/registration/ndt_registration_manual.cpp:14-22) */
orc_ndt *orc_ndt_create(float res, float step_size, float trans_eps, int max_iter, float outlier_ratio);
void orc_ndt_free(orc_ndt *h);
/* 0 (default): in-tree NDTMANUAL arithmetic. 1: PCL-semantics mode for the "NDT" plugin
 * (S/models/registration/ndt_registration.cpp:46-66 -> pcl::NormalDistributionsTransform, PCL 1.7, not in the tree):
 * zero-seeded covariance, accurate eigen-solver, no static weight, More-Thuente phi_t assigned. UNPINNED. Call before
 * orc_ndt_set_target. */
void orc_ndt_set_mode(orc_ndt *h, int mode);
int orc_ndt_set_target(orc_ndt *h, const void *pts, size_t n, size_t stride, size_t w_off);
const orc_grid *orc_ndt_grid(const orc_ndt *h);
void orc_ndt_gauss(const orc_ndt *h, double *d1, double *d2);
/* one computeDerivatives pass (NS/...:391-445) at f32 transform T and f64 pose p.
 * g[6], H[36] row-major; returns score. pairs (may be NULL) receives #(point,voxel) pairs. */
double orc_ndt_derivatives(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16],
                           const double p[6], int compute_hessian, double g[6], double H[36], long *pairs);
/* computeHessian pass (NS/...:901-936) */
void orc_ndt_hessian(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16],
                     const double p[6], double H[36]);
/* Registration::align + computeTransformation (NS/Registration.cpp:76-95, NS/Normal...:310-389) */
int orc_ndt_align(orc_ndt *h, const void *src, size_t n, size_t stride, const float guess[16],
                  float out_pose[16], orc_ndt_stats *stats);
/* optional per-pass trace: rows of [p(6), score, step_len, kind] appended; kind 0=deriv+H,1=deriv,2=hessian */
void orc_ndt_set_trace(orc_ndt *h, double *buf, int cap_rows, int *n_rows);
/* getFitnessScore (NS/...:940-965) with the octree-bounded nearest-voxel search */
double orc_ndt_fitness(orc_ndt *h, const void *src, size_t n, size_t stride, const float T[16]);

/* ---- generic GN/More-Thuente driver with caller-supplied evaluator ----- */
/* evaluator: given x_t (6) and f32 T, kind (0 = score+g+H, 1 = score+g, 2 = H only) fill out28:
 * [score, g0..g5, H upper-tri 21 (row-major i<=j)]  — used by the world_size-2 gloo tests */
typedef int (*orc_eval_fn)(void *user, const double p[6], const float T[16], int kind, double out28[28]);

/* ---- PCL VoxelGrid filter (PCL 1.7 filters/impl/voxel_grid.hpp) -------- */
/* returns 0 ok, 1 = "leaf size too small" (output = copy of input), <0 error. out is n_out x 4 floats (x,y,z,intensity) */
int orc_voxel_filter(const void *pts, size_t n, size_t stride, size_t w_off, float lx, float ly, float lz,
                     float *out_xyzi, size_t cap, size_t *n_out);

/* BoxFilter::Filter (S/models/cloud_filter/box_filter.cpp:27-37 -> pcl::CropBox, UNPINNED): input order kept */
int orc_box_filter(const void *pts, size_t n, size_t stride, size_t w_off, const float size[6], const float origin[3],
                   float *out_xyzi, size_t cap, size_t *n_out);

/* ---- ICP (S/models/registration/icp_registration_manual.cpp:125-178) --- */
/* one iteration's sums of ICPRegistrationManual over a (shard of the) source at pose T: the exchange step's payload (icp_oracle.c) */
int orc_icp_sums(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride, float max_correspond_dis,
                 const float T[16], double out32[32]);
int orc_icp_manual(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                   float max_correspond_dis, int max_iter, const float guess[16], float out_pose[16]);

/* Point-to-plane ICP: build extension (BASELINE.json configs[2]; not in the reference), defined in icp_oracle.c.
 * stats: [iterations, correspondences of the last iteration] */
int orc_icp_plane(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                  float max_correspond_dis, int max_iter, float normal_leaf, const float guess[16], float out_pose[16],
                  double stats[2]);

/* ---- Matching's position-only initialisation (yaw_oracle.c; S/matching/matching.cpp:267-307,344-422; restated, UNPINNED) */
typedef struct orc_yaw orc_yaw;
/* generateGauss2DMapCells over (map - origin); grid_resolution as in the YAML (clamped to >= 0.1, :137-138) */
orc_yaw *orc_yaw_build(const void *map, size_t n, size_t stride, const float origin[3], double grid_resolution);
void orc_yaw_free(orc_yaw *y);
void orc_yaw_info(const orc_yaw *y, int wh[2], float mn[3], float mx[3]);
/* cell k = cell_x * height + cell_y */
void orc_yaw_cells(const orc_yaw *y, float *mu, float *sigma, int *cnt);
void orc_yaw_rotation(int i, float R[9]);
/* getInitialYawAngle; probs270 (optional) receives the 270 scores */
double orc_yaw_angle(const orc_yaw *y, const void *scan, size_t n, size_t stride, double *probs270);

/* ---- PCL-semantics pieces (pcl_oracle.c; third-party PCL 1.7, absent from the tree: UNPINNED) -------------- */
/* symmetric 3x3 eigen-decomposition, eigenvalues ascending, eigenvectors in columns (row-major storage):
 * what Eigen::SelfAdjointEigenSolver<Matrix3d> returns up to rounding and sign. cyclic Jacobi. */
void orc_eig3_jacobi(const double a[9], double evals[3], double evecs[9]);
/* pcl::Registration::getFitnessScore(max_range) (registration/impl/registration.hpp): mean over transformed source
 * points of the SQUARED f32 distance to the nearest TARGET POINT, pairs with d2 > max_range dropped; DBL_MAX if none.
 * Consumer: S/mapping/loop_closing/loop_closing.cpp:253. */
double orc_fitness_pcl(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                       const float T[16], double max_range);
/* pcl::IterativeClosestPoint::computeTransformation with the parameters of S/models/registration/icp_registration.cpp:
 * 30-33 (nearest-neighbour correspondences gated at max_correspond_dis, TransformationEstimationSVD = f32 Umeyama,
 * DefaultConvergenceCriteria: max_iter / translation^2 <= trans_eps && cos(angle) >= 1 - trans_eps / relative MSE change
 * < euclid_eps / absolute MSE change < 1e-12). stats: [iterations, converged, last #correspondences, last mse]. */
int orc_icp_pcl(const void *tgt, size_t nt, size_t tstride, const void *src, size_t ns, size_t sstride,
                float euclid_eps, float max_correspond_dis, float trans_eps, int max_iter, const float guess[16],
                float out_pose[16], double stats[4]);

#ifdef __cplusplus
}
#endif
#endif
