// ndt_grid.hip — target voxel table for NDT: per-cell mean / covariance / inverse covariance / static weight.
//
// Replaces lidar_localization::VoxelGrid::setInput (reference: lidar_localization/src/models/registration/
// ndt_registration_manual/VoxelGrid.cpp:326-358 = findBoundaries :360-400 + scatterPointsToVoxelGrid :244-270 +
// computeCentroidAndCovariance :273-323) with the sparse voxel index of voxel_index.hip:
//   bbox -> occupancy bitmap + popcount ranks -> order-independent fixed-point sums of (p - corner), its outer
//   product and the intensity -> one finalize thread per occupied cell.
// The reference keeps a DENSE grid of ~232 B per cell (160 k cells at KITTI size, 861 k for the 2 M-pt map);
// here the lookup structure is 12 B per 32 cells plus 80 B per OCCUPIED cell, which stays in the 4 MiB per-XCD L2.
#include <cstdlib>

#include "eig3_dev.h"
#include "ndt_internal.h"

namespace {
constexpr int kBlock = 256;

// ---- 3x3 symmetric eigen-solver exactly as the reference's hand-rolled one, including its two quirks
// (reference: include/.../ndt_registration_manual/SymmetricEigenSolver.h:55-273; quirks :172-174 and :179-180).
// They matter: with the mis-normalised first eigenvector the "eigenvectors" are an orthogonal but rotated frame,
// and the regularised covariance V L' V^-1 (VoxelGrid.cpp:317) inherits that rotation.
__device__ void cross3(const double u[3], const double v[3], double o[3]) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}

__device__ void eig3_reference(const double a_in[6] /*xx,xy,xz,yy,yz,zz*/, double evals[3], double V[3][3] /*columns*/) {
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
__device__ void inv3(const double m[3][3], double o[3][3]) {
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

// one thread per occupied cell (computeCentroidAndCovariance, VoxelGrid.cpp:273-323)
__global__ __launch_bounds__(kBlock) void k_ndt_finalize(long long *__restrict__ acc, int32_t *__restrict__ cnt,
                                                         const int32_t *__restrict__ cell_of_rank, size_t nocc, CellMap m, FixScales fs,
                                                         double *__restrict__ rec, int32_t *__restrict__ state, int32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ vbits, int pcl) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nocc) return;
  long long *a = acc + r * 10;
  int ni = cnt[r];
  double n = (double)ni;
  int cell = cell_of_rank[r];
  double o[3];
  lsa_cell_origin(m, cell, o);
  double md[3] = {((double)a[0] * fs.inv_s1) / n, ((double)a[1] * fs.inv_s1) / n, ((double)a[2] * fs.inv_s1) / n};
  double *R = rec + r * kRecDoubles;
  R[0] = o[0] + md[0]; R[1] = o[1] + md[1]; R[2] = o[2] + md[2];  // centroid (:286)
  R[9] = pcl ? 1.0 : ((double)a[9] * fs.inv_sw) / n;                // mean static weight (:287); PCL has none
  for (int k = 3; k < 9; k++) R[k] = 0.0;
  int st = 0;  // 0: fewer than min_points_per_voxel_ (6) points, 1: valid, -1: rejected eigenvalues
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
  state[r] = st;
  if (st == 1) atomicOr(vbits + (cell >> 5), 1u << (cell & 31));
  // leave the index clean for the next SetInputTarget (no memset of the sums / counts then: 6.6 MB for a 66 k-point map);
  // the point count, which lsa_ndt_grid_dump reports, moves to its own array
  counts[r] = ni;
#pragma unroll
  for (int k = 0; k < 10; k++) a[k] = 0;
  cnt[r] = 0;
}

// {occupancy, valid, prefix} per 32 cells in one 16-byte word; the index's own bitmap word is cleared on the way (the
// packed copy is what every later reader uses), which completes the clean-up started in k_ndt_finalize
__global__ __launch_bounds__(kBlock) void k_pack_words(uint32_t *__restrict__ bits, const uint32_t *__restrict__ vbits,
                                                       const uint32_t *__restrict__ prefix, size_t W, uint4 *__restrict__ words) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W) return;
  words[i] = make_uint4(bits[i], vbits[i], prefix[i], 0u);
  bits[i] = 0u;
}

// dense probe table of the pass kernel: one 16-B entry per CELL of the box (not per occupied cell): the centroid relative
// to the cell's low corner in f32 and the record rank; cells without a usable voxel keep the 0xFF fill (rank -1)
__global__ __launch_bounds__(kBlock) void k_dense_fill(const double *__restrict__ rec, const int32_t *__restrict__ state, const int32_t *__restrict__ cell_of_rank,
                                                       size_t nocc, CellMap m, float4 *__restrict__ dense) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nocc || state[r] != 1) return;
  int cell = cell_of_rank[r];
  int ijk[3];
  lsa_cell_unravel(m, cell, ijk);
  const double *R = rec + r * kRecDoubles;
  // the corner exactly as the pass kernel forms it: (float)(absolute cell index) * res, in f32
  float cx = (float)ijk[0] * m.leaf[0], cy = (float)ijk[1] * m.leaf[1], cz = (float)ijk[2] * m.leaf[2];
  dense[cell] = make_float4((float)(R[0] - (double)cx), (float)(R[1] - (double)cy), (float)(R[2] - (double)cz), __int_as_float((int)r));
}

__global__ __launch_bounds__(kBlock) void k_count_valid(const int32_t *__restrict__ state, size_t nocc, unsigned *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned v = (i < nocc && state[i] == 1) ? 1u : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}
}  // namespace

int ndt_count_valid(lsa_ndt *h) {
  if (h->n_valid_known) return LSA_OK;
  lsa_context *ctx = h->ctx;
  const size_t V = h->vi.nocc;
  int rc;
  if ((rc = h->d_small.reserve(ctx, 64)) || (rc = h->h_out.reserve(ctx, kOutDoubles))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->d_small.p, 0, 64 * sizeof(double), ctx->stream));
  if (V) hipLaunchKernelGGL(k_count_valid, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, h->state.p, V, (unsigned *)h->d_small.p);
  LSA_HIP(ctx, hipGetLastError());
  LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, h->d_small.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  unsigned nvalid = 0;
  memcpy(&nvalid, h->h_out.p, sizeof(unsigned));
  h->n_valid = nvalid;
  h->n_valid_known = true;
  return LSA_OK;
}

int ndt_build_grid(lsa_ndt *h, const float4 *d_pts, size_t n) {
  lsa_context *ctx = h->ctx;
  VoxelIndex &vi = h->vi;
  h->has_target = false;
  int rc, status = VI_OK;
  const float leaf[3] = {h->resolution, h->resolution, h->resolution};
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = vi_enqueue(ctx, vi, d_pts, n, /*mode NDT*/ 1, leaf, 10))) return rc;
    if ((rc = vi_fetch(ctx, vi))) return rc;
    if ((rc = vi_wait(ctx, vi, &status))) return rc;
    if (status != VI_CAPACITY) break;
    if ((rc = vi_grow(ctx, vi))) return rc;
  }
  if (status == VI_EMPTY) return lsa_fail(ctx, LSA_ERR_INVALID, "set_input_target: no finite points");
  if (status != VI_OK) return lsa_fail(ctx, LSA_ERR_RANGE, "set_input_target: voxel grid of %d x %d x %d cells overflows int32", vi.map.dim[0], vi.map.dim[1], vi.map.dim[2]);
  size_t V = vi.nocc;
  if ((rc = h->rec.reserve(ctx, V * kRecDoubles + kRecDoubles, 1.25))) return rc;
  if ((rc = h->state.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = h->vbits.reserve(ctx, vi.W, 1.25))) return rc;
  if ((rc = h->words.reserve(ctx, vi.W, 1.25))) return rc;
  if ((rc = h->d_small.reserve(ctx, 64))) return rc;
  if ((rc = h->counts.reserve(ctx, V + 1, 1.25))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(h->vbits.p, 0, vi.W * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(k_ndt_finalize, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, vi.acc.p, vi.cnt.p, vi.cell_of_rank.p, V, vi.map, vi.fs,
                     h->rec.p, h->state.p, h->counts.p, h->vbits.p, h->mode == LSA_NDT_MODE_PCL ? 1 : 0);
  hipLaunchKernelGGL(k_pack_words, dim3(lsa_div_up(vi.W, kBlock)), dim3(kBlock), 0, ctx->stream, vi.bits.p, h->vbits.p, vi.prefix.p, vi.W, h->words.p);
  vi.clean = true;  // sums, counts and bitmap words of this build are zero again
  // dense probe table when the cell box is small enough (16 B per cell; default limit 16 Mi cells = 256 MiB)
  h->use_dense = vi.map.ncells <= h->dense_max_cells;
  if (h->use_dense) {
    size_t nc = (size_t)vi.map.ncells + 2;  // a row load reads 3 entries from any in-box cell
    if ((rc = h->dense.reserve(ctx, nc, 1.1))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(h->dense.p, 0xFF, nc * sizeof(float4), ctx->stream));
    hipLaunchKernelGGL(k_dense_fill, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, h->rec.p, h->state.p, vi.cell_of_rank.p, V, vi.map, h->dense.p);
  }
  LSA_HIP(ctx, hipGetLastError());
  // Nothing after the bounding-box / count round trip above needs the host, so the call returns with the table still being
  // finished on the stream (everything that uses it is ordered behind); the number of valid voxels is counted only when
  // lsa_ndt_grid_info asks for it.  The input points are no longer read at this point.
  h->n_valid_known = false;
  h->n_target = n;
  h->d_target_pts = d_pts;
  ndt_fitness_invalidate(h);
  h->nn.built = false;
  h->has_target = true;
  return LSA_OK;
}
