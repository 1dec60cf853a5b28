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

#include "ndt_cell_dev.h"
#include "ndt_internal.h"

namespace {
constexpr int kBlock = 256;

// (the per-cell arithmetic — eigen-solver with the reference's quirks, eigenvalue floor, inverse — is ndt_cell_dev.h)

// one thread per occupied cell (computeCentroidAndCovariance, VoxelGrid.cpp:273-323)
__global__ __launch_bounds__(kBlock) void k_ndt_finalize(long long *__restrict__ acc, int32_t *__restrict__ cnt,
                                                         const int32_t *__restrict__ cell_of_rank, size_t nocc, CellMap m, FixScales fs,
                                                         double *__restrict__ rec, int32_t *__restrict__ state, int32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ vbits, int pcl) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nocc) return;
  long long *a = acc + r * vi_row_stride(10);   // ten sums and, behind them, the count (voxel_index.h)
  int ni = (int)a[10];
  int cell = cell_of_rank[r];
  double o[3];
  lsa_cell_origin(m, cell, o);
  long long av[10];
#pragma unroll
  for (int k = 0; k < 10; k++) av[k] = a[k];
  const int st = ndt_cell_record(av, ni, o, fs, pcl, rec + r * kRecDoubles);
  state[r] = st;
  if (st == 1) atomicOr(vbits + (cell >> 5), 1u << (cell & 31));
  // leave the index clean for the next SetInputTarget (no memset of the sums / counts then: 6.6 MB for a 66 k-point map);
  // the point count, which lsa_ndt_grid_dump reports, moves to its own array
  counts[r] = ni;
#pragma unroll
  for (int k = 0; k <= 10; k++) a[k] = 0;
}

// {occupancy, valid, prefix} per 32 cells in one 16-byte word; the index's own bitmap word is cleared on the way (the
// packed copy is what every later reader uses), which completes the clean-up started in k_ndt_finalize
__global__ __launch_bounds__(kBlock) void k_pack_words(uint32_t *__restrict__ bits, uint32_t *__restrict__ vbits,
                                                       const uint32_t *__restrict__ prefix, size_t W, uint4 *__restrict__ words) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W) return;
  words[i] = make_uint4(bits[i], vbits[i], prefix[i], 0u);
  bits[i] = 0u;
  vbits[i] = 0u;   // (the valid bitmap is zero again, too: the next build needs no fill launch in front of k_ndt_finalize)
}

// dense probe table of the pass kernel in ONE sweep over the cells of the box: the entry of a valid voxel (its rank among the occupied
// cells from the packed words: prefix + occupied bits below), all-ones (rank -1) for every other cell and for the two pad entries a
// row load may touch.  (Was a 0xFF fill of the whole table and a scatter over the occupied cells behind it: one launch more.)
// PACK: the sweep also does k_pack_words' work — a word's 32 cells sit in 32 adjacent lanes of ONE wave (a wave takes 64 consecutive
// cells per step), which read the three bitmaps' words and then, by the lane of the word's first cell, write the packed word and
// clear the two bitmaps: a third launch less behind the scan.  (Every word has a cell below ncells, so every word is packed.)
template <bool PACK>
__global__ __launch_bounds__(kBlock) void k_dense_build(uint4 *__restrict__ words, const double *__restrict__ rec, CellMap m, size_t ncells, size_t nc,
                                                        float4 *__restrict__ dense, uint32_t *__restrict__ bits, uint32_t *__restrict__ vbits,
                                                        const uint32_t *__restrict__ prefix) {
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += (size_t)gridDim.x * blockDim.x) {
    const unsigned all = 0xFFFFFFFFu;
    float4 e = make_float4(__uint_as_float(all), __uint_as_float(all), __uint_as_float(all), __uint_as_float(all));
    if (c < ncells) {
      uint4 w;
      if (PACK) {
        const size_t i = c >> 5;
        w = make_uint4(bits[i], vbits[i], prefix[i], 0u);
        // (all 32 lanes of the word have issued their reads in this very instruction; the stores below come later in program order of
        // the same wave)
        if ((c & 31) == 0) { words[i] = w; bits[i] = 0u; vbits[i] = 0u; }
      } else {
        w = words[c >> 5];
      }
      const unsigned bit = (unsigned)(c & 31);
      if ((w.y >> bit) & 1u) {   // valid (k_ndt_finalize: state == 1)
        const unsigned r = w.z + (unsigned)__popc(w.x & ((1u << bit) - 1u));
        int ijk[3];
        lsa_cell_unravel(m, (int)c, ijk);
        e = ndt_dense_entry(rec + (size_t)r * kRecDoubles, ijk, m.leaf, (int)r);
      }
    }
    dense[c] = e;
  }
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

void ndt_set_scale_floor(lsa_ndt *h, unsigned long long n) { h->vi.scale_n_min = n; }

int ndt_adopt_external(lsa_ndt *h, const double *rec, const float4 *dense, const int min_b[3], const int dim[3], size_t n_target,
                       int (*materialize)(void *), void *user) {
  h->rec_ext = rec; h->dense_ext = dense;
  h->ext_materialize = materialize; h->ext_user = user;
  CellMap &m = h->vi.map;
  m.mode = 1;
  m.ncells = 1;
  for (int a = 0; a < 3; a++) {
    m.leaf[a] = h->resolution; m.inv[a] = 1.0f / h->resolution;
    m.min_b[a] = min_b[a]; m.dim[a] = dim[a];
    m.ncells *= dim[a];
  }
  h->vi.nocc = 0;
  h->use_dense = true;
  h->d_target_pts = nullptr;
  ndt_touch_external(h, n_target);
  h->has_target = true;
  return LSA_OK;
}

void ndt_touch_external(lsa_ndt *h, size_t n_target) {
  h->target_gen++;            // an armed kernel parked for the previous contents is not reused
  h->n_valid_known = false;
  h->n_target = n_target;
  ndt_fitness_invalidate(h);
  h->nn.built = false;
}

int ndt_materialize(lsa_ndt *h) {
  if (!h->rec_ext) return LSA_OK;
  if (!h->ext_materialize) return lsa_fail(h->ctx, LSA_ERR_INVALID, "NDT handle with an external table and no owner to materialise it");
  int (*fn)(void *) = h->ext_materialize;
  void *user = h->ext_user;
  const bool had_result = h->has_result;
  int rc = fn(user);   // runs ndt_build_grid on this handle, which leaves the external mode
  if (rc) return rc;
  h->has_result = had_result;   // (the table is the same one, cell for cell: the last ScanMatch's pose still belongs to it)
  return h->rec_ext ? lsa_fail(h->ctx, LSA_ERR_INVALID, "the owner of the external NDT table did not materialise it") : LSA_OK;
}

namespace {
// hipMemsetAsync costs the host 6-14 us per call on this runtime (it picks a fill kernel, splits unaligned sizes in two launches); a
// plain launch costs 4-5.  n16 = number of 16-byte units.
__global__ __launch_bounds__(256) void k_fill16(uint4 *__restrict__ p, size_t n16, unsigned v) {
  const uint4 q = make_uint4(v, v, v, v);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = q;
}
void fill16(lsa_context *ctx, void *p, size_t bytes, unsigned v) {   // bytes rounded up to 16: every buffer filled here has the room (DevBuf headroom / +2 pad)
  const size_t n16 = (bytes + 15) / 16;
  if (!n16) return;
  size_t g = (n16 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_fill16, dim3((unsigned)g), dim3(256), 0, ctx->stream, reinterpret_cast<uint4 *>(p), n16, v);
}
}  // namespace

int ndt_build_grid(lsa_ndt *h, const float4 *d_pts, size_t n) {
  lsa_context *ctx = h->ctx;
  VoxelIndex &vi = h->vi;
  h->has_target = false;
  h->rec_ext = nullptr; h->dense_ext = nullptr;   // an ordinary table from here on
  int rc, status = VI_OK;
  const float leaf[3] = {h->resolution, h->resolution, h->resolution};
  vi.mirror_at_scan = true;   // the definition and the occupied-cell count reach the host behind k_scan (no copy command, k_accum still running)
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = vi_enqueue(ctx, vi, d_pts, n, /*mode NDT*/ 1, leaf, 10))) return rc;
    if ((rc = vi_wait(ctx, vi, &status))) return rc;
    if (status != VI_CAPACITY) break;
    if ((rc = vi_grow(ctx, vi))) return rc;
  }
  if (status == VI_EMPTY) return lsa_fail(ctx, LSA_ERR_INVALID, "set_input_target: no finite points");
  if (status != VI_OK) return lsa_fail(ctx, LSA_ERR_RANGE, "set_input_target: voxel grid of %d x %d x %d cells overflows int32", vi.map.dim[0], vi.map.dim[1], vi.map.dim[2]);
  size_t V = vi.nocc;
  if ((rc = h->rec.reserve(ctx, V * kRecDoubles + kRecDoubles, 1.25))) return rc;
  if ((rc = h->state.reserve(ctx, V + 1, 1.25))) return rc;
  if ((rc = h->vbits.reserve(ctx, vi.W + 4, 1.25))) return rc;
  if ((rc = h->words.reserve(ctx, vi.W, 1.25))) return rc;
  if ((rc = h->d_small.reserve(ctx, 64))) return rc;
  if ((rc = h->counts.reserve(ctx, V + 1, 1.25))) return rc;
  // the valid bitmap: zeroed once when it is (re)allocated — over its whole capacity —, left zero by every build's k_pack_words
  if (!h->vbits_zero || h->vbits.cap != h->vbits_zero_cap) {
    fill16(ctx, h->vbits.p, (h->vbits.cap & ~(size_t)3) * sizeof(uint32_t), 0u);   // (cap >= W + 4: every word a build touches lies below cap & ~3)
    h->vbits_zero_cap = h->vbits.cap;
  }
  h->vbits_zero = false;   // (until this build's k_pack_words is enqueued)
  hipLaunchKernelGGL(k_ndt_finalize, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, vi.acc.p, vi.cnt.p, vi.cell_of_rank.p, V, vi.map, vi.fs,
                     h->rec.p, h->state.p, h->counts.p, h->vbits.p, h->mode == LSA_NDT_MODE_PCL ? 1 : 0);
  // dense probe table when the cell box is small enough (16 B per cell; default limit 16 Mi cells = 256 MiB); its sweep over the cells
  // packs the words on the way, a box too large for it gets the pack kernel alone
  h->use_dense = vi.map.ncells <= h->dense_max_cells;
  if (h->use_dense) {
    size_t nc = (size_t)vi.map.ncells + 2;  // a row load reads 3 entries from any in-box cell
    if ((rc = h->dense.reserve(ctx, nc, 1.1))) return rc;
    size_t g = (nc + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_dense_build<true>, dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, h->words.p, h->rec.p, vi.map, (size_t)vi.map.ncells, nc, h->dense.p,
                       vi.bits.p, h->vbits.p, vi.prefix.p);
  } else {
    hipLaunchKernelGGL(k_pack_words, dim3(lsa_div_up(vi.W, kBlock)), dim3(kBlock), 0, ctx->stream, vi.bits.p, h->vbits.p, vi.prefix.p, vi.W, h->words.p);
  }
  if (hipGetLastError() == hipSuccess) h->vbits_zero = true;
  vi.clean = true;  // sums, counts and bitmap words of this build are zero again
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
