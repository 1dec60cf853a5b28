// ndt_fitness.hip — GetFitnessScore for the NDT plugin.
//
// Replaces NormalDistributionsTransform::getFitnessScore (reference: lidar_localization/src/models/registration/
// ndt_registration_manual/NormalDistributionsTransform.cpp:940-965) and what it reaches:
//   VoxelGrid::nearestNeighborDistance / nearestVoxel   VoxelGrid.cpp:483-543
//   Octree::setInput / nearestOctreeNode                 Octree.cpp:50-228, :638-667 (+ initRange / goUp / goDown)
// Semantics: for every source point transformed by the final pose, (1) find the occupied octree LEAF — a block of
// 4 x 4 x 2 voxels (Octree.cpp:20-22) — whose point centroid is nearest (the reference's branch-and-bound prunes on a
// valid lower bound, so it returns that global argmin), (2) take the leaf's POINT bounding box, (3) among the voxels
// whose index lies in floor(box / leaf) and that hold >= 1 point, take the nearest centroid, (4) average the distances
// (not squared) over the source points.  max_range is DBL_MAX, so every point counts.
//
// Device layout: dense block table over the occupied voxel bounds divided by (4,4,2): {fixed-point point sums, count,
// float min/max} filled by one streaming pass over the target with atomics, compacted to a list of occupied blocks;
// the query is a brute-force argmin over that list (a few hundred to a few thousand entries, staged through LDS).
// Built lazily on the first GetFitnessScore after a SetInputTarget: only loop closing calls it.
#include "ndt_internal.h"

#include <cfloat>

namespace {
constexpr int kBlock = 256;

struct BlockGrid {
  int lo[3], dim[3];
  float leaf;
  double s1, inv_s1;
};

__device__ __forceinline__ int fdiv_floor(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }  // Octree.cpp:44-47

// order-preserving float <-> int maps for atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return (i >= 0) ? i : (i ^ 0x7fffffff); }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float((i >= 0) ? i : (i ^ 0x7fffffff)); }

__global__ __launch_bounds__(kBlock) void k_blocks_init(int *__restrict__ mm, size_t nblk) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  int *m = mm + i * 6;
  m[0] = m[1] = m[2] = INT_MAX;
  m[3] = m[4] = m[5] = INT_MIN;
}

__global__ __launch_bounds__(kBlock) void k_blocks_accum(const float4 *__restrict__ pts, size_t n, BlockGrid g, long long *__restrict__ sums,
                                                         int *__restrict__ cnt, int *__restrict__ mm) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (!lsa_finite3(p.x, p.y, p.z)) continue;
    // voxel ids as VoxelGrid::setInput computes them (VoxelGrid.cpp:344-346), then the leaf that holds them (Octree.cpp:150-152)
    int vx = (int)floorf(lsa_fdiv(p.x, g.leaf)), vy = (int)floorf(lsa_fdiv(p.y, g.leaf)), vz = (int)floorf(lsa_fdiv(p.z, g.leaf));
    int bx = fdiv_floor(vx, 4) - g.lo[0], by = fdiv_floor(vy, 4) - g.lo[1], bz = fdiv_floor(vz, 2) - g.lo[2];
    if ((unsigned)bx >= (unsigned)g.dim[0] || (unsigned)by >= (unsigned)g.dim[1] || (unsigned)bz >= (unsigned)g.dim[2]) continue;
    size_t b = (size_t)bx + (size_t)by * g.dim[0] + (size_t)bz * g.dim[0] * g.dim[1];
    double ox = (double)((bx + g.lo[0]) * 4) * (double)g.leaf, oy = (double)((by + g.lo[1]) * 4) * (double)g.leaf, oz = (double)((bz + g.lo[2]) * 2) * (double)g.leaf;
    unsigned long long *s = (unsigned long long *)(sums + b * 3);
    atomicAdd(s + 0, (unsigned long long)__double2ll_rn(((double)p.x - ox) * g.s1));
    atomicAdd(s + 1, (unsigned long long)__double2ll_rn(((double)p.y - oy) * g.s1));
    atomicAdd(s + 2, (unsigned long long)__double2ll_rn(((double)p.z - oz) * g.s1));
    atomicAdd(cnt + b, 1);
    int *m = mm + b * 6;
    atomicMin(m + 0, f2ord(p.x)); atomicMin(m + 1, f2ord(p.y)); atomicMin(m + 2, f2ord(p.z));
    atomicMax(m + 3, f2ord(p.x)); atomicMax(m + 4, f2ord(p.y)); atomicMax(m + 5, f2ord(p.z));
  }
}

struct LeafRec {
  double c[3];   // point centroid of the leaf
  float box[6];  // point bounding box lx,ly,lz,ux,uy,uz
  int id;        // dense block id (tie-break: lowest id, like ascending node order)
  int pad;
};

__global__ __launch_bounds__(kBlock) void k_blocks_compact(const long long *__restrict__ sums, const int *__restrict__ cnt, const int *__restrict__ mm,
                                                           size_t nblk, BlockGrid g, LeafRec *__restrict__ out, unsigned *__restrict__ n_out) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk || cnt[b] <= 0) return;
  unsigned slot = atomicAdd(n_out, 1u);
  int bz = (int)(b / ((size_t)g.dim[0] * g.dim[1]));
  int rem = (int)(b - (size_t)bz * g.dim[0] * g.dim[1]);
  int by = rem / g.dim[0], bx = rem - by * g.dim[0];
  double n = (double)cnt[b];
  LeafRec r;
  r.c[0] = (double)((bx + g.lo[0]) * 4) * (double)g.leaf + ((double)sums[b * 3 + 0] * g.inv_s1) / n;
  r.c[1] = (double)((by + g.lo[1]) * 4) * (double)g.leaf + ((double)sums[b * 3 + 1] * g.inv_s1) / n;
  r.c[2] = (double)((bz + g.lo[2]) * 2) * (double)g.leaf + ((double)sums[b * 3 + 2] * g.inv_s1) / n;
  for (int a = 0; a < 6; a++) r.box[a] = ord2f(mm[b * 6 + a]);
  r.id = (int)b;
  r.pad = 0;
  out[slot] = r;
}

struct T12s { float v[12]; };

// kFitLanes adjacent lanes share one source point: each takes every kFitLanes-th leaf of the brute-force argmin and every
// kFitLanes-th voxel of the winning leaf's box, and the partial results are merged with the reference's tie rules (the
// comparison is on the ROUNDED f64 distance, lower leaf id on equality, so the f64 sqrt cannot be dropped — but it is only
// taken for leaves whose squared distance does not already rule them out).  One lane per point walked ~1500 leaves and
// up to 32 dependent bitmap -> prefix -> record chains alone: 182 us for 7.7 k points on 30 workgroups.
constexpr int kFitLanes = 8;
__global__ __launch_bounds__(kBlock) void k_fitness(const float4 *__restrict__ src, size_t n, T12s Tm, const LeafRec *__restrict__ leaves,
                                                    unsigned nleaves, CellMap m, const uint4 *__restrict__ words,
                                                    const double *__restrict__ rec, double *__restrict__ partials) {
  __shared__ LeafRec tile[kBlock];
  __shared__ double sm[kBlock / 64][2];
  const int sub = threadIdx.x & (kFitLanes - 1);
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / kFitLanes;
  bool live = i < n;
  float qx = 0, qy = 0, qz = 0;
  const float *T12 = Tm.v;
  if (live) {
    float4 s = src[i];
    qx = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T12[0], s.x), lsa_fmul(T12[1], s.y)), lsa_fmul(T12[2], s.z)), T12[3]);
    qy = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T12[4], s.x), lsa_fmul(T12[5], s.y)), lsa_fmul(T12[6], s.z)), T12[7]);
    qz = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(T12[8], s.x), lsa_fmul(T12[9], s.y)), lsa_fmul(T12[10], s.z)), T12[11]);
    live = lsa_finite3(qx, qy, qz);
  }
  double best = DBL_MAX, best2_hi = DBL_MAX;  // best2_hi: a squared distance above it cannot reach `best` after the sqrt
  int best_id = INT_MAX;
  float bbox[6] = {0, 0, 0, 0, 0, 0};
  for (unsigned base = 0; base < nleaves; base += kBlock) {
    __syncthreads();
    if (base + threadIdx.x < nleaves) tile[threadIdx.x] = leaves[base + threadIdx.x];
    __syncthreads();
    unsigned cntl = min((unsigned)kBlock, nleaves - base);
    if (live) {
      for (unsigned k = sub; k < cntl; k += kFitLanes) {
        const LeafRec &L = tile[k];
        double dx = L.c[0] - (double)qx, dy = L.c[1] - (double)qy, dz = L.c[2] - (double)qz;
        double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 > best2_hi) continue;
        double d = sqrt(d2);
        if (d < best || (d == best && L.id < best_id)) {
          best = d; best_id = L.id;
          best2_hi = d * d * (1.0 + 1e-14);
#pragma unroll
          for (int a = 0; a < 6; a++) bbox[a] = L.box[a];
        }
      }
    }
  }
  // merge the kFitLanes partial winners (all lanes of a group end with the group's)
#pragma unroll
  for (int off = 1; off < kFitLanes; off <<= 1) {
    double ob = __shfl_xor(best, off, 64);
    int oid = __shfl_xor(best_id, off, 64);
    float obox[6];
#pragma unroll
    for (int a = 0; a < 6; a++) obox[a] = __shfl_xor(bbox[a], off, 64);
    if (ob < best || (ob == best && oid < best_id)) {
      best = ob; best_id = oid;
#pragma unroll
      for (int a = 0; a < 6; a++) bbox[a] = obox[a];
    }
  }
  double md = DBL_MAX;
  if (live && best_id != INT_MAX) {
    // nearestVoxel (VoxelGrid.cpp:483-521): voxels floor(lower / leaf) .. floor(upper / leaf) holding >= 1 point
    int lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      lo[a] = (int)floorf(lsa_fdiv(bbox[a], m.leaf[a])) - m.min_b[a];
      hi[a] = (int)floorf(lsa_fdiv(bbox[3 + a], m.leaf[a])) - m.min_b[a];
      lo[a] = max(lo[a], 0); hi[a] = min(hi[a], m.dim[a] - 1);
    }
    const int nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
    if (nx > 0 && ny > 0 && nz > 0) {
      const long long total = (long long)nx * ny * nz;
      for (long long c = sub; c < total; c += kFitLanes) {
        int iz = (int)(c % nz), iy = (int)((c / nz) % ny), ix = (int)(c / ((long long)nz * ny));
        int cell = (lo[0] + ix) + (lo[1] + iy) * m.dim[0] + (lo[2] + iz) * m.dim[0] * m.dim[1];
        const uint4 wd = words[cell >> 5];  // {occupancy, valid, prefix}: the table's packed copy of the bitmap
        const uint32_t w = wd.x, bit = 1u << (cell & 31);
        if (!(w & bit)) continue;
        unsigned rank = wd.z + __popc(w & (bit - 1u));
        const double *R = rec + (size_t)rank * kRecDoubles;
        double dx = (double)qx - R[0], dy = (double)qy - R[1], dz = (double)qz - R[2];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < md) md = d;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < kFitLanes; off <<= 1) md = fmin(md, __shfl_xor(md, off, 64));
  double dist = 0.0, cnt = 0.0;
  if (sub == 0 && live && best_id != INT_MAX && md < DBL_MAX) { dist = md; cnt = 1.0; }  // distance < max_range (= DBL_MAX) (:954-957)
  dist = lsa_wave_sum(dist);
  cnt = lsa_wave_sum(cnt);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sm[wave][0] = dist; sm[wave][1] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < kBlock / 64; w++) { a += sm[w][0]; b += sm[w][1]; }
    partials[(size_t)blockIdx.x * 2] = a;
    partials[(size_t)blockIdx.x * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(kBlock) void k_fitness_final(const double *__restrict__ partials, int nb, double *__restrict__ out) {
  __shared__ double s[kBlock][2];
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < nb; i += kBlock) { a += partials[2 * (size_t)i]; b += partials[2 * (size_t)i + 1]; }
  s[threadIdx.x][0] = a; s[threadIdx.x][1] = b;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) { s[threadIdx.x][0] += s[threadIdx.x + st][0]; s[threadIdx.x][1] += s[threadIdx.x + st][1]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s[0][0]; out[1] = s[0][1]; }
}

int ceil_log2_host(double v) {
  int e;
  double f = frexp(v, &e);
  return (f == 0.5) ? e - 1 : e;
}
}  // namespace

struct FitnessState {
  DevBuf<long long> sums;
  DevBuf<int> cnt, mm;
  DevBuf<LeafRec> leaves;
  DevBuf<double> partials, small;
  unsigned nleaves = 0;
  bool built = false;
};

void ndt_fitness_invalidate(lsa_ndt *h) {
  if (h->fitness) h->fitness->built = false;
}

void ndt_fitness_release(lsa_ndt *h) {
  if (!h->fitness) return;
  FitnessState *f = h->fitness;
  f->sums.release(); f->cnt.release(); f->mm.release(); f->leaves.release(); f->partials.release(); f->small.release();
  delete f;
  h->fitness = nullptr;
}

static int fitness_build(lsa_ndt *h) {
  lsa_context *ctx = h->ctx;
  if (!h->fitness) h->fitness = new FitnessState();
  FitnessState *f = h->fitness;
  if (!h->d_target_pts || !h->n_target) return lsa_fail(ctx, LSA_ERR_NO_TARGET, "GetFitnessScore: target points are not available");
  const CellMap &m = h->vi.map;
  BlockGrid g;
  const int L[3] = {4, 4, 2};
  for (int a = 0; a < 3; a++) {
    int lo = m.min_b[a], hi = m.min_b[a] + m.dim[a] - 1;
    auto fd = [](int v, int d) { return (v >= 0) ? v / d : -((-v + d - 1) / d); };
    g.lo[a] = fd(lo, L[a]);
    g.dim[a] = fd(hi, L[a]) - g.lo[a] + 1;
  }
  g.leaf = h->resolution;
  int bits_n = ceil_log2_host((double)h->n_target + 1.0);
  int bits_d = ceil_log2_host(2.0 * 4.0 * (double)h->resolution);
  int k1 = 62 - bits_n - bits_d;
  if (k1 > 44) k1 = 44;
  g.s1 = ldexp(1.0, k1); g.inv_s1 = ldexp(1.0, -k1);
  size_t nblk = (size_t)g.dim[0] * g.dim[1] * g.dim[2];
  int rc;
  if ((rc = f->sums.reserve(ctx, nblk * 3, 1.25))) return rc;
  if ((rc = f->cnt.reserve(ctx, nblk, 1.25))) return rc;
  if ((rc = f->mm.reserve(ctx, nblk * 6, 1.25))) return rc;
  if ((rc = f->leaves.reserve(ctx, nblk, 1.25))) return rc;
  if ((rc = f->small.reserve(ctx, 8))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(f->sums.p, 0, nblk * 3 * sizeof(long long), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(f->cnt.p, 0, nblk * sizeof(int), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(f->small.p, 0, 8 * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_blocks_init, dim3(lsa_div_up(nblk, kBlock)), dim3(kBlock), 0, ctx->stream, f->mm.p, nblk);
  size_t n = h->n_target;
  unsigned gsz = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)ctx->num_cus * 8);
  hipLaunchKernelGGL(k_blocks_accum, dim3(gsz ? gsz : 1), dim3(kBlock), 0, ctx->stream, h->d_target_pts, n, g, f->sums.p, f->cnt.p, f->mm.p);
  hipLaunchKernelGGL(k_blocks_compact, dim3(lsa_div_up(nblk, kBlock)), dim3(kBlock), 0, ctx->stream, f->sums.p, f->cnt.p, f->mm.p, nblk, g, f->leaves.p,
                     (unsigned *)f->small.p);
  LSA_HIP(ctx, hipGetLastError());
  if ((rc = h->h_out.reserve(ctx, kOutDoubles))) return rc;
  LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, f->small.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(&f->nleaves, h->h_out.p, sizeof(unsigned));
  f->built = true;
  return LSA_OK;
}

int ndt_fitness(lsa_ndt *h, float *score) {
  lsa_context *ctx = h->ctx;
  int rc;
  if (!h->fitness || !h->fitness->built) {
    if ((rc = fitness_build(h))) return rc;
  }
  FitnessState *f = h->fitness;
  size_t n = h->n_source;
  if (!h->allreduce && (n == 0 || !h->d_source || f->nleaves == 0)) {
    *score = FLT_MAX;  // the reference returns DBL_MAX when no point qualifies (:964)
    return LSA_OK;
  }
  if ((rc = h->h_out.reserve(ctx, kOutDoubles))) return rc;
  if (n == 0 || !h->d_source || f->nleaves == 0) {   // an empty shard of a sharded source still takes part in the collective
    LSA_HIP(ctx, hipMemsetAsync(f->small.p, 0, 2 * sizeof(double), ctx->stream));
    n = 0;
  }
  T12s Tm;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Tm.v[r * 4 + c] = h->last_T[c * 4 + r];
  unsigned nb = lsa_div_up(n * kFitLanes, kBlock);
  if ((rc = f->partials.reserve(ctx, (size_t)nb * 2 + 2, 1.25))) return rc;
  if (n) {
    hipLaunchKernelGGL(k_fitness, dim3(nb), dim3(kBlock), 0, ctx->stream, h->d_source, n, Tm, f->leaves.p, f->nleaves, h->vi.map, h->words.p,
                       h->rec.p, f->partials.p);
    hipLaunchKernelGGL(k_fitness_final, dim3(1), dim3(kBlock), 0, ctx->stream, f->partials.p, (int)nb, f->small.p);
  }
  LSA_HIP(ctx, hipGetLastError());
  if (h->allreduce) {
    // source sharded over the ranks ("one extra all-reduce of 2 scalars for fitness"): {sum of distances, points}
    int crc = h->allreduce(h->allreduce_user, f->small.p, 2, (void *)ctx->stream);
    if (crc) return lsa_fail(ctx, LSA_ERR_COLLECTIVE, "GetFitnessScore: all-reduce callback returned %d", crc);
  }
  LSA_HIP(ctx, hipMemcpyAsync(h->h_out.p, f->small.p, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double sum = h->h_out.p[0], cnt = h->h_out.p[1];
  *score = (cnt > 0) ? (float)(sum / cnt) : FLT_MAX;
  return LSA_OK;
}
