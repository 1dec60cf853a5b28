// voxel_index.hip — sparse voxel index: bbox reduce, grid definition on the device (or reused from the previous build),
// occupancy bitmap, single-pass popcount prefix scan, fixed-point per-cell accumulation.  HBM-bound streaming kernels over packed float4 points (16 B/point, one
// global_load_dwordx4 per lane); at KITTI size (131 k points = 2.1 MB = 0.3 us of HBM time) they are launch-latency
// bound, which is why the pipeline is enqueued back to back with no host round trip in between.
#include "voxel_index_dev.h"

#include <cfloat>
#include <cstdlib>
#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kScanTile = 2048;  // words per scan block: 256 threads x 8 words

__global__ __launch_bounds__(kBlock) void k_bbox_partial(const float4 *__restrict__ pts, size_t n, float *__restrict__ partial) {
  float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
  unsigned cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (lsa_finite3(p.x, p.y, p.z)) {
      mn0 = fminf(mn0, p.x); mn1 = fminf(mn1, p.y); mn2 = fminf(mn2, p.z);
      mx0 = fmaxf(mx0, p.x); mx1 = fmaxf(mx1, p.y); mx2 = fmaxf(mx2, p.z);
      cnt++;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn0 = fminf(mn0, __shfl_down(mn0, off, 64)); mn1 = fminf(mn1, __shfl_down(mn1, off, 64)); mn2 = fminf(mn2, __shfl_down(mn2, off, 64));
    mx0 = fmaxf(mx0, __shfl_down(mx0, off, 64)); mx1 = fmaxf(mx1, __shfl_down(mx1, off, 64)); mx2 = fmaxf(mx2, __shfl_down(mx2, off, 64));
    cnt += __shfl_down(cnt, off, 64);
  }
  __shared__ float s[kBlock / 64][8];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    s[wave][0] = mn0; s[wave][1] = mn1; s[wave][2] = mn2; s[wave][3] = mx0; s[wave][4] = mx1; s[wave][5] = mx2;
    s[wave][6] = __uint_as_float(cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned c = 0;
    for (int w = 0; w < kBlock / 64; w++) {
      mn0 = fminf(mn0, s[w][0]); mn1 = fminf(mn1, s[w][1]); mn2 = fminf(mn2, s[w][2]);
      mx0 = fmaxf(mx0, s[w][3]); mx1 = fmaxf(mx1, s[w][4]); mx2 = fmaxf(mx2, s[w][5]);
      c += __float_as_uint(s[w][6]);
    }
    float *o = partial + (size_t)blockIdx.x * 8;
    o[0] = mn0; o[1] = mn1; o[2] = mn2; o[3] = mx0; o[4] = mx1; o[5] = mx2; o[6] = __uint_as_float(c); o[7] = 0.f;
  }
}


// single block: final bbox reduction, then the grid definition by the requested rule set, on the device
struct ReqParams {  // the request travels as a kernel argument: no H2D copy on the critical path
  int mode;
  float leaf[3];
  unsigned long long W_cap;
  int expand;  // define the cell box with a margin so that the next scans can reuse it (vi_enqueue allow_spec)
  unsigned long long scale_n_min;  // VoxelIndex::scale_n_min
};

__global__ __launch_bounds__(kBlock) void k_bbox_final_params(const float *__restrict__ partial, int nb, ReqParams req, DevParams *__restrict__ dp) {
  float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
  unsigned long long cnt = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    const float *p = partial + (size_t)i * 8;
    mn0 = fminf(mn0, p[0]); mn1 = fminf(mn1, p[1]); mn2 = fminf(mn2, p[2]);
    mx0 = fmaxf(mx0, p[3]); mx1 = fmaxf(mx1, p[4]); mx2 = fmaxf(mx2, p[5]);
    cnt += __float_as_uint(p[6]);
  }
  __shared__ float s[kBlock][6];
  __shared__ unsigned long long sc[kBlock];
  s[threadIdx.x][0] = mn0; s[threadIdx.x][1] = mn1; s[threadIdx.x][2] = mn2;
  s[threadIdx.x][3] = mx0; s[threadIdx.x][4] = mx1; s[threadIdx.x][5] = mx2;
  sc[threadIdx.x] = cnt;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      for (int a = 0; a < 3; a++) s[threadIdx.x][a] = fminf(s[threadIdx.x][a], s[threadIdx.x + st][a]);
      for (int a = 3; a < 6; a++) s[threadIdx.x][a] = fmaxf(s[threadIdx.x][a], s[threadIdx.x + st][a]);
      sc[threadIdx.x] += sc[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  float mn[3] = {s[0][0], s[0][1], s[0][2]}, mx[3] = {s[0][3], s[0][4], s[0][5]};
  unsigned long long nfin = sc[0];
  // the definition the device still holds from the previous build of this index (speculative builds of a sensor's scans:
  // the new box becomes the UNION with it, see below)
  const CellMap prev = dp->map;
  const bool prev_ok = req.expand && dp->expand && dp->status == VI_OK && dp->req_mode == req.mode && dp->req_leaf[0] == req.leaf[0] &&
                       dp->req_leaf[1] == req.leaf[1] && dp->req_leaf[2] == req.leaf[2] && prev.dim[0] > 0 && prev.dim[1] > 0 && prev.dim[2] > 0;
  CellMap m;
  m.mode = req.mode;
  dp->req_mode = req.mode;
  dp->W_cap = req.W_cap;
  int status = VI_OK;
  for (int a = 0; a < 3; a++) {
    dp->mn[a] = mn[a]; dp->mx[a] = mx[a];
    m.leaf[a] = req.leaf[a];
    dp->req_leaf[a] = req.leaf[a];
    m.inv[a] = 1.0f / m.leaf[a];
    m.min_b[a] = 0; m.dim[a] = 1;
  }
  m.ncells = 1;
  dp->n_finite = nfin;
  if (nfin == 0) {
    status = VI_EMPTY;
  } else if (m.mode == 0) {
    // PCL 1.7 filters/impl/voxel_grid.hpp applyFilter: overflow check on (max - min) * inverse_leaf, then
    // min_b / max_b = floor(min|max * inverse_leaf), div_b = max_b - min_b + 1.  (PCL multiplies the three extents in
    // int64, which itself overflows past ~9.2e18 cells; the product is compared in double so that case is caught too.)
    double prod = 1.0;
    for (int a = 0; a < 3; a++) prod *= (double)((long long)(lsa_fmul(lsa_fsub(mx[a], mn[a]), m.inv[a])) + 1);
    if (prod > (double)INT32_MAX) status = VI_RANGE;
    for (int a = 0; a < 3; a++) {
      m.min_b[a] = (int)floorf(lsa_fmul(mn[a], m.inv[a]));
      int max_b = (int)floorf(lsa_fmul(mx[a], m.inv[a]));
      m.dim[a] = max_b - m.min_b[a] + 1;
    }
    if (req.expand && status == VI_OK) {
      // margin of max(2 cells, 5 % of the extent) per side: neither the output order nor the values depend on min_b /
      // dims (the order is lexicographic in (k, j, i), the sums are taken against absolute cell corners), only the
      // overflow verdict above does, and that used the exact box.  Keep the exact box if the padded one would overflow.
      int mb[3], dm[3];
      double nc = 1.0;
      for (int a = 0; a < 3; a++) {
        int pad = max(2, m.dim[a] / 20);
        mb[a] = m.min_b[a] - pad; dm[a] = m.dim[a] + 2 * pad;
        nc *= (double)dm[a];
      }
      bool done = false;
      if (prev_ok) {
        // A scan that reaches further than its predecessor misses the reused box and is built twice; with scans of
        // alternating extent that happened on every other scan (the dense workload: the frame filter at 110 us instead of
        // 40).  In the sensor frame the extent is bounded by the range, so the box is made the UNION with the previous one:
        // it only grows, and after a few scans it covers what the sensor can see and speculative builds stop missing.  (Order
        // and values of the output do not depend on the box, only on floor(x * inv_leaf).)  Clouds that wander — key-frames
        // in map coordinates through one handle — must not drag an ever-growing box along: the union is dropped as soon as it
        // would hold more than four times the cells of the padded box of the cloud at hand.
        int ub[3], ud[3];
        double unc = 1.0;
        for (int a = 0; a < 3; a++) {
          int lo = min(mb[a], prev.min_b[a]), hi = max(mb[a] + dm[a], prev.min_b[a] + prev.dim[a]);
          ub[a] = lo; ud[a] = hi - lo;
          unc *= (double)ud[a];
        }
        if (unc <= 4.0 * nc && unc <= (double)INT32_MAX) {
          // (a union beyond the bitmap's capacity is reported as VI_CAPACITY below: the host sizes the bitmap for it once)
          for (int a = 0; a < 3; a++) { m.min_b[a] = ub[a]; m.dim[a] = ud[a]; }
          done = true;
        }
      }
      if (!done && nc <= (double)INT32_MAX && (unsigned long long)((nc + 31.0) / 32.0) <= req.W_cap)
        for (int a = 0; a < 3; a++) { m.min_b[a] = mb[a]; m.dim[a] = dm[a]; }
    }
  } else {
    // reference VoxelGrid::findBoundaries (VoxelGrid.cpp:366-372): real_min/real_max = floor(min|max / leaf).  The
    // reference additionally pads its DENSE allocation to multiples of 16x16x8 cells (:379-392); every lookup is clipped
    // to the real bounds (:449-455), so the padding is an allocation detail with no observable effect.
    for (int a = 0; a < 3; a++) {
      m.min_b[a] = (int)floorf(lsa_fdiv(mn[a], m.leaf[a]));
      int max_b = (int)floorf(lsa_fdiv(mx[a], m.leaf[a]));
      m.dim[a] = max_b - m.min_b[a] + 1;
    }
  }
  if (status == VI_OK) {
    double nc = (double)m.dim[0] * (double)m.dim[1] * (double)m.dim[2];
    if (m.dim[0] <= 0 || m.dim[1] <= 0 || m.dim[2] <= 0 || nc > (double)INT32_MAX) status = VI_RANGE;
    else m.ncells = (long long)m.dim[0] * m.dim[1] * m.dim[2];
  }
  unsigned long long W = (unsigned long long)((m.ncells + 31) / 32);
  if (status == VI_OK && W > req.W_cap) status = VI_CAPACITY;
  // fixed-point scales (vi_make_scales); with `expand` they leave room for twice the points, so that later (speculative)
  // builds can reuse them; a caller-set floor (scale_n_min) makes them a constant of the handle
  double leaf_max = fmax(fmax((double)m.leaf[0], (double)m.leaf[1]), (double)m.leaf[2]);
  unsigned long long n_cap = 0;
  FixScales fs = vi_make_scales(fmax((req.expand ? 2.0 : 1.0) * (double)nfin, (double)req.scale_n_min), leaf_max, &n_cap);
  dp->map = m;
  dp->fs = fs;
  dp->W = W;
  dp->nsb = (unsigned)((W + kScanTile - 1) / kScanTile);
  dp->V = 0;
  dp->status = status;
  dp->n_cap = n_cap;
  dp->spec_miss = 0;
  dp->expand = req.expand;
}

// ---- LDS tile hashing ---------------------------------------------------------------------------------------
// Consecutive lidar returns hit the same few hundred voxels, so a tile of kTile consecutive points is first
// de-duplicated / pre-aggregated in an LDS open-addressing table keyed by the cell index; only one global atomic per
// DISTINCT cell of the tile leaves the CU.  (Global atomics execute at the memory side at ~1e10/s and serialise on a
// shared word; LDS atomics are two orders of magnitude cheaper.)  A point that cannot be placed because the table is
// full falls back to direct global atomics, so correctness never depends on the tile's diversity.
// Tile geometry: points per workgroup TILE and LDS slots CAP (power of two, = TILE / 2).  Big tiles de-duplicate best
// (2 M-point maps); a KITTI scan (131 k points) cut into 2048-point tiles is only 64 workgroups on a 256-CU chip and each
// kernel is a latency chain, so small inputs use 512-point tiles: four times the workgroups, a quarter of the chain.
constexpr int kTileBig = 2048, kTileSmall = 512, kTilePull = 1024;   // (pull: four 16-byte link reads per lane in flight, 128 workgroups per 131 k-point scan: 48 GB/s measured)
constexpr size_t kSmallInput = 512 * 1024;  // points; below this the small tile is used

__device__ __forceinline__ unsigned lds_hash(int cell) { return ((unsigned)cell * 2654435761u) >> 22; }  // top 10 bits

// returns the slot of `cell` in keys[] (inserting it if new), or -1 if no slot was found within kMaxProbes steps.  The
// bound matters for inputs that are already sparse (a filtered map filtered again, the NDT table of a 0.3 m-filtered
// local map): nearly every point is its own cell, the table fills up, and unbounded probing walked all CAP slots per
// point (k_accum at 52 us for 66 k points); such tiles gain nothing from pre-aggregation and go straight to memory.
constexpr int kMaxProbes = 8;
template <int CAP>
__device__ __forceinline__ int lds_insert(int *keys, int cell) {
  unsigned h = lds_hash(cell) & (CAP - 1);
  for (int probe = 0; probe < kMaxProbes; probe++) {
    int old = keys[h];
    if (old == cell) return (int)h;
    if (old == -1) {
      old = atomicCAS(&keys[h], -1, cell);
      if (old == -1 || old == cell) return (int)h;
    }
    h = (h + 1) & (CAP - 1);
  }
  return -1;
}

// diagnostics (lsa_voxel_filter_stamps): thread 0 of a workgroup notes the 100 MHz wall clock at a phase boundary
__device__ __forceinline__ void vi_stamp(unsigned long long *stamps, int k) {
  if (stamps && threadIdx.x == 0 && blockIdx.x < kViStampWgs) stamps[(size_t)blockIdx.x * 8 + k] = wall_clock64();
}

// Which point a lane holds in its k-th register: wave w of a tile takes the tile's 64-point groups w * PER .. w * PER + PER - 1, its
// lane l the l-th point of each — every load instruction still reads 64 consecutive points (1 KB), and a lane's PER points are 64
// apart in the cloud.  A spinning lidar delivers its beams azimuth step by azimuth step, so points 64 (or 128) apart are neighbours
// on one ring: at 2 M points / 0.3 m a lane's eight points fall into 3.0 runs of equal cells on average (4.8 with the lanes 256
// apart, 5.1 for eight consecutive points), at 1.0 m into 2.0 — and a run costs the LDS one set of adds, not one per point
// (vi_stamps: the adds of eight points per lane were 10-12 us of a workgroup's 22, the LDS array busy throughout).  Any other
// order of the cloud only merges less; nothing but speed depends on it.
template <int PER, bool LINEAR = false>
__device__ __forceinline__ size_t vi_point_index(size_t base, int k) {
  // LINEAR: the workgroup's k-th 256 points.  The kernel that pulls its rows over the link keeps this order: with the wave-group order
  // the same bytes took 66 instead of 57 us to arrive (131 k rows; `profiles/r05_voxel_index_ab.log`) — a load step of the whole
  // workgroup then asks for four separate 1 KB pieces instead of one 4 KB piece
  if (LINEAR) return base + (size_t)k * kBlock + threadIdx.x;
  return base + (size_t)(((threadIdx.x >> 6) * PER + k) << 6) + (threadIdx.x & 63);
}

// The same for B cells of one lane at once: the B key reads are issued together, then the B compare-and-swaps of the lanes that met an
// empty slot, then the next probe step for whoever is still unplaced — a lane's cells walk their probe sequences side by side instead of one
// after the other.  (Phase stamps of the 2 M-point build, `profiles/r05_vi_stamps_before.log`: eight dependent insert + add chains per lane
// took 10.5 us of a workgroup's 22 in k_accum and 5.4 of 13 in k_mark, whether 768 workgroups were resident or 209: latency, not
// throughput.)  cell < 0: nothing to place.  A lane's own duplicates resolve like anybody else's: LDS operations of a wave execute in order,
// the second compare-and-swap finds the first one's key.  Which slot a cell lands in may differ from the one-by-one walk; nothing depends on it.
template <int CAP, int B>
__device__ __forceinline__ void lds_insert_batch(int *keys, const int (&cell)[B], int (&slot)[B]) {
  unsigned h[B];
  unsigned pending = 0;
#pragma unroll
  for (int k = 0; k < B; k++) {
    h[k] = lds_hash(cell[k]) & (CAP - 1);
    slot[k] = -1;
    if (cell[k] >= 0) pending |= 1u << k;
  }
  for (int probe = 0; probe < kMaxProbes; probe++) {
    if (!__any(pending != 0)) break;
    int old[B];
#pragma unroll
    for (int k = 0; k < B; k++) old[k] = (pending >> k) & 1u ? keys[h[k]] : 0;
#pragma unroll
    for (int k = 0; k < B; k++)
      if (((pending >> k) & 1u) && old[k] == -1) old[k] = atomicCAS(&keys[h[k]], -1, cell[k]);
#pragma unroll
    for (int k = 0; k < B; k++) {
      if (!((pending >> k) & 1u)) continue;
      if (old[k] == -1 || old[k] == cell[k]) { slot[k] = (int)h[k]; pending &= ~(1u << k); }
      else h[k] = (h[k] + 1) & (CAP - 1);
    }
  }
}

// PULL (host clouds, lsa_voxel_filter_apply): the rows are not resident yet when this kernel starts.  The host packs them into
// device-visible pinned memory chunk by chunk and sets ready[chunk] = seq behind each; a tile waits for its chunk, reads its rows
// over the link itself (no copy command: four 0.5 MB copies cost 73 us on one queue against 44 us for the bytes), leaves them in
// device memory for k_accum and marks them in the same pass — the transfer runs under the packing and the mark under the transfer.
// Only with a known grid definition (kp.valid: the speculative box of the previous scan), so nothing has to see all rows first.
template <int TILE, bool PULL>
__global__ __launch_bounds__(kBlock) void k_mark(const float4 *__restrict__ pts, size_t n, DevParams *__restrict__ dp, KnownParams kp,
                                                 uint32_t *__restrict__ bits, int32_t *__restrict__ cell_idx, PullSource pull,
                                                 unsigned long long *__restrict__ stamps) {
  constexpr int kTile = TILE, kHashCap = TILE / 2;
  vi_stamp(stamps, 0);
  if (!kp.valid && dp->status != VI_OK) {
    // KEEP (a device cloud the caller wants a copy of, pull.d_copy without pull.h_rows): the copy does not depend on the grid
    // definition — a build that ends in a range or an empty report still leaves it complete
    if (!PULL && pull.d_copy) {
      for (int k = 0; k < kTile / kBlock; k++) {
        size_t i = (size_t)blockIdx.x * kTile + (size_t)k * kBlock + threadIdx.x;
        if (i < n) pull.d_copy[i] = pts[i];
      }
    }
    return;
  }
  const CellMap m = kp.valid ? kp.map : dp->map;
  __shared__ int pull_ok;
  if (PULL && threadIdx.x == 0) {
    const size_t first = (size_t)blockIdx.x * kTile, last = (first + kTile < n ? first + kTile : n) - 1;
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    // (a flag per 64-byte line, a look about once a microsecond: every look is a read over the link that snoops the line out of the
    // packing core's cache, and 128 workgroups looking at one shared line every 0.1 us made that core's flag stores — hence its
    // packing — five times slower)
    for (size_t c = first >> pull.chunk_shift; ok && c <= (last >> pull.chunk_shift); c++)
      while (__hip_atomic_load(pull.ready + c * kPullFlagStride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != pull.seq) {
        if (wall_clock64() - t0 > pull.patience_ticks) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    pull_ok = ok;
  }
  // the table is keyed by bitmap WORD (32 x-adjacent cells) and holds the OR of the tile's bits in it: the memory-side
  // atomics of the flush serialise on a shared word, and x-neighbours in one word are the common case (measured at 2 M
  // points / 0.3 m: 269 distinct cells but 107 distinct words per tile; one atomic per cell made the flush 42 of 56 us)
  __shared__ int keys[kHashCap];
  __shared__ uint32_t vals[kHashCap];
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) { keys[k] = -1; vals[k] = 0u; }
  __syncthreads();
  vi_stamp(stamps, 1);
  size_t base = (size_t)blockIdx.x * kTile;
  if (PULL) {
    if (!pull_ok) {   // the host never delivered (it is gone, or held up for seconds): flag the build, touch nothing else
      for (int k = 0; k < kTile / kBlock; k++) {
        size_t i = base + (size_t)k * kBlock + threadIdx.x;
        if (i < n && cell_idx) cell_idx[i] = -1;
      }
      if (threadIdx.x == 0) atomicMax(&dp->spec_miss, 2);
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // (every wave: its loads of the rows come behind the flag thread 0 saw)
  }
  // all 8 loads of a lane are issued before the first use: one HBM latency per tile instead of eight
  constexpr int kPer = kTile / kBlock;
  float4 pv[kPer];
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    size_t i = vi_point_index<kPer, PULL>(base, k);
    pv[k] = (i < n) ? (PULL ? pull.h_rows[i] : pts[i]) : make_float4(NAN, NAN, NAN, 0.f);
  }
  if (PULL || pull.d_copy) {   // (KEEP: wave-uniform; the rows are in registers anyway, the stores are not waited for)
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      size_t i = vi_point_index<kPer, PULL>(base, k);
      if (i < n) pull.d_copy[i] = pv[k];
    }
  }
  // a lane's points run by run (equal bitmap words in a row): a run's bits are OR-ed in a register and reach the table once.  A
  // finished run is noted at the point that ends it (the first of the next run), the last one in an extra entry: every index static.
  int word[kPer + 1];
  uint32_t rbits[kPer + 1];
  {
    int cw = -1;
    uint32_t cb = 0u;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      size_t i = vi_point_index<kPer, PULL>(base, k);
      word[k] = -1; rbits[k] = 0u;
      if (i >= n) continue;
      float4 p = pv[k];
      int cell = -1;
      if (lsa_finite3(p.x, p.y, p.z)) {
        int ijk[3];
        lsa_cell_ijk(m, p.x, p.y, p.z, ijk);
        cell = lsa_cell_linear(m, ijk);
        if (cell < 0) atomicMax(&dp->spec_miss, 1);  // only possible when the grid definition was reused from an earlier build (rare; never hides a 2)
      }
      if (cell_idx) cell_idx[i] = cell;   // (only for an owner that asked for it: VoxelIndex::want_cell_idx)
      if (cell < 0) continue;
      const int w = cell >> 5;
      const uint32_t bit = 1u << (cell & 31);
      if (w == cw) cb |= bit;
      else { word[k] = cw; rbits[k] = cb; cw = w; cb = bit; }
    }
    word[kPer] = cw; rbits[kPer] = cb;
  }
  int slot[kPer + 1];
  if constexpr (kPer > 2) {
    lds_insert_batch<kHashCap, kPer + 1>(keys, word, slot);
  } else {
    // (two points per lane — a lidar scan cut into 512-point tiles: nothing to walk side by side, and the batch's wave-wide exit test
    // is a dependent step more in a kernel that is one latency chain: 5.96 -> 6.59 us with it at 131 k points)
#pragma unroll
    for (int k = 0; k <= kPer; k++) slot[k] = word[k] >= 0 ? lds_insert<kHashCap>(keys, word[k]) : -1;
  }
#pragma unroll
  for (int k = 0; k <= kPer; k++) {
    if (word[k] < 0) continue;
    if (slot[k] < 0) atomicOr(bits + word[k], rbits[k]);
    else if ((vals[slot[k]] & rbits[k]) != rbits[k]) atomicOr(&vals[slot[k]], rbits[k]);  // racy read: at worst a redundant LDS atomic
  }
  vi_stamp(stamps, 2);
  __syncthreads();
  vi_stamp(stamps, 3);
  // one fire-and-forget atomic per distinct word of the tile.  No read-before-write: a plain load may be served from a
  // stale L2 line, and a device-scope load of words that hundreds of tiles hit at once was measured slower than the
  // atomics it saves (k_mark at 2 M points: 29 us without, 59 us with)
#pragma unroll
  for (int k = 0; k < kHashCap / kBlock; k++) {
    int word = keys[k * kBlock + threadIdx.x];
    if (word >= 0) atomicOr(bits + word, vals[k * kBlock + threadIdx.x]);
  }
  vi_stamp(stamps, 4);
  if (stamps) { __builtin_amdgcn_s_waitcnt(0); vi_stamp(stamps, 5); }   // (diagnostics only: until the flush has been acknowledged)
}

// ---- large clouds: cells marked by plain byte stores, packed into the bitmap afterwards (VoxelIndex::occ) -----------------------
// A streaming kernel: no LDS table, no barrier, no flush.  A lane's PER points are 64 apart in the cloud (vi_point_index): runs of
// equal cells among them are stored once.
template <int PER>
__global__ __launch_bounds__(kBlock) void k_mark_bytes(const float4 *__restrict__ pts, size_t n, DevParams *__restrict__ dp, KnownParams kp,
                                                       uint8_t *__restrict__ occ, int32_t *__restrict__ cell_idx, float4 *__restrict__ d_keep,
                                                       unsigned long long *__restrict__ stamps) {
  vi_stamp(stamps, 0);
  const size_t base = (size_t)blockIdx.x * (kBlock * PER);
  if (!kp.valid && dp->status != VI_OK) {
    if (d_keep) {   // (see k_mark: the caller's copy does not depend on the grid definition)
      for (int k = 0; k < PER; k++) {
        size_t i = vi_point_index<PER>(base, k);
        if (i < n) d_keep[i] = pts[i];
      }
    }
    return;
  }
  const CellMap m = kp.valid ? kp.map : dp->map;
  float4 pv[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    size_t i = vi_point_index<PER>(base, k);
    pv[k] = (i < n) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
  }
  if (d_keep) {
#pragma unroll
    for (int k = 0; k < PER; k++) {
      size_t i = vi_point_index<PER>(base, k);
      if (i < n) d_keep[i] = pv[k];
    }
  }
  vi_stamp(stamps, 1);
  int run = -1;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    size_t i = vi_point_index<PER>(base, k);
    if (i >= n) continue;
    const float4 p = pv[k];
    int cell = -1;
    if (lsa_finite3(p.x, p.y, p.z)) {
      int ijk[3];
      lsa_cell_ijk(m, p.x, p.y, p.z, ijk);
      cell = lsa_cell_linear(m, ijk);
      if (cell < 0) atomicMax(&dp->spec_miss, 1);  // only possible when the grid definition was reused from an earlier build
    }
    if (cell_idx) cell_idx[i] = cell;
    if (cell >= 0 && cell != run) { occ[cell] = (uint8_t)1; run = cell; }
  }
  vi_stamp(stamps, 2);
  if (stamps) {   // (diagnostics only: the remaining phase slots of k_mark's record, and when the stores have been acknowledged)
    vi_stamp(stamps, 3); vi_stamp(stamps, 4);
    __builtin_amdgcn_s_waitcnt(0);
    vi_stamp(stamps, 5);
  }
}

// one lane per bitmap word: its 32 cells' bytes -> bits, and the bytes cleared for the next build.  Every word of the build's box is
// written (zero where nothing was marked); the grid covers the bitmap's capacity, the box's word count is only known on the device.
__device__ __forceinline__ uint32_t occ_nibble(uint32_t x) {   // bytes b0..b3 (0 / 1) -> b0 | b1 << 1 | b2 << 2 | b3 << 3
  x &= 0x01010101u;
  return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
}
__global__ __launch_bounds__(kBlock) void k_occ_to_bits(uint4 *__restrict__ occ16, const DevParams *__restrict__ dp, KnownParams kp, uint32_t *__restrict__ bits) {
  if (!kp.valid && dp->status != VI_OK) return;   // (nothing was marked)
  const size_t W = kp.valid ? (size_t)kp.W : (size_t)dp->W;
  const size_t w = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= W) return;
  const uint4 a = occ16[2 * w], b = occ16[2 * w + 1];
  const uint32_t word = occ_nibble(a.x) | (occ_nibble(a.y) << 4) | (occ_nibble(a.z) << 8) | (occ_nibble(a.w) << 12) | (occ_nibble(b.x) << 16) |
                        (occ_nibble(b.y) << 20) | (occ_nibble(b.z) << 24) | (occ_nibble(b.w) << 28);
  bits[w] = word;
  if (word) {
    occ16[2 * w] = make_uint4(0u, 0u, 0u, 0u);
    occ16[2 * w + 1] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ---- popcount prefix scan of the bitmap, single pass -------------------------------------------------------------
// One launch instead of three (tile sums -> scan of the sums -> apply): decoupled look-back.  A workgroup takes a
// ticket (its tile index, so tiles are started in order and the look-back can never wait on a tile that has not been
// scheduled), publishes its tile's popcount as an AGGREGATE, and reads its predecessors' states 64 at a time until it
// meets one that already carries an inclusive PREFIX.  A state word = {epoch : 30, flag : 2, value : 32}; the epoch
// makes words left by earlier launches invalid without clearing the array.  The word IS the message (no other memory
// is handed from tile to tile), so the device-scope atomics are RELAXED: an acquire / release pair would add an L2
// invalidate / write-back per tile on this eight-L2 part.
constexpr unsigned long long kFlagAgg = 1ull, kFlagPrefix = 2ull;
constexpr unsigned kScanRedundantMax = 96;  // tiles (x 2048 words = 6.3 M cells): above this the look-back scan runs
__device__ __forceinline__ unsigned long long scan_word(unsigned epoch, unsigned long long flag, unsigned value) {
  return ((unsigned long long)epoch << 34) | (flag << 32) | (unsigned long long)value;
}

// 512 threads of 4 words each (was 256 of 8): the kernel is one dependent chain per wave — loads of the bits in front, a scan, then a
// bit-by-bit walk whose every step is a store — and at one wave per SIMD nothing hides a step's latency; two waves per SIMD share a
// tile's walk (phase stamps, profiles/r05_scan_stamps.log: the one tile that holds the ground around the sensor, 10-15 k of a
// 2 M-point map's 48 k cells at 0.3 m, spent 10 us of the kernel's 12.8 writing its cells' list)
constexpr int kScanBlock = 512, kScanPer = kScanTile / kScanBlock;
static_assert(kScanPer == 4, "a thread's words are one 16-byte vector");
__global__ __launch_bounds__(kScanBlock) void k_scan(const uint32_t *__restrict__ bits, DevParams *__restrict__ dp, KnownParams kp,
                                                 unsigned long long *__restrict__ state, unsigned *__restrict__ ticket, unsigned epoch,
                                                 uint32_t *__restrict__ prefix, int32_t *__restrict__ cell_of_rank, DevParams *__restrict__ host_mirror,
                                                 unsigned long long *__restrict__ stamps) {
  vi_stamp(stamps, 0);
  // host_mirror (optional, host-mapped pinned memory): the grid definition + occupied-cell count + status as soon as they exist, i.e.
  // behind this kernel — the caller that has to size its own buffers by them (the NDT table) waits for THIS kernel's event while the
  // sums are still being accumulated, instead of for a copy command behind everything
  __shared__ unsigned s_tile, s_excl;
  __shared__ unsigned wt[kScanBlock / 64], wr[kScanBlock / 64];
  // Small bitmaps (a lidar scan at 0.5 m: ~24 tiles) do not go through the look-back at all: the chain of tile-to-tile
  // hand-offs through memory is serial (~0.5 us per hop across XCDs: k_scan took 12.8 us of the 31 us filter chain), while
  // every workgroup can simply count the bits in FRONT of its tile itself — at most kScanRedundantMax x 8 KB of an
  // L2-resident bitmap, no dependency on any other workgroup.  Which path runs is decided on the device (the tile count is
  // only known there); the ticket counter is touched by the look-back path alone.
  // the tile this workgroup will most likely own is loaded before anything is known of the grid definition (the grid
  // covers the bitmap's capacity, so the address is valid whatever the definition says): its latency overlaps the
  // definition's
  uint4 lo = *reinterpret_cast<const uint4 *>(bits + (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPer);
  const unsigned nsb = kp.valid ? kp.nsb : dp->nsb;
  const bool redundant = nsb <= kScanRedundantMax;
  if (!redundant) {
    if (threadIdx.x == 0) {
      unsigned t = atomicAdd(ticket, 1u);
      if (t == gridDim.x - 1) *ticket = 0u;  // every ticket of this launch has been handed out: reset for the next one
      s_tile = t;
    }
    __syncthreads();
  }
  const unsigned tile = redundant ? blockIdx.x : s_tile;
  if ((!kp.valid && dp->status != VI_OK) || tile >= nsb) {
    if (host_mirror && !kp.valid && dp->status != VI_OK && blockIdx.x == 0 && threadIdx.x == 0) *host_mirror = *dp;   // (no tile runs: the status goes out here)
    return;
  }
  const size_t W = kp.valid ? (size_t)kp.W : (size_t)dp->W;
  vi_stamp(stamps, 1);
  unsigned front = 0;  // redundant path: this thread's share of the popcount of words [0, tile * kScanTile)
  if (redundant) {
    // a predecessor tile is kScanTile / 4 = 512 uint4 = one per thread: the loads of up to sixteen tiles are issued before the
    // first popcount, so the front count costs ~one L2 latency per sixteen tiles
    const uint4 *all = reinterpret_cast<const uint4 *>(bits) + threadIdx.x;
    unsigned t = 0;
    for (; t + 16 <= tile; t += 16) {
      uint4 v[16];
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = all[(size_t)(t + k) * kScanBlock];
#pragma unroll
      for (int k = 0; k < 16; k++) front += __popc(v[k].x) + __popc(v[k].y) + __popc(v[k].z) + __popc(v[k].w);
    }
    for (; t + 4 <= tile; t += 4) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = all[(size_t)(t + k) * kScanBlock];
#pragma unroll
      for (int k = 0; k < 4; k++) front += __popc(v[k].x) + __popc(v[k].y) + __popc(v[k].z) + __popc(v[k].w);
    }
    for (; t < tile; t++) {
      uint4 v0 = all[(size_t)t * kScanBlock];
      front += __popc(v0.x) + __popc(v0.y) + __popc(v0.z) + __popc(v0.w);
    }
  }
  if (stamps) { __builtin_amdgcn_s_waitcnt(0); vi_stamp(stamps, 2); }   // (diagnostics only: the bits in front of this tile have arrived)
  size_t base = (size_t)tile * kScanTile + (size_t)threadIdx.x * kScanPer;
  if (tile != blockIdx.x) lo = *reinterpret_cast<const uint4 *>(bits + base);  // look-back path: tiles are handed out by ticket
  uint32_t wv[kScanPer] = {lo.x, lo.y, lo.z, lo.w};
  unsigned local = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; k++) {
    if (base + k >= W) wv[k] = 0u;
    local += __popc(wv[k]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wt[wave] = incl;
  if (redundant) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) front += __shfl_down(front, off, 64);
    if (lane == 0) wr[wave] = front;
  }
  __syncthreads();
  vi_stamp(stamps, 3);
  unsigned agg = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; w++) agg += wt[w];
  if (redundant) {
    if (threadIdx.x == 0) {
      unsigned excl = 0;
#pragma unroll
      for (int w = 0; w < kScanBlock / 64; w++) excl += wr[w];
      s_excl = excl;
      if (tile == nsb - 1) {
        dp->V = excl + agg;  // #occupied cells
        if (host_mirror) { DevParams t = *dp; t.V = excl + agg; *host_mirror = t; }
      }
    }
  } else if (wave == 0) {
    if (lane == 0) __hip_atomic_store(state + tile, scan_word(epoch, tile == 0 ? kFlagPrefix : kFlagAgg, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    int j = (int)tile - 1;  // nearest predecessor not yet accounted for
    while (j >= 0) {        // wave-uniform
      int idx = j - lane;
      unsigned long long w = 0;
      bool ready;
      do {
        ready = true;
        if (idx >= 0) {
          w = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ready = ((unsigned)(w >> 34) == epoch) && (((w >> 32) & 3ull) != 0ull);
        }
      } while (!__all(ready));
      unsigned long long pmask = __ballot(idx >= 0 && ((w >> 32) & 3ull) == kFlagPrefix);
      int stop = pmask ? (__ffsll((long long)pmask) - 1) : 63;  // nearest predecessor that already holds a prefix
      unsigned v = (idx >= 0 && lane <= stop) ? (unsigned)w : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      excl += __shfl(v, 0, 64);
      if (pmask) break;
      j -= 64;
    }
    if (lane == 0) {
      if (tile != 0) __hip_atomic_store(state + tile, scan_word(epoch, kFlagPrefix, excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
      if (tile == nsb - 1) {
        dp->V = excl + agg;  // #occupied cells
        if (host_mirror) { DevParams t = *dp; t.V = excl + agg; *host_mirror = t; }
      }
    }
  }
  __syncthreads();
  vi_stamp(stamps, 4);
  unsigned run = s_excl + incl - local;
  for (int w = 0; w < wave; w++) run += wt[w];
  unsigned pf[kScanPer];
#pragma unroll
  for (int k = 0; k < kScanPer; k++) {
    pf[k] = run;
    run += __popc(wv[k]);
  }
  *reinterpret_cast<uint4 *>(prefix + base) = make_uint4(pf[0], pf[1], pf[2], pf[3]);
  // inverse map (rank -> cell).  A lane's 4 consecutive words are 128 x-adjacent cells, and occupied cells come in x-runs
  // (the ground next to the sensor, a wall): written by the lane that scanned them, the densest lane set the kernel's
  // duration (3.5 us for a tile whose other lanes had nothing to write).  The words and their ranks go through LDS and are
  // dealt out again round-robin, so that a run of n dense words is written by n different lanes.
  __shared__ uint32_t s_w[kScanTile], s_pf[kScanTile];
#pragma unroll
  for (int k = 0; k < kScanPer; k++) { s_w[threadIdx.x * kScanPer + k] = wv[k]; s_pf[threadIdx.x * kScanPer + k] = pf[k]; }
  __syncthreads();
  vi_stamp(stamps, 5);
  const size_t tile_cell0 = (size_t)tile * kScanTile * 32;
#pragma unroll 2
  for (int k = 0; k < kScanPer; k++) {
    const int wi = k * kScanBlock + threadIdx.x;
    uint32_t b = s_w[wi];
    unsigned r = s_pf[wi];
    while (b) {
      int bit = __ffs(b) - 1;
      b &= b - 1;
      cell_of_rank[r++] = (int32_t)(tile_cell0 + (size_t)wi * 32 + bit);
    }
  }
  vi_stamp(stamps, 6);
  if (stamps) { __builtin_amdgcn_s_waitcnt(0); vi_stamp(stamps, 7); }   // (diagnostics only: until the stores have been acknowledged)
}

__device__ __forceinline__ unsigned cell_rank(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ prefix, int cell) {
  return prefix[cell >> 5] + __popc(bits[cell >> 5] & ((1u << (cell & 31)) - 1u));
}

// per-cell fixed-point sums: LDS pre-aggregation per tile, then one set of global atomics per distinct cell.  Rows of acc[]: NS = 4
// (VoxelFilter) one 64-byte line per key — four sums and the count as the fifth 64-bit word, so that a cell's adds are ONE atomic
// instruction touching ONE line, one request to the memory side (vi_row_stride) —, NS = 10 (voxel table) two lines: ten sums and the count.
// BY_CELL (large clouds of an owner that cleans up by key, VoxelIndex::allow_by_cell; NS = 4 only): the rows are keyed by the
// CELL, not by its rank among the occupied cells — the sums then need nothing the scan produces, and this kernel is the build's
// only pass over the points: it also leaves the byte marks of the cells it met (no k_mark_bytes: one 32 MB read, one launch and
// the rank look-ups of the flush less; 268 MB of rows for a 4.2 M-cell box, which 288 GB do not notice).  The scan runs behind it
// for the output order alone, and the owner's finalize kernel reads row cell_of_rank[r] instead of row r.
// PULL (BY_CELL only; host clouds, see k_mark<.., PULL>): the rows are still being packed by the host when this kernel starts; a tile
// waits for its chunk's flag, reads its rows over the link (workgroup-linear order), leaves them in pull.d_copy and sums them in the same
// pass — mark, pull and accumulate of a host scan in one kernel.
template <int NS, int TILE, bool BY_CELL, bool PULL = false>
__global__ __launch_bounds__(kBlock) void k_accum(const float4 *__restrict__ pts, size_t n,
                                                  DevParams *__restrict__ dp, KnownParams kp, const uint32_t *__restrict__ bits,
                                                  const uint32_t *__restrict__ prefix, long long *__restrict__ acc, int32_t *__restrict__ cnt,
                                                  uint8_t *__restrict__ occ, uint32_t *__restrict__ mark_bits, float4 *__restrict__ d_keep,
                                                  unsigned long long *__restrict__ stamps, PullSource pull) {
  static_assert(!PULL || BY_CELL, "the pulling variant sums by cell");
  vi_stamp(stamps, 0);
  // (BY_CELL marks: byte stores into occ[] for large clouds — a pack kernel follows —, or, occ == null, one fire-and-forget atomic OR
  // per cell and tile straight into the bitmap for scans, where the pack kernel would cost more than the ~20 k atomics)
  // table slots: TILE / 2, but TILE / 4 for the 10-sum rows (80 B each): 46 KB instead of 92 KB of LDS lets three
  // workgroups share a CU instead of one; a tile with more distinct cells than slots overflows to direct atomics.
  // NS = 4, TILE = 2048: keys + counts + sums = 4 + 4 + 32 KB = exactly a quarter of the CU's 160 KB, so the 977 tiles of a 2 M-point
  // cloud are all resident at once (with a fourth array they came to 44 KB: 768 resident, and the 209 left over ran a second round
  // of their own — 14 of the kernel's 44 us).
  constexpr int kTile = TILE, kPer = TILE / kBlock, kHashCap = (NS > 4 && TILE >= 1024) ? TILE / 4 : TILE / 2;
  if (!kp.valid && dp->status != VI_OK) {
    if (BY_CELL && d_keep) {   // (see k_mark: the caller's copy does not depend on the grid definition)
      for (int k = 0; k < kPer; k++) {
        size_t i = vi_point_index<kPer>((size_t)blockIdx.x * kTile, k);
        if (i < n) d_keep[i] = pts[i];
      }
    }
    return;
  }
  // a build k_mark has flagged (a point outside the reused cell box; rows of a host cloud that never arrived and were not copied)
  // is thrown away by the host: nothing to sum — and, for the second kind, nothing to be trusted in the rows.  (BY_CELL: this
  // kernel is the one that finds such points; what it has summed by then the owner's finalize kernel clears with the rest.)
  if (!BY_CELL && dp->spec_miss) return;
  const CellMap m = kp.valid ? kp.map : dp->map;
  const FixScales fs = kp.valid ? kp.fs : dp->fs;
  __shared__ int pull_ok;
  if (PULL && threadIdx.x == 0) {   // (as in k_mark<.., PULL>: a flag per 64-byte line, a look about once a microsecond)
    const size_t first = (size_t)blockIdx.x * kTile, last = (first + kTile < n ? first + kTile : n) - 1;
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    for (size_t c = first >> pull.chunk_shift; ok && c <= (last >> pull.chunk_shift); c++)
      while (__hip_atomic_load(pull.ready + c * kPullFlagStride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != pull.seq) {
        if (wall_clock64() - t0 > pull.patience_ticks) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    pull_ok = ok;
  }
  __shared__ int keys[kHashCap];
  __shared__ int counts[kHashCap];
  __shared__ unsigned long long sums[kHashCap][NS];
  for (int k = threadIdx.x; k < kHashCap; k += kBlock) {
    keys[k] = -1;
    counts[k] = 0;
#pragma unroll
    for (int j = 0; j < NS; j++) sums[k][j] = 0ull;
  }
  __syncthreads();
  vi_stamp(stamps, 1);
  size_t base = (size_t)blockIdx.x * kTile;
  if (PULL) {
    if (!pull_ok) {   // the host never delivered: flag the build (the owner sends the rows again or fails), touch nothing else
      if (threadIdx.x == 0) atomicMax(&dp->spec_miss, 2);
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // (every wave: its loads of the rows come behind the flag thread 0 saw)
  }
  float4 pv[kPer];
  int cv[kPer];
#pragma unroll
  for (int k = 0; k < kPer; k++) {  // issue every load of the lane up front
    size_t i = vi_point_index<kPer, PULL>(base, k);
    pv[k] = (i < n) ? (PULL ? pull.h_rows[i] : pts[i]) : make_float4(NAN, NAN, NAN, 0.f);
  }
  if (BY_CELL && (PULL || d_keep)) {
    float4 *dst = PULL ? pull.d_copy : d_keep;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      size_t i = vi_point_index<kPer, PULL>(base, k);
      if (i < n) dst[i] = pv[k];
    }
  }
  // a point's cell by the very expressions k_mark marked it with (the per-point cell index k_mark used to leave in memory was an
  // 8 MB store, an 8 MB load and a kernel-end write-back at 2 M points, for ~20 instructions per point)
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    cv[k] = -1;
    if (lsa_finite3(pv[k].x, pv[k].y, pv[k].z)) {
      int ijk[3];
      lsa_cell_ijk(m, pv[k].x, pv[k].y, pv[k].z, ijk);
      cv[k] = lsa_cell_linear(m, ijk);
      if (BY_CELL && cv[k] < 0) atomicMax(&dp->spec_miss, 1);  // only possible when the grid definition was reused from an earlier build
    }
  }
  // the first cell of every run of equal cells among the lane's points finds its slot (one probe chain for all of them) ...
  int hc[kPer];
  {
    int prev = -1;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      hc[k] = (cv[k] >= 0 && cv[k] != prev) ? cv[k] : -1;
      if (cv[k] >= 0) prev = cv[k];
    }
  }
  int slot[kPer];
  if constexpr (kPer > 2) {
    lds_insert_batch<kHashCap, kPer>(keys, hc, slot);
  } else {
#pragma unroll
    for (int k = 0; k < kPer; k++) slot[k] = hc[k] >= 0 ? lds_insert<kHashCap>(keys, hc[k]) : -1;
  }
  // ... then point after point: the terms (vector arithmetic only) are summed in registers while the run lasts; a finished run's
  // sums go to its slot by adds that return nothing and are not waited for
  int n_direct = 0;   // (diagnostics: runs of this lane that found no slot)
  auto emit = [&](int sl, int cell, const unsigned long long (&rs)[NS], int rn) {
    if (sl < 0) n_direct++;
    if (sl >= 0) {
#pragma unroll
      for (int j = 0; j < NS; j++) atomicAdd(&sums[sl][j], rs[j]);
      atomicAdd(&counts[sl], rn);
    } else {  // table full: straight to memory
      if constexpr (BY_CELL) {
        unsigned long long *a = (unsigned long long *)acc + (size_t)(unsigned)cell * kByCellStride;
#pragma unroll
        for (int j = 0; j < NS; j++) atomicAdd(a + j, rs[j]);
        atomicAdd(a + NS, (unsigned long long)(unsigned)rn);
        if (occ) occ[cell] = (uint8_t)1;
        else atomicOr(mark_bits + (cell >> 5), 1u << (cell & 31));
      } else {
        unsigned rank = cell_rank(bits, prefix, cell);
        unsigned long long *a = (unsigned long long *)acc + (size_t)rank * vi_row_stride(NS);
#pragma unroll
        for (int j = 0; j < NS; j++) atomicAdd(a + j, rs[j]);
        atomicAdd(a + NS, (unsigned long long)(unsigned)rn);
      }
    }
  };
  bool open = false;
  int oslot = -1, ocell = -1, on = 0;
  unsigned long long os[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) os[j] = 0ull;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int cell = cv[k];
    if (cell < 0) continue;
    const float4 p = pv[k];
    // the cell's low corner from the point itself (the very expression k_mark took the cell from) instead of unravelling the linear
    // index: two integer divisions, ~80 instructions, a quarter of this kernel's vector instructions at 2 M points
    int ijk[3];
    lsa_cell_ijk(m, p.x, p.y, p.z, ijk);
    const int abs_ijk[3] = {ijk[0] + m.min_b[0], ijk[1] + m.min_b[1], ijk[2] + m.min_b[2]};
    double o[3];
    lsa_abs_origin(abs_ijk, m.leaf, o);
    unsigned long long v[NS];
    vi_point_terms<NS>(o, fs, p, v);
    if (hc[k] >= 0) {   // a new run starts here
      if (open) emit(oslot, ocell, os, on);
      open = true; oslot = slot[k]; ocell = cell; on = 1;
#pragma unroll
      for (int j = 0; j < NS; j++) os[j] = v[j];
    } else {
      on++;
#pragma unroll
      for (int j = 0; j < NS; j++) os[j] += v[j];
    }
  }
  if (open) emit(oslot, ocell, os, on);
  vi_stamp(stamps, 2);
  __syncthreads();
  vi_stamp(stamps, 3);
  if (stamps) {   // stamp 7 of the workgroup's record: how many of its runs went straight to memory (table full)
    if (threadIdx.x == 0 && blockIdx.x < kViStampWgs) stamps[(size_t)blockIdx.x * 8 + 7] = 0ull;
    __syncthreads();
    if (n_direct && blockIdx.x < kViStampWgs) atomicAdd(stamps + (size_t)blockIdx.x * 8 + 7, (unsigned long long)n_direct);
  }
  // flush, step 1: a lane's slots are independent, their rank look-ups are issued together; the rank replaces the cell in keys[]
  // (each slot is read and rewritten by its one lane; -1 stays -1).  BY_CELL: the cell IS the row.
  if constexpr (!BY_CELL) {
    int fcell[kHashCap / kBlock];
    uint32_t fw[kHashCap / kBlock], fp[kHashCap / kBlock];
#pragma unroll
    for (int k = 0; k < kHashCap / kBlock; k++) {
      fcell[k] = keys[k * kBlock + threadIdx.x];
      int c = fcell[k] < 0 ? 0 : fcell[k];
      fw[k] = bits[c >> 5];
      fp[k] = prefix[c >> 5];
    }
#pragma unroll
    for (int k = 0; k < kHashCap / kBlock; k++)
      if (fcell[k] >= 0) keys[k * kBlock + threadIdx.x] = (int)(fp[k] + __popc(fw[k] & ((1u << (fcell[k] & 31)) - 1u)));
    __syncthreads();
  }
  vi_stamp(stamps, 4);
  // step 2: one lane per (slot, term): the NS atomics of a cell leave from adjacent lanes and land in one or two 64-byte
  // lines of acc[], instead of NS instructions whose 64 lanes each touch a different line.  Every slot is visited (a quarter of them
  // is in use: the walk is 5 reads per lane more than over a compacted list, and the list's array is what the fourth resident
  // workgroup needed)
  constexpr int kFlush = kHashCap * (NS + 1);
  for (int e0 = 0; e0 < kFlush; e0 += 4 * kBlock) {
    int rk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = e0 + q * kBlock + (int)threadIdx.x;
      rk[q] = e < kFlush ? keys[e / (NS + 1)] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = e0 + q * kBlock + (int)threadIdx.x;
      if (rk[q] < 0) continue;
      const int sl = e / (NS + 1), j = e - sl * (NS + 1);
      if constexpr (BY_CELL) {
        // the cell's NS sums and its count: NS + 1 adjacent lanes, ONE 64-bit atomic instruction, one 64-byte line
        atomicAdd((unsigned long long *)acc + (size_t)(unsigned)rk[q] * kByCellStride + j, j < NS ? sums[sl][j < NS ? j : 0] : (unsigned long long)(unsigned)counts[sl]);
        if (j == NS) {   // the cell's mark: a byte (plain store: see k_mark_bytes) or, for scans, its bit
          if (occ) occ[(unsigned)rk[q]] = (uint8_t)1;
          else atomicOr(mark_bits + ((unsigned)rk[q] >> 5), 1u << ((unsigned)rk[q] & 31u));
        }
      } else {   // (the same rows keyed by the rank: one line for NS = 4, two for NS = 10 — vi_row_stride)
        atomicAdd((unsigned long long *)acc + (size_t)(unsigned)rk[q] * vi_row_stride(NS) + j, j < NS ? sums[sl][j < NS ? j : 0] : (unsigned long long)(unsigned)counts[sl]);
      }
    }
  }
  vi_stamp(stamps, 5);
  if (stamps) { __builtin_amdgcn_s_waitcnt(0); vi_stamp(stamps, 6); }   // (diagnostics only: until the flush has been acknowledged)
}

// zero what this build touched: the bitmap words of the occupied cells and the V rows of sums / counts
__global__ __launch_bounds__(kBlock) void k_cleanup(const DevParams *__restrict__ dp, const int32_t *__restrict__ cell_of_rank, uint32_t *__restrict__ bits,
                                                    long long *__restrict__ acc, int32_t *__restrict__ cnt, int ns, int by_cell) {
  if (dp->status != VI_OK) return;
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= dp->V) return;
  const int cell = cell_of_rank[r];
  bits[cell >> 5] = 0u;
  const size_t key = by_cell ? (size_t)cell : r;
  for (int j = 0; j <= ns; j++) acc[key * vi_row_stride(ns) + j] = 0;
}

inline unsigned stream_grid(lsa_context *ctx, size_t n) {
  size_t blocks = (n + kBlock - 1) / kBlock;
  size_t cap = (size_t)ctx->num_cus * 8;  // >> 256 workgroups, grid-stride beyond (guide: ~2048 blocks)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

// every buffer a build over `n` points with `ns` sums per cell needs, and a bitmap of at least w_min words (0: whatever is there,
// 4096 words on first use — the device reports VI_CAPACITY with the bounding box and vi_grow() sizes it)
int vi_prepare(lsa_context *ctx, VoxelIndex &vi, size_t n, int ns, size_t w_min) {
  int rc;
  unsigned nb = stream_grid(ctx, n);
  if ((rc = vi.scratch.reserve(ctx, (size_t)nb * 8 + 8, 1.25))) return rc;
  if ((rc = vi.dparams.reserve(ctx, 1))) return rc;
  if ((rc = vi.h_params.reserve(ctx, 2))) return rc;
  if (!vi.ev_done) LSA_HIP(ctx, hipEventCreateWithFlags(&vi.ev_done, hipEventDisableTiming | hipEventReleaseToSystem));
  // buffers sized by the point count (#occupied <= #points)
  size_t old_cor = vi.cell_of_rank.cap, old_cnt = vi.cnt.cap, old_acc = vi.acc.cap;
  if (vi.want_cell_idx && (rc = vi.cell_idx.reserve(ctx, n, 1.25))) return rc;
  if ((rc = vi.cell_of_rank.reserve(ctx, n, 1.25))) return rc;
  if ((rc = vi.cnt.reserve(ctx, n + 1, 1.25))) return rc;
  if ((rc = vi.acc.reserve(ctx, (n + 1) * vi_row_stride(ns), 1.25))) return rc;
  if (vi.cnt.cap != old_cnt || vi.acc.cap != old_acc || vi.cell_of_rank.cap != old_cor || vi.acc_ns != ns) vi.clean = false;
  if (vi.W_cap == 0 && w_min < 4096) w_min = 4096;
  if (vi.W_cap < w_min) {
    w_min = (w_min + kScanTile - 1) / kScanTile * kScanTile;  // whole scan tiles: k_scan reads / writes 16-byte vectors
    if (vi.bits.p) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the reserves free the old bitmap)
    if ((rc = vi.bits.reserve(ctx, w_min))) return rc;
    if ((rc = vi.prefix.reserve(ctx, w_min))) return rc;
    if ((rc = vi.scan_state.reserve(ctx, w_min / kScanTile + 2))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(vi.scan_state.p, 0, vi.scan_state.cap * sizeof(unsigned long long), ctx->stream));
    vi.scan_epoch = 0;
    vi.W_cap = w_min;
    vi.clean = false;
  }
  if (!vi.tickets.p) {
    if ((rc = vi.tickets.reserve(ctx, 2))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(vi.tickets.p, 0, 2 * sizeof(unsigned), ctx->stream));
  }
  return LSA_OK;
}

// what a LARGE build (>= kSmallInput points) needs beyond vi_prepare, allocated and cleared ahead of the first one: the byte map of the
// bitmap's capacity and, for an owner that allows it, the rows keyed by the cell (lsa_voxel_filter_reserve)
int vi_prepare_large(lsa_context *ctx, VoxelIndex &vi, int ns, bool with_occ) {
  int rc;
  if (with_occ && vi.occ_words != vi.W_cap) {
    if (vi.occ.p) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    vi.occ.release();
    if ((rc = vi.occ.reserve(ctx, vi.W_cap * 32))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(vi.occ.p, 0, vi.W_cap * 32, ctx->stream));
    vi.occ_words = vi.W_cap;
  }
  const size_t cells_cap = vi.W_cap * 32;
  if (vi.allow_by_cell && ns == 4 && !vi.mirror_at_scan && cells_cap * kByCellStride * 8 <= kByCellMaxBytes && vi.acc.cap < cells_cap * kByCellStride) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = vi.acc.reserve(ctx, cells_cap * kByCellStride))) return rc;
    vi.clean = false;
  }
  return LSA_OK;
}

bool vi_spec_ready(const VoxelIndex &vi, size_t n, int mode, const float leaf[3]) {
  return mode == 0 && vi.spec_valid && vi.spec_mode == mode && vi.spec_leaf[0] == leaf[0] && vi.spec_leaf[1] == leaf[1] && vi.spec_leaf[2] == leaf[2] &&
         (unsigned long long)n <= vi.spec_n_cap && vi.spec_W_cap == vi.W_cap && vi.W_cap != 0;
}

int vi_enqueue(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, int mode, const float leaf[3], int ns, bool allow_spec, const PullSource *pull,
               float4 *d_keep) {
  int rc;
  unsigned nb = stream_grid(ctx, n);
  if ((rc = vi_prepare(ctx, vi, n, ns, 0))) return rc;
  // rows by cell (see VoxelIndex::allow_by_cell): large device clouds whose box capacity keeps the rows below kByCellMaxBytes
  // (LSA_VI_BY_CELL=0: the A/B arm and the equality test of the two keyings)
  const bool by_cell_on = !(getenv("LSA_VI_BY_CELL") && atoi(getenv("LSA_VI_BY_CELL")) == 0);
  const size_t cells_cap = vi.W_cap * 32;
  // (not for an owner that waits behind the scan while the sums are still running, mirror_at_scan: by cell the sums come FIRST,
  // and SetInputTarget of 2 M points measured 0.106 ms that way against 0.098 with its host work under the accumulation)
  // (scans too, LSA_VI_BY_CELL_SMALL=0 switches that off: one kernel — sums by cell + the cells' bits — instead of mark and accumulate
  // on either side of the scan; every dependent launch of this chain costs ~4.5 us whatever it does)
  const bool by_cell_small = !(getenv("LSA_VI_BY_CELL_SMALL") && atoi(getenv("LSA_VI_BY_CELL_SMALL")) == 0);
  const bool by_cell = vi.allow_by_cell && ns == 4 && !vi.mirror_at_scan && (n >= kSmallInput || by_cell_small || pull) &&
                       cells_cap * kByCellStride * 8 <= kByCellMaxBytes && by_cell_on && !(pull && !by_cell_small);
  if (by_cell && vi.acc.cap < cells_cap * kByCellStride) {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the reserve frees the old rows)
    if ((rc = vi.acc.reserve(ctx, cells_cap * kByCellStride))) return rc;
    vi.clean = false;
  }
  vi.by_cell = by_cell;
  if (++vi.scan_epoch >= (1u << 30)) {  // epoch wrap: invalidate every state word the old way
    LSA_HIP(ctx, hipMemsetAsync(vi.scan_state.p, 0, vi.scan_state.cap * sizeof(unsigned long long), ctx->stream));
    vi.scan_epoch = 1;
  }
  if (!vi.clean) {
    LSA_HIP(ctx, hipMemsetAsync(vi.bits.p, 0, vi.W_cap * sizeof(uint32_t), ctx->stream));
    LSA_HIP(ctx, hipMemsetAsync(vi.acc.p, 0, vi.acc.cap * sizeof(long long), ctx->stream));
    LSA_HIP(ctx, hipMemsetAsync(vi.cnt.p, 0, vi.cnt.cap * sizeof(int32_t), ctx->stream));
  }
  vi.clean = false;
  vi.acc_ns = ns;
  // request block -> device
  ReqParams req;
  req.mode = mode;
  for (int a = 0; a < 3; a++) req.leaf[a] = leaf[a];
  req.W_cap = vi.W_cap;
  req.scale_n_min = vi.scale_n_min;
  // (round 4: for large inputs too.  At 2 M points the two bounding-box launches are 16 of ~100 us — a full read of the cloud —
  // and a map assembled from a sliding window of key-frames keeps its extent from one build to the next: the box is defined with
  // a 10 % margin, the window moves ~2 m per key-frame, so a miss — detected by k_mark, answered with one exact rebuild — comes
  // about once in a dozen builds.)
  req.expand = (allow_spec && mode == 0) ? 1 : 0;
  // reuse the previous build's grid definition (still in vi.dparams on the device) when it is known to be good for
  // this request: same rule set and leaf, scales valid for n, bitmap capacity unchanged
  const bool spec = req.expand && vi.spec_valid && vi.spec_mode == mode && vi.spec_leaf[0] == leaf[0] && vi.spec_leaf[1] == leaf[1] &&
                    vi.spec_leaf[2] == leaf[2] && (unsigned long long)n <= vi.spec_n_cap && vi.spec_W_cap == vi.W_cap;
  vi.last_allow_spec = req.expand != 0;
  vi.last_was_spec = spec;
  vi.spec_mode = mode;
  for (int a = 0; a < 3; a++) vi.spec_leaf[a] = leaf[a];
  unsigned nsb_cap = (unsigned)((vi.W_cap + kScanTile - 1) / kScanTile);
  DevParams *dp = vi.dparams.p;
  if (!spec) {
    hipLaunchKernelGGL(k_bbox_partial, dim3(nb), dim3(kBlock), 0, ctx->stream, d_pts, n, vi.scratch.p);
    hipLaunchKernelGGL(k_bbox_final_params, dim3(1), dim3(kBlock), 0, ctx->stream, vi.scratch.p, (int)nb, req, vi.dparams.p);
  }
  const bool small = n < kSmallInput;
  KnownParams kp;
  memset(&kp, 0, sizeof(kp));
  if (spec) {  // the definition the device holds is the one vi_wait() read back after the previous build
    kp.valid = 1;
    kp.map = vi.map; kp.fs = vi.fs;
    kp.W = (unsigned long long)vi.W;
    kp.nsb = (unsigned)((vi.W + kScanTile - 1) / kScanTile);
  }
  PullSource no_pull;
  memset(&no_pull, 0, sizeof(no_pull));
  no_pull.d_copy = d_keep;   // (a pulled build leaves the rows in pull->d_copy already)
  if (pull) {
    if (!spec) return lsa_fail(ctx, LSA_ERR_INVALID, "vi_enqueue: rows can only be pulled into a known grid definition (vi_spec_ready)");
    if (by_cell)   // pull, marks and sums of a host scan in ONE kernel (bits set by atomic ORs: a scan-sized cloud whatever n says)
      hipLaunchKernelGGL((k_accum<4, kTilePull, true, true>), dim3(lsa_div_up(n, kTilePull)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.prefix.p,
                         vi.acc.p, vi.cnt.p, (uint8_t *)nullptr, vi.bits.p, (float4 *)nullptr, vi.stamps.p ? vi.stamps.p + kViStampWgs * 8 : nullptr, *pull);
    else
      hipLaunchKernelGGL((k_mark<kTilePull, true>), dim3(lsa_div_up(n, kTilePull)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.cell_idx.p, *pull, vi.stamps.p);
  } else if (small && by_cell) {
    hipLaunchKernelGGL((k_accum<4, kTileSmall, true>), dim3(lsa_div_up(n, kTileSmall)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.prefix.p,
                       vi.acc.p, vi.cnt.p, (uint8_t *)nullptr, vi.bits.p, d_keep, vi.stamps.p ? vi.stamps.p + kViStampWgs * 8 : nullptr, no_pull);
  } else if (small) hipLaunchKernelGGL((k_mark<kTileSmall, false>), dim3(lsa_div_up(n, kTileSmall)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.cell_idx.p, no_pull, vi.stamps.p);
  else {
    // large clouds: byte marks, then the bytes of the box packed into the bitmap (see VoxelIndex::occ)
    unsigned long long *st_acc = vi.stamps.p ? vi.stamps.p + kViStampWgs * 8 : nullptr;
    if (vi.occ_words != vi.W_cap) {
      if (vi.occ.p) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the reserve frees the old map)
      vi.occ.release();
      if ((rc = vi.occ.reserve(ctx, vi.W_cap * 32))) return rc;
      LSA_HIP(ctx, hipMemsetAsync(vi.occ.p, 0, vi.W_cap * 32, ctx->stream));
      vi.occ_words = vi.W_cap;
    }
    if (by_cell) {   // ONE pass over the points: the sums by cell and the byte marks (k_accum<.., BY_CELL>); pack and scan run behind it
      hipLaunchKernelGGL((k_accum<4, kTileBig, true>), dim3(lsa_div_up(n, kTileBig)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.prefix.p,
                         vi.acc.p, vi.cnt.p, vi.occ.p, (uint32_t *)nullptr, d_keep, st_acc, no_pull);
    } else {
      hipLaunchKernelGGL((k_mark_bytes<kTileBig / kBlock>), dim3(lsa_div_up(n, kTileBig)), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.occ.p, vi.cell_idx.p, d_keep,
                         vi.stamps.p);
    }
    hipLaunchKernelGGL(k_occ_to_bits, dim3(lsa_div_up(vi.W_cap, kBlock)), dim3(kBlock), 0, ctx->stream, reinterpret_cast<uint4 *>(vi.occ.p), dp, kp, vi.bits.p);
  }
  hipLaunchKernelGGL(k_scan, dim3(nsb_cap), dim3(kScanBlock), 0, ctx->stream, vi.bits.p, vi.dparams.p, kp, vi.scan_state.p, vi.tickets.p + 1, vi.scan_epoch,
                     vi.prefix.p, vi.cell_of_rank.p, vi.mirror_at_scan ? vi.h_params.p : (DevParams *)nullptr, vi.stamps.p ? vi.stamps.p + 2 * kViStampWgs * 8 : nullptr);
  if (vi.mirror_at_scan) LSA_HIP(ctx, hipEventRecord(vi.ev_done, ctx->stream));   // vi_wait() returns behind the scan; the accumulation is still running then
  unsigned g = lsa_div_up(n, small ? kTileSmall : kTileBig);
#define LSA_LAUNCH_ACCUM(NS_, TILE_) hipLaunchKernelGGL((k_accum<NS_, TILE_, false>), dim3(g), dim3(kBlock), 0, ctx->stream, d_pts, n, dp, kp, vi.bits.p, vi.prefix.p, vi.acc.p, vi.cnt.p, (uint8_t *)nullptr, (uint32_t *)nullptr, (float4 *)nullptr, vi.stamps.p ? vi.stamps.p + kViStampWgs * 8 : nullptr, no_pull)
  if (!by_cell) {   // (rows by rank: behind the scan)
    if (ns == 4) { if (small) LSA_LAUNCH_ACCUM(4, kTileSmall); else LSA_LAUNCH_ACCUM(4, kTileBig); }
    else { if (small) LSA_LAUNCH_ACCUM(10, kTileSmall); else LSA_LAUNCH_ACCUM(10, kTileBig); }
  }
#undef LSA_LAUNCH_ACCUM
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

int vi_fetch(lsa_context *ctx, VoxelIndex &vi) {
  LSA_HIP(ctx, hipMemcpyAsync(vi.h_params.p, vi.dparams.p, sizeof(DevParams), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipEventRecord(vi.ev_done, ctx->stream));
  return LSA_OK;
}

int vi_mark_done(lsa_context *ctx, VoxelIndex &vi) {
  LSA_HIP(ctx, hipEventRecord(vi.ev_done, ctx->stream));
  return LSA_OK;
}

int vi_wait(lsa_context *ctx, VoxelIndex &vi, int *status) {
  LSA_HIP(ctx, hipEventSynchronize(vi.ev_done));
  const DevParams &p = vi.h_params.p[0];
  vi.map = p.map;
  vi.fs = p.fs;
  vi.W = (size_t)p.W;
  vi.nocc = (p.status == VI_OK) ? p.V : 0;
  *status = p.status;
  // the device-side definition can be reused by the next build if this one was good and was defined with a margin
  vi.spec_valid = vi.last_allow_spec && p.status == VI_OK && !p.spec_miss && p.expand;
  vi.spec_n_cap = p.n_cap;
  vi.spec_W_cap = vi.W_cap;
  if (p.spec_miss) *status = VI_SPEC_MISS;
  vi.last_pull_timeout = p.spec_miss == 2;
  return LSA_OK;
}

int vi_grow(lsa_context *ctx, VoxelIndex &vi) {
  // called after a VI_CAPACITY report: h_params holds the grid the device wanted
  const DevParams &p = vi.h_params.p[0];
  size_t need = (size_t)p.W + 1;
  size_t want = need + (p.expand ? need : need / 4);  // room for the speculative margin (1.1^3 of the cells + 4 cells per axis)
  want = (want + kScanTile - 1) / kScanTile * kScanTile;  // whole scan tiles: k_scan reads / writes 16-byte vectors
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int rc;
  if ((rc = vi.bits.reserve(ctx, want))) return rc;
  if ((rc = vi.prefix.reserve(ctx, want))) return rc;
  if ((rc = vi.scan_state.reserve(ctx, want / kScanTile + 2))) return rc;
  LSA_HIP(ctx, hipMemsetAsync(vi.scan_state.p, 0, vi.scan_state.cap * sizeof(unsigned long long), ctx->stream));
  vi.scan_epoch = 0;
  vi.W_cap = want;
  vi.clean = false;
  return LSA_OK;
}

int vi_cleanup(lsa_context *ctx, VoxelIndex &vi, size_t n) {
  hipLaunchKernelGGL(k_cleanup, dim3(lsa_div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, vi.dparams.p, vi.cell_of_rank.p, vi.bits.p, vi.acc.p, vi.cnt.p,
                     vi.acc_ns, vi.by_cell ? 1 : 0);
  LSA_HIP(ctx, hipGetLastError());
  vi.clean = true;
  return LSA_OK;
}
