// local_map.hip — the front end's incremental sliding local map: see local_map.h for what it replaces
// (FrontEnd::UpdateWithNewFrame, front_end.cpp:348-424) and for the design.
//
// Kernels (all HBM / atomic bound streaming work, nothing GEMM shaped):
//   k_lm_points<NS, MODE, SWAP>  one tile of points per workgroup (a swap tile: 256 of the evicted key-frame's and 256 of the new
//                           one's): pre-aggregated per cell in an LDS table (consecutive lidar returns share cells), then each
//                           distinct cell finds or claims its line in the global table (64-bit CAS on the key) and receives its
//                           NS sums and its count as ONE wave-instruction's adjacent lanes into that line — a returning add:
//                           the lane that carried the count learns from the old value whether its tile is the cell's first
//                           toucher and lists the cell in the TILE's region of the dirty list.  Level F (NS 4, PCL's cell rule)
//                           or, while the map is unfiltered, level N directly (NS 10, the reference VoxelGrid's cell rule).
//                           Bytes per point: 16 read (+ 16 written for the new key-frame's); per distinct (tile, cell): one line.
//   k_lm_filter_dirty       one workgroup per tile region of the dirty list, up to four touched F cells per lane: new
//                           output point from the sums (vi_filter_mean: the VoxelFilter's own arithmetic), old point out of / new
//                           point into level N, summed per N cell in LDS first.
//   k_lm_ndt_dirty          one lane per touched N cell: the voxel record (ndt_cell_record: the full build's own function) and
//                           the dense probe table entry; the last workgroup to finish publishes the status block into host-mapped
//                           memory.
#include "local_map.h"

#include <chrono>

#include "ndt_cell_dev.h"

namespace {
constexpr int kBlock = 256;
constexpr int kTile = 256;          // points per workgroup tile (per cloud): one point of each cloud per lane — the tile's work is a dependent chain
                                    // (load -> cell -> LDS table -> line -> atomics), shorter tiles on more workgroups overlap more of it
constexpr int kLdsProbes = 16;
constexpr int kGlobalProbes = 512;
constexpr unsigned long long kEmpty = 0ull;   // a zero-filled table is empty
constexpr int kCoordOff = 1 << 20;  // 21 bits per axis; the cell (-2^20, -2^20, -2^20) would pack to 0 and is out of range
constexpr unsigned kMaxRegions = 4096;   // regions of a dirty list one consumer launch can scan (its prefix lives in LDS)
constexpr unsigned kMaxTiles = kMaxRegions - 16;   // tiles of one k_lm_points launch: k_lm_filter_dirty adds its overflow workgroups' regions
constexpr unsigned kFdRegion = 512;      // level-N region of a k_lm_filter_dirty workgroup
constexpr unsigned kTouchMask = 0x7fffffffu;

__host__ __device__ inline unsigned long long lm_pack(int i, int j, int k) {
  return (unsigned long long)(unsigned)(i + kCoordOff) | ((unsigned long long)(unsigned)(j + kCoordOff) << 21) | ((unsigned long long)(unsigned)(k + kCoordOff) << 42);
}
__host__ __device__ inline void lm_unpack(unsigned long long key, int ijk[3]) {
  ijk[0] = (int)(key & 0x1fffffull) - kCoordOff;
  ijk[1] = (int)((key >> 21) & 0x1fffffull) - kCoordOff;
  ijk[2] = (int)((key >> 42) & 0x1fffffull) - kCoordOff;
}
__device__ __forceinline__ bool lm_in_range(const int ijk[3]) {
  return (unsigned)(ijk[0] + kCoordOff - 1) < (1u << 21) - 1u && (unsigned)(ijk[1] + kCoordOff - 1) < (1u << 21) - 1u &&
         (unsigned)(ijk[2] + kCoordOff - 1) < (1u << 21) - 1u;
}
__device__ __forceinline__ unsigned lm_hash(unsigned long long key, unsigned cap) {
  return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 32) & (cap - 1u);
}

// Probe sequence of a cell in a GLOBAL table: cells are hashed by 4 x 4 x 2 BLOCK, and a block's 32 cells occupy the 32
// consecutive slots behind the block's hash (a collision moves a cell on by whole groups of 32).  Neighbouring cells — which
// consecutive lidar returns, and the members of one NDT voxel, are — so land in neighbouring lines of the table.
__device__ __forceinline__ unsigned lm_probe0(unsigned long long key, unsigned cap) {
  int ijk[3];
  lm_unpack(key, ijk);
  const unsigned long long block = lm_pack(ijk[0] >> 2, ijk[1] >> 2, ijk[2] >> 1);
  const unsigned local = (unsigned)(ijk[0] & 3) | ((unsigned)(ijk[1] & 3) << 2) | ((unsigned)(ijk[2] & 1) << 4);
  const unsigned groups = cap >> 5;
  return (((unsigned)((block * 0x9E3779B97F4A7C15ull) >> 32) & (groups - 1u)) << 5) + local;
}
constexpr unsigned kProbeStep = 32u;

// absolute cell of a point.  MODE 0: pcl::VoxelGrid, floor(x * inverse_leaf) (what applyFilter subtracts min_b from);
// MODE 1: the reference's VoxelGrid, floor(x / leaf) (VoxelGrid.cpp:218-222) — the rules of lsa_cell_ijk without the box.
template <int MODE>
__device__ __forceinline__ void lm_cell_abs(const float4 &p, const float leaf[3], const float inv[3], int ijk[3]) {
  if (MODE == 0) {
    ijk[0] = (int)floorf(lsa_fmul(p.x, inv[0])); ijk[1] = (int)floorf(lsa_fmul(p.y, inv[1])); ijk[2] = (int)floorf(lsa_fmul(p.z, inv[2]));
  } else {
    ijk[0] = (int)floorf(lsa_fdiv(p.x, leaf[0])); ijk[1] = (int)floorf(lsa_fdiv(p.y, leaf[1])); ijk[2] = (int)floorf(lsa_fdiv(p.z, leaf[2]));
  }
}

// a table seen as lines of STRIDE 64-bit words: word 0 the key, words 1..NS the sums, word NS + 1 the count word
template <int NS>
struct TableView {
  static constexpr unsigned kStride = NS == 4 ? 8u : 16u;
  unsigned long long *w;
  unsigned cap;
  __device__ __forceinline__ unsigned long long *line(unsigned slot) const { return w + (size_t)slot * kStride; }
};

// diagnostics (lsa_front_end_map_stamps): thread 0 of a workgroup notes the 100 MHz wall clock at phase boundaries, 8 per workgroup
__device__ __forceinline__ void lm_stamp(unsigned long long *stamps, int k) {
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + k] = wall_clock64();
}

struct DirtyView {
  unsigned *slot, *len;      // len[0]: entries of the overflow region slot[0 .. over_cap); len[1 + r]: entries of region r
  unsigned regions, region_cap, over_cap;
  __device__ __forceinline__ unsigned *region(unsigned r) const { return slot + over_cap + (size_t)r * region_cap; }
};

// Slot of `key` in the global table, claiming one if it is new.  Keys never change once written, so a plain (possibly stale,
// the eight L2s are not coherent with each other) read can only err towards "empty", which the CAS then settles.
template <int NS>
__device__ __forceinline__ int lm_find_insert(const TableView<NS> &H, unsigned long long key, unsigned *n_new, unsigned *overflow) {
  unsigned h = lm_probe0(key, H.cap);
  for (int probe = 0; probe < kGlobalProbes; probe++) {
    unsigned long long k = *H.line(h);
    if (k == key) return (int)h;
    if (k == kEmpty) {
      unsigned long long old = atomicCAS(H.line(h), kEmpty, key);
      if (old == kEmpty) { atomicAdd(n_new, 1u); return (int)h; }
      if (old == key) return (int)h;
    }
    h = (h + kProbeStep) & (H.cap - 1u);
  }
  atomicOr(overflow, 1u);
  return -1;
}

// a cell that bypassed a workgroup's LDS table (table full), or that no longer fits the workgroup's region: the overflow region
__device__ __forceinline__ void lm_dirty_overflow(const DirtyView &D, unsigned g, unsigned *overflow) {
  const unsigned i = atomicAdd(D.len, 1u);
  if (i < D.over_cap) D.slot[i] = g; else atomicOr(overflow, 1u);
}

// slot of `key` in the workgroup's LDS table (inserting it), or -1 when no slot is free within kLdsProbes steps
template <int CAP>
__device__ __forceinline__ int lm_lds_insert(unsigned long long *keys, unsigned long long key) {
  unsigned h = lm_hash(key, CAP);
  for (int probe = 0; probe < kLdsProbes; probe++) {
    unsigned long long old = keys[h];
    if (old == key) return (int)h;
    if (old == kEmpty) {
      old = atomicCAS(&keys[h], kEmpty, key);
      if (old == kEmpty || old == key) return (int)h;
    }
    h = (h + 1u) & (CAP - 1u);
  }
  return -1;
}

// The flush of a workgroup's LDS table (`CAP` slots: keys / counts / sums) into the global table H, and the cells it is the
// first toucher of into `won` (LDS; *n_won entries on return, behind a barrier).  Every step issues ALL of a lane's memory
// operations before it looks at the first result — a lane owns up to CAP / 256 cells, and taken one after the other their
// round trips to the memory side (~2 us each for a returning atomic) were most of the kernel:
//   step 0: the used LDS slots are listed (LDS only);
//   step 1: every listed cell finds (or claims) its line: the first probe of all of a lane's cells is loaded together;
//   step 2: the NS sums of a cell leave as adjacent lanes of plain (non-returning) adds into its line — nobody waits for them —
//           and the count word takes count + 2^32 by a RETURNING add, all of a lane's issued together: the old value's high half
//           says whether the cell was clean, i.e. whether this workgroup is its first toucher and lists it as dirty.
template <int NS, int CAP>
__device__ __forceinline__ void lm_flush(const TableView<NS> &H, const unsigned long long *keys, const int *counts, const unsigned long long (*sums)[NS],
                                         int *used, int *gslot, unsigned *won, int *n_used, unsigned *n_won, unsigned *n_new_keys, unsigned *overflow,
                                         unsigned long long *stamps = nullptr) {
  constexpr int kPer = CAP / kBlock;
  for (int k = threadIdx.x; k < CAP; k += kBlock)
    if (keys[k] != kEmpty) used[atomicAdd(n_used, 1)] = k;
  __syncthreads();
  const int nu = *n_used;
  {
    unsigned long long key[kPer], first[kPer];
    unsigned h0[kPer];
#pragma unroll
    for (int m = 0; m < kPer; m++) {
      const int u = threadIdx.x + m * kBlock;
      key[m] = u < nu ? keys[used[u]] : kEmpty;
      h0[m] = u < nu ? lm_probe0(key[m], H.cap) : 0u;
    }
#pragma unroll
    for (int m = 0; m < kPer; m++) first[m] = *H.line(h0[m]);   // (idle lanes read line 0: harmless)
#pragma unroll
    for (int m = 0; m < kPer; m++) {
      const int u = threadIdx.x + m * kBlock;
      if (u < nu) gslot[u] = first[m] == key[m] ? (int)h0[m] : lm_find_insert<NS>(H, key[m], n_new_keys, overflow);
    }
  }
  __syncthreads();
  lm_stamp(stamps, 3);
  for (int e = threadIdx.x; e < nu * NS; e += kBlock) {
    const int u = e / NS, j = e - u * NS;
    const int g = gslot[u];
    if (g >= 0) atomicAdd(H.line((unsigned)g) + 1 + j, sums[used[u]][j]);
  }
  {
    unsigned long long old[kPer];
#pragma unroll
    for (int m = 0; m < kPer; m++) {
      const int u = threadIdx.x + m * kBlock;
      const int g = u < nu ? gslot[u] : -1;
      old[m] = g >= 0 ? atomicAdd(H.line((unsigned)g) + 1 + NS, (unsigned long long)(long long)counts[used[u]] + kLmTouch) : ~0ull;
    }
#pragma unroll
    for (int m = 0; m < kPer; m++) {
      const int u = threadIdx.x + m * kBlock;
      if (u < nu && gslot[u] >= 0 && ((unsigned)(old[m] >> 32) & kTouchMask) == 0u) won[atomicAdd(n_won, 1u)] = (unsigned)gslot[u];
    }
  }
  __syncthreads();
  lm_stamp(stamps, 4);
}

struct PointsArgs {
  float leaf[3], inv[3];
  FixScales fs;
  int sign;          // +1 add, -1 subtract (the points of `pts`; plain launches)
  int to_level_n;    // which counters of the status block this table uses
  // swap (k_lm_points with SWAP): `pts` is a slot of the local-map buffer whose old contents leave the sums (if evict) while
  // the key-frame cloud `raw` (n_raw points, rigid transform T, f32 like pcl::transformPointCloud) is written into it and enters
  int evict;
  const float4 *raw;
  size_t n_raw;
  float T[12];
  float4 *keep;      // optional: the key-frame's copy of its (untransformed) cloud
  int force_w_one;
};

template <int NS, bool SWAP>
struct PointsCap { static constexpr int value = SWAP ? (NS == 4 ? 512 : 256) : 256; };

template <int NS, int MODE, bool SWAP>
__global__ __launch_bounds__(kBlock) void k_lm_points(float4 *__restrict__ pts, size_t n, PointsArgs A, TableView<NS> H, DirtyView D, LmStatus *__restrict__ st,
                                                      unsigned long long *stamps) {
  constexpr int kLdsCap = PointsCap<NS, SWAP>::value;   // a swap tile carries two clouds' points; == D.region_cap
  lm_stamp(stamps, 0);
  __shared__ unsigned long long keys[kLdsCap];
  __shared__ int counts[kLdsCap];
  __shared__ unsigned long long sums[kLdsCap][NS];
  __shared__ int gslot[kLdsCap];
  __shared__ int used[kLdsCap];
  __shared__ unsigned won[kLdsCap];     // global slots this tile was the first to touch
  __shared__ int n_used;
  __shared__ unsigned n_won, n_new_keys;
  if (threadIdx.x == 0) { n_used = 0; n_won = 0u; n_new_keys = 0u; }
  for (int k = threadIdx.x; k < kLdsCap; k += kBlock) {
    keys[k] = kEmpty;
    counts[k] = 0;
#pragma unroll
    for (int j = 0; j < NS; j++) sums[k][j] = 0ull;
  }
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kTile;
  // One pass (plain) or two (swap: first the slot's old points with sign -1, then the new key-frame's with sign +1, written
  // into the slot on the way); both passes aggregate into the same LDS table, so a cell both clouds touch is flushed once
  // with its net change.
  // every load of the lane — both clouds' in a swap — is issued up front: one HBM latency per tile
  float4 pv_all[SWAP ? 2 : 1][kTile / kBlock];
#pragma unroll
  for (int k = 0; k < kTile / kBlock; k++) {
    size_t i = base + (size_t)k * kBlock + threadIdx.x;
    if (SWAP) {
      pv_all[0][k] = (A.evict && i < n) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
      pv_all[1][k] = (i < A.n_raw) ? A.raw[i] : make_float4(NAN, NAN, NAN, 0.f);
    } else {
      pv_all[0][k] = (i < n) ? pts[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
  }
  lm_stamp(stamps, 1);
#pragma unroll
  for (int pass = 0; pass < (SWAP ? 2 : 1); pass++) {
    const bool old_pass = SWAP && pass == 0;
    if (old_pass && !A.evict) continue;
    const int sign = SWAP ? (old_pass ? -1 : +1) : A.sign;
    const unsigned long long sgn = sign < 0 ? ~0ull : 0ull;  // two's complement negation: (v ^ sgn) - sgn
#pragma unroll
    for (int k = 0; k < kTile / kBlock; k++) {
      float4 p = pv_all[pass][k];
      if (SWAP && !old_pass) {
        // pcl::transformPointCloud: xyz' = M * xyz + t in f32, left to right, no FMA (front_end.hip: k_transform_slot); the
        // rest of the slot is NaN padding
        size_t i = base + (size_t)k * kBlock + threadIdx.x;
        if (i < A.n_raw) {
          if (A.force_w_one) p.w = 1.0f;
          if (A.keep) A.keep[i] = p;
          float4 q;
          q.x = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(A.T[0], p.x), lsa_fmul(A.T[1], p.y)), lsa_fmul(A.T[2], p.z)), A.T[3]);
          q.y = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(A.T[4], p.x), lsa_fmul(A.T[5], p.y)), lsa_fmul(A.T[6], p.z)), A.T[7]);
          q.z = lsa_fadd(lsa_fadd(lsa_fadd(lsa_fmul(A.T[8], p.x), lsa_fmul(A.T[9], p.y)), lsa_fmul(A.T[10], p.z)), A.T[11]);
          q.w = p.w;
          p = q;
        } else {
          p = make_float4(NAN, NAN, NAN, 0.f);
        }
        if (i < n) pts[i] = p;
      }
      if (!lsa_finite3(p.x, p.y, p.z)) continue;
      int ijk[3];
      lm_cell_abs<MODE>(p, A.leaf, A.inv, ijk);
      if (!lm_in_range(ijk)) { atomicOr(&st->overflow, 2u); continue; }   // beyond +-2^20 cells: the owner falls back to the full build
      double o[3];
      lsa_abs_origin(ijk, A.leaf, o);
      unsigned long long v[NS];
      vi_point_terms<NS>(o, A.fs, p, v);
      const unsigned long long key = lm_pack(ijk[0], ijk[1], ijk[2]);
      int slot = lm_lds_insert<kLdsCap>(keys, key);
      if (slot >= 0) {
#pragma unroll
        for (int j = 0; j < NS; j++) atomicAdd(&sums[slot][j], (v[j] ^ sgn) - sgn);
        atomicAdd(&counts[slot], sign);
      } else {  // the tile's table is full (a tile of an already sparse cloud): straight to the global table
        int g = lm_find_insert<NS>(H, key, &n_new_keys, &st->overflow);
        if (g >= 0) {
          unsigned long long *a = H.line((unsigned)g) + 1;
#pragma unroll
          for (int j = 0; j < NS; j++) atomicAdd(a + j, (v[j] ^ sgn) - sgn);
          const unsigned long long old = atomicAdd(a + NS, (unsigned long long)(long long)sign + kLmTouch);
          if (((unsigned)(old >> 32) & kTouchMask) == 0u) lm_dirty_overflow(D, (unsigned)g, &st->overflow);
        }
      }
    }
    if (SWAP) __syncthreads();   // (the slot's old points are read by the lanes that overwrite them: no hazard; this keeps the passes apart in LDS only)
  }
  __syncthreads();
  lm_stamp(stamps, 2);
  lm_flush<NS, kLdsCap>(H, keys, counts, sums, used, gslot, won, &n_used, &n_won, &n_new_keys, &st->overflow, stamps);
  // the tile's region of the dirty list (every region's length is written, the empty ones' too: the consumer does not clear)
  unsigned *reg = D.region(blockIdx.x);
  for (unsigned e = threadIdx.x; e < n_won; e += kBlock) reg[e] = won[e];
  if (threadIdx.x == 0) {
    D.len[1 + blockIdx.x] = n_won;
    if (n_new_keys) atomicAdd(A.to_level_n ? &st->keys_n : &st->keys_f, n_new_keys);
  }
  lm_stamp(stamps, 5);
}

// Scan of a dirty list's region lengths by one workgroup: pre[q] = entries in front of region q (q = 0 the overflow region,
// q = 1 + r region r), pre[R + 1] = the total.  R + 1 <= kMaxRegions + 1.  The lengths are read in ONE coalesced sweep (a lane's
// loads are independent: one round trip), then scanned in LDS.
__device__ __forceinline__ unsigned lm_scan_regions(const DirtyView &D, unsigned *pre, unsigned *wave_sums) {
  const unsigned R1 = D.regions + 1u;
  for (unsigned q = threadIdx.x; q < R1; q += kBlock) {
    const unsigned c = D.len[q];
    const unsigned room = q == 0u ? D.over_cap : D.region_cap;
    pre[q] = c < room ? c : room;
  }
  __syncthreads();
  const unsigned per = (R1 + kBlock - 1u) / kBlock;   // consecutive lengths per lane
  const unsigned q0 = threadIdx.x * per;
  unsigned local = 0u;
  for (unsigned k = 0; k < per; k++)
    if (q0 + k < R1) local += pre[q0 + k];
  unsigned incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    unsigned t = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wave_sums[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned run = incl - local;
  for (unsigned w = 0; w < (threadIdx.x >> 6); w++) run += wave_sums[w];
  for (unsigned k = 0; k < per; k++) {
    const unsigned q = q0 + k;
    if (q < R1) { const unsigned c = pre[q]; pre[q] = run; run += c; }
  }
  if (threadIdx.x == kBlock - 1) pre[R1] = run;
  __syncthreads();
  return pre[R1];
}
// entry i of the concatenated list
__device__ __forceinline__ unsigned lm_dirty_entry(const DirtyView &D, const unsigned *pre, unsigned i) {
  unsigned lo = 0u, hi = D.regions + 1u;   // largest q with pre[q] <= i (lengths may be 0: pre is non-decreasing)
  while (hi - lo > 1u) {
    const unsigned mid = (lo + hi) >> 1;
    if (pre[mid] <= i) lo = mid; else hi = mid;
  }
  const unsigned off = i - pre[lo];
  return lo == 0u ? D.slot[off] : D.region(lo - 1u)[off];
}

struct FilterDirtyArgs {
  float leaf_f[3];
  float leaf_n[3], inv_n[3];
  FixScales fs_f, fs_n;
};

// One workgroup per TILE region of the level-F dirty list (plus a few that share the overflow region), up to kFdPer touched cells
// per lane, everything a lane needs of ALL its cells requested before the first is looked at: the region's length and the entries
// in one round trip (the entries are read whatever the length says), the cells' lines in the next.  A tile's cells are spatial
// neighbours (in the order the scan first touches them), so the workgroup's cells fall into a few dozen level-N cells: their
// changes are summed per N cell in an LDS table first and leave the workgroup as one line's worth of atomics per N cell.
constexpr int kFdCap = 256;    // LDS slots (N cells) of a workgroup; what does not fit goes straight to memory
constexpr int kFdPer = 2;      // cells per lane and step: a swap tile's region holds up to 512
constexpr unsigned kFdOverWg = 8;   // workgroups that share the overflow region (usually empty)
__global__ __launch_bounds__(kBlock) void k_lm_filter_dirty(TableView<4> F, DirtyView DF, TableView<10> N, DirtyView DN, FilterDirtyArgs A, LmStatus *__restrict__ st,
                                                            unsigned long long *stamps) {
  lm_stamp(stamps, 0);
  __shared__ unsigned long long keys[kFdCap];
  __shared__ int counts[kFdCap];
  __shared__ unsigned long long sums[kFdCap][10];
  __shared__ int gslot[kFdCap];
  __shared__ int used[kFdCap];
  __shared__ unsigned won[kFdCap];
  __shared__ int n_used;
  __shared__ unsigned n_won, n_new_keys;
  const bool over = blockIdx.x >= DF.regions;
  const unsigned *list = over ? DF.slot : DF.region(blockIdx.x);
  const unsigned room = over ? DF.over_cap : DF.region_cap;
  unsigned nd = over ? DF.len[0] : DF.len[1 + blockIdx.x];
  nd = nd < room ? nd : room;
  const unsigned step = kBlock * kFdPer;
  unsigned fill = 0u;   // entries of this workgroup's level-N region so far (uniform)
  int d_live = 0;
  for (unsigned base = over ? (blockIdx.x - DF.regions) * step : 0u; base < nd; base += over ? kFdOverWg * step : step) {
    if (threadIdx.x == 0) { n_used = 0; n_won = 0u; n_new_keys = 0u; }
    for (int k = threadIdx.x; k < kFdCap; k += kBlock) {
      keys[k] = kEmpty;
      counts[k] = 0;
#pragma unroll
      for (int j = 0; j < 10; j++) sums[k][j] = 0ull;
    }
    __syncthreads();
    unsigned s[kFdPer];
    bool live[kFdPer];
    ulonglong2 q0[kFdPer], q1[kFdPer], q2[kFdPer];
    float4 oldp[kFdPer];
#pragma unroll
    for (int m = 0; m < kFdPer; m++) {
      const unsigned t = base + (unsigned)m * kBlock + threadIdx.x;
      live[m] = t < nd;
      s[m] = live[m] ? list[t] : 0u;
    }
#pragma unroll
    for (int m = 0; m < kFdPer; m++) {
      // the cell's line: key | 4 sums | count word | output point  (idle lanes read line 0: harmless)
      const ulonglong2 *ln = reinterpret_cast<const ulonglong2 *>(F.line(s[m]));
      q0[m] = ln[0]; q1[m] = ln[1]; q2[m] = ln[2];
      oldp[m] = reinterpret_cast<const float4 *>(F.line(s[m]))[3];
    }
    if (base == 0u || over) lm_stamp(stamps, 1);
#pragma unroll
    for (int m = 0; m < kFdPer; m++) {
      if (!live[m]) continue;
      const float4 old = oldp[m];
      const long long a[4] = {(long long)q0[m].y, (long long)q1[m].x, (long long)q1[m].y, (long long)q2[m].x};
      const unsigned long long cw = q2[m].y;
      const int cnt = (int)(unsigned)cw;
      const bool had = (cw & kLmHasOut) != 0ull, has = cnt > 0;
      float4 nw = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has) {
        int ijk[3];
        lm_unpack(q0[m].x, ijk);
        double o[3];
        lsa_abs_origin(ijk, A.leaf_f, o);
        nw = vi_filter_mean(a, cnt, o, A.fs_f);
      }
      d_live += (int)has - (int)had;
      const bool same = had && has && __float_as_uint(old.x) == __float_as_uint(nw.x) && __float_as_uint(old.y) == __float_as_uint(nw.y) &&
                        __float_as_uint(old.z) == __float_as_uint(nw.z) && __float_as_uint(old.w) == __float_as_uint(nw.w);
      // the count word back to "clean" (no contributions of a running update), with the output point's presence
      F.line(s[m])[5] = (unsigned long long)(unsigned)cnt | (has ? kLmHasOut : 0ull);
      if (same) continue;
      // the old output point leaves level N, the new one enters it
      reinterpret_cast<float4 *>(F.line(s[m]))[3] = nw;
#pragma unroll
      for (int side = 0; side < 2; side++) {
        if (side == 0 ? !had : !has) continue;
        const float4 p = side == 0 ? old : nw;
        const int sign = side == 0 ? -1 : +1;
        const unsigned long long sgn = sign < 0 ? ~0ull : 0ull;
        int c[3];
        lm_cell_abs<1>(p, A.leaf_n, A.inv_n, c);
        if (!lm_in_range(c)) { atomicOr(&st->overflow, 2u); continue; }
        double o[3];
        lsa_abs_origin(c, A.leaf_n, o);
        unsigned long long v[10];
        vi_point_terms<10>(o, A.fs_n, p, v);
        const unsigned long long key = lm_pack(c[0], c[1], c[2]);
        int slot = lm_lds_insert<kFdCap>(keys, key);
        if (slot >= 0) {
#pragma unroll
          for (int j = 0; j < 10; j++) atomicAdd(&sums[slot][j], (v[j] ^ sgn) - sgn);
          atomicAdd(&counts[slot], sign);
        } else {
          int g = lm_find_insert<10>(N, key, &n_new_keys, &st->overflow);
          if (g >= 0) {
            unsigned long long *a2 = N.line((unsigned)g) + 1;
#pragma unroll
            for (int j = 0; j < 10; j++) atomicAdd(a2 + j, (v[j] ^ sgn) - sgn);
            const unsigned long long oldw = atomicAdd(a2 + 10, (unsigned long long)(long long)sign + kLmTouch);
            if (((unsigned)(oldw >> 32) & kTouchMask) == 0u) lm_dirty_overflow(DN, (unsigned)g, &st->overflow);
          }
        }
      }
    }
    __syncthreads();
    lm_stamp(stamps, 2);
    lm_flush<10, kFdCap>(N, keys, counts, sums, used, gslot, won, &n_used, &n_won, &n_new_keys, &st->overflow, stamps);
    // the cells this step was the first toucher of: into the workgroup's region while it has room, the rest into the overflow region
    {
      unsigned *reg = DN.region(blockIdx.x);
      const unsigned nw_ = n_won;
      for (unsigned e = threadIdx.x; e < nw_; e += kBlock) {
        if (fill + e < DN.region_cap) reg[fill + e] = won[e];
        else lm_dirty_overflow(DN, won[e], &st->overflow);
      }
      fill = fill + nw_ < DN.region_cap ? fill + nw_ : DN.region_cap;
      if (threadIdx.x == 0 && n_new_keys) atomicAdd(&st->keys_n, n_new_keys);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) DN.len[1 + blockIdx.x] = fill;
  lm_stamp(stamps, 5);
  // one atomic per wave for the live-cell counter
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d_live += __shfl_down(d_live, off, 64);
  if ((threadIdx.x & 63) == 0 && d_live != 0) atomicAdd(&st->live_f, d_live);
}

struct NdtDirtyArgs {
  float leaf[3];
  FixScales fs;
  int pcl;
  int wmin[3], wdim[3];
  int have_window;
};

// one lane per touched level-N cell: voxel record + dense table entry
__global__ __launch_bounds__(kBlock) void k_lm_ndt_dirty(TableView<10> N, DirtyView DN, unsigned *__restrict__ f_over_len, double *__restrict__ rec, float4 *__restrict__ dense,
                                                         NdtDirtyArgs A, LmStatus *__restrict__ st, LmStatus *__restrict__ host_mirror, unsigned long long epoch,
                                                         unsigned long long *stamps) {
  lm_stamp(stamps, 0);
  __shared__ unsigned pre[kMaxRegions + 2];
  __shared__ unsigned wave_sums[kBlock / 64];
  const unsigned nd = lm_scan_regions(DN, pre, wave_sums);
  lm_stamp(stamps, 1);
  // workgroups that have a chunk (workgroup 0 always): the others leave without a word
  const unsigned chunks = (nd + kBlock - 1u) / kBlock;
  const unsigned active = chunks < gridDim.x ? (chunks ? chunks : 1u) : gridDim.x;
  if (blockIdx.x >= active) return;
  int d_live = 0;
  for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < nd; t += active * kBlock) {
    const unsigned s = lm_dirty_entry(DN, pre, t);
    const ulonglong2 *ln = reinterpret_cast<const ulonglong2 *>(N.line(s));
    ulonglong2 q[7];
#pragma unroll
    for (int k = 0; k < 7; k++) q[k] = ln[k];
    const unsigned long long key = q[0].x, cw = q[5].y;
    const long long a[10] = {(long long)q[0].y, (long long)q[1].x, (long long)q[1].y, (long long)q[2].x, (long long)q[2].y,
                             (long long)q[3].x, (long long)q[3].y, (long long)q[4].x, (long long)q[4].y, (long long)q[5].x};
    const int cnt = (int)(unsigned)cw;
    const int was = (int)(q[6].x >> 32);   // {state, alive} packed in one word: state low, alive high
    int ijk[3];
    lm_unpack(key, ijk);
    lm_stamp(stamps, 2);
    double *R = rec + (size_t)s * 10;
    int stt = 0;
    if (cnt > 0) {
      double o[3];
      lsa_abs_origin(ijk, A.leaf, o);
      stt = ndt_cell_record(a, cnt, o, A.fs, A.pcl, R);
      lm_stamp(stamps, 3);
    } else {
#pragma unroll
      for (int j = 0; j < 10; j++) R[j] = 0.0;
    }
    const int is = cnt > 0 ? 1 : 0;
    // count word back to clean; state and alive
    ulonglong2 tail;
    tail.x = (unsigned long long)(unsigned)stt | ((unsigned long long)(unsigned)is << 32);
    tail.y = 0ull;
    N.line(s)[11] = (unsigned long long)(unsigned)cnt;
    reinterpret_cast<ulonglong2 *>(N.line(s))[6] = tail;
    d_live += is - was;
    if (A.have_window) {
      const int i = ijk[0] - A.wmin[0], j = ijk[1] - A.wmin[1], k = ijk[2] - A.wmin[2];
      const bool inside = (unsigned)i < (unsigned)A.wdim[0] && (unsigned)j < (unsigned)A.wdim[1] && (unsigned)k < (unsigned)A.wdim[2];
      if (inside) {
        const size_t cell = (size_t)i + (size_t)j * A.wdim[0] + (size_t)k * A.wdim[0] * A.wdim[1];
        dense[cell] = stt == 1 ? ndt_dense_entry(R, ijk, A.leaf, (int)s) : make_float4(__int_as_float(-1), __int_as_float(-1), __int_as_float(-1), __int_as_float(-1));
      } else if (stt == 1) {
        atomicOr(&st->window_miss, 1u);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d_live += __shfl_down(d_live, off, 64);
  if ((threadIdx.x & 63) == 0 && d_live != 0) atomicAdd(&st->live_n, d_live);
  // The LAST active workgroup to finish publishes the status block to the host and resets the overflow regions' lengths.  Every
  // word of the block is only ever changed by device-scope atomics, which are performed at the memory side, so no cache
  // write-back is involved: each wave drains its own outstanding operations (s_waitcnt), the workgroup meets, one lane counts
  // it done; the workgroup that finds itself last reads the words with device-scope loads.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  lm_stamp(stamps, 4);
  __shared__ unsigned s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(&st->done, 1u) == active - 1u ? 1u : 0u;
  __syncthreads();
  lm_stamp(stamps, 5);
  if (s_last && threadIdx.x < 64) {
    // one lane per 32-bit word of the block in front of the epoch, all loads in ONE instruction
    constexpr int kW = (int)(offsetof(LmStatus, epoch) / 4);
    constexpr int kDone = (int)(offsetof(LmStatus, done) / 4);
    const int lane = (int)threadIdx.x;
    unsigned *sw = reinterpret_cast<unsigned *>(st), *hw = reinterpret_cast<unsigned *>(host_mirror);
    const bool reset = lane == kDone;
    unsigned v = 0u;
    if (lane < kW && !reset) v = __hip_atomic_load(sw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < kW) __hip_atomic_store(hw + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (reset) __hip_atomic_store(sw + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 62) __hip_atomic_store(DN.len, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 63 && f_over_len) __hip_atomic_store(f_over_len, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the epoch word last, behind the completion of the others (a drain of the wave's stores: a RELEASE at system scope
    // would write back the whole L2 — the records and table entries this kernel has just written — for nothing)
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) __hip_atomic_store(&host_mirror->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    lm_stamp(stamps, 6);
  }
}

// a rebuild's first launch: the status block zeroed, the sentinels the window re-fit reduces the bounds into (k_lm_bounds leaves them
// in place again for the next re-fit), the overflow regions' lengths of both dirty lists zeroed.  One kernel instead of three copy
// commands: the runtime's fill path costs the host 6-26 us a call, a launch 4-5.
__global__ void k_lm_reset(LmStatus *__restrict__ st, unsigned *__restrict__ len_f, unsigned *__restrict__ len_n) {
  unsigned *w = reinterpret_cast<unsigned *>(st);
  for (unsigned i = threadIdx.x; i < sizeof(LmStatus) / sizeof(unsigned); i += blockDim.x) w[i] = 0u;
  __syncthreads();
  if (threadIdx.x < 3) { st->bmin[threadIdx.x] = INT_MAX; st->bmax[threadIdx.x] = INT_MIN; }
  if (threadIdx.x == 3 && len_f) len_f[0] = 0u;
  if (threadIdx.x == 4 && len_n) len_n[0] = 0u;
}
// 16-byte fill (the dense window's "no voxel" entries are all-ones)
__global__ __launch_bounds__(kBlock) void k_lm_fill16(uint4 *__restrict__ p, size_t n16, unsigned v) {
  const uint4 q = make_uint4(v, v, v, v);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBlock) p[i] = q;
}

// bounds of the valid voxels (window re-fit); the LAST workgroup to finish hands them to the host through the host-mapped status block
// (no copy command; the host waits for the stream once)
__global__ __launch_bounds__(kBlock) void k_lm_bounds(const LmCellN *__restrict__ cells, unsigned cap, LmStatus *__restrict__ st, LmStatus *__restrict__ host_mirror) {
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += gridDim.x * blockDim.x) {
    if (cells[s].key == kEmpty || cells[s].state != 1) continue;
    int ijk[3];
    lm_unpack(cells[s].key, ijk);
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], ijk[a]); mx[a] = max(mx[a], ijk[a]); }
  }
  // wave, then workgroup (LDS), then ONE set of six atomics per workgroup: the valid voxels are hash-spread over the table, so every
  // wave has some, and six same-address atomics from each of 2 048 waves were most of this kernel (46 us in the trace of the cold
  // front-end drives against ~5 us for reading the 16 MB; the memory side serialises on an address)
  __shared__ int s_mn[kBlock / 64][3], s_mx[kBlock / 64][3];
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[a] = min(mn[a], __shfl_down(mn[a], off, 64)); mx[a] = max(mx[a], __shfl_down(mx[a], off, 64)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][a] = mn[a]; s_mx[threadIdx.x >> 6][a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = (int)threadIdx.x;
    int lo = INT_MAX, hi = INT_MIN;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { lo = min(lo, s_mn[w][a]); hi = max(hi, s_mx[w][a]); }
    if (lo != INT_MAX) atomicMin(&st->bmin[a], lo);
    if (hi != INT_MIN) atomicMax(&st->bmax[a], hi);
  }
  // the issuing wave drains its own atomics (they are performed at the memory side: nothing to write back, a wait for the wave's
  // outstanding operations is enough — an agent-scope fence here wrote back the L2 once per workgroup), then the workgroup counts itself done
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&st->pad, 1u);
    if (t == gridDim.x - 1) {
      st->pad = 0u;
      for (int a = 0; a < 3; a++) {
        // (exchanged for the sentinels: the block is ready for the next re-fit, no initialising launch in front of this kernel)
        const int lo = __hip_atomic_exchange(&st->bmin[a], INT_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_exchange(&st->bmax[a], INT_MIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&host_mirror->bmin[a], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_mirror->bmax[a], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_lm_dense_fill(const LmCellN *__restrict__ cells, const double *__restrict__ rec, unsigned cap, NdtDirtyArgs A,
                                                          float4 *__restrict__ dense, LmStatus *__restrict__ st) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) st->window_miss = 0u;   // the flag that asked for this re-fit is consumed
  if (s >= cap || cells[s].key == kEmpty || cells[s].state != 1) return;
  int ijk[3];
  lm_unpack(cells[s].key, ijk);
  const int i = ijk[0] - A.wmin[0], j = ijk[1] - A.wmin[1], k = ijk[2] - A.wmin[2];
  if ((unsigned)i >= (unsigned)A.wdim[0] || (unsigned)j >= (unsigned)A.wdim[1] || (unsigned)k >= (unsigned)A.wdim[2]) return;
  dense[(size_t)i + (size_t)j * A.wdim[0] + (size_t)k * A.wdim[0] * A.wdim[1]] = ndt_dense_entry(rec + (size_t)s * 10, ijk, A.leaf, (int)s);
}

// verification (tests): every occupied cell of a FULL table build must be in level N with the same state and the same ten
// doubles, bit for bit; out[0] cells compared, [1] mismatches
__global__ __launch_bounds__(kBlock) void k_lm_verify(const LmCellN *__restrict__ cells, unsigned cap, const double *__restrict__ rec, const int32_t *__restrict__ full_cor,
                                                      const double *__restrict__ full_rec, const int32_t *__restrict__ full_state, size_t V, CellMap m,
                                                      unsigned long long *__restrict__ out) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  int ijk[3];
  lsa_cell_unravel(m, full_cor[r], ijk);
  const unsigned long long key = lm_pack(ijk[0], ijk[1], ijk[2]);
  unsigned h = lm_probe0(key, cap);
  int slot = -1;
  for (int probe = 0; probe < kGlobalProbes; probe++) {
    unsigned long long k = cells[h].key;
    if (k == key) { slot = (int)h; break; }
    if (k == kEmpty) break;
    h = (h + kProbeStep) & (cap - 1u);
  }
  bool same = slot >= 0 && (int)(unsigned)cells[slot].cnt > 0 && (cells[slot].cnt >> 32) == 0ull && cells[slot].state == full_state[r];
  if (same)
    for (int j = 0; j < 10; j++)
      same = same && __double_as_longlong(rec[(size_t)slot * 10 + j]) == __double_as_longlong(full_rec[r * 10 + j]);
  atomicAdd(out, 1ull);
  if (!same) atomicAdd(out + 1, 1ull);
}
__global__ __launch_bounds__(kBlock) void k_lm_count(const LmCellN *__restrict__ cells, unsigned cap, unsigned long long *__restrict__ out) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  if (cells[s].alive) atomicAdd(out + 2, 1ull);
  if (cells[s].alive && cells[s].state == 1) atomicAdd(out + 3, 1ull);
}

TableView<4> view_f(LocalMap &lm) { return TableView<4>{reinterpret_cast<unsigned long long *>(lm.f.p), lm.cap_f}; }
TableView<10> view_n(LocalMap &lm) { return TableView<10>{reinterpret_cast<unsigned long long *>(lm.n.p), lm.cap_n}; }
DirtyView view(const LmDirty &d) { return DirtyView{d.slot.p, d.len.p, d.regions, d.region_cap, d.over_cap}; }

// geometry of a dirty list for the next producing launch; (re)allocates when it does not fit.  The overflow region's length
// (len[0]) is zero between updates: the consumer's last workgroup resets it.
int dirty_shape(lsa_context *ctx, LmDirty &d, unsigned regions, unsigned region_cap, unsigned over_cap) {
  int rc;
  const size_t need = (size_t)over_cap + (size_t)regions * region_cap;
  const bool fresh = need > d.slot.cap || (size_t)regions + 1 > d.len.cap || over_cap != d.over_cap;
  if (fresh) {
    // (nothing is in flight that reads the old buffers: a list is reshaped between updates only, behind lm_check / a rebuild's wait)
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (need > d.slot.cap) { d.slot.release(); if ((rc = d.slot.reserve(ctx, need, 1.25))) return rc; }
    if ((size_t)regions + 1 > d.len.cap) { d.len.release(); if ((rc = d.len.reserve(ctx, (size_t)kMaxRegions + 1))) return rc; }
    LSA_HIP(ctx, hipMemsetAsync(d.len.p, 0, d.len.cap * sizeof(unsigned), ctx->stream));
  }
  d.regions = regions; d.region_cap = region_cap; d.over_cap = over_cap;
  return LSA_OK;
}

// diagnostics: where kernel `which` (0 points, 1 filter_dirty, 2 ndt_dirty) of the NEXT update notes its phase boundaries; null unless enabled
unsigned long long *lm_stamps(LocalMap &lm, int which) { return lm.stamps.p ? lm.stamps.p + (size_t)which * kMaxRegions * 8 : nullptr; }

NdtDirtyArgs ndt_args(const LocalMap &lm, bool with_window) {
  NdtDirtyArgs A;
  for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.wmin[a] = lm.wmin[a]; A.wdim[a] = lm.wdim[a]; }
  A.fs = lm.fs_n;
  A.pcl = lm.pcl;
  A.have_window = (with_window && lm.window_set) ? 1 : 0;
  return A;
}

int alloc_level_n(LocalMap &lm, unsigned cap) {
  lsa_context *ctx = lm.ctx;
  int rc;
  if (cap != lm.cap_n) { lm.n.release(); lm.rec.release(); lm.n_zero = false; }
  if ((rc = lm.n.reserve(ctx, cap)) || (rc = lm.rec.reserve(ctx, (size_t)cap * 10))) return rc;
  lm.cap_n = cap;
  return LSA_OK;
}

int alloc_level_f(LocalMap &lm, unsigned cap) {
  lsa_context *ctx = lm.ctx;
  int rc;
  if (cap != lm.cap_f) { lm.f.release(); lm.f_zero = false; }
  if ((rc = lm.f.reserve(ctx, cap))) return rc;
  lm.cap_f = cap;
  return LSA_OK;
}

// one launch of k_lm_points over `n` points (plain: sign +-1) with the dirty list shaped for its tiles; at most kMaxTiles tiles
int lm_points(LocalMap &lm, const float4 *d_pts, size_t n, int sign) {
  lsa_context *ctx = lm.ctx;
  if (n == 0) return LSA_OK;
  PointsArgs A;
  A.sign = sign;
  A.evict = 0; A.raw = nullptr; A.n_raw = 0; A.keep = nullptr; A.force_w_one = 0;
  for (int k = 0; k < 12; k++) A.T[k] = 0.f;
  const unsigned g = lsa_div_up(n, kTile);
  int rc;
  if (lm.filtered) {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_f[a]; A.inv[a] = 1.0f / lm.leaf_f[a]; }
    A.fs = lm.fs_f;
    A.to_level_n = 0;
    if ((rc = dirty_shape(ctx, lm.dirty_f, g, PointsCap<4, false>::value, lm.cap_f))) return rc;
    hipLaunchKernelGGL((k_lm_points<4, 0, false>), dim3(g), dim3(kBlock), 0, ctx->stream, const_cast<float4 *>(d_pts), n, A, view_f(lm), view(lm.dirty_f), lm.d_status.p, (unsigned long long *)nullptr);
  } else {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.inv[a] = 1.0f / lm.leaf_n; }
    A.fs = lm.fs_n;
    A.to_level_n = 1;
    if ((rc = dirty_shape(ctx, lm.dirty_n, g, PointsCap<10, false>::value, lm.cap_n))) return rc;
    hipLaunchKernelGGL((k_lm_points<10, 1, false>), dim3(g), dim3(kBlock), 0, ctx->stream, const_cast<float4 *>(d_pts), n, A, view_n(lm), view(lm.dirty_n), lm.d_status.p, (unsigned long long *)nullptr);
  }
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}
}  // namespace

int lm_configure(LocalMap &lm, lsa_context *ctx, const float leaf_f[3], float leaf_n, int pcl, unsigned long long scale_n) {
  lm.ctx = ctx;
  for (int a = 0; a < 3; a++) lm.leaf_f[a] = leaf_f[a];
  lm.leaf_n = leaf_n;
  lm.pcl = pcl;
  const double lf = fmax(fmax((double)leaf_f[0], (double)leaf_f[1]), (double)leaf_f[2]);
  // exactly the scales a full build with VoxelIndex::scale_n_min = scale_n derives (k_bbox_final_params)
  lm.fs_f = vi_make_scales((double)scale_n, lf, nullptr);
  lm.fs_n = vi_make_scales((double)scale_n, (double)leaf_n, nullptr);
  lm.valid = false;
  int rc;
  if ((rc = lm.d_status.reserve(ctx, 1)) || (rc = lm.h_status.reserve(ctx, 1))) return rc;
  if (!lm.cap_n && (rc = alloc_level_n(lm, 1u << 17))) return rc;
  return LSA_OK;
}

// Everything a session with the default capacities allocates on the way — the status blocks at the first key-frame, level N, its
// records and dirty list, the dense window at the first re-fit, level F and its dirty list at the switch to the filtered map (the
// tenth key-frame) — allocated now instead: ~120 MB.  Tables that turn out too small are still grown by lm_rebuild.
int lm_reserve(LocalMap &lm, lsa_context *ctx, bool with_level_f) {
  lm.ctx = ctx;
  int rc;
  if ((rc = lm.d_status.reserve(ctx, 1)) || (rc = lm.h_status.reserve(ctx, 1))) return rc;
  if (!lm.cap_n) {
    if ((rc = alloc_level_n(lm, 1u << 17))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(lm.n.p, 0, (size_t)lm.cap_n * sizeof(LmCellN), ctx->stream));
    lm.n_zero = true;
  }
  if (with_level_f && !lm.cap_f) {
    if ((rc = alloc_level_f(lm, 1u << 20))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(lm.f.p, 0, (size_t)lm.cap_f * sizeof(LmCellF), ctx->stream));
    lm.f_zero = true;
  }
  // the dirty lists at the largest shape a launch may ask for (kMaxRegions regions of 512 entries + the overflow region)
  if ((rc = dirty_shape(ctx, lm.dirty_n, kMaxRegions, 512, lm.cap_n))) return rc;
  if (with_level_f && (rc = dirty_shape(ctx, lm.dirty_f, kMaxRegions, 512, lm.cap_f))) return rc;
  if (!lm.dense.p && (rc = lm.dense.reserve(ctx, (size_t)2 << 20))) return rc;
  return LSA_OK;
}

int lm_swap(LocalMap &lm, float4 *d_slot, size_t slot_cap, bool evict, const float4 *d_raw, size_t n_raw, const float T_rowmajor12[12],
            float4 *d_keep, bool force_w_one) {
  lsa_context *ctx = lm.ctx;
  if (slot_cap == 0) return LSA_OK;
  PointsArgs A;
  A.sign = +1;
  A.evict = evict ? 1 : 0;
  A.raw = d_raw; A.n_raw = n_raw;
  for (int k = 0; k < 12; k++) A.T[k] = T_rowmajor12[k];
  A.keep = d_keep; A.force_w_one = force_w_one ? 1 : 0;
  const unsigned g = lsa_div_up(slot_cap, kTile);
  if (g > kMaxTiles) return lsa_fail(ctx, LSA_ERR_CAPACITY, "incremental local map: a key-frame of %zu points exceeds %u tiles", slot_cap, kMaxTiles);
  int rc;
  if (lm.filtered) {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_f[a]; A.inv[a] = 1.0f / lm.leaf_f[a]; }
    A.fs = lm.fs_f;
    A.to_level_n = 0;
    if ((rc = dirty_shape(ctx, lm.dirty_f, g, PointsCap<4, true>::value, lm.cap_f))) return rc;
    hipLaunchKernelGGL((k_lm_points<4, 0, true>), dim3(g), dim3(kBlock), 0, ctx->stream, d_slot, slot_cap, A, view_f(lm), view(lm.dirty_f), lm.d_status.p, lm_stamps(lm, 0));
  } else {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.inv[a] = 1.0f / lm.leaf_n; }
    A.fs = lm.fs_n;
    A.to_level_n = 1;
    if ((rc = dirty_shape(ctx, lm.dirty_n, g, PointsCap<10, true>::value, lm.cap_n))) return rc;
    hipLaunchKernelGGL((k_lm_points<10, 1, true>), dim3(g), dim3(kBlock), 0, ctx->stream, d_slot, slot_cap, A, view_n(lm), view(lm.dirty_n), lm.d_status.p, lm_stamps(lm, 0));
  }
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

int lm_finish(LocalMap &lm) {
  int rc = lm_finish_begin(lm);
  return rc ? rc : lm_finish_end(lm);
}

// first half of an update's tail: the dirty cells of the filter level carried into the sums of level N (filtered map only).  Nothing a
// registration pass reads is written yet — the records and the probe table are lm_finish_end's — so a pass still in flight on another
// stream is not disturbed by anything up to here
int lm_finish_begin(LocalMap &lm) {
  lsa_context *ctx = lm.ctx;
  int rc;
  if (lm.filtered) {
    FilterDirtyArgs A;
    for (int a = 0; a < 3; a++) { A.leaf_f[a] = lm.leaf_f[a]; A.leaf_n[a] = lm.leaf_n; A.inv_n[a] = 1.0f / lm.leaf_n; }
    A.fs_f = lm.fs_f; A.fs_n = lm.fs_n;
    // one workgroup per tile region + a few for the overflow region; each owns a region of the level-N dirty list
    const unsigned g = lm.dirty_f.regions + kFdOverWg;
    if ((rc = dirty_shape(ctx, lm.dirty_n, g, kFdRegion, lm.cap_n))) return rc;
    hipLaunchKernelGGL(k_lm_filter_dirty, dim3(g), dim3(kBlock), 0, ctx->stream, view_f(lm), view(lm.dirty_f), view_n(lm), view(lm.dirty_n), A, lm.d_status.p, lm_stamps(lm, 1));
    LSA_HIP(ctx, hipGetLastError());
  }
  return LSA_OK;
}

// second half: the records and the probe-table entries of the dirty level-N cells, the status to the host
int lm_finish_end(LocalMap &lm) {
  lsa_context *ctx = lm.ctx;
  lm.epoch++;
  // (a small grid: a key-frame touches ~6 k level-N cells, one per lane of 24 workgroups, and every ACTIVE workgroup ends with one
  // same-address atomic — ~20 ns each at the memory side)
  hipLaunchKernelGGL(k_lm_ndt_dirty, dim3(64), dim3(kBlock), 0, ctx->stream, view_n(lm), view(lm.dirty_n), lm.filtered ? lm.dirty_f.len.p : nullptr, lm.rec.p,
                     lm.dense.p, ndt_args(lm, true), lm.d_status.p, lm.h_status.p, lm.epoch, lm_stamps(lm, 2));
  LSA_HIP(ctx, hipGetLastError());
  lm.pending = true;
  lm.updates++;
  return LSA_OK;
}

namespace {
// window re-fit, first half: the bounds of the valid voxels, reduced into the status block's sentinels (set by k_lm_reset, left in place
// again by the kernel's last workgroup) and handed to the host through the host-mapped block
int lm_bounds_launch(LocalMap &lm) {
  lsa_context *ctx = lm.ctx;
  // (one cell per thread: 16 MB of table lines at the default capacity, ~5 us over 512 workgroups; 128 workgroups of 4 cells per thread took 17)
  hipLaunchKernelGGL(k_lm_bounds, dim3(lsa_div_up(lm.cap_n, kBlock)), dim3(kBlock), 0, ctx->stream, lm.n.p, lm.cap_n, lm.d_status.p, lm.h_status.p);
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}
int lm_window_from_bounds(LocalMap &lm, bool wait = true);
}  // namespace

int lm_refit_window(LocalMap &lm) {
  int rc;
  if ((rc = lm_bounds_launch(lm))) return rc;
  LSA_HIP(lm.ctx, hipStreamSynchronize(lm.ctx->stream));
  return lm_window_from_bounds(lm);
}

namespace {
// second half, behind a host wait for the bounds: the dense window sized, cleared and filled
int lm_window_from_bounds(LocalMap &lm, bool wait) {
  lsa_context *ctx = lm.ctx;
  LmStatus got = *lm.h_status.p;   // (bmin / bmax: written by the kernel's last workgroup)
  int mn[3], dm[3];
  // margin: 24 m of travel horizontally (a dozen key-frames at the 2 m default), 8 m vertically, in cells
  const int mxy = (int)ceilf(24.0f / lm.leaf_n), mz = (int)ceilf(8.0f / lm.leaf_n);
  double nc = 1.0;
  for (int a = 0; a < 3; a++) {
    const int pad = a < 2 ? mxy : mz;
    if (got.bmin[a] > got.bmax[a]) { mn[a] = -pad; dm[a] = 2 * pad + 1; }   // no valid voxel at all
    else { mn[a] = got.bmin[a] - pad; dm[a] = got.bmax[a] - got.bmin[a] + 1 + 2 * pad; }
    nc *= (double)dm[a];
  }
  if (nc > (double)lm.dense_max_cells) return lsa_fail(ctx, LSA_ERR_CAPACITY, "incremental local map: the dense window of %d x %d x %d cells exceeds the limit", dm[0], dm[1], dm[2]);
  int rc;
  const size_t cells = (size_t)nc + 2;   // a row load of the pass kernel reads 3 entries from any in-box cell
  if (cells > lm.dense.cap) {
    lm.dense.release();
    if ((rc = lm.dense.reserve(ctx, cells, 1.25))) return rc;
  }
  for (int a = 0; a < 3; a++) { lm.wmin[a] = mn[a]; lm.wdim[a] = dm[a]; }
  lm.window_set = true;
  {
    size_t g = (cells + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_lm_fill16, dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, reinterpret_cast<uint4 *>(lm.dense.p), cells, 0xFFFFFFFFu);
  }
  hipLaunchKernelGGL(k_lm_dense_fill, dim3(lsa_div_up(lm.cap_n, kBlock)), dim3(kBlock), 0, ctx->stream, lm.n.p, lm.rec.p, lm.cap_n, ndt_args(lm, true), lm.dense.p, lm.d_status.p);
  LSA_HIP(ctx, hipGetLastError());
  // (the table is read next on the REGISTRATION's stream, which is not necessarily this one: the re-fit ends with a wait — unless the
  // caller orders that stream behind this one by an event of its own.  The host-side flag below is the mirror's: nothing on the device
  // writes the mirror before the next update's last kernel.)
  if (wait) LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  lm.h_status.p->window_miss = 0;
  lm.refits++;
  return LSA_OK;
}
}  // namespace

int lm_rebuild(LocalMap &lm, const float4 *const *d_ranges, const size_t *n_ranges, int nranges, bool filtered, bool leave_in_flight) {
  lsa_context *ctx = lm.ctx;
  int rc;
  lm.valid = false;
  lm.filtered = filtered;
  unsigned cap_f = lm.cap_f ? lm.cap_f : (1u << 20), cap_n = lm.cap_n ? lm.cap_n : (1u << 17);
  for (int attempt = 0; attempt < 8; attempt++) {
    if ((rc = alloc_level_n(lm, cap_n))) return rc;
    if (filtered && (rc = alloc_level_f(lm, cap_f))) return rc;
    // (tables fresh from lm_reserve are zero already: the first build of a session — and the first filtered one — skip the 16 / 64 MB fill)
    static_assert(sizeof(LmCellN) % 16 == 0 && sizeof(LmCellF) % 16 == 0, "the tables are cleared 16 bytes at a time");
    if (!lm.n_zero) hipLaunchKernelGGL(k_lm_fill16, dim3(2048), dim3(kBlock), 0, ctx->stream, reinterpret_cast<uint4 *>(lm.n.p), (size_t)lm.cap_n * (sizeof(LmCellN) / 16), 0u);
    if (filtered && !lm.f_zero) hipLaunchKernelGGL(k_lm_fill16, dim3(2048), dim3(kBlock), 0, ctx->stream, reinterpret_cast<uint4 *>(lm.f.p), (size_t)lm.cap_f * (sizeof(LmCellF) / 16), 0u);
    lm.n_zero = false;
    if (filtered) lm.f_zero = false;
    hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(64), 0, ctx->stream, lm.d_status.p, lm.dirty_f.len.p, lm.dirty_n.len.p);
    LSA_HIP(ctx, hipGetLastError());
    // every cell ends up in the dirty lists: finalise them all (no window yet: the table is filled by the re-fit below), in rounds
    // of at most kMaxTiles tiles — a round's dirty list has one region per tile
    lm.window_set = false;
    for (int r = 0; r < nranges; r++) {
      const size_t round = (size_t)kMaxTiles * kTile;
      for (size_t at = 0; at < n_ranges[r]; at += round) {
        const size_t m = n_ranges[r] - at < round ? n_ranges[r] - at : round;
        if ((rc = lm_points(lm, d_ranges[r] + at, m, +1))) return rc;
        if ((rc = lm_finish(lm))) return rc;
      }
    }
    // (the re-fit's bounds ride behind the build: one host wait serves the build's status and the window's bounds; a build that has to be
    // repeated larger throws them away)
    if ((rc = lm_bounds_launch(lm))) return rc;
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const LmStatus &s = *lm.h_status.p;
    if (s.overflow & 2u) return lsa_fail(ctx, LSA_ERR_RANGE, "incremental local map: a point lies beyond +-2^20 cells");
    const bool grow_f = filtered && (s.overflow || s.keys_f > lm.cap_f / 4), grow_n = s.overflow || s.keys_n > lm.cap_n / 4;
    if (!grow_f && !grow_n) break;
    if (attempt == 7 || (grow_f && cap_f >= (1u << 26)) || (grow_n && cap_n >= (1u << 24)))
      return lsa_fail(ctx, LSA_ERR_CAPACITY, "incremental local map: %u / %u cells do not fit the tables", s.keys_f, s.keys_n);
    if (grow_f) cap_f *= 2;
    if (grow_n) cap_n *= 2;
  }
  if ((rc = lm_window_from_bounds(lm, !leave_in_flight))) return rc;
  lm.pending = false;
  lm.valid = true;
  lm.rebuilds++;
  return LSA_OK;
}

int lm_stamps_enable(LocalMap &lm, bool on) {
  if (!on) { lm.stamps.release(); return LSA_OK; }
  int rc = lm.stamps.reserve(lm.ctx, (size_t)3 * kMaxRegions * 8);
  if (rc) return rc;
  LSA_HIP(lm.ctx, hipMemsetAsync(lm.stamps.p, 0, (size_t)3 * kMaxRegions * 8 * sizeof(unsigned long long), lm.ctx->stream));
  return LSA_OK;
}
int lm_stamps_fetch(LocalMap &lm, unsigned long long *out, size_t cap, size_t *per_kernel) {
  if (per_kernel) *per_kernel = (size_t)kMaxRegions * 8;
  if (!lm.stamps.p) return LSA_ERR_INVALID;
  size_t n = (size_t)3 * kMaxRegions * 8;
  if (n > cap) n = cap;
  LSA_HIP(lm.ctx, hipMemcpyAsync(out, lm.stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, lm.ctx->stream));
  LSA_HIP(lm.ctx, hipStreamSynchronize(lm.ctx->stream));
  LSA_HIP(lm.ctx, hipMemsetAsync(lm.stamps.p, 0, (size_t)3 * kMaxRegions * 8 * sizeof(unsigned long long), lm.ctx->stream));
  return LSA_OK;
}

int lm_check(LocalMap &lm) {
  if (!lm.pending) return 0;
  const LmStatus &s = *lm.h_status.p;
  if (__atomic_load_n(&lm.h_status.p->epoch, __ATOMIC_ACQUIRE) != lm.epoch) {
    // The update is still running (a batch of Updates filters on another stream, so nothing has ordered the host behind it):
    // its last kernel writes the epoch, ~40 us after the key-frame.  Past any plausible run time: ask the runtime.
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(&lm.h_status.p->epoch, __ATOMIC_ACQUIRE) != lm.epoch) {
      __builtin_ia32_pause();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) { (void)hipStreamSynchronize(lm.ctx->stream); break; }
    }
  }
  lm.pending = false;
  if (s.overflow) return 2;
  if ((lm.filtered && s.keys_f > (lm.cap_f / 10) * 6) || s.keys_n > (lm.cap_n / 10) * 6) return 2;
  if (s.window_miss) return 1;
  return 0;
}

int lm_verify(LocalMap &lm, const int32_t *full_cell_of_rank, const double *full_rec, const int32_t *full_state, size_t V, const CellMap &map,
              unsigned long long out[4]) {
  lsa_context *ctx = lm.ctx;
  unsigned long long *d_out = nullptr;
  LSA_HIP(ctx, hipMalloc((void **)&d_out, 4 * sizeof(unsigned long long)));
  (void)hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), ctx->stream);
  if (V) hipLaunchKernelGGL(k_lm_verify, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, lm.n.p, lm.cap_n, lm.rec.p, full_cell_of_rank, full_rec, full_state, V, map, d_out);
  hipLaunchKernelGGL(k_lm_count, dim3(lsa_div_up(lm.cap_n, kBlock)), dim3(kBlock), 0, ctx->stream, lm.n.p, lm.cap_n, d_out);
  hipError_t e = hipMemcpyAsync(out, d_out, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_out);
  LSA_HIP(ctx, e);
  return LSA_OK;
}
