// local_map.hip — the front end's incremental sliding local map: see local_map.h for what it replaces
// (FrontEnd::UpdateWithNewFrame, front_end.cpp:348-424) and for the design.
//
// Kernels (all HBM / atomic bound streaming work, nothing GEMM shaped):
//   k_lm_points<NS, MODE>   one 512-point tile per workgroup: the tile's points are pre-aggregated per cell in an LDS table
//                           (consecutive lidar returns share cells), then each distinct cell finds or claims its slot in the
//                           global table (64-bit CAS on the key) and receives NS + 1 atomics; level F (NS 4, PCL's cell rule)
//                           or, while the map is unfiltered, level N directly (NS 10, the reference VoxelGrid's cell rule).
//                           Bytes per point: 16 read; per distinct (tile, cell): 8 (key) + 8 NS + 4 + 4 of atomic traffic.
//   k_lm_filter_dirty       one lane per touched F cell: new output point from the sums (vi_filter_mean: the VoxelFilter's own
//                           arithmetic), old point out of / new point into level N (one set of atomics when both lie in the
//                           same N cell, which is the rule).
//   k_lm_ndt_dirty          one lane per touched N cell: the voxel record (ndt_cell_record: the full build's own function) and
//                           the dense probe table entry.
//                           the last workgroup to finish publishes the status block into host-mapped memory and resets the
//                           dirty counters.
#include "local_map.h"

#include <chrono>

#include "ndt_cell_dev.h"

namespace {
constexpr int kBlock = 256;
constexpr int kTile = 512;          // points per workgroup tile
constexpr int kLdsProbes = 8;
constexpr int kGlobalProbes = 512;
constexpr unsigned long long kEmpty = ~0ull;
constexpr int kCoordOff = 1 << 20;  // 21 bits per axis

__host__ __device__ inline unsigned long long lm_pack(int i, int j, int k) {
  return (unsigned long long)(unsigned)(i + kCoordOff) | ((unsigned long long)(unsigned)(j + kCoordOff) << 21) | ((unsigned long long)(unsigned)(k + kCoordOff) << 42);
}
__host__ __device__ inline void lm_unpack(unsigned long long key, int ijk[3]) {
  ijk[0] = (int)(key & 0x1fffffull) - kCoordOff;
  ijk[1] = (int)((key >> 21) & 0x1fffffull) - kCoordOff;
  ijk[2] = (int)((key >> 42) & 0x1fffffull) - kCoordOff;
}
__device__ __forceinline__ bool lm_in_range(const int ijk[3]) {
  return (unsigned)(ijk[0] + kCoordOff) < (1u << 21) && (unsigned)(ijk[1] + kCoordOff) < (1u << 21) && (unsigned)(ijk[2] + kCoordOff) < (1u << 21);
}
__device__ __forceinline__ unsigned lm_hash(unsigned long long key, unsigned cap) {
  return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 32) & (cap - 1u);
}

// Probe sequence of a cell in a GLOBAL table: cells are hashed by 4 x 4 x 2 BLOCK, and a block's 32 cells occupy the 32
// consecutive slots behind the block's hash (a collision moves a cell on by whole groups of 32).  Neighbouring cells — which
// consecutive lidar returns, and the members of one NDT voxel, are — so land in neighbouring rows of the sums: their atomics
// share cache lines instead of each pulling a line of its own from a 70 MB table.
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

struct HashView {
  unsigned long long *keys;
  int32_t *cnt;
  long long *acc;
  unsigned *flag, *dirty;
  unsigned cap;
};

// Slot of `key` in the global table, claiming one if it is new.  Keys never change once written, so a plain (possibly stale,
// the eight L2s are not coherent with each other) read can only err towards "empty", which the CAS then settles.
__device__ __forceinline__ int lm_find_insert(const HashView &H, unsigned long long key, unsigned *n_keys, unsigned *overflow) {
  unsigned h = lm_probe0(key, H.cap);
  for (int probe = 0; probe < kGlobalProbes; probe++) {
    unsigned long long k = H.keys[h];
    if (k == key) return (int)h;
    if (k == kEmpty) {
      unsigned long long old = atomicCAS(H.keys + h, kEmpty, key);
      if (old == kEmpty) { atomicAdd(n_keys, 1u); return (int)h; }
      if (old == key) return (int)h;
    }
    h = (h + kProbeStep) & (H.cap - 1u);
  }
  atomicOr(overflow, 1u);
  return -1;
}

__device__ __forceinline__ void lm_mark_dirty(const HashView &H, int slot, unsigned *n_dirty) {
  if (atomicExch(H.flag + slot, 1u) == 0u) H.dirty[atomicAdd(n_dirty, 1u)] = (unsigned)slot;   // (each slot at most once: the list cannot overflow)
}

// slot of `key` in the tile's LDS table (inserting it), or -1 when no slot is free within kLdsProbes steps
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

struct PointsArgs {
  float leaf[3], inv[3];
  FixScales fs;
  int sign;          // +1 add, -1 subtract (the points of `pts`)
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

template <int NS, int MODE, bool SWAP>
__global__ __launch_bounds__(kBlock) void k_lm_points(float4 *__restrict__ pts, size_t n, PointsArgs A, HashView H, LmStatus *__restrict__ st) {
  constexpr int kLdsCap = SWAP ? 512 : 256;   // a swap tile carries two clouds' points
  __shared__ unsigned long long keys[kLdsCap];
  __shared__ int counts[kLdsCap];
  __shared__ unsigned long long sums[kLdsCap][NS];
  __shared__ int gslot[kLdsCap];
  __shared__ int used[kLdsCap];
  __shared__ unsigned won[kLdsCap];     // global slots this tile was the first to mark dirty
  __shared__ int n_used;
  __shared__ unsigned n_won, n_new_keys, dirty_base;
  if (threadIdx.x == 0) { n_used = 0; n_won = 0u; n_new_keys = 0u; }
  for (int k = threadIdx.x; k < kLdsCap; k += kBlock) {
    keys[k] = kEmpty;
    counts[k] = 0;
#pragma unroll
    for (int j = 0; j < NS; j++) sums[k][j] = 0ull;
  }
  __syncthreads();
  unsigned *n_keys = A.to_level_n ? &st->keys_n : &st->keys_f;
  unsigned *n_dirty = A.to_level_n ? &st->n_dirty_n : &st->n_dirty_f;
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
        int g = lm_find_insert(H, key, n_keys, &st->overflow);
        if (g >= 0) {
          unsigned long long *a = (unsigned long long *)(H.acc + (size_t)g * NS);
#pragma unroll
          for (int j = 0; j < NS; j++) atomicAdd(a + j, (v[j] ^ sgn) - sgn);
          atomicAdd(H.cnt + g, sign);
          lm_mark_dirty(H, g, n_dirty);
        }
      }
    }
    if (SWAP) __syncthreads();   // (the slot's old points are read by the lanes that overwrite them: no hazard; this keeps the passes apart in LDS only)
  }
  __syncthreads();
  // flush, step 1: every used LDS slot finds (or claims) its global slot.  The dirty list and the key counter are single
  // words that every tile of the launch appends to (~40 k cells per cloud): appended cell by cell, those same-address atomics
  // WERE the kernel (41-61 us for a 131 k-point key-frame); a tile now reserves its share with one atomic each.
  for (int k = threadIdx.x; k < kLdsCap; k += kBlock) {
    if (keys[k] == kEmpty) continue;
    int g = lm_find_insert(H, keys[k], &n_new_keys, &st->overflow);
    if (g < 0) continue;
    int u = atomicAdd(&n_used, 1);
    used[u] = k;
    gslot[u] = g;
    if (atomicExch(H.flag + g, 1u) == 0u) won[atomicAdd(&n_won, 1u)] = (unsigned)g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dirty_base = n_won ? atomicAdd(n_dirty, n_won) : 0u;
    if (n_new_keys) atomicAdd(n_keys, n_new_keys);
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < n_won; e += kBlock) H.dirty[dirty_base + e] = won[e];
  // step 2: one lane per (cell, term): the NS + 1 atomics of a cell leave from adjacent lanes into one or two lines
  const int total = n_used * (NS + 1);
  for (int e = threadIdx.x; e < total; e += kBlock) {
    int u = e / (NS + 1), j = e - u * (NS + 1);
    int slot = used[u], g = gslot[u];
    if (j < NS) atomicAdd((unsigned long long *)(H.acc + (size_t)g * NS) + j, sums[slot][j]);
    else atomicAdd(H.cnt + g, counts[slot]);
  }
}

struct FilterDirtyArgs {
  float leaf_f[3];
  float leaf_n[3], inv_n[3];
  FixScales fs_f, fs_n;
};

// one lane per touched level-F cell.  Consecutive entries of the dirty list are spatial neighbours (cells enter it in the order
// the scan first touches them), so the 256 cells of a chunk fall into a few dozen level-N cells: their changes are summed per N
// cell in an LDS table first and leave the workgroup as ONE set of atomics per N cell (cell by cell, ~80 k x 10 atomics landed on
// ~6 k x 10 words: 30 us).
constexpr int kFdCap = 128;   // LDS slots (N cells) per chunk of 256 F cells; what does not fit goes straight to memory
__global__ __launch_bounds__(kBlock) void k_lm_filter_dirty(HashView F, float4 *__restrict__ out, int32_t *__restrict__ has_out, HashView N, FilterDirtyArgs A,
                                                            LmStatus *__restrict__ st) {
  __shared__ unsigned long long keys[kFdCap];
  __shared__ int counts[kFdCap];
  __shared__ unsigned long long sums[kFdCap][10];
  __shared__ int gslot[kFdCap];
  __shared__ int used[kFdCap];
  __shared__ unsigned won[kFdCap];
  __shared__ int n_used;
  __shared__ unsigned n_won, n_new_keys, dirty_base;
  const unsigned nd = st->n_dirty_f;
  int d_live = 0;
  for (unsigned chunk = blockIdx.x; chunk * kBlock < nd; chunk += gridDim.x) {
    if (threadIdx.x == 0) { n_used = 0; n_won = 0u; n_new_keys = 0u; }
    for (int k = threadIdx.x; k < kFdCap; k += kBlock) {
      keys[k] = kEmpty;
      counts[k] = 0;
#pragma unroll
      for (int j = 0; j < 10; j++) sums[k][j] = 0ull;
    }
    __syncthreads();
    const unsigned t = chunk * kBlock + threadIdx.x;
    if (t < nd) {
      const unsigned s = F.dirty[t];
      F.flag[s] = 0u;
      const int cnt = F.cnt[s];
      const bool had = has_out[s] != 0, has = cnt > 0;
      const float4 old = out[s];
      float4 nw = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has) {
        int ijk[3];
        lm_unpack(F.keys[s], ijk);
        double o[3];
        lsa_abs_origin(ijk, A.leaf_f, o);
        const long long a[4] = {F.acc[(size_t)s * 4], F.acc[(size_t)s * 4 + 1], F.acc[(size_t)s * 4 + 2], F.acc[(size_t)s * 4 + 3]};
        nw = vi_filter_mean(a, cnt, o, A.fs_f);
      }
      d_live += (int)has - (int)had;
      const bool same = had && has && __float_as_uint(old.x) == __float_as_uint(nw.x) && __float_as_uint(old.y) == __float_as_uint(nw.y) &&
                        __float_as_uint(old.z) == __float_as_uint(nw.z) && __float_as_uint(old.w) == __float_as_uint(nw.w);
      if (!same) {  // the old output point leaves level N, the new one enters it
        out[s] = nw;
        has_out[s] = has ? 1 : 0;
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
            int g = lm_find_insert(N, key, &st->keys_n, &st->overflow);
            if (g >= 0) {
              unsigned long long *a = (unsigned long long *)(N.acc + (size_t)g * 10);
#pragma unroll
              for (int j = 0; j < 10; j++) atomicAdd(a + j, (v[j] ^ sgn) - sgn);
              atomicAdd(N.cnt + g, sign);
              lm_mark_dirty(N, g, &st->n_dirty_n);
            }
          }
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kFdCap; k += kBlock) {
      if (keys[k] == kEmpty) continue;
      int g = lm_find_insert(N, keys[k], &n_new_keys, &st->overflow);
      if (g < 0) continue;
      int u = atomicAdd(&n_used, 1);
      used[u] = k;
      gslot[u] = g;
      if (atomicExch(N.flag + g, 1u) == 0u) won[atomicAdd(&n_won, 1u)] = (unsigned)g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      dirty_base = n_won ? atomicAdd(&st->n_dirty_n, n_won) : 0u;
      if (n_new_keys) atomicAdd(&st->keys_n, n_new_keys);
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < n_won; e += kBlock) N.dirty[dirty_base + e] = won[e];
    const int total = n_used * 11;
    for (int e = threadIdx.x; e < total; e += kBlock) {
      int u = e / 11, j = e - u * 11;
      int slot = used[u], g = gslot[u];
      if (j < 10) atomicAdd((unsigned long long *)(N.acc + (size_t)g * 10) + j, sums[slot][j]);
      else atomicAdd(N.cnt + g, counts[slot]);
    }
    __syncthreads();
  }
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
__global__ __launch_bounds__(kBlock) void k_lm_ndt_dirty(HashView N, double *__restrict__ rec, int32_t *__restrict__ state, int32_t *__restrict__ alive,
                                                         float4 *__restrict__ dense, NdtDirtyArgs A, LmStatus *__restrict__ st, LmStatus *__restrict__ host_mirror,
                                                         unsigned long long epoch) {
  const unsigned nd = __hip_atomic_load(&st->n_dirty_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int d_live = 0;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < nd; t += gridDim.x * blockDim.x) {
    const unsigned s = N.dirty[t];
    N.flag[s] = 0u;
    const int cnt = N.cnt[s];
    int ijk[3];
    lm_unpack(N.keys[s], ijk);
    double *R = rec + (size_t)s * 10;
    int stt = 0;
    if (cnt > 0) {
      double o[3];
      lsa_abs_origin(ijk, A.leaf, o);
      long long a[10];
#pragma unroll
      for (int j = 0; j < 10; j++) a[j] = N.acc[(size_t)s * 10 + j];
      stt = ndt_cell_record(a, cnt, o, A.fs, A.pcl, R);
    } else {
#pragma unroll
      for (int j = 0; j < 10; j++) R[j] = 0.0;
    }
    state[s] = stt;
    const int was = alive[s], is = cnt > 0 ? 1 : 0;
    alive[s] = is;
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
  // The LAST workgroup to finish publishes the status block to the host and resets the dirty counters (this was a launch of
  // its own: ~4 us of kernel + a dispatch gap per key-frame).  Every word of the block is only ever changed by device-scope
  // atomics, which are performed at the memory side, so no cache write-back is involved: each wave drains its own
  // outstanding operations (s_waitcnt), the workgroup meets, one lane counts it done; the workgroup that finds itself last
  // reads the words with device-scope loads.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ unsigned s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(&st->done, 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x < 64) {
    // one lane per 32-bit word of the block in front of the epoch, all loads in ONE instruction (read one after the other
    // by a single lane, six dependent trips to the memory side cost the kernel 11 us)
    constexpr int kW = (int)(offsetof(LmStatus, epoch) / 4);
    constexpr int kNdF = (int)(offsetof(LmStatus, n_dirty_f) / 4), kNdN = (int)(offsetof(LmStatus, n_dirty_n) / 4), kDone = (int)(offsetof(LmStatus, done) / 4);
    const int lane = (int)threadIdx.x;
    unsigned *sw = reinterpret_cast<unsigned *>(st), *hw = reinterpret_cast<unsigned *>(host_mirror);
    const bool reset = lane == kNdF || lane == kNdN || lane == kDone;
    unsigned v = 0u;
    if (lane < kW && !reset) v = __hip_atomic_load(sw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < kW) __hip_atomic_store(hw + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (reset) __hip_atomic_store(sw + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the epoch word last, behind the completion of the others (a drain of the wave's stores: a RELEASE at system scope
    // would write back the whole L2 — the records and table entries this kernel has just written — for nothing)
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) __hip_atomic_store(&host_mirror->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// bounds of the valid voxels (window re-fit)
__global__ __launch_bounds__(kBlock) void k_lm_bounds(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ state, unsigned cap,
                                                      LmStatus *__restrict__ st) {
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += gridDim.x * blockDim.x) {
    if (keys[s] == kEmpty || state[s] != 1) continue;
    int ijk[3];
    lm_unpack(keys[s], ijk);
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], ijk[a]); mx[a] = max(mx[a], ijk[a]); }
  }
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[a] = min(mn[a], __shfl_down(mn[a], off, 64)); mx[a] = max(mx[a], __shfl_down(mx[a], off, 64)); }
    if ((threadIdx.x & 63) == 0) {
      if (mn[a] != INT_MAX) atomicMin(&st->bmin[a], mn[a]);
      if (mx[a] != INT_MIN) atomicMax(&st->bmax[a], mx[a]);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_lm_dense_fill(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ state, const double *__restrict__ rec,
                                                          unsigned cap, NdtDirtyArgs A, float4 *__restrict__ dense) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap || keys[s] == kEmpty || state[s] != 1) return;
  int ijk[3];
  lm_unpack(keys[s], ijk);
  const int i = ijk[0] - A.wmin[0], j = ijk[1] - A.wmin[1], k = ijk[2] - A.wmin[2];
  if ((unsigned)i >= (unsigned)A.wdim[0] || (unsigned)j >= (unsigned)A.wdim[1] || (unsigned)k >= (unsigned)A.wdim[2]) return;
  dense[(size_t)i + (size_t)j * A.wdim[0] + (size_t)k * A.wdim[0] * A.wdim[1]] = ndt_dense_entry(rec + (size_t)s * 10, ijk, A.leaf, (int)s);
}

// verification (tests): every occupied cell of a FULL table build must be in level N with the same state and the same ten
// doubles, bit for bit; out[0] cells compared, [1] mismatches
__global__ __launch_bounds__(kBlock) void k_lm_verify(HashView N, const double *__restrict__ rec, const int32_t *__restrict__ state, const int32_t *__restrict__ full_cor,
                                                      const double *__restrict__ full_rec, const int32_t *__restrict__ full_state, size_t V, CellMap m,
                                                      unsigned long long *__restrict__ out) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= V) return;
  int ijk[3];
  lsa_cell_unravel(m, full_cor[r], ijk);
  const unsigned long long key = lm_pack(ijk[0], ijk[1], ijk[2]);
  unsigned h = lm_probe0(key, N.cap);
  int slot = -1;
  for (int probe = 0; probe < kGlobalProbes; probe++) {
    unsigned long long k = N.keys[h];
    if (k == key) { slot = (int)h; break; }
    if (k == kEmpty) break;
    h = (h + kProbeStep) & (N.cap - 1u);
  }
  bool same = slot >= 0 && N.cnt[slot] > 0 && state[slot] == full_state[r];
  if (same)
    for (int j = 0; j < 10; j++)
      same = same && __double_as_longlong(rec[(size_t)slot * 10 + j]) == __double_as_longlong(full_rec[r * 10 + j]);
  atomicAdd(out, 1ull);
  if (!same) atomicAdd(out + 1, 1ull);
}
__global__ __launch_bounds__(kBlock) void k_lm_count(const int32_t *__restrict__ alive, const int32_t *__restrict__ state, unsigned cap, unsigned long long *__restrict__ out) {
  unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  if (alive[s]) atomicAdd(out + 2, 1ull);
  if (alive[s] && state[s] == 1) atomicAdd(out + 3, 1ull);
}

HashView view(LmHash &h) { return HashView{h.keys.p, h.cnt.p, h.acc.p, h.flag.p, h.dirty.p, h.cap}; }

int hash_alloc(lsa_context *ctx, LmHash &h, unsigned cap, int ns) {
  int rc;
  if (cap == h.cap && ns == h.ns) return LSA_OK;
  h.release();
  if ((rc = h.keys.reserve(ctx, cap)) || (rc = h.cnt.reserve(ctx, cap)) || (rc = h.acc.reserve(ctx, (size_t)cap * ns)) || (rc = h.flag.reserve(ctx, cap)) ||
      (rc = h.dirty.reserve(ctx, cap)))
    return rc;
  h.cap = cap; h.ns = ns;
  return LSA_OK;
}

int hash_clear(lsa_context *ctx, LmHash &h) {
  LSA_HIP(ctx, hipMemsetAsync(h.keys.p, 0xFF, (size_t)h.cap * sizeof(unsigned long long), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(h.cnt.p, 0, (size_t)h.cap * sizeof(int32_t), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(h.acc.p, 0, (size_t)h.cap * h.ns * sizeof(long long), ctx->stream));
  LSA_HIP(ctx, hipMemsetAsync(h.flag.p, 0, (size_t)h.cap * sizeof(unsigned), ctx->stream));
  return LSA_OK;
}

NdtDirtyArgs ndt_args(const LocalMap &lm, bool with_window) {
  NdtDirtyArgs A;
  for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.wmin[a] = lm.wmin[a]; A.wdim[a] = lm.wdim[a]; }
  A.fs = lm.fs_n;
  A.pcl = lm.pcl;
  A.have_window = (with_window && lm.window_set) ? 1 : 0;
  return A;
}

unsigned dirty_grid(const lsa_context *ctx) { return (unsigned)ctx->num_cus * 2u; }   // grid-stride over a count only the device knows

int alloc_level_n(LocalMap &lm, unsigned cap) {
  lsa_context *ctx = lm.ctx;
  int rc;
  if ((rc = hash_alloc(ctx, lm.n, cap, 10))) return rc;
  if (lm.rec.cap != (size_t)cap * 10) { lm.rec.release(); lm.state.release(); lm.alive.release(); }
  if ((rc = lm.rec.reserve(ctx, (size_t)cap * 10)) || (rc = lm.state.reserve(ctx, cap)) || (rc = lm.alive.reserve(ctx, cap))) return rc;
  return LSA_OK;
}

int alloc_level_f(LocalMap &lm, unsigned cap) {
  lsa_context *ctx = lm.ctx;
  int rc;
  if ((rc = hash_alloc(ctx, lm.f, cap, 4))) return rc;
  if (lm.out.cap != cap) { lm.out.release(); lm.has_out.release(); }
  if ((rc = lm.out.reserve(ctx, cap)) || (rc = lm.has_out.reserve(ctx, cap))) return rc;
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
  if (!lm.n.cap && (rc = alloc_level_n(lm, 1u << 17))) return rc;
  return LSA_OK;
}

int lm_points(LocalMap &lm, const float4 *d_pts, size_t n, int sign) {
  lsa_context *ctx = lm.ctx;
  if (n == 0) return LSA_OK;
  PointsArgs A;
  A.sign = sign;
  A.evict = 0; A.raw = nullptr; A.n_raw = 0; A.keep = nullptr; A.force_w_one = 0;
  for (int k = 0; k < 12; k++) A.T[k] = 0.f;
  const unsigned g = lsa_div_up(n, kTile);
  if (lm.filtered) {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_f[a]; A.inv[a] = 1.0f / lm.leaf_f[a]; }
    A.fs = lm.fs_f;
    A.to_level_n = 0;
    hipLaunchKernelGGL((k_lm_points<4, 0, false>), dim3(g), dim3(kBlock), 0, ctx->stream, const_cast<float4 *>(d_pts), n, A, view(lm.f), lm.d_status.p);
  } else {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.inv[a] = 1.0f / lm.leaf_n; }
    A.fs = lm.fs_n;
    A.to_level_n = 1;
    hipLaunchKernelGGL((k_lm_points<10, 1, false>), dim3(g), dim3(kBlock), 0, ctx->stream, const_cast<float4 *>(d_pts), n, A, view(lm.n), lm.d_status.p);
  }
  LSA_HIP(ctx, hipGetLastError());
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
  if (lm.filtered) {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_f[a]; A.inv[a] = 1.0f / lm.leaf_f[a]; }
    A.fs = lm.fs_f;
    A.to_level_n = 0;
    hipLaunchKernelGGL((k_lm_points<4, 0, true>), dim3(g), dim3(kBlock), 0, ctx->stream, d_slot, slot_cap, A, view(lm.f), lm.d_status.p);
  } else {
    for (int a = 0; a < 3; a++) { A.leaf[a] = lm.leaf_n; A.inv[a] = 1.0f / lm.leaf_n; }
    A.fs = lm.fs_n;
    A.to_level_n = 1;
    hipLaunchKernelGGL((k_lm_points<10, 1, true>), dim3(g), dim3(kBlock), 0, ctx->stream, d_slot, slot_cap, A, view(lm.n), lm.d_status.p);
  }
  LSA_HIP(ctx, hipGetLastError());
  return LSA_OK;
}

int lm_finish(LocalMap &lm) {
  lsa_context *ctx = lm.ctx;
  if (lm.filtered) {
    FilterDirtyArgs A;
    for (int a = 0; a < 3; a++) { A.leaf_f[a] = lm.leaf_f[a]; A.leaf_n[a] = lm.leaf_n; A.inv_n[a] = 1.0f / lm.leaf_n; }
    A.fs_f = lm.fs_f; A.fs_n = lm.fs_n;
    hipLaunchKernelGGL(k_lm_filter_dirty, dim3(dirty_grid(ctx)), dim3(kBlock), 0, ctx->stream, view(lm.f), lm.out.p, lm.has_out.p, view(lm.n), A, lm.d_status.p);
  }
  lm.epoch++;
  // (a small grid: every workgroup ends with one same-address atomic — ~20 ns each at the memory side, 512 of them were 11 us —
  // and a key-frame touches ~6 k level-N cells, one per lane of 24 workgroups)
  hipLaunchKernelGGL(k_lm_ndt_dirty, dim3(64), dim3(kBlock), 0, ctx->stream, view(lm.n), lm.rec.p, lm.state.p, lm.alive.p, lm.dense.p, ndt_args(lm, true),
                     lm.d_status.p, lm.h_status.p, lm.epoch);
  LSA_HIP(ctx, hipGetLastError());
  lm.pending = true;
  lm.updates++;
  return LSA_OK;
}

int lm_refit_window(LocalMap &lm) {
  lsa_context *ctx = lm.ctx;
  // bounds of the valid voxels
  LmStatus init;
  memset(&init, 0, sizeof(init));
  for (int a = 0; a < 3; a++) { init.bmin[a] = INT_MAX; init.bmax[a] = INT_MIN; }
  LSA_HIP(ctx, hipMemcpyAsync((char *)lm.d_status.p + offsetof(LmStatus, bmin), init.bmin, 6 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_lm_bounds, dim3(lsa_div_up(lm.n.cap, kBlock * 4)), dim3(kBlock), 0, ctx->stream, lm.n.keys.p, lm.state.p, lm.n.cap, lm.d_status.p);
  LmStatus got;
  LSA_HIP(ctx, hipMemcpyAsync(&got, lm.d_status.p, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
  LSA_HIP(ctx, hipMemsetAsync(lm.dense.p, 0xFF, cells * sizeof(float4), ctx->stream));
  hipLaunchKernelGGL(k_lm_dense_fill, dim3(lsa_div_up(lm.n.cap, kBlock)), dim3(kBlock), 0, ctx->stream, lm.n.keys.p, lm.state.p, lm.rec.p, lm.n.cap, ndt_args(lm, true), lm.dense.p);
  // the flag is consumed
  const unsigned zero = 0;
  LSA_HIP(ctx, hipMemcpyAsync((char *)lm.d_status.p + offsetof(LmStatus, window_miss), &zero, sizeof(zero), hipMemcpyHostToDevice, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (`zero` and `init` are stack variables)
  lm.h_status.p->window_miss = 0;
  lm.refits++;
  return LSA_OK;
}

int lm_rebuild(LocalMap &lm, const float4 *const *d_ranges, const size_t *n_ranges, int nranges, bool filtered) {
  lsa_context *ctx = lm.ctx;
  int rc;
  lm.valid = false;
  lm.filtered = filtered;
  unsigned cap_f = lm.f.cap ? lm.f.cap : (1u << 20), cap_n = lm.n.cap ? lm.n.cap : (1u << 17);
  for (int attempt = 0; attempt < 8; attempt++) {
    if ((rc = alloc_level_n(lm, cap_n))) return rc;
    if (filtered && (rc = alloc_level_f(lm, cap_f))) return rc;
    if ((rc = hash_clear(ctx, lm.n))) return rc;
    LSA_HIP(ctx, hipMemsetAsync(lm.state.p, 0, (size_t)lm.n.cap * sizeof(int32_t), ctx->stream));
    LSA_HIP(ctx, hipMemsetAsync(lm.alive.p, 0, (size_t)lm.n.cap * sizeof(int32_t), ctx->stream));
    if (filtered) {
      if ((rc = hash_clear(ctx, lm.f))) return rc;
      LSA_HIP(ctx, hipMemsetAsync(lm.has_out.p, 0, (size_t)lm.f.cap * sizeof(int32_t), ctx->stream));
    }
    LSA_HIP(ctx, hipMemsetAsync(lm.d_status.p, 0, sizeof(LmStatus), ctx->stream));
    for (int r = 0; r < nranges; r++)
      if ((rc = lm_points(lm, d_ranges[r], n_ranges[r], +1))) return rc;
    // every cell is in the dirty lists: finalise them all (no window yet: the table is filled by the re-fit below)
    lm.window_set = false;
    if ((rc = lm_finish(lm))) return rc;
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const LmStatus &s = *lm.h_status.p;
    if (s.overflow & 2u) return lsa_fail(ctx, LSA_ERR_RANGE, "incremental local map: a point lies beyond +-2^20 cells");
    const bool grow_f = filtered && (s.overflow || s.keys_f > lm.f.cap / 4), grow_n = s.overflow || s.keys_n > lm.n.cap / 4;
    if (!grow_f && !grow_n) break;
    if (attempt == 7 || (grow_f && cap_f >= (1u << 26)) || (grow_n && cap_n >= (1u << 24)))
      return lsa_fail(ctx, LSA_ERR_CAPACITY, "incremental local map: %u / %u cells do not fit the tables", s.keys_f, s.keys_n);
    if (grow_f) cap_f *= 2;
    if (grow_n) cap_n *= 2;
  }
  if ((rc = lm_refit_window(lm))) return rc;
  lm.pending = false;
  lm.valid = true;
  lm.rebuilds++;
  return LSA_OK;
}

int lm_check(LocalMap &lm) {
  if (!lm.pending) return 0;
  const LmStatus &s = *lm.h_status.p;
  if (__atomic_load_n(&lm.h_status.p->epoch, __ATOMIC_ACQUIRE) != lm.epoch) {
    // The update is still running (a batch of Updates filters on another stream, so nothing has ordered the host behind it):
    // its last kernel writes the epoch, ~60 us after the key-frame.  Past any plausible run time: ask the runtime.
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(&lm.h_status.p->epoch, __ATOMIC_ACQUIRE) != lm.epoch) {
      __builtin_ia32_pause();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) { (void)hipStreamSynchronize(lm.ctx->stream); break; }
    }
  }
  lm.pending = false;
  if (s.overflow) return 2;
  if ((lm.filtered && s.keys_f > (lm.f.cap / 10) * 6) || s.keys_n > (lm.n.cap / 10) * 6) return 2;
  if (s.window_miss) return 1;
  return 0;
}

int lm_verify(LocalMap &lm, const int32_t *full_cell_of_rank, const double *full_rec, const int32_t *full_state, size_t V, const CellMap &map,
              unsigned long long out[4]) {
  lsa_context *ctx = lm.ctx;
  unsigned long long *d_out = nullptr;
  LSA_HIP(ctx, hipMalloc((void **)&d_out, 4 * sizeof(unsigned long long)));
  (void)hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), ctx->stream);
  if (V) hipLaunchKernelGGL(k_lm_verify, dim3(lsa_div_up(V, kBlock)), dim3(kBlock), 0, ctx->stream, view(lm.n), lm.rec.p, lm.state.p, full_cell_of_rank, full_rec, full_state, V, map, d_out);
  hipLaunchKernelGGL(k_lm_count, dim3(lsa_div_up(lm.n.cap, kBlock)), dim3(kBlock), 0, ctx->stream, lm.alive.p, lm.state.p, lm.n.cap, d_out);
  hipError_t e = hipMemcpyAsync(out, d_out, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_out);
  LSA_HIP(ctx, e);
  return LSA_OK;
}
