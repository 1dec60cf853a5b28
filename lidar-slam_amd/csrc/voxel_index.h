// voxel_index.h — internal: sparse voxel index shared by the VoxelFilter and the NDT target grid.
//
// Data layout in HBM (all device-resident, grow-only):
//   bits[W]        u32   occupancy bitmap over the dense cell box, cell = i + j*dx + k*dx*dy (x fastest),
//                        W = ceil(ncells / 32).  At KITTI sizes this is 20 KB (NDT, 1 m cells) to
//                        ~1 MB (0.3 m filter over a 250 m map): L2-resident, unlike a dense record grid.
//   prefix[W]      u32   exclusive prefix of popcount(bits): rank(cell) = prefix[w] + popc(bits[w] & below(b)).
//                        Ranks enumerate occupied cells in ASCENDING linear index, which is exactly the
//                        output order PCL's sort produces, so no sort is needed anywhere.
//   cell_of_rank[V] i32  inverse map (cell linear index per occupied rank).
//   acc[V][NS]     i64   per-cell fixed-point sums (order-independent => bitwise reproducible),
//   cnt[V]         i32   per-cell point counts.
#ifndef LSA_VOXEL_INDEX_H_
#define LSA_VOXEL_INDEX_H_

#include "lsa_common.h"

struct CellMap {
  int mode;  // 0: PCL VoxelGrid (floor(x * inv_leaf) - (float)min_b); 1: NDT VoxelGrid (floor(x / leaf) - min_b)
  float leaf[3];
  float inv[3];
  int min_b[3];
  int dim[3];
  long long ncells;
};

struct FixScales {
  double s1, inv_s1;  // first moments  (metres)
  double s2, inv_s2;  // second moments (metres^2)
  double sw, inv_sw;  // intensity / static weight
};

struct VoxelIndex {
  DevBuf<uint32_t> bits, prefix, blocksums;
  DevBuf<int32_t> cell_of_rank, cnt, cell_idx;
  DevBuf<long long> acc;
  DevBuf<float> scratch;  // bbox partials
  PinnedBuf<float> h_small;
  size_t W = 0;
  size_t nocc = 0;
  CellMap map{};
  FixScales fs{};
  void release() {
    bits.release(); prefix.release(); blocksums.release(); cell_of_rank.release(); cnt.release(); cell_idx.release();
    acc.release(); scratch.release(); h_small.release();
  }
};

// min/max over finite points; out[0..2] = min, [3..5] = max, [6] = #finite (as float bits of an int). One sync.
int vi_bbox(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, float mn[3], float mx[3], size_t *n_finite);
// PCL rule set (filter) / reference NDT rule set; returns LSA_ERR_RANGE on int32 overflow of the cell count
int vi_make_map_pcl(const float mn[3], const float mx[3], const float leaf[3], CellMap *m);
int vi_make_map_ndt(const float mn[3], const float mx[3], float leaf, CellMap *m);
void vi_make_scales(const CellMap &m, size_t n, FixScales *fs);
// mark + scan + inverse map; sets vi.nocc. One sync.
int vi_build(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n);
// accumulate NS sums per point: NS = 4 (dx,dy,dz,w) or 10 (dx,dy,dz,dxx,dxy,dxz,dyy,dyz,dzz,w). No sync.
int vi_accumulate(lsa_context *ctx, VoxelIndex &vi, const float4 *d_pts, size_t n, int ns);

#endif
