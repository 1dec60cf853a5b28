// nn_grid.h — internal: exact nearest-neighbour structure over a point cloud (what pcl::KdTreeFLANN / pcl::search::KdTree
// give the reference: icp_registration_manual.cpp:52,146; PCL's Registration::getFitnessScore and
// CorrespondenceEstimation behind the "NDT" / "ICP" plugins).
//
// A kd-tree is pointer chasing; on the GPU the same exact answer comes from a uniform grid: the points are
// counting-sorted by the occupied-cell rank of the sparse voxel index (voxel_index.h), so the points of one cell are
// one contiguous run of float4 {x, y, z, original index}, and a query walks Chebyshev rings of cells around its own cell
// until the best distance is provably minimal.  Distances are the f32 expression FLANN's L2_Simple evaluates;
// ties go to the lowest original index.
#ifndef LSA_NN_GRID_H_
#define LSA_NN_GRID_H_

#include "voxel_index.h"

struct NNGrid {
  VoxelIndex vi;
  DevBuf<uint32_t> cnt, start, fill;
  DevBuf<float4> sorted;
  DevBuf<uint2> dense;           // per cell {first slot, #points}; built when the cell box has <= kNNDenseMaxCells cells
  bool use_dense = false;
  DevBuf<double> fit_partials;   // nn_fitness: per-workgroup {sum d2, count}
  PinnedBuf<double> fit_host;
  size_t n = 0;         // points handed to the build
  size_t n_sorted = 0;  // finite points = entries of sorted[]
  float cell = 0;
  bool built = false;
  void release() { vi.release(); cnt.release(); start.release(); fill.release(); sorted.release(); dense.release(); use_dense = false; fit_partials.release(); fit_host.release(); n = 0; built = false; }
};

constexpr long long kNNDenseMaxCells = 16ll << 20;  // 128 MiB of table at most

// kernel-argument view
struct NNView {
  CellMap map;
  const uint32_t *bits, *prefix, *start;
  const float4 *sorted;
  float cell;
  unsigned n_sorted;  // points in sorted[]
  const uint2 *dense;  // per cell {first slot, #points}, or null when the cell box is too large for a dense table
};

// Build over n device points with the given cell edge (synchronises the context's stream once).
int nn_grid_build(lsa_context *ctx, NNGrid &g, const float4 *d_pts, size_t n, float cell, const char *who);
// pcl::Registration::getFitnessScore(max_range) (PCL 1.7 registration/impl/registration.hpp): mean over the source points,
// transformed by the f32 column-major T, of the SQUARED f32 distance to the nearest point of the grid; pairs with
// d2 > max_range are dropped; DBL_MAX (std::numeric_limits<double>::max()) when no pair is left.
// allreduce (optional): the source is one rank's shard; {sum, count} are summed over the ranks before the division
int nn_fitness(lsa_context *ctx, NNGrid &g, const float4 *d_src, size_t n, const float T[16], float max_range, double *out, lsa_allreduce_fn allreduce = nullptr,
               void *allreduce_user = nullptr);
inline NNView nn_view(const NNGrid &g) { return NNView{g.vi.map, g.vi.bits.p, g.vi.prefix.p, g.start.p, g.sorted.p, g.cell, (unsigned)g.n_sorted, g.use_dense ? g.dense.p : nullptr}; }

#endif
