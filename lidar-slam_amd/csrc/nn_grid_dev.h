// nn_grid_dev.h — internal device side of nn_grid.h: the ring search.
#ifndef LSA_NN_GRID_DEV_H_
#define LSA_NN_GRID_DEV_H_

#include "nn_grid.h"
#include "voxel_index_dev.h"

struct NNHit {
  float d2;       // f32 squared distance, +inf when nothing was found
  unsigned id;    // original index of the neighbour, 0xffffffff when nothing was found
  float x, y, z;  // the neighbour
  unsigned slot;  // its position in the cell-sorted array (index of per-slot side tables)
};

__device__ __forceinline__ void nn_visit_cell(const NNView &g, int ix, int iy, int iz, float qx, float qy, float qz, NNHit &h) {
  const CellMap &m = g.map;
  int cell = ix + iy * m.dim[0] + iz * m.dim[0] * m.dim[1];
  uint32_t w = g.bits[cell >> 5], bit = 1u << (cell & 31);
  if (!(w & bit)) return;
  unsigned rank = g.prefix[cell >> 5] + __popc(w & (bit - 1u));
  unsigned a = g.start[rank], b = g.start[rank + 1];
  for (unsigned k = a; k < b; k++) {
    float4 t = g.sorted[k];
    float ex = lsa_fsub(qx, t.x), ey = lsa_fsub(qy, t.y), ez = lsa_fsub(qz, t.z);
    float d = lsa_fadd(lsa_fadd(lsa_fmul(ex, ex), lsa_fmul(ey, ey)), lsa_fmul(ez, ez));  // FLANN L2_Simple, f32
    unsigned id = __float_as_uint(t.w);
    if (d < h.d2 || (d == h.d2 && id < h.id)) { h.d2 = d; h.id = id; h.x = t.x; h.y = t.y; h.z = t.z; h.slot = k; }
  }
}

// Exact nearest neighbour of a FINITE query.  max_ring bounds the search: with a gate g a caller that only accepts
// neighbours within the gate passes max_ring = ceil(g / cell) + 1 (anything farther out is farther than the gate);
// INT_MAX searches until the answer is proven.  Every wave-divergent loop below has a trip count bounded by the grid
// dimensions, so all lanes finish.
//
// SUB lanes (1, 2, 4 or 8 ADJACENT lanes, same query in each) share one search: the cells of a ring are dealt round-robin
// to the SUB lanes and the partial results are merged after every ring, so the dependent bitmap -> prefix -> start ->
// points chain a single lane would walk for 27-125 cells is cut SUB-fold.  All SUB lanes return the same hit.
template <int SUB>
__device__ __forceinline__ NNHit nn_nearest_sub(const NNView &g, float qx, float qy, float qz, int max_ring, int sub) {
  const CellMap &m = g.map;
  NNHit h;
  h.d2 = INFINITY; h.id = 0xffffffffu; h.x = h.y = h.z = 0.0f; h.slot = 0u;
  // unclamped cell coordinates relative to the grid corner (mode-1 rule: floor(x / leaf) - min_b), saturated so that
  // the ring arithmetic cannot overflow
  float fx = floorf(lsa_fdiv(qx, m.leaf[0])), fy = floorf(lsa_fdiv(qy, m.leaf[1])), fz = floorf(lsa_fdiv(qz, m.leaf[2]));
  const float lim = 5.0e8f;
  int c0 = (int)fminf(fmaxf(fx, -lim), lim) - m.min_b[0];
  int c1 = (int)fminf(fmaxf(fy, -lim), lim) - m.min_b[1];
  int c2 = (int)fminf(fmaxf(fz, -lim), lim) - m.min_b[2];
  int out0 = c0 < 0 ? -c0 : (c0 > m.dim[0] - 1 ? c0 - (m.dim[0] - 1) : 0);
  int out1 = c1 < 0 ? -c1 : (c1 > m.dim[1] - 1 ? c1 - (m.dim[1] - 1) : 0);
  int out2 = c2 < 0 ? -c2 : (c2 > m.dim[2] - 1 ? c2 - (m.dim[2] - 1) : 0);
  int rho0 = max(out0, max(out1, out2));  // first ring that touches the grid
  int far0 = max(abs(c0), abs(m.dim[0] - 1 - c0)), far1 = max(abs(c1), abs(m.dim[1] - 1 - c1)), far2 = max(abs(c2), abs(m.dim[2] - 1 - c2));
  int rho_end = max(far0, max(far1, far2));  // last ring that touches the grid
  if (max_ring < rho_end) rho_end = max_ring;
  for (int rho = rho0; rho <= rho_end; rho++) {
    int z0 = c2 - rho, z1 = c2 + rho, y0 = c1 - rho, y1 = c1 + rho, x0 = c0 - rho, x1 = c0 + rho;
    int zc0 = max(z0, 0), zc1 = min(z1, m.dim[2] - 1), yc0 = max(y0, 0), yc1 = min(y1, m.dim[1] - 1), xc0 = max(x0, 0), xc1 = min(x1, m.dim[0] - 1);
    int turn = 0;  // round-robin over the SUB lanes, identical sequence in each of them
    for (int iz = zc0; iz <= zc1; iz++)
      for (int iy = yc0; iy <= yc1; iy++) {
        if (iz == z0 || iz == z1 || iy == y0 || iy == y1) {
          for (int ix = xc0; ix <= xc1; ix++)
            if (SUB == 1 || ((turn++) & (SUB - 1)) == sub) nn_visit_cell(g, ix, iy, iz, qx, qy, qz, h);
        } else {
          if (x0 >= 0 && x0 < m.dim[0])
            if (SUB == 1 || ((turn++) & (SUB - 1)) == sub) nn_visit_cell(g, x0, iy, iz, qx, qy, qz, h);
          if (x1 != x0 && x1 >= 0 && x1 < m.dim[0])
            if (SUB == 1 || ((turn++) & (SUB - 1)) == sub) nn_visit_cell(g, x1, iy, iz, qx, qy, qz, h);
        }
      }
    if (SUB > 1) {  // merge across the SUB lanes (xor butterfly: every lane ends with the group's best)
#pragma unroll
      for (int off = 1; off < SUB; off <<= 1) {
        NNHit o;
        o.d2 = __shfl_xor(h.d2, off, 64); o.id = __shfl_xor(h.id, off, 64);
        o.x = __shfl_xor(h.x, off, 64); o.y = __shfl_xor(h.y, off, 64); o.z = __shfl_xor(h.z, off, 64); o.slot = __shfl_xor(h.slot, off, 64);
        if (o.d2 < h.d2 || (o.d2 == h.d2 && o.id < h.id)) h = o;
      }
    }
    // every cell not visited yet lies at least rho * cell away from the query (minus slack for the f32 cell rounding)
    float lb = (float)rho * g.cell - 1e-3f;
    if (h.id != 0xffffffffu && lb > 0.0f && h.d2 <= lb * lb) break;
  }
  return h;
}

__device__ __forceinline__ NNHit nn_nearest(const NNView &g, float qx, float qy, float qz, int max_ring) {
  return nn_nearest_sub<1>(g, qx, qy, qz, max_ring, 0);
}

#endif
