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

// compare the query with the points sorted[a, b) of one cell.  Four points per step, their loads issued together: a cell
// holds a handful of points and a load-compare-load chain would cost one memory round trip per point.
// TRACK: `sec` follows the smallest squared distance among the visited points OTHER than the best one (see nn_nearest_sub).
template <bool TRACK = false>
__device__ __forceinline__ void nn_visit_range(const NNView &g, unsigned a, unsigned b, float qx, float qy, float qz, NNHit &h, float &sec) {
  for (unsigned k0 = a; k0 < b; k0 += 4) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) t[u] = g.sorted[min(k0 + u, b - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u >= b) break;
      float ex = lsa_fsub(qx, t[u].x), ey = lsa_fsub(qy, t[u].y), ez = lsa_fsub(qz, t[u].z);
      float d = lsa_fadd(lsa_fadd(lsa_fmul(ex, ex), lsa_fmul(ey, ey)), lsa_fmul(ez, ez));  // FLANN L2_Simple, f32
      unsigned id = __float_as_uint(t[u].w);
      if (d < h.d2 || (d == h.d2 && id < h.id)) {
        if (TRACK) sec = fminf(sec, h.d2);   // the best so far becomes a runner-up (+inf when there was none)
        h.d2 = d; h.id = id; h.x = t[u].x; h.y = t[u].y; h.z = t[u].z; h.slot = k0 + u;
      } else if (TRACK && id != h.id) {      // (the seed is met again in its cell: not a runner-up of itself)
        sec = fminf(sec, d);
      }
    }
  }
}
__device__ __forceinline__ void nn_visit_range(const NNView &g, unsigned a, unsigned b, float qx, float qy, float qz, NNHit &h) {
  float unused = INFINITY;
  nn_visit_range<false>(g, a, b, qx, qy, qz, h, unused);
}

// visit the points of one occupied cell given its bitmap word (already loaded)
template <bool TRACK = false>
__device__ __forceinline__ void nn_visit_bit(const NNView &g, int cell, uint32_t w, float qx, float qy, float qz, NNHit &h, float &sec) {
  uint32_t bit = 1u << (cell & 31);
  unsigned rank = g.prefix[cell >> 5] + __popc(w & (bit - 1u));
  nn_visit_range<TRACK>(g, g.start[rank], g.start[rank + 1], qx, qy, qz, h, sec);
}

// visit the occupied cells among x in [xa, xb] (clipped by the caller) of row (iy, iz); only_ends: just xa and xb.
// Cells of a row are consecutive bits, so an EMPTY stretch costs one or two bitmap words instead of a load per cell —
// what a query with no neighbour nearby spends its time on.
template <bool TRACK = false>
__device__ __forceinline__ void nn_visit_row(const NNView &g, int xa, int xb, int iy, int iz, bool only_ends, float qx, float qy, float qz, NNHit &h, float &sec) {
  const CellMap &m = g.map;
  const int base = iy * m.dim[0] + iz * m.dim[0] * m.dim[1];
  const int ca = base + xa, cb = base + xb;
  for (int wi = ca >> 5; wi <= (cb >> 5); wi++) {
    uint32_t w = g.bits[wi];
    if (!w) continue;
    int lo = max(ca, wi << 5) & 31, hi = min(cb, (wi << 5) + 31) & 31;
    uint32_t mask = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
    if (only_ends) {
      uint32_t ends = 0u;
      if ((ca >> 5) == wi) ends |= 1u << (ca & 31);
      if ((cb >> 5) == wi) ends |= 1u << (cb & 31);
      mask &= ends;
    }
    uint32_t bitsel = w & mask;
    while (bitsel) {
      int b = __ffs(bitsel) - 1;
      bitsel &= bitsel - 1;
      nn_visit_bit<TRACK>(g, (wi << 5) + b, w, qx, qy, qz, h, sec);
    }
  }
}

// Exact nearest neighbour of a FINITE query.  max_ring bounds the search: with a gate g a caller that only accepts
// neighbours within the gate passes max_ring = ceil(g / cell) + 1 (anything farther out is farther than the gate);
// INT_MAX searches until the answer is proven.  Every wave-divergent loop below has a trip count bounded by the grid
// dimensions, so all lanes finish.
//
// SUB lanes (1, 2, 4 or 8 ADJACENT lanes, same query in each) share one search: the (y, z) rows of a ring are dealt round-robin
// to the SUB lanes and the partial results are merged after every ring, so the dependent bitmap -> prefix -> start ->
// points chain a single lane would walk for 27-125 cells is cut SUB-fold.  All SUB lanes return the same hit.
// gate_d2: squared distance beyond which the caller rejects the neighbour anyway (INFINITY: none).  Cells — and whole
// rows — that cannot hold a point closer than min(gate_d2, best so far) are skipped without touching memory: the cube
// of rings shrinks to the sphere that matters (about half the cells), which is what bounds a query with no neighbour.
constexpr int kPhasedMaxRing = 2;
// seed_hit (optional, instead of seed_slot): a hit the caller has already evaluated — e.g. against another grid over the same
// points; its `slot` is passed through untouched unless a better point of THIS grid replaces it.
//
// TRACK (ICP's search-skipping, icp.hip): inside the rings the search walks anyway, cells are ruled out against
// min(gate, best) + `widen` metres instead of min(gate, best), and *others_lb receives a lower bound, in metres, of the distance
// from the query to every target point other than the returned one: the runner-up among the visited points, the widened reach,
// or the rings' own coverage, whichever is smallest.  The answer itself is the same point either way.
template <int SUB, bool TRACK = false>
__device__ __forceinline__ NNHit nn_nearest_sub(const NNView &g, float qx, float qy, float qz, int max_ring, int sub, float gate_d2 = INFINITY,
                                                unsigned seed_slot = 0xffffffffu, const NNHit *seed_hit = nullptr, float widen = 0.0f,
                                                float *others_lb = nullptr) {
  const CellMap &m = g.map;
  float sec = INFINITY;
  const float gate_r = TRACK ? sqrtf(gate_d2) : 0.0f;
  // squared radius beyond which a cell cannot matter
  auto reach2 = [&](float best_d2) -> float {
    if (!TRACK) return fminf(gate_d2, best_d2);
    const float r = fminf(gate_r, sqrtf(best_d2)) + widen;
    return r * r;
  };
  NNHit h;
  h.d2 = INFINITY; h.id = 0xffffffffu; h.x = h.y = h.z = 0.0f; h.slot = 0u;
  if (seed_hit) h = *seed_hit;
  // seed_slot: a point of the sorted array known to be a good guess (the neighbour found for this query one ICP iteration
  // ago).  It is treated as already visited, which changes nothing in the answer — every cell that could hold a closer
  // (or equally close, lower-index) point is still searched — but shrinks the sphere the search has to cover from the
  // start; from the second iteration on most queries end after their own cell and a few face neighbours.
  if (seed_slot < g.n_sorted) {
    const float4 t = g.sorted[seed_slot];
    const float ex = lsa_fsub(qx, t.x), ey = lsa_fsub(qy, t.y), ez = lsa_fsub(qz, t.z);
    h.d2 = lsa_fadd(lsa_fadd(lsa_fmul(ex, ex), lsa_fmul(ey, ey)), lsa_fmul(ez, ez));
    h.id = __float_as_uint(t.w); h.x = t.x; h.y = t.y; h.z = t.z; h.slot = seed_slot;
  }
  // unclamped cell coordinates relative to the grid corner (mode-1 rule: floor(x / leaf) - min_b), saturated so that
  // the ring arithmetic cannot overflow
  const float vx = lsa_fdiv(qx, m.leaf[0]), vy = lsa_fdiv(qy, m.leaf[1]), vz = lsa_fdiv(qz, m.leaf[2]);
  float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
  const float lim = 5.0e8f;
  int c0 = (int)fminf(fmaxf(fx, -lim), lim) - m.min_b[0];
  int c1 = (int)fminf(fmaxf(fy, -lim), lim) - m.min_b[1];
  int c2 = (int)fminf(fmaxf(fz, -lim), lim) - m.min_b[2];
  // position of the query inside its own cell, in cells: [0, 1).  A point that the grid puts into cell i has
  // floor(x / leaf) == i under the same f32 division, so along one axis it is at least (i - c - u) cells (i > c) or
  // (c - i - 1 + u) cells (i < c) away from the query, up to the rounding of that division; kSlack (in metres, far above an
  // ulp of any coordinate the grid accepts) is taken off every such bound.
  const float ux = fminf(fmaxf(vx - fx, 0.0f), 1.0f), uy = fminf(fmaxf(vy - fy, 0.0f), 1.0f), uz = fminf(fmaxf(vz - fz, 0.0f), 1.0f);
  const float kSlack = 1e-3f;
  const float umin = fminf(fminf(fminf(ux, 1.0f - ux), fminf(uy, 1.0f - uy)), fminf(uz, 1.0f - uz));
  int out0 = c0 < 0 ? -c0 : (c0 > m.dim[0] - 1 ? c0 - (m.dim[0] - 1) : 0);
  int out1 = c1 < 0 ? -c1 : (c1 > m.dim[1] - 1 ? c1 - (m.dim[1] - 1) : 0);
  int out2 = c2 < 0 ? -c2 : (c2 > m.dim[2] - 1 ? c2 - (m.dim[2] - 1) : 0);
  int rho0 = max(out0, max(out1, out2));  // first ring that touches the grid
  int far0 = max(abs(c0), abs(m.dim[0] - 1 - c0)), far1 = max(abs(c1), abs(m.dim[1] - 1 - c1)), far2 = max(abs(c2), abs(m.dim[2] - 1 - c2));
  int rho_end = max(far0, max(far1, far2));  // last ring that touches the grid
  if (max_ring < rho_end) rho_end = max_ring;
  // (no ring at all when the grid begins beyond max_ring: everything is at least rho0 - 1 + umin cells away)
  float lb_last = rho0 > rho_end ? ((float)(rho0 - 1) + umin) * g.cell - kSlack : 0.0f;
  for (int rho = rho0; rho <= rho_end; rho++) {
    if (g.dense && rho <= kPhasedMaxRing) {
      // PHASED ring (dense per-cell {first slot, #points} table).  The lanes of a wave run in lockstep, so a loop that
      // walks a lane's rows one after the other costs, per trip, the longest chain any lane has in that trip (word ->
      // prefix -> start -> points wherever one lane meets an occupied cell): measured 13.7 us per wave for queries whose
      // neighbour is 10 cm away.  Here every lane first lists kSlots candidate cells of the ring (cube enumeration, the
      // interior and everything the slab bounds rule out dropped), all their table entries are fetched together, and the
      // occupied ones are then visited — the loop trips are the number of occupied cells per lane, not of rows.
      // (The cube enumeration is O(rho^3) against the shell's O(rho^2): rings beyond kPhasedMaxRing — unbounded searches
      // of queries far from every point — take the row walk below.)
      constexpr int kSlots = 8;
      // Rings 0 and 1 in ONE step when the query's own cell is in the grid: every dependent load costs ~1 us here (the
      // table and the points are cold in this XCD's L2 at the start of a launch), and with a seed the cells of ring 1 that
      // can matter are known before ring 0 has been looked at — so their entries are fetched with the own cell's.
      const bool fused = rho == 0 && rho_end >= 1;
      const int inner = rho;       // lowest Chebyshev distance this step covers
      if (fused) rho = 1;          // ... and the highest; the termination test below then speaks for ring 1
      const int w = 2 * rho + 1, ww = w * w, ncube = ww * w;
      const float inv_w = 1.0f / (float)w, inv_ww = 1.0f / (float)ww;
      const float bound2 = reach2(h.d2);  // fixed for the ring: conservative
      for (int base = 0; base < ncube; base += SUB * kSlots) {
        uint2 ent[kSlots];
#pragma unroll
        for (int k = 0; k < kSlots; k++) {
          const int t = base + k * SUB + sub;
          // t -> (dx, dy, dz) in [-rho, rho]^3 (exact: t < 9^3 and the quotients stay far from the rounding edge)
          const int tz = (int)(((float)t + 0.5f) * inv_ww), r2 = t - tz * ww, ty = (int)(((float)r2 + 0.5f) * inv_w), tx = r2 - ty * w;
          const int dx = tx - rho, dy = ty - rho, dz = tz - rho;
          const int ix = c0 + dx, iy = c1 + dy, iz = c2 + dz;
          bool ok = t < ncube && max(abs(dx), max(abs(dy), abs(dz))) >= inner && ix >= 0 && ix < m.dim[0] && iy >= 0 && iy < m.dim[1] && iz >= 0 &&
                    iz < m.dim[2];
          const float gx = dx > 0 ? (float)dx - ux : (dx < 0 ? (float)(-dx - 1) + ux : 0.0f);
          const float gy = dy > 0 ? (float)dy - uy : (dy < 0 ? (float)(-dy - 1) + uy : 0.0f);
          const float gz = dz > 0 ? (float)dz - uz : (dz < 0 ? (float)(-dz - 1) + uz : 0.0f);
          const float mx = fmaxf(gx * g.cell - kSlack, 0.0f), my = fmaxf(gy * g.cell - kSlack, 0.0f), mz = fmaxf(gz * g.cell - kSlack, 0.0f);
          ok = ok && (mx * mx + my * my + mz * mz <= bound2);
          ent[k] = ok ? g.dense[ix + iy * m.dim[0] + iz * m.dim[0] * m.dim[1]] : make_uint2(0u, 0u);
        }
        unsigned todo = 0u;
#pragma unroll
        for (int k = 0; k < kSlots; k++) todo |= ent[k].y ? 1u << k : 0u;
        while (todo) {  // one occupied cell of this lane per trip
          const int k = __ffs(todo) - 1;
          todo &= todo - 1u;
          uint2 e = ent[0];
#pragma unroll
          for (int j = 1; j < kSlots; j++) e = (k == j) ? ent[j] : e;
          nn_visit_range<TRACK>(g, e.x, e.x + e.y, qx, qy, qz, h, sec);
        }
      }
    } else {
    int z0 = c2 - rho, z1 = c2 + rho, y0 = c1 - rho, y1 = c1 + rho, x0 = c0 - rho, x1 = c0 + rho;
    int zc0 = max(z0, 0), zc1 = min(z1, m.dim[2] - 1), yc0 = max(y0, 0), yc1 = min(y1, m.dim[1] - 1), xc0 = max(x0, 0), xc1 = min(x1, m.dim[0] - 1);
    int turn = 0;  // rows are dealt round-robin to the SUB lanes, identical sequence in each of them
    for (int iz = zc0; iz <= zc1; iz++)
      for (int iy = yc0; iy <= yc1; iy++) {
        if (SUB != 1 && ((turn++) & (SUB - 1)) != sub) continue;
        // lower bound of the distance from the query to this row / its cells (exact slab distances, see above)
        const float bound2 = reach2(h.d2);
        const float gy = iy > c1 ? (float)(iy - c1) - uy : (iy < c1 ? (float)(c1 - iy - 1) + uy : 0.0f);
        const float gz = iz > c2 ? (float)(iz - c2) - uz : (iz < c2 ? (float)(c2 - iz - 1) + uz : 0.0f);
        const float my = fmaxf(gy * g.cell - kSlack, 0.0f), mz = fmaxf(gz * g.cell - kSlack, 0.0f);
        const float ryz2 = my * my + mz * mz;
        if (ryz2 > bound2) continue;
        const bool shell = iz == z0 || iz == z1 || iy == y0 || iy == y1;  // a face row: every x of the ring's extent
        if (shell) {
          // cells more than sqrt(bound2 - ryz2) away along x cannot matter: nl cells to the left of c0, nr to the right
          int xa = xc0, xb = xc1;
          if (bound2 != INFINITY) {
            const float rc = (sqrtf(bound2 - ryz2) + kSlack) / g.cell;
            const float tl = rc - ux, tr = rc - (1.0f - ux);
            const int nl = tl < 0.0f ? 0 : (int)fminf(tl, 1.0e6f) + 1, nr = tr < 0.0f ? 0 : (int)fminf(tr, 1.0e6f) + 1;
            xa = max(xc0, c0 - nl);
            xb = min(xc1, c0 + nr);
          }
          if (xa <= xb) nn_visit_row<TRACK>(g, xa, xb, iy, iz, false, qx, qy, qz, h, sec);
        } else {
          // an inner row contributes only the two end cells x0, x1 (when inside the grid and close enough)
          const float ml = fmaxf(((float)(rho - 1) + ux) * g.cell - kSlack, 0.0f), mr = fmaxf(((float)rho - ux) * g.cell - kSlack, 0.0f);
          const bool in0 = x0 >= 0 && x0 < m.dim[0] && ryz2 + ml * ml <= bound2;
          const bool in1 = x1 >= 0 && x1 < m.dim[0] && x1 != x0 && ryz2 + mr * mr <= bound2;
          if (in0 && in1) nn_visit_row<TRACK>(g, x0, x1, iy, iz, true, qx, qy, qz, h, sec);
          else if (in0) nn_visit_row<TRACK>(g, x0, x0, iy, iz, true, qx, qy, qz, h, sec);
          else if (in1) nn_visit_row<TRACK>(g, x1, x1, iy, iz, true, qx, qy, qz, h, sec);
        }
      }
    }
    if (SUB > 1) {  // merge across the SUB lanes (xor butterfly: every lane ends with the group's best)
#pragma unroll
      for (int off = 1; off < SUB; off <<= 1) {
        NNHit o;
        o.d2 = __shfl_xor(h.d2, off, 64); o.id = __shfl_xor(h.id, off, 64);
        o.x = __shfl_xor(h.x, off, 64); o.y = __shfl_xor(h.y, off, 64); o.z = __shfl_xor(h.z, off, 64); o.slot = __shfl_xor(h.slot, off, 64);
        if (TRACK) {  // the group's runner-up: either lane's, or the loser of the two bests when they are different points
          const float osec = __shfl_xor(sec, off, 64);
          sec = fminf(sec, osec);
          if (o.id != h.id) sec = fminf(sec, fmaxf(o.d2, h.d2));
        }
        if (o.d2 < h.d2 || (o.d2 == h.d2 && o.id < h.id)) h = o;
      }
    }
    // every cell not visited yet differs from the query's by more than rho along some axis, i.e. lies at least
    // (rho + umin) cells away
    float lb = ((float)rho + umin) * g.cell - kSlack;
    if (TRACK) lb_last = lb;   // (the rings end where the answer is proven: `widen` never opens a ring of its own)
    if (h.id != 0xffffffffu && lb > 0.0f && h.d2 <= lb * lb) break;
    // ... and nothing farther than the gate is wanted at all
    if (lb > 0.0f && gate_d2 < lb * lb) break;
  }
  if (TRACK && others_lb) {
    // a point that was not visited lies in a cell ruled out against the reach (at least the final one: the reach only
    // shrinks), or beyond the last ring
    const float reach = fminf(gate_r, sqrtf(h.d2)) + widen;
    *others_lb = fminf(fminf(sqrtf(sec), reach), fmaxf(lb_last, 0.0f));
  }
  return h;
}

__device__ __forceinline__ NNHit nn_nearest(const NNView &g, float qx, float qy, float qz, int max_ring) {
  return nn_nearest_sub<1>(g, qx, qy, qz, max_ring, 0);
}

#endif
