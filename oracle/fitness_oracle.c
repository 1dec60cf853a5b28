/*
 * fitness_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lsa_oracle.h header).
 *
 * Restates the part of NS/Octree.cpp that NormalDistributionsTransform::getFitnessScore reaches:
 *   Octree::setInput            NS/Octree.cpp:50-228   (leaf = 4x4x2 voxels, Octree.cpp:20-22)
 *   Octree::nearestOctreeNode   NS/Octree.cpp:638-667  (initRange :703-806, goUp :863-914, goDown :810-859)
 * PARITY STATUS: UNPINNED (restated from text).
 *
 * The hierarchical search prunes a subtree only when the distance from q to the subtree's point
 * bounding box exceeds the best leaf-centroid distance found so far (goDown :835-839); a leaf's
 * centroid lies inside its box, so the prune is a valid lower bound and the search returns the
 * leaf with the globally smallest ||centroid - q||.  It is therefore restated as an argmin over
 * occupied leaves (first-minimum in ascending node id on exact ties; ties have measure zero).
 * Note: the reference's goDown falls through at level 0 into reserved_size_[-1] (:842-843, undefined
 * behaviour); the restatement returns at the leaf, the evident intent.
 */
#include "lsa_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float lx, ux, ly, uy, lz, uz;
  double c[3];
  int point_num;
} onode;

typedef struct {
  int lo[3], dim[3];
  onode *nodes;
  size_t count;
} octree;

static int fdiv(int input, int divisor) { /* NS/Octree.cpp:44-47 */
  return (input < 0) ? (-(-input + divisor - 1) / divisor) : (input / divisor);
}

void *orc_octree_build(const void *pts, size_t n, size_t stride, float leaf) {
  const int LX = 4, LY = 4, LZ = 2;
  octree *o = (octree *)calloc(1, sizeof(octree));
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (size_t i = 0; i < n; i++) {
    const float *p = (const float *)((const char *)pts + i * stride);
    int v[3] = {(int)floorf(p[0] / leaf), (int)floorf(p[1] / leaf), (int)floorf(p[2] / leaf)}; /* VoxelGrid.cpp:344-346 */
    int nd[3] = {fdiv(v[0], LX), fdiv(v[1], LY), fdiv(v[2], LZ)};
    for (int a = 0; a < 3; a++) { if (nd[a] < mn[a]) mn[a] = nd[a]; if (nd[a] > mx[a]) mx[a] = nd[a]; }
  }
  for (int a = 0; a < 3; a++) { o->lo[a] = mn[a]; o->dim[a] = mx[a] - mn[a] + 1; }
  o->count = (size_t)o->dim[0] * o->dim[1] * o->dim[2];
  o->nodes = (onode *)calloc(o->count, sizeof(onode));
  for (size_t i = 0; i < n; i++) {
    const float *p = (const float *)((const char *)pts + i * stride);
    int v[3] = {(int)floorf(p[0] / leaf), (int)floorf(p[1] / leaf), (int)floorf(p[2] / leaf)};
    int nd[3] = {fdiv(v[0], LX) - o->lo[0], fdiv(v[1], LY) - o->lo[1], fdiv(v[2], LZ) - o->lo[2]};
    onode *nn = &o->nodes[(size_t)nd[0] + (size_t)nd[1] * o->dim[0] + (size_t)nd[2] * o->dim[0] * o->dim[1]];
    if (nn->point_num == 0) { /* Octree.cpp:166-177 */
      nn->lx = nn->ux = p[0]; nn->ly = nn->uy = p[1]; nn->lz = nn->uz = p[2];
      nn->point_num = 1;
      nn->c[0] = p[0]; nn->c[1] = p[1]; nn->c[2] = p[2];
    } else { /* :182-212 */
      if (p[0] < nn->lx) nn->lx = p[0];
      if (p[1] < nn->ly) nn->ly = p[1];
      if (p[2] < nn->lz) nn->lz = p[2];
      if (p[0] > nn->ux) nn->ux = p[0];
      if (p[1] > nn->uy) nn->uy = p[1];
      if (p[2] > nn->uz) nn->uz = p[2];
      for (int a = 0; a < 3; a++) nn->c[a] = nn->c[a] * nn->point_num + (double)p[a];
      nn->point_num++;
      for (int a = 0; a < 3; a++) nn->c[a] /= nn->point_num;
    }
  }
  return o;
}

void orc_octree_free(void *ov) {
  octree *o = (octree *)ov;
  if (!o) return;
  free(o->nodes);
  free(o);
}

void orc_octree_nearest(void *ov, const float q[3], float bounds[6]) {
  octree *o = (octree *)ov;
  double best = DBL_MAX;
  size_t bi = 0;
  for (size_t i = 0; i < o->count; i++) {
    const onode *nn = &o->nodes[i];
    if (nn->point_num == 0) continue;
    double d = sqrt((nn->c[0] - q[0]) * (nn->c[0] - q[0]) + (nn->c[1] - q[1]) * (nn->c[1] - q[1]) + (nn->c[2] - q[2]) * (nn->c[2] - q[2]));
    if (d < best) { best = d; bi = i; }
  }
  const onode *nn = &o->nodes[bi];
  bounds[0] = nn->lx; bounds[1] = nn->ly; bounds[2] = nn->lz;
  bounds[3] = nn->ux; bounds[4] = nn->uy; bounds[5] = nn->uz;
}
