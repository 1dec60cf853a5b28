/*
 * filter_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lsa_oracle.h header).
 *
 * Restates pcl::VoxelGrid<pcl::PointXYZI>::applyFilter as called by
 *   S/models/cloud_filter/voxel_filter.cpp:25 (setLeafSize) and :36-41 (setInputCloud + filter).
 * The arithmetic lives in third-party PCL, pinned by the reference only as
 * `find_package(PCL 1.7 REQUIRED)` (lidar_localization/cmake/PCL.cmake:1) and absent from
 * /root/reference, so this follows the published PCL 1.7.2 algorithm
 * (filters/include/pcl/filters/impl/voxel_grid.hpp, applyFilter) from its text:
 *   1. getMinMax3D over finite points (f32)
 *   2. dx,dy,dz = (int64)((max-min) * inverse_leaf) + 1; if dx*dy*dz > INT32_MAX -> warn, output = input
 *   3. min_b = floor(min * inverse_leaf), max_b likewise, div_b = max_b - min_b + 1, divb_mul = (1, div0, div0*div1)
 *   4. per point: ijk = (int)(floor(x * inverse_leaf) - (float)min_b); idx = ijk . divb_mul
 *   5. sort by idx; one output point per distinct idx (min_points_per_voxel_ = 0), in ascending idx
 *   6. centroid over ALL fields (downsample_all_data_ = true: x, y, z, intensity), f32 accumulate, / (float)count
 * PARITY STATUS: UNPINNED (no PCL here, no reference fixtures). std::sort is unstable, so PCL's
 * f32 summation order inside a voxel is unspecified; this restatement sums in input order.
 * inverse_leaf_size_ = 1.0f / leaf (Eigen::Array4f::Ones() / leaf_size_.array()).
 */
#include "lsa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned int idx; unsigned int pid; } idx_pair;

static int cmp_pair(const void *a, const void *b) {
  const idx_pair *pa = (const idx_pair *)a, *pb = (const idx_pair *)b;
  if (pa->idx != pb->idx) return (pa->idx < pb->idx) ? -1 : 1;
  return (pa->pid < pb->pid) ? -1 : (pa->pid > pb->pid); /* stable: input order inside a voxel */
}

int orc_voxel_filter(const void *pts, size_t n, size_t stride, size_t w_off, float lx, float ly, float lz,
                     float *out, size_t cap, size_t *n_out) {
  *n_out = 0;
  if (n == 0) return 0;
  float inv[3] = {1.0f / lx, 1.0f / ly, 1.0f / lz};
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  size_t finite = 0;
  for (size_t i = 0; i < n; i++) {
    const float *p = (const float *)((const char *)pts + i * stride);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
    finite++;
  }
  if (finite == 0) return 0;
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv[0]) + 1;
  int64_t dy = (int64_t)((mx[1] - mn[1]) * inv[1]) + 1;
  int64_t dz = (int64_t)((mx[2] - mn[2]) * inv[2]) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) {
    /* "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = *input_ */
    size_t k = 0;
    for (size_t i = 0; i < n && k < cap; i++, k++) {
      const float *p = (const float *)((const char *)pts + i * stride);
      out[4 * k] = p[0]; out[4 * k + 1] = p[1]; out[4 * k + 2] = p[2];
      out[4 * k + 3] = *(const float *)((const char *)pts + i * stride + w_off);
    }
    *n_out = n;
    return 1;
  }
  int min_b[3], max_b[3], div_b[3], mul[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = (int)floorf(mn[a] * inv[a]);
    max_b[a] = (int)floorf(mx[a] * inv[a]);
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
  idx_pair *iv = (idx_pair *)malloc(finite * sizeof(idx_pair));
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const float *p = (const float *)((const char *)pts + i * stride);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int ijk0 = (int)(floorf(p[0] * inv[0]) - (float)min_b[0]);
    int ijk1 = (int)(floorf(p[1] * inv[1]) - (float)min_b[1]);
    int ijk2 = (int)(floorf(p[2] * inv[2]) - (float)min_b[2]);
    int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
    iv[m].idx = (unsigned int)idx; iv[m].pid = (unsigned int)i; m++;
  }
  qsort(iv, m, sizeof(idx_pair), cmp_pair);
  size_t total = 0, index = 0;
  while (index < m) {
    size_t i = index + 1;
    while (i < m && iv[i].idx == iv[index].idx) ++i;
    if (total < cap) {
      float c[4];
      const char *p0 = (const char *)pts + (size_t)iv[index].pid * stride;
      c[0] = ((const float *)p0)[0]; c[1] = ((const float *)p0)[1]; c[2] = ((const float *)p0)[2];
      c[3] = *(const float *)(p0 + w_off);
      for (size_t j = index + 1; j < i; j++) {
        const char *pj = (const char *)pts + (size_t)iv[j].pid * stride;
        c[0] += ((const float *)pj)[0]; c[1] += ((const float *)pj)[1]; c[2] += ((const float *)pj)[2];
        c[3] += *(const float *)(pj + w_off);
      }
      float cnt = (float)(i - index);
      out[4 * total] = c[0] / cnt; out[4 * total + 1] = c[1] / cnt; out[4 * total + 2] = c[2] / cnt; out[4 * total + 3] = c[3] / cnt;
    }
    total++;
    index = i;
  }
  free(iv);
  *n_out = total;
  return 0;
}

/* ---- BoxFilter (S/models/cloud_filter/box_filter.cpp:27-37,62-68 -> pcl::CropBox<PointXYZI>::applyFilter, PCL 1.7
 * filters/impl/crop_box.hpp; identity transform, negative_ = false; third-party: UNPINNED) ------------------------- */
int orc_box_filter(const void *pts, size_t n, size_t stride, size_t w_off, const float size[6], const float origin[3],
                   float *out_xyzi, size_t cap, size_t *n_out) {
  float edge[6];
  for (int i = 0; i < 3; i++) { edge[2 * i] = size[2 * i] + origin[i]; edge[2 * i + 1] = size[2 * i + 1] + origin[i]; }
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const float *p = (const float *)((const char *)pts + i * stride);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue; /* !input_->is_dense branch */
    if ((p[0] < edge[0] || p[1] < edge[2] || p[2] < edge[4]) || (p[0] > edge[1] || p[1] > edge[3] || p[2] > edge[5])) continue;
    if (m >= cap) return -1;
    out_xyzi[4 * m] = p[0]; out_xyzi[4 * m + 1] = p[1]; out_xyzi[4 * m + 2] = p[2];
    out_xyzi[4 * m + 3] = *(const float *)((const char *)p + w_off);
    m++;
  }
  *n_out = m;
  return 0;
}
