/*
 * yaw_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY): Matching's position-only initialisation
 * (S/matching/matching.cpp): generateGauss2DMapCells (:344-394) + resetMapRange (:396-422) build a 2-D grid of per-cell
 * height statistics over the local map, getInitialYawAngle (:267-307) scores 270 yaw hypotheses of the current scan
 * against it and returns the best angle.  `init_type: OnlyPosition` is the default of config/matching/matching.yaml:4.
 * PARITY STATUS: restated from the reference text (the file needs PCL / ROS headers: unbuildable here), UNPINNED.
 * The mixed float / double / int arithmetic of the text is kept operation by operation.
 */
#include "lsa_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_yaw {
  double res;       /* grid_map_resolution_ = max(0.1, grid_resolution) (:137-138) */
  int width, height;
  float mn[3], mx[3];
  float *mu, *sigma;
  int *cnt;
};

static inline const float *pxyz(const void *pts, size_t stride, size_t i) {
  return (const float *)((const char *)pts + i * stride);
}

void orc_yaw_free(orc_yaw *y) {
  if (!y) return;
  free(y->mu); free(y->sigma); free(y->cnt); free(y);
}

orc_yaw *orc_yaw_build(const void *map, size_t n, size_t stride, const float origin[3], double grid_resolution) {
  orc_yaw *y = (orc_yaw *)calloc(1, sizeof(orc_yaw));
  y->res = grid_resolution > 0.1 ? grid_resolution : 0.1;
  for (int a = 0; a < 3; a++) { y->mn[a] = FLT_MAX; y->mx[a] = -FLT_MAX; } /* SetInitPose :331-334 */
  float *c = (float *)malloc((n ? n : 1) * 3 * sizeof(float));
  size_t m = 0;
  for (size_t i = 0; i < n; i++) { /* :348-355 */
    const float *p = pxyz(map, stride, i);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    float pc[3] = {p[0] - origin[0], p[1] - origin[1], p[2] - origin[2]};
    memcpy(c + m * 3, pc, sizeof(pc));
    m++;
    for (int a = 0; a < 3; a++) { /* resetMapRange */
      if (pc[a] < y->mn[a]) y->mn[a] = pc[a];
      if (pc[a] > y->mx[a]) y->mx[a] = pc[a];
    }
  }
  y->width = (int)round((double)(y->mx[0] - y->mn[0]) / y->res);  /* :357-358: float difference / double */
  y->height = (int)round((double)(y->mx[1] - y->mn[1]) / y->res);
  if (m == 0 || y->width < 0 || y->height < 0) { y->width = y->height = 0; }
  size_t nc = (size_t)y->width * (size_t)y->height;
  y->mu = (float *)calloc(nc ? nc : 1, sizeof(float));
  y->sigma = (float *)calloc(nc ? nc : 1, sizeof(float));
  y->cnt = (int *)calloc(nc ? nc : 1, sizeof(int));
  for (size_t i = 0; i < m; i++) { /* :367-393 */
    const float *pt = c + i * 3;
    int cell_x = (int)round((double)(pt[0] - y->mn[0]) / y->res);
    int cell_y = (int)round((double)(pt[1] - y->mn[1]) / y->res);
    if (cell_x < 0 || cell_y < 0 || cell_x >= y->width || cell_y >= y->height) continue;
    size_t k = (size_t)cell_x * (size_t)y->height + (size_t)cell_y;
    float mu = y->mu[k], sigma = y->sigma[k];
    int point_cnt = y->cnt[k];
    if (point_cnt == 0) {
      y->mu[k] = pt[2];
      y->sigma[k] = 0;
      y->cnt[k] = 1;
    } else {
      float new_mu = ((float)point_cnt * mu + pt[2]) / (float)(point_cnt + 1);
      y->mu[k] = new_mu;
      /* (cnt-1)*sigma [float] + pow(z-mu,2) [double] + (cnt+1)*pow(new_mu-mu,2) [double] + 2*(new_mu-mu)*(z-new_mu) [float] */
      float t0 = (float)(point_cnt - 1) * sigma;
      float d1 = pt[2] - mu, d2 = new_mu - mu;
      double acc = (double)t0 + (double)d1 * (double)d1;
      acc = acc + (double)(point_cnt + 1) * ((double)d2 * (double)d2);
      float t3 = (2.0f * (new_mu - mu)) * (pt[2] - new_mu);
      acc = acc + (double)t3;
      float s = (float)acc;
      s = s / (float)point_cnt;
      y->sigma[k] = s;
      y->cnt[k] = point_cnt + 1;
    }
  }
  free(c);
  return y;
}

void orc_yaw_info(const orc_yaw *y, int wh[2], float mn[3], float mx[3]) {
  wh[0] = y->width; wh[1] = y->height;
  memcpy(mn, y->mn, sizeof(y->mn));
  memcpy(mx, y->mx, sizeof(y->mx));
}

void orc_yaw_cells(const orc_yaw *y, float *mu, float *sigma, int *cnt) {
  size_t nc = (size_t)y->width * (size_t)y->height;
  memcpy(mu, y->mu, nc * sizeof(float));
  memcpy(sigma, y->sigma, nc * sizeof(float));
  memcpy(cnt, y->cnt, nc * sizeof(int));
}

/* Eigen::AngleAxisf(angle, UnitZ).matrix() (Eigen 3.2.92 Geometry/AngleAxis.h toRotationMatrix), f32 */
void orc_yaw_rotation(int i, float R[9]) {
  float delta_angle = (float)(2 * M_PI / 270); /* float delta_angle = 2 * M_PI / angle_size (:270) */
  float angle = delta_angle * (float)i;
  float s = sinf(angle), c = cosf(angle);
  for (int k = 0; k < 9; k++) R[k] = 0.0f;
  R[0] = c; R[1] = -s; R[3] = s; R[4] = c;
  R[8] = (1.0f - c) * 1.0f + c;
}

double orc_yaw_angle(const orc_yaw *y, const void *scan, size_t n, size_t stride, double *probs270) {
  const int angle_size = 270;
  float delta_angle = (float)(2 * M_PI / angle_size);
  float max_prob = -FLT_MAX;
  double probs[270];
  for (int i = 0; i < angle_size; i++) {
    float R[9], T[16];
    orc_yaw_rotation(i, R);
    memset(T, 0, sizeof(T));
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) T[c * 4 + r] = R[r * 3 + c];
    T[15] = 1.0f;
    probs[i] = 0;
    for (size_t j = 0; j < n; j++) {
      float q[3];
      orc_transform_point(T, pxyz(scan, stride, j), q); /* pcl::transformPointCloud, f32 */
      if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
      int cell_x = (int)round((double)(q[0] - y->mn[0]) / y->res);
      int cell_y = (int)round((double)(q[1] - y->mn[1]) / y->res);
      if (cell_x < 0 || cell_y < 0 || cell_x >= y->width || cell_y >= y->height) continue;
      size_t k = (size_t)cell_x * (size_t)y->height + (size_t)cell_y;
      if (y->cnt[k] == 0) continue;
      float mu = y->mu[k], sigma = y->sigma[k];
      float dz = q[2] - mu;
      probs[i] += exp(-((double)dz * (double)dz) / (double)(2.0f * sigma)); /* -pow(z - mu, 2) / (2 * sigma) (:296) */
    }
  }
  double max_prob_angle = 0;
  for (int it = 0; it < angle_size; it++) {
    if (probs[it] > (double)max_prob) { /* max_prob is a float (:271) */
      max_prob = (float)probs[it];
      max_prob_angle = (double)((float)it * delta_angle);
    }
  }
  if (probs270) memcpy(probs270, probs, sizeof(probs));
  return max_prob_angle;
}
