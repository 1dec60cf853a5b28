// host_pack.cpp — host-only helpers with no HIP dependency (also built with -fsanitize=address,undefined by
// tests/cpp/Makefile): repacking of caller-owned point rows.
#include <cstddef>
#include <cstdint>
#include <cstring>

// host AoS (pcl::PointXYZI-like rows of `stride` bytes: x, y, z at 0, intensity at w_off) -> packed float4 {x, y, z, w};
// w_off == SIZE_MAX: the cloud carries no intensity, w = 1.  The caller has checked that xyz and the intensity fit the row.
void lsa_pack_xyzw(const void *pts, size_t n, size_t stride, size_t w_off, float *out4) {
  const char *p = (const char *)pts;
  if (w_off != SIZE_MAX) {
    for (size_t i = 0; i < n; i++, p += stride) {
      memcpy(out4 + 4 * i, p, 12);  // x, y, z: one 16-byte move pair for the optimiser instead of three scalar ones
      memcpy(out4 + 4 * i + 3, p + w_off, 4);
    }
  } else {
    for (size_t i = 0; i < n; i++, p += stride) {
      memcpy(out4 + 4 * i, p, 12);
      out4[4 * i + 3] = 1.0f;
    }
  }
}
