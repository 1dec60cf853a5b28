// host_pack.cpp — host-only helpers with no HIP dependency (also built with -fsanitize=address,undefined by
// tests/cpp/Makefile): repacking of caller-owned point rows.
#include <cstddef>
#include <cstdint>
#include <cstring>

// host AoS (pcl::PointXYZI-like rows of `stride` bytes: x, y, z at 0, intensity at w_off) -> packed float4 {x, y, z, w};
// w_off == SIZE_MAX: the cloud carries no intensity, w = 1.  The caller has checked that xyz and the intensity fit the row.
#if defined(__x86_64__)
#include <immintrin.h>
#include <cstdlib>
// pcl::PointXYZI rows (32 bytes, intensity at 16) into 16-byte-aligned output: one 16-byte load, one 4-byte load, one insert and
// one NON-TEMPORAL 16-byte store per point — the packed cloud is read next by the copy engine, from memory, not by this core, so
// it need not displace the caller's data in the cache (scalar loop: ~0.5 ns per point, this one ~0.25)
__attribute__((target("sse4.1"))) static void pack_xyzi32_sse41(const char *p, size_t n, float *out4) {
  for (size_t i = 0; i < n; i++, p += 32) {
    const __m128 xyz_ = _mm_loadu_ps(reinterpret_cast<const float *>(p));
    const __m128 w = _mm_load_ss(reinterpret_cast<const float *>(p + 16));
    _mm_stream_ps(out4 + 4 * i, _mm_insert_ps(xyz_, w, 0x30));   // lane 3 <- w[0]
  }
  _mm_sfence();
}
#endif

void lsa_pack_xyzw(const void *pts, size_t n, size_t stride, size_t w_off, float *out4) {
  const char *p = (const char *)pts;
#if defined(__x86_64__)
  static const bool fast = __builtin_cpu_supports("sse4.1") && getenv("LSA_PACK_SCALAR") == nullptr;
  if (fast && stride == 32 && w_off == 16 && (reinterpret_cast<uintptr_t>(out4) & 15u) == 0) {
    pack_xyzi32_sse41(p, n, out4);
    return;
  }
#endif
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
