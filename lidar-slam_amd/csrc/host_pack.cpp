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

// ---- check-worded workgroup rows (ndt_derivatives.hip: ndt_reduce_partials_host), the rows that are already there --------------
// A pass delivers one 32-double row per workgroup into host-mapped memory; column 31 = key ^ XOR_c rotl(word_c, c + 1) over the
// other 31 words (voxel_index_dev.h: lsa_store_row32).  The calling thread sits between the last row's arrival and the next pass's
// mail, so what it spends per row is on the critical path of every ScanMatch: 31 scalar rotates + XORs, a 256-byte copy and 32
// scalar additions per row, 135 rows per KITTI-sized pass.  Here: eight 32-byte loads, the rotation by per-lane shift counts, the
// additions four at a time straight into the partial sums.  Takes rows b, b + 1, ... while their check words match and returns the
// index of the first one that does not (the caller waits for that one the ordinary, timed way and comes back).  Same additions in
// the same order as the scalar loop: sm[b % 32][col] += row_b[col], ascending b.
#if defined(__x86_64__)
bool lsa_rows_fast_available() {
  static const bool ok = __builtin_cpu_supports("avx2") && getenv("LSA_ROWS_SCALAR") == nullptr;
  return ok;
}
__attribute__((target("avx2"))) static unsigned rows_take_avx2(const double *partials, unsigned b, unsigned nb, unsigned long long key, double (*sm)[32]) {
  __m256i shl[8], shr[8];
  for (int k = 0; k < 8; k++) {
    shl[k] = _mm256_set_epi64x(4 * k + 4, 4 * k + 3, 4 * k + 2, 4 * k + 1);   // lane i of vector k is column 4k + i: rotate by column + 1
    shr[k] = _mm256_sub_epi64(_mm256_set1_epi64x(64), shl[k]);
  }
  const __m256i payload = _mm256_set_epi64x(0, -1, -1, -1);   // of the last vector: columns 28..30; column 31 is the check word
  for (; b < nb; b++) {
    const __m256i *row = reinterpret_cast<const __m256i *>(partials + (size_t)b * 32);
    __m256i w[8], x = _mm256_setzero_si256();
    for (int k = 0; k < 8; k++) w[k] = _mm256_loadu_si256(row + k);
    const unsigned long long check = (unsigned long long)_mm256_extract_epi64(w[7], 3);
    w[7] = _mm256_and_si256(w[7], payload);   // (a zero rotates to zero; and the sums' column 31 takes +0.0, as in the scalar loop)
    for (int k = 0; k < 8; k++) x = _mm256_xor_si256(x, _mm256_or_si256(_mm256_sllv_epi64(w[k], shl[k]), _mm256_srlv_epi64(w[k], shr[k])));
    const __m128i y = _mm_xor_si128(_mm256_castsi256_si128(x), _mm256_extracti128_si256(x, 1));
    const unsigned long long xx = (unsigned long long)_mm_extract_epi64(y, 0) ^ (unsigned long long)_mm_extract_epi64(y, 1);
    if ((xx ^ key) != check) break;
    double *dst = sm[b & 31];
    for (int k = 0; k < 8; k++) _mm256_storeu_pd(dst + 4 * k, _mm256_add_pd(_mm256_loadu_pd(dst + 4 * k), _mm256_castsi256_pd(w[k])));
  }
  return b;
}
// the same with 64-byte registers and the rotate instruction AVX-512 has (vprolvq): four loads, four rotates, four additions per row
__attribute__((target("avx512f"))) static unsigned rows_take_avx512(const double *partials, unsigned b, unsigned nb, unsigned long long key, double (*sm)[32]) {
  __m512i rot[4];
  for (int k = 0; k < 4; k++) rot[k] = _mm512_set_epi64(8 * k + 8, 8 * k + 7, 8 * k + 6, 8 * k + 5, 8 * k + 4, 8 * k + 3, 8 * k + 2, 8 * k + 1);
  const __mmask8 payload = 0x7f;   // of the last vector: columns 24..30; column 31 is the check word
  for (; b < nb; b++) {
    const char *row = reinterpret_cast<const char *>(partials + (size_t)b * 32);
    __m512i w[4];
    for (int k = 0; k < 4; k++) w[k] = _mm512_loadu_si512(row + 64 * k);
    const unsigned long long check = (unsigned long long)_mm_extract_epi64(_mm512_extracti32x4_epi32(w[3], 3), 1);
    w[3] = _mm512_maskz_mov_epi64(payload, w[3]);
    const __m512i x = _mm512_xor_si512(_mm512_xor_si512(_mm512_rolv_epi64(w[0], rot[0]), _mm512_rolv_epi64(w[1], rot[1])),
                                       _mm512_xor_si512(_mm512_rolv_epi64(w[2], rot[2]), _mm512_rolv_epi64(w[3], rot[3])));
    const __m256i y4 = _mm256_xor_si256(_mm512_castsi512_si256(x), _mm512_extracti64x4_epi64(x, 1));
    const __m128i y2 = _mm_xor_si128(_mm256_castsi256_si128(y4), _mm256_extracti128_si256(y4, 1));
    const unsigned long long xx = (unsigned long long)_mm_extract_epi64(y2, 0) ^ (unsigned long long)_mm_extract_epi64(y2, 1);
    if ((xx ^ key) != check) break;
    double *dst = sm[b & 31];
    for (int k = 0; k < 4; k++) _mm512_storeu_pd(dst + 8 * k, _mm512_add_pd(_mm512_loadu_pd(dst + 8 * k), _mm512_castsi512_pd(w[k])));
  }
  return b;
}
unsigned lsa_rows_take_fast(const double *partials, unsigned b, unsigned nb, unsigned long long key, double (*sm)[32]) {
  static const bool wide = __builtin_cpu_supports("avx512f") && getenv("LSA_ROWS_AVX2") == nullptr;
  return wide ? rows_take_avx512(partials, b, nb, key, sm) : rows_take_avx2(partials, b, nb, key, sm);
}
#else
bool lsa_rows_fast_available() { return false; }
unsigned lsa_rows_take_fast(const double *, unsigned b, unsigned, unsigned long long, double (*)[32]) { return b; }
#endif
