// lsa_common.h — internal: context, grow-only device buffers, error plumbing.
#ifndef LSA_COMMON_H_
#define LSA_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "lidar_slam_amd.h"

struct lsa_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cus = 256;
  std::string err;
  // A handle may leave a kernel PARKED on this context's stream between two calls (an armed NDT pass waiting for the next
  // ScanMatch's parameters, ndt_api.hip).  Everything else that is about to use the stream releases it first.
  void (*unpark)(void *) = nullptr;
  void *unpark_arg = nullptr;
};

inline void lsa_ctx_unpark(lsa_context *ctx, const void *except = nullptr) {
  if (ctx->unpark && ctx->unpark_arg != except) ctx->unpark(ctx->unpark_arg);
}
// entry of every public call that touches the stream: release a parked kernel, select the device
inline hipError_t lsa_enter(lsa_context *ctx) {
  lsa_ctx_unpark(ctx);
  return hipSetDevice(ctx->device);
}

// Base of a handle's pass / mail sequence numbers: distinct for every handle of the process, so that the check words
// derived from them (rows handed to the host as they arrive, the armed passes' mail) can never match what another handle
// — destroyed, its pinned memory recycled — left behind.  Bit 63 stays clear (the mail's cancel flag); 2^36 numbers per handle.
unsigned long long lsa_seq_base();
size_t lsa_upload_chunk();   // points per host -> device copy command (lsa_upload_xyzw)

void lsa_set_global_error(const char *fmt, ...);
int lsa_fail(lsa_context *ctx, int code, const char *fmt, ...);

#define LSA_HIP(ctx, call)                                                                          \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess)                                                                          \
      return lsa_fail((ctx), (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice) ? LSA_ERR_NO_DEVICE : LSA_ERR_HIP, \
                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// diagnostics (LSA_ALLOC_TRACE=1): one line on stderr per device / pinned allocation, so that a loop that should not allocate shows that it does not
void lsa_alloc_trace(const char *kind, size_t bytes, const char *who);

// grow-only device allocation
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;  // elements
  int reserve(lsa_context *ctx, size_t n, double headroom = 1.0) {
    if (n <= cap) return LSA_OK;
    size_t want = (size_t)((double)n * headroom);
    if (want < n) want = n;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    lsa_alloc_trace("device", want * sizeof(T), __PRETTY_FUNCTION__);
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e != hipSuccess) return lsa_fail(ctx, LSA_ERR_HIP, "hipMalloc(%zu B) failed: %s", want * sizeof(T), hipGetErrorString(e));
    cap = want;
    return LSA_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
};

template <typename T>
struct PinnedBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve(lsa_context *ctx, size_t n) {
    if (n <= cap) return LSA_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = n + n / 4;
    lsa_alloc_trace("pinned", want * sizeof(T), __PRETTY_FUNCTION__);
    hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return lsa_fail(ctx, LSA_ERR_HIP, "hipHostMalloc(%zu B) failed: %s", want * sizeof(T), hipGetErrorString(e));
    // The runtime recycles pinned memory WITH its contents.  Several of these buffers receive check-worded rows that the
    // host accepts by key the moment they look whole: a block inherited from a destroyed handle must not offer it that
    // handle's rows (see lsa_seq_base: the keys are per-handle too).
    memset(p, 0, want * sizeof(T));
    cap = want;
    return LSA_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
  }
};

// host AoS (pcl::PointXYZI-like) -> packed float4 {x,y,z,w}; w_off == SIZE_MAX -> w = 1
void lsa_pack_xyzw(const void *pts, size_t n, size_t stride, size_t w_off, float *out4);
int lsa_upload_xyzw(lsa_context *ctx, const void *pts, size_t n, size_t stride, size_t w_off, float *h_stage, float4 *d_dst);
hipError_t lsa_pull_rows(hipStream_t stream, const float *h_rows, float4 *d_rows, size_t n);

static inline unsigned lsa_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

#endif
