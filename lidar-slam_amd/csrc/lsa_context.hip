// lsa_context.hip — context lifecycle, error text, host-side packing helpers.
#include "lsa_common.h"

#include <atomic>
#include <mutex>

#include <cstdlib>

static thread_local std::string g_global_err;
// ctx->err is written by whichever thread fails (the async frame filter reports from its helper thread) and read by the
// caller's: both sides go through this mutex, and lsa_last_error hands out a per-thread copy
static std::mutex g_err_mu;

void lsa_set_global_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_global_err = buf;
}

void lsa_alloc_trace(const char *kind, size_t bytes, const char *who) {
  static const bool on = getenv("LSA_ALLOC_TRACE") && atoi(getenv("LSA_ALLOC_TRACE")) != 0;
  if (on) fprintf(stderr, "[lsa alloc] %s %zu B  %s\n", kind, bytes, who);
}

int lsa_fail(lsa_context *ctx, int code, const char *fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  {
    std::lock_guard<std::mutex> lock(g_err_mu);
    if (ctx) ctx->err = buf;
  }
  g_global_err = buf;
  return code;
}

// Host points -> device float4: packed into the pinned staging buffer and copied in a few chunks, each copy enqueued as
// soon as its chunk is packed, so the PCIe transfer of chunk k runs under the packing of chunk k + 1 (a 131 k-point scan:
// ~70 us of packing with ~40 us of copy behind it instead of after it).  The caller owns the ordering of h_stage reuse.
size_t lsa_upload_chunk() {
  static const size_t c = getenv("LSA_UPLOAD_CHUNK") ? (size_t)atol(getenv("LSA_UPLOAD_CHUNK")) : 65536;   // points per copy command: 1 MB (two of a KITTI scan run side by side, one per copy queue: 115-118 us per host Filter against 119-123 with four of 512 KB)
  return c < 1024 ? 1024 : c;
}

// Small clouds (a filtered scan: 7-25 k rows, 120-400 KB) are taken out of the pinned staging buffer by a KERNEL, not by a copy command:
// the bytes need ~4 us on the link either way, but a copy between two kernels of one stream is a change of engine on both sides (queue ->
// SDMA -> queue: two cross-engine dependencies) and the runtime's copy path costs the host more than a launch.  Measured on the
// host-pointer ScanMatch of a 7.7 k-point source: 91-98 -> 68 us per call (profiles/r04_host_boundary_ab.log).
constexpr size_t kPullRowsMax = 65536;
__global__ __launch_bounds__(256) void k_pull_rows(const float4 *__restrict__ h_rows, float4 *__restrict__ d_rows, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d_rows[i] = h_rows[i];
}

// packed rows in pinned memory -> device, by a kernel on `stream` (any size; no copy command: the first one on a fresh stream
// sets up a copy queue, measured at ~7 ms)
hipError_t lsa_pull_rows(hipStream_t stream, const float *h_rows, float4 *d_rows, size_t n) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_pull_rows, dim3(lsa_div_up(n, 256)), dim3(256), 0, stream, reinterpret_cast<const float4 *>(h_rows), d_rows, n);
  return hipGetLastError();
}

int lsa_upload_xyzw(lsa_context *ctx, const void *pts, size_t n, size_t stride, size_t w_off, float *h_stage, float4 *d_dst) {
  // an intensity offset that does not fit the row would read the next point (or past the buffer): (n,3) xyz rows have
  // no intensity and must say so with w_off == SIZE_MAX
  if (n && (stride < 12 || (w_off != SIZE_MAX && w_off + 4 > stride)))
    return lsa_fail(ctx, LSA_ERR_INVALID, "point rows of %zu bytes cannot hold xyz + an intensity at offset %zu (pass SIZE_MAX for clouds without one)", stride, w_off);
  if (n && n <= kPullRowsMax) {
    lsa_pack_xyzw(pts, n, stride, w_off, h_stage);
    hipLaunchKernelGGL(k_pull_rows, dim3(lsa_div_up(n, 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const float4 *>(h_stage), d_dst, n);
    LSA_HIP(ctx, hipGetLastError());
    return LSA_OK;
  }
  const size_t kChunk = lsa_upload_chunk();
  for (size_t off = 0; off < n; off += kChunk) {
    const size_t m = n - off < kChunk ? n - off : kChunk;
    lsa_pack_xyzw((const char *)pts + off * stride, m, stride, w_off, h_stage + 4 * off);
    // (a kernel per chunk, not a copy command: the same bytes at the link's rate, no copy queue to set up on a stream that has never
    // copied — ~7 ms the first time —, no change of engine between this stream's kernels)
    LSA_HIP(ctx, lsa_pull_rows(ctx->stream, h_stage + 4 * off, d_dst + off, m));
  }
  return LSA_OK;
}

unsigned long long lsa_seq_base() {
  static std::atomic<unsigned long long> next{1};
  return (next.fetch_add(1, std::memory_order_relaxed) & 0x3ffffffull) << 36;
}

extern "C" int lsa_version(void) { return LSA_VERSION_MAJOR * 100 + LSA_VERSION_MINOR; }

extern "C" const char *lsa_last_global_error(void) { return g_global_err.c_str(); }

extern "C" const char *lsa_last_error(const lsa_context *ctx) {
  if (!ctx) return g_global_err.c_str();
  static thread_local std::string copy;
  {
    std::lock_guard<std::mutex> lock(g_err_mu);
    copy = ctx->err;
  }
  return copy.c_str();
}

extern "C" int lsa_context_create(lsa_context **out, int device, void *hip_stream) {
  if (!out) { lsa_set_global_error("lsa_context_create: out is null"); return LSA_ERR_INVALID; }
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    lsa_set_global_error("lsa_context_create: no HIP device available (%s); this library has no CPU path",
                         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return LSA_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    lsa_set_global_error("lsa_context_create: device %d out of range (have %d)", device, count);
    return LSA_ERR_NO_DEVICE;
  }
  // The hot path waits for a ~10 us kernel dozens of times per scan: ask the runtime to spin rather than block in
  // hipStreamSynchronize / hipEventSynchronize (refused once the device is active in this process, which is harmless).
  if (!getenv("LSA_NO_SPIN")) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
  (void)hipGetLastError();
  e = hipSetDevice(device);
  if (e != hipSuccess) { lsa_set_global_error("hipSetDevice(%d): %s", device, hipGetErrorString(e)); return LSA_ERR_NO_DEVICE; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { lsa_set_global_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("LSA_ALLOW_ANY_ARCH")) {
    lsa_set_global_error("lsa_context_create: device %d is %s; this library ships gfx950 (MI355X) code objects only", device, prop.gcnArchName);
    return LSA_ERR_NO_DEVICE;
  }
  lsa_context *ctx = new lsa_context();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else {
    // LSA_STREAM_PRIORITY (experiments): the HIP priority of the streams this library creates itself (-1 = high on this runtime)
    static const char *prio_env = getenv("LSA_STREAM_PRIORITY");
    if (prio_env) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, atoi(prio_env));
    else e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { lsa_set_global_error("hipStreamCreate: %s", hipGetErrorString(e)); delete ctx; return LSA_ERR_HIP; }
    ctx->own_stream = true;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    lsa_set_global_error("hipEventCreate failed");
    delete ctx;
    return LSA_ERR_HIP;
  }
  *out = ctx;
  return LSA_OK;
}

extern "C" void lsa_context_destroy(lsa_context *ctx) {
  if (!ctx) return;
  (void)lsa_enter(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" void *lsa_context_stream(lsa_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int lsa_context_synchronize(lsa_context *ctx) {
  if (!ctx) return LSA_ERR_INVALID;
  LSA_HIP(ctx, lsa_enter(ctx));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}
