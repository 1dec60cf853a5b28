// micro-benchmark: how fast does a kernel pull a 2.1 MB packed scan out of host-mapped pinned memory, against one copy command?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_pull(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_pull4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {   // 4 rows per thread, issued together
  size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float4 v[4];
  for (int k = 0; k < 4; k++) { size_t i = base + k * 256; if (i < n) v[k] = src[i]; }
  for (int k = 0; k < 4; k++) { size_t i = base + k * 256; if (i < n) dst[i] = v[k]; }
}
int main() {
  const size_t n = 131072;
  float4 *h, *d;
  CK(hipHostMalloc((void **)&h, n * 16, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipMalloc((void **)&d, n * 16));
  for (size_t i = 0; i < n; i++) h[i] = make_float4((float)i, 1, 2, 3);
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 3; variant++) {
    for (int rep = 0; rep < 6; rep++) {
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, s));
      if (variant == 0) CK(hipMemcpyAsync(d, h, n * 16, hipMemcpyHostToDevice, s));
      else if (variant == 1) hipLaunchKernelGGL(k_pull, dim3(n / 256), dim3(256), 0, s, h, d, n);
      else hipLaunchKernelGGL(k_pull4, dim3(n / 1024), dim3(256), 0, s, h, d, n);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      auto t1 = std::chrono::steady_clock::now();
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) printf("%s: events %.1f us, host wall %.1f us (%.1f GB/s)\n", variant == 0 ? "hipMemcpyAsync" : variant == 1 ? "k_pull 512 WG" : "k_pull4 128 WG", ms * 1e3,
             std::chrono::duration<double, std::micro>(t1 - t0).count(), n * 16 / (ms * 1e-3) / 1e9);
    }
  }
  float4 chk; CK(hipMemcpy(&chk, d + 777, 16, hipMemcpyDeviceToHost));
  printf("check %g\n", chk.x);
  return 0;
}
