// r04_graph_chain.hip — would a hipGraph shorten the frame filter's chain (four dependent ~7 us kernels, `k_mark -> k_scan -> k_accum ->
// k_filter_finalize`)?  Stand-alone measurement, not part of the library: four dependent kernels of a set duration (256 workgroups each, every
// wave spins on the 100 MHz wall clock) launched (a) one by one on a stream, (b) as ONE captured graph; per repetition the caller waits for the
// chain (as lsa_voxel_filter_wait does).  Reported: host time to enqueue, wall time enqueue -> chain complete, both per chain.
// build: hipcc -O2 --offload-arch=gfx950 r04_graph_chain.hip -o /tmp/graph_chain     run: /tmp/graph_chain [us per kernel]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ void k_spin(unsigned long long ticks, unsigned long long *sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0 && blockIdx.x == 0) *sink = t0;
}
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main(int argc, char **argv) {
  const double us = argc > 1 ? atof(argv[1]) : 7.0;
  const unsigned long long ticks = (unsigned long long)(us * 100.0);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *sink; CK(hipMalloc(&sink, 8));
  const int kKernels = 4, reps = 2000;
  auto chain = [&]() { for (int k = 0; k < kKernels; k++) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, ticks, sink); };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int mode = 0; mode < 2; mode++) {
    std::vector<double> enq, tot;
    for (int r = 0; r < reps + 100; r++) {
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) chain(); else CK(hipGraphLaunch(ge, s));
      const auto t1 = std::chrono::steady_clock::now();
      while (hipStreamQuery(s) == hipErrorNotReady) {}
      const auto t2 = std::chrono::steady_clock::now();
      if (r >= 100) { enq.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count()); tot.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count()); }
    }
    printf("%s: %d kernels of %.1f us: host enqueue %.1f us, enqueue -> complete %.1f us (medians of %d)\n", mode == 0 ? "stream launches" : "one graph launch ", kKernels, us, median(enq), median(tot), reps);
  }
  return 0;
}
