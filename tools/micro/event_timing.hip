// Developer microbenchmark: which in-process timer agrees with rocprofv3's kernel duration?
//
// A kernel of known length (every wave spins on the 100 MHz wall clock for `us`
// microseconds) is launched N times back to back on one stream and timed four ways:
//   (a) hipEventRecord before / after every launch (what the library did in round 1),
//   (b) hipExtLaunchKernelGGL with a start and a stop event,
//   (c) hipExtLaunchKernelGGL with a stop event only, elapsed(stop, stop),
//   (d) wall time of the N launches / N.
// Run under `rocprofv3 --kernel-trace --stats` to get the profiler's figure next to them.
//
// hipcc -O3 --offload-arch=gfx950 tools/micro/event_timing.hip -o tools/micro/event_timing
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                             \
  do {                                                                       \
    hipError_t e = (x);                                                      \
    if (e != hipSuccess) {                                                   \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                   \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();
  if (sink != nullptr && t == 1) *sink = t;
}

int main(int argc, char** argv) {
  const double us = argc > 1 ? atof(argv[1]) : 35.0;
  const int n = argc > 2 ? atoi(argv[2]) : 2000;
  const unsigned long long ticks = (unsigned long long)(us * 100.0);
  const dim3 grid(512), block(256);
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  std::vector<hipEvent_t> e0(n), e1(n);
  for (int i = 0; i < n; ++i) {
    CHECK(hipEventCreate(&e0[i]));
    CHECK(hipEventCreate(&e1[i]));
  }
  auto mean = [&](bool same) {
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, same ? e1[i] : e0[i], e1[i]));
      total += ms;
    }
    return total / n * 1e3;
  };
  for (int i = 0; i < 200; ++i)
    hipLaunchKernelGGL(spin_kernel, grid, block, 0, stream, ticks, (unsigned long long*)nullptr);
  CHECK(hipStreamSynchronize(stream));

  for (int i = 0; i < n; ++i) {
    CHECK(hipEventRecord(e0[i], stream));
    hipLaunchKernelGGL(spin_kernel, grid, block, 0, stream, ticks, (unsigned long long*)nullptr);
    CHECK(hipEventRecord(e1[i], stream));
  }
  CHECK(hipStreamSynchronize(stream));
  printf("(a) hipEventRecord pair around each launch : %.2f us\n", mean(false));

  for (int i = 0; i < n; ++i)
    hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, stream, e0[i], e1[i], 0, ticks,
                          (unsigned long long*)nullptr);
  CHECK(hipStreamSynchronize(stream));
  printf("(b) hipExtLaunchKernelGGL start + stop     : %.2f us\n", mean(false));

  for (int i = 0; i < n; ++i)
    hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, stream, nullptr, e1[i], 0, ticks,
                          (unsigned long long*)nullptr);
  CHECK(hipStreamSynchronize(stream));
  printf("(c) hipExtLaunchKernelGGL stop, (stop,stop): %.2f us\n", mean(true));

  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL(spin_kernel, grid, block, 0, stream, ticks, (unsigned long long*)nullptr);
  CHECK(hipStreamSynchronize(stream));
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("(d) wall time of %d back-to-back launches / n: %.2f us (spin %.2f us)\n", n,
         wall / n * 1e6, us);
  return 0;
}
