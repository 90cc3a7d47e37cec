// Launch-path latency of one short kernel whose result the host polls in page-locked memory:
// plain hipLaunchKernelGGL against hipGraphLaunch of the same single-kernel graph (the kernel
// reads its input from page-locked memory in both cases, so the graph needs no parameter
// update per call).  Answers: would a pre-built graph shorten the un-batched predict() call?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/graph_launch.hip -o tools/micro/graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const unsigned long long* in, unsigned long long* out) {
  if (threadIdx.x == 0) {
    const unsigned long long v = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(out + blockIdx.x, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main() {
  unsigned long long *in, *out;
  const int blocks = 13;
  CHECK(hipHostMalloc(&in, 64, hipHostMallocDefault));
  CHECK(hipHostMalloc(&out, 64 * sizeof(unsigned long long), hipHostMallocDefault));
  for (int b = 0; b < 64; ++b) out[b] = 0;
  hipStream_t stream;
  CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 0, stream, in, out);
  CHECK(hipStreamEndCapture(stream, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  volatile unsigned long long* flags = out;
  unsigned long long epoch = 0;
  // launch-path floor against the launch geometry (plain launch, polling)
  for (int geometry = 0; geometry < 6; ++geometry) {
    const int nb = geometry < 3 ? 1 : 13, nt = geometry % 3 == 0 ? 64 : geometry % 3 == 1 ? 256 : 1024;
    const int n = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; ++k) {
      *in = ++epoch;
      hipLaunchKernelGGL(probe, dim3(nb), dim3(nt), 0, stream, in, out);
      for (int b = 0; b < nb; ++b) while (flags[b] != epoch) __builtin_ia32_pause();
      if ((k & 255) == 0) CHECK(hipStreamSynchronize(stream));
    }
    CHECK(hipStreamSynchronize(stream));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("%2d workgroups x %4d threads, poll: %6.2f us per call\n", nb, nt, us);
  }
  for (int mode = 0; mode < 4; ++mode) {
    const int n = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; ++k) {
      *in = ++epoch;
      if (mode % 2 == 0) hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 0, stream, in, out);
      else CHECK(hipGraphLaunch(exec, stream));
      if (mode < 2) {
        for (int b = 0; b < blocks; ++b) while (flags[b] != epoch) __builtin_ia32_pause();
        if ((k & 255) == 0) CHECK(hipStreamSynchronize(stream));
      } else {
        CHECK(hipStreamSynchronize(stream));
      }
    }
    CHECK(hipStreamSynchronize(stream));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("%-22s %-22s %6.2f us per call\n", mode % 2 ? "hipGraphLaunch" : "hipLaunchKernelGGL",
           mode < 2 ? "poll host memory" : "hipStreamSynchronize", us);
  }
  return 0;
}
