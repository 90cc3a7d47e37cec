// Does hipExtAnyOrderLaunch let a kernel start while an earlier kernel of the SAME stream is
// still running on gfx950?  K1 spins ~200 us; K2 (one wave) records its start time.  Prints
// K2's start relative to K1's start and end, launched normally and with the flag.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/any_order.hip -o tools/micro/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned long long* stamps, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void mark(unsigned long long* stamps) {
  if (threadIdx.x == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  unsigned long long* stamps;
  CHECK(hipHostMalloc(&stamps, 64, hipHostMallocDefault));
  hipStream_t stream;
  CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  for (int flag = 0; flag < 2; ++flag)
    for (int rep = 0; rep < 3; ++rep) {
      stamps[0] = stamps[1] = stamps[2] = 0;
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, stream, stamps, 20000ull);   // 200 us at 100 MHz
      hipExtLaunchKernelGGL(mark, dim3(1), dim3(64), 0, stream, nullptr, nullptr,
                            flag ? hipExtAnyOrderLaunch : 0, stamps);
      CHECK(hipGetLastError());
      CHECK(hipStreamSynchronize(stream));
      printf("%-22s K2 starts %7.1f us after K1 starts (K1 runs %.1f us)\n",
             flag ? "hipExtAnyOrderLaunch" : "in order",
             (double)((long long)stamps[2] - (long long)stamps[0]) / 100.0,
             (double)(stamps[1] - stamps[0]) / 100.0);
    }
  return 0;
}
