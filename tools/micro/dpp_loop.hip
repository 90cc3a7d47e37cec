// Developer microbenchmark: does the FP64 DPP-FMA rate survive a long unrolled body with
// varying broadcast lanes, a v_mul and an LDS read per 20 FMAs (the contraction's shape)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int N> __device__ __forceinline__ void fmac_bcast(double& acc, double t, double w) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(w), "n"(N));
}
template <int E, int... Rs>
__device__ __forceinline__ void entry(double (&acc)[20], const double (&t)[5], double w, std::integer_sequence<int, Rs...>) {
  (fmac_bcast<((E * 20 + Rs) & 15)>(acc[Rs], t[(E * 20 + Rs) >> 4], w), ...);
}
template <int KIND>
__global__ __launch_bounds__(256) void bench(double* out, int iters, unsigned long long* clk) {
  __shared__ double lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
  __syncthreads();
  double acc[20];
  for (int r = 0; r < 20; ++r) acc[r] = r;
  double t[5];
  for (int g = 0; g < 5; ++g) t[g] = 1.0 + (threadIdx.x & 15) + g;
  const int lane = threadIdx.x & 63;
  double ni = 1.0 + lane * 1e-9;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < (KIND >= 2 ? 4 : 1); ++u) {
      double w0, w1, w2, w3;
      if (KIND == 0) { w0 = w1 = w2 = w3 = ni; }
      else {
        w0 = ni * lds[((it + 0) & 63) * 64 + lane];
        w1 = ni * lds[((it + 1) & 63) * 64 + lane];
        w2 = ni * lds[((it + 2) & 63) * 64 + lane];
        w3 = ni * lds[((it + 3) & 63) * 64 + lane];
      }
      entry<0>(acc, t, w0, std::make_integer_sequence<int, 20>());
      entry<1>(acc, t, w1, std::make_integer_sequence<int, 20>());
      entry<2>(acc, t, w2, std::make_integer_sequence<int, 20>());
      entry<3>(acc, t, w3, std::make_integer_sequence<int, 20>());
    }
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = r1 - r0; clk[2 * blockIdx.x + 1] = c1 - c0; }
  double s = 0;
  for (int r = 0; r < 20; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd, iters = KIND >= 2 ? 1000 : 4000;
  double* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned long long* clk; (void)hipMalloc(&clk, (size_t)blocks * 16);
  bench<KIND><<<blocks, 256>>>(out, 10, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); bench<KIND><<<blocks, 256>>>(out, iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("   [block 0: %.1f us, shader clock %.3f GHz] ", h[0] / 100.0, h[1] / (h[0] * 10.0));
  double fmas = 80.0 * iters * (KIND >= 2 ? 4 : 1);
  printf("%-34s waves/SIMD=%d %.3f ms  %.2f nominal cycles/FMA/SIMD  %.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / (fmas * waves_per_simd), fmas * blocks * 4 * 128 / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  for (int w : {4, 5, 6}) {
    run<0>("80 dpp fma, varying lanes", w);
    run<1>("80 dpp fma + 4 (lds read + mul)", w);
    run<2>("320 dpp fma + 16 (lds read + mul)", w);
  }
}
