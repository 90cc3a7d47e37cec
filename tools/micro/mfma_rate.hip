// Developer microbenchmark: FP64 MFMA issue rate on gfx950, alone and next to FP64 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void bench(double* out, int iters) {
  double4v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = double4v{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-6, b = 2.0 - threadIdx.x * 1e-6;
  double vacc[8];
  for (int i = 0; i < 8; ++i) vacc[i] = i;
  double s1 = 0;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0 || KIND == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i].x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i].x, 0, 0, 0);
    }
    if (KIND == 2 || KIND == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(vacc[i]) : "v"(a), "v"(b));
    }
  }
  double s = s1;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 8; ++i) s += vacc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, int waves_per_simd, double flops_per_iter_per_wave) {
  int blocks = 256 * waves_per_simd, iters = 20000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<KIND><<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
  hipEventRecord(e0); bench<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s waves/SIMD=%d %.3f ms  %.1f nominal cycles/iter/SIMD  %.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd), flops_per_iter_per_wave * iters * blocks * 4 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("4x mfma_f64_16x16x4", w, 4 * 2048.0);
    run<1>("4x mfma_f64_4x4x4(4b)", w, 4 * 512.0);
    run<3>("8x v_fmac_f64", w, 8 * 128.0);
    run<2>("4x mfma16 + 8x v_fmac_f64", w, 4 * 2048.0 + 8 * 128.0);
  }
}
