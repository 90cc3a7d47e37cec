// Developer microbenchmark: SUSTAINED FP64 rates (hundreds of ms, power-limited regime)
// of the instruction kinds a contraction kernel could be built from.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void bench(double* out, int iters) {
  double4v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = double4v{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-6, b = 2.0 - threadIdx.x * 1e-6;
  double vacc[20];
  for (int i = 0; i < 20; ++i) vacc[i] = i;
  float facc[64];
  for (int i = 0; i < 64; ++i) facc[i] = 0.0f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (KIND == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i].x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i].x, 0, 0, 0);
    }
    if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 20; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(vacc[i]) : "v"(a), "v"(b));
    }
    if (KIND == 4) {
      typedef float float4v __attribute__((ext_vector_type(4)));
      float4v* f = (float4v*)facc;      // 8 independent accumulators of 4 floats
#pragma unroll
      for (int i = 0; i < 8; ++i)
        f[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, f[i], 0, 0, 0);
    }
    if (KIND == 5) {
      typedef float float16v __attribute__((ext_vector_type(16)));
      float16v* f = (float16v*)facc;    // 4 independent accumulators of 16 floats
#pragma unroll
      for (int i = 0; i < 4; ++i)
        f[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a, (float)b, f[i], 0, 0, 0);
    }
    if (KIND == 3) {
#pragma unroll
      for (int i = 0; i < 20; ++i)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(vacc[i]) : "v"(a), "v"(b));
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 20; ++i) s += vacc[i];
  for (int i = 0; i < 64; ++i) s += facc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, double flops_per_iter_per_wave, int iters) {
  const int blocks = 256 * 4;   // 4 waves per SIMD
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // ~150 ms warm-up, then ~150 ms timed
  float one = 0;
  hipEventRecord(e0); bench<KIND><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&one, e0, e1);
  const int reps = (int)(150.0f / one) + 1;
  for (int r = 0; r < reps; ++r) bench<KIND><<<blocks, 256>>>(out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) bench<KIND><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-26s first launch %.2f ms -> %.1f TFLOP/s; sustained (%d launches, %.0f ms) %.1f TFLOP/s\n", name, one,
         flops_per_iter_per_wave * iters * blocks * 4 / (one * 1e-3) / 1e12, reps, ms,
         flops_per_iter_per_wave * iters * blocks * 4 * reps / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<0>("mfma_f64_16x16x4", 4 * 2048.0, 5000);
  run<1>("mfma_f64_4x4x4 (4 blocks)", 16 * 512.0, 5000);
  run<2>("v_fmac_f64", 20 * 128.0, 16000);
  run<3>("v_fmac_f64_dpp newbcast", 20 * 128.0, 16000);
  run<0>("mfma_f64_16x16x4 (again)", 4 * 2048.0, 5000);
  // float32 matrix instructions (MI355X: 157.3 TFLOP/s dense): 156.0 / 155.3 sustained
  run<4>("mfma_f32_16x16x4", 8 * 2048.0, 5000);
  run<5>("mfma_f32_32x32x2", 4 * 4096.0, 5000);
}
