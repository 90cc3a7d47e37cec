// Developer microbenchmark: issue rate of FP64 FMA forms on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N> __device__ __forceinline__ void fmac_bcast(double& acc, double t, double w) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(w), "n"(N));
}
__device__ __forceinline__ void fmac_plain(double& acc, double t, double w) {
  asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(t), "v"(w));
}
__device__ __forceinline__ void fmac_sgpr(double& acc, double t, double w) {
  asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "s"(t), "v"(w));
}
__device__ __forceinline__ void fma_f32(float& acc, float t, float w) {
  asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(t), "v"(w));
}
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pk_fma_f32(float2v& acc, float2v t, float2v w) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(t), "v"(w));
}
__device__ __forceinline__ void pk_fma_f32_s(float2v& acc, float2v t, float2v w) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(t), "v"(w));
}

template <int KIND>
__global__ void bench(double* out, int iters, double tval) {
  double acc[20];
  for (int r = 0; r < 20; ++r) acc[r] = r;
  double t = tval + (threadIdx.x & 15);
  double ts = __builtin_amdgcn_readfirstlane((int)tval) * 1.5;
  double w = 1.0 + threadIdx.x * 1e-9;
  float facc[20]; float2v pacc[20];
  for (int r = 0; r < 20; ++r) { facc[r] = r; pacc[r] = float2v{(float)r, (float)r}; }
  float ft = (float)t, fw = (float)w;
  float2v pt = {ft, ft}, pw = {fw, fw};
  float2v pts = {(float)ts, (float)ts};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 20; ++r) {
      if (KIND == 0) fmac_plain(acc[r], t, w);
      if (KIND == 1) fmac_sgpr(acc[r], ts, w);
      if (KIND == 2) fmac_bcast<5>(acc[r], t, w);
      if (KIND == 3) fma_f32(facc[r], ft, fw);
      if (KIND == 4) pk_fma_f32(pacc[r], pt, pw);
      if (KIND == 5) pk_fma_f32_s(pacc[r], pts, pw);
    }
  }
  double s = 0;
  for (int r = 0; r < 20; ++r) s += acc[r] + facc[r] + pacc[r].x + pacc[r].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, int waves_per_simd, double flops_per_inst) {
  int blocks = 256 * waves_per_simd;   // 4 waves per block -> one per SIMD
  int iters = 20000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<KIND><<<blocks, 256>>>(out, 100, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<KIND><<<blocks, 256>>>(out, iters, 1.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_wave = 20.0 * iters;
  double cycles = ms * 1e-3 * 2.4e9;   // at nominal 2.4 GHz
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f cycles/inst/SIMD (at 2.4 GHz)  %.1f TFLOP/s\n", name, waves_per_simd, ms,
         cycles / (insts_per_wave * waves_per_simd), insts_per_wave * blocks * 4 * 64 * flops_per_inst / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fmac_f64 vgpr", w, 2);
    run<1>("v_fmac_f64 sgpr", w, 2);
    run<2>("v_fmac_f64_dpp bcast", w, 2);
    run<3>("v_fmac_f32", w, 2);
    run<4>("v_pk_fma_f32", w, 4);
    run<5>("v_pk_fma_f32 sgpr", w, 4);
  }
  return 0;
}
