// Developer microbenchmark: do vector instructions of one wave overlap with the FP64 matrix
// instructions of another wave on the same SIMD?  Workgroups of 8 waves (two per SIMD): waves
// 0..3 issue v_mfma_f64_16x16x4 (or nothing), waves 4..7 issue one kind of vector instruction
// (or nothing).  If both kinds shared nothing, "both" would take max(matrix, vector); if they
// share the pipe, the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

// VKIND: 0 none, 1 v_fma_f64, 2 v_fma_f32, 3 v_add_u32, 4 v_pk_fma_f32
template <bool MFMA, int VKIND>
__global__ __launch_bounds__(512, 2) void bench(double* out, int mfma_iters, int valu_iters) {
  const int wave = threadIdx.x >> 6;
  double s = 0.0;
  if (wave < 4) {
    if (MFMA) {
      double4v acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = double4v{0, 0, 0, 0};
      const double a = 1.0 + threadIdx.x * 1e-6, b = 2.0 - threadIdx.x * 1e-6;
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
  } else if (VKIND != 0) {
    double d[16];
    float f[16];
    unsigned u[16];
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[16];
    for (int i = 0; i < 16; ++i) {
      d[i] = i;
      f[i] = i;
      u[i] = i;
      p[i] = float2v{(float)i, (float)i};
    }
    const double da = 1.0 + threadIdx.x * 1e-9, db = 1e-9;
    const float fa = 1.0f + threadIdx.x * 1e-6f, fb = 1e-6f;
    const float2v pa = {fa, fa}, pb = {fb, fb};
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (VKIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
        if (VKIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
        if (VKIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(threadIdx.x));
        if (VKIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
      }
    }
    for (int i = 0; i < 16; ++i) s += d[i] + f[i] + u[i] + p[i].x + p[i].y;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool MFMA, int VKIND>
float run(double* out, int mfma_iters, int valu_iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256;        // one 8-wave workgroup per CU: two waves per SIMD
  for (int r = 0; r < 20; ++r) bench<MFMA, VKIND><<<blocks, 512>>>(out, mfma_iters, valu_iters);
  hipEventRecord(e0);
  const int reps = 50;
  for (int r = 0; r < reps; ++r) bench<MFMA, VKIND><<<blocks, 512>>>(out, mfma_iters, valu_iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

template <int VKIND>
void row(const char* name, double* out, int mfma_iters, int valu_iters) {
  const float m = run<true, 0>(out, mfma_iters, valu_iters);
  const float v = run<false, VKIND>(out, mfma_iters, valu_iters);
  const float b = run<true, VKIND>(out, mfma_iters, valu_iters);
  printf("%-14s matrix alone %7.1f us, vector alone %7.1f us, both %7.1f us  (sum %7.1f, max %7.1f)\n",
         name, m, v, b, m + v, m > v ? m : v);
}

int main() {
  double* out;
  hipMalloc(&out, (size_t)256 * 512 * 8);
  // 4 matrix instructions x 64 cycles x 1000 = 256 k cycles; vector: 16 x 4 cycles x iters
  const int mfma_iters = 1000;
  row<1>("v_fma_f64", out, mfma_iters, 2000);     // 128 k cycles of vector work
  row<2>("v_fma_f32", out, mfma_iters, 2000);
  row<3>("v_add_u32", out, mfma_iters, 2000);
  row<4>("v_pk_fma_f32", out, mfma_iters, 2000);
  row<1>("v_fma_f64 x2", out, mfma_iters, 4000);  // as much vector as matrix work
  row<3>("v_add_u32 x2", out, mfma_iters, 4000);
  hipFree(out);
  return 0;
}
