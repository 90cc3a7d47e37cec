// Developer microbenchmark: what, next to 20 independent v_mfma_f64_4x4x4 per step, costs
// matrix-pipe time: FP64 vector multiplies, LDS gathers, vector loads.
#include <hip/hip_runtime.h>
#include <cstdio>

// AHEAD: the multiplies of step t + 1 are issued before the MFMAs of step t (two weight sets)
template <int MULS, int LDS, int LOADS, int AHEAD = 0>
__global__ __launch_bounds__(256) void bench(const double* table, double* out, int iters) {
  __shared__ double rows[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) rows[i] = 1.0 + i * 1e-9;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double acc[20];
  for (int r = 0; r < 20; ++r) acc[r] = 0.0;
  double a[5], w[4] = {1.0, 1.0, 1.0, 1.0};
  for (int u = 0; u < 5; ++u) a[u] = 1.0 + u;
  const double* p = table + (size_t)(blockIdx.x % 64) * 4096 + lane;
  double ni[4] = {1, 1, 1, 1}, nj[4] = {1, 1, 1, 1};
  double w2[4] = {1.0, 1.0, 1.0, 1.0};
  if (AHEAD == 2) {
    // gathers two steps ahead: two density register sets
    double ni2[4] = {1, 1, 1, 1}, nj2[4] = {1, 1, 1, 1};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        double (&ci)[4] = half ? ni2 : ni;
        double (&cj)[4] = half ? nj2 : nj;
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(w[s]) : "v"(ci[s]), "v"(cj[s]));
        const double* rj = rows + (((it + half) * 7 + (lane >> 4)) & 63) * 64 + (lane & 15);
        const double* ri = rows + (((it + half) * 3) & 63) * 64 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 4; ++s) { cj[s] = rj[16 * s]; ci[s] = ri[16 * s]; }
#pragma unroll
        for (int u = 0; u < LOADS; ++u) a[u] = p[(size_t)(((it + half) * 5 + u) & 63) * 64];
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[u * 4 + s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], w[s], acc[u * 4 + s], 0, 0, 0);
      }
    }
    w2[0] += ni2[0] + nj2[0];
  } else if (AHEAD) {
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        double (&wn)[4] = half ? w : w2;       // weights of the next step
        double (&wc)[4] = half ? w2 : w;       // weights of this step
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(wn[s]) : "v"(ni[s]), "v"(nj[s]));
        const double* rj = rows + (((it + half) * 7 + (lane >> 4)) & 63) * 64 + (lane & 15);
        const double* ri = rows + (((it + half) * 3) & 63) * 64 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 4; ++s) { nj[s] = rj[16 * s]; ni[s] = ri[16 * s]; }
#pragma unroll
        for (int u = 0; u < LOADS; ++u) a[u] = p[(size_t)(((it + half) * 5 + u) & 63) * 64];
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[u * 4 + s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], wc[s], acc[u * 4 + s], 0, 0, 0);
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    if (MULS) {
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(w[s]) : "v"(ni[s]), "v"(nj[s]));
    }
    if (LDS) {
      const double* rj = rows + ((it * 7 + (lane >> 4)) & 63) * 64 + (lane & 15);
      const double* ri = rows + ((it * 3) & 63) * 64 + (lane & 15);
#pragma unroll
      for (int s = 0; s < 4; ++s) nj[s] = rj[16 * s];
      if (LDS == 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ni[s] = ri[16 * s];
      }
    }
    if (LOADS) {
#pragma unroll
      for (int u = 0; u < LOADS; ++u) a[u] = p[(size_t)((it * 5 + u) & 63) * 64];
    }
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[u * 4 + s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], w[s], acc[u * 4 + s], 0, 0, 0);
  }
  double s = 0;
  for (int r = 0; r < 20; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + w[0] + ni[0] + w2[0];
}

template <int MULS, int LDS, int LOADS, int AHEAD = 0> void run(const char* name, const double* table, int per_simd = 5) {
  const int blocks = 256 * per_simd, iters = 4000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) bench<MULS, LDS, LOADS, AHEAD><<<blocks, 256>>>(table, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) bench<MULS, LDS, LOADS, AHEAD><<<blocks, 256>>>(table, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * blocks * 4 * (double)iters * 20 * 512;
  printf("%-44s %.2f ms  %.1f TFLOP/s of MFMA\n", name, ms, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  double* table; hipMalloc(&table, (size_t)64 * 4096 * 8 + 1024); hipMemset(table, 0, (size_t)64 * 4096 * 8);
  run<0, 0, 0>("20 MFMA", table);
  run<1, 0, 0>("20 MFMA + 4 v_mul_f64", table);
  run<1, 1, 0>("20 MFMA + 4 mul + 8 LDS reads", table);
  run<1, 1, 3>("20 MFMA + 4 mul + 8 LDS reads + 3 loads", table);
  run<1, 2, 0>("20 MFMA + 4 mul + 4 LDS reads (n_i kept)", table);
  run<1, 2, 3>("20 MFMA + 4 mul + 4 LDS reads + 3 loads", table);
  run<0, 0, 3>("20 MFMA + 3 loads", table);
  run<0, 0, 5>("20 MFMA + 5 loads", table);
  run<1, 1, 0, 1>("mul ahead: 20 MFMA + 4 mul + 8 LDS", table);
  run<1, 1, 3, 1>("mul ahead: 20 MFMA + 4 mul + 8 LDS + 3 loads", table);
  run<1, 1, 3, 1>("mul ahead, 4 waves/SIMD: full mix", table, 4);
  run<1, 1, 3, 0>("4 waves/SIMD: full mix", table, 4);
  run<1, 1, 0, 2>("gathers 2 steps ahead: 20 MFMA + 4 mul + 8 LDS", table);
  run<1, 1, 3, 2>("gathers 2 steps ahead: full mix", table);
  run<1, 1, 3, 2>("gathers 2 steps ahead: full mix, 4 waves", table, 4);
  run<1, 1, 2, 1>("mul ahead: 20 MFMA + 4 mul + 8 LDS + 2 loads", table);
  run<1, 1, 1, 1>("mul ahead: 20 MFMA + 4 mul + 8 LDS + 1 load", table);
  return 0;
}
