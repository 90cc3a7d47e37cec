// Developer microbenchmark: the instruction mix of an MFMA-hybrid FP64 contraction step
// (4 entries x 64 draws x 19 r): 4 x v_mfma_f64_16x16x4 (16 r) + 12 DPP FMAs (3 r) +
// 8 v_mul_f64 + 8 ds_read_b64, against the all-VALU mix (80 DPP FMAs + 4 muls + 4 reads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void fmac_bcast(double& acc, double t, double w) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(w), "n"(N));
}
template <int KIND>
__global__ __launch_bounds__(256) void bench(double* out, int iters, unsigned long long* clk) {
  __shared__ double lds[64 * 80];
  for (int i = threadIdx.x; i < 64 * 80; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double4v acc[4];
  for (int m = 0; m < 4; ++m) acc[m] = double4v{0, 0, 0, 0};
  double tail[4] = {0, 0, 0, 0};
  double vacc[20];
  for (int r = 0; r < 20; ++r) vacc[r] = r;
  double ta = 1.0 + lane * 1e-6, tt = 2.0 + (lane & 15) * 1e-6;
  double t5[5];
  for (int g = 0; g < 5; ++g) t5[g] = 1.0 + (lane & 15) + g;
  double ni[4] = {1.0 + lane * 1e-9, 1.1, 1.2, 1.3};
  double niv = 1.5 + lane * 1e-9;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int row = it & 15;
    if (KIND == 0) {
      // MFMA part: per draw group m one LDS read (lane-dependent row) and one mul
      const double* pb = lds + (row + (lane >> 4)) * 64 + (lane & 15);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double w = ni[m] * pb[m * 16];
        acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, w, acc[m], 0, 0, 0);
      }
      // VALU tail: 3 r values x 4 entries, draws in lanes
      const double* pv = lds + row * 64 + lane;
      double w0 = niv * pv[0], w1 = niv * pv[64], w2 = niv * pv[128], w3 = niv * pv[192];
      fmac_bcast<0>(tail[0], tt, w0); fmac_bcast<1>(tail[1], tt, w0); fmac_bcast<2>(tail[2], tt, w0);
      fmac_bcast<4>(tail[0], tt, w1); fmac_bcast<5>(tail[1], tt, w1); fmac_bcast<6>(tail[2], tt, w1);
      fmac_bcast<8>(tail[0], tt, w2); fmac_bcast<9>(tail[1], tt, w2); fmac_bcast<10>(tail[2], tt, w2);
      fmac_bcast<12>(tail[0], tt, w3); fmac_bcast<13>(tail[1], tt, w3); fmac_bcast<14>(tail[2], tt, w3);
    } else {
      const double* pv = lds + row * 64 + lane;
      double w[4] = {niv * pv[0], niv * pv[64], niv * pv[128], niv * pv[192]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int r = 0; r < 20; ++r) {
          const int idx = e * 20 + r;
          // (compile-time lane selection is emulated with a fixed lane: same cost)
          fmac_bcast<5>(vacc[r], t5[idx >> 4], w[e]);
        }
      }
    }
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = r1 - r0; clk[2 * blockIdx.x + 1] = c1 - c0; }
  double s = tail[0] + tail[1] + tail[2] + tail[3];
  for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
  for (int r = 0; r < 20; ++r) s += vacc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd, iters = 4000;
  double* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  unsigned long long* clk; (void)hipMalloc(&clk, (size_t)blocks * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  bench<KIND><<<blocks, 256>>>(out, 10, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); bench<KIND><<<blocks, 256>>>(out, iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // useful flop per iteration per wave: 4 entries x 64 draws x 19 r x 2
  double flop = 4.0 * 64 * 19 * 2 * iters * blocks * 4;
  printf("%-30s waves/SIMD=%d  %.3f ms  %.1f ns per 4-entry step per SIMD  useful %.1f TFLOP/s  (clock %.2f GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / (iters * (double)waves_per_simd), flop / (ms * 1e-3) / 1e12, h[1] / (h[0] * 10.0));
  (void)hipFree(out); (void)hipFree(clk);
}
int main() {
  for (int w : {3, 4, 5, 6}) {
    run<0>("hybrid: 4 mfma16 + 12 dpp fma", w);
    run<1>("valu: 80 dpp fma", w);
  }
}
