// Developer microbenchmark: issue cost of the FP64 helper instructions the occupation
// kernel's erf / log / exp use (rounding, conversions, frexp, ldexp, min / max, compares)
// relative to v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP1(name, text)                                                       \
  struct name {                                                               \
    static __device__ __forceinline__ void apply(double& x, int& i) {         \
      asm volatile(text : "+v"(x), "+v"(i));                                  \
    }                                                                         \
    static const char* label() { return #name; }                              \
  };

OP1(fma_f64, "v_fma_f64 %0, %0, %0, %0")
OP1(add_f64, "v_add_f64 %0, %0, %0")
OP1(mul_f64, "v_mul_f64 %0, %0, %0")
OP1(min_f64, "v_min_f64 %0, %0, %0")
OP1(max_f64, "v_max_f64 %0, %0, %0")
OP1(rndne_f64, "v_rndne_f64 %0, %0")
OP1(trunc_f64, "v_trunc_f64 %0, %0")
OP1(cvt_i32_f64, "v_cvt_i32_f64 %1, %0")
OP1(cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
OP1(frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
OP1(frexp_exp_f64, "v_frexp_exp_i32_f64 %1, %0")
OP1(ldexp_f64, "v_ldexp_f64 %0, %0, %1")
OP1(cmp_f64, "v_cmp_lt_f64 vcc, %0, %0")
OP1(cndmask_b32, "v_cndmask_b32 %1, %1, %1, vcc")
OP1(and_b32, "v_and_b32 %1, %1, %1")
OP1(lshl_add_u32, "v_lshl_add_u32 %1, %1, 3, %1")
OP1(bfe_u32, "v_bfe_u32 %1, %1, 12, 8")
OP1(rcp_f64, "v_rcp_f64 %0, %0")
OP1(cndmask_e64_sgpr, "v_cndmask_b32_e64 %1, 0, %1, s[10:11]")
OP1(mul_lo_u32, "v_mul_lo_u32 %1, %1, %1")
OP1(mul_u32_u24, "v_mul_u32_u24 %1, %1, %1")
OP1(mov_b64, "v_mov_b64 %0, %0")
OP1(mov_b32, "v_mov_b32 %1, %1")
OP1(bfi_b32, "v_bfi_b32 %1, %1, %1, %1")
OP1(ashrrev_i32, "v_ashrrev_i32 %1, 8, %1")
OP1(mul_f64_clamp, "v_mul_f64 %0, %0, %0 clamp")
OP1(add_u32, "v_add_u32 %1, %1, %1")
OP1(xor_b32, "v_xor_b32 %1, %1, %1")
OP1(fract_f64, "v_fract_f64 %0, %0")

template <typename Op>
__global__ void bench(double* out, int iters) {
  double x[8];
  int k[8];
  for (int r = 0; r < 8; ++r) { x[r] = 1.0 + r + threadIdx.x * 1e-6; k[r] = r; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 8; ++r) Op::apply(x[r], k[r]);
  }
  double s = 0;
  for (int r = 0; r < 8; ++r) s += x[r] + k[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename Op> void run(double ref_ms, double* ref_out) {
  const int blocks = 256 * 4, iters = 4000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<Op><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<Op><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (ref_out) *ref_out = ms;
  printf("%-18s %.3f ms   x%.2f of v_fma_f64\n", Op::label(), ms, ref_ms > 0 ? ms / ref_ms : 1.0);
  hipFree(out);
}

int main() {
  double ref = 0;
  run<fma_f64>(0, &ref);
  run<add_f64>(ref, nullptr); run<mul_f64>(ref, nullptr); run<min_f64>(ref, nullptr);
  run<max_f64>(ref, nullptr); run<rndne_f64>(ref, nullptr); run<trunc_f64>(ref, nullptr);
  run<fract_f64>(ref, nullptr);
  run<cvt_i32_f64>(ref, nullptr); run<cvt_f64_i32>(ref, nullptr);
  run<frexp_mant_f64>(ref, nullptr); run<frexp_exp_f64>(ref, nullptr);
  run<ldexp_f64>(ref, nullptr); run<cmp_f64>(ref, nullptr); run<cndmask_b32>(ref, nullptr);
  run<and_b32>(ref, nullptr); run<lshl_add_u32>(ref, nullptr); run<bfe_u32>(ref, nullptr);
  run<rcp_f64>(ref, nullptr);
  run<cndmask_e64_sgpr>(ref, nullptr); run<mul_lo_u32>(ref, nullptr);
  run<mul_u32_u24>(ref, nullptr); run<mov_b64>(ref, nullptr); run<mov_b32>(ref, nullptr);
  run<bfi_b32>(ref, nullptr); run<ashrrev_i32>(ref, nullptr); run<mul_f64_clamp>(ref, nullptr);
  run<add_u32>(ref, nullptr); run<xor_b32>(ref, nullptr);
  return 0;
}
