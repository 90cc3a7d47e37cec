// VERDICT r03 item 9, the throughput half of the gate: at the instruction level, how much
// faster than v_mfma_f64_16x16x4_f64 is a contraction of float32-exact tables through the
// integer matrix cores (v_mfma_i32_16x16x64_i8) at float64 accuracy?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ozaki_gate.hip -o tools/micro/ozaki_gate
// Unit of work = one tile of 16 rows x 16 draws x 64 pairs:
//   float64: 16 matrix instructions (K = 4 each);
//   integer: P slice products (tools/micro/ozaki_accuracy.py: 19 with 5 slices per operand
//   reach 4e-11, 21 with 6 slices 2e-12), the products of one diagonal i + j accumulated in one
//   int32 tile, then per diagonal 4 x (v_cvt_f64_i32 + v_fma_f64 with the blocks' scale) per
//   lane; optionally the slicing of the pair weights (per 64 pairs x 16 draws: 16 elements per
//   lane, S slices each), amortised over the 47.5 row tiles of R = 760.
// Registers only: operands never change, so this is an upper bound of what a kernel can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(call)                                                             \
  do {                                                                          \
    hipError_t status_ = (call);                                                \
    if (status_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(status_));           \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ __launch_bounds__(256) void tiles_f64(double* out, int n_tiles) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  f64x4 acc[4] = {};
  for (int t = 0; t < n_tiles; ++t) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k & 3], 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// SLICES per operand, products with i + j < DIAGONALS; WEIGHTS: also cut 16 pair weights per
// lane into slices once per ROW_TILES tiles.
template <int SLICES, int DIAGONALS, bool WEIGHTS>
__global__ __launch_bounds__(256) void tiles_i8(double* out, int n_tiles, double scale_in,
                                                const double* weights) {
  i32x4 a[SLICES], b[SLICES];
#pragma unroll
  for (int s = 0; s < SLICES; ++s) {
    a[s] = i32x4{(int)threadIdx.x + s, 0x01020304, 0x7f807f80, s};
    b[s] = i32x4{0x01010101, (int)threadIdx.x, s, 0x02020202};
  }
  f64x4 total = {};
  double scale = scale_in;
  constexpr int kRowTiles = 47;
  for (int t = 0; t < n_tiles; ++t) {
    if (WEIGHTS && t % kRowTiles == 0) {
      // 16 pair weights of this lane -> SLICES x 16 bytes (four registers per slice): block
      // maximum over the lane's elements (the cross-lane part left out), then S times
      // multiply, round, subtract, convert, pack
      double w[16], top = 0.0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        w[e] = weights[(t / kRowTiles * 16 + e) & 1023] * scale;
        top = fmax(top, fabs(w[e]));
      }
      const double inverse = 1.0 / (top + 1e-300);
#pragma unroll
      for (int e = 0; e < 16; ++e) w[e] *= inverse * 0.5;
#pragma unroll
      for (int s = 0; s < SLICES; ++s) {
        int packed[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          w[e] *= 128.0;
          const double piece = __builtin_rint(w[e]);
          w[e] -= piece;
          packed[e >> 2] |= ((int)piece & 0xff) << (8 * (e & 3));
        }
        b[s] = i32x4{packed[0], packed[1], packed[2], packed[3]};
      }
      scale = top;
    }
#pragma unroll
    for (int d = 0; d < DIAGONALS; ++d) {
      i32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < SLICES; ++i) {
        const int j = d - i;
        if (j >= 0 && j < SLICES) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc, 0, 0, 0);
      }
      const double factor = scale * (1.0 / (double)(1ll << (7 * (d + 2))));
#pragma unroll
      for (int v = 0; v < 4; ++v) total[v] = fma((double)acc[v], factor, total[v]);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}

template <typename Launch>
static double time_ms(Launch launch) {
  hipEvent_t start, stop;
  CHECK(hipEventCreate(&start));
  CHECK(hipEventCreate(&stop));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(start));
  launch();
  CHECK(hipEventRecord(stop));
  CHECK(hipEventSynchronize(stop));
  float ms = 0.0f;
  CHECK(hipEventElapsedTime(&ms, start, stop));
  return ms;
}

int main() {
  const int blocks = 256 * 8, n_tiles = 20000;      // 8 waves per SIMD
  double* out = nullptr;
  double* weights = nullptr;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
  CHECK(hipMalloc(&weights, 1024 * sizeof(double)));
  double host[1024];
  for (int i = 0; i < 1024; ++i) host[i] = 1e-9 * (1 + i % 37);
  CHECK(hipMemcpy(weights, host, sizeof(host), hipMemcpyHostToDevice));
  const double tiles = (double)blocks * 4 * n_tiles;            // tiles in a launch
  const double macs = tiles * 16 * 16 * 64;
  const double f64 = time_ms([&] { hipLaunchKernelGGL(tiles_f64, dim3(blocks), dim3(256), 0, 0, out, n_tiles); });
  printf("float64 matrix instructions: %.2f ms, %.1f TFLOP/s\n", f64, 2 * macs / f64 * 1e-9);
  auto report = [&](const char* name, double ms) {
    printf("%-58s %.2f ms = x%.2f the float64 rate\n", name, ms, f64 / ms);
  };
  report("5 slices, 15 products, 5 conversions",
         time_ms([&] { hipLaunchKernelGGL((tiles_i8<5, 5, false>), dim3(blocks), dim3(256), 0, 0, out, n_tiles, 1.0, weights); }));
  report("5 slices, 19 products, 6 conversions (4e-11)",
         time_ms([&] { hipLaunchKernelGGL((tiles_i8<5, 6, false>), dim3(blocks), dim3(256), 0, 0, out, n_tiles, 1.0, weights); }));
  report("6 slices, 21 products, 6 conversions (2e-12)",
         time_ms([&] { hipLaunchKernelGGL((tiles_i8<6, 6, false>), dim3(blocks), dim3(256), 0, 0, out, n_tiles, 1.0, weights); }));
  report("5 slices, 19 products, 6 conversions + weights cut",
         time_ms([&] { hipLaunchKernelGGL((tiles_i8<5, 6, true>), dim3(blocks), dim3(256), 0, 0, out, n_tiles, 1.0, weights); }));
  report("6 slices, 21 products, 6 conversions + weights cut",
         time_ms([&] { hipLaunchKernelGGL((tiles_i8<6, 6, true>), dim3(blocks), dim3(256), 0, 0, out, n_tiles, 1.0, weights); }));
  return 0;
}
