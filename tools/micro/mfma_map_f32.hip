// Developer probe: lane <-> matrix element mapping of v_mfma_f32_16x16x4_f32 on gfx950.
// Checks the hypothesis A[m = l % 16][k = l / 16], B[k = l / 16][n = l % 16],
// D[m = 4 (l / 16) + v][n = l % 16] in register v, by multiplying index-coded matrices.
//
// hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_map_f32.hip -o tools/micro/mfma_map_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // A[m][k] = 1 + m + 16 k (exact small integers), B[k][n] = (k == kk) for a chosen kk -> D = A[:, kk]
  for (int kk = 0; kk < 4; ++kk) {
    const float a = 1.0f + (l % 16) + 16.0f * (l / 16);
    const float b = (l / 16) == kk ? 1.0f + (l % 16) : 0.0f;     // B[kk][n] = 1 + n
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[(kk * 64 + l) * 4 + v] = d[v];
  }
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 4 * 64 * 4 * sizeof(float));
  probe<<<1, 64>>>(d_out);
  std::vector<float> h(4 * 64 * 4);
  hipMemcpy(h.data(), d_out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int kk = 0; kk < 4; ++kk)
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 4; ++v) {
        const int m = 4 * (l / 16) + v, n = l % 16;
        const float expect = (1.0f + m + 16.0f * kk) * (1.0f + n);     // A[m][kk] B[kk][n]
        if (h[(kk * 64 + l) * 4 + v] != expect) {
          if (bad < 8) printf("kk %d lane %d v %d: got %g expected %g\n", kk, l, v, h[(kk * 64 + l) * 4 + v], expect);
          ++bad;
        }
      }
  printf(bad ? "%d mismatches: the hypothesis is wrong\n" : "mapping confirmed: A[m=l%%16][k=l/16], B[k=l/16][n=l%%16], D[m=4(l/16)+v][n=l%%16]\n", bad);
  return bad != 0;
}
