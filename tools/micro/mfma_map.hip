// Developer probe: lane <-> matrix element mapping of v_mfma_f64_4x4x4_4b_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out) {   // out[la][lb] = lane of D that becomes nonzero, or -1
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long mask = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = mask ? __builtin_ctzll(mask) + 64 * (__builtin_popcountll(mask) - 1) : -1;
    }
  }
}

int main() {
  int* d_out; hipMalloc(&d_out, 64 * 64 * 4);
  probe<<<1, 64>>>(d_out);
  std::vector<int> h(64 * 64);
  hipMemcpy(h.data(), d_out, 64 * 64 * 4, hipMemcpyDeviceToHost);
  // for every A lane: which B lanes pair with it, and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
