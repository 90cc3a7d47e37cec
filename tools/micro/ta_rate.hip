// Developer microbenchmark: vector-memory (TA) cost per load instruction for the access
// patterns of the matrix-core contraction: per-lane 8 or 16 bytes, with the addresses
// replicated across lane groups, streaming through an L2-resident table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double2v __attribute__((ext_vector_type(2)));

// KIND 0: dwordx2, 64 distinct lanes (512 B unique per instruction)
// KIND 1: dwordx2, 16 distinct values replicated over 4 lane groups (128 B unique)
// KIND 2: dwordx4, 16 distinct 16-byte values replicated (256 B unique)
// KIND 3: dwordx2, 4 distinct values replicated 16x (32 B unique)
// KIND 4: dwordx4, 64 distinct lanes (1024 B unique)
template <int KIND>
__global__ void bench(const double* table, size_t table_doubles, double* out, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  size_t base = (wave * 4099) % (table_doubles - 64 * 1024);
  double s = 0;
  for (int it = 0; it < iters; ++it) {
    const double* p = table + base + (size_t)(it & 255) * 160;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      if (KIND == 0) s += p[u * 64 + lane];
      if (KIND == 1) s += p[u * 16 + (lane >> 4) * 4 + (lane & 3)];
      if (KIND == 2) { double2v v = *(const double2v*)(p + u * 32 + ((lane >> 4) * 4 + (lane & 3)) * 2); s += v.x + v.y; }
      if (KIND == 3) s += p[u * 4 + (lane >> 4)];
      if (KIND == 4) { double2v v = *(const double2v*)(p + u * 128 + lane * 2); s += v.x + v.y; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, const double* table, size_t n) {
  const int blocks = 256 * 5, iters = 2000;   // 5 blocks of 4 waves per CU
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<KIND><<<blocks, 256>>>(table, n, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<KIND><<<blocks, 256>>>(table, n, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 5.0 * iters * 20;     // 20 waves per CU
  printf("%-46s %.3f ms  %.1f ns per load instruction per CU (%.1f cycles at 2.1 GHz)\n", name, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
  hipFree(out);
}

int main() {
  const size_t n = 1 << 20;   // 8 MB of doubles: L2 / MALL resident
  double* table; hipMalloc(&table, n * 8);
  hipMemset(table, 0, n * 8);
  run<0>("dwordx2, 64 distinct (512 B)", table, n);
  run<1>("dwordx2, 16 distinct x4 (128 B)", table, n);
  run<2>("dwordx4, 16 distinct x4 (256 B)", table, n);
  run<3>("dwordx2, 4 distinct x16 (32 B)", table, n);
  run<4>("dwordx4, 64 distinct (1024 B)", table, n);
  return 0;
}
