// Un-batched calls: single_draw_kernel, resident_draw_kernel, resident_ensemble_kernel -- their
// launches, in a translation unit of their own.  Compiled with -ffp-contract=on (build.py):
// single_draw_kernel and resident_draw_kernel are two instances of one body, and with the
// compiler's default (contraction across statements, decided per instance) the same draw came
// out ~1e-14 apart depending on which of the two served it.
#include <mutex>

#define TC_UNIT_SINGLE
#include "internal.h"
#include "kernels.hip.h"

namespace tc {
namespace host {

int launch_single_kernel(int blocks, hipStream_t stream, const tc::SingleArgs& sa) {
  hipLaunchKernelGGL(tc::single_draw_kernel, dim3((unsigned)blocks), dim3(tc::kSingleThreads), 0,
                     stream, sa);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_resident_kernel(int blocks, hipStream_t stream, const tc::SingleArgs& sa) {
  hipLaunchKernelGGL(tc::resident_draw_kernel, dim3((unsigned)blocks), dim3(tc::kSingleThreads),
                     0, stream, sa);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_ensemble_kernel(int device, int grid, int lds_bytes, hipStream_t stream,
                           const tc::EnsembleArgs& ea) {
  {
    // (the attribute belongs to the function ON a device: once per device, and handles may
    // be used from different threads)
    static std::mutex attribute_mutex;
    static bool attribute_set[64] = {};
    std::lock_guard<std::mutex> lock(attribute_mutex);
    if (!(device >= 0 && device < 64 && attribute_set[device])) {
      TC_HIP(hipFuncSetAttribute((const void*)tc::resident_ensemble_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      if (device >= 0 && device < 64) attribute_set[device] = true;
    }
  }
  hipLaunchKernelGGL(tc::resident_ensemble_kernel, dim3((unsigned)grid),
                     dim3(tc::kEnsembleThreads), (size_t)lds_bytes, stream, ea);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

// The same for every table of an interpolator in one launch (interp.cpp fills the per-class
// pointer arrays): host_ws holds (n_tables, 2) densities, then (n_tables, blocks, rt)
// partial sums.
int launch_single_draw_tables(tc_table* t0, const tc::SingleArgs& prepared, int n_tables,
                              int blocks_per_table, hipStream_t stream) {
  tc::SingleArgs sa = prepared;
  sa.n_tables = n_tables;
  sa.blocks_per_table = blocks_per_table;
  sa.stamps = nullptr;
  sa.theta_many = nullptr;
  sa.n_walkers = 0;
  (void)t0;
  return launch_single_kernel(n_tables * blocks_per_table, stream, sa);
}

}  // namespace host
}  // namespace tc
