// One-launch kernels of mode cross (predict_cross_small_kernel, predict_cross_fused_kernel):
// the instances launch.hip: run_cross_fused can select, in a translation unit of their own.
#include "internal.h"
#include "kernels.hip.h"

namespace tc {
namespace host {

namespace {
template <bool AB, bool MO, bool DE = false>
int launch_cross_fused(int device, int rows, dim3 grid, dim3 block, int lds, hipStream_t stream,
                       hipEvent_t k0, hipEvent_t k1, const tc::CrossFusedArgs& ca) {
  switch (rows / tc::kCrossWaves) {      // rows per wave
#define TC_CASE(N, DEFER)                                                                     \
  case N: {                                                                                   \
    static bool limit_set[64] = {};                                                           \
    if (lds > 64 * 1024 && !(device >= 0 && device < 64 && limit_set[device])) {              \
      /* (the kernel holds a few bytes of static LDS besides) */                               \
      TC_HIP(hipFuncSetAttribute(                                                             \
          (const void*)tc::predict_cross_fused_kernel<N, AB, MO, DEFER>,                      \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));                     \
      if (device >= 0 && device < 64) limit_set[device] = true;                               \
    }                                                                                         \
    hipExtLaunchKernelGGL((tc::predict_cross_fused_kernel<N, AB, MO, DEFER>), grid, block,    \
                          lds, stream, k0, k1, 0, ca);                                        \
    break;                                                                                    \
  }
    TC_CASE(4, DE) TC_CASE(8, DE) TC_CASE(16, false)
#undef TC_CASE
    case 2:     // up to 16 rows: the sums in every wave's registers
      // (the instance with the deferred pairs exists in the source and is not shipped: 16 row
      // sums next to the expansions do not fit 128 registers -- 59.2 against 58.9 us per 10^4
      // draws of the AbacusSummit table with the expansions of round 5's first half, 80 with the
      // group records; undecorated batches go through the 32-row chunk form instead:
      // choose_cross_fused)
      hipExtLaunchKernelGGL((tc::predict_cross_small_kernel<AB, MO, false>), grid, block, lds,
                            stream, k0, k1, 0, ca);
      break;
    default:
      return fail(TC_ERR_UNSUPPORTED, "no cross kernel for %d rows", rows);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}
}  // namespace

int launch_cross_instance(bool assembias, bool modulate, bool defer, int device, int rows,
                          dim3 grid, dim3 block, int lds, hipStream_t stream, hipEvent_t k0,
                          hipEvent_t k1, const tc::CrossFusedArgs& ca) {
#define TC_ARGS device, rows, grid, block, lds, stream, k0, k1, ca
  if (defer) return launch_cross_fused<false, false, true>(TC_ARGS);
  return assembias ? (modulate ? launch_cross_fused<true, true>(TC_ARGS)
                               : launch_cross_fused<true, false>(TC_ARGS))
                   : (modulate ? launch_cross_fused<false, true>(TC_ARGS)
                               : launch_cross_fused<false, false>(TC_ARGS));
#undef TC_ARGS
}

}  // namespace host
}  // namespace tc
