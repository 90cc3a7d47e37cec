// launch_fused<...>: one instance of predict_fused_kernel per number of r sub-tiles (shared by
// the inst_fused*.hip units, each of which instantiates its part of the family).
#pragma once
#include "internal.h"
#include "kernels.hip.h"

namespace tc {
namespace host {
namespace {
template <int NG, bool AB, bool MO, bool LE = false, int W = tc::kFusedWaves, int DL = 64,
          bool GR = false, int SD = 0>
int launch_fused(int device, int n_u, dim3 grid, dim3 block, int lds, hipStream_t stream,
                 hipEvent_t k0, hipEvent_t k1, const tc::FusedArgs& fa) {
  switch (n_u) {
#define TC_CASE(N)                                                                            \
  case N: {                                                                                   \
    /* (the attribute belongs to the function ON a device: once per device) */                \
    static bool limit_set[64] = {};                                                           \
    if (lds > 64 * 1024 && !(device >= 0 && device < 64 && limit_set[device])) {              \
      TC_HIP(hipFuncSetAttribute(                                                             \
          (const void*)tc::predict_fused_kernel<NG, N, AB, MO, LE, W, DL, GR, SD>,                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));    \
      if (device >= 0 && device < 64) limit_set[device] = true;                               \
    }                                                                                         \
    hipExtLaunchKernelGGL((tc::predict_fused_kernel<NG, N, AB, MO, LE, W, DL, GR, SD>), grid,     \
                          block, lds, stream, k0, k1, 0, fa);                                 \
    break;                                                                                    \
  }
    TC_CASE(1) TC_CASE(2) TC_CASE(3) TC_CASE(4) TC_CASE(5)
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no fused kernel for %d r sub-tiles", n_u);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}
}  // namespace

// (the parts of the family: 8 waves x 64 draws in inst_fused.hip, 8 x 32 in inst_fused32.hip,
// 16 x 64 in inst_fused16.hip, 8 x 40 -- the latency form -- in inst_fused40.hip)
int launch_fused_instance_32(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                             int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                             const tc::FusedArgs& fa);
int launch_fused_instance_40(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                             int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                             const tc::FusedArgs& fa);
int launch_fused_instance_16(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                             int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                             const tc::FusedArgs& fa);

#define TC_FUSED_ARGS device, n_u, grid, block, lds, stream, k0, k1, fa
}  // namespace host
}  // namespace tc
