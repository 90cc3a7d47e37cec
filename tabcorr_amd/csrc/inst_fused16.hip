// predict_fused_kernel, sixteen waves x 64 draws (one workgroup per CU: forced or measured).
#include "inst_fused.h"

namespace tc {
namespace host {

int launch_fused_instance_16(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                          int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                          const tc::FusedArgs& fa) {
  const bool assembias = in.assembias, modulate = in.modulate;
#define TC_FUSED(NG, AB, MO, LE) launch_fused<NG, AB, MO, LE, 16>(TC_FUSED_ARGS)
#define TC_FUSED_GROUPED(AB, MO) launch_fused<10, AB, MO, false, 16, 64, true>(TC_FUSED_ARGS)
  if (in.grouped)
    return assembias ? (modulate ? TC_FUSED_GROUPED(true, true) : TC_FUSED_GROUPED(true, false))
                     : (modulate ? TC_FUSED_GROUPED(false, true) : TC_FUSED_GROUPED(false, false));
  if (in.leauthaud) return modulate ? TC_FUSED(0, false, true, true) : TC_FUSED(0, false, false, true);
  if (in.n_gauss != 10) return TC_FUSED(0, false, false, false);
  if (!assembias && !modulate) return TC_FUSED(10, false, false, false);
  if (!assembias) return TC_FUSED(10, false, true, false);
  if (!modulate) return TC_FUSED(10, true, false, false);
  return TC_FUSED(10, true, true, false);
#undef TC_FUSED
#undef TC_FUSED_GROUPED
}

}  // namespace host
}  // namespace tc
