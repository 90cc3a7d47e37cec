// predict_fused_kernel, eight waves x 32 draws (batches below 8192 draws, tables of 105-208 bins).
#include "inst_fused.h"

namespace tc {
namespace host {

int launch_fused_instance_32(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                          int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                          const tc::FusedArgs& fa) {
  const bool assembias = in.assembias, modulate = in.modulate;
#define TC_FUSED32(AB, MO) launch_fused<10, AB, MO, false, 8, 32>(TC_FUSED_ARGS)
#define TC_FUSED_GROUPED(AB, MO) launch_fused<10, AB, MO, false, 8, 32, true>(TC_FUSED_ARGS)
  if (in.grouped)
    return assembias ? (modulate ? TC_FUSED_GROUPED(true, true) : TC_FUSED_GROUPED(true, false))
                     : (modulate ? TC_FUSED_GROUPED(false, true) : TC_FUSED_GROUPED(false, false));
  if (in.leauthaud)
    return modulate ? launch_fused<0, false, true, true, 8, 32>(TC_FUSED_ARGS)
                    : launch_fused<0, false, false, true, 8, 32>(TC_FUSED_ARGS);
  return assembias ? (modulate ? TC_FUSED32(true, true) : TC_FUSED32(true, false))
                   : (modulate ? TC_FUSED32(false, true) : TC_FUSED32(false, false));
#undef TC_FUSED32
#undef TC_FUSED_GROUPED
}

}  // namespace host
}  // namespace tc
