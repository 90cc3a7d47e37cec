// predict_fused_kernel, eight waves x 40 draws: the latency form (one workgroup per CU, 250 for
// 10^4 draws, v_mfma_f64_4x4x4) for calls that have the chip to themselves -- undecorated
// Zheng07, ten nodes, total correlation function; node loops in place, the satellites' or both
// galaxy types' expansions from records with deferred pairs.
#include "inst_fused.h"

namespace tc {
namespace host {

int launch_fused_instance_40(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                             int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                             const tc::FusedArgs& fa) {
  if (in.assembias || in.modulate || in.leauthaud || in.grouped || in.n_gauss != 10)
    return fail(TC_ERR_UNSUPPORTED, "internal: no 40-draw instance for these flags");
  if (in.defer == 2) return launch_fused<10, false, false, false, 8, 40, false, 2>(TC_FUSED_ARGS);
  if (in.defer == 1) return launch_fused<10, false, false, false, 8, 40, false, 1>(TC_FUSED_ARGS);
  return launch_fused<10, false, false, false, 8, 40>(TC_FUSED_ARGS);
}

}  // namespace host
}  // namespace tc
