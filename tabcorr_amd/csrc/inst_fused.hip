// One-launch kernels of mode auto (predict_fused_kernel), eight waves x 64 draws -- the
// instances launch.hip: run_fused can select, in translation units of their own (the family is
// most of the library's device code; the units compile in parallel).
#include "inst_fused.h"

namespace tc {
namespace host {

int launch_fused_instance(const FusedInstance& in, int device, int n_u, dim3 grid, dim3 block,
                       int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                       const tc::FusedArgs& fa) {
  const bool assembias = in.assembias, modulate = in.modulate;
  if (in.draws == 40) return launch_fused_instance_40(in, TC_FUSED_ARGS);
  if (in.draws == 32) return launch_fused_instance_32(in, TC_FUSED_ARGS);
  if (in.waves == 16) return launch_fused_instance_16(in, TC_FUSED_ARGS);
#define TC_FUSED(NG, AB, MO, LE) launch_fused<NG, AB, MO, LE, 8>(TC_FUSED_ARGS)
#define TC_FUSED_GROUPED(AB, MO) launch_fused<10, AB, MO, false, 8, 64, true>(TC_FUSED_ARGS)
  if (in.grouped)
    return assembias ? (modulate ? TC_FUSED_GROUPED(true, true) : TC_FUSED_GROUPED(true, false))
                     : (modulate ? TC_FUSED_GROUPED(false, true) : TC_FUSED_GROUPED(false, false));
  if (in.leauthaud) return modulate ? TC_FUSED(0, false, true, true) : TC_FUSED(0, false, false, true);
  if (in.n_gauss != 10) return TC_FUSED(0, false, false, false);
  if (!assembias && !modulate && in.defer == 2)
    return launch_fused<10, false, false, false, 8, 64, false, 2>(TC_FUSED_ARGS);
  if (!assembias && !modulate && in.defer == 1)
    return launch_fused<10, false, false, false, 8, 64, false, 1>(TC_FUSED_ARGS);
  if (!assembias && !modulate) return TC_FUSED(10, false, false, false);
  if (!assembias) return TC_FUSED(10, false, true, false);
  if (!modulate) return TC_FUSED(10, true, false, false);
  return TC_FUSED(10, true, true, false);
#undef TC_FUSED
#undef TC_FUSED_GROUPED
}

}  // namespace host
}  // namespace tc
