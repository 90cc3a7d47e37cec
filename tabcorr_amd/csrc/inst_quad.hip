// Kernels of the three-kernel path (occupations, contraction, finalisation), the segment
// kernels, the interpolator's coefficients and the likelihood: their launches, in a translation
// unit of their own.
#define TC_UNIT_QUAD
#include "internal.h"
#include "kernels.hip.h"

namespace tc {
namespace host {

// The occupation kernel of the three-kernel path for these flags (instances: Zheng07 family
// with ten or any number of nodes, per bin or per group of bins; Leauthaud11).
int launch_occupation(const tc::OccArgs& oa, unsigned flags, int n_gauss, bool grouped,
                      int64_t grid_blocks, hipStream_t stream) {
  const dim3 grid((unsigned)grid_blocks), block(tc::kOccWaves * 64);
  const bool assembias = (flags & TC_FLAG_ASSEMBIAS) != 0;
  const bool modulate = (flags & TC_FLAG_MODULATE_WITH_CENOCC) != 0;
#define TC_OCC(NG, AB, MO)                                                           \
  hipLaunchKernelGGL((tc::occ_zheng07_kernel<NG, AB, MO>), grid, block, 0, stream, oa)
  if (flags & TC_FLAG_LEAUTHAUD11) {
    if (modulate)
      hipLaunchKernelGGL(tc::occ_leauthaud11_kernel<true>, grid, block, 0, stream, oa);
    else
      hipLaunchKernelGGL(tc::occ_leauthaud11_kernel<false>, grid, block, 0, stream, oa);
  } else if (grouped) {
#define TC_OCC_GROUPED(AB, MO)                                                        \
  hipLaunchKernelGGL((tc::occ_zheng07_kernel<10, AB, MO, true>), grid, block, 0, stream, oa)
    if (!assembias && !modulate) TC_OCC_GROUPED(false, false);
    else if (!assembias) TC_OCC_GROUPED(false, true);
    else if (!modulate) TC_OCC_GROUPED(true, false);
    else TC_OCC_GROUPED(true, true);
#undef TC_OCC_GROUPED
  } else if (n_gauss == 10) {
    if (!assembias && !modulate) TC_OCC(10, false, false);
    else if (!assembias) TC_OCC(10, false, true);
    else if (!modulate) TC_OCC(10, true, false);
    else TC_OCC(10, true, true);
  } else {
    if (!assembias && !modulate) TC_OCC(0, false, false);
    else if (!assembias) TC_OCC(0, false, true);
    else if (!modulate) TC_OCC(0, true, false);
    else TC_OCC(0, true, true);
  }
#undef TC_OCC
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_contract_quad(int n_u, bool interp, const tc::QuadArgs& args, int lds_bytes,
                         hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  const dim3 grid((unsigned)((args.n_waves + tc::kQuadWavesPerBlock - 1) /
                             tc::kQuadWavesPerBlock));
  const dim3 block(64 * tc::kQuadWavesPerBlock);
  if (args.n_waves == 0) return TC_OK;
  switch (n_u) {
#define TC_CASE(N)                                                                          \
  case N:                                                                                   \
    if (interp)                                                                             \
      hipExtLaunchKernelGGL((tc::contract_quad_kernel<N, true>), grid, block, lds_bytes,  \
                            stream, start, stop, 0, args);                                \
    else                                                                                    \
      hipExtLaunchKernelGGL((tc::contract_quad_kernel<N, false>), grid, block, lds_bytes, \
                            stream, start, stop, 0, args);                                \
    break;
    TC_CASE(1) TC_CASE(2) TC_CASE(3) TC_CASE(4) TC_CASE(5)
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no kernel for %d r sub-tiles", n_u);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_contract_quad_f32_interp(int n_u, const tc::QuadArgs& args, int lds_bytes,
                                    hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  const dim3 grid((unsigned)((args.n_waves + tc::kQuadWavesPerBlock - 1) /
                             tc::kQuadWavesPerBlock));
  const dim3 block(64 * tc::kQuadWavesPerBlock);
  if (args.n_waves == 0) return TC_OK;
  switch (n_u) {
#define TC_CASE(N)                                                                          \
  case N:                                                                                   \
    hipExtLaunchKernelGGL((tc::contract_quad_f32_kernel<N, true>), grid, block, lds_bytes, \
                          stream, start, stop, 0, args);                                  \
    break;
    TC_CASE(1) TC_CASE(2) TC_CASE(3) TC_CASE(4)
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no float32 kernel for %d r sub-tiles", n_u);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_contract_quad_f32(int n_u, const tc::QuadArgs& args, int lds_bytes,
                             hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  const dim3 grid((unsigned)((args.n_waves + tc::kQuadWavesPerBlock - 1) /
                             tc::kQuadWavesPerBlock));
  const dim3 block(64 * tc::kQuadWavesPerBlock);
  if (args.n_waves == 0) return TC_OK;
  switch (n_u) {
#define TC_CASE(N)                                                                        \
  case N:                                                                                 \
    hipExtLaunchKernelGGL((tc::contract_quad_f32_kernel<N, false>), grid, block,          \
                          lds_bytes,                                                      \
                          stream, start, stop, 0, args);                                  \
    break;
    TC_CASE(1) TC_CASE(2) TC_CASE(3) TC_CASE(4)
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no float32 kernel for %d r sub-tiles", n_u);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_finalize_quad(const tc::FinalizeQuadArgs& args, const Tuning& tuning,
                         hipStream_t stream, bool f32) {
  // geometry as launch_finalize: one block per 64 draws, small batches split the rows
  const int64_t n_tiles = args.ldb / 64;
  // (fused likelihood: 16 waves share the rows of the quadratic form -- next to a
  // contraction every vector instruction of a wave waits for a matrix instruction)
  const int threads = tuning.finalize_threads > 0 ? tuning.finalize_threads
                      : n_tiles < 128 || args.chi2 != nullptr ? 1024 : 256;
  const int n_rows = args.n_comp * args.n_r;
  // (many rows -- hundreds of r values -- are split over row blocks of at least 16 rows until
  // the grid has ~2048 blocks: one block per draw tile walked 760 rows serially, 2.4 ms)
  const int row_blocks =
      args.chi2 != nullptr
          ? 1   // (the fused likelihood needs every row of a draw in one workgroup)
          : std::min(n_rows, tuning.finalize_row_blocks > 0
                                 ? tuning.finalize_row_blocks
                                 : n_tiles < 128
                                       ? (int)std::max<int64_t>(1, 512 / n_tiles)
                                       : (int)std::max<int64_t>(
                                             1, std::min<int64_t>(n_rows / 16, 2048 / n_tiles)));
  if (f32)
    hipLaunchKernelGGL((tc::finalize_quad_kernel<float, tc::kQuadTileF32>),
                       dim3((unsigned)n_tiles, (unsigned)row_blocks), dim3(threads), 0, stream,
                       args);
  else
    hipLaunchKernelGGL((tc::finalize_quad_kernel<double, tc::kQuadTile>),
                       dim3((unsigned)n_tiles, (unsigned)row_blocks), dim3(threads), 0, stream,
                       args);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

#define TC_RT_CASES                                                           \
  TC_CASE(4) TC_CASE(8) TC_CASE(12) TC_CASE(16) TC_CASE(20) TC_CASE(24)       \
  TC_CASE(28) TC_CASE(32)

int launch_contract_rt(int rt, dim3 grid, dim3 block, int lds, hipStream_t stream,
                       const tc::ContractArgs& args, hipEvent_t start, hipEvent_t stop) {
  switch (rt) {
#define TC_CASE(N)                                                            \
  case N:                                                                     \
    if (args.n_tables > 0)                                                    \
      hipExtLaunchKernelGGL((tc::contract_mfma_kernel<N, true>), grid, block, \
                            lds, stream, start, stop, 0, args);               \
    else                                                                      \
      hipExtLaunchKernelGGL((tc::contract_mfma_kernel<N, false>), grid, block, \
                            lds, stream, start, stop, 0, args);               \
    break;
    TC_RT_CASES
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no kernel for r tile %d", rt);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

// float32 variant (one kernel for every r tile: always 32 wide)
int launch_contract_f32(dim3 grid, dim3 block, int lds, hipStream_t stream,
                        const tc::ContractArgs& args, hipEvent_t start, hipEvent_t stop) {
  if (lds > 64 * 1024) {
    TC_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&tc::contract_f32_kernel<false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TC_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&tc::contract_f32_kernel<true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  if (args.n_tables > 0)
    hipExtLaunchKernelGGL(tc::contract_f32_kernel<true>, grid, block, lds, stream, start, stop,
                          0, args);
  else
    hipExtLaunchKernelGGL(tc::contract_f32_kernel<false>, grid, block, lds, stream, start, stop,
                          0, args);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int set_lds_limit_rt(int rt, int lds) {
  switch (rt) {
#define TC_CASE(N)                                                            \
  case N:                                                                     \
    TC_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&tc::contract_mfma_kernel<N, false>),   \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    TC_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&tc::contract_mfma_kernel<N, true>),    \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    break;
    TC_RT_CASES
#undef TC_CASE
    default:
      break;
  }
  return TC_OK;
}

int launch_finalize(const FinalizeArgs& args, const Tuning& tuning, hipStream_t stream) {
  // one block per draw tile; a wave sums one (component, r) row at a time over the slabs,
  // so small batches (few blocks, latency-bound) get 16 waves per block instead of 4
  // and split the rows over several blocks (up to ~512 blocks in all)
  const int64_t n_tiles = args.ldb / 64;
  const int threads =
      tuning.finalize_threads > 0 ? tuning.finalize_threads : n_tiles < 128 ? 1024 : 256;
  const int n_rows = args.n_comp * args.n_r;
  const int row_blocks = std::min(
      n_rows, tuning.finalize_row_blocks > 0
                  ? tuning.finalize_row_blocks
                  : n_tiles < 128 ? (int)std::max<int64_t>(1, 512 / n_tiles) : 1);
  hipLaunchKernelGGL(tc::finalize_kernel, dim3((unsigned)n_tiles, (unsigned)row_blocks),
                     dim3(threads), 0, stream, args);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_interp_coef(const InterpArgs& args, hipStream_t stream) {
  if (args.n_draws <= 16 && args.n_tables <= tc::kCoefSmallTables)
    hipLaunchKernelGGL(tc::interp_coef_small_kernel, dim3((unsigned)args.n_draws), dim3(64),
                       0, stream, args);
  else
    hipLaunchKernelGGL(tc::interp_coef_kernel, dim3((unsigned)(args.ldb / 64)), dim3(64), 0,
                       stream, args);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_chi2(const double* xi, int64_t n_draws, int n_r, const double* data,
                const double* precision, double* chi2, hipStream_t stream) {
  // draws per workgroup: 8 unless their deviations would not fit the LDS budget
  const size_t row = (size_t)n_r * sizeof(double);
  TC_CHECK(n_r >= 1 && row <= (size_t)tc::kChi2LdsBytes, "chi2: too many r bins");
  const int per_block =
      (int)std::min<size_t>(tc::kChi2DrawsPerBlock, (size_t)tc::kChi2LdsBytes / row);
  const size_t lds =
      per_block * row + (n_r <= tc::kChi2LdsMatrix ? (size_t)n_r * row : (size_t)0);
  hipLaunchKernelGGL(tc::chi2_kernel, dim3((unsigned)((n_draws + per_block - 1) / per_block)),
                     dim3(32 * per_block), lds, stream, xi, n_draws, n_r, data, precision,
                     chi2);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int launch_occ_from_array_kernel(const double* occupation_device, int64_t n_draws, int64_t ldb,
                                 int n_bins, int n_central, const double* n_h,
                                 const int32_t* perm, double* nbuf, double* ngal2, float* nbuf32,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(tc::occ_from_array_kernel, dim3((unsigned)((ldb + 255) / 256)),
                     dim3(256), 0, stream, occupation_device, n_draws, ldb, n_bins, n_central,
                     n_h, perm, nbuf, ngal2, nbuf32);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

}  // namespace host
}  // namespace tc
