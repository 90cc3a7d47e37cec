// Device kernels of libtabcorr_hip.so (gfx950 / CDNA4 only).
//
// Common layout: one LANE per parameter draw.  A wavefront owns 64 consecutive
// draws ("draw tile"); everything that does not depend on the draw -- table
// values, quadrature constants, work descriptors -- is wave-uniform, is read
// through the scalar data cache (s_load_*) and enters the FP64 FMAs as an SGPR
// operand.  Per-draw values live in VGPRs / LDS with the draw index fastest, so
// every vector access is a contiguous 512-byte row.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "hostmath.h"

namespace tc {

// Pointers whose loads must go through the scalar cache.
typedef const __attribute__((address_space(4))) double* sc_f64;
typedef const __attribute__((address_space(4))) float* sc_f32;
typedef const __attribute__((address_space(4))) int32_t* sc_i32;

constexpr int kLanes = 64;

struct OccArgs {
  const double* theta;     // (n_draws, n_theta) row-major
  int n_theta;
  int64_t n_draws;
  int64_t ldb;             // draws rounded up to a multiple of 64
  int n_bins;
  int n_central;
  int n_gauss;
  unsigned flags;
  double split;
  const double* log_m;     // (n_bins, n_gauss) log10 of the node masses
  const double* m;         // (n_bins, n_gauss) node masses
  const double* weight;    // (n_bins, n_gauss) normalised quadrature weights
  const double* n_h;       // (n_bins)
  const double* percentile;  // (n_bins)
  const int32_t* perm;     // library bin -> reference row
  double* nbuf;            // (n_bins, ldb) number density per bin and draw
  double* ngal;            // (2, ldb) centrals / satellites number density
  double* occupation;      // optional (n_draws, n_bins) in reference order
};

constexpr unsigned kFlagSeparate = 1u;
constexpr unsigned kFlagModulate = 2u;
constexpr unsigned kFlagAssembias = 4u;

__device__ inline double heaviside_assembias(double n, double strength,
                                             bool above, double f1, double f2,
                                             bool bounded_above) {
  // Hearin et al. (2016): shift +d above the split, -d f1/f2 below, with
  // |d| limited so that both stay within [0, 1] (centrals) or [0, inf).
  double up = bounded_above ? 1.0 - n : __builtin_huge_val();
  double dmax = strength >= 0.0 ? fmin(up, n * f2 / f1) : fmin(n, up * f2 / f1);
  double d1 = strength * dmax;
  return above ? n + d1 : n - d1 * f1 / f2;
}

// Mean occupation of every bin for every draw: tabcorr/tabcorr.py:537-578 with
// the two halotools callbacks of :556-563 evaluated inline (Zheng et al. 2007
// eqs. 1 and 3).  Block = kOccWaves waves sharing one draw tile; wave w handles
// bins w, w + kOccWaves, ...
constexpr int kOccWaves = 8;

__global__ __launch_bounds__(kOccWaves * kLanes) void occ_zheng07_kernel(
    OccArgs a) {
  __shared__ double red[2][kOccWaves][kLanes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b0 = (int64_t)blockIdx.x * kLanes + lane;
  const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;

  const double* th = a.theta + b * a.n_theta;
  const double log_m_min = th[0];
  const double inv_sigma = 1.0 / th[1];
  const double m0 = exp10(th[2]);
  const double inv_m1 = 1.0 / exp10(th[3]);
  const double m1 = exp10(th[3]);
  const double alpha = th[4];
  const bool assembias = (a.flags & kFlagAssembias) != 0;
  const bool modulate = (a.flags & kFlagModulate) != 0;
  const double a_cen = assembias ? th[5] : 0.0;
  const double a_sat = assembias ? th[6] : 0.0;
  const double f1 = 1.0 - a.split, f2 = a.split;
  (void)inv_m1;

  sc_f64 log_m = (sc_f64)a.log_m;
  sc_f64 mass = (sc_f64)a.m;
  sc_f64 weight = (sc_f64)a.weight;
  sc_f64 n_h = (sc_f64)a.n_h;
  sc_f64 percentile = (sc_f64)a.percentile;
  sc_i32 perm = (sc_i32)a.perm;

  double sum_cen = 0.0, sum_sat = 0.0;
  for (int g = wave; g < a.n_bins; g += kOccWaves) {
    const bool central = g < a.n_central;
    const bool above = percentile[g] > a.split;
    double acc = 0.0;
    for (int k = 0; k < a.n_gauss; ++k) {
      const double lm = log_m[g * a.n_gauss + k];
      double n;
      if (central) {
        n = 0.5 * (1.0 + erf((lm - log_m_min) * inv_sigma));
        if (assembias) n = heaviside_assembias(n, a_cen, above, f1, f2, true);
      } else {
        const double x = (mass[g * a.n_gauss + k] - m0) / m1;
        n = x > 0.0 ? exp(alpha * log(x)) : 0.0;
        if (modulate) n *= 0.5 * (1.0 + erf((lm - log_m_min) * inv_sigma));
        if (assembias) n = heaviside_assembias(n, a_sat, above, f1, f2, false);
      }
      acc = fma(weight[g * a.n_gauss + k], n, acc);
    }
    if (a.occupation != nullptr && b0 < a.n_draws)
      a.occupation[b0 * a.n_bins + perm[g]] = acc;
    const double dens = acc * n_h[g];
    a.nbuf[(int64_t)g * a.ldb + (int64_t)blockIdx.x * kLanes + lane] = dens;
    if (central) sum_cen += dens; else sum_sat += dens;
  }
  red[0][wave][lane] = sum_cen;
  red[1][wave][lane] = sum_sat;
  __syncthreads();
  if (wave < 2) {
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < kOccWaves; ++w) total += red[wave][w][lane];
    a.ngal[(int64_t)wave * a.ldb + (int64_t)blockIdx.x * kLanes + lane] = total;
  }
}

// Occupations supplied by the caller (the ndarray seam, tabcorr.py:616-623):
// nbuf[g'][b] = occupation[b][perm[g']] * n_h[g'] and the two sums.
__global__ __launch_bounds__(256) void occ_from_array_kernel(
    const double* occupation, int64_t n_draws, int64_t ldb, int n_bins,
    int n_central, const double* n_h, const int32_t* perm, double* nbuf,
    double* ngal) {
  const int64_t b0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b0 >= ldb) return;
  const int64_t b = b0 < n_draws ? b0 : n_draws - 1;
  double sum_cen = 0.0, sum_sat = 0.0;
  for (int g = 0; g < n_bins; ++g) {
    const double dens = occupation[b * n_bins + perm[g]] * n_h[g];
    nbuf[(int64_t)g * ldb + b0] = dens;
    if (g < n_central) sum_cen += dens; else sum_sat += dens;
  }
  ngal[b0] = sum_cen;
  ngal[ldb + b0] = sum_sat;
}

struct ContractArgs {
  const double* nbuf;       // (n_bins, ldb)
  int64_t ldb;
  const void* table;        // (n_rtiles, n_entries, RT) re-laid-out matrix
  int64_t n_entries;
  const Segment* segments;
  const Chunk* chunks;
  const Group* groups;
  int n_components_out;     // 1 (total only) or the table's component count
  int r_stride;             // n_rtiles * RT: padded number of r values
  double* partial;          // (n_groups, n_components_out, r_stride, ldb)
};

// Contraction of the re-laid-out table with the pair weights of 64 draws:
// tabcorr.py:641-649 (total) and :652-683 (per component), without the final
// division.  grid = (draw tiles, groups, r tiles); wave w of a block works on
// chunk w of its group.  The per-draw densities of the rows the group touches
// are staged once in LDS; per table entry a wave does one ds_read_b64 (n_j),
// one v_mul_f64 (n_i n_j) and RT v_fma_f64 whose table operand is an SGPR pair
// fetched with s_load_dwordx16.
template <int RT, typename TableT>
__global__ __launch_bounds__(1024) void contract_kernel(ContractArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  typedef const __attribute__((address_space(4))) TableT* sc_table;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * kLanes;

  const Group group = a.groups[blockIdx.y];
  const int n_rows = group.row_hi - group.row_lo;
  for (int idx = threadIdx.x; idx < n_rows * kLanes; idx += blockDim.x) {
    const int row = idx >> 6;
    lds[idx] = a.nbuf[(int64_t)(group.row_lo + row) * a.ldb + col + (idx & 63)];
  }
  __syncthreads();

  TableT acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0;

  int component = 0;
  if (wave < group.n_chunks) {
    const Chunk chunk = a.chunks[group.chunk_begin + wave];
    component = chunk.component;
    sc_table table = (sc_table)a.table + (int64_t)blockIdx.z * a.n_entries * RT;
    for (int s = chunk.seg_begin; s < chunk.seg_end; ++s) {
      const Segment seg = a.segments[s];
      const double ni =
          seg.i >= 0 ? lds[(seg.i - group.row_lo) * kLanes + lane] : 1.0;
      sc_table row = table + (int64_t)seg.e0 * RT;
      const double* nj = lds + (seg.j0 - group.row_lo) * kLanes + lane;
      for (int t = 0; t < seg.len; ++t) {
        const TableT w = (TableT)(ni * nj[t * kLanes]);
#pragma unroll
        for (int r = 0; r < RT; ++r)
          acc[r] = __builtin_fma(row[t * RT + r], w, acc[r]);
      }
    }
  }
  __syncthreads();  // the staged densities are dead; reuse LDS for the sums

  // Deterministic in-block reduction: waves add their accumulators in wave
  // order into red[component][r][lane].
  const int n_comp = a.n_components_out;
  double* red = lds;
  for (int idx = threadIdx.x; idx < n_comp * RT * kLanes; idx += blockDim.x)
    red[idx] = 0.0;
  __syncthreads();
  const int slot = n_comp == 1 ? 0 : component;
  for (int w = 0; w < n_waves; ++w) {
    if (w == wave && wave < group.n_chunks) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
        red[(slot * RT + r) * kLanes + lane] += (double)acc[r];
    }
    __syncthreads();
  }
  double* out = a.partial +
                ((int64_t)blockIdx.y * n_comp * a.r_stride +
                 (int64_t)blockIdx.z * RT) * a.ldb + col;
  for (int idx = threadIdx.x; idx < n_comp * RT * kLanes; idx += blockDim.x) {
    const int c = idx / (RT * kLanes);
    const int r = (idx >> 6) % RT;
    out[((int64_t)c * a.r_stride + r) * a.ldb + (idx & 63)] = red[idx];
  }
}

struct FinalizeArgs {
  const double* partial;   // (n_groups, n_comp, r_stride, ldb)
  const double* ngal_in;   // (2, ldb)
  int n_groups;
  int n_comp;
  int r_stride;
  int n_r;
  int mode;
  int64_t ldb;
  int64_t n_draws;
  double* ngal;            // (n_draws) or (n_draws, 2)
  double* xi;              // (n_draws, n_comp, n_r)
};

// Sum the per-group partials in fixed order, divide by the total pair weight
// (tabcorr.py:646-649, 653-655: sum(ngal_sq) = (sum ngal)^2 in mode auto) and
// write the results in the reference's output order.
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.n_draws) return;
  const double n_cen = a.ngal_in[b], n_sat = a.ngal_in[a.ldb + b];
  const double total = n_cen + n_sat;
  const double norm = a.mode == 0 ? total * total : total;
  if (a.n_comp == 1) {
    a.ngal[b] = total;
  } else {
    a.ngal[2 * b] = n_cen;
    a.ngal[2 * b + 1] = n_sat;
  }
  for (int c = 0; c < a.n_comp; ++c) {
    for (int r = 0; r < a.n_r; ++r) {
      double sum = 0.0;
      for (int g = 0; g < a.n_groups; ++g)
        sum += a.partial[(((int64_t)g * a.n_comp + c) * a.r_stride + r) * a.ldb + b];
      a.xi[((int64_t)b * a.n_comp + c) * a.n_r + r] = sum / norm;
    }
  }
}

}  // namespace tc
