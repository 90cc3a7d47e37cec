// Device kernels of libtabcorr_hip.so (gfx950 / CDNA4 only).
//
// Common layout: a workgroup owns 64 consecutive draws ("draw tile").  In the occupation
// and finalisation kernels one LANE is one draw and everything that does not depend on
// the draw (quadrature constants, work descriptors) is wave-uniform and read through the
// scalar data cache.  The contraction runs on the FP64 matrix cores with the draws as one
// matrix dimension.  Per-draw values live in global memory / LDS with the draw index
// fastest, so every vector access is a contiguous 512-byte row.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

#include "fastmath.h"
#include "hostmath.h"
#include "kernel_args.h"
#include "series.h"

// Developer timelines (per-workgroup / per-wave time stamps behind tc_table_set_option
// "trace") exist only in developer builds (-DTC_DEVELOPER_KNOBS, tools/build_dev.sh); the
// release kernels carry none of their branches.
#ifdef TC_DEVELOPER_KNOBS
#define TC_TRACE(pointer) ((pointer) != nullptr)
#else
#define TC_TRACE(pointer) false
#endif

namespace tc {

static_assert(sizeof(QuadRun) == 32 && sizeof(QuadCompArgs) == 32, "read as 8 x int32");

// Pointers whose loads must go through the scalar cache.
typedef const __attribute__((address_space(4))) double* sc_f64;
typedef const __attribute__((address_space(4))) float* sc_f32;
typedef const __attribute__((address_space(4))) int32_t* sc_i32;
// Pointers that were themselves loaded from memory are generic to the compiler; this
// tells it they point to global memory (global_load with counted vmcnt, not flat_load).
typedef const __attribute__((address_space(1))) double* gl_f64;

// wave priority chosen at run time (experiments: who wins the FP64 pipe)
__device__ inline void set_priority(int priority) {
  switch (priority) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
}

__device__ inline double heaviside_assembias(double n, double strength,
                                             bool above, double f2_over_f1,
                                             double f1_over_f2,
                                             bool bounded_above) {
  // Hearin et al. (2016): shift +d above the split, -d f1/f2 below, with
  // |d| limited so that both stay within [0, 1] (centrals) or [0, inf).
  double up = bounded_above ? 1.0 - n : __builtin_huge_val();
  double dmax = strength >= 0.0 ? fmin(up, n * f2_over_f1) : fmin(n, up * f2_over_f1);
  double d1 = strength * dmax;
  return above ? n + d1 : n - d1 * f1_over_f2;
}

// Per-draw constants of the Zheng07 occupation functions and the flags of draws the fast
// node loop cannot represent.  The table-driven erf / log2 / exp2 clamp their arguments, so
// a NaN parameter would come out as a finite occupation where the reference's NumPy
// callbacks (tabcorr.py:556-563; oracle/tabcorr_oracle.py) return NaN -- and an MCMC
// likelihood relies on that NaN to reject the draw.  Such parameters are replaced by
// harmless finite ones here and the affected bins are set to NaN after their node loop:
//   kBadCen  logMmin or sigma_logM is NaN (or both infinite): every <N_cen> is NaN;
//   kBadSat  logM1 or alpha is NaN: <N_sat> is NaN at every node with M > M0, 0 elsewhere
//            (a NaN logM0 makes "M - M0 > 0" false everywhere: <N_sat> = 0);
//   kTieCen  sigma_logM == 0: a step function, NaN (0 / 0) at a node with log M == logMmin;
//   kInfSat  M1 = 10^logM1 underflows (logM1 < -290 is treated as M1 = 0) with alpha > 0:
//            <N_sat> is infinite at every node with M > M0.
// Assembly-bias strengths are clipped to [-1, 1] as halotools' HeavisideAssembias does (a
// NaN strength stays NaN and propagates through the decoration by itself).
constexpr int kBadCen = 1, kBadSat = 2, kTieCen = 4, kInfSat = 8;

struct DrawSetup {
  double log_m_min, inv_sigma, m0, log2_m1, sat_scale, alpha, a_cen, a_sat;
  int bad;
};

__device__ inline double clip_strength(double a) {
  a = a > 1.0 ? 1.0 : a;
  return a < -1.0 ? -1.0 : a;
}

__device__ inline DrawSetup prepare_draw(const double* table, const fm::Consts& kc,
                                         double log_m_min, double sigma, double log_m0,
                                         double log_m1, double alpha, double a_cen,
                                         double a_sat) {
  DrawSetup d;
  d.bad = 0;
  if (log_m_min != log_m_min || sigma != sigma ||
      (__builtin_isinf(log_m_min) && __builtin_isinf(sigma))) {
    d.bad |= kBadCen;
    log_m_min = 12.0;
    sigma = 1.0;
  }
  if (sigma == 0.0) d.bad |= kTieCen;
  if (log_m1 != log_m1 || alpha != alpha) {
    d.bad |= kBadSat;
    log_m1 = 13.0;
    alpha = 1.0;
  }
  bool no_satellites = log_m0 != log_m0;   // "M - M0 > 0" is false for every node
  if (log_m1 < -290.0) {                   // M1 = 0: (x / 0)^alpha = inf, 1 or 0
    if (alpha > 0.0) d.bad |= kInfSat;
    if (alpha < 0.0) no_satellites = true;
    log_m1 = 13.0;
  }
  d.log_m_min = log_m_min;
  d.inv_sigma = 1.0 / sigma;
  // (logM0 = -inf, or so low that 10^logM0 underflows: M0 = 0, as NumPy's 10**x gives -- the
  // two-part product of exp10_fast would make inf - inf = NaN of it and lose every satellite)
  d.m0 = no_satellites ? 1e300 : log_m0 < -300.0 ? 0.0 : fm::exp10_fast(table, kc, log_m0);
  // ((M - M0) / M1)^alpha = 2^(alpha (log2(M - M0) - log2 M1)); log2 M1 is carried in two
  // parts, the low one applied to the finished bin sum as 2^(-alpha lo)
  const double hi = log_m1 * fm::kLog2Of10Hi;
  const double lo = __builtin_isinf(log_m1)
                        ? 0.0
                        : fma(log_m1, fm::kLog2Of10Hi, -hi) + log_m1 * fm::kLog2Of10Lo;
  d.log2_m1 = hi;
  d.sat_scale = fma(-alpha * fm::kLn2, lo, 1.0);
  d.alpha = alpha;
  d.a_cen = clip_strength(a_cen);
  d.a_sat = clip_strength(a_sat);
  return d;
}

// Mean occupation of bin g for the lane's draw (Zheng et al. 2007 eqs. 1 and 3 averaged over
// the bin's quadrature nodes): the body shared by occ_zheng07_kernel and predict_fused_kernel.
struct DrawParams {
  double log_m_min, inv_sigma, m0, log2_m1, sat_scale, alpha, a_cen, a_sat;
  int bad;
  bool any_bad;    // wave-uniform: does any draw of the tile need the NaN fix-ups?
  // The moment expansions (series.h).  A LANE adds as many terms as its own draw needs --
  // inv_sigma_hi: high dword of the draw's |1 / sigma| (central bins); m0_hi: of its M0, or
  // INT_MAX where the expansion does not apply to the draw (centrals to fix up; satellites: alpha
  // outside [0, 4], any fix-up) -- so that a draw's result does not depend on its neighbours in
  // the wave.  INT_MAX: off.
  int inv_sigma_hi = 0x7fffffff, m0_hi = 0x7fffffff;
};

// Constants of the moment expansion (launch.hip: get_quadrature): per bin (or member, in group
// order) series::kStride doubles and series::kThresholds int32.  consts == nullptr: off.
struct SeriesConsts {
  sc_f64 consts = nullptr;
  sc_i32 thresholds = nullptr;
  sc_f64 sat_consts = nullptr;         // series::sat::kStride doubles per bin / member
  sc_i32 sat_thresholds = nullptr;
  // the longest satellite expansion a lane takes in place: 12 + 4 sat_cap terms (kernels that
  // defer what is left: a wave pays for its longest lane)
  int sat_cap = series::sat::kSteps - 1;
};

// The expansions' per-lane keys (series.h): what a draw's term counts follow (non-negative
// doubles order like their high dwords; NaN and infinity come out above every threshold).
template <bool MODULATE>
__device__ inline void series_setup(DrawParams& dp, bool on, bool sat_on) {
  const bool sat_ok = sat_on && !MODULATE && dp.alpha >= 0.0 && dp.alpha <= 4.0 && dp.bad == 0;
  // (a draw whose centrals are to be fixed up -- kBadCen: its parameters were replaced -- takes
  // the node path, where the fix-ups are)
  dp.inv_sigma_hi = on && !(dp.bad & kBadCen) ? (int)(fm::bits_of(fabs(dp.inv_sigma)) >> 32)
                                              : 0x7fffffff;
  // (fabs: a NaN M0 -- 10^inf from the table-driven exp10 -- may carry a sign bit)
  dp.m0_hi = sat_ok ? (int)(fm::bits_of(fabs(dp.m0)) >> 32) : 0x7fffffff;
}

// (Mc - M0)^alpha / M1^alpha times the binomial sum of a satellite bin's moments.
__device__ __forceinline__ double sat_series_value(const double* table, const fm::Consts& kc,
                                                   sc_f64 consts, sc_i32 thresholds,
                                                   const DrawParams& d) {
  const double base = consts[0] - d.m0;
  const double eps = consts[0] * series::sat::reciprocal(base);
  const double sum = series::sat::binomial_sum(consts, eps, d.alpha, thresholds, d.m0_hi);
  return sum * fm::exp2_fast(table, kc,
                             d.alpha * fm::log2_fast_offset(table, kc, base, d.log2_m1));
}


// The node loop of bin g (tabcorr.py:556-578 with the Zheng07 callbacks inline) and the fix-ups
// of draws it cannot represent: what occ_bin_zheng07 runs where no shortcut or expansion
// applies.  Ptr = scalar-cache pointers with a wave-uniform g (the constants are scalar
// operands), or plain pointers with a bin PER LANE, whose constants then come by vector loads;
// `central`, `above` and d.any_bad are per lane resp. recomputed by the caller there.
template <int NGAUSS, bool ASSEMBIAS, bool MODULATE, typename Ptr>
__device__ __forceinline__ double occ_nodes_zheng07(const double* table, const fm::Consts& kc,
                                                    int g, int n_gauss, bool central, bool above,
                                                    Ptr log_m, Ptr mass, Ptr weight,
                                                    Ptr weight_sum, const DrawParams& d,
                                                    double f1, double f2) {
  constexpr bool assembias = ASSEMBIAS;
  constexpr bool modulate = MODULATE;
  const double log_m_min = d.log_m_min, inv_sigma = d.inv_sigma, m0 = d.m0;
  const double log2_m1 = d.log2_m1, sat_scale = d.sat_scale, alpha = d.alpha;
  const double a_cen = d.a_cen, a_sat = d.a_sat;
  const int bad = d.bad;
  const bool median = assembias && f1 == 1.0 && f2 == 1.0;
  const double s_cen = above ? a_cen : -a_cen, s_sat = above ? a_sat : -a_sat;
  double acc = 0.0;
  if (central && !assembias) {
    // sum_k w_k (1 + erf_k) / 2 = (W + sum_k w_k erf_k) / 2 with W = sum_k w_k from the host
    // (get_quadrature): one instruction per node less than forming every <N_cen> first
#pragma unroll NGAUSS > 0 ? NGAUSS : 1
    for (int k = 0; k < n_gauss; ++k) {
      const double lm = log_m[g * n_gauss + k];
      acc = fma(weight[g * n_gauss + k],
                fm::erf_fast(table, kc, (lm - log_m_min) * inv_sigma), acc);
    }
    acc = fma(0.5, acc, 0.5 * weight_sum[g]);
  } else if (central) {
#pragma unroll NGAUSS > 0 ? NGAUSS : 1
    for (int k = 0; k < n_gauss; ++k) {
      const double lm = log_m[g * n_gauss + k];
      double n = fma(0.5, fm::erf_fast(table, kc, (lm - log_m_min) * inv_sigma), 0.5);
      n = median ? fma(s_cen, fmin(n, 1.0 - n), n)
                 : heaviside_assembias(n, a_cen, above, f2, f1, true);
      acc = fma(weight[g * n_gauss + k], n, acc);
    }
  } else {
#pragma unroll NGAUSS > 0 ? NGAUSS : 1
    for (int k = 0; k < n_gauss; ++k) {
      const double x = mass[g * n_gauss + k] - m0;
      // 1e-300 keeps log2's input a positive normal number on the lanes with
      // M <= M0, whose result the scaling step of exp2 then sets to exactly 0
      double n = fm::exp2_fast(
          table, kc,
          alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, log2_m1),
          x > 0.0);
      // (another split: the decoration is not linear in n, everything per node)
      if (assembias && !median) n *= sat_scale;
      if (modulate) {
        const double lm = log_m[g * n_gauss + k];
        n *= fma(0.5, fm::erf_fast(table, kc, (lm - log_m_min) * inv_sigma), 0.5);
      }
      if (assembias && !median) n = heaviside_assembias(n, a_sat, above, f2, f1, false);
      acc = fma(weight[g * n_gauss + k], n, acc);
    }
    if (!assembias || median) acc *= sat_scale;
    if (median) acc = fma(s_sat, acc, acc);
  }
  if (d.any_bad) {
    bool tie = false;
    if ((bad & kTieCen) && (central || modulate))
      for (int k = 0; k < n_gauss; ++k) tie = tie || log_m[g * n_gauss + k] == log_m_min;
    const bool cen_nan = (bad & kBadCen) || tie;
    if (!central && (bad & kInfSat) && acc != 0.0) {
      // (decorated: the shift limit is inf - inf = NaN in the reference's arithmetic)
      acc = assembias ? __builtin_nan("") : __builtin_huge_val();
    }
    if (central ? cen_nan : (((bad & kBadSat) && acc != 0.0) || (modulate && cen_nan)))
      acc = __builtin_nan("");
  }
  return acc;
}

template <int NGAUSS, bool ASSEMBIAS, bool MODULATE>
__device__ __forceinline__ double occ_bin_zheng07(const double* table, const fm::Consts& kc,
                                                  int g, int n_gauss, bool central, bool above,
                                                  sc_f64 log_m, sc_f64 mass, sc_f64 weight,
                                                  sc_f64 weight_sum, const DrawParams& d,
                                                  double f1, double f2,
                                                  const SeriesConsts& sr = SeriesConsts(),
                                                  bool* defer_sat = nullptr) {
  constexpr bool assembias = ASSEMBIAS;
  constexpr bool modulate = MODULATE;
  const double log_m_min = d.log_m_min, inv_sigma = d.inv_sigma, m0 = d.m0;
  const double sat_scale = d.sat_scale, a_sat = d.a_sat;
  const bool any_bad = d.any_bad;
  // Split at the median (f1 = f2 = 1, what the device path is given: models.device_spec): the
  // limit of the shift is min(n, 1 - n) for either sign of the strength (centrals) and n itself
  // (satellites, unbounded above), so the decoration is n + s min(n, 1 - n) resp. n (1 + s) with
  // s = +-strength above / below the split -- three instructions per central node instead of
  // ten, and for the satellites one factor on the finished bin sum.
  const bool median = assembias && f1 == 1.0 && f2 == 1.0;
  const double s_sat = above ? a_sat : -a_sat;
  double acc = 0.0;
  // Wave-uniform shortcuts (the draws of a sampler's ensemble cluster around the posterior,
  // so whole bins sit on the plateaus for all 64 draws of a tile): a bin whose nodes all
  // have |z| >= 6 evaluates to erf = +-1 exactly (erf_fast clamps there), a satellite bin
  // below every draw's M0 to 0.  Same bits as the node loop, none of its instructions.
  int shortcut = 0;      // 1: all ones, 2: all zeros
#ifndef TC_NO_OCC_SHORTCUTS
  if (!assembias && !any_bad) {
    if (central) {
      const double z_a = (log_m[g * n_gauss] - log_m_min) * inv_sigma;
      const double z_b = (log_m[g * n_gauss + n_gauss - 1] - log_m_min) * inv_sigma;
      const double z_lo = z_a < z_b ? z_a : z_b, z_hi = z_a < z_b ? z_b : z_a;
      if (__builtin_amdgcn_ballot_w64(!(z_lo >= 6.0)) == 0) shortcut = 1;
      else if (__builtin_amdgcn_ballot_w64(!(z_hi <= -6.0)) == 0) shortcut = 2;
    } else {
      const double m_a = mass[g * n_gauss], m_b = mass[g * n_gauss + n_gauss - 1];
      if (__builtin_amdgcn_ballot_w64((m_a > m_b ? m_a : m_b) > m0) == 0) shortcut = 2;
    }
  }
#endif
  // the node sum of an undecorated central bin by its moment expansion (series.h) on the lanes
  // whose own draw allows it -- (half bin width) / sigma small enough; the others run the node
  // loop (both paths in a wave that holds both kinds of draws).  Per lane, so that a draw's
  // bits do not depend on its neighbours.
  bool series_cen = false, series_sat = false;     // this lane takes the expansion
  if (central && !assembias && shortcut == 0 && sr.consts != nullptr && n_gauss >= 4)
    series_cen = series::eligible(sr.thresholds + g * series::kThresholds, d.inv_sigma_hi);
  if (!central && !modulate && (!assembias || median) && shortcut == 0 &&
      sr.sat_consts != nullptr && n_gauss >= 4)
    series_sat = d.m0_hi < (sr.sat_thresholds + g * series::sat::kThresholds)[sr.sat_cap];
  if (shortcut != 0) {
    if (shortcut == 1) acc = weight_sum[g];
  } else if (series_cen) {
    acc = series::central_sum(table, kc, log_m_min, inv_sigma, sr.consts + g * series::kStride,
                              weight_sum[g], sr.thresholds + g * series::kThresholds,
                              d.inv_sigma_hi);
    acc = fma(0.5, acc, 0.5 * weight_sum[g]);
  } else if (series_sat) {
    // a satellite bin well above the draw's M0: the binomial expansion of its node sum
    acc = sat_series_value(table, kc, sr.sat_consts + g * series::sat::kStride,
                           sr.sat_thresholds + g * series::sat::kThresholds, d);
    acc *= sat_scale;
    if (median) acc = fma(s_sat, acc, acc);
  } else if (defer_sat != nullptr && !central) {
    // (predict_fused_kernel's deferred pairs: a satellite bin that no expansion serves for this
    // lane's draw is evaluated later, whole waves of such (bin, draw) pairs at a time -- unless
    // every node lies at or below the draw's M0: 0, as the node loop would give)
    const double m_a = mass[g * n_gauss], m_b = mass[g * n_gauss + n_gauss - 1];
    *defer_sat = !(d.bad == 0 && !((m_a > m_b ? m_a : m_b) > m0));
  } else {
    // (no shortcut, no expansion for this lane's draw: the node loop, and the fix-ups of the
    // draws it cannot represent -- such draws never qualify for an expansion, series_setup)
    acc = occ_nodes_zheng07<NGAUSS, ASSEMBIAS, MODULATE>(table, kc, g, n_gauss, central, above,
                                                        log_m, mass, weight, weight_sum, d, f1,
                                                        f2);
  }
  return acc;
}

// An undecorated central bin of predict_fused_kernel's instance that defers the centrals too,
// from its record (series.h, namespace cen_record): plateau nodes, thresholds, centre, m_0 and
// all 24 moments with one round trip of scalar loads; a lane adds the passes ITS draw asks for
// (the arithmetic of series::central_sum).  *deferred where no expansion serves the draw.
__device__ __forceinline__ double occ_cen_record(const double* table, const fm::Consts& kc,
                                                 sc_f64 rec, const DrawParams& d,
                                                 bool* deferred) {
  namespace record = series::cen_record;
  series::record::f64x8_t head, m0, m1, m2;
  record::load_record(rec, head, m0, m1, m2);
  const double log_m_min = d.log_m_min, inv_sigma = d.inv_sigma;
  const double weight_sum = head.v[record::kSum];
#ifndef TC_NO_OCC_SHORTCUTS
  if (!d.any_bad) {
    const double z_a = (head.v[record::kLow] - log_m_min) * inv_sigma;
    const double z_b = (head.v[record::kHigh] - log_m_min) * inv_sigma;
    const double z_lo = z_a < z_b ? z_a : z_b, z_hi = z_a < z_b ? z_b : z_a;
    if (__builtin_amdgcn_ballot_w64(!(z_lo >= 6.0)) == 0) return weight_sum;
    if (__builtin_amdgcn_ballot_w64(!(z_hi <= -6.0)) == 0) return 0.0;
  }
#endif
  const series::Thresholds limit = series::record::thresholds_of(head);
  double acc = 0.0;
  if (d.inv_sigma_hi < limit.v[series::kSteps - 1]) {
    const int n_blocks = series::passes<series::kSteps>(limit, d.inv_sigma_hi, 2);
    double g0, z0;
    const double e = fm::erf_gauss_fast(
        table, kc, (head.v[record::kCentre] - log_m_min) * inv_sigma, &g0, &z0);
    const double a = 2.0 * z0 * inv_sigma;
    // (b through a value the optimiser cannot see through: the multiples n b of all 24 unrolled
    // terms would otherwise be formed once per draw, outside the loop of bins, and held in
    // registers across it -- 129 of them in all)
    double b = -2.0 * inv_sigma * inv_sigma;
    asm volatile("" : "+v"(b));
    double p_prev = 0.0, p = inv_sigma, nb = -b, sum = 0.0;
    auto term = [&](double moment) {
      sum = fma(p, moment, sum);
      nb += b;
      const double next = fma(a, p, nb * p_prev);
      p_prev = p;
      p = next;
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) term(m0.v[k]);
    if (n_blocks > 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) term(m1.v[k]);
      if (n_blocks > 3) {
#pragma unroll
        for (int k = 4; k < 8; ++k) term(m1.v[k]);
        if (n_blocks > 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) term(m2.v[k]);
          if (n_blocks > 5) {
#pragma unroll
            for (int k = 4; k < 8; ++k) term(m2.v[k]);
          }
        }
      }
    }
    acc = fma(0.5, fma(g0, sum, weight_sum * e), 0.5 * weight_sum);
  } else {
    *deferred = true;
  }
  return acc;
}

// An undecorated satellite bin of predict_fused_kernel's deferring instance from its record
// (series.h, namespace sat_record): head and the first sixteen moments with one round trip of
// scalar loads.  In place only the draws the bin's SHORTEST expansion serves (the same number
// of terms for all of them: the record's); *deferred for the others unless every node lies at
// or below their M0.
__device__ __forceinline__ double occ_sat_record(const double* table, const fm::Consts& kc,
                                                 sc_f64 rec, const DrawParams& d,
                                                 bool* deferred) {
  namespace record = series::sat_record;
  series::record::f64x8_t head, first, second;
  record::load_record(rec, head, first, second);
  const double m0 = d.m0, largest = head.v[record::kLargest];
#ifndef TC_NO_OCC_SHORTCUTS
  if (!d.any_bad && __builtin_amdgcn_ballot_w64(largest > m0) == 0) return 0.0;
#endif
  const unsigned long long word = fm::bits_of(head.v[record::kLimit]);
  const int limit = (int)(unsigned)(word & 0xffffffffull), n_passes = (int)(word >> 32);
  double acc = 0.0;
  if (d.m0_hi < limit) {
    const double centre = head.v[record::kCentre], alpha = d.alpha;
    const double base = centre - m0;
    const double eps = centre * series::sat::reciprocal(base);
    const double power =
        fm::exp2_fast(table, kc, alpha * fm::log2_fast_offset(table, kc, base, d.log2_m1));
    double c = 1.0, g = eps * alpha, sum = head.v[record::kSum];
    auto term = [&](double moment) {
      c *= g;
      g -= eps;
      sum = fma(c, moment, sum);
    };
    // (sixteen terms whatever the bin asks for: more than needed only adds terms below the
    // tolerance)
#pragma unroll
    for (int k = 0; k < 8; ++k) term(first.v[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) term(second.v[k]);
    sc_f64 further = rec + record::kHead + 4 * record::kBlock;
#pragma unroll 1
    for (int pass = 4; pass < n_passes; ++pass) {
      const series::f64x4_t m = series::load_four(further, 0, 0);
      further += record::kBlock;
#pragma unroll
      for (int k = 0; k < 4; ++k) term(m[k]);
    }
    acc = sum * power * d.sat_scale;
  } else {
    *deferred = !(d.bad == 0 && !(largest > m0));
  }
  return acc;
}

// The same for workgroups of 32 draws (predict_fused_kernel<..., DL = 32>): lane = (draw, half),
// each half of the wave takes five of the bin's ten nodes -- their constants come from memory
// by vector loads, two addresses per wave, instead of scalar registers -- and the halves are
// added through a lane exchange; then the bin's affine map / scale and the fix-ups of draws the
// node loop cannot represent, as in occ_bin_zheng07 (same values to the last bits or two: five
// + five nodes are added instead of ten in a row).  Undecorated, n_gauss_prim = 10.
template <bool ASSEMBIAS, bool MODULATE>
__device__ __forceinline__ double occ_bin_zheng07_halves(const double* table, const fm::Consts& kc,
                                                         int g, bool central, bool above, int half,
                                                         const double* log_m_v,
                                                         const double* mass_v,
                                                         const double* weight_v, sc_f64 log_m,
                                                         sc_f64 weight_sum,
                                                         const DrawParams& d) {
  constexpr int kNodes = 10, kHalf = 5;
  const double* lm_p = log_m_v + g * kNodes + half * kHalf;
  const double* m_p = mass_v + g * kNodes + half * kHalf;
  const double* w_p = weight_v + g * kNodes + half * kHalf;
  // (Heaviside assembly bias at the median split, as occ_bin_zheng07: n + s min(n, 1 - n) per
  // central node, the factor 1 + s on a satellite bin's sum, s = +-strength above / below)
  const double s_cen = above ? d.a_cen : -d.a_cen, s_sat = above ? d.a_sat : -d.a_sat;
  double acc = 0.0;
  if (central && ASSEMBIAS) {
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
      const double n =
          fma(0.5, fm::erf_fast(table, kc, (lm_p[k] - d.log_m_min) * d.inv_sigma), 0.5);
      acc = fma(w_p[k], fma(s_cen, fmin(n, 1.0 - n), n), acc);
    }
  } else if (central) {
    // (each half closes its own affine map 0.5 (sum_k w_k erf_k + sum_k w_k) with the weights
    // added in the order of the node sum: where every erf is -1 -- no centrals at all -- the
    // two sums cancel EXACTLY, as the reference's 0.5 (1 + erf) = 0 per node does
    // (tabcorr.py:556-578); against the bin's sum over all ten weights the halves' rounding
    // left -4e-19 there, and 0 / 0 = NaN became a finite number)
    double ws = 0.0;
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
      acc = fma(w_p[k], fm::erf_fast(table, kc, (lm_p[k] - d.log_m_min) * d.inv_sigma), acc);
      ws += w_p[k];
    }
    acc = fma(0.5, acc, 0.5 * ws);
  } else {
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
      const double x = m_p[k] - d.m0;
      double n = fm::exp2_fast(
          table, kc,
          d.alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, d.log2_m1),
          x > 0.0);
      if (MODULATE)
        n *= fma(0.5, fm::erf_fast(table, kc, (lm_p[k] - d.log_m_min) * d.inv_sigma), 0.5);
      acc = fma(w_p[k], n, acc);
    }
  }
  acc += __shfl_xor(acc, 32, 64);            // both halves hold the bin's sum
  if (!central) acc *= d.sat_scale;
  if (!central && ASSEMBIAS) acc = fma(s_sat, acc, acc);
  if (d.any_bad) {
    const int bad = d.bad;
    bool tie = false;
    if ((bad & kTieCen) && (central || MODULATE))
      for (int k = 0; k < kNodes; ++k) tie = tie || log_m[g * kNodes + k] == d.log_m_min;
    const bool cen_nan = (bad & kBadCen) || tie;
    if (!central && (bad & kInfSat) && acc != 0.0)
      acc = ASSEMBIAS ? __builtin_nan("") : __builtin_huge_val();
    if (central ? cen_nan : (((bad & kBadSat) && acc != 0.0) || (MODULATE && cen_nan)))
      acc = __builtin_nan("");
  }
  return acc;
}

// ---- bins that share their quadrature nodes ------------------------------------------------
//
// The secondary-percentile bins of one mass bin (tabcorr.py:186-205: the reference cuts every
// log_prim_haloprop bin into sec_haloprop_percentile bins -- two in its AbacusSummit database,
// scripts/tabulate_snapshot.py:193) have the same log_prim_haloprop_min / max and galaxy type,
// hence the same ten nodes M_gk (tabcorr.py:548-549); only the weights W_gk (through
// prim_haloprop_dist_index) and the side of the assembly-bias split differ.  The expensive part
// of a node -- one erf, or one log2 + exp2 -- is therefore evaluated ONCE per group of such
// bins (table.cpp finds the groups at upload) and every member accumulates it with its own
// weights: per member and node one FMA (undecorated), or the four instructions of the
// Heaviside decoration.  For every bin the instructions and their order are those of
// occ_bin_zheng07: same bits.  n_gauss_prim = 10 (the default) only.
//
// Constants (launch.hip: get_quadrature): the nodes per GROUP (log_m, mass: [group][10]), the
// weights, their sums, n_h and the percentile per MEMBER in group order ([member index][10],
// ...), so that no address depends on a loaded bin index; member[mi] = the library bin of
// member index mi (only the results' row needs it).  Members are processed in pairs: the
// constants of both are in flight together and their two FMA chains interleave.
//
// emit(mi, g, acc): called once per member with its index, its bin g and its mean occupation.
struct GroupConsts {
  sc_f64 log_m, mass;          // (n_groups, 10)
  sc_f64 weight, weight_sum;   // (n_bins, 10), (n_bins) in member order
  sc_f64 percentile;           // (n_bins) in member order
  sc_i32 member;               // (n_bins)
  SeriesConsts series;         // moment expansion, per member in group order (or off)
};

// Defer (predict_cross_fused_kernel, round 5): what happens to the lanes of a group that neither a
// shortcut nor an expansion serves.  NoDefer: they run the node path here, under their part of
// the execution mask -- a wave that holds one such lane pays the whole node loop.  A deferring
// hook (active) gets a call under that mask instead and the members' occupations are
// emitted as 0 for those lanes: the caller evaluates the (group, draw) pairs later, whole waves
// of them at a time.
// A double of lane `lane` (wave-uniform index) as a scalar.
__device__ __forceinline__ double readlane_f64(double value, int lane) {
  const unsigned long long bits = fm::bits_of(value);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffu), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

struct NoDefer {
  static constexpr bool active = false;
  __device__ void operator()(int) const {}
};

// ... the hook of predict_cross_fused_kernel: a word per group in LDS takes the mask of the
// lanes (draws) it is called under (a group is one wave's).
template <bool ACTIVE>
struct MarkPairs {
  static constexpr bool active = ACTIVE;
  unsigned long long* bitmap;
  __device__ void operator()(int group) const {
    bitmap[group] = __builtin_amdgcn_ballot_w64(true);
  }
};

template <bool ASSEMBIAS, bool MODULATE, bool SERIES = true, typename Emit,
          typename Defer = NoDefer>
__device__ __forceinline__ void occ_group_zheng07(const double* table, const fm::Consts& kc,
                                                  int group, int m_begin, int m_end,
                                                  bool central, const GroupConsts& q,
                                                  double split, const DrawParams& d,
                                                  Emit&& emit, Defer defer = Defer()) {
  constexpr int kNodes = 10;
  constexpr bool assembias = ASSEMBIAS;
  constexpr bool modulate = MODULATE;
  const double log_m_min = d.log_m_min, inv_sigma = d.inv_sigma, m0 = d.m0;
  const double log2_m1 = d.log2_m1, sat_scale = d.sat_scale, alpha = d.alpha;
  const int bad = d.bad;
  const bool any_bad = d.any_bad;
  // (the split at the median, f1 = f2 = 1: the only one the device path is given,
  // models.device_spec -- the decoration is n + s min(n, 1 - n) resp. n (1 + s))
  constexpr bool median = assembias;
  sc_f64 log_m = q.log_m + group * kNodes, mass = q.mass + group * kNodes;
  int shortcut = 0;                    // as occ_bin_zheng07: 1 all ones, 2 all zeros
#ifndef TC_NO_OCC_SHORTCUTS
  if (!assembias && !any_bad) {
    if (central) {
      const double z_a = (log_m[0] - log_m_min) * inv_sigma;
      const double z_b = (log_m[kNodes - 1] - log_m_min) * inv_sigma;
      const double z_lo = z_a < z_b ? z_a : z_b, z_hi = z_a < z_b ? z_b : z_a;
      if (__builtin_amdgcn_ballot_w64(!(z_lo >= 6.0)) == 0) shortcut = 1;
      else if (__builtin_amdgcn_ballot_w64(!(z_hi <= -6.0)) == 0) shortcut = 2;
    } else {
      const double m_a = mass[0], m_b = mass[kNodes - 1];
      if (__builtin_amdgcn_ballot_w64((m_a > m_b ? m_a : m_b) > m0) == 0) shortcut = 2;
    }
  }
#endif
  // (undecorated centrals: the moment expansion where the wave's draws allow it -- series.h;
  // the members share the recurrence, each adds its own moments)
  // (SERIES = false: compiled without it -- predict_cross_small_kernel, whose 16 row sums per
  // wave leave no registers for it: 142 with it, one workgroup per CU)
  // Per LANE (series.h): a lane whose own draw allows the expansion takes it, the others the node
  // path below -- `done` marks the former; a wave of both kinds runs both.
  bool done = false;
  if (SERIES && !central && !modulate && shortcut == 0 && q.series.sat_consts != nullptr) {
    sc_i32 thresholds = q.series.sat_thresholds + m_begin * series::sat::kThresholds;
    if (series::sat::eligible(thresholds, d.m0_hi)) {
      // a group of satellite bins well above the draw's M0: the binomial expansion; the
      // members share Mc, eps and the coefficients
      sc_f64 first = q.series.sat_consts + m_begin * series::sat::kStride;
      const double base = first[0] - m0;
      const double eps = first[0] * series::sat::reciprocal(base);
      const double power = fm::exp2_fast(
          table, kc, alpha * fm::log2_fast_offset(table, kc, base, log2_m1));
      for (int mi = m_begin; mi < m_end; mi += 2) {
        const int mj = mi + 1 < m_end ? mi + 1 : mi;
        double acc_i, acc_j;
        series::sat::binomial_sum_pair(q.series.sat_consts + mi * series::sat::kStride,
                                       q.series.sat_consts + mj * series::sat::kStride, eps,
                                       alpha, thresholds, d.m0_hi, &acc_i, &acc_j);
        acc_i = acc_i * power * sat_scale;
        acc_j = acc_j * power * sat_scale;
        if (median) {
          const double s_i = q.percentile[mi] > split ? d.a_sat : -d.a_sat;
          const double s_j = q.percentile[mj] > split ? d.a_sat : -d.a_sat;
          acc_i = fma(s_i, acc_i, acc_i);
          acc_j = fma(s_j, acc_j, acc_j);
        }
        emit(mi, q.member[mi], acc_i);
        if (mj != mi) emit(mj, q.member[mj], acc_j);
      }
      done = true;
    }
  }
  if (SERIES && central && !assembias && shortcut == 0 && q.series.consts != nullptr) {
    sc_i32 thresholds = q.series.thresholds + m_begin * series::kThresholds;
    if (series::eligible(thresholds, d.inv_sigma_hi)) {
      // (a path of its own, so that the node path below stays one straight line whose scalar
      // loads the compiler can start early)
      for (int mi = m_begin; mi < m_end; mi += 2) {
        const int mj = mi + 1 < m_end ? mi + 1 : mi;
        double acc_i, acc_j;
        series::central_sum_pair(table, kc, log_m_min, inv_sigma,
                                 q.series.consts + mi * series::kStride,
                                 q.series.consts + mj * series::kStride, q.weight_sum[mi],
                                 q.weight_sum[mj], thresholds, d.inv_sigma_hi, &acc_i, &acc_j);
        emit(mi, q.member[mi], fma(0.5, acc_i, 0.5 * q.weight_sum[mi]));
        if (mj != mi) emit(mj, q.member[mj], fma(0.5, acc_j, 0.5 * q.weight_sum[mj]));
      }
      done = true;
    }
  }
  if (done) return;
  if (Defer::active && shortcut == 0) {
    // a satellite group at or below the draw's M0: every node gives 0, as the node loop would;
    // everything else goes to the caller's list of pairs
    const bool zero = !central && bad == 0 &&
                      !((mass[0] > mass[kNodes - 1] ? mass[0] : mass[kNodes - 1]) > m0);
    if (!zero) defer(group);
    for (int mi = m_begin; mi < m_end; ++mi) emit(mi, q.member[mi], 0.0);
    return;
  }
  // v[k]: centrals erf(z_k) (decorated: <N_cen> itself), satellites <N_sat> before the scale
  double v[kNodes];
  if (shortcut == 0) {
    if (central) {
#pragma unroll
      for (int k = 0; k < kNodes; ++k) {
        const double e = fm::erf_fast(table, kc, (log_m[k] - log_m_min) * inv_sigma);
        v[k] = assembias ? fma(0.5, e, 0.5) : e;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kNodes; ++k) {
        const double x = mass[k] - m0;
        double n = fm::exp2_fast(
            table, kc,
            alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, log2_m1),
            x > 0.0);
        if (modulate)
          n *= fma(0.5, fm::erf_fast(table, kc, (log_m[k] - log_m_min) * inv_sigma), 0.5);
        v[k] = n;
      }
    }
  }
  bool tie = false;
  if (any_bad && (bad & kTieCen) && (central || modulate))
    for (int k = 0; k < kNodes; ++k) tie = tie || log_m[k] == log_m_min;
  // one member's sum over the nodes, from its weights on (the arithmetic of occ_bin_zheng07)
  auto member_sum = [&](int mi) {
    sc_f64 weight = q.weight + mi * kNodes;
    const bool above = assembias ? q.percentile[mi] > split : false;
    const double s_cen = above ? d.a_cen : -d.a_cen, s_sat = above ? d.a_sat : -d.a_sat;
    double acc = 0.0;
    if (shortcut != 0) {
      if (shortcut == 1) acc = q.weight_sum[mi];
    } else if (central && !assembias) {
#pragma unroll
      for (int k = 0; k < kNodes; ++k) acc = fma(weight[k], v[k], acc);
      acc = fma(0.5, acc, 0.5 * q.weight_sum[mi]);
    } else if (central) {
#pragma unroll
      for (int k = 0; k < kNodes; ++k)
        acc = fma(weight[k], fma(s_cen, fmin(v[k], 1.0 - v[k]), v[k]), acc);
    } else {
#pragma unroll
      for (int k = 0; k < kNodes; ++k) acc = fma(weight[k], v[k], acc);
      acc *= sat_scale;
      if (median) acc = fma(s_sat, acc, acc);
    }
    if (any_bad) {
      const bool cen_nan = (bad & kBadCen) || tie;
      if (!central && (bad & kInfSat) && acc != 0.0)
        acc = assembias ? __builtin_nan("") : __builtin_huge_val();
      if (central ? cen_nan : (((bad & kBadSat) && acc != 0.0) || (modulate && cen_nan)))
        acc = __builtin_nan("");
    }
    return acc;
  };
  for (int mi = m_begin; mi < m_end; mi += 2) {
    const int mj = mi + 1 < m_end ? mi + 1 : mi;       // (clamped: a single member twice)
    const double acc_i = member_sum(mi), acc_j = member_sum(mj);
    emit(mi, q.member[mi], acc_i);
    if (mj != mi) emit(mj, q.member[mj], acc_j);
  }
}

// A group of one or two undecorated bins from its record (series.h, namespace record) for the
// kernels that defer what no expansion serves: the head and the passes every eligible draw
// takes arrive with ONE round trip of scalar loads whose addresses depend on `group` alone; the
// arithmetic per member is that of series::central_sum_pair / sat::binomial_sum_pair -- same bits.
// emit(member index, 0, occupation) once per member; a lane that is neither on a plateau nor
// eligible is handed to defer(group) and emits 0.
template <bool CENTRAL, typename Emit, typename Defer>
__device__ __forceinline__ void occ_record_zheng07(const double* table, const fm::Consts& kc,
                                                   int group, sc_f64 rec, const DrawParams& d,
                                                   Emit&& emit, Defer defer) {
  namespace record = series::record;
  constexpr bool central = CENTRAL;
  record::f64x8_t head, b0, b1, b2;
  record::load_record(rec, head, b0, b1, b2);
  const series::Thresholds limit = record::thresholds_of(head);
  const int members = record::members_of(central, limit, head);
  const int m_begin = members >> 1;
  const bool two = (members & 1) != 0;
  double out_i = 0.0, out_j = 0.0;
  if (central) {
    if (!d.any_bad) {
      // (the plateaus of occ_group_zheng07: every node of every lane at 1, or at 0)
      const double z_a = (head.v[record::kLow] - d.log_m_min) * d.inv_sigma;
      const double z_b = (head.v[record::kHigh] - d.log_m_min) * d.inv_sigma;
      const double z_lo = z_a < z_b ? z_a : z_b, z_hi = z_a < z_b ? z_b : z_a;
      const bool ones = __builtin_amdgcn_ballot_w64(!(z_lo >= 6.0)) == 0;
      if (ones || __builtin_amdgcn_ballot_w64(!(z_hi <= -6.0)) == 0) {
        emit(m_begin, 0, ones ? head.v[record::kFirstSum] : 0.0);
        if (two) emit(m_begin + 1, 0, ones ? head.v[record::kFirstSum + 1] : 0.0);
        return;
      }
    }
    if (d.inv_sigma_hi < limit.v[series::kSteps - 1]) {
      const double inv_sigma = d.inv_sigma;
      double g0, z0;
      const double e = fm::erf_gauss_fast(
          table, kc, (head.v[record::kCentre] - d.log_m_min) * inv_sigma, &g0, &z0);
      const double a = 2.0 * z0 * inv_sigma, b = -2.0 * inv_sigma * inv_sigma;
      double p_prev = 0.0, p = inv_sigma, nb = -b, sum_i = 0.0, sum_j = 0.0;
      auto pass = [&](const record::f64x8_t& m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sum_i = fma(p, m.v[k], sum_i);
          sum_j = fma(p, m.v[4 + k], sum_j);
          nb += b;
          const double next = fma(a, p, nb * p_prev);
          p_prev = p;
          p = next;
        }
      };
      pass(b0);
      pass(b1);
      // (most waves need no more: a lane's count of passes only where one of them does)
      if (__builtin_amdgcn_ballot_w64(d.inv_sigma_hi >= limit.v[0]) != 0) {
        const int n_blocks = series::passes<series::kSteps>(limit, d.inv_sigma_hi, 2);
        if (n_blocks > 2) pass(b2);
        sc_f64 further = rec + record::kHead + 3 * record::kBlock;
#pragma unroll 1
        for (int block = 3; block < n_blocks; ++block) {
          pass(record::load_eight(further));
          further += record::kBlock;
        }
      }
      const double m0_i = head.v[record::kFirstSum], m0_j = head.v[record::kFirstSum + 1];
      out_i = fma(0.5, fma(g0, sum_i, m0_i * e), 0.5 * m0_i);
      out_j = fma(0.5, fma(g0, sum_j, m0_j * e), 0.5 * m0_j);
    } else {
      defer(group);
    }
  } else {
    const double m0 = d.m0, largest = head.v[record::kLow];
    if (!d.any_bad && __builtin_amdgcn_ballot_w64(largest > m0) == 0) {
      emit(m_begin, 0, 0.0);
      if (two) emit(m_begin + 1, 0, 0.0);
      return;
    }
    if (d.m0_hi < limit.v[series::sat::kSteps - 1]) {
      const double centre = head.v[record::kCentre], alpha = d.alpha;
      const double base = centre - m0;
      const double eps = centre * series::sat::reciprocal(base);
      const double power = fm::exp2_fast(
          table, kc, alpha * fm::log2_fast_offset(table, kc, base, d.log2_m1));
      double c = 1.0, g = eps * alpha;
      double sum_i = head.v[record::kFirstSum], sum_j = head.v[record::kFirstSum + 1];
      auto pass = [&](const record::f64x8_t& m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          c *= g;
          g -= eps;
          sum_i = fma(c, m.v[k], sum_i);
          sum_j = fma(c, m.v[4 + k], sum_j);
        }
      };
      pass(b0);
      pass(b1);
      pass(b2);
      if (__builtin_amdgcn_ballot_w64(d.m0_hi >= limit.v[0]) != 0) {
        const int n_blocks = series::passes<series::sat::kSteps>(limit, d.m0_hi, 3);
        sc_f64 further = rec + record::kHead + 3 * record::kBlock;
#pragma unroll 1
        for (int block = 3; block < n_blocks; ++block) {
          pass(record::load_eight(further));
          further += record::kBlock;
        }
      }
      out_i = sum_i * power * d.sat_scale;
      out_j = sum_j * power * d.sat_scale;
    } else if (!(d.bad == 0 && !(largest > m0))) {
      // (at or below the draw's M0 every node gives 0, as the node loop would)
      defer(group);
    }
  }
  emit(m_begin, 0, out_i);
  if (two) emit(m_begin + 1, 0, out_j);
}

// The same for the 32-draw workgroups (lane = (draw, half of the nodes), occ_bin_zheng07_halves):
// five node values per lane, every member's weights by vector loads -- those of a pair of
// members requested BEFORE the nodes are evaluated, so that they arrive under that arithmetic
// --, the halves added through one lane exchange per member.
template <bool ASSEMBIAS, bool MODULATE, typename Emit>
__device__ __forceinline__ void occ_group_zheng07_halves(
    const double* table, const fm::Consts& kc, int group, int m_begin, int m_end, bool central,
    int half, const double* log_m_v, const double* mass_v, const double* weight_v,
    const GroupConsts& q, double split, const DrawParams& d, Emit&& emit) {
  constexpr int kNodes = 10, kHalf = 5;
  const double* lm_p = log_m_v + group * kNodes + half * kHalf;
  const double* m_p = mass_v + group * kNodes + half * kHalf;
  double v[kHalf];
  bool tie = false;
  if (d.any_bad && (d.bad & kTieCen) && (central || MODULATE))
    for (int k = 0; k < kNodes; ++k) tie = tie || q.log_m[group * kNodes + k] == d.log_m_min;
  auto finish = [&](int mi, double acc) {
    const bool above = ASSEMBIAS ? q.percentile[mi] > split : false;
    const double s_sat = above ? d.a_sat : -d.a_sat;
    acc += __shfl_xor(acc, 32, 64);            // both halves hold the bin's sum
    if (!central) acc *= d.sat_scale;
    if (!central && ASSEMBIAS) acc = fma(s_sat, acc, acc);
    if (d.any_bad) {
      const int bad = d.bad;
      const bool cen_nan = (bad & kBadCen) || tie;
      if (!central && (bad & kInfSat) && acc != 0.0)
        acc = ASSEMBIAS ? __builtin_nan("") : __builtin_huge_val();
      if (central ? cen_nan : (((bad & kBadSat) && acc != 0.0) || (MODULATE && cen_nan)))
        acc = __builtin_nan("");
    }
    return acc;
  };
  for (int mi = m_begin; mi < m_end; mi += 2) {
    const int mj = mi + 1 < m_end ? mi + 1 : mi;
    double w_i[kHalf], w_j[kHalf];
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
      w_i[k] = weight_v[mi * kNodes + half * kHalf + k];
      w_j[k] = weight_v[mj * kNodes + half * kHalf + k];
    }
    if (mi == m_begin) {
      if (central) {
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
          const double e = fm::erf_fast(table, kc, (lm_p[k] - d.log_m_min) * d.inv_sigma);
          v[k] = ASSEMBIAS ? fma(0.5, e, 0.5) : e;
        }
      } else {
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
          const double x = m_p[k] - d.m0;
          double n = fm::exp2_fast(
              table, kc,
              d.alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, d.log2_m1),
              x > 0.0);
          if (MODULATE)
            n *= fma(0.5, fm::erf_fast(table, kc, (lm_p[k] - d.log_m_min) * d.inv_sigma), 0.5);
          v[k] = n;
        }
      }
    }
    double acc_i = 0.0, acc_j = 0.0;
    if (central && ASSEMBIAS) {
      const double s_i = q.percentile[mi] > split ? d.a_cen : -d.a_cen;
      const double s_j = q.percentile[mj] > split ? d.a_cen : -d.a_cen;
#pragma unroll
      for (int k = 0; k < kHalf; ++k) {
        const double low = fmin(v[k], 1.0 - v[k]);
        acc_i = fma(w_i[k], fma(s_i, low, v[k]), acc_i);
        acc_j = fma(w_j[k], fma(s_j, low, v[k]), acc_j);
      }
    } else {
#pragma unroll
      for (int k = 0; k < kHalf; ++k) {
        acc_i = fma(w_i[k], v[k], acc_i);
        acc_j = fma(w_j[k], v[k], acc_j);
      }
      if (central) {
        // (the half's own affine map, weights added in the order of its node sum: exact
        // cancellation where every erf is -1 -- occ_bin_zheng07_halves)
        double ws_i = 0.0, ws_j = 0.0;
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
          ws_i += w_i[k];
          ws_j += w_j[k];
        }
        acc_i = fma(0.5, acc_i, 0.5 * ws_i);
        acc_j = fma(0.5, acc_j, 0.5 * ws_j);
      }
    }
    acc_i = finish(mi, acc_i);
    emit(mi, q.member[mi], acc_i);
    if (mj != mi) emit(mj, q.member[mj], finish(mj, acc_j));
  }
}

// Mean occupation of every bin for every draw: tabcorr/tabcorr.py:537-578 with
// the two halotools callbacks of :556-563 evaluated inline (Zheng et al. 2007
// eqs. 1 and 3).  Work items = (draw tile, bin split); the kOccWaves waves of a
// block share an item's draw tile and interleave over the bins of its split, and
// the blocks stride over the items (the math tables are staged once per block).
// An item's centrals / satellites density sums go to ngal_part[split][2][ldb].

// NGAUSS > 0: n_gauss known at compile time, node loop fully unrolled (the scalar loads
// of a bin's constants are batched and the independent polynomial chains interleave);
// NGAUSS == 0: any n_gauss.
// GROUPED (NGAUSS == 10): the work items are ranges of GROUPS of bins that share their nodes
// (occ_group_zheng07); a.n_groups / a.n_central_groups take the place of the bin counts.
template <int NGAUSS, bool ASSEMBIAS, bool MODULATE, bool GROUPED = false>
__global__ __launch_bounds__(kOccWaves * kLanes) void occ_zheng07_kernel(
    OccArgs a) {
  static_assert(!GROUPED || NGAUSS == 10, "groups of bins: ten nodes per bin");
  __shared__ double red[2][kOccWaves][kLanes];
  __shared__ double prm[9][kLanes];
  __shared__ __attribute__((aligned(16))) double table[fm::kTableDoubles];
  const fm::Consts kc = fm::make_consts();
  // short kernel on the critical path of its lane: run ahead of the contraction waves
  // of neighbouring batches it shares the CUs with
  set_priority((int)(a.flags >> 8) & 3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_gauss = NGAUSS > 0 ? NGAUSS : a.n_gauss;
  const int n_items = a.n_tiles * a.n_splits;
  // (ranges of bins, or of groups of bins)
  const int n_units = GROUPED ? a.n_groups : a.n_bins;
  const int n_central_units = GROUPED ? a.n_central_groups : a.n_central;
  const int per_block = (n_units + a.n_splits - 1) / a.n_splits;
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    // a block with a single all-satellites item does not need the erf rows (the exp2
    // table is always needed: 10^logM0)
    const bool need_erf = (int)gridDim.x < n_items || MODULATE ||
                          ((int)blockIdx.x / a.n_tiles) * per_block < n_central_units;
    const int lo = need_erf ? 0 : fm::kLogOffset / 2;
    const int hi = fm::kTableDoubles / 2;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    int i = lo + threadIdx.x;
    for (; i + 3 * (int)blockDim.x < hi; i += 4 * blockDim.x) {
      double2v v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[i + u * blockDim.x];
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[i + u * blockDim.x] = v[u];
    }
    for (; i < hi; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  sc_f64 log_m = (sc_f64)a.log_m;
  sc_f64 mass = (sc_f64)a.m;
  sc_f64 weight = (sc_f64)a.weight;
  sc_f64 weight_sum = weight + a.n_bins * n_gauss;     // (get_quadrature: the bins' sums)
  sc_f64 n_h = (sc_f64)a.n_h;
  sc_f64 percentile = (sc_f64)a.percentile;
  sc_i32 perm = (sc_i32)a.perm;
  constexpr bool assembias = ASSEMBIAS;
  const double f1 = (1.0 - a.split) / a.split, f2 = a.split / (1.0 - a.split);
  // (f1 = f1/f2 and f2 = f2/f1 of Hearin et al.'s population fractions)

  // items = (draw tile, bin split): a contiguous range of bins, so that most items are
  // all-centrals or all-satellites; the blocks stride over the items
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int tile = item % a.n_tiles, split = item / a.n_tiles;
    const int64_t b0 = (int64_t)tile * kLanes + lane;
    const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;
    const int g_begin = split * per_block;
    const int g_end = g_begin + per_block < n_units ? g_begin + per_block : n_units;
    // per-draw quantities: computed by wave 0, shared with the other waves through LDS
    if (wave == 0) {
      const double* th = a.theta + b * a.n_theta;
      const DrawSetup d = prepare_draw(table, kc, th[0], th[1], th[2], th[3], th[4],
                                       assembias ? th[5] : 0.0, assembias ? th[6] : 0.0);
      prm[0][lane] = d.log_m_min;
      prm[1][lane] = d.inv_sigma;
      prm[2][lane] = d.m0;
      prm[3][lane] = d.log2_m1;
      prm[4][lane] = d.sat_scale;
      prm[5][lane] = d.alpha;
      prm[6][lane] = d.a_cen;
      prm[7][lane] = d.a_sat;
      prm[8][lane] = (double)d.bad;
    }
    __syncthreads();
    DrawParams dp;
    dp.log_m_min = prm[0][lane];
    dp.inv_sigma = prm[1][lane];
    dp.m0 = prm[2][lane];
    dp.log2_m1 = prm[3][lane];
    dp.sat_scale = prm[4][lane];
    dp.alpha = prm[5][lane];
    dp.a_cen = assembias ? prm[6][lane] : 0.0;
    dp.a_sat = assembias ? prm[7][lane] : 0.0;
    dp.bad = (int)prm[8][lane];
    // wave-uniform: does any draw of this tile need the NaN fix-ups after a bin's node loop?
    dp.any_bad = __builtin_amdgcn_ballot_w64(dp.bad != 0) != 0;
    series_setup<MODULATE>(dp, (GROUPED ? a.group.series : a.series) != nullptr,
                           (GROUPED ? a.group.sat_series : a.sat_series) != nullptr);

    double sum_cen = 0.0, sum_sat = 0.0;
    auto emit = [&](int g, bool central, double acc, double n_h_g) {
      if (a.occupation != nullptr && b0 < a.n_draws)
        a.occupation[b0 * a.n_bins + perm[g]] = acc;
      const double dens = acc * n_h_g;
      a.nbuf[(int64_t)g * a.ldb + (int64_t)tile * kLanes + lane] = dens;
      if (a.nbuf32 != nullptr)
        a.nbuf32[(int64_t)g * a.ldb + (int64_t)tile * kLanes + lane] = (float)dens;
      if (central) sum_cen += dens; else sum_sat += dens;
    };
    for (int g = g_begin + wave; g < g_end; g += kOccWaves) {
      if (GROUPED) {
        sc_i32 group_begin = (sc_i32)a.group.begin;
        sc_f64 n_h_m = (sc_f64)a.group.n_h;
        const GroupConsts gq{(sc_f64)a.group.log_m, (sc_f64)a.group.m, (sc_f64)a.group.weight,
                             (sc_f64)a.group.weight + a.n_bins * 10,
                             (sc_f64)a.group.percentile, (sc_i32)a.group.member,
                             SeriesConsts{(sc_f64)a.group.series, (sc_i32)a.group.series_thr,
                                          (sc_f64)a.group.sat_series, (sc_i32)a.group.sat_series_thr}};
        const bool central = g < a.n_central_groups;
        occ_group_zheng07<ASSEMBIAS, MODULATE>(
            table, kc, g, group_begin[g], group_begin[g + 1], central, gq, a.split, dp,
            [&](int mi, int bin, double acc) { emit(bin, central, acc, n_h_m[mi]); });
      } else {
        const bool central = g < a.n_central;
        const bool above = percentile[g] > a.split;
        emit(g, central,
             occ_bin_zheng07<NGAUSS, ASSEMBIAS, MODULATE>(
                 table, kc, g, n_gauss, central, above, log_m, mass, weight, weight_sum, dp, f1,
                 f2, SeriesConsts{(sc_f64)a.series, (sc_i32)a.series_thr, (sc_f64)a.sat_series,
                              (sc_i32)a.sat_series_thr}),
             n_h[g]);
      }
    }
    red[0][wave][lane] = sum_cen;
    red[1][wave][lane] = sum_sat;
    __syncthreads();
    if (wave < 2) {
      double total = 0.0;
#pragma unroll
      for (int w = 0; w < kOccWaves; ++w) total += red[wave][w][lane];
      a.ngal[((int64_t)split * 2 + wave) * a.ldb + (int64_t)tile * kLanes + lane] = total;
    }
    __syncthreads();
  }
}

// ---- Leauthaud et al. (2011) occupation family ------------------------------------------
//
// The second family evaluated on the device (the callbacks the
// reference would get from halotools' Leauthaud11Cens / Leauthaud11Sats at
// tabcorr.py:556-563), restated from the papers:
//   stellar-to-halo mass relation (Behroozi, Conroy & Wechsler 2010, eq. 21), h = the Hubble
//   parameter of the relation (halotools' Behroozi10SmHm: 0.7), h_s that of the satellite
//   terms (halotools' Leauthaud11Sats: 0.72):
//     log10 M_h(M*) = logm1 + beta x + 10^(delta x) / (1 + 10^(-gamma x)) - 1/2 - log10 h,
//     x = log10 M* + 2 log10 h - logm0
//   centrals (Leauthaud et al. 2011, eq. 8):
//     <N_cen>(M_h) = 1/2 [1 - erf((threshold - log10 M*(M_h)) / (sqrt 2 scatter))]
//     with M*(M_h) the INVERSE of the relation above;
//   satellites (eq. 12): <N_sat>(M_h) = [<N_cen>] (M_h h_s / M_sat)^alphasat
//     exp(-M_cut / (M_h h_s)), M_sat = 1e12 bsat (M_knee / 1e12)^betasat, M_cut = 1e12 bcut
//     (M_knee / 1e12)^betacut, M_knee = h_s M_h(M* = 10^threshold).
// theta columns: logm0, logm1, beta, delta, gamma, scatter, alphasat, bsat, betasat, bcut,
// betacut, threshold, h, h_s.  The inverse relation is solved per quadrature node by
// Newton's method on g(x) = beta x + 10^(delta x) / (1 + 10^(-gamma x)) = log10(M_h h) + 1/2
// - logm1, started right of the root (x0 = min(T / beta, log10(2 T) / delta)), to the last
// bit (<= 11 iterations over the prior box, 5 typically); halotools itself interpolates a
// 100-point table of the relation with a cubic spline, which is NOT reproduced here: only
// this package's own Leauthaud11Model is routed to this kernel (models.device_spec).
constexpr int kLeauthaudTheta = 14;
constexpr unsigned kFlagLeauthaud11 = 16u;

struct SmhmSetup {
  double log_m0, beta, delta, gamma, offset;   // offset: log10 h + 1/2 - logm1
  double two_log_h;
};

// log10 M* (in h = 1 units) of a halo of log10 mass `log_mh`.
__device__ inline double smhm_log_mstar(const double* table, const fm::Consts& kc,
                                        const SmhmSetup& s, double log_mh) {
  const double target = log_mh + s.offset;
  double x = target / s.beta;
  if (target > 0.5) {
    const double alt = fm::log2_fast(table, kc, 2.0 * target) * (0.30102999566398119521 / s.delta);
    x = alt < x ? alt : x;
  }
  constexpr double kLn10 = 2.30258509299404568402;
  for (int iteration = 0; iteration < 24; ++iteration) {
    const double a = fm::exp10_fast(table, kc, s.delta * x);
    const double b = fm::exp10_fast(table, kc, -s.gamma * x);
    const double inv = 1.0 / (1.0 + b);
    const double g = fma(s.beta, x, a * inv);
    const double slope = fma(kLn10 * a * inv, fma(s.gamma * b, inv, s.delta), s.beta);
    const double step = (g - target) / slope;
    x -= step;
    const double scale = fabs(x) > 1.0 ? fabs(x) : 1.0;
    // wave-uniform exit: every lane within one part in 10^15 (NaN lanes never hold it up)
    if (__builtin_amdgcn_ballot_w64(fabs(step) > 1e-15 * scale) == 0) break;
  }
  return x + s.log_m0 - s.two_log_h;
}

// Per-draw constants of the Leauthaud11 occupation functions (theta columns: see above) and
// the mean occupation of one bin: shared by occ_leauthaud11_kernel and predict_fused_kernel.
struct LeauthaudDraw {
  SmhmSetup smhm;
  double inv_scatter, threshold, alphasat, log2_msat, cut;
};

__device__ inline LeauthaudDraw prepare_leauthaud(const double* table, const fm::Consts& kc,
                                                  const double* th) {
  constexpr double kLog2Of10 = 3.32192809488736234787, kLog10Of2 = 0.30102999566398119521;
  constexpr double kLog2E = 1.44269504088896340736;
  const double log_m0 = th[0], log_m1 = th[1], beta = th[2], delta = th[3], gamma = th[4];
  const double scatter = th[5], alphasat = th[6], bsat = th[7], betasat = th[8];
  const double bcut = th[9], betacut = th[10], threshold = th[11], h = th[12];
  const double h_sat = th[13];
  const double log_h = fm::log2_fast(table, kc, h > 1e-300 ? h : 1e-300) * kLog10Of2;
  const double log_hs = fm::log2_fast(table, kc, h_sat > 1e-300 ? h_sat : 1e-300) * kLog10Of2;
  // halo mass of the threshold: the forward relation, then the knee in units of 1e12
  const double x_t = threshold + 2.0 * log_h - log_m0;
  const double a_t = fm::exp10_fast(table, kc, delta * x_t);
  const double b_t = fm::exp10_fast(table, kc, -gamma * x_t);
  const double log_knee = log_m1 + beta * x_t + a_t / (1.0 + b_t) - 0.5 - log_h + log_hs - 12.0;
  // log2 M_sat and M_cut * log2 e (bsat <= 0 or bcut < 0 have no real power law: NaN)
  const double log2_msat =
      bsat > 0.0 ? fm::log2_fast(table, kc, bsat) + (12.0 + betasat * log_knee) * kLog2Of10
                 : __builtin_nan("");
  const double mcut =
      bcut > 0.0 ? fm::exp2_fast(table, kc, fm::log2_fast(table, kc, bcut) +
                                                (12.0 + betacut * log_knee) * kLog2Of10)
                 : (bcut == 0.0 ? 0.0 : __builtin_nan(""));
  LeauthaudDraw d;
  d.smhm.log_m0 = log_m0;
  d.smhm.beta = beta;
  d.smhm.delta = delta;
  d.smhm.gamma = gamma;
  d.smhm.offset = log_h + 0.5 - log_m1;
  d.smhm.two_log_h = 2.0 * log_h;
  d.inv_scatter = 1.0 / (1.41421356237309504880 * scatter);
  d.threshold = threshold;
  d.alphasat = alphasat;
  d.log2_msat = log2_msat - log_hs * kLog2Of10;     // (M h_s / M_sat): log2 M - this
  d.cut = -mcut * kLog2E / h_sat;                   // exp(-M_cut / (M h_s)) = 2^(this / M)
  return d;
}

template <bool MODULATE>
__device__ __forceinline__ double occ_bin_leauthaud11(const double* table, const fm::Consts& kc,
                                                      int g, int n_gauss, bool central,
                                                      sc_f64 log_m, sc_f64 mass, sc_f64 weight,
                                                      const LeauthaudDraw& d) {
  constexpr double kLog2Of10 = 3.32192809488736234787;
  double acc = 0.0;
  for (int k = 0; k < n_gauss; ++k) {
    const double lm = log_m[g * n_gauss + k];
    double n_cen = 1.0;
    if (central || MODULATE) {
      const double log_mstar = smhm_log_mstar(table, kc, d.smhm, lm);
      const double za = (d.threshold - log_mstar) * d.inv_scatter;
      n_cen = fma(-0.5, fm::erf_fast(table, kc, za), 0.5);
      n_cen = za != za ? za : n_cen;      // NaN parameters stay NaN (erf_fast clamps)
    }
    double n = n_cen;
    if (!central) {
      const double m = mass[g * n_gauss + k];
      // (M h / M_sat)^alphasat exp(-M_cut / (M h)) as one power of two
      const double z = fma(d.alphasat, lm * kLog2Of10 - d.log2_msat, d.cut / m);
      n = fm::exp2_fast(table, kc, z);
      n = z != z ? z : n;                 // NaN parameters stay NaN (exp2_fast clamps)
      if (MODULATE) n *= n_cen;
    }
    acc = fma(weight[g * n_gauss + k], n, acc);
  }
  return acc;
}

// The same for the 32-draw workgroups of predict_fused_kernel: lane = (draw, half), five of the
// bin's ten nodes per half (constants by vector loads), the halves added by a lane exchange.
template <bool MODULATE>
__device__ __forceinline__ double occ_bin_leauthaud11_halves(const double* table,
                                                             const fm::Consts& kc, int g,
                                                             bool central, int half,
                                                             const double* log_m_v,
                                                             const double* mass_v,
                                                             const double* weight_v,
                                                             const LeauthaudDraw& d) {
  constexpr double kLog2Of10 = 3.32192809488736234787;
  constexpr int kNodes = 10, kHalf = 5;
  const double* lm_p = log_m_v + g * kNodes + half * kHalf;
  const double* m_p = mass_v + g * kNodes + half * kHalf;
  const double* w_p = weight_v + g * kNodes + half * kHalf;
  double acc = 0.0;
  for (int k = 0; k < kHalf; ++k) {
    const double lm = lm_p[k];
    double n_cen = 1.0;
    if (central || MODULATE) {
      const double log_mstar = smhm_log_mstar(table, kc, d.smhm, lm);
      const double za = (d.threshold - log_mstar) * d.inv_scatter;
      n_cen = fma(-0.5, fm::erf_fast(table, kc, za), 0.5);
      n_cen = za != za ? za : n_cen;
    }
    double n = n_cen;
    if (!central) {
      const double z = fma(d.alphasat, lm * kLog2Of10 - d.log2_msat, d.cut / m_p[k]);
      n = fm::exp2_fast(table, kc, z);
      n = z != z ? z : n;
      if (MODULATE) n *= n_cen;
    }
    acc = fma(w_p[k], n, acc);
  }
  return acc + __shfl_xor(acc, 32, 64);
}

// Same work decomposition, outputs and launch geometry as occ_zheng07_kernel.
template <bool MODULATE>
__global__ __launch_bounds__(kOccWaves * kLanes) void occ_leauthaud11_kernel(OccArgs a) {
  __shared__ double red[2][kOccWaves][kLanes];
  __shared__ double prm[12][kLanes];
  __shared__ __attribute__((aligned(16))) double table[fm::kTableDoubles];
  const fm::Consts kc = fm::make_consts();
  set_priority((int)(a.flags >> 8) & 3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_gauss = a.n_gauss;
  const int n_items = a.n_tiles * a.n_splits;
  const int per_block = (a.n_bins + a.n_splits - 1) / a.n_splits;
  for (int i = threadIdx.x; i < fm::kTableDoubles; i += blockDim.x) table[i] = a.math_table[i];
  __syncthreads();
  sc_f64 log_m = (sc_f64)a.log_m;
  sc_f64 mass = (sc_f64)a.m;
  sc_f64 weight = (sc_f64)a.weight;
  sc_f64 n_h = (sc_f64)a.n_h;
  sc_i32 perm = (sc_i32)a.perm;

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int tile = item % a.n_tiles, split = item / a.n_tiles;
    const int64_t b0 = (int64_t)tile * kLanes + lane;
    const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;
    const int g_begin = split * per_block;
    const int g_end = g_begin + per_block < a.n_bins ? g_begin + per_block : a.n_bins;
    if (wave == 0) {
      const LeauthaudDraw d = prepare_leauthaud(table, kc, a.theta + b * a.n_theta);
      prm[0][lane] = d.smhm.log_m0;
      prm[1][lane] = d.smhm.beta;
      prm[2][lane] = d.smhm.delta;
      prm[3][lane] = d.smhm.gamma;
      prm[4][lane] = d.smhm.offset;
      prm[5][lane] = d.smhm.two_log_h;
      prm[6][lane] = d.inv_scatter;
      prm[7][lane] = d.threshold;
      prm[8][lane] = d.alphasat;
      prm[9][lane] = d.log2_msat;
      prm[10][lane] = d.cut;
    }
    __syncthreads();
    LeauthaudDraw draw;
    draw.smhm.log_m0 = prm[0][lane];
    draw.smhm.beta = prm[1][lane];
    draw.smhm.delta = prm[2][lane];
    draw.smhm.gamma = prm[3][lane];
    draw.smhm.offset = prm[4][lane];
    draw.smhm.two_log_h = prm[5][lane];
    draw.inv_scatter = prm[6][lane];
    draw.threshold = prm[7][lane];
    draw.alphasat = prm[8][lane];
    draw.log2_msat = prm[9][lane];
    draw.cut = prm[10][lane];

    double sum_cen = 0.0, sum_sat = 0.0;
    for (int g = g_begin + wave; g < g_end; g += kOccWaves) {
      const bool central = g < a.n_central;
      const double acc = occ_bin_leauthaud11<MODULATE>(table, kc, g, n_gauss, central, log_m,
                                                       mass, weight, draw);
      if (a.occupation != nullptr && b0 < a.n_draws)
        a.occupation[b0 * a.n_bins + perm[g]] = acc;
      const double dens = acc * n_h[g];
      a.nbuf[(int64_t)g * a.ldb + (int64_t)tile * kLanes + lane] = dens;
      if (a.nbuf32 != nullptr)
        a.nbuf32[(int64_t)g * a.ldb + (int64_t)tile * kLanes + lane] = (float)dens;
      if (central) sum_cen += dens; else sum_sat += dens;
    }
    red[0][wave][lane] = sum_cen;
    red[1][wave][lane] = sum_sat;
    __syncthreads();
    if (wave < 2) {
      double total = 0.0;
#pragma unroll
      for (int w = 0; w < kOccWaves; ++w) total += red[wave][w][lane];
      a.ngal[((int64_t)split * 2 + wave) * a.ldb + (int64_t)tile * kLanes + lane] = total;
    }
    __syncthreads();
  }
}

// Occupations supplied by the caller (the ndarray seam, tabcorr.py:616-623):
// nbuf[g'][b] = occupation[b][perm[g']] * n_h[g'] and the two sums.
#ifdef TC_UNIT_QUAD   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(256) void occ_from_array_kernel(
    const double* occupation, int64_t n_draws, int64_t ldb, int n_bins,
    int n_central, const double* n_h, const int32_t* perm, double* nbuf,
    double* ngal, float* nbuf32) {
  const int64_t b0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b0 >= ldb) return;
  const int64_t b = b0 < n_draws ? b0 : n_draws - 1;
  double sum_cen = 0.0, sum_sat = 0.0;
  for (int g = 0; g < n_bins; ++g) {
    const double dens = occupation[b * n_bins + perm[g]] * n_h[g];
    nbuf[(int64_t)g * ldb + b0] = dens;
    if (nbuf32 != nullptr) nbuf32[(int64_t)g * ldb + b0] = (float)dens;
    if (g < n_central) sum_cen += dens; else sum_sat += dens;
  }
  ngal[b0] = sum_cen;
  ngal[ldb + b0] = sum_sat;
}
#endif

// ---- FP64 matrix-core contraction -----------------------------------------------------
//
// partial[slab][r][draw] = sum over the slab's table positions q of T[q][r] w[q][draw],
// w = n_i n_j of the position's bin pair (tabcorr.py:626-655 with the pair prefactor
// folded into T).  grid = (8 * ceil(tiles / 8) * slabs, 1, r tiles), XCD-aware: linear
// block id b runs on XCD b % 8 (each with its own L2) and all slabs of a draw tile get
// ids with the same b % 8, so the tile's density rows, re-read by every slab, come from
// that XCD's L2.  A workgroup stages
// the density rows of its group's bins for its draw tile in LDS, each wave walks one
// chunk of positions, the waves' sums are combined by a fixed-order tree in LDS.
//
// The inner product runs on v_mfma_f64_4x4x4_4b_f64: 4 blocks of D(4x4) += A(4x4) B(4x4).
// The FP64 matrix rate equals the FP64 vector rate on gfx950, but vector FMAs are power
// throttled under sustained load (tools/micro/sustained.hip: 63-67 TFLOP/s against 75
// for this instruction), and the matrix form needs a fifth of the vector work around it
// (the first version of this kernel, vector FMAs with DPP row broadcasts of the table
// values, reached 45.7 TFLOP/s: profiles/r01_notes.md).
//
// Operand mapping (probed with tools/micro/mfma_map.hip): lane l holds A[i][k] and
// B[k][j] with i or j = l % 4, block = (l / 4) % 4, k = l / 16, and D[i][j] with
// j = l % 4, block = (l / 4) % 4, i = l / 16.  One instruction covers 4 consecutive
// table positions (k), 4 r values (i) and 16 draws (block, j):
//   A (per r sub-tile u): T[position q0 + l / 16][r = 4 u + l % 4], the same in all blocks;
//     table layout [q0 / 8][u][k][i][step parity]: one 16-byte load per lane serves two
//     consecutive steps;
//   B (per draw set s): w[draw 16 s + l % 16][position q0 + l / 16] = n_i n_j, formed from
//     two LDS gathers per set at row offsets the lane gets from pos_off[q0 + l / 16];
//   D (per set s and sub-tile u): draw 16 s + l % 16, r = 4 u + l / 16.
// Per 4 positions: (RT / 4 + 1) / 2 vector loads, 8 LDS reads (4 ds_read2_b64), 4 v_mul_f64
// and RT MFMAs.
template <int RT, bool INTERP>
__global__ __launch_bounds__(512) void contract_mfma_kernel(ContractArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NT = RT / 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  int tile, slab;
  if (a.xcd_map) {   // full chip (8 XCDs) and at least 8 draw tiles
    const int xcd = blockIdx.x & 7;
    const int rest = blockIdx.x >> 3;
    tile = (rest / a.n_slabs) * 8 + xcd;
    slab = rest % a.n_slabs;
  } else {   // fewer draw tiles than XCDs, or a partitioned device: plain order
    tile = blockIdx.x / a.n_slabs;
    slab = blockIdx.x % a.n_slabs;
  }
  if (tile >= a.n_tiles) return;
  const int64_t col = (int64_t)tile * kLanes;

  constexpr bool interp = INTERP;
  unsigned long long t_start = 0, t_staged = 0, t_main = 0, c_staged = 0, c_main = 0;
  if (TC_TRACE(a.trace)) t_start = __builtin_amdgcn_s_memrealtime();
  const int k_splits = interp ? a.k_splits : 1;
  const Group group = a.groups[slab / k_splits];
  const int n_rows_j = group.j_hi - group.j_lo;
  const int n_rows = n_rows_j + (group.i_hi - group.i_lo);
  int k_begin = 0, k_end = 1;
  if (interp) {
    const int split = slab % k_splits;
    k_begin = (int)((int64_t)a.n_tables * split / k_splits);
    k_end = (int)((int64_t)a.n_tables * (split + 1) / k_splits);
  }

  double acc[4][NT];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[s][u] = 0.0;

  int staged_class = -1;
  for (int k = k_begin; k < k_end; ++k) {
    const int density_class = interp ? a.table_class[k] : 0;
    if (density_class != staged_class) {
      if (staged_class >= 0) __syncthreads();
      typedef double __attribute__((ext_vector_type(2))) double2v;
      typedef const __attribute__((address_space(1))) double2v* gl_f64x2;
      const int n_items = n_rows * (kLanes / 2);
      gl_f64 src = (gl_f64)(interp ? a.nbufs[density_class] : a.nbuf) + col;
      auto source = [&](int id) {
        const int row = id >> 5;
        const int bin = row < n_rows_j ? group.j_lo + row : group.i_lo + row - n_rows_j;
        return (gl_f64x2)(src + (int64_t)bin * a.ldb + (id & 31) * 2);
      };
      const int nthreads = blockDim.x;
      int it = threadIdx.x;
      for (; it + 3 * nthreads < n_items; it += 4 * nthreads) {
        double2v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *source(it + u * nthreads);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int id = it + u * nthreads;
          *(double2v*)(lds + (id >> 5) * kLanes + (id & 31) * 2) = v[u];
        }
      }
      for (; it < n_items; it += nthreads)
        *(double2v*)(lds + (it >> 5) * kLanes + (it & 31) * 2) = *source(it);
      // mode cross has no row bin: its "n_i" is a row of ones behind the staged rows
      if (a.mode != 0 && threadIdx.x < kLanes) lds[n_rows * kLanes + threadIdx.x] = 1.0;
      __syncthreads();
      staged_class = density_class;
      if (TC_TRACE(a.trace)) {
        t_staged = __builtin_amdgcn_s_memrealtime();
        c_staged = __builtin_amdgcn_s_memtime();
      }
    }

    if (wave < group.n_chunks) {
      const Chunk chunk = a.chunks[group.chunk_begin + wave];
      // Steps of 4 positions, loaded two at a time: the vector-memory pipe costs ~17
      // cycles per load INSTRUCTION per CU whatever its width (tools/micro/ta_rate.hip),
      // so every load is 16 bytes per lane: the table values of a lane for two
      // consecutive steps, and the LDS offsets of its position in both steps.
      const int n_pairs = (chunk.q_end - chunk.q_begin) / 8;
      const int last = n_pairs > 0 ? n_pairs - 1 : 0;
      typedef double __attribute__((ext_vector_type(2))) double2v;
      typedef const __attribute__((address_space(1))) double2v* gl_f64x2;
      typedef int __attribute__((ext_vector_type(4))) int4v;
      typedef const __attribute__((address_space(1))) int4v* gl_i32x4;
      // A operand: k = lane / 16, i = lane % 4 inside the [u][k][i][step parity] block
      gl_f64x2 table =
          (gl_f64x2)((interp ? a.tables[k] : (const double*)a.table) +
                     ((int64_t)blockIdx.z * a.n_positions + chunk.q_begin) * RT) +
          (lane >> 4) * 4 + (lane & 3);
      gl_i32x4 pos_base = (gl_i32x4)a.pos_off + (chunk.q_begin / 8) * 4 + (lane >> 4);
      auto pos = [&](int pair) { return pos_base[(int64_t)pair * 4]; };
      // byte offsets of this lane's first draw (set 0) in LDS rows i and j
      const int lane_col = (lane & 15) * 8;
      const int base_j = lane_col - group.j_lo * (kLanes * 8);
      const int base_i =
          lane_col + (a.mode == 0 ? group.i_shift : n_rows) * (kLanes * 8);
      double scale[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        scale[s] = interp ? a.coef[(int64_t)k * a.ldb + col + 16 * s + (lane & 15)] : 1.0;

      auto load_table = [&](double2v (&t)[NT], int pair) {
#pragma unroll
        for (int u = 0; u < NT; ++u) t[u] = table[(int64_t)pair * 4 * RT + u * 16];
      };
      // densities of the step's position in this lane's four draws
      auto gather = [&](double (&ni)[4], double (&nj)[4], int off_i, int off_j) {
        const double* rj = (const double*)((const char*)lds + (off_j + base_j));
        const double* ri = (const double*)((const char*)lds + (off_i + base_i));
#pragma unroll
        for (int s = 0; s < 4; ++s) nj[s] = rj[16 * s];
#pragma unroll
        for (int s = 0; s < 4; ++s) ni[s] = ri[16 * s];
      };
      auto weights = [&](double (&w)[4], const double (&ni)[4], const double (&nj)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          w[s] = ni[s] * nj[s];
          if (interp) w[s] *= scale[s];
        }
      };
      double2v t[NT];
      double ni[4], nj[4], w[4];
      unsigned long long stamps[5] = {0, 0, 0, 0, 0};
      if (TC_TRACE(a.wave_trace)) stamps[0] = __builtin_amdgcn_s_memrealtime();
      if (n_pairs > 0) {
        load_table(t, 0);
        int4v pa = pos(0);
        gather(ni, nj, pa.x, pa.y);
        const int quarter = (n_pairs / 4) > 0 ? (n_pairs / 4) : 1;
        // One register set for the table values: each sub-tile's 16 bytes are reloaded for
        // the next pair of steps right after their last use; the other resident waves of
        // the SIMD cover the latency (the registers saved buy a fifth wave per SIMD).
        // Every prefetch is unconditional (clamped to the last pair).
        for (int pair = 0; pair < n_pairs; ++pair) {
          const int next = pair + 1 < last ? pair + 1 : last;
          const int4v pn = pos(next);
          weights(w, ni, nj);
          gather(ni, nj, pa.z, pa.w);
#pragma unroll
          for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[s][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[u].x, w[s], acc[s][u], 0, 0, 0);
          weights(w, ni, nj);
          gather(ni, nj, pn.x, pn.y);
#pragma unroll
          for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[s][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(t[u].y, w[s], acc[s][u], 0, 0, 0);
            t[u] = table[(int64_t)next * 4 * RT + u * 16];
          }
          pa = pn;
          if (TC_TRACE(a.wave_trace) && (pair + 1) % quarter == 0 && (pair + 1) / quarter <= 3)
            stamps[(pair + 1) / quarter] = __builtin_amdgcn_s_memrealtime();
#ifndef TC_DEVELOPER_KNOBS
          // Keep the iterations apart for the instruction scheduler: with the whole loop body
          // as one region it interleaves the next pair's loads differently and the kernel
          // takes 6.15 instead of 4.65 ms on BASELINE configs[4] (float64).  (The developer
          // build's time-stamp branch at this place had the same effect.)
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      if (TC_TRACE(a.wave_trace)) {
        stamps[4] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
          unsigned long long* rec =
              a.wave_trace + 6 * (((unsigned long long)tile * a.n_slabs + slab) * n_waves + wave);
          for (int q = 0; q < 5; ++q) rec[q] = stamps[q];
          rec[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        }
      }
    }
  }
  if (TC_TRACE(a.trace)) {
    t_main = __builtin_amdgcn_s_memrealtime();
    c_main = __builtin_amdgcn_s_memtime();
  }
  __syncthreads();  // the staged densities are dead; reuse LDS for the sums

  // deterministic tree reduction over the waves (any layout: all waves share it)
  int span = 1;
  while (span < n_waves) span <<= 1;
  for (int half = span >> 1; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      double* slot = lds + (wave - half) * RT * kLanes + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < NT; ++u) slot[(s * NT + u) * kLanes] = acc[s][u];
    }
    __syncthreads();
    if (wave < half && wave + half < n_waves) {
      const double* slot = lds + wave * RT * kLanes + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[s][u] += slot[(s * NT + u) * kLanes];
    }
    __syncthreads();
  }
  if (wave == 0) {
    // D layout: draw = 16 s + lane % 16, r = 4 u + lane / 16
    double* out = a.partial +
                  ((int64_t)slab * a.r_stride + (int64_t)blockIdx.z * RT + (lane >> 4)) *
                      a.ldb + col + (lane & 15);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int u = 0; u < NT; ++u) out[(int64_t)(4 * u) * a.ldb + 16 * s] = acc[s][u];
  }
  if (TC_TRACE(a.trace) && threadIdx.x == 0) {
    const unsigned long long block =
        tile + (unsigned long long)a.n_tiles * (slab + a.n_slabs * blockIdx.z);
    unsigned long long* rec = a.trace + 6 * block;
    rec[0] = t_start;
    rec[1] = t_staged;
    rec[2] = t_main;
    rec[3] = __builtin_amdgcn_s_memrealtime();
    rec[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    rec[5] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf) |
             ((c_main - c_staged) << 4);   // XCC_ID, shader cycles of the main loop
  }
}

// ---- contraction as a quadratic form (mode auto, float64) -----------------------------
//
//   sum_p c_p T[r][p] n_i n_j = sum_i n_i ( sum_{j <= i} c_ij T_r[i][j] n_j )
//
// (tabcorr.py:626-655 regrouped).  The inner sum is a matrix product over j whose second
// operand is the density row itself: no pair weights to form, no gathers.  One
// v_mfma_f64_16x16x4_f64 covers a 4 x 4 block of bin pairs (4 i x 4 j), 4 r values and 16
// draws; operand lanes (MI355X programming guide, f64 forms): A[m = l % 16][k = l / 16],
// B[k = l / 16][n = l % 16], D[m = l / 16 + 4 v][n = l % 16] in register v = 0..3.  With
// m = r_local + 4 i_local, k = j_local and n = draw:
//   A = c_ij T[r = 4 u + m % 4][i = i0 + m / 4][j = j0 + k]   one double per lane from the
//       re-laid-out table, two r sub-tiles u per 16-byte load;
//   B = n[j0 + l / 16][draws 2 (l % 16), 2 (l % 16) + 1]       one 16-byte load serves the
//       wave's two column sets (even / odd draws of its 32-draw tile);
//   D[u][set][v]: r = 4 u + l / 16, i = i0 + v, draw = 2 (l % 16) + set.
// After the last block of a block row: F[u][set] += sum_v D[u][set][v] n[i0 + v][draw]
// (40 FMAs per ~13 x 10 matrix instructions).  Per unit (4 x 4 bin block, all U sub-tiles,
// 32 draws): 2 U matrix instructions, (U + 1) / 2 + 1 loads, nothing else: every address
// is a buffer resource + constant lane offset + scalar offset, so the vector ALU (which
// shares the FP64 pipe with the matrix instructions) stays idle.  Reads past the end of a
// resource return zero: bins beyond the last one need no padding.
//
// No workgroup structure, no LDS, no barriers: the (draw tile, r tile, component, table,
// unit) space is cut into equal contiguous ranges, one per resident wave (hostmath.h:
// QuadSchedule), so every SIMD retires the same number of matrix instructions whatever
// the batch size; a wave writes its sums to a slab of the partial buffer whenever it
// leaves an output group, finalize_quad_kernel adds the slabs in fixed order.
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const double* f64_ptr;
typedef const __attribute__((address_space(4))) f64_ptr* sc_ptrs;   // array of pointers

constexpr unsigned kBufferFlags = 0x00020000;   // raw buffer, 32-bit data format (gfx9 family)

template <int IMM>
__device__ inline f64x2 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_offset,
                                      unsigned wave_offset) {
  return __builtin_bit_cast(
      f64x2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_offset + IMM, wave_offset, 0));
}

template <int U, bool INTERP>
__global__ __launch_bounds__(64 * kQuadWavesPerBlock, 2) void contract_quad_kernel(QuadArgs a) {
  constexpr int UP = (U + 1) / 2;
  constexpr bool interp = INTERP;
  extern __shared__ __attribute__((aligned(16))) double stage[];   // merge slots (4 U, 32) each
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = (int)blockIdx.x * kQuadWavesPerBlock + wave_in_block;
  set_priority(a.priority);
  sc_i32 wave_runs = (sc_i32)a.wave_runs;
  sc_i32 runs = (sc_i32)a.runs;       // QuadRun = 8 x int32
  sc_i32 comps = (sc_i32)a.comps;     // QuadCompArgs = 8 x int32
  // (a workgroup beyond the last wave has no runs but still meets the others at the barrier)
  // developer timeline: entry, first operands in, last matrix instruction issued, sums
  // written, end (100 MHz) and the shader clock cycles between the second and the third
  unsigned long long t_entry = 0, t_first = 0, t_main = 0, t_flushed = 0, c_first = 0, c_main = 0;
  if (TC_TRACE(a.stamps)) t_entry = __builtin_amdgcn_s_memrealtime();
  // (one 64-byte record per wave: its range of runs, its first run and that run's component;
  // three dependent scalar loads at the start of every wave otherwise)
  sc_i32 head = (sc_i32)a.wave_head + 16 * (wave < a.n_waves ? wave : 0);
  const bool have_head = a.wave_head != nullptr;
  const int run_begin = wave >= a.n_waves ? 0 : have_head ? head[0] : wave_runs[2 * wave];
  const int run_end = wave >= a.n_waves ? 0 : have_head ? head[1] : wave_runs[2 * wave + 1];
  const int c = lane & 15, kq = lane >> 4;
  const unsigned row_bytes = (unsigned)(a.ldb * 8);
  const unsigned off_a = lane * 16;                    // table: (unit, u pair, lane) x 16 B
  const unsigned off_e = c * 16;                       // densities of draws 2 c, 2 c + 1
  const unsigned off_b = kq * row_bytes + c * 16;      // ... of bin j0 + l / 16
  double F[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) F[u][0] = F[u][1] = 0.0;

  for (int ri = run_begin; ri < run_end; ++ri) {
    const bool first_run = have_head && ri == run_begin;
    sc_i32 run = first_run ? head + 2 : runs + ri * 8;
    const int tile = run[0], rtile = run[1], comp = run[2];
    const int table = run[3], rb0 = run[4], cb0 = run[5];
    const int count = run[6], slab = run[7];
    sc_i32 cmp = first_run ? head + 10 : comps + comp * 8;
    const bool triangular = cmp[0] != 0;
    const int i_bin0 = cmp[1], j_bin0 = cmp[2];
    const int n_cb = cmp[3];
    const unsigned unit_base = (unsigned)cmp[4];

    const double* densities =
        interp ? ((sc_ptrs)a.nbufs)[((sc_i32)a.table_class)[table]] : a.nbuf;
    const char* matrix = (const char*)(interp ? (const void*)((sc_ptrs)a.tables)[table]
                                              : a.table) + (int64_t)rtile * a.rtile_bytes;
    // the tile's 32 draws of every bin; rows past the last bin read as zero
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(densities + (int64_t)tile * kQuadTile), 0,
        (unsigned)a.n_bins * row_bytes - (unsigned)tile * (kQuadTile * 8), kBufferFlags);
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc((void*)matrix, 0, a.rtile_bytes, kBufferFlags);
    f64x2 cf = {1.0, 1.0};
    if (interp)   // spline weight / pair-weight norm of this table for the lane's two draws
      cf = *(const __attribute__((address_space(1))) f64x2*)(
          a.coef + (int64_t)table * a.ldb + (int64_t)tile * kQuadTile + 2 * c);

    int rb = rb0, cb = cb0, left = count;
    unsigned ua = (unit_base + (unsigned)(triangular ? rb * (rb + 1) / 2 + cb : rb * n_cb + cb)) *
                  (UP * 1024);
    f64x2 t0[UP], t1[UP], b0, b1;
    f64x4 D[U][2];
    // Operands of the unit at table offset `ua` and block column `col`.  Unconditional
    // (one basic block per phase keeps every prefetch where it is written); at worst a
    // wave reads one unit nobody uses.
    auto fetch = [&](f64x2 (&t)[UP], f64x2& b, int col) {
      t[0] = buffer_load16<0>(rs_t, off_a, ua);
      if (UP > 1) t[1] = buffer_load16<1024>(rs_t, off_a, ua);
      if (UP > 2) t[2] = buffer_load16<2048>(rs_t, off_a, ua);
      b = buffer_load16<0>(rs_n, off_b, (unsigned)(j_bin0 + 4 * col) * row_bytes);
      ua += UP * 1024;
    };
    auto mma = [&](const f64x2 (&t)[UP], const f64x2& b, bool first) {
      if (first) {
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double av = (u & 1) ? t[u >> 1].y : t[u >> 1].x;
          D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, zero, 0, 0, 0);
          D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, zero, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double av = (u & 1) ? t[u >> 1].y : t[u >> 1].x;
          D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, D[u][0], 0, 0, 0);
          D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, D[u][1], 0, 0, 0);
        }
      }
    };
    fetch(t0, b0, cb);
    if (TC_TRACE(a.stamps) && ri == run_begin) {
      // (waits for the first operands: reading them forces the loads to land)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_first = __builtin_amdgcn_s_memrealtime();
      c_first = __builtin_amdgcn_s_memtime();
    }
    while (left > 0) {
      // the units of one block row inside this run: n >= 1; on entry the first one is
      // in (t0, b0)
      const int row_length = triangular ? rb + 1 : n_cb;
      const int n = row_length - cb < left ? row_length - cb : left;
      left -= n;
      f64x2 e[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        e[v] = buffer_load16<0>(rs_n, off_e, (unsigned)(i_bin0 + 4 * rb + v) * row_bytes);
      // (the scheduling barriers keep the compiler from sinking a prefetch next to its
      // use: the loads of unit t + 1 are in flight while unit t runs on the matrix core)
      fetch(t1, b1, n > 1 ? cb + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(t0, b0, true);
      __builtin_amdgcn_sched_barrier(0);
      int t = 1;
      for (; t + 1 < n; t += 2) {
        fetch(t0, b0, cb + t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
        fetch(t1, b1, t + 2 < n ? cb + t + 2 : 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t0, b0, false);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t < n) {
        // one unit left, in (t1, b1); the next row's first unit goes to (t0, b0)
        fetch(t0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // the next row's first unit sits in (t1, b1)
#pragma unroll
        for (int p = 0; p < UP; ++p) t0[p] = t1[p];
        b0 = b1;
      }
      // the row's outer factor: F += D n_i (x the table's spline weight)
      if (interp) {
#pragma unroll
        for (int v = 0; v < 4; ++v) e[v] *= cf;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          F[u][0] = fma(D[u][0][v], e[v].x, F[u][0]);
          F[u][1] = fma(D[u][1][v], e[v].y, F[u][1]);
        }
      ++rb;
      cb = 0;
    }
    if (TC_TRACE(a.stamps)) {
      t_main = __builtin_amdgcn_s_memrealtime();
      c_main = __builtin_amdgcn_s_memtime();
    }
    if (slab != -1) {
      // r = 4 u + l / 16, draws 2 c and 2 c + 1 of the tile: to a slab of the partial buffer,
      // or to an LDS slot of the workgroup that is merged with its neighbours below
      double* out = slab >= 0
                        ? (double*)a.partial + ((int64_t)slab * (4 * U) + kq) * kQuadTile + 2 * c
                        : stage + ((-2 - slab) * (4 * U) + kq) * kQuadTile + 2 * c;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f64x2 value = {F[u][0], F[u][1]};
        *(f64x2*)(out + (4 * u) * kQuadTile) = value;
        F[u][0] = F[u][1] = 0.0;
      }
    }
  }
  // Workgroup-level merge (hostmath.h: QuadMergePlan): the waves of a workgroup mostly end
  // and start inside the same output group; their sums are added here, in slot order, and
  // leave as ONE slab -- a third of the partial-buffer traffic of one slab per wave.
  if (TC_TRACE(a.stamps)) t_flushed = __builtin_amdgcn_s_memrealtime();
  auto write_stamps = [&]() {
    if (TC_TRACE(a.stamps) && lane == 0 && wave < a.n_waves) {
      unsigned long long* out = a.stamps + (size_t)wave * 6;
      out[0] = t_entry;
      out[1] = t_first;
      out[2] = t_main;
      out[3] = t_flushed;
      out[4] = __builtin_amdgcn_s_memrealtime();
      out[5] = c_main - c_first;     // shader clock cycles of the main loop(s)
    }
  };
  sc_i32 merge_range = (sc_i32)a.merge_range;
  const int merge_begin = merge_range[2 * blockIdx.x], merge_end = merge_range[2 * blockIdx.x + 1];
  if (merge_begin == merge_end) {          // (uniform over the workgroup)
    write_stamps();
    return;
  }
  __syncthreads();
  sc_i32 merges = (sc_i32)a.merges;
  for (int e = merge_begin + wave_in_block; e < merge_end; e += kQuadWavesPerBlock) {
    const int slab = merges[4 * e], first = merges[4 * e + 1], count = merges[4 * e + 2];
    double* out = (double*)a.partial + ((int64_t)slab * (4 * U) + kq) * kQuadTile + 2 * c;
    const double* in = stage + (first * (4 * U) + kq) * kQuadTile + 2 * c;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f64x2 sum = *(const f64x2*)(in + (4 * u) * kQuadTile);
      for (int s = 1; s < count; ++s)
        sum += *(const f64x2*)(in + (s * (4 * U) + 4 * u) * kQuadTile);
      *(f64x2*)(out + (4 * u) * kQuadTile) = sum;
    }
  }
  write_stamps();
}

// ---- the same in float32 (BASELINE configs[4]: hundreds of r values, tolerance 1e-5) ------
//
// v_mfma_f32_16x16x4_f32 (probed: tools/micro/mfma_map_f32.hip): A[m = l % 16][k = l / 16],
// B[k = l / 16][n = l % 16], D[m = 4 (l / 16) + v][n = l % 16] in register v.  With
// m = i_local + 4 r_local, k = j_local, n = draw: register v is bin i0 + v and the lane group
// l / 16 the r value, so the outer factor is applied per register as in the float64 kernel.
// The matrix instruction is twice as fast as the float64 one, so a wave owns 64 draws (four
// column sets: draws 4 c + s of its tile, one 16-byte density load per lane serves all four)
// and an r tile holds 16 values (four sub-tiles in ONE 16-byte table load per lane): 16
// matrix instructions per two loads.  Densities arrive as floats (the occupation kernel
// writes a float copy), sums stay in float, slabs are (4 U, 64) floats.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int IMM>
__device__ inline f32x4 buffer_load16f(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_offset,
                                       unsigned wave_offset) {
  return __builtin_bit_cast(
      f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_offset + IMM, wave_offset, 0));
}

template <int U, bool INTERP>
__global__ __launch_bounds__(64 * kQuadWavesPerBlock, 2) void contract_quad_f32_kernel(QuadArgs a) {
  constexpr bool interp = INTERP;   // (the unit space gains a table dimension, as in float64)
  extern __shared__ __attribute__((aligned(16))) float stage32[];   // merge slots (4 U, 64)
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = (int)blockIdx.x * kQuadWavesPerBlock + wave_in_block;
  set_priority(a.priority);
  sc_i32 wave_runs = (sc_i32)a.wave_runs;
  sc_i32 runs = (sc_i32)a.runs;
  sc_i32 comps = (sc_i32)a.comps;
  const int run_begin = wave < a.n_waves ? wave_runs[2 * wave] : 0;
  const int run_end = wave < a.n_waves ? wave_runs[2 * wave + 1] : 0;
  const int c = lane & 15, kq = lane >> 4;
  const unsigned row_bytes = (unsigned)(a.ldb * 4);
  const unsigned off_a = lane * 16;                    // table: (unit, lane) x 16 B
  const unsigned off_e = c * 16;                       // densities of draws 4 c .. 4 c + 3
  const unsigned off_b = kq * row_bytes + c * 16;      // ... of bin j0 + l / 16
  float F[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) F[u][s] = 0.0f;

  for (int ri = run_begin; ri < run_end; ++ri) {
    const int tile = runs[ri * 8 + 0], rtile = runs[ri * 8 + 1], comp = runs[ri * 8 + 2];
    const int table = runs[ri * 8 + 3];
    const int rb0 = runs[ri * 8 + 4], cb0 = runs[ri * 8 + 5];
    const int count = runs[ri * 8 + 6], slab = runs[ri * 8 + 7];
    const bool triangular = comps[comp * 8 + 0] != 0;
    const int i_bin0 = comps[comp * 8 + 1], j_bin0 = comps[comp * 8 + 2];
    const int n_cb = comps[comp * 8 + 3];
    const unsigned unit_base = (unsigned)comps[comp * 8 + 4];
    typedef const float* f32_ptr;
    typedef const __attribute__((address_space(4))) f32_ptr* sc_f32_ptrs;
    const float* densities =
        interp ? ((sc_f32_ptrs)a.nbufs32)[((sc_i32)a.table_class)[table]] : a.nbuf32;
    const char* matrix = (const char*)(interp ? (const void*)((sc_ptrs)a.tables)[table]
                                              : a.table) + (int64_t)rtile * a.rtile_bytes;
    (void)table;
    f32x4 cf = {1.0f, 1.0f, 1.0f, 1.0f};
    if (interp) {   // spline weight / pair-weight norm of this table for the lane's four draws
      const double* w = a.coef + (int64_t)table * a.ldb + (int64_t)tile * kQuadTileF32 + 4 * c;
      cf = f32x4{(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
    }
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(densities + (int64_t)tile * kQuadTileF32), 0,
        (unsigned)a.n_bins * row_bytes - (unsigned)tile * (kQuadTileF32 * 4), kBufferFlags);
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc((void*)matrix, 0, a.rtile_bytes, kBufferFlags);

    int rb = rb0, cb = cb0, left = count;
    unsigned ua = (unit_base + (unsigned)(triangular ? rb * (rb + 1) / 2 + cb : rb * n_cb + cb)) *
                  1024u;
    f32x4 t0, t1, b0, b1;
    f32x4 D[U][4];
    auto fetch = [&](f32x4& t, f32x4& b, int col) {
      t = buffer_load16f<0>(rs_t, off_a, ua);
      b = buffer_load16f<0>(rs_n, off_b, (unsigned)(j_bin0 + 4 * col) * row_bytes);
      ua += 1024u;
    };
    auto mma = [&](const f32x4& t, const f32x4& b, bool first) {
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          D[u][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[u], b[s], first ? zero : D[u][s], 0, 0, 0);
    };
    fetch(t0, b0, cb);
    while (left > 0) {
      const int row_length = triangular ? rb + 1 : n_cb;
      const int n = row_length - cb < left ? row_length - cb : left;
      left -= n;
      f32x4 e[4];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        e[v] = buffer_load16f<0>(rs_n, off_e, (unsigned)(i_bin0 + 4 * rb + v) * row_bytes);
      fetch(t1, b1, n > 1 ? cb + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(t0, b0, true);
      __builtin_amdgcn_sched_barrier(0);
      int t = 1;
      for (; t + 1 < n; t += 2) {
        fetch(t0, b0, cb + t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
        fetch(t1, b1, t + 2 < n ? cb + t + 2 : 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t0, b0, false);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t < n) {
        fetch(t0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        t0 = t1;
        b0 = b1;
      }
      // the row's outer factor: F += D n_i (x the table's spline weight)
      if (interp) {
#pragma unroll
        for (int v = 0; v < 4; ++v) e[v] *= cf;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int v = 0; v < 4; ++v) F[u][s] = fmaf(D[u][s][v], e[v][s], F[u][s]);
      ++rb;
      cb = 0;
    }
    if (slab != -1) {
      // r = 4 u + l / 16, draws 4 c .. 4 c + 3 of the tile
      float* out = slab >= 0
                       ? (float*)a.partial + ((int64_t)slab * (4 * U) + kq) * kQuadTileF32 + 4 * c
                       : stage32 + ((-2 - slab) * (4 * U) + kq) * kQuadTileF32 + 4 * c;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f32x4 value = {F[u][0], F[u][1], F[u][2], F[u][3]};
        *(f32x4*)(out + (4 * u) * kQuadTileF32) = value;
#pragma unroll
        for (int s = 0; s < 4; ++s) F[u][s] = 0.0f;
      }
    }
  }
  sc_i32 merge_range = (sc_i32)a.merge_range;
  const int merge_begin = merge_range[2 * blockIdx.x], merge_end = merge_range[2 * blockIdx.x + 1];
  if (merge_begin == merge_end) return;          // (uniform over the workgroup)
  __syncthreads();
  sc_i32 merges = (sc_i32)a.merges;
  for (int e = merge_begin + wave_in_block; e < merge_end; e += kQuadWavesPerBlock) {
    const int slab = merges[4 * e], first = merges[4 * e + 1], n_slots = merges[4 * e + 2];
    float* out = (float*)a.partial + ((int64_t)slab * (4 * U) + kq) * kQuadTileF32 + 4 * c;
    const float* in = stage32 + (first * (4 * U) + kq) * kQuadTileF32 + 4 * c;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 sum = *(const f32x4*)(in + (4 * u) * kQuadTileF32);
      for (int s = 1; s < n_slots; ++s)
        sum += *(const f32x4*)(in + (s * (4 * U) + 4 * u) * kQuadTileF32);
      *(f32x4*)(out + (4 * u) * kQuadTileF32) = sum;
    }
  }
}

// Sums the slabs of every output group in slab order, normalises and writes the results
// in the reference's order: finalize_kernel for the partial layout of contract_quad_kernel.
// One block per 64 draws (two 32-draw tiles, each with its own slab lists).
template <typename P, int TILE>
__global__ __launch_bounds__(1024) void finalize_quad_kernel(FinalizeQuadArgs a) {
  __shared__ double tile[kFinalizeRows][kLanes + 1];
  __shared__ double part_sum[16][kLanes];
  __shared__ double norm_inv[kLanes];
  // fused likelihood: the data vector and the weight matrix, staged once per workgroup
  __shared__ double chi2_lds[kFinalizeRows * (kFinalizeRows + 1)];
  set_priority(a.priority);
  if (a.chi2 != nullptr) {
    const int count = a.n_r * (a.n_r + 1);
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x) chi2_lds[idx] = a.chi2_data[idx];
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = (int64_t)blockIdx.x * kLanes;
  const int64_t n_valid = a.n_draws - col < kLanes ? a.n_draws - col : kLanes;

  if (wave == 0 && a.ngal_part == nullptr) norm_inv[lane] = 1.0;
  if (wave == 0 && a.ngal_part != nullptr) {
    double n_cen = 0.0, n_sat = 0.0;
    for (int p = 0; p < a.n_ngal_parts; ++p) {
      n_cen += a.ngal_part[((int64_t)p * 2 + 0) * a.ldb + col + lane];
      n_sat += a.ngal_part[((int64_t)p * 2 + 1) * a.ldb + col + lane];
    }
    const double total = n_cen + n_sat;
    norm_inv[lane] = a.mode == 0 ? total * total : total;
    if (lane < n_valid && blockIdx.y == 0) {
      if (a.n_comp == 1) {
        a.ngal[col + lane] = total;
      } else {
        a.ngal[2 * (col + lane)] = n_cen;
        a.ngal[2 * (col + lane) + 1] = n_sat;
      }
    }
  }
  __syncthreads();
  const double norm = norm_inv[lane];
  // Separated by galaxy type the reference forms every term / sum first and masks
  // afterwards (tabcorr.py:653-681): with a non-finite sum some term is inf / inf = NaN
  // and NaN x False = NaN reaches every component.
  const bool poisoned = a.n_comp > 1 && !(fabs(norm) <= 1.79769313486231570815e308);
  auto normalise = [&](double sum) { return poisoned ? __builtin_nan("") : sum / norm; };

  const int n_rows = a.n_comp * a.n_r;
  const int n_waves = blockDim.x >> 6;
  const int rows_per_block = (n_rows + gridDim.y - 1) / gridDim.y;
  const int row_begin = blockIdx.y * rows_per_block;
  const int row_end = row_begin + rows_per_block < n_rows ? row_begin + rows_per_block : n_rows;
  // (a block serves 64 draws: two tiles of 32, or one of 64)
  const int64_t tile32 = TILE == 32 ? (int64_t)blockIdx.x * 2 + (lane >> 5) : (int64_t)blockIdx.x;
  const int64_t slab_stride = (int64_t)a.rt * TILE;
  // sum of part `part` of `parts` equal ranges of a row's slabs, eight independent loads
  // in flight, the additions in slab order
  auto sum_slabs = [&](int row, int part, int parts) {
    const int comp = row / a.n_r, r = row % a.n_r;
    const int rtile = r / a.r_per_tile, r_local = r % a.r_per_tile;
    const int64_t group = (tile32 * a.n_rtiles + rtile) * a.groups_per_rtile +
                          (a.groups_per_rtile > 1 ? comp : 0);
    const int first = a.group_begin[group], count = a.group_begin[group + 1] - first;
    const int begin = first + (int)((int64_t)count * part / parts);
    const int end = first + (int)((int64_t)count * (part + 1) / parts);
    const P* src = (const P*)a.partial + ((int64_t)begin * a.rt + r_local) * TILE + (lane & (TILE - 1));
    double sum = 0.0;
    for (int s0 = begin; s0 < end; s0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = s0 + u < end ? (double)src[(s0 - begin + u) * slab_stride] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    return sum;
  };
  for (int row0 = row_begin; row0 < row_end; row0 += kFinalizeRows) {
    const int rows = row_end - row0 < kFinalizeRows ? row_end - row0 : kFinalizeRows;
    if (rows * 2 <= n_waves) {
      // few rows (small batches, split over row blocks; an un-batched interpolator call
      // leaves ~1000 slabs per row): several waves per row, each over a contiguous range
      // of slabs, combined in fixed order
      const int parts = n_waves / rows;
      const int rr = wave / parts, part = wave % parts;
      if (rr < rows) part_sum[wave][lane] = sum_slabs(row0 + rr, part, parts);
      __syncthreads();
      if (wave < rows) {
        double sum = 0.0;
        for (int p = 0; p < parts; ++p) sum += part_sum[wave * parts + p][lane];
        tile[wave][lane] = normalise(sum);
      }
    } else {
      for (int rr = wave; rr < rows; rr += n_waves)
        tile[rr][lane] = normalise(sum_slabs(row0 + rr, 0, 1));
    }
    __syncthreads();
    if (a.chi2 != nullptr) {
      // (host side: every row of the draws is in `tile` now -- one pass, one row block)
      // wave w sums the rows i = w, w + n_waves, ... of delta_i (P delta)_i for its lane's
      // draw; the matrix from LDS (every lane reads the same word: a broadcast)
      const double* data = chi2_lds;
      const double* matrix = chi2_lds + rows;
      double part = 0.0;
      for (int i = wave; i < rows; i += n_waves) {
        double inner = 0.0;
        for (int j = 0; j < rows; ++j)
          inner = fma(matrix[i * rows + j], tile[j][lane] - data[j], inner);
        part = fma(tile[i][lane] - data[i], inner, part);
      }
      part_sum[wave][lane] = part;
      __syncthreads();
      if (wave == 0 && lane < n_valid) {
        double total = 0.0;
        for (int w = 0; w < n_waves; ++w) total += part_sum[w][lane];
        a.chi2[col + lane] = total;
      }
      __syncthreads();
      continue;
    }
    for (int idx = threadIdx.x; idx < rows * kLanes; idx += blockDim.x) {
      const int d = idx / rows, rr = idx % rows;
      if (d < n_valid) a.xi[(col + d) * (int64_t)n_rows + row0 + rr] = tile[rr][d];
    }
    __syncthreads();
  }
}

// ---- one launch per batch: theta -> (ngal, xi[, chi2]) inside a workgroup ------------------
//
// tabcorr.py:537-578 + :623-650 for 64 draws per workgroup, nothing handed over through
// memory.  The three-kernel path pays for its hand-offs twice: 8 MB of densities and ~5 MB of
// partial sums per 10^4 draws travel through the L2, and -- what costs the time -- every lane's
// chain has three stages that each wait for the previous one to DRAIN (a lane's cycle is
// occupation 75 us, contraction 68 us, finalisation 26 us with four lanes in flight:
// profiles/r03_notes.md).  Here a workgroup of kFusedWaves waves
//   1. evaluates the occupations of its 64 draws (lane = draw, bins strided over the waves,
//      occ_bin_zheng07 as in occ_zheng07_kernel) into an LDS array dens[bin][64],
//   2. contracts: wave w takes the 32-draw tile w / 4 and quarter w % 4 of the triangle's
//      units -- the loop of contract_quad_kernel with the density operands read from LDS (the
//      matrix still streams from the L2 through buffer loads), walked once per PAIR of r
//      sub-tiles: 32 accumulator registers instead of 80, so that four waves per SIMD fit
//      (two workgroups per CU: whatever phase the neighbour is in, a SIMD has two waves
//      issuing matrix instructions),
//   3. adds the four quarters through LDS, normalises and writes ngal, xi (or the likelihood).
// Workgroups of different launches share a CU, so one's occupation phase (vector ALU) runs
// under the other's matrix instructions; there is no inter-workgroup step.
// Mode auto, total correlation function or its three components, one r tile; the Zheng07
// family (with its Heaviside assembly bias / modulate_with_cenocc variants) or, LEAUTHAUD, the
// Leauthaud11 family.
// (W waves per workgroup: 8 = two 32-draw tiles x four parts of the units, two workgroups per
// CU; 16 = eight parts per tile, one workgroup with up to 160 KB of LDS per CU -- tables of
// more than 104 bins)
constexpr int fused_slot_doubles(int waves, int draws = 64) {
  // (the waves' sums: per wave 4 U rows of one 32-draw tile, or -- 40 draws per workgroup --
  // of all its draws)
  return waves * 4 * kQuadMaxU * (draws == 40 ? 40 : kQuadTile);
}
// 40 draws per workgroup (the latency form, below): where draw d of bin row `row` lies in the
// LDS density array -- the ten draws d = 4 dg + j of one j side by side, so that a lane of
// v_mfma_f64_4x4x4 (which wants dens[k][4 dg + j] for every dg) reads them as five 16-byte
// words; the sixteen (row % 4, j) starts 320 row + 80 j bytes fall into sixteen different
// 16-byte bank groups of the 256-byte LDS line: no conflicts.
__device__ __forceinline__ constexpr int dens40(int row, int draw) {
  return row * 40 + (draw & 3) * 10 + (draw >> 2);
}
constexpr int fused_scratch_doubles(int waves) { return fm::kTableDoubles + 2 * waves * kLanes; }
static_assert(kFusedWaves == 8 && kFusedMaxParts == 8, "8 or 16 waves: 4 or 8 parts per tile");
static_assert(20 * (kLanes + 1) + 20 * 21 <= fm::kTableDoubles,
              "results tile + likelihood data in the place of the math table");

// One pass over `count` units from block (rb, cb) on: UU (1 or 2) r sub-tiles whose table
// operands are the pair at lane offset off_a; F[uu][set] += ... as in contract_quad_kernel.
template <int UU, int DL>
__device__ __forceinline__ void fused_quad_pass(__amdgpu_buffer_rsrc_t rs_t, unsigned off_a,
                                                unsigned unit_bytes, const double* dens_b,
                                                const double* dens_e, int rb, int cb, int left,
                                                bool triangular, int n_cb, unsigned unit_base,
                                                double (&F)[2][2]) {
  unsigned ua = (unit_base + (unsigned)(triangular ? rb * (rb + 1) / 2 + cb : rb * n_cb + cb)) *
                unit_bytes;
  f64x2 t0, t1, b0, b1;
  f64x4 D[UU][2];
  auto fetch = [&](f64x2& t, f64x2& b, int column) {
    t = buffer_load16<0>(rs_t, off_a, ua);
    b = *(const f64x2*)(dens_b + 4 * column * DL);
    ua += unit_bytes;
  };
  auto mma = [&](const f64x2& t, const f64x2& b, bool first) {
    if (first) {
      const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const double av = u ? t.y : t.x;
        D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, zero, 0, 0, 0);
        D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, zero, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const double av = u ? t.y : t.x;
        D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, D[u][0], 0, 0, 0);
        D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, D[u][1], 0, 0, 0);
      }
    }
  };
  fetch(t0, b0, cb);
  while (left > 0) {
    const int row_length = triangular ? rb + 1 : n_cb;
    const int n = row_length - cb < left ? row_length - cb : left;
    left -= n;
    f64x2 e[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) e[v] = *(const f64x2*)(dens_e + (4 * rb + v) * DL);
    fetch(t1, b1, n > 1 ? cb + 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(t0, b0, true);
    __builtin_amdgcn_sched_barrier(0);
    int t = 1;
    for (; t + 1 < n; t += 2) {
      fetch(t0, b0, cb + t + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(t1, b1, false);
      __builtin_amdgcn_sched_barrier(0);
      fetch(t1, b1, t + 2 < n ? cb + t + 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(t0, b0, false);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t < n) {
      fetch(t0, b0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(t1, b1, false);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      t0 = t1;
      b0 = b1;
    }
#pragma unroll
    for (int u = 0; u < UU; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        F[u][0] = fma(D[u][0][v], e[v].x, F[u][0]);
        F[u][1] = fma(D[u][1][v], e[v].y, F[u][1]);
      }
    ++rb;
    cb = 0;
  }
}

// Inclusive prefix sum of `value` over the 64 lanes: inside the rows of 16 lanes by shifts, then
// across them by the two row broadcasts (data-parallel primitives: no LDS round trips).
__device__ __forceinline__ int wave_prefix_sum(int value) {
  auto shifted = [](int x, auto control, auto rows) {
    return __builtin_amdgcn_update_dpp(0, x, decltype(control)::value, decltype(rows)::value, 0xf,
                                       false);
  };
  typedef std::integral_constant<int, 0xf> all_rows;
  value += shifted(value, std::integral_constant<int, 0x111>(), all_rows());      // row_shr:1
  value += shifted(value, std::integral_constant<int, 0x112>(), all_rows());      // row_shr:2
  value += shifted(value, std::integral_constant<int, 0x114>(), all_rows());      // row_shr:4
  value += shifted(value, std::integral_constant<int, 0x118>(), all_rows());      // row_shr:8
  // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
  value += shifted(value, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>());
  value += shifted(value, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>());
  return value;
}

// D[i][j] += sum_k A[i][k] value(lane 16 k + 4 block + j) with A[i][k] = 1 for i = m, else 0
// (`select`: the lane's element of that A, 1 where lane % 4 == m): the four row shares of a (r,
// draw) of sum m, added into the lanes 16 m + ... -- four sums per chain of four instructions,
// every lane of the wave ends with one of them.
__device__ __forceinline__ double fused_quad_rows40(double select, double value, double sums) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(select, value, sums, 0, 0, 0);
}

// The latency form's pass (40 draws per workgroup, v_mfma_f64_4x4x4_4b_f64): `count` units from
// block (rb, cb) on for the NS r sub-tiles S0 .. S0 + NS - 1 and ALL 40 draws of the workgroup.
// The table's unit layout serves this instruction with the lanes permuted inside a group of 16:
// lane 16 k + 4 i + r of a unit holds T[r][row 4 rb + i][column 4 cb + k]; lane l = 16 k + 4 r +
// i of the WAVE fetches it (off_a: the caller's permutation) as A_block=r [i][k]; with B_block [k][j]
// = dens[4 cb + k][4 dg + j] (the same for every block) lane l receives D = sum_k T[r = (l / 4)
// % 4][row i = l / 16][k] dens[k][draw j = l % 4] (lane <-> element map: tools/micro/
// mfma_map.hip), which times dens[4 rb + i][4 dg + j] is the lane's share of F[r][draw]: the four
// i of a (r, j) -- lanes 16 apart, the k index of a B operand -- are added once, after the walk,
// on the matrix pipe as well (fused_quad_rows40: A = a row of ones).  F[s][dg] += ... per lane.
template <int S0, int NS>
__device__ __forceinline__ void fused_quad_pass40(__amdgpu_buffer_rsrc_t rs_t, unsigned off_a,
                                                  unsigned unit_bytes, const double* dens_b,
                                                  const double* dens_e, int rb, int cb, int left,
                                                  bool triangular, int n_cb, unsigned unit_base,
                                                  double (&F)[NS][10]) {
  unsigned ua = (unit_base + (unsigned)(triangular ? rb * (rb + 1) / 2 + cb : rb * n_cb + cb)) *
                unit_bytes;
  // (a unit holds the sub-tiles in pairs, 16 bytes per lane and pair: whole pairs come by one
  // 16-byte load, a single sub-tile of a pair by an 8-byte load of its half -- a 16-byte load
  // whose other half nobody reads invites the register allocator to reuse that half at once,
  // and the wave then waits for the load where it writes the register)
  double ta[2][NS];
  f64x2 tb[2][5];
  double D[NS][10];
  auto fetch = [&](int set, int column) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      constexpr int kPairBytes = 1024;
      const int sub = S0 + s;
      const bool first_of_pair = (sub & 1) == 0 && s + 1 < NS;
      const bool second_of_pair = (sub & 1) == 1 && s >= 1;
      if (first_of_pair) {
        const f64x2 word = __builtin_bit_cast(
            f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs_t, off_a + (sub / 2) * kPairBytes, ua,
                                                         0));
        ta[set][s] = word.x;
        ta[set][s + 1 < NS ? s + 1 : s] = word.y;
      } else if (!second_of_pair) {
        ta[set][s] = __builtin_bit_cast(
            double, __builtin_amdgcn_raw_buffer_load_b64(
                        rs_t, off_a + (sub / 2) * kPairBytes + (sub & 1) * 8, ua, 0));
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) tb[set][q] = *(const f64x2*)(dens_b + 4 * column * 40 + 2 * q);
    ua += unit_bytes;
  };
  // (the first unit of a block row starts the sums: C = 0 instead of 30 register clears)
  auto mma = [&](int set, bool first) {
    if (first) {
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double av = ta[set][s];
          D[s][2 * q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, tb[set][q].x, 0.0, 0, 0, 0);
          D[s][2 * q + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, tb[set][q].y, 0.0, 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double av = ta[set][s];
          D[s][2 * q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, tb[set][q].x, D[s][2 * q], 0, 0, 0);
          D[s][2 * q + 1] =
              __builtin_amdgcn_mfma_f64_4x4x4f64(av, tb[set][q].y, D[s][2 * q + 1], 0, 0, 0);
        }
    }
  };
  // (no scheduling barriers between the fetches and the matrix instructions: the compiler's own
  // interleaving is 0.4 us ahead of loads-then-products)
  fetch(0, cb);
  while (left > 0) {
    const int row_length = triangular ? rb + 1 : n_cb;
    const int n = row_length - cb < left ? row_length - cb : left;
    left -= n;
    // (the row's own densities, requested here: they have arrived long before the row ends)
    f64x2 e[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) e[q] = *(const f64x2*)(dens_e + 4 * rb * 40 + 2 * q);
    int t = 0;
    if (n > 1) {
      fetch(1, cb + 1);
      mma(0, true);
      fetch(0, 2 < n ? cb + 2 : 0);
      mma(1, false);
      t = 2;
    }
    for (; t + 1 < n; t += 2) {
      fetch(1, cb + t + 1);
      mma(0, false);
      fetch(0, t + 2 < n ? cb + t + 2 : 0);
      mma(1, false);
    }
    if (t < n) {
      fetch(1, 0);
      if (t == 0) mma(0, true); else mma(0, false);
#pragma unroll
      for (int s = 0; s < NS; ++s) ta[0][s] = ta[1][s];
#pragma unroll
      for (int q = 0; q < 5; ++q) tb[0][q] = tb[1][q];
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        F[s][2 * q] = fma(D[s][2 * q], e[q].x, F[s][2 * q]);
        F[s][2 * q + 1] = fma(D[s][2 * q + 1], e[q].y, F[s][2 * q + 1]);
      }
    ++rb;
    cb = 0;
  }
}

template <int NGAUSS, int U, bool ASSEMBIAS, bool MODULATE, bool LEAUTHAUD = false,
          int W = kFusedWaves, int DL = 64, bool GROUPED = false, int SATDEFER = 0>
__global__ __launch_bounds__(64 * W, DL == 40 ? 2 : W == 8 ? (SATDEFER == 2 ? 4 : 2) : 1) void predict_fused_kernel(
    FusedArgs a) {
  // SATDEFER (round 5; undecorated Zheng07, ten nodes, 64 draws): the satellites' expansion for
  // the lanes it serves, and the (bin, draw) pairs it does not serve evaluated after the
  // wave's bins, 64 pairs per pass of the node loop -- see "deferred pairs" below.  2: the
  // central bins likewise (records, the centrals no expansion serves deferred: the node loop of
  // the centrals leaves the loop of bins as well)
  static_assert(!SATDEFER || (NGAUSS == 10 && !ASSEMBIAS && !MODULATE && !LEAUTHAUD && DL != 32 &&
                              !GROUPED), "deferred pairs");
  // GROUPED: the waves stride over the groups of bins that share their nodes (occ_group_zheng07)
  static_assert(!GROUPED || (NGAUSS == 10 && !LEAUTHAUD), "groups of bins: Zheng07, ten nodes");
  // DL = draws per workgroup: 64 (two 32-draw tiles, W = 8 or 16 waves), or 32 (ONE tile, eight
  // waves = eight parts of the units, lanes = (draw, half of a bin's nodes) in the occupation
  // phase, occ_bin_zheng07_halves; two workgroups of up to 80 KB per CU): a quarter of the
  // 64-draw workgroup's lifetime -- batches below 8192 draws --, and tables of 105-208 bins
  // with the wave count per SIMD of the 64-draw form
  // DL = 40 (round 6, the LATENCY form): ONE workgroup per CU and 250 of them for 10^4 draws,
  // so that a launch that has the chip to itself fills it (64-draw workgroups: 157 of 256 CUs).
  // Phase 1 as for 64 draws with lanes 40 .. 63 idle; phase 2 on v_mfma_f64_4x4x4_4b_f64, whose
  // draw granularity is 4 instead of 16 (fused_quad_pass40): every wave takes an eighth of the
  // units for all 40 draws.  Undecorated Zheng07, ten nodes, total correlation function.
  static_assert((DL == 64 && (W == 8 || W == 16)) || ((DL == 32 || DL == 40) && W == 8),
                "workgroup shape");
  static_assert(DL == 64 || NGAUSS == 10 || LEAUTHAUD, "32 / 40 draws: ten nodes per bin");
  static_assert(DL != 40 || (!ASSEMBIAS && !MODULATE && !LEAUTHAUD && !GROUPED), "latency form");
  constexpr int PARTS = DL == 40 ? W : W * 32 / DL;       // waves per tile
  static_assert(!LEAUTHAUD || (NGAUSS == 0 && !ASSEMBIAS), "Leauthaud11: any n_gauss, undecorated");
  constexpr int UP = (U + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double fused_lds[];
  // region B: densities, later the waves' sums; region A: math table, later the results tile
  // and the likelihood's data | ngal sums
  constexpr int kSlotDoubles = fused_slot_doubles(W, DL);
  const int region_b = a.dens_rows * DL > kSlotDoubles ? a.dens_rows * DL : kSlotDoubles;
  double* dens = fused_lds;
  double* table = fused_lds + region_b;
  double(*red)[W][kLanes] = (double(*)[W][kLanes])(table + fm::kTableDoubles);
  const fm::Consts kc = fm::make_consts();
#ifdef TC_DEVELOPER_KNOBS
  // (diagnosis, TC_FUSED_SKIP: 1 no occupations -- every density 1e-3 --, 2 no matrix phase)
  const int skip = (a.priority >> 8) & 3;
  const int n_bins_occ = (skip & 1) ? 0 : a.n_bins;
  // (TC_FUSED_STAMPS: 32 slots of 100 MHz stamps per workgroup; wave 0 -- 0 entry, 1 math table staged,
  // 2 draws set up, 3 bins done, 4 sums exchanged, 5 matrix phase done, 6 waves' parts added,
  // 7 end; 8 all waves through the matrix phase, 9 this wave's part stored, 10 all parts stored
  // -- in the place of the likelihood's data; tools/r06_stamps.py)
  unsigned long long* const stamps =
      ((a.priority >> 10) & 1) && a.chi2 == nullptr ? (unsigned long long*)a.chi2_data : nullptr;
  // (waves 0 and 4 of eight, slots 16 + 2 wave ...: 0 the wave's bins done, 2 its deferred pairs
  // counted, 3 their list built, 4 their node loops done, 1 their values added to the sums)
  auto wave_stamp = [&](int which) {
    if (stamps != nullptr && (threadIdx.x & 63) == 0 && W == 8 && ((threadIdx.x >> 6) & 3) == 0)
      stamps[blockIdx.x * 32 + 16 + 8 * (threadIdx.x >> 8) + which] =
          __builtin_amdgcn_s_memrealtime();
  };
  auto stamp = [&](int which) {
    if (stamps != nullptr && threadIdx.x == 0)
      stamps[blockIdx.x * 32 + which] = __builtin_amdgcn_s_memrealtime();
  };
  if (skip & 1)
    for (int idx = threadIdx.x; idx < a.n_bins * DL; idx += blockDim.x) fused_lds[idx] = 1e-3;
#else
  constexpr int skip = 0;
  const int n_bins_occ = a.n_bins;
  auto stamp = [](int) {};
  auto wave_stamp = [](int) {};
#endif
  stamp(0);
  set_priority((a.priority >> 2) & 3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_gauss = NGAUSS > 0 ? NGAUSS : a.n_gauss;
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    const int hi = fm::kTableDoubles / 2;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    for (int i = threadIdx.x; i < hi; i += blockDim.x) dst[i] = src[i];
  }
  for (int idx = a.n_bins * DL + threadIdx.x; idx < a.dens_rows * DL; idx += blockDim.x)
    dens[idx] = 0.0;
  __syncthreads();
  stamp(1);

  // ---- 1. occupations ----
  const int draw = DL != 32 ? lane : (lane & 31);
  const int half = DL != 32 ? 0 : (lane >> 5);
  // (40 draws: lanes 40 .. 63 evaluate draws of the next workgroup -- or the batch's last --
  // and neither write nor defer anything)
  const bool live = DL != 40 || lane < 40;
  auto dens_at = [&](int row, int d) -> double& {
    return dens[DL == 40 ? dens40(row, d) : row * DL + d];
  };
  const int64_t col = (int64_t)blockIdx.x * DL;
  const int64_t b0 = col + draw;
  const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;
  double norm;
  {
    // (every wave sets up its lanes' draws itself: cheaper than a hand-over through LDS)
    const double* th = a.theta + b * a.n_theta;
    // (Leauthaud11: the draw's constants in the place of the Zheng07 ones, which are then
    // harmless finite numbers the compiler drops with the unused branch)
    LeauthaudDraw ld;
    if (LEAUTHAUD) ld = prepare_leauthaud(table, kc, th);
    const DrawSetup d =
        LEAUTHAUD ? prepare_draw(table, kc, 12.0, 1.0, 12.0, 13.0, 1.0, 0.0, 0.0)
                  : prepare_draw(table, kc, th[0], th[1], th[2], th[3], th[4],
                                 ASSEMBIAS ? th[5] : 0.0, ASSEMBIAS ? th[6] : 0.0);
    DrawParams dp;
    dp.log_m_min = d.log_m_min;
    dp.inv_sigma = d.inv_sigma;
    dp.m0 = d.m0;
    dp.log2_m1 = d.log2_m1;
    dp.sat_scale = d.sat_scale;
    dp.alpha = d.alpha;
    dp.a_cen = d.a_cen;
    dp.a_sat = d.a_sat;
    dp.bad = d.bad;
    dp.any_bad = __builtin_amdgcn_ballot_w64(dp.bad != 0) != 0;
    series_setup<MODULATE>(dp, (GROUPED ? a.group.series : a.series) != nullptr && DL != 32,
                           (GROUPED ? a.group.sat_series : a.sat_series) != nullptr && DL != 32);
    sc_f64 log_m = (sc_f64)a.log_m;
    sc_f64 mass = (sc_f64)a.m;
    sc_f64 weight = (sc_f64)a.weight;
    sc_f64 weight_sum = weight + a.n_bins * n_gauss;   // (get_quadrature: the bins' sums)
    sc_f64 n_h = (sc_f64)a.n_h;
    sc_f64 percentile = (sc_f64)a.percentile;
    const double f1 = (1.0 - a.split) / a.split, f2 = a.split / (1.0 - a.split);
    double sum_cen = 0.0, sum_sat = 0.0;
    stamp(2);
    if (GROUPED) {
      sc_i32 group_begin = (sc_i32)a.group.begin;
      sc_f64 n_h_m = (sc_f64)a.group.n_h;
      const GroupConsts gq{(sc_f64)a.group.log_m, (sc_f64)a.group.m, (sc_f64)a.group.weight,
                           (sc_f64)a.group.weight + a.n_bins * 10, (sc_f64)a.group.percentile,
                           (sc_i32)a.group.member,
                             SeriesConsts{(sc_f64)a.group.series, (sc_i32)a.group.series_thr,
                                          (sc_f64)a.group.sat_series, (sc_i32)a.group.sat_series_thr}};
      for (int gr = wave; gr < a.n_groups; gr += W) {
        const bool central = gr < a.n_central_groups;
        auto emit = [&](int mi, int g, double acc) {
          const double value = acc * n_h_m[mi];
          if (half == 0) dens[g * DL + draw] = value;
          if (central) sum_cen += value; else sum_sat += value;
        };
        if (DL == 32)
          occ_group_zheng07_halves<ASSEMBIAS, MODULATE>(
              table, kc, gr, group_begin[gr], group_begin[gr + 1], central, half, a.group.log_m,
              a.group.m, a.group.weight, gq, a.split, dp, emit);
        else
          occ_group_zheng07<ASSEMBIAS, MODULATE>(table, kc, gr, group_begin[gr],
                                                 group_begin[gr + 1], central, gq, a.split, dp,
                                                 emit);
      }
    }
    unsigned mine = 0;        // SATDEFER: bit j = this lane's draw defers the wave's j-th bin
    for (int g = wave; g < (GROUPED ? 0 : n_bins_occ); g += W) {
      const bool central = g < a.n_central;
      const bool above = ASSEMBIAS ? percentile[g] > a.split : false;
      bool deferred = false;
      const double acc =
          SATDEFER == 2 && central
              ? occ_cen_record(table, kc,
                               (sc_f64)a.cen_records + g * series::cen_record::kStride, dp,
                               &deferred) :
          SATDEFER && !central && (SATDEFER == 2 || a.sat_cap == series::sat::kShortest)
              ? occ_sat_record(table, kc,
                               (sc_f64)a.sat_records + g * series::sat_record::kStride, dp,
                               &deferred) :
          DL == 32 && LEAUTHAUD
              ? occ_bin_leauthaud11_halves<MODULATE>(table, kc, g, central, half, a.log_m, a.m,
                                                     a.weight, ld)
          : DL == 32  ? occ_bin_zheng07_halves<ASSEMBIAS, MODULATE>(table, kc, g, central, above,
                                                                    half, a.log_m, a.m, a.weight,
                                                                    log_m, weight_sum, dp)
          : LEAUTHAUD ? occ_bin_leauthaud11<MODULATE>(table, kc, g, n_gauss, central, log_m, mass,
                                                      weight, ld)
                      : occ_bin_zheng07<NGAUSS, ASSEMBIAS, MODULATE>(
                            table, kc, g, n_gauss, central, above, log_m, mass, weight,
                            weight_sum, dp, f1, f2,
                            SeriesConsts{(sc_f64)a.series, (sc_i32)a.series_thr, (sc_f64)a.sat_series,
                              (sc_i32)a.sat_series_thr,
                              SATDEFER ? a.sat_cap : series::sat::kSteps - 1},
                            SATDEFER ? &deferred : nullptr);
      if (SATDEFER && deferred && live) mine |= 1u << ((g - wave) / W);
      const double value = acc * n_h[g];
      if (half == 0 && live) dens_at(g, draw) = value;
      if (central) sum_cen += value; else sum_sat += value;
    }
    wave_stamp(0);
    if (SATDEFER) {
      // ---- deferred pairs: this wave's (bin, draw) pairs, 64 per pass (lane = pair) ----
      // A lane marks in `mine` the bins of its wave its draw needs the node loop for; the
      // pairs are numbered in (draw, bin) order -- a prefix sum of the lanes' counts, then every
      // lane enters its own pairs: a draw defers a bin or two of a wave's dozen, so that is a
      // loop of two or three steps where a loop over the wave's bins with a ballot each took 1.6
      // of this phase's 14 us for counting and numbering (tools/r06_stamps.py) --, entry p goes
      // to lane p of a pass: the draw's constants from the lane that holds the draw, the bin's
      // by vector loads, the result to the bin's row of the densities (this wave's rows: no
      // barrier), from where the draw's own lane adds it to its sum in bin order.  What a draw
      // defers depends on the draw alone: the same bits wherever it sits in the batch.
      unsigned* list = (unsigned*)&red[1][wave][0];
      const int count = __builtin_popcount(mine);
      const int inclusive = wave_prefix_sum(count);
      const int total = __builtin_amdgcn_readlane(inclusive, 63);
      wave_stamp(2);
      for (int base = 0; base < total; base += 64) {
        {
          unsigned left = mine;
          int position = inclusive - count - base;
          while (__builtin_amdgcn_ballot_w64(left != 0) != 0) {
            if (left != 0) {
              const int j = __builtin_ctz(left);
              left &= left - 1;
              if (position >= 0 && position < 64) list[position] = ((unsigned)j << 6) | (unsigned)lane;
              ++position;
            }
          }
        }
        wave_stamp(3);
        const int n = total - base < 64 ? total - base : 64;
        const bool active = lane < n;
        const unsigned entry = list[active ? lane : 0];
        const int g = wave + W * (int)(entry >> 6), from = (int)(entry & 63);
        const double m0 = __shfl(dp.m0, from, 64);
        const double log2_m1 = __shfl(dp.log2_m1, from, 64);
        const double sat_scale = __shfl(dp.sat_scale, from, 64);
        const double alpha = __shfl(dp.alpha, from, 64);
        const int bad = __shfl(dp.bad, from, 64);
        double acc = 0.0;
        const bool cen_pair = SATDEFER == 2 && g < a.n_central;
        if (SATDEFER == 2 && __builtin_amdgcn_ballot_w64(active && cen_pair) != 0) {
          // (pairs of centrals: draws no expansion serves -- a step-like sigma_logM, parameters
          // to fix up; occ_nodes_zheng07's arithmetic)
          const double log_m_min = __shfl(dp.log_m_min, from, 64);
          const double inv_sigma = __shfl(dp.inv_sigma, from, 64);
          double sum = 0.0;
          bool tie = false;
#pragma unroll 1
          for (int k0 = 0; k0 < 10; k0 += 5) {
            double node[5], w[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              node[k] = a.log_m[g * 10 + k0 + k];
              w[k] = a.weight[g * 10 + k0 + k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              sum = fma(w[k], fm::erf_fast(table, kc, (node[k] - log_m_min) * inv_sigma), sum);
              tie = tie || node[k] == log_m_min;
            }
          }
          sum = fma(0.5, sum, 0.5 * a.weight[a.n_bins * 10 + g]);
          if ((bad & kBadCen) || ((bad & kTieCen) && tie)) sum = __builtin_nan("");
          if (cen_pair) acc = sum;
        }
        // (five nodes at a time: their constants requested together, then evaluated)
        if (SATDEFER != 2 || __builtin_amdgcn_ballot_w64(active && !cen_pair) != 0) {
        double sat = 0.0;
#pragma unroll 1
        for (int k0 = 0; k0 < 10; k0 += 5) {
          double node[5], w[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            node[k] = a.m[g * 10 + k0 + k];
            w[k] = a.weight[g * 10 + k0 + k];
          }
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const double x = node[k] - m0;
            sat = fma(w[k],
                      fm::exp2_fast(table, kc,
                                    alpha * fm::log2_fast_offset(
                                                table, kc, x > 1e-300 ? x : 1e-300, log2_m1),
                                    x > 0.0),
                      sat);
          }
        }
        sat *= sat_scale;
        if (bad != 0) {           // (occ_nodes_zheng07's fix-ups of an undecorated satellite bin)
          if ((bad & kInfSat) && sat != 0.0) sat = __builtin_huge_val();
          if ((bad & kBadSat) && sat != 0.0) sat = __builtin_nan("");
        }
        if (!cen_pair) acc = sat;
        }
        if (active) dens_at(g, from) = acc * a.n_h[g];
        wave_stamp(4);
      }
      // (the draw's deferred bins, in bin order)
      for (unsigned left = mine; __builtin_amdgcn_ballot_w64(left != 0) != 0;) {
        if (left != 0) {
          const int j = __builtin_ctz(left);
          left &= left - 1;
          const double value = dens_at(wave + W * j, draw);
          if (SATDEFER == 2 && wave + W * j < a.n_central) sum_cen += value;
          else sum_sat += value;
        }
      }
    }
    wave_stamp(1);
    stamp(3);
    red[0][wave][lane] = sum_cen;
    red[1][wave][lane] = sum_sat;
    __syncthreads();
    stamp(4);
    double n_cen = 0.0, n_sat = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      n_cen += red[0][w][lane];
      n_sat += red[1][w][lane];
    }
    const double total = (skip & 1) ? 1.0 : n_cen + n_sat;
    norm = total * total;
    if (wave == 0 && half == 0 && live && b0 < a.n_draws) {
      if (a.separate) {
        a.ngal[2 * b0] = n_cen;
        a.ngal[2 * b0 + 1] = n_sat;
      } else {
        a.ngal[b0] = total;
      }
    }
  }

  // ---- 2. quadratic form ----
  set_priority(a.priority & 3);
  const int c = lane & 15, kq = lane >> 4;
  double F[UP][2][2];
  // (40 draws: three, then the other r sub-tiles -- accumulators and sums of up to 3 x 10
  // (sub-tile, four draws) pairs per lane in either pass)
  constexpr int UA = DL == 40 ? (U > 3 ? 3 : U) : 1, UB = DL == 40 && U > 3 ? U - 3 : 1;
  double F40a[UA][10], F40b[UB][10];
  if (DL == 40) {
    const int part = wave;
    // (the unit's lane 16 k + 4 i + r for the wave's lane 16 k + 4 r + i: fused_quad_pass40)
    const unsigned off_a = ((lane & 0x30) | ((lane & 3) << 2) | ((lane >> 2) & 3)) * 16;
    const double* dens_b = dens + (a.part_j_row0[part] + (lane >> 4)) * 40 + (lane & 3) * 10;
    const double* dens_e = dens + (a.part_i_row0[part] + (lane >> 4)) * 40 + (lane & 3) * 10;
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.table, 0, a.table_bytes, kBufferFlags);
    const int rb = a.part_rb0[part], cb = a.part_cb0[part], count = a.part_count[part];
    const bool triangular = a.part_triangular[part] != 0;
    const int n_cb = a.part_n_cb[part];
    const unsigned unit_base = (unsigned)a.part_unit_base[part];
#pragma unroll
    for (int s = 0; s < UA; ++s)
#pragma unroll
      for (int g = 0; g < 10; ++g) F40a[s][g] = 0.0;
#pragma unroll
    for (int s = 0; s < UB; ++s)
#pragma unroll
      for (int g = 0; g < 10; ++g) F40b[s][g] = 0.0;
#ifdef TC_DEVELOPER_KNOBS
    // (TC_FUSED_SAME_UNIT: every unit's matrix operands from the table's first unit -- wrong
    // results, loads that hit the first-level cache: what the operands' way from the L2 costs)
    const unsigned unit_bytes = ((a.priority >> 12) & 1) ? 0u : UP * 1024u;
    const unsigned first_unit = ((a.priority >> 12) & 1) ? 0u : unit_base;
#else
    constexpr unsigned unit_bytes = UP * 1024u;
    const unsigned first_unit = unit_base;
#endif
    if (!(skip & 2))
    fused_quad_pass40<0, UA>(rs_t, off_a, unit_bytes, dens_b, dens_e, rb, cb, count, triangular,
                             n_cb, first_unit, F40a);
    if (U > 3 && !(skip & 2))
      fused_quad_pass40<(U > 3 ? 3 : 0), UB>(rs_t, off_a, unit_bytes, dens_b, dens_e, rb, cb,
                                             count, triangular, n_cb, first_unit, F40b);
  } else {
    const int sub = wave / PARTS, part = wave % PARTS;
    const unsigned off_a = lane * 16;
    // (the component's columns are density rows j_row0 ..., its rows i_row0 ...)
    const double* dens_b =
        dens + (a.part_j_row0[part] + kq) * DL + sub * kQuadTile + 2 * c;   // + 4 col rows
    const double* dens_e = dens + a.part_i_row0[part] * DL + sub * kQuadTile + 2 * c;
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.table, 0, a.table_bytes, kBufferFlags);
    const int rb = a.part_rb0[part], cb = a.part_cb0[part], count = a.part_count[part];
    const bool triangular = a.part_triangular[part] != 0;
    const int n_cb = a.part_n_cb[part];
    const unsigned unit_base = (unsigned)a.part_unit_base[part];
#pragma unroll
    for (int p = 0; p < UP; ++p) {
      F[p][0][0] = F[p][0][1] = F[p][1][0] = F[p][1][1] = 0.0;
      if (skip & 2) continue;
      if (2 * p + 1 < U)
        fused_quad_pass<2, DL>(rs_t, off_a + p * 1024, UP * 1024, dens_b, dens_e, rb, cb, count,
                               triangular, n_cb, unit_base, F[p]);
      else
        fused_quad_pass<1, DL>(rs_t, off_a + p * 1024, UP * 1024, dens_b, dens_e, rb, cb, count,
                               triangular, n_cb, unit_base, F[p]);
    }
  }
  stamp(5);
  __syncthreads();       // the densities are dead: their place takes the waves' sums
  stamp(8);
  if (DL == 40) {
    // lane l holds the share of row i = l / 16 of F[r = 4 s + (l / 4) % 4][draw 4 g + l % 4]:
    // the four rows are added on the matrix pipe, four sums of one r sub-tile per chain of four
    // instructions (fused_quad_rows40), lanes 16 q + ... receiving sum q of the chain -- 50
    // instructions of 16 cycles and 15 stores by all lanes.  (Rotations across the lanes and
    // vector additions, 300 instructions at 4 - 8 cycles with two waves per SIMD, and 25 stores by
    // a quarter of the lanes were 2.4 of the workgroup's 52 us, tools/r06_stamps.py.)  A chain
    // multiplies the other three sums' shares by 0: only where every share is a finite number
    // -- the matrix has no other entries, none beyond 1e20 in size (a.priority, bit 11), and
    // every draw of the workgroup a pair-weight sum below 1e280: no share can overflow.
    // Elsewhere every sum by an instruction of its own (A = 1, the same additions in the same
    // order: the same bits), stored by the lanes of row 0.
    // The wave's sums as (4 U, 40).
    const int q = lane >> 4;
    auto share = [&](int sub, int g) {
      return sub < UA ? F40a[sub < UA ? sub : 0][g] : F40b[sub >= UA ? sub - UA : 0][g];
    };
    double* out = dens + (wave * (4 * U) + ((lane >> 2) & 3)) * 40 + (lane & 3);
    if (((a.priority >> 11) & 1) && __builtin_amdgcn_ballot_w64(!(norm < 1e280)) == 0) {
      double select[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) select[m] = (lane & 3) == m ? 1.0 : 0.0;
      // draws 4 g + j, g = 4 t + q: t = 0, 1, and g = 8 + q for the lanes of q = 0, 1
#pragma unroll
      for (int sub = 0; sub < U; ++sub)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          double sums = 0.0;
#pragma unroll
          for (int m = 0; m < (t < 2 ? 4 : 2); ++m)
            sums = fused_quad_rows40(select[m], share(sub, 4 * t + m), sums);
          if (t < 2 || q < 2) out[(4 * sub) * 40 + 16 * t + 4 * q] = sums;
        }
    } else {
#pragma unroll
      for (int sub = 0; sub < U; ++sub)
#pragma unroll
        for (int g = 0; g < 10; ++g) {
          const double sum = fused_quad_rows40(1.0, share(sub, g), 0.0);
          if (q == 0) out[(4 * sub) * 40 + 4 * g] = sum;
        }
    }
  } else {
    // r = 4 u + l / 16, draws 2 c and 2 c + 1 of the wave's tile
    double* out = dens + (wave * (4 * U) + kq) * kQuadTile + 2 * c;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f64x2 value = {F[u >> 1][u & 1][0], F[u >> 1][u & 1][1]};
      *(f64x2*)(out + (4 * u) * kQuadTile) = value;
    }
  }
  // ---- 3. the four quarters of every tile, normalisation, results ----
  stamp(9);
  set_priority((a.priority >> 4) & 3);
  double(*tile)[kLanes + 1] = (double(*)[kLanes + 1])table;
  if (a.separate) {
    // cen-cen = the first quarter of the tile's waves, cen-sat = the two middle quarters,
    // sat-sat = the last quarter, one component after the other through the results tile.  The reference forms every term / sum first
    // and masks afterwards (tabcorr.py:653-681): with a non-finite pair-weight sum some term
    // is inf / inf = NaN and NaN x False = NaN reaches every component.
    const bool poisoned = !(fabs(norm) <= 1.79769313486231570815e308);
    const int64_t n_valid = a.n_draws - col < DL ? a.n_draws - col : DL;
    const int sub = lane >> 5, d = lane & 31;
    for (int comp = 0; comp < 3; ++comp) {
      __syncthreads();
      constexpr int Q = PARTS / 4;
      const int p_begin = comp == 0 ? 0 : comp == 1 ? Q : 3 * Q;
      const int p_count = comp == 1 ? 2 * Q : Q;
      for (int rr = wave; rr < a.n_r && lane < DL; rr += W) {
        const double* first = dens + ((PARTS * sub + p_begin) * (4 * U) + rr) * kQuadTile + d;
        double sum = first[0];
        for (int p = 1; p < p_count; ++p) sum += first[p * (4 * U) * kQuadTile];
        tile[rr][lane] = poisoned ? __builtin_nan("") : sum / norm;
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < a.n_r * DL; idx += blockDim.x) {
        const int dd = idx / a.n_r, rr = idx % a.n_r;
        if (dd < n_valid) a.xi[((col + dd) * 3 + comp) * (int64_t)a.n_r + rr] = tile[rr][dd];
      }
    }
    return;
  }
  double* chi2_lds = table + 20 * (kLanes + 1);
  if (a.chi2 != nullptr) {
    const int count = a.n_r * (a.n_r + 1);
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x) chi2_lds[idx] = a.chi2_data[idx];
  }
  __syncthreads();
  stamp(10);
  if (DL == 40) {
    for (int rr = wave; rr < a.n_r && lane < DL; rr += W) {
      const double* first = dens + rr * 40 + lane;
      double sum = first[0];
#pragma unroll
      for (int part = 1; part < W; ++part) sum += first[part * (4 * U) * 40];
      tile[rr][lane] = sum / norm;
    }
  } else {
    const int sub = lane >> 5, d = lane & 31;
    for (int rr = wave; rr < a.n_r && lane < DL; rr += W) {
      const double* first = dens + ((PARTS * sub) * (4 * U) + rr) * kQuadTile + d;
      double sum = first[0];
#pragma unroll
      for (int part = 1; part < PARTS; ++part) sum += first[part * (4 * U) * kQuadTile];
      tile[rr][lane] = sum / norm;
    }
  }
  __syncthreads();
  stamp(6);
  const int64_t n_valid = a.n_draws - col < DL ? a.n_draws - col : DL;
  if (a.chi2 != nullptr) {
    // chi2 = delta^T P delta for the lane's draw (finalize_quad_kernel's fused likelihood)
    const int rows = a.n_r;
    const double* data = chi2_lds;
    const double* matrix = chi2_lds + rows;
    double part = 0.0;
    for (int i = wave; i < rows; i += W) {
      double inner = 0.0;
      for (int j = 0; j < rows; ++j)
        inner = fma(matrix[i * rows + j], tile[j][lane] - data[j], inner);
      part = fma(tile[i][lane] - data[i], inner, part);
    }
    red[0][wave][lane] = part;
    __syncthreads();
    if (wave == 0 && lane < n_valid) {
      double total = 0.0;
      for (int w = 0; w < W; ++w) total += red[0][w][lane];
      a.chi2[col + lane] = total;
    }
    return;
  }
  for (int idx = threadIdx.x; idx < a.n_r * DL; idx += blockDim.x) {
    const int d = idx / a.n_r, rr = idx % a.n_r;
    if (d < n_valid) a.xi[(col + d) * (int64_t)a.n_r + rr] = tile[rr][d];
  }
  stamp(7);
}

// The deferred (group, draw) pairs of the mode-cross kernels (round 5): lanes that neither a
// shortcut nor an expansion served are marked in `bitmap` (one word per group: MarkPairs) and
// entered the row sums as 0.  Wave w owns the draws w, w + 8, ...: it collects their pairs of
// the groups [g_first, g_last) from the bitmap in group order (64 groups at a time: a prefix
// sum over the lanes, into `list`: 512 words of this wave), evaluates 64 pairs per pass of the
// node loop (lane = pair: the draw's constants from the lane that holds the draw, the group's
// by vector loads, five nodes at a time) and adds coefficient x occupation to ITS draws' sums
// in `res` ((rows, 64) doubles) -- lane = (row, share of the eight draws), eight pairs'
// coefficients requested together --, in that order: a fixed order per draw, whatever the
// rest of the batch is.  `type`: -1 every pair, 0 / 1 only the pairs of centrals / satellites
// (separated by galaxy type: one call per component).  coefficient(bin, row): the row's
// coefficient of member bin `bin`.  ROW_LANES: 16, 32 or 64 rows (lanes beyond: further draws).
template <int ROW_LANES, typename Coefficient>
__device__ __forceinline__ void cross_deferred_pairs(
    const CrossFusedArgs& a, const double* table, const fm::Consts& kc, const DrawParams& dp,
    const unsigned long long* bitmap, unsigned* list, int wave, int lane, int g_first, int g_last,
    int type, Coefficient&& coefficient, double* res) {
  constexpr int SHARES = 64 / ROW_LANES;        // lanes per row: shares of the wave's 8 draws
  constexpr int SLOTS = 8 / SHARES;             // draws per lane: j = share + SHARES slot
  const int row = lane % ROW_LANES, share = lane / ROW_LANES;
  const unsigned long long owner = 0x0101010101010101ull << wave;
  double extra[SLOTS];
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) extra[slot] = 0.0;
  // The pairs of 64 groups at a time go to the list; it is evaluated when the next 64 groups'
  // pairs might not fit (512 entries: 64 groups x the wave's 8 draws) and at the end -- a draw
  // defers a handful of groups, so a wave's pairs of ALL groups are usually one pass of the
  // node loop (evaluated per block of 64 groups they were ~5 nearly empty passes: a sixth of the
  // kernel's vector instructions on the AbacusSummit tables).
  auto evaluate = [&](int total) {
#ifdef TC_DEVELOPER_KNOBS
    // (TC_FUSED_STAMPS, slot 12: the workgroup's deferred pairs; slot 13: passes of 64)
    if (((a.priority >> 10) & 1) && a.chi2 == nullptr && lane == 0) {
      atomicAdd((unsigned long long*)a.chi2_data + blockIdx.x * 16 + 12, (unsigned long long)total);
      atomicAdd((unsigned long long*)a.chi2_data + blockIdx.x * 16 + 13,
                (unsigned long long)((total + 63) / 64));
    }
#endif
    for (int e0 = 0; e0 < total; e0 += 64) {
      const int n = total - e0 < 64 ? total - e0 : 64;
      const bool active = lane < n;
      const unsigned entry = list[e0 + (active ? lane : 0)];
      const int g = (int)(entry >> 6), draw = (int)(entry & 63);
      const bool central = g < a.n_central_groups;
      const int m_begin = a.group.begin[g], m_end = a.group.begin[g + 1];
      const double log_m_min = __shfl(dp.log_m_min, draw, 64);
      const double inv_sigma = __shfl(dp.inv_sigma, draw, 64);
      const double m0 = __shfl(dp.m0, draw, 64);
      const double log2_m1 = __shfl(dp.log2_m1, draw, 64);
      const double sat_scale = __shfl(dp.sat_scale, draw, 64);
      const double alpha = __shfl(dp.alpha, draw, 64);
      const int bad = __shfl(dp.bad, draw, 64);
      // member by member: the occupation (the nodes of the lane's group for the lane's draw,
      // evaluated per member -- a handful of pairs per tile, and twenty registers less than
      // keeping the node values), then its row contributions to the owner's sums
      for (int t = 0; __builtin_amdgcn_ballot_w64(active && m_begin + t < m_end) != 0; ++t) {
        const int mi = m_begin + t;
        const bool valid = active && mi < m_end && (type < 0 || central == (type == 0));
        const int mi_safe = valid ? mi : m_begin;
        const double* weight = a.group.weight + (int64_t)mi_safe * 10;
        double acc = 0.0;
        bool tie = false;
        // (five nodes at a time: their constants requested together, then evaluated)
#pragma unroll 1
        for (int k0 = 0; k0 < 10; k0 += 5) {
          double node[5], w[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            node[k] = central ? a.group.log_m[g * 10 + k0 + k] : a.group.m[g * 10 + k0 + k];
            w[k] = weight[k0 + k];
          }
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            double value;
            if (central) {
              value = fm::erf_fast(table, kc, (node[k] - log_m_min) * inv_sigma);
              tie = tie || node[k] == log_m_min;
            } else {
              const double x = node[k] - m0;
              value = fm::exp2_fast(
                  table, kc,
                  alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, log2_m1),
                  x > 0.0);
            }
            acc = fma(w[k], value, acc);
          }
        }
        const bool cen_nan = (bad & kBadCen) || ((bad & kTieCen) && tie);
        if (central) acc = fma(0.5, acc, 0.5 * a.group.weight[(int64_t)a.n_bins * 10 + mi_safe]);
        else acc *= sat_scale;
        if (bad != 0) {
          if (!central && (bad & kInfSat) && acc != 0.0) acc = __builtin_huge_val();
          if (central ? cen_nan : ((bad & kBadSat) && acc != 0.0)) acc = __builtin_nan("");
        }
        // (eight pairs at a time: their coefficients requested together; a pair that is not
        // valid -- beyond the batch, beyond its group's members, the other galaxy type --
        // adds coefficient x 0)
        // The pair's draw is j = share + SHARES slot of the wave's eight: its share of the lanes
        // adds to the sums of that slot -- a wave-uniform choice, one product per pair where a
        // comparison, a selection and a product per SLOT stood until round 6.
        const double nbar_lane = valid ? acc : 0.0;
        for (int p0 = 0; p0 < n; p0 += 8) {
          double c[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            c[u] = coefficient(__builtin_amdgcn_readlane(mi_safe, (p0 + u) & 63), row);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (p0 + u >= n) break;
            const double nbar = readlane_f64(nbar_lane, (p0 + u) & 63);
            const int j = __builtin_amdgcn_readlane(draw, (p0 + u) & 63) >> 3;
            const double mine = share == j % SHARES ? nbar : 0.0;
            const int slot_of_pair = j / SHARES;
            // (a switch over constants: written as a loop with a comparison the compiler turns
            // it into an indexed access of `extra` -- in scratch memory)
            switch (slot_of_pair) {
#define TC_SLOT(K)                                                     \
  case K:                                                              \
    if constexpr (K < SLOTS) extra[K < SLOTS ? K : 0] = fma(c[u], mine, extra[K < SLOTS ? K : 0]); \
    break;
              TC_SLOT(0) TC_SLOT(1) TC_SLOT(2) TC_SLOT(3) TC_SLOT(4) TC_SLOT(5) TC_SLOT(6) TC_SLOT(7)
#undef TC_SLOT
              default: break;
            }
          }
        }
      }
    }
  };
  // (segments of blocks that fit the list; the evaluation stands in ONE place, outside the
  // loop of blocks: inlined twice, or with a block's words alive across it, registers spill)
  int gb = g_first;
  while (gb < g_last) {
    int filled = 0;
    for (; gb < g_last; gb += 64) {
      const int g_lane = gb + lane;
      unsigned long long word = g_lane < g_last ? bitmap[g_lane] & owner : 0ull;
      // (most blocks of 64 groups hold no pair of this wave's draws)
      if (__builtin_amdgcn_ballot_w64(word != 0) == 0) continue;
      const int count = __builtin_popcountll(word);
      const int inclusive = wave_prefix_sum(count);
      const int total = __builtin_amdgcn_readlane(inclusive, 63);
      if (filled + total > 512) break;            // (the next segment starts with this block)
      int slot_out = filled + inclusive - count;
      while (word != 0) {
        list[slot_out++] = (unsigned)(g_lane << 6) | (unsigned)__builtin_ctzll(word);
        word &= word - 1;
      }
      filled += total;
    }
    evaluate(filled);
  }
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot)
    res[row * kLanes + wave + 8 * (share + SHARES * slot)] += extra[slot];
}

// ---- mode cross, one launch per batch: theta -> (ngal, xi[, chi2]) ------------------------
//
// tabcorr.py:537-578 + :623-683 for tables with mode = 'cross' (excess surface density, any
// galaxy-matter statistic: tpcf_matrix has one column per halo bin, xi = T n / sum n), and
// interpolator.py:124-216 over K such tables whose bins share their mass edges.  Per draw the
// contraction is 2 K R G flop against G x n_gauss occupation nodes -- the step is ALL vector
// arithmetic, and the three-kernel path spends as much time moving the G x draws density array
// through memory (AbacusSummit fixture, G = 1104: 87 MB written + 89 MB read per 10^4 draws) as
// computing.  Here a workgroup of eight waves owns 64 draws (lane = draw) and walks the bins in
// chunks of whole groups (<= 32 bins):
//   A. the waves stride over the chunk's groups (occ_group_zheng07: nodes once per group) and
//      put every member's mean occupation N into an LDS buffer (bin in chunk, draw);
//   B. wave w owns rows w RW .. w RW + RW - 1 of the K (R + 1) row sums
//          acc[k (R + 1) + r] += N T_k[r][g] n_h,k[g]      (r < R),
//          acc[k (R + 1) + R] += N n_h,k[g]                (the number density of table k)
//      and adds every bin of the chunk: one LDS read (conflict-free: a row per bin) and RW FMAs
//      with the coefficients as scalar operands ((n_bins in group order, ROWS) doubles, rows
//      beyond K (R + 1) zero).
// Two buffers, so that one barrier per chunk is enough: a wave that writes chunk c + 1 has
// passed the barrier of chunk c, which every wave reaches only after reading chunk c - 1.
// Nothing but theta comes in and nothing but the results go out.  Then per
// draw xi[r] = sum_k c_k acc[k][r] / ngal_k with the spline weights c_k of the draw's extra
// parameters (interpolator.py:275-331; one table: c = 1).  Separated by galaxy type the sums
// of the centrals are set aside after their last chunk (no chunk holds both types).
//
// B on the matrix pipe (round 5).  acc[row][draw] += sum_bin coefficient[bin][row] N[bin][draw]
// is a (ROWS x bins) x (bins x 64 draws) product: one v_mfma_f64_16x16x4_f64 takes 16 rows x 4
// bins (A: one double per lane from the rows re-laid out per (4-bin block, 16-row block), one
// coalesced 512-byte load) and 4 bins x 16 draws (B: one ds_read_b64 per lane from the chunk
// buffer).  The ROWS / 16 x 4 output tiles are dealt to the eight waves (RW / 4 each: 1, 2 or
// 4; two tiles of a wave share their A operand).  Round 4 issued per bin and wave one LDS read,
// one 64-byte SCALAR load of coefficients (from a 565 KB table the 16 KB scalar cache cannot
// hold) and RW FMAs, and spilled 61 scalar registers around the unrolled loads: 79.4 us per 10^4
// draws of the reference's AbacusSummit interpolator against 45 us of vector issue.  The same
// flop on the same pipe (matrix and vector FP64 share it), an eighth of the instructions, no
// scalar loads.  The buffer's rows are skewed by 16 draws per bin (row r holds draw d at column
// (d + 16 (r % 4)) % 64) so that the four bins a B operand spans fall into different banks.
template <int RW, bool ASSEMBIAS, bool MODULATE, bool DEFER = false>
// (second launch bound = waves per SIMD: four -- two workgroups per CU, 128 vector registers --
// for up to 8 rows per wave; the 16-row instances (65 - 128 rows: 32 registers of sums live across
// the occupation arithmetic) would spill there and take one workgroup per CU instead)
__global__ __launch_bounds__(64 * kCrossWaves, RW <= 8 ? 4 : 2) void predict_cross_fused_kernel(
    CrossFusedArgs a) {
  constexpr int W = kCrossWaves, ROWS = W * RW;
  constexpr int NT = ROWS / 16;                       // row blocks of 16
  constexpr int N_RB = RW >= 16 ? 2 : 1;              // row blocks per wave
  constexpr int N_DB = RW >= 8 ? 2 : 1;               // blocks of 16 draws per wave
  constexpr int PF = 6;                               // A operands in flight (4-bin steps; 4: +0.3–1 %, 8: level)
  static_assert(ROWS <= kCrossMaxRows && fm::kTableDoubles == kCrossTableDoubles, "kernel_args.h");
  static_assert(N_RB * N_DB * 4 == RW && kCrossChunkBins % 4 == 0, "tiles per wave");
  extern __shared__ __attribute__((aligned(16))) double cross_lds[];
  double* table = cross_lds;
  double* buffers = cross_lds + kCrossTableDoubles;    // 2 x (kCrossChunkBins, 64); then the sums
  const fm::Consts kc = fm::make_consts();
#ifdef TC_DEVELOPER_KNOBS
  // (TC_FUSED_STAMPS: 16 slots of 100 MHz stamps per workgroup, wave 0, in the place of the
  // likelihood's data -- 0 entry, 1 draws set up, 2 chunks done, 3 sums in LDS, 4 deferred pairs
  // done, 5 weights and norms, 6 end; 8 / 9 / 10: ticks wave 0 spent in the occupations of the
  // chunks, waiting at their barriers, in their products; tools/r06_stamps_cross.py)
  unsigned long long* const stamps =
      ((a.priority >> 10) & 1) && a.chi2 == nullptr ? (unsigned long long*)a.chi2_data : nullptr;
  auto stamp = [&](int which) {
    if (stamps != nullptr && threadIdx.x == 0)
      stamps[blockIdx.x * 16 + which] = __builtin_amdgcn_s_memrealtime();
  };
  unsigned long long ticks_a = 0, ticks_wait = 0, ticks_b = 0, lap_from = 0;
  auto lap = [&](unsigned long long& into) {
    if (stamps != nullptr) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      into += now - lap_from;
      lap_from = now;
    }
  };
#else
  auto stamp = [](int) {};
  unsigned long long ticks_a = 0, ticks_wait = 0, ticks_b = 0;
  auto lap = [](unsigned long long&) {};
#endif
  stamp(0);
  set_priority(a.priority & 3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    for (int i = threadIdx.x; i < fm::kTableDoubles / 2; i += blockDim.x) dst[i] = src[i];
  }
  // (the deferred pairs: instances of their own -- undecorated, up to 64 rows)
  static_assert(!DEFER || (!ASSEMBIAS && !MODULATE && ROWS <= 64), "deferred pairs");
  constexpr bool kDeferrable = DEFER;
  constexpr bool deferring = DEFER;
  unsigned long long* bitmap = (unsigned long long*)(cross_lds + a.lds_bitmap);
  if (deferring)
    for (int i = threadIdx.x; i < a.n_groups; i += blockDim.x) bitmap[i] = 0;
  __syncthreads();
  // (medium batches: n_splits workgroups per tile of 64 draws, each with a range of the chunks;
  // the last one to arrive adds the shares in split order and finishes the tile -- as
  // predict_cross_small_kernel)
  const int n_splits = a.n_splits > 1 ? a.n_splits : 1;
  const int tile_index = (int)blockIdx.x / n_splits, split = (int)blockIdx.x % n_splits;
  const int chunk_begin = a.n_splits > 1 ? a.split_all[split] : 0;
  const int chunk_end = a.n_splits > 1 ? a.split_all[split + 1] : a.n_chunks;
  const int64_t col = (int64_t)tile_index * kLanes;
  const int64_t b0 = col + lane;
  const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;
  DrawParams dp;              // (the lane's draw: also what the deferred pairs fetch)

  // this wave's output tiles: row blocks rb0 (+ 4), draw blocks db0 (+ 1); D[m = l / 16 + 4 v]
  // [n = l % 16] in register v (the lane map of contract_quad_kernel)
  const int rb0 = RW == 4 ? (wave & 1) : (wave >> 1);
  const int db0 = RW == 4 ? (wave >> 1) : 2 * (wave & 1);
  f64x4 acc[N_RB][N_DB];
#pragma unroll
  for (int j = 0; j < N_RB; ++j)
#pragma unroll
    for (int i = 0; i < N_DB; ++i) acc[j][i] = f64x4{0.0, 0.0, 0.0, 0.0};
  // element (j, i, v) of the wave's sums sits at [row][draw] of a (ROWS, 64) array
  auto sum_index = [&](int j, int i, int v) {
    return (16 * (rb0 + 4 * j) + (lane >> 4) + 4 * v) * kLanes + 16 * (db0 + i) + (lane & 15);
  };
  {
    const double* th = a.theta + b * a.n_theta;
    const DrawSetup d = prepare_draw(table, kc, th[0], th[1], th[2], th[3], th[4],
                                     ASSEMBIAS ? th[5] : 0.0, ASSEMBIAS ? th[6] : 0.0);
    dp.log_m_min = d.log_m_min;
    dp.inv_sigma = d.inv_sigma;
    dp.m0 = d.m0;
    dp.log2_m1 = d.log2_m1;
    dp.sat_scale = d.sat_scale;
    dp.alpha = d.alpha;
    dp.a_cen = d.a_cen;
    dp.a_sat = d.a_sat;
    dp.bad = d.bad;
    dp.any_bad = __builtin_amdgcn_ballot_w64(dp.bad != 0) != 0;
    series_setup<MODULATE>(dp, a.group.series != nullptr, a.group.sat_series != nullptr);
    sc_i32 group_begin = (sc_i32)a.group.begin;
    sc_i32 chunk_group = (sc_i32)a.chunk_group;
    sc_i32 chunk_block = (sc_i32)a.chunk_block;
    const MarkPairs<kDeferrable> mark{bitmap};
    const GroupConsts gq{(sc_f64)a.group.log_m, (sc_f64)a.group.m, (sc_f64)a.group.weight,
                         (sc_f64)a.group.weight + a.n_bins * 10, (sc_f64)a.group.percentile,
                         (sc_i32)a.group.member,
                             SeriesConsts{(sc_f64)a.group.series, (sc_i32)a.group.series_thr,
                                          (sc_f64)a.group.sat_series, (sc_i32)a.group.sat_series_thr}};
    if (a.separate && chunk_begin >= a.n_central_chunks) {
      // (a share without centrals: their sums are zero)
      double* res0 = cross_lds + a.lds_res0;
#pragma unroll
      for (int j = 0; j < N_RB; ++j)
#pragma unroll
        for (int i = 0; i < N_DB; ++i)
#pragma unroll
          for (int v = 0; v < 4; ++v) res0[sum_index(j, i, v)] = 0.0;
    }
    stamp(1);
#ifdef TC_DEVELOPER_KNOBS
    if (stamps != nullptr) lap_from = __builtin_amdgcn_s_memrealtime();
#endif
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
      if (a.separate && chunk == a.n_central_chunks && chunk > chunk_begin) {
        // (the sums of the centrals are complete: set them aside; nobody reads them before the
        // barrier behind the last chunk)
        double* res0 = cross_lds + a.lds_res0;
#pragma unroll
        for (int j = 0; j < N_RB; ++j)
#pragma unroll
          for (int i = 0; i < N_DB; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              res0[sum_index(j, i, v)] = acc[j][i][v];
              acc[j][i][v] = 0.0;
            }
      }
      double* buffer = buffers + (chunk & 1) * (kCrossChunkBins * kLanes);
      const int g0 = chunk_group[chunk], g1 = chunk_group[chunk + 1];
      const int m0 = group_begin[g0], m1 = group_begin[g1];
      const int n_steps = (m1 - m0 + 3) >> 2;        // 4-bin steps of phase B
      // A. mean occupations of the chunk's bins (row = bin in chunk, skewed columns)
      auto emit = [&](int mi, int, double nbar) {
        const int row = mi - m0;
        buffer[row * kLanes + ((lane + 16 * (row & 3)) & 63)] = nbar;
      };
      if constexpr (kDeferrable) {
        // (launch.hip: only tables that have the records; no chunk holds both types: a loop
        // per type)
        sc_f64 records = (sc_f64)a.group.records;
        if (chunk < a.n_central_chunks) {
          for (int gr = g0 + wave; gr < g1; gr += W)
            occ_record_zheng07<true>(table, kc, gr,
                                     records + (int64_t)gr * series::record::kStride, dp, emit,
                                     mark);
        } else {
          for (int gr = g0 + wave; gr < g1; gr += W)
            occ_record_zheng07<false>(table, kc, gr,
                                      records + (int64_t)gr * series::record::kStride, dp, emit,
                                      mark);
        }
      } else {
        for (int gr = g0 + wave; gr < g1; gr += W)
          occ_group_zheng07<ASSEMBIAS, MODULATE>(table, kc, gr, group_begin[gr],
                                                 group_begin[gr + 1], gr < a.n_central_groups,
                                                 gq, a.split, dp, emit, mark);
      }
      // (the rows that fill the last step: zero, whatever an earlier chunk left there)
      for (int row = m1 - m0 + wave; row < 4 * n_steps; row += W) buffer[row * kLanes + lane] = 0.0;
      // B. this wave's tiles over every bin of the chunk; the first A operands are requested
      // before the barrier.  (The lane's offsets are formed here, per chunk, from a lane index
      // the optimiser cannot see through: hoisted out of the chunk loop they would cost six
      // registers across phase A, which has none to spare at four waves per SIMD.)
      int l = lane;
      asm volatile("" : "+v"(l));
      // B operand: bins 4 q + l / 16 of the chunk buffer, draws 16 (db0 + i) + l % 16, skewed
      int b_col[N_DB];
#pragma unroll
      for (int i = 0; i < N_DB; ++i)
        b_col[i] = (l >> 4) * kLanes + ((16 * (db0 + i) + (l & 15) + 16 * (l >> 4)) & 63);
      const double* a_lane = a.rows + ((int64_t)chunk_block[chunk] * NT + rb0) * 64 + l;
      double av[PF][N_RB];
#pragma unroll
      for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int j = 0; j < N_RB; ++j)
          av[s][j] = s < n_steps ? a_lane[(s * NT + 4 * j) * 64] : 0.0;
      lap(ticks_a);
      __syncthreads();
      lap(ticks_wait);
      for (int q = 0; q < n_steps; q += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          if (q + s < n_steps) {
            double bv[N_DB];
#pragma unroll
            for (int i = 0; i < N_DB; ++i) bv[i] = buffer[(q + s) * 4 * kLanes + b_col[i]];
#pragma unroll
            for (int j = 0; j < N_RB; ++j)
#pragma unroll
              for (int i = 0; i < N_DB; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s][j], bv[i], acc[j][i], 0, 0, 0);
            if (q + s + PF < n_steps) {
#pragma unroll
              for (int j = 0; j < N_RB; ++j)
                av[s][j] = a_lane[((q + s + PF) * NT + 4 * j) * 64];
            }
          }
        }
      }
      lap(ticks_b);
    }
  }
  stamp(2);
  __syncthreads();      // every wave has read the last chunk: the buffers take the sums
  double* res1 = buffers;                               // all bins, or the satellites
  double* res0 = a.separate ? cross_lds + a.lds_res0 : buffers;
  {
    // (a share that ends before the satellites begin holds sums of centrals only)
    const bool only_centrals = a.separate && chunk_end <= a.n_central_chunks;
#pragma unroll
    for (int j = 0; j < N_RB; ++j)
#pragma unroll
      for (int i = 0; i < N_DB; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int index = sum_index(j, i, v);
          if (only_centrals) res0[index] = acc[j][i][v];
          res1[index] = only_centrals ? 0.0 : acc[j][i][v];
        }
  }
  __syncthreads();
  stamp(3);
  if (kDeferrable && deferring) {
    // ---- the deferred (group, draw) pairs (cross_deferred_pairs) ----
    sc_i32 chunk_groups = (sc_i32)a.chunk_group;
    sc_i32 bin_operand = (sc_i32)a.bin_operand;
    unsigned* list = (unsigned*)(cross_lds + a.lds_list) + wave * 512;
    for (int pass = 0; pass < (a.separate ? 2 : 1); ++pass)
      cross_deferred_pairs<ROWS <= 32 ? 32 : 64>(
          a, table, kc, dp, bitmap, list, wave, lane, chunk_groups[chunk_begin],
          chunk_groups[chunk_end], a.separate ? pass : -1,
          [&](int bin, int row) {
            return a.rows[bin_operand[bin] + (row >> 4) * 64 + (row & 15)];
          },
          pass == 0 ? res0 : res1);
    __syncthreads();
  }
  stamp(4);
  if (n_splits > 1) {
    // the shares of the tile's workgroups: device-scope write-through stores, the last arrival
    // (a counter per tile, reset for the next launch on this lane) adds them in split order
    __shared__ int is_last;
    const int n_comp_rows = (a.separate ? 2 : 1) * ROWS;
    const int count = n_comp_rows * kLanes;
    unsigned long long* mine =
        (unsigned long long*)a.partial + ((int64_t)tile_index * n_splits + split) * count;
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x) {
      const double value = idx < ROWS * kLanes ? res0[idx] : res1[idx - ROWS * kLanes];
      __hip_atomic_store(mine + idx, __builtin_bit_cast(unsigned long long, value),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the wave's stores have arrived
    __syncthreads();
    if (threadIdx.x == 0) {
      const int arrived = __hip_atomic_fetch_add(a.counters + tile_index, 1, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
      is_last = arrived == n_splits - 1;
      if (is_last)
        __hip_atomic_store(a.counters + tile_index, 0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    const unsigned long long* shares =
        (const unsigned long long*)a.partial + (int64_t)tile_index * n_splits * count;
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x) {
      double sum = 0.0;
      for (int part = 0; part < n_splits; ++part)
        sum += __builtin_bit_cast(
            double, __hip_atomic_load(shares + (int64_t)part * count + idx, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT));
      if (idx < ROWS * kLanes) res0[idx] = sum; else res1[idx - ROWS * kLanes] = sum;
    }
    __syncthreads();
  }

  // ---- per draw: spline weights / norms of the tables, number densities ----
  set_priority((a.priority >> 4) & 3);
  // (the draw's index once more, from a lane index the optimiser cannot trace back: kept from
  // the top of the kernel it would sit in three registers across every phase -- spilled)
  int lane_again = lane;
  asm volatile("" : "+v"(lane_again));
  const int64_t b0_end = col + lane_again;
  const int64_t b_end = b0_end < a.n_draws ? b0_end : a.n_draws - 1;
  const int n_comp = a.separate ? 2 : 1;
  const int per_table = a.n_r + 1;
  double* coef = cross_lds + a.lds_tile;        // (K, 64): c_k / ngal_k
  double* tile = coef + a.n_tables * kLanes;    // (n_comp n_r, 65)
  if (wave == 0) {
    double n_cen = 0.0, n_sat = 0.0;
    for (int k = 0; k < a.n_tables; ++k) {
      double c = 1.0;
      if (a.interp) {
        for (int dim = 0; dim < a.n_dim; ++dim) {
          const int n = a.n_axis[dim];
          const double* xp = a.xp + a.axis_offset[dim];
          const double x = a.x[b_end * a.n_dim + dim];
          int seg = -1;
          for (int i = 0; i < n; ++i) seg += xp[i] <= x ? 1 : 0;   // np.digitize(x, xp) - 1
          if (x == xp[n - 1]) seg = n - 2;
          seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
          const double* m = a.a + a.a_offset[dim] + (int64_t)seg * 4 * n;
          const int j = a.table_node[k * a.n_dim + dim];
          const double x2 = x * x, x3 = x2 * x;
          c *= m[j] + m[n + j] * x + m[2 * n + j] * x2 + m[3 * n + j] * x3;
        }
      }
      const double cen = res0[(k * per_table + a.n_r) * kLanes + lane];
      const double sat = a.separate ? res1[(k * per_table + a.n_r) * kLanes + lane] : 0.0;
      const double total = cen + sat;
      // (separated by galaxy type a non-finite sum poisons both components of the table, as
      // the reference's mask-after-divide does: tabcorr.py:653-681, finalize_kernel)
      const bool poisoned = a.separate && !(fabs(total) <= 1.79769313486231570815e308);
      coef[k * kLanes + lane] = c / (poisoned ? __builtin_nan("") : total);
      n_cen += c * cen;
      n_sat += c * sat;
    }
    if (b0_end < a.n_draws) {
      if (a.separate) {
        a.ngal[2 * b0_end] = n_cen;
        a.ngal[2 * b0_end + 1] = n_sat;
      } else {
        a.ngal[b0_end] = n_cen + n_sat;
      }
    }
  }
  __syncthreads();
  stamp(5);
  const int n_rows = n_comp * a.n_r;
  for (int row = wave; row < n_rows; row += W) {
    const int comp = row / a.n_r, r = row % a.n_r;
    const double* res = comp == 0 ? res0 : res1;
    double sum = 0.0;
    for (int k = 0; k < a.n_tables; ++k)
      sum = fma(coef[k * kLanes + lane], res[(k * per_table + r) * kLanes + lane], sum);
    tile[row * (kLanes + 1) + lane] = sum;
  }
  __syncthreads();
  const int64_t n_valid = a.n_draws - col < kLanes ? a.n_draws - col : kLanes;
  if (a.chi2 != nullptr) {
    // chi2 = delta^T P delta for the lane's draw (total prediction; finalize_quad_kernel)
    const int rows_r = a.n_r;
    const double* data = a.chi2_data;
    const double* matrix = a.chi2_data + rows_r;
    double part = 0.0;
    for (int i = wave; i < rows_r; i += W) {
      double inner = 0.0;
      for (int j = 0; j < rows_r; ++j)
        inner = fma(matrix[i * rows_r + j], tile[j * (kLanes + 1) + lane] - data[j], inner);
      part = fma(tile[i * (kLanes + 1) + lane] - data[i], inner, part);
    }
    res1[wave * kLanes + lane] = part;           // (the sums are dead: all in the tile)
    __syncthreads();
    if (wave == 0 && lane < n_valid) {
      double total = 0.0;
      for (int w = 0; w < W; ++w) total += res1[w * kLanes + lane];
      a.chi2[col + lane] = total;
    }
    return;
  }
  for (int idx = threadIdx.x; idx < n_rows * kLanes; idx += blockDim.x) {
    const int dd = idx / n_rows, row = idx % n_rows;
    if (dd < n_valid) a.xi[(col + dd) * (int64_t)n_rows + row] = tile[row * (kLanes + 1) + dd];
  }
  stamp(6);
#ifdef TC_DEVELOPER_KNOBS
  if (stamps != nullptr && threadIdx.x == 0) {
    stamps[blockIdx.x * 16 + 8] = ticks_a;
    stamps[blockIdx.x * 16 + 9] = ticks_wait;
    stamps[blockIdx.x * 16 + 10] = ticks_b;
  }
#endif
}

// The same for up to 16 rows (one table with up to 15 r values -- the reference's excess surface
// density tables have 13 --, a few tables with a handful): the row sums fit every wave's
// registers next to the occupation arithmetic (two workgroups per CU all the same), so the waves
// simply stride over the groups, a member's mean occupation goes straight into the wave's 16
// sums, and the waves' sums meet ONCE, at the end, through LDS -- no buffer, no barrier per
// chunk: the AbacusSummit table takes 57.9 us per 10^4 draws this way against 73.7 through the
// chunks above (where the 2 rows per wave make phase B mostly LDS reads and scalar loads).
// Separated by galaxy type the first `cen_waves` waves take the groups of centrals.
// (four waves per SIMD = two workgroups per CU -- the second launch bound is waves per SIMD in
// HIP: left to itself the register allocator takes 130 registers since the moment expansion
// joined, and one workgroup per CU costs 58 -> 94 us per 10^4 draws of the AbacusSummit table)
template <bool ASSEMBIAS, bool MODULATE, bool DEFER = false>
__global__ __launch_bounds__(64 * kCrossWaves, DEFER ? 4 : 2) void predict_cross_small_kernel(
    CrossFusedArgs a) {
  constexpr int ROWS = kCrossSmallRows, kCrossChunk = kCrossSmallChunk;
  static_assert(fm::kTableDoubles == kCrossTableDoubles, "kernel_args.h");
  constexpr int W = kCrossWaves;
  extern __shared__ __attribute__((aligned(16))) double cross_lds[];
  // math table | stage (W, kCrossChunk, 64): the waves' sums of one chunk of rows, later the spline
  // weights / norms (K, 64) and the results tile | res (n_comp, ROWS, 64)
  double* table = cross_lds;
  double* stage = cross_lds + kCrossTableDoubles;
  double* res = stage + W * kCrossChunk * kLanes;
  const fm::Consts kc = fm::make_consts();
  set_priority(a.priority & 3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    for (int i = threadIdx.x; i < fm::kTableDoubles / 2; i += blockDim.x) dst[i] = src[i];
  }
  // (the deferred pairs of cross_deferred_pairs: compiled for the undecorated instance, which
  // then also takes the expansions of series.h -- without the node loop in its main loop the
  // registers suffice)
  static_assert(!DEFER || (!ASSEMBIAS && !MODULATE), "deferred pairs");
  constexpr bool kDeferrable = DEFER;
  constexpr bool deferring = DEFER;
  unsigned long long* bitmap = (unsigned long long*)(cross_lds + a.lds_bitmap);
  if (deferring)
    for (int i = threadIdx.x; i < a.n_groups; i += blockDim.x) bitmap[i] = 0;
  __syncthreads();
  // (medium batches: n_splits workgroups per tile of 64 draws, each with a share of the groups;
  // the last one to arrive adds the shares in split order and finishes the tile)
  const int n_splits = a.n_splits > 1 ? a.n_splits : 1;
  const int tile_index = (int)blockIdx.x / n_splits, split = (int)blockIdx.x % n_splits;
  const int64_t col = (int64_t)tile_index * kLanes;
  const int64_t b0 = col + lane;
  const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;

  // ---- occupations -> row sums ----
  double acc[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) acc[j] = 0.0;
  DrawParams dp;
  {
    const double* th = a.theta + b * a.n_theta;
    const DrawSetup d = prepare_draw(table, kc, th[0], th[1], th[2], th[3], th[4],
                                     ASSEMBIAS ? th[5] : 0.0, ASSEMBIAS ? th[6] : 0.0);
    dp.log_m_min = d.log_m_min;
    dp.inv_sigma = d.inv_sigma;
    dp.m0 = d.m0;
    dp.log2_m1 = d.log2_m1;
    dp.sat_scale = d.sat_scale;
    dp.alpha = d.alpha;
    dp.a_cen = d.a_cen;
    dp.a_sat = d.a_sat;
    dp.bad = d.bad;
    dp.any_bad = __builtin_amdgcn_ballot_w64(dp.bad != 0) != 0;
    // (the decorated instances are compiled without the moment expansions: dp's keys stay at
    // INT_MAX there)
    if (kDeferrable)
      series_setup<MODULATE>(dp, deferring && a.group.series != nullptr,
                             deferring && a.group.sat_series != nullptr);
    const MarkPairs<kDeferrable> mark{bitmap};
    sc_i32 group_begin = (sc_i32)a.group.begin;
    const GroupConsts gq{(sc_f64)a.group.log_m, (sc_f64)a.group.m, (sc_f64)a.group.weight,
                         (sc_f64)a.group.weight + a.n_bins * 10, (sc_f64)a.group.percentile,
                         (sc_i32)a.group.member,
                             SeriesConsts{(sc_f64)a.group.series, (sc_i32)a.group.series_thr,
                                          (sc_f64)a.group.sat_series, (sc_i32)a.group.sat_series_thr}};
    sc_f64 rows = (sc_f64)a.rows;
    auto emit = [&](int mi, int, double nbar) {
      sc_f64 coefficient = rows + (int64_t)mi * a.row_stride;
#pragma unroll
      for (int j = 0; j < ROWS; ++j) acc[j] = fma(coefficient[j], nbar, acc[j]);
    };
    // this workgroup's groups: [split_all[s], split_all[s + 1]), or -- separated by galaxy
    // type -- that share of the centrals for the first cen_waves waves and of the satellites
    // for the others
    int first = a.split_all[split] + wave, end = a.split_all[split + 1], stride = W;
    if (a.separate) {
      const bool cen = wave < a.cen_waves;
      first = cen ? a.split_cen[split] + wave : a.split_sat[split] + (wave - a.cen_waves);
      end = cen ? a.split_cen[split + 1] : a.split_sat[split + 1];
      stride = cen ? a.cen_waves : W - a.cen_waves;
    }
    for (int gr = first; gr < end; gr += stride) {
      if constexpr (kDeferrable) {        // (launch.hip: only tables that have the records)
        sc_f64 rec = (sc_f64)a.group.records + (int64_t)gr * series::record::kStride;
        if (gr < a.n_central_groups) occ_record_zheng07<true>(table, kc, gr, rec, dp, emit, mark);
        else occ_record_zheng07<false>(table, kc, gr, rec, dp, emit, mark);
      } else
        occ_group_zheng07<ASSEMBIAS, MODULATE, kDeferrable>(
            table, kc, gr, group_begin[gr], group_begin[gr + 1], gr < a.n_central_groups, gq,
            a.split, dp, emit, mark);
    }
  }

  // ---- the waves' sums, kCrossChunk rows at a time, in wave order ----
  const int n_comp = a.separate ? 2 : 1;
#pragma unroll
  for (int chunk = 0; chunk < ROWS / kCrossChunk; ++chunk) {
#pragma unroll
    for (int j = 0; j < kCrossChunk; ++j)
      stage[(wave * kCrossChunk + j) * kLanes + lane] = acc[chunk * kCrossChunk + j];
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_comp * kCrossChunk * kLanes; idx += blockDim.x) {
      const int comp = idx / (kCrossChunk * kLanes), rest = idx % (kCrossChunk * kLanes);
      const int w_begin = a.separate && comp == 1 ? a.cen_waves : 0;
      const int w_end = a.separate && comp == 0 ? a.cen_waves : W;
      double sum = 0.0;
      for (int w = w_begin; w < w_end; ++w) sum += stage[w * kCrossChunk * kLanes + rest];
      res[(comp * ROWS + chunk * kCrossChunk) * kLanes + rest] = sum;
    }
    __syncthreads();
  }

  if (kDeferrable && deferring) {
    // ---- the deferred (group, draw) pairs of this workgroup's groups (cross_deferred_pairs) ----
    unsigned* list = (unsigned*)stage + wave * 512;         // (the stage is free again)
    auto coefficient = [&](int bin, int row) { return a.rows[(int64_t)bin * a.row_stride + row]; };
    if (a.separate) {
      cross_deferred_pairs<ROWS>(a, table, kc, dp, bitmap, list, wave, lane, a.split_cen[split],
                                 a.split_cen[split + 1], -1, coefficient, res);
      cross_deferred_pairs<ROWS>(a, table, kc, dp, bitmap, list, wave, lane, a.split_sat[split],
                                 a.split_sat[split + 1], -1, coefficient, res + ROWS * kLanes);
    } else {
      cross_deferred_pairs<ROWS>(a, table, kc, dp, bitmap, list, wave, lane, a.split_all[split],
                                 a.split_all[split + 1], -1, coefficient, res);
    }
    __syncthreads();
  }

  if (n_splits > 1) {
    // this workgroup's share goes to the tile's slot `split` of the partial buffer; the last
    // of the tile's workgroups to get here (a counter per tile, which it resets for the next
    // launch on this lane) adds the shares in split order: deterministic
    // (the shares travel as device-scope write-through stores / loads and the wave waits until
    // its stores are acknowledged: a release fence by 128 workgroups -- a write-back and an
    // invalidation of the whole L2 each -- cost 40 us per 1024 draws)
    __shared__ int is_last;
    const int count = n_comp * ROWS * kLanes;
    unsigned long long* mine =
        (unsigned long long*)a.partial + ((int64_t)tile_index * n_splits + split) * count;
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x)
      __hip_atomic_store(mine + idx, __builtin_bit_cast(unsigned long long, res[idx]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the wave's stores have arrived
    __syncthreads();
    if (threadIdx.x == 0) {
      const int arrived = __hip_atomic_fetch_add(a.counters + tile_index, 1, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
      is_last = arrived == n_splits - 1;
      if (is_last)
        __hip_atomic_store(a.counters + tile_index, 0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    const unsigned long long* shares =
        (const unsigned long long*)a.partial + (int64_t)tile_index * n_splits * count;
    for (int idx = threadIdx.x; idx < count; idx += blockDim.x) {
      double sum = 0.0;
      for (int part = 0; part < n_splits; ++part)
        sum += __builtin_bit_cast(
            double, __hip_atomic_load(shares + (int64_t)part * count + idx, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT));
      res[idx] = sum;
    }
    __syncthreads();
  }

  // ---- per draw: spline weights / norms of the tables, number densities ----
  set_priority((a.priority >> 4) & 3);
  // (the draw's index once more, from a lane index the optimiser cannot trace back: kept from
  // the top of the kernel it would sit in registers across every phase)
  int lane_again = lane;
  asm volatile("" : "+v"(lane_again));
  const int64_t b0_end = col + lane_again;
  const int64_t b_end = b0_end < a.n_draws ? b0_end : a.n_draws - 1;
  const int per_table = a.n_r + 1;
  double* coef = stage;                         // (K, 64): c_k / ngal_k
  double* tile = stage + a.n_tables * kLanes;   // (n_comp n_r, 65)
  if (wave == 0) {
    double n_cen = 0.0, n_sat = 0.0;
    for (int k = 0; k < a.n_tables; ++k) {
      double c = 1.0;
      if (a.interp) {
        for (int dim = 0; dim < a.n_dim; ++dim) {
          const int n = a.n_axis[dim];
          const double* xp = a.xp + a.axis_offset[dim];
          const double x = a.x[b_end * a.n_dim + dim];
          int seg = -1;
          for (int i = 0; i < n; ++i) seg += xp[i] <= x ? 1 : 0;   // np.digitize(x, xp) - 1
          if (x == xp[n - 1]) seg = n - 2;
          seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
          const double* m = a.a + a.a_offset[dim] + (int64_t)seg * 4 * n;
          const int j = a.table_node[k * a.n_dim + dim];
          const double x2 = x * x, x3 = x2 * x;
          c *= m[j] + m[n + j] * x + m[2 * n + j] * x2 + m[3 * n + j] * x3;
        }
      }
      const double cen = res[(k * per_table + a.n_r) * kLanes + lane];
      const double sat = a.separate ? res[(ROWS + k * per_table + a.n_r) * kLanes + lane] : 0.0;
      const double total = cen + sat;
      // (separated by galaxy type a non-finite sum poisons both components of the table, as
      // the reference's mask-after-divide does: tabcorr.py:653-681, finalize_kernel)
      const bool poisoned = a.separate && !(fabs(total) <= 1.79769313486231570815e308);
      coef[k * kLanes + lane] = c / (poisoned ? __builtin_nan("") : total);
      n_cen += c * cen;
      n_sat += c * sat;
    }
    if (b0_end < a.n_draws) {
      if (a.separate) {
        a.ngal[2 * b0_end] = n_cen;
        a.ngal[2 * b0_end + 1] = n_sat;
      } else {
        a.ngal[b0_end] = n_cen + n_sat;
      }
    }
  }
  __syncthreads();
  const int n_rows = n_comp * a.n_r;
  for (int row = wave; row < n_rows; row += W) {
    const int comp = row / a.n_r, r = row % a.n_r;
    double sum = 0.0;
    for (int k = 0; k < a.n_tables; ++k)
      sum = fma(coef[k * kLanes + lane], res[(comp * ROWS + k * per_table + r) * kLanes + lane],
                sum);
    tile[row * (kLanes + 1) + lane] = sum;
  }
  __syncthreads();
  const int64_t n_valid = a.n_draws - col < kLanes ? a.n_draws - col : kLanes;
  if (a.chi2 != nullptr) {
    // chi2 = delta^T P delta for the lane's draw (total prediction; finalize_quad_kernel)
    const int rows_r = a.n_r;
    const double* data = a.chi2_data;
    const double* matrix = a.chi2_data + rows_r;
    double part = 0.0;
    for (int i = wave; i < rows_r; i += W) {
      double inner = 0.0;
      for (int j = 0; j < rows_r; ++j)
        inner = fma(matrix[i * rows_r + j], tile[j * (kLanes + 1) + lane] - data[j], inner);
      part = fma(tile[i * (kLanes + 1) + lane] - data[i], inner, part);
    }
    res[wave * kLanes + lane] = part;
    __syncthreads();
    if (wave == 0 && lane < n_valid) {
      double total = 0.0;
      for (int w = 0; w < W; ++w) total += res[w * kLanes + lane];
      a.chi2[col + lane] = total;
    }
    return;
  }
  for (int idx = threadIdx.x; idx < n_rows * kLanes; idx += blockDim.x) {
    const int dd = idx / n_rows, row = idx % n_rows;
    if (dd < n_valid) a.xi[(col + dd) * (int64_t)n_rows + row] = tile[row * (kLanes + 1) + dd];
  }
}

// ---- un-batched predict(): one draw, one launch ---------------------------------------
//
// The reference's usage is one predict() per MCMC step.  Three launches for one draw are
// latency, not work, so a single draw goes through one kernel: every workgroup evaluates
// all G * n_gauss occupation nodes (one node per thread and pass, per-bin sums in fixed
// order), contracts its share of the table positions (thread = r value x slice of
// positions, all of a thread's loads in flight at once) and writes its partial sums
// STRAIGHT INTO PAGE-LOCKED HOST MEMORY; the host adds the workgroups' partial sums in
// fixed order and normalises (table.cpp) -- deterministic like the batched path.  There is
// no inter-workgroup step on the device: the first version combined the partial sums in
// the last workgroup to arrive (device-scope fence + counter), and per-phase stamps
// (tools/archive/single_trace.py) showed that this tail -- fence 2.6 us, atomic 1.7 us, re-read and
// write-out 6.3 us -- was 10 of the kernel's 17 us.  Total correlation function only, one r
// tile, G * n_gauss <= kSingleMaxNodes.
constexpr int kSingleThreads = 1024;
constexpr int kSingleMaxNodes = 4096;
constexpr int kSingleMaxBins = 1024;

template <bool RESIDENT>
__device__ __forceinline__ void single_draw_body(SingleArgs a) {
  // (resident: one node per thread; what a thread reads from the table between the calls
  // stays in LDS -- registers would not hold it next to the body's own 80)
  __shared__ double node_value[RESIDENT ? kSingleThreads : kSingleMaxNodes];
  __shared__ unsigned kept_bins[RESIDENT ? 8 : 1][RESIDENT ? kSingleThreads : 1];
  __shared__ double kept_value[RESIDENT ? 8 : 1][RESIDENT ? kSingleThreads : 1];
  __shared__ double density[kSingleMaxBins];
  __shared__ double slice_sum[kSingleThreads];
  __shared__ double totals[2];
  __shared__ __attribute__((aligned(16))) double table[fm::kTableDoubles];
  const int tid = threadIdx.x;
  auto stamp = [&](int phase) {
#ifdef TC_DEVELOPER_KNOBS
    if (a.stamps != nullptr && tid == 0)
      a.stamps[blockIdx.x * 8 + phase] = __builtin_amdgcn_s_memrealtime();
#else
    (void)phase;
#endif
  };
  stamp(0);

  // everything here is latency: the draw arrives in the kernel arguments, the math tables
  // are staged in LDS
  const bool assembias = (a.flags & kFlagAssembias) != 0;
  const bool modulate = (a.flags & kFlagModulate) != 0;
  const int n_nodes = a.n_bins * a.n_gauss;
  // which table and which part of it (one table: everything direct)
  int part = blockIdx.x, n_parts = gridDim.x, which_table = 0;
  if (a.n_tables > 0) {
    which_table = blockIdx.x / a.blocks_per_table;
    part = blockIdx.x % a.blocks_per_table;
    n_parts = a.blocks_per_table;
    const int cls = a.table_class[which_table];
    a.table = a.tables[which_table];
    a.log_m = a.class_log_m[cls];
    a.m = a.class_m[cls];
    a.weight = a.class_weight[cls];
    a.n_h = a.class_n_h[cls];
    a.percentile = a.class_percentile[cls];
    a.ngal += 2 * which_table;
    a.partial += (int64_t)which_table * a.blocks_per_table * a.rt;
  } else if (a.n_walkers > 0) {
    const int walker = blockIdx.x / a.blocks_per_table;
    part = blockIdx.x % a.blocks_per_table;
    n_parts = a.blocks_per_table;
    for (int i = 0; i < 7; ++i)
      a.theta_value[i] = i < a.n_theta ? a.theta_many[walker * a.n_theta + i] : 0.0;
    a.ngal += 2 * walker;
    a.partial += (int64_t)walker * a.blocks_per_table * a.rt;
  }
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    for (int i = tid; i < fm::kTableDoubles / 2; i += kSingleThreads) dst[i] = src[i];
  }
  __syncthreads();
  stamp(1);
  // Resident form: the rest of the body once per call, the calls taken from the mailbox.
  // Wave 0 polls it (uncached loads over PCIe; a.poll_waves > 1: that many waves, started a
  // fraction of a round trip apart -- more samples per round trip, but every reader of the
  // mailbox's two cache lines delays the host's next stores to them: 13 workgroups x 4 waves
  // 25 us per call against 14.5 with one wave each) and the others sleep at the barrier; every
  // path out of the polling loop is bounded by the two time limits.
  __shared__ volatile int resident_decided;
  __shared__ int resident_leave;
  __shared__ double resident_theta[7];
  __shared__ unsigned long long resident_seen;
  unsigned long long serving = a.epoch;
  unsigned long long t_begin = 0, t_last = 0;
  if (RESIDENT) {
    t_begin = t_last = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) resident_decided = 0;
    __syncthreads();
  }
  // What a thread reads from memory does not change between the calls of a resident launch:
  // its quadrature node stays in registers, its eight table positions in LDS (tables of up to
  // 1024 nodes and launches sized for a single pass: launch.hip, resident_eligible).
  const int rt = a.rt;
  const int n_slices = kSingleThreads / rt;
  const int r = tid % rt, slice = tid / rt;
  const int64_t per_block = (a.n_positions + n_parts - 1) / n_parts;
  const int64_t q_begin = per_block * part;
  const int64_t q_end = q_begin + per_block < a.n_positions ? q_begin + per_block
                                                             : a.n_positions;
  auto load_positions = [&](int64_t q0, int (&off_i)[8], int (&off_j)[8], double (&value)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t qu = q0 + (int64_t)u * n_slices;
      const int64_t q = qu < q_end ? qu : q_begin;
      // layouts of table.cpp: blocks of 8 positions, LDS row offsets per position
      const int64_t slot = ((q >> 3) * 4 + (q & 3)) * 4 + ((q >> 2) & 1) * 2;
      off_i[u] = a.pos_off[slot];
      off_j[u] = a.pos_off[slot + 1];
      const int64_t index = (q >> 3) * 8 * rt +
                            ((((r >> 2) * 16 + (q & 3) * 4 + (r & 3)) << 1) + ((q >> 2) & 1));
      value[u] = qu < q_end ? a.table[index] : 0.0;
    }
  };
  double kept_lm = 0.0, kept_mass = 0.0, kept_wk = 0.0, kept_n_h = 0.0;
  bool kept_above = false;
  if (RESIDENT) {
    if (tid < n_nodes) {
      kept_lm = a.log_m[tid];
      kept_mass = a.m[tid];
      kept_wk = a.weight[tid];
      kept_above = a.percentile[tid / a.n_gauss] > a.split;
    }
    if (tid < a.n_bins) kept_n_h = a.n_h[tid];
    if (slice < n_slices && q_begin < q_end) {
      int off_i[8], off_j[8];
      double value[8];
      load_positions(q_begin + slice, off_i, off_j, value);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // (the positions' bins, 16 bits each)
        kept_bins[u][tid] = (unsigned)(off_i[u] >> 9) | ((unsigned)(off_j[u] >> 9) << 16);
        kept_value[u][tid] = value[u];
      }
    }
  }
  for (;;) {
  if (RESIDENT) {
    // One 16-byte entry {value, call number} per parameter, written by the host in that
    // order: lane i of a polling wave reads entry i with ONE uncached load (a 16-byte read
    // inside a cache line is a snapshot of it), so a single PCIe round trip both detects the
    // call and fetches its parameters; the call is there when all seven entries carry its
    // number.  The first wave to decide publishes the parameters and the decision.
    if (tid < 64 * a.poll_waves) {
      typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
      const int lane = tid & 63;
      const u64x2* entry = (const u64x2*)a.mailbox + (lane < 7 ? lane : 0);
      u64x2 word = {0ull, 0ull};
      for (int i = 0; i < (tid >> 6); ++i) __builtin_amdgcn_s_sleep(24);
      while (resident_decided == 0) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(word) : "v"(entry) : "memory");
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        const bool here = word.y == serving;
        const bool stop = word.y == kResidentStop;
        int leave = -1;
        if (__builtin_amdgcn_ballot_w64(!here) == 0) leave = 0;
        else if (__builtin_amdgcn_ballot_w64(stop) != 0 || now - t_last > a.idle_ticks ||
                 now - t_begin > a.life_ticks) leave = 1;
        if (leave >= 0) {
          // (several waves may decide at once: a call that is there is there for all of them,
          // and "leave" can only lose against "go" -- the later writer -- when the call
          // arrived in between, which the host handles like any workgroup that has left)
          if (lane < 7) resident_theta[lane] = __builtin_bit_cast(double, word.x);
          if (lane == 0) {
            resident_leave = leave;
            resident_seen = now;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) resident_decided = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    const int leave = resident_leave;
    for (int i = 0; i < 7; ++i) a.theta_value[i] = i < a.n_theta ? resident_theta[i] : 0.0;
    __syncthreads();
    if (tid == 0) resident_decided = 0;
    if (leave != 0) break;
    a.epoch = serving;
  }
  // (made per call: the resident form would otherwise hold their registers while it waits)
  const fm::Consts kc = fm::make_consts();
  const DrawSetup d = prepare_draw(table, kc, a.theta_value[0], a.theta_value[1],
                                   a.theta_value[2], a.theta_value[3], a.theta_value[4],
                                   assembias ? a.theta_value[5] : 0.0,
                                   assembias ? a.theta_value[6] : 0.0);
  const double log_m_min = d.log_m_min, inv_sigma = d.inv_sigma, m0 = d.m0;
  const double log2_m1 = d.log2_m1, sat_scale = d.sat_scale, alpha = d.alpha;
  const double a_cen = d.a_cen, a_sat = d.a_sat;
  const double f1 = (1.0 - a.split) / a.split, f2 = a.split / (1.0 - a.split);

  // (resident: one node per thread, its constants in registers -- launch.hip admits only
  // tables of up to kSingleThreads nodes and launches sized for a single pass)
  const int node_end = RESIDENT ? (tid < n_nodes ? tid + 1 : 0) : n_nodes;
  for (int idx = tid; idx < node_end; idx += kSingleThreads) {
    const int g = idx / a.n_gauss;
    const double lm = RESIDENT ? kept_lm : a.log_m[idx], mass = RESIDENT ? kept_mass : a.m[idx];
    const double wk = RESIDENT ? kept_wk : a.weight[idx];
    const bool above = RESIDENT ? kept_above : a.percentile[g] > a.split;
    double n;
    if (g < a.n_central) {
      n = fma(0.5, fm::erf_fast(table, kc, (lm - log_m_min) * inv_sigma), 0.5);
      if (assembias) n = heaviside_assembias(n, a_cen, above, f2, f1, true);
    } else {
      const double x = mass - m0;
      n = fm::exp2_fast(
          table, kc,
          alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, log2_m1),
          x > 0.0);
      n *= sat_scale;
      if (modulate)
        n *= fma(0.5, fm::erf_fast(table, kc, (lm - log_m_min) * inv_sigma), 0.5);
      if (assembias) n = heaviside_assembias(n, a_sat, above, f2, f1, false);
    }
    if (d.bad) {   // (prepare_draw: parameters the fast path cannot represent)
      const bool cen_nan = (d.bad & kBadCen) || ((d.bad & kTieCen) && lm == log_m_min);
      if (g >= a.n_central && (d.bad & kInfSat) && n != 0.0) {
        n = assembias ? __builtin_nan("") : __builtin_huge_val();
      }
      if (g < a.n_central ? cen_nan
                          : (((d.bad & kBadSat) && n != 0.0) || (modulate && cen_nan)))
        n = __builtin_nan("");
    }
    node_value[idx] = wk * n;
  }
  __syncthreads();
  stamp(2);
  for (int g = tid; g < a.n_bins; g += kSingleThreads) {
    double acc = 0.0;
    for (int k = 0; k < a.n_gauss; ++k) acc += node_value[g * a.n_gauss + k];
    density[g] = acc * (RESIDENT ? kept_n_h : a.n_h[g]);
  }
  __syncthreads();
  if (tid < 128) {   // centrals / satellites totals: one wave each, fixed order
    const int which = tid >> 6, lane = tid & 63;
    const int lo = which == 0 ? 0 : a.n_central, hi = which == 0 ? a.n_central : a.n_bins;
    double total = 0.0;
    for (int g = lo + lane; g < hi; g += 64) total += density[g];
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) total += __shfl_down(total, offset, 64);
    // (every workgroup holds the same sums; the first one reports them, below)
    if (lane == 0) totals[which] = total;
  }

  // contraction of this workgroup's positions: thread = (slice, r); eight positions per
  // pass, their loads issued together (the launch sizes the grid for a single pass)
  double acc = 0.0;
  if (slice < n_slices && q_begin < q_end) {
    const int64_t q_last = RESIDENT ? q_begin + slice + 1 : q_end;
    for (int64_t q0 = q_begin + slice; q0 < q_last; q0 += 8 * n_slices) {
      int off_i[8], off_j[8];
      double value[8];
      if (RESIDENT) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned bins = kept_bins[u][tid];
          off_i[u] = (int)(bins & 0xffffu) << 9;
          off_j[u] = (int)(bins >> 16) << 9;
          value[u] = kept_value[u][tid];
        }
      } else {
        load_positions(q0, off_i, off_j, value);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        double w = density[off_j[u] >> 9];
        if (a.mode == 0) w *= density[off_i[u] >> 9];
        acc = fma(value[u], w, acc);
      }
    }
  }
  slice_sum[tid] = acc;
  __syncthreads();
  stamp(3);
  // all stores to host memory come from wave 0 (rt <= 32): the partial sums, the number
  // densities, then -- behind a system-scope fence -- the workgroup's completion word
  // the slices of every r value are added in two levels (eight partial sums per r value, then
  // one: 51 dependent LDS reads by 20 threads cost 1.4 us of a 7 us kernel), in fixed order
  constexpr int kFan = 8;
  double level = 0.0;
  if (tid < rt * kFan) {
    const int rr = tid % rt;
    for (int s = tid / rt; s < n_slices; s += kFan) level += slice_sum[s * rt + rr];
  }
  __syncthreads();
  if (tid < rt * kFan) slice_sum[tid] = level;
  __syncthreads();
  if (tid < 64) {
    // Write-through stores to the page-locked results (system scope, no cache holds them), the
    // wave waits until all of them are acknowledged, then the completion word the same way:
    // what a system-scope fence + release store would add -- two write-backs and an
    // invalidation of the whole L2 -- has nothing to write back here and costs ~1 us per call.
    auto store_host = [](double* address, double value) {
      __hip_atomic_store((unsigned long long*)address,
                         __builtin_bit_cast(unsigned long long, value), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if (tid < rt) {
      double total = 0.0;
      for (int s = 0; s < kFan; ++s) total += slice_sum[s * rt + tid];
      store_host(a.partial + (int64_t)part * rt + tid, total);
    } else if (tid < rt + 2 && part == 0) {
      // (the address formed here, from an index the optimiser cannot hoist out of the resident
      // form's loop of calls: kept across the loop it was spilled, and every call of workgroup
      // 0 waited for a scratch load in front of its number densities)
      int which = tid - rt;
      asm volatile("" : "+v"(which));
      store_host(a.ngal + which, totals[which]);
    }
    if (a.done != nullptr) {
      __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): the wave's stores have arrived
      if (tid == 0)
        __hip_atomic_store(a.done + blockIdx.x, a.epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  stamp(4);
  if (!RESIDENT) break;
  ++serving;
  {
    // (diagnosis, tools/archive/r03_resident.py: 100 MHz ticks from the sight of the call to here)
    t_last = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) a.exited[kResidentBusyOffset + blockIdx.x] = t_last - resident_seen;
  }
  }   // (calls of the resident form)
  if (RESIDENT && tid == 0) {
    __threadfence_system();
    __hip_atomic_store(a.exited + blockIdx.x, a.launch_id, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#ifdef TC_UNIT_SINGLE   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(kSingleThreads) void single_draw_kernel(SingleArgs a) {
  single_draw_body<false>(a);
}
#endif

// The same body, resident: one launch serves every un-batched call until the host says stop,
// none has arrived for a.idle_ticks, or a.life_ticks have passed (SingleArgs, kernel_args.h).
// One table, one draw per call (a.n_tables == 0, a.n_walkers == 0).
#ifdef TC_UNIT_SINGLE   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(kSingleThreads) void resident_draw_kernel(SingleArgs a) {
  single_draw_body<true>(a);
}
#endif

// ---- ensembles without a launch: the resident ensemble kernel ---------------------------
//
// kernel_args.h (EnsembleArgs) describes the phases and the protocol.  Every loop that waits
// for another workgroup or for the host is bounded by a time limit, after which the workgroup
// leaves and says so: the grid drains whatever happens to the others.
namespace ens {
__device__ __forceinline__ void store_agent(double* address, double value) {
  __hip_atomic_store((unsigned long long*)address, __builtin_bit_cast(unsigned long long, value),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* address) {
  return __builtin_bit_cast(double,
                            __hip_atomic_load((const unsigned long long*)address,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_host(double* address, double value) {
  __hip_atomic_store((unsigned long long*)address, __builtin_bit_cast(unsigned long long, value),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void wait_stores() { __builtin_amdgcn_s_waitcnt(0x0f70); }  // vmcnt(0)
}  // namespace ens

#ifdef TC_UNIT_SINGLE   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(kEnsembleThreads) void resident_ensemble_kernel(EnsembleArgs a) {
  // (512 threads: eight waves with 256 registers each keep the loads of a phase in flight
  // together; sixteen waves of 128 registers spilled and walked the LDS one read at a time)
  extern __shared__ __attribute__((aligned(16))) unsigned char ens_lds[];
  double* table = (double*)ens_lds;
  // phase A: node_value[1024] | density[1024]; phase B: other[32][64]; phase C: sums[8][64]
  double* area = (double*)(ens_lds + a.lds_area);
  double* node_value = area;
  double* density = area + 1024;
  double* dens_lds = (double*)(ens_lds + a.lds_dens);     // (n_bins + 2, 65)
  double* t_lds = (double*)(ens_lds + a.lds_t);           // (4 * per_quarter, 32)
  unsigned* ij_lds = (unsigned*)(ens_lds + a.lds_ij);     // (4 * per_quarter)
  __shared__ int s_abort;
  __shared__ int s_leave, s_walkers;
  __shared__ unsigned long long s_epoch, s_seen;
  __shared__ double s_theta[8];
  __shared__ double s_totals[2];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int c = b & 3, slice = b >> 2;
  const int rt = a.rt;
  const int pq = a.per_quarter;
  const bool assembias = (a.flags & kFlagAssembias) != 0;
  const bool modulate = (a.flags & kFlagModulate) != 0;
  const int n_nodes = a.n_bins * a.n_gauss;

  // once per launch: the math tables, this slice's positions, this thread's two nodes
  {
    typedef double __attribute__((ext_vector_type(2))) double2v;
    const double2v* src = (const double2v*)a.math_table;
    double2v* dst = (double2v*)table;
    for (int i = tid; i < fm::kTableDoubles / 2; i += kEnsembleThreads) dst[i] = src[i];
  }
  const int64_t slice_begin = (int64_t)slice * 4 * pq;
  for (int idx = tid; idx < 4 * pq * 32; idx += kEnsembleThreads) {
    const int pl = idx >> 5, r = idx & 31;
    const int64_t q = slice_begin + pl;
    // layouts of table.cpp, as single_draw_body reads them; 32 rows per position here, zeros
    // beyond rt
    const int64_t index = (q >> 3) * 8 * rt +
                          ((((r >> 2) * 16 + (q & 3) * 4 + (r & 3)) << 1) + ((q >> 2) & 1));
    t_lds[idx] = q < a.n_positions && r < rt ? a.table[index] : 0.0;
  }
  for (int pl = tid; pl < 4 * pq; pl += kEnsembleThreads) {
    const int64_t q = slice_begin + pl;
    unsigned bins = 0xffffffffu;            // (no position)
    if (q < a.n_positions) {
      const int64_t slot = ((q >> 3) * 4 + (q & 3)) * 4 + ((q >> 2) & 1) * 2;
      bins = (unsigned)(a.pos_off[slot] >> 9) | ((unsigned)(a.pos_off[slot + 1] >> 9) << 16);
    }
    ij_lds[pl] = bins;
  }
  double kept_lm[2] = {0.0, 0.0}, kept_mass[2] = {0.0, 0.0}, kept_wk[2] = {0.0, 0.0};
  bool kept_above[2] = {false, false};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int node = tid + u * kEnsembleThreads;
    if (node < n_nodes) {
      kept_lm[u] = a.log_m[node];
      kept_mass[u] = a.m[node];
      kept_wk[u] = a.weight[node];
      kept_above[u] = a.percentile[node / a.n_gauss] > a.split;
    }
  }
  double kept_n_h[2] = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + u * kEnsembleThreads < a.n_bins) kept_n_h[u] = a.n_h[tid + u * kEnsembleThreads];
  if (tid == 0) s_abort = 0;
  __syncthreads();

  unsigned long long serving = a.epoch;
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_last = t_begin;
  // (a wait inside a call is over when its time is up or the host says stop -- it does when it
  // has seen a workgroup leave that this call needed; workgroup 0 passes the word on where the
  // others cannot see the header)
  const unsigned long long* source = b == 0 || a.direct ? a.mailbox : a.callword;
  unsigned waited = 0;       // (the stop word every 16th look: a look costs a trip to memory)
  auto timed_out = [&](unsigned long long since) {
    if ((++waited & 15u) == 0) {
      unsigned long long word;
      asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=v"(word) : "v"(source) : "memory");
      if (word == kResidentStop) {
        if (b == 0 && !a.direct)
          __hip_atomic_store(a.callword, kResidentStop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
    }
    return __builtin_amdgcn_s_memrealtime() - since > a.call_ticks;
  };
  // (diagnosis, tools/r04_ensemble.py: 100 MHz stamps of workgroup 0 in page-locked memory)
  // (developer builds only: a store to host memory in front of a phase's `s_waitcnt vmcnt(0)`
  // makes workgroup 0 -- walker 0, reducer of row 0 -- wait for the link: 2 us per call)
  auto stamp = [&](int which, unsigned long long value) {
#ifdef TC_DEVELOPER_KNOBS
    if (b == 0 && tid == 0)
      __hip_atomic_store(a.exited + gridDim.x + which, value, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
#else
    (void)which;
    (void)value;
#endif
  };

  for (;;) {
    // (the thread's indices of THIS call, from a value the optimiser cannot trace back to
    // threadIdx: addresses formed from it stay inside the loop of calls.  Hoisted out of
    // it they filled 19 more registers than the 255 there are, and every call waited for
    // their scratch loads in each of its phases)
    int tid_call = tid;
    asm volatile("" : "+v"(tid_call));
    const int lane_call = tid_call & 63;
    // ---- the call.  a.direct: the mailbox lies in device memory that the host writes through
    // the PCIe aperture, every workgroup polls its header there (local reads) and a walker's
    // workgroup takes its parameters from behind it.  Otherwise the mailbox is page-locked
    // host memory: workgroup 0 alone polls the header and passes it on through a word in
    // device memory that the others poll (256 pollers of host memory keep the link so busy
    // that the host's own stores take tens of microseconds), and the parameters cross the link
    // when a workgroup asks for them.  The header is written behind the parameters.
    if (wave == 0) {
      // (the header and the forwarded word: number of the call << 10 | walkers)
      int leave = -1, walkers = 0;
      unsigned long long epoch = 0, now = 0;
      for (;;) {
        unsigned long long word;
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(word) : "v"(source) : "memory");
        now = __builtin_amdgcn_s_memrealtime();
        if (word == kResidentStop) {
          leave = 1;
        } else if ((word >> 10) >= serving) {
          leave = 0;
          walkers = (int)(word & 1023);
          epoch = word >> 10;
        } else if (now - t_last > a.idle_ticks || now - t_begin > a.life_ticks) {
          leave = 1;
        }
        if (leave >= 0) break;
        if (b == 0) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(4);
      }
      if (b == 0 && !a.direct && lane_call == 0)
        __hip_atomic_store(a.callword,
                           leave != 0 ? kResidentStop : (epoch << 10) | (unsigned long long)walkers,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (leave == 0 && b < walkers && lane_call < 7) {
        // (parameters: 8 doubles per walker behind the header's line)
        const double* theta = (const double*)a.mailbox + 8 + (size_t)b * 8 + lane_call;
        double value;
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(value) : "v"(theta) : "memory");
        s_theta[lane_call] = value;
      }
      if (lane_call == 0) {
        s_leave = leave;
        s_walkers = walkers;
        s_epoch = epoch;
        s_seen = now;
      }
    }
    __syncthreads();
    if (s_leave != 0) break;
    const int n_walkers = s_walkers;
    const unsigned long long epoch = s_epoch;
    const bool is_walker = b < n_walkers;
    stamp(0, s_seen);

    // ---- A: the occupation of walker b ----------------------------------------------------
    if (is_walker && (a.skip & 1)) {
      if (tid_call == 0) __hip_atomic_store(a.flag_a + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (is_walker) {
      const fm::Consts kc = fm::make_consts();
      const DrawSetup d = prepare_draw(table, kc, s_theta[0], s_theta[1], s_theta[2], s_theta[3],
                                       s_theta[4], assembias ? s_theta[5] : 0.0,
                                       assembias ? s_theta[6] : 0.0);
      const double f1 = (1.0 - a.split) / a.split, f2 = a.split / (1.0 - a.split);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int node = tid_call + u * kEnsembleThreads;
        if (node < n_nodes) {
          const int g = node / a.n_gauss;
          const double lm = kept_lm[u], mass = kept_mass[u];
          double n;
          if (g < a.n_central) {
            n = fma(0.5, fm::erf_fast(table, kc, (lm - d.log_m_min) * d.inv_sigma), 0.5);
            if (assembias) n = heaviside_assembias(n, d.a_cen, kept_above[u], f2, f1, true);
          } else {
            const double x = mass - d.m0;
            n = fm::exp2_fast(
                table, kc,
                d.alpha * fm::log2_fast_offset(table, kc, x > 1e-300 ? x : 1e-300, d.log2_m1),
                x > 0.0);
            n *= d.sat_scale;
            if (modulate)
              n *= fma(0.5, fm::erf_fast(table, kc, (lm - d.log_m_min) * d.inv_sigma), 0.5);
            if (assembias) n = heaviside_assembias(n, d.a_sat, kept_above[u], f2, f1, false);
          }
          if (d.bad) {   // (prepare_draw: parameters the fast path cannot represent)
            const bool cen_nan = (d.bad & kBadCen) || ((d.bad & kTieCen) && lm == d.log_m_min);
            if (g >= a.n_central && (d.bad & kInfSat) && n != 0.0)
              n = assembias ? __builtin_nan("") : __builtin_huge_val();
            if (g < a.n_central ? cen_nan
                                : (((d.bad & kBadSat) && n != 0.0) || (modulate && cen_nan)))
              n = __builtin_nan("");
          }
          node_value[node] = kept_wk[u] * n;
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = tid_call + u * kEnsembleThreads;
        if (g < a.n_bins) {
          double acc = 0.0;
          for (int k = 0; k < a.n_gauss; ++k) acc += node_value[g * a.n_gauss + k];
          density[g] = acc * kept_n_h[u];
        }
      }
      __syncthreads();
      if (tid_call < 128) {   // centrals / satellites totals: one wave each, fixed order
        const int which = tid_call >> 6;
        const int lo = which == 0 ? 0 : a.n_central, hi = which == 0 ? a.n_central : a.n_bins;
        double total = 0.0;
        for (int g = lo + lane_call; g < hi; g += 64) total += density[g];
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) total += __shfl_down(total, offset, 64);
        if (lane_call == 0) s_totals[which] = total;
      }
      // the densities go out while the totals are formed; the totals behind them
      for (int g = tid_call; g < a.n_bins; g += kEnsembleThreads)
        ens::store_agent(a.dens + (size_t)b * a.dens_stride + g, density[g]);
      __syncthreads();
      if (tid_call < 2) ens::store_agent(a.dens + (size_t)b * a.dens_stride + a.n_bins + tid_call, s_totals[tid_call]);
      ens::wait_stores();
      __syncthreads();
      if (tid_call == 0)
        __hip_atomic_store(a.flag_a + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(1, __builtin_amdgcn_s_memrealtime());

    // ---- B: this workgroup's part of its slice for one group of 64 walkers ----------------
    const int n_wg = (n_walkers + 63) >> 6;
    int grp, q_first, q_count;
    if (n_wg == 1) {
      grp = 0; q_first = c; q_count = 1;
    } else if (n_wg == 2) {
      grp = c & 1; q_first = 2 * (c >> 1); q_count = 2;
    } else {
      grp = c; q_first = 0; q_count = 4;
    }
    const bool active = grp < n_wg;
    if (active) {
      if (wave == 0) {
        const int walker = grp * 64 + lane_call;
        const unsigned long long since = __builtin_amdgcn_s_memrealtime();
        for (;;) {
          const unsigned long long flag =
              walker < n_walkers ? __hip_atomic_load(a.flag_a + walker, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)
                                 : epoch;
          if (__builtin_amdgcn_ballot_w64(flag != epoch) == 0) break;
          if (timed_out(since)) {
            s_abort = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (s_abort) break;
      stamp(2, __builtin_amdgcn_s_memrealtime());
      if (!(a.skip & 4)) {
        // the group's densities, (walker, bin) in memory -> (bin, walker) in LDS; eight loads
        // in flight per thread
        const int stride = a.dens_stride;
        const int count = 64 * stride;
        for (int idx0 = tid_call; idx0 < count; idx0 += 8 * kEnsembleThreads) {
          double value[8];
          int at[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = idx0 + u * kEnsembleThreads;
            const int w = idx / stride, g = idx - w * stride;
            const int walker = grp * 64 + w;
            const bool valid = idx < count && g < a.n_bins + 2;
            at[u] = valid ? g * kEnsembleDensPad + w : -1;
            value[u] = valid && walker < n_walkers
                           ? ens::load_agent(a.dens + (size_t)walker * stride + g) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (at[u] >= 0) dens_lds[at[u]] = value[u];
        }
      }
      __syncthreads();
      stamp(3, __builtin_amdgcn_s_memrealtime());
      const unsigned long long cycles_begin = __builtin_amdgcn_s_memtime();

      // Waves = 2 x 4 groups of 8 rows (the rows beyond rt are zeros in t_lds).  A quarter is
      // ALWAYS summed as (its even positions) + (its odd positions), each half in position
      // order: with one quarter per workgroup the two waves of a row group take a half each
      // and meet through the LDS; with two or four quarters a wave sums whole quarters (both
      // halves, two accumulators) -- one exchange per call instead of one per quarter.
      // (sel and the rows as per-lane_call values: with wave-uniform ones the compiler moves the
      // positions' bins into scalar registers one LDS read, one wait and one branch at a time)
      const int sel = (tid_call >> 6) & 1, row0 = (tid_call >> 7) * 8;
      const bool rows_used = (wave >> 1) * 8 < rt;
      // (one instance of the inner code, real loops: unrolled over 16 guarded positions and
      // inlined five times the phase was 7000 instructions, more than the instruction cache
      // holds, with a spilled scalar condition per position: 4100 cycles per half)
      auto quarter_sum = [&](int base, int parity_begin, int parity_end, double (&out)[8]) {
#pragma unroll 1
        for (int parity = parity_begin; parity < parity_end; ++parity) {
          double acc[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = 0.0;
          const int n_half = (pq - parity + 1) >> 1;
#pragma unroll 1
          for (int u0 = 0; u0 < n_half; u0 += 4) {
            int p[4];
            unsigned bins[4];
            double w[4];
            // (a slot beyond the quarter: w = 0 times a finite row of the first position)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int at = base + parity + 2 * (u0 + u);
              p[u] = at < base + pq ? at : base;
              bins[u] = at < base + pq ? ij_lds[p[u]] : 0xffffffffu;
            }
            // (the rows of all four positions requested before anything waits for them: left
            // to itself the compiler reads two values, waits, multiplies, reads the next two)
            double t[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int k = 0; k < 8; ++k) t[u][k] = t_lds[p[u] * 32 + row0 + k];
            double first[4], second[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const bool valid = bins[u] != 0xffffffffu;
              const unsigned bi = valid ? bins[u] & 0xffffu : 0u, bj = valid ? bins[u] >> 16 : 0u;
              first[u] = dens_lds[bj * kEnsembleDensPad + lane_call];
              second[u] = a.mode == 0 ? dens_lds[bi * kEnsembleDensPad + lane_call] : 1.0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              double value = first[u];
              if (a.mode == 0) value *= second[u];
              w[u] = bins[u] != 0xffffffffu ? value : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int k = 0; k < 8; ++k) acc[k] = fma(t[u][k], w[u], acc[k]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) out[k] = parity == parity_begin ? acc[k] : out[k] + acc[k];
        }
      };
      double mine[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) mine[k] = 0.0;
      if (rows_used && !(a.skip & 2)) {
        // one quarter: wave `sel` sums its half; two quarters: wave `sel` sums quarter sel;
        // four: quarters 2 sel and 2 sel + 1, added at once -- (q0 + q1) and (q2 + q3)
        const int n_mine = q_count == 1 ? 1 : q_count >> 1;
#pragma unroll 1
        for (int j = 0; j < n_mine; ++j) {
          double quarter[8];
          const int base = (q_first + (q_count == 1 ? 0 : sel * n_mine) + j) * pq;
          quarter_sum(base, q_count == 1 ? sel : 0, q_count == 1 ? sel + 1 : 2, quarter);
#pragma unroll
          for (int k = 0; k < 8; ++k) mine[k] = j == 0 ? quarter[k] : mine[k] + quarter[k];
        }
        if (sel == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) area[(row0 + k) * 64 + lane_call] = mine[k];
        }
      }
      __syncthreads();
      stamp(4, __builtin_amdgcn_s_memrealtime());
      stamp(6, __builtin_amdgcn_s_memtime() - cycles_begin);    // (shader cycles of the sums)
      if (sel == 0 && rows_used) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (row0 + k < rt)
            ens::store_agent(a.partial + ((size_t)b * rt + row0 + k) * 64 + lane_call,
                             mine[k] + area[(row0 + k) * 64 + lane_call]);
      }
      ens::wait_stores();
      __syncthreads();
      if (tid_call == 0)
        __hip_atomic_store(a.flag_b + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(5, __builtin_amdgcn_s_memrealtime());

    // ---- C: one row of one group over all slices ------------------------------------------
    if (c < n_wg && slice < rt + 2) {
      const int row = slice;
      double* out = a.out + ((size_t)c * (rt + 2) + row) * 64;
      if (row < rt) {
        // a wave = eight slices: the flags of the 32 workgroups it reads from, then their
        // partial sums, all loads in flight together
        const int first = wave * 8;
        {
          const int from = 4 * first + lane_call;
          const bool needed = lane_call < 32 && (from >> 2) < a.n_slices &&
                              (n_wg == 1 ? true : n_wg == 2 ? (lane_call & 1) == c : (lane_call & 3) == c);
          const unsigned long long since = __builtin_amdgcn_s_memrealtime();
          for (;;) {
            const unsigned long long flag =
                needed ? __hip_atomic_load(a.flag_b + from, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : epoch;
            if (__builtin_amdgcn_ballot_w64(flag != epoch) == 0) break;
            if (timed_out(since)) {
              s_abort = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const int n_parts = n_wg == 1 ? 4 : n_wg == 2 ? 2 : 1;
        double part[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int s = first + u;
            const int which = n_wg == 1 ? v : n_wg == 2 ? c + 2 * v : c;
            part[u][v] =
                s < a.n_slices && v < n_parts && !(a.skip & 8)
                    ? ens::load_agent(a.partial + ((size_t)(4 * s + which) * rt + row) * 64 + lane_call)
                    : 0.0;
          }
        }
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          double value;
          if (n_wg == 1) value = (part[u][0] + part[u][1]) + (part[u][2] + part[u][3]);
          else if (n_wg == 2) value = part[u][0] + part[u][1];
          else value = part[u][0];
          if (first + u < a.n_slices) sum = u == 0 ? value : sum + value;
        }
        area[wave * 64 + lane_call] = sum;
        __syncthreads();
        if (s_abort) break;
        if (wave == 0) {
          double each[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) each[w] = area[w * 64 + lane_call];
          double total = each[0];
          const int n_waves = (a.n_slices + 7) >> 3;
#pragma unroll
          for (int w = 1; w < 8; ++w)
            if (w < n_waves) total += each[w];
          ens::store_host(out + lane_call, total);
        }
      } else if (wave == 0) {
        ens::store_host(out + lane_call, dens_lds[(a.n_bins + row - rt) * kEnsembleDensPad + lane_call]);
      }
      if (wave == 0) {
        ens::wait_stores();
        if (lane_call == 0)
          __hip_atomic_store(a.done + c * (rt + 2) + row, epoch, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();    // (area and dens_lds are free again)
    serving = epoch + 1;
    t_last = __builtin_amdgcn_s_memrealtime();
    stamp(7, t_last);
  }
  // (with the number of the first call this workgroup has NOT served: one that leaves -- idle --
  // behind its part of a call that slower workgroups are still working on has not failed it)
  if (tid == 0) {
    __threadfence_system();
    __hip_atomic_store(a.exited + b, (a.launch_id << 40) | serving, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#endif

// ---- float32 variant for tables with many correlation-function bins -------------------
//
// BASELINE configs[4]: R = 760 (rp x pi), G ~ 200.  With hundreds of r values the
// contraction is a GEMM with a wide N: out[r][draw] += T[r][entry] * w[entry][draw], and
// f32 inputs may use the matrix cores: v_mfma_f32_32x32x2_f32 computes a 32 (r) x 32
// (draws) tile over 2 entries in 64 cycles -- the f32 vector rate, but with one table
// float and one weight float per lane per instruction instead of three VALU operands,
// and the VALU stays free to form the weights.  A wave owns 64 draws (two 32-draw
// accumulator tiles) x 32 r values and walks its chunk of entries 8 at a time:
//   one global_load_dwordx4: the table values of the 8 entries (4 MFMA k-steps),
//   one global_load_dwordx4: the bin pairs (i, j) of its 4 entries (lane half = k),
//   per k-step and draw half: two ds_read_b32 (n_i, n_j, float in LDS), one v_mul_f32,
//   one MFMA.
// Table layout: [r tile][block of 8 entries][k (2)][r (32)][k-step (4)] floats, so that
// lane l = k * 32 + r reads its 4 k-steps as one 16-byte vector.  Results are accumulated
// in f32; the partial sums leave the kernel as f64.  Stated tolerance 1e-5 relative.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool INTERP>
__global__ __launch_bounds__(512) void contract_f32_kernel(ContractArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  int tile, slab;
  if (a.xcd_map) {   // full chip (8 XCDs) and at least 8 draw tiles
    const int xcd = blockIdx.x & 7;
    const int rest = blockIdx.x >> 3;
    tile = (rest / a.n_slabs) * 8 + xcd;
    slab = rest % a.n_slabs;
  } else {   // fewer draw tiles than XCDs, or a partitioned device: plain order
    tile = blockIdx.x / a.n_slabs;
    slab = blockIdx.x % a.n_slabs;
  }
  if (tile >= a.n_tiles) return;
  const int64_t col = (int64_t)tile * kLanes;
  constexpr bool interp = INTERP;
  const int k_splits = interp ? a.k_splits : 1;
  const Group group = a.groups[slab / k_splits];
  const int n_rows_j = group.j_hi - group.j_lo;
  const int n_rows = n_rows_j + (group.i_hi - group.i_lo);
  int k_begin = 0, k_end = 1;
  if (interp) {
    const int split = slab % k_splits;
    k_begin = (int)((int64_t)a.n_tables * split / k_splits);
    k_end = (int)((int64_t)a.n_tables * (split + 1) / k_splits);
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }

  int staged_class = -1;
  for (int k = k_begin; k < k_end; ++k) {
    // interpolator: the block loops over the tables of its split (as contract_mfma_kernel)
    const int density_class = interp ? a.table_class[k] : 0;
    if (density_class != staged_class) {
      if (staged_class >= 0) __syncthreads();
      // stage the density rows as float
      gl_f64 src = (gl_f64)(interp ? a.nbufs[density_class] : a.nbuf) + col;
      for (int id = threadIdx.x; id < n_rows * kLanes; id += blockDim.x) {
        const int row = id >> 6;
        const int bin = row < n_rows_j ? group.j_lo + row : group.i_lo + row - n_rows_j;
        ldsf[id] = (float)src[(int64_t)bin * a.ldb + (id & 63)];
      }
      __syncthreads();
      staged_class = density_class;
    }

    if (wave < group.n_chunks) {
      const Chunk chunk = a.chunks[group.chunk_begin + wave];
      const int n_blocks = (chunk.q_end - chunk.q_begin) / kF32Block;
      typedef const __attribute__((address_space(1))) f32x4* gl_f32x4;
      typedef const __attribute__((address_space(1))) i32x4* gl_i32x4;
      gl_f32x4 table =
          (gl_f32x4)((const float*)(interp ? (const void*)a.tables[k] : a.table) +
                     ((int64_t)blockIdx.z * a.n_positions + chunk.q_begin) * kF32Tile) +
          lane;
      gl_i32x4 pairs = (gl_i32x4)(a.pos_ij + chunk.q_begin) + (lane >> 5);
      const int draw = lane & 31;
      const bool auto_mode = a.mode == 0;
      // spline weight / pair-weight norm of this table for the lane's two draws
      const float scale0 = interp ? (float)a.coef[(int64_t)k * a.ldb + col + draw] : 1.0f;
      const float scale1 =
          interp ? (float)a.coef[(int64_t)k * a.ldb + col + draw + 32] : 1.0f;
      f32x4 ta = table[0];
      i32x4 pa = pairs[0];
      for (int blk = 0; blk < n_blocks; ++blk) {
        const int next = blk + 1 < n_blocks ? blk + 1 : blk;
        const f32x4 tn = table[(int64_t)next * 64];
        const i32x4 pn = pairs[(int64_t)next * 2];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int jj = (pa[p] & 0xffff) - group.j_lo;
          const int ii = (pa[p] >> 16) + group.i_shift;
          const float* rj = ldsf + jj * kLanes + draw;
          float w0 = rj[0], w1 = rj[32];
          if (auto_mode) {
            const float* ri = ldsf + ii * kLanes + draw;
            w0 *= ri[0];
            w1 *= ri[32];
          }
          if (interp) {
            w0 *= scale0;
            w1 *= scale1;
          }
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[p], w0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[p], w1, acc1, 0, 0, 0);
        }
        ta = tn;
        pa = pn;
      }
    }
  }
  __syncthreads();

  // tree reduction over the waves (all of one segment), then wave 0 writes doubles
  int span = 1;
  while (span < n_waves) span <<= 1;
  for (int half = span >> 1; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* slot = ldsf + (wave - half) * 32 * kLanes + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        slot[r * kLanes] = acc0[r];
        slot[(16 + r) * kLanes] = acc1[r];
      }
    }
    __syncthreads();
    if (wave < half && wave + half < n_waves) {
      const float* slot = ldsf + wave * 32 * kLanes + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += slot[r * kLanes];
        acc1[r] += slot[(16 + r) * kLanes];
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
    // D layout of the 32x32 MFMA: column (draw) = lane & 31,
    // row (r) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    double* out = a.partial +
                  ((int64_t)slab * a.r_stride + (int64_t)blockIdx.z * kF32Tile) * a.ldb +
                  col + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      out[(int64_t)r * a.ldb] = (double)acc0[reg];
      out[(int64_t)r * a.ldb + 32] = (double)acc1[reg];
    }
  }
}

// Sum the per-group partials in fixed order, divide by the total pair weight
// (tabcorr.py:646-649, 653-655: sum(ngal_sq) = (sum ngal)^2 in mode auto) and
// write the results in the reference's output order.  One block per draw tile;
// reads are coalesced over draws, the transposition to the draw-major output
// goes through LDS.
#ifdef TC_UNIT_QUAD   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(1024) void finalize_kernel(FinalizeArgs a) {
  __shared__ double tile[kFinalizeRows][kLanes + 1];
  __shared__ double part_sum[16][kLanes];
  __shared__ double norm_inv[kLanes];
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t col = (int64_t)blockIdx.x * kLanes;
  const int64_t n_valid = a.n_draws - col < kLanes ? a.n_draws - col : kLanes;

  if (wave == 0 && a.ngal_part == nullptr) norm_inv[lane] = 1.0;
  if (wave == 0 && a.ngal_part != nullptr) {
    double n_cen = 0.0, n_sat = 0.0;
    for (int p = 0; p < a.n_ngal_parts; ++p) {
      n_cen += a.ngal_part[((int64_t)p * 2 + 0) * a.ldb + col + lane];
      n_sat += a.ngal_part[((int64_t)p * 2 + 1) * a.ldb + col + lane];
    }
    const double total = n_cen + n_sat;
    norm_inv[lane] = a.mode == 0 ? total * total : total;
    if (lane < n_valid && blockIdx.y == 0) {
      if (a.n_comp == 1) {
        a.ngal[col + lane] = total;
      } else {
        a.ngal[2 * (col + lane)] = n_cen;
        a.ngal[2 * (col + lane) + 1] = n_sat;
      }
    }
  }
  __syncthreads();
  // (separated by galaxy type a non-finite sum poisons every component: see
  // finalize_quad_kernel)
  const double norm = a.n_comp > 1 && !(fabs(norm_inv[lane]) <= 1.79769313486231570815e308)
                          ? __builtin_nan("")
                          : norm_inv[lane];

  const int n_rows = a.n_comp * a.n_r;
  const int n_waves = blockDim.x >> 6;
  const int n_slabs = a.n_groups * a.k_splits;
  // rows of this block (grid.y row blocks; one for large batches)
  const int rows_per_block = (n_rows + gridDim.y - 1) / gridDim.y;
  const int row_begin = blockIdx.y * rows_per_block;
  const int row_end = row_begin + rows_per_block < n_rows ? row_begin + rows_per_block : n_rows;
  const int64_t slab = (int64_t)a.r_stride * a.ldb;
  // sum of the slabs [g_begin, g_end) of one row, eight independent loads in flight, the
  // additions in slab order
  auto sum_slabs = [&](int row, int g_begin, int g_end) {
    const int c = row / a.n_r, r = row % a.n_r;
    const double* src = a.partial + (int64_t)r * a.ldb + col + lane;
    double sum = 0.0;
    for (int g0 = g_begin; g0 < g_end; g0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + u;
        const bool use = g < g_end &&
                         (a.n_comp == 1 || a.groups[g / a.k_splits].component == c);
        v[u] = use ? src[g * slab] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    return sum;
  };
  for (int row0 = row_begin; row0 < row_end; row0 += kFinalizeRows) {
    const int rows = row_end - row0 < kFinalizeRows ? row_end - row0 : kFinalizeRows;
    if (rows * 2 <= n_waves) {
      // few rows (small batches, split over row blocks): several waves per row, each
      // over a contiguous range of slabs, combined in fixed order
      const int parts = n_waves / rows;
      const int rr = wave / parts, part = wave % parts;
      if (rr < rows) {
        const int g_begin = (int)((int64_t)n_slabs * part / parts);
        const int g_end = (int)((int64_t)n_slabs * (part + 1) / parts);
        part_sum[wave][lane] = sum_slabs(row0 + rr, g_begin, g_end);
      }
      __syncthreads();
      if (wave < rows) {
        double sum = 0.0;
        for (int p = 0; p < parts; ++p) sum += part_sum[wave * parts + p][lane];
        tile[wave][lane] = sum / norm;
      }
    } else {
      for (int rr = wave; rr < rows; rr += n_waves)
        tile[rr][lane] = sum_slabs(row0 + rr, 0, n_slabs) / norm;
    }
    __syncthreads();
    // out[(col + d) * n_rows + row0 + rr]
    for (int idx = threadIdx.x; idx < rows * kLanes; idx += blockDim.x) {
      const int d = idx / rows, rr = idx % rows;
      if (d < n_valid)
        a.xi[(col + d) * (int64_t)n_rows + row0 + rr] = tile[rr][d];
    }
    __syncthreads();
  }
}
#endif

// Per draw: the tensor-product spline weight of every table at the draw's extra
// parameters (interpolator.py:275-331; segment search as np.digitize with the right
// edge special-cased, out-of-range clamped to the outermost segment), divided by the
// table's total pair weight so that the contraction can accumulate all tables into
// one sum (interpolation is linear in the per-table xi); and the interpolated ngal.
#ifdef TC_UNIT_QUAD   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(64) void interp_coef_kernel(InterpArgs a) {
  __shared__ double weight[kMaxInterpDim][kMaxInterpAxis][kLanes];
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kLanes + lane;
  const int64_t b = b0 < a.n_draws ? b0 : a.n_draws - 1;
  for (int d = 0; d < a.n_dim; ++d) {
    const int n = a.n_axis[d];
    const double* xp = a.xp + a.axis_offset[d];
    const double x = a.x[b * a.n_dim + d];
    int seg = -1;
    for (int i = 0; i < n; ++i) seg += xp[i] <= x ? 1 : 0;   // np.digitize(x, xp) - 1
    if (x == xp[n - 1]) seg = n - 2;
    seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
    const double* m = a.a + a.a_offset[d] + (int64_t)seg * 4 * n;
    const double x2 = x * x, x3 = x2 * x;
    for (int j = 0; j < n; ++j)
      weight[d][j][lane] = m[j] + m[n + j] * x + m[2 * n + j] * x2 + m[3 * n + j] * x3;
  }
  double n_cen = 0.0, n_sat = 0.0;
  int summed_class = -1;
  double cen = 0.0, sat = 0.0;
  for (int k = 0; k < a.n_tables; ++k) {
    double c = 1.0;
    for (int d = 0; d < a.n_dim; ++d)
      c *= weight[d][a.table_node[k * a.n_dim + d]][lane];
    // the densities of a class of identical halo tables are summed once (tables of one
    // class are usually all of them)
    const int density_class = a.table_class[k];
    if (density_class != summed_class) {
      const double* parts = a.ngal_parts[density_class];
      cen = 0.0;
      sat = 0.0;
      for (int p = 0; p < a.n_ngal_parts; ++p) {
        cen += parts[((int64_t)p * 2 + 0) * a.ldb + b0];
        sat += parts[((int64_t)p * 2 + 1) * a.ldb + b0];
      }
      summed_class = density_class;
    }
    const double total = cen + sat;
    a.coef[(int64_t)k * a.ldb + b0] = c / (a.mode == 0 ? total * total : total);
    n_cen += c * cen;
    n_sat += c * sat;
  }
  if (b0 < a.n_draws) {
    if (a.separate) {
      a.ngal[2 * b0] = n_cen;
      a.ngal[2 * b0 + 1] = n_sat;
    } else {
      a.ngal[b0] = n_cen + n_sat;
    }
  }
}
#endif

// The same for a handful of draws (the un-batched Interpolator.predict of an MCMC step):
// one block per draw, the LANES over the tables, instead of one lane per draw looping over
// all tables -- 18 -> ~5 us for a 5 x 5 grid.  Table weights land in LDS and are summed
// in table order by one thread (deterministic).
constexpr int kCoefSmallTables = 1024;

#ifdef TC_UNIT_QUAD   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(64) void interp_coef_small_kernel(InterpArgs a) {
  __shared__ double weight[kMaxInterpDim][kMaxInterpAxis];
  __shared__ double part_cen[kCoefSmallTables], part_sat[kCoefSmallTables];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  for (int d = 0; d < a.n_dim; ++d) {
    const int n = a.n_axis[d];
    const double* xp = a.xp + a.axis_offset[d];
    const double x = a.x[b * a.n_dim + d];
    int seg = -1;
    for (int i = 0; i < n; ++i) seg += xp[i] <= x ? 1 : 0;   // np.digitize(x, xp) - 1
    if (x == xp[n - 1]) seg = n - 2;
    seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
    const double* m = a.a + a.a_offset[d] + (int64_t)seg * 4 * n;
    const double x2 = x * x, x3 = x2 * x;
    if (lane < n)
      weight[d][lane] = m[lane] + m[n + lane] * x + m[2 * n + lane] * x2 + m[3 * n + lane] * x3;
  }
  __syncthreads();
  for (int k = lane; k < a.n_tables; k += kLanes) {
    double c = 1.0;
    for (int d = 0; d < a.n_dim; ++d) c *= weight[d][a.table_node[k * a.n_dim + d]];
    const double* parts = a.ngal_parts[a.table_class[k]];
    double cen = 0.0, sat = 0.0;
    for (int p = 0; p < a.n_ngal_parts; ++p) {
      cen += parts[((int64_t)p * 2 + 0) * a.ldb + b];
      sat += parts[((int64_t)p * 2 + 1) * a.ldb + b];
    }
    const double total = cen + sat;
    a.coef[(int64_t)k * a.ldb + b] = c / (a.mode == 0 ? total * total : total);
    part_cen[k] = c * cen;
    part_sat[k] = c * sat;
  }
  __syncthreads();
  if (lane == 0) {
    double n_cen = 0.0, n_sat = 0.0;
    for (int k = 0; k < a.n_tables; ++k) {
      n_cen += part_cen[k];
      n_sat += part_sat[k];
    }
    if (a.separate) {
      a.ngal[2 * b] = n_cen;
      a.ngal[2 * b + 1] = n_sat;
    } else {
      a.ngal[b] = n_cen + n_sat;
    }
  }
}
#endif

// Gaussian likelihood fused behind predict(): chi2[b] = (xi_b - d)^T P (xi_b - d).  This is
// the step every MCMC likelihood performs on the host right after predict() (README.md:7 of
// the reference); doing it here leaves one double per draw to copy back / gather.  32 lanes
// per draw (two draws per wave): lane i forms delta_i (sum_j P_ij delta_j) for the rows
// i = lane, lane + 32, ..., a fixed-order butterfly adds the 32 lanes.  The deviations of the
// block's draws and (for up to 64 r values) the precision matrix sit in LDS.  (The first
// version used one lane per draw: 40 workgroups for 10^4 draws, 30-50 us of a chain that the
// next batch's kernels wait for.)
constexpr int kChi2DrawsPerBlock = 8;      // 256 threads; fewer when n_r is large (LDS)
constexpr int kChi2LdsMatrix = 64;         // largest n_r whose precision matrix is staged
constexpr int kChi2LdsBytes = 48 * 1024;

#ifdef TC_UNIT_QUAD   // (emitted by the unit that launches it, which defines the macro)
static __global__ __launch_bounds__(256) void chi2_kernel(const double* xi, int64_t n_draws,
                                                   int n_r, const double* data,
                                                   const double* precision,
                                                   double* chi2) {
  extern __shared__ double chi2_lds[];     // (draws per block, n_r) deviations [+ matrix]
  const int per_block = blockDim.x >> 5;
  double* delta = chi2_lds;
  double* matrix = chi2_lds + per_block * n_r;
  const bool staged = n_r <= kChi2LdsMatrix;
  const int64_t first = (int64_t)blockIdx.x * per_block;
  for (int idx = threadIdx.x; idx < per_block * n_r; idx += blockDim.x) {
    const int64_t b = first + idx / n_r;
    delta[idx] = b < n_draws ? xi[b * n_r + idx % n_r] - data[idx % n_r] : 0.0;
  }
  if (staged)
    for (int idx = threadIdx.x; idx < n_r * n_r; idx += blockDim.x) matrix[idx] = precision[idx];
  __syncthreads();
  const int local = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const double* d = delta + local * n_r;
  const double* p = staged ? matrix : precision;
  // lanes over the columns (consecutive addresses), rows in order: sum_j d_j sum_i d_i P_ij
  double total = 0.0;
  for (int j = lane; j < n_r; j += 32) {
    double inner = 0.0;
    for (int i = 0; i < n_r; ++i) inner = fma(d[i], p[(int64_t)i * n_r + j], inner);
    total = fma(d[j], inner, total);
  }
#pragma unroll
  for (int offset = 16; offset >= 1; offset >>= 1) total += __shfl_xor(total, offset, 32);
  if (lane == 0 && first + local < n_draws) chi2[first + local] = total;
}
#endif

}  // namespace tc
