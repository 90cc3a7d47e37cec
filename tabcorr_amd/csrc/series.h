// Moment expansion of a central bin's Gauss-Legendre sum (host + device).
//
// tabcorr.py:556-578 averages <N_cen>(M) = (1 + erf((log M - logMmin) / sigma)) / 2 over the
// nodes of a bin with the weights W_k.  The nodes lie within half a bin width of the bin's
// centre c, so with z0 = (c - logMmin) / sigma and d_k = log M_k - c
//
//   sum_k W_k erf(z0 + d_k / sigma)
//       = m_0 erf(z0) + 2/sqrt(pi) exp(-z0^2) sum_{n >= 1} (-1)^(n-1) H_{n-1}(z0) sigma^-n m_n / n!
//
// (Taylor series of erf around z0; H: physicists' Hermite polynomials) with the bin's moments
// m_n = sum_k W_k d_k^n, which do not depend on the draw: the SAME sum over the nodes,
// re-ordered, truncated where the tail is below 1e-16 -- Cramer's bound
// |H_n(z) exp(-z^2 / 2)| <= 1.09 2^(n/2) sqrt(n!) gives the number of terms from
// h = (half bin width) / sigma alone.  One erf, its derivative and four instructions per term
// (the recurrence p_n = H_n(z0) sigma^-(n+1): p_(n+1) = (2 z0 / sigma) p_n - (2 n / sigma^2)
// p_(n-1), and one FMA with the moment as a scalar operand) replace n_gauss erf evaluations:
// 8 terms serve h <= 0.027, 12 h <= 0.11, 16 h <= 0.24, 20 h <= 0.39, 24 h <= 0.55; beyond
// that (sigma below a tenth of a bin width ...) the node loop runs as before.
#pragma once

#include <cstdint>

#include "fastmath.h"

namespace tc {
namespace series {

constexpr int kMaxTerms = 24;
constexpr int kSteps = 5;                    // 8, 12, 16, 20, 24 terms
constexpr int kFirst = 4;                    // the moments start at consts[kFirst]: blocks of
                                             // four, 32 bytes each, 32-byte aligned
constexpr int kStride = kFirst + kMaxTerms;  // per bin: centre, (3 unused), M_1 .. M_24
// (round 5's first loops requested the block behind the one they used, also behind the last:
// arrays of constants still end with kPad more doubles)
constexpr int kPad = 4;
constexpr int kThresholds = 8;               // int32 per bin (kSteps used)
constexpr double kTolerance = 1e-16;

// Largest h = (half bin width) / sigma for which `n_terms` terms leave a tail below kTolerance.
double h_max(int n_terms);

// Per-bin constants from the bin's nodes (log10 M_k) and normalised weights: consts[0] = centre,
// consts[kFirst + n - 1] = (-1)^(n-1) m_n / n!; thresholds[s] = high dword of the largest
// |1 / sigma| for which 8 + 4 s terms suffice (0: never).
void bin_consts(int n_gauss, const double* log_m, const double* weight, double log_min,
                double log_max, double* consts, int32_t* thresholds);

// Number of terms for a draw whose |1 / sigma| has the high dword `inv_sigma_hi` (0: the
// expansion does not apply, run the node loop).
//
// In the kernels a LANE adds the terms its OWN draw asks for, so that a draw's result depends
// on nothing but the draw -- the same bits wherever it sits in whatever batch: a lane takes
// the expansion when eligible(thresholds, its key) and leaves the loop of passes (four terms
// each) when its key is below the next pass's threshold; the wave's loop ends with its last
// lane.  The thresholds increase with the number of terms (0: never); keys are non-negative.
template <typename IntPtr>
TC_HD int terms_for(IntPtr thresholds, int inv_sigma_hi) {
  int n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = kSteps - 1; s >= 0; --s)
    if (inv_sigma_hi < thresholds[s]) n = 8 + 4 * s;
  return n;
}

template <typename IntPtr>
TC_HD bool eligible(IntPtr thresholds, int inv_sigma_hi) {
  return inv_sigma_hi < thresholds[kSteps - 1];
}

// The loops below: passes of four terms, NOT unrolled (the unrolled forms take 135-205 vector
// registers in the kernels).  The moments of a pass are ONE aligned 32-byte scalar load through
// a pointer that advances by a block per pass, requested at the top of the pass (the other
// waves of the SIMD cover its latency; requested a pass ahead, the rotation of the two register
// sets cost eight scalar moves per pass of a pair of bins -- 39.4 -> 39.3 us per 10^4 draws of
// the headline table, 75.8 -> 75.1 for the AbacusSummit interpolator); the thresholds are eight
// scalar registers from one load before the first pass, from which a lane counts its passes
// (`passes`): the loop ends with the wave's last lane.  Round 4's loop recomputed its
// addresses from the pass number -- clamped, so that eight separate 8-byte loads and three scalar
// instructions per vector instruction came out.
struct f64x4_t {
  double v[4];
  TC_HD double operator[](int k) const { return v[k]; }
};

// Four consecutive constants from consts[first + 4 block] on (first + 4 block a multiple of 4:
// one 32-byte load on the device).
template <typename Ptr>
TC_HD f64x4_t load_four(Ptr consts, int first, int block) {
  f64x4_t m;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double __attribute__((ext_vector_type(4))) f64x4v;
  typedef const __attribute__((address_space(4))) f64x4v* sc_f64x4v;
  const f64x4v value = *(sc_f64x4v)(consts + first + 4 * block);
  m.v[0] = value.x;
  m.v[1] = value.y;
  m.v[2] = value.z;
  m.v[3] = value.w;
#else
  m.v[0] = consts[first + 4 * block];
  m.v[1] = consts[first + 1 + 4 * block];
  m.v[2] = consts[first + 2 + 4 * block];
  m.v[3] = consts[first + 3 + 4 * block];
#endif
  return m;
}

// The bin's thresholds (kThresholds = 8 int32: one 32-byte load on the device).
struct Thresholds {
  int v[kThresholds];
};
template <typename IntPtr>
TC_HD Thresholds load_thresholds(IntPtr thresholds) {
  Thresholds t;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int __attribute__((ext_vector_type(8))) i32x8v;
  typedef const __attribute__((address_space(4))) i32x8v* sc_i32x8v;
  const i32x8v value = *(sc_i32x8v)thresholds;
#pragma unroll
  for (int i = 0; i < kThresholds; ++i) t.v[i] = value[i];
#else
  for (int i = 0; i < kThresholds; ++i) t.v[i] = thresholds[i];
#endif
  return t;
}

// Passes of an ELIGIBLE draw with the key `hi`: `fewest`, plus one per threshold it reaches
// (= terms_for(...) / 4).
template <int STEPS>
TC_HD int passes(const Thresholds& limit, int hi, int fewest) {
  int n = fewest;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = 0; s < STEPS - 1; ++s) n += hi >= limit.v[s] ? 1 : 0;
  return n;
}

// sum_k W_k erf((log M_k - log_m_min) inv_sigma) of the bin whose constants are `consts`, by
// the terms of the expansion an ELIGIBLE draw needs; m0 = sum_k W_k.  Uniform steps from
// (p_-1, p_0) = (0, 1 / sigma): term n adds p_(n-1) M_n, then p_n = a p_(n-1) + (n - 1) b p_(n-2).
template <typename Ptr, typename IntPtr>
TC_HD double central_sum(const double* table, const fm::Consts& kc, double log_m_min,
                         double inv_sigma, Ptr consts, double m0, IntPtr thresholds,
                         int inv_sigma_hi) {
  // (the recurrence runs on z0 clamped to [-6, 6]: beyond, on the plateaus, g0 = 0 and the sum
  // is m0 erf(z0) = -+m0 whatever the terms are -- but they must stay finite: with an infinite
  // or huge logMmin the unclamped recurrence overflows and 0 x inf would be NaN where the node
  // loop and the reference give -+1, N_cen = 0 or 1)
  const int n_blocks = passes<kSteps>(load_thresholds(thresholds), inv_sigma_hi, 2);
  Ptr block_ptr = consts + kFirst;
  double g0, z0;
  const double e = fm::erf_gauss_fast(table, kc, (consts[0] - log_m_min) * inv_sigma, &g0, &z0);
  const double a = 2.0 * z0 * inv_sigma, b = -2.0 * inv_sigma * inv_sigma;
  double p_prev = 0.0, p = inv_sigma, nb = -b, sum = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int block = 0; block < n_blocks; ++block) {
    const f64x4_t m = load_four(block_ptr, 0, 0);
    block_ptr += 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
      sum = fma(p, m[k], sum);
      nb += b;
      const double next = fma(a, p, nb * p_prev);
      p_prev = p;
      p = next;
    }
  }
  return fma(g0, sum, m0 * e);
}

// The same for two bins with the same nodes (one group: same centre, different moments): the
// recurrence once, one FMA per term and bin.
template <typename Ptr, typename IntPtr>
TC_HD void central_sum_pair(const double* table, const fm::Consts& kc, double log_m_min,
                            double inv_sigma, Ptr consts_i, Ptr consts_j, double m0_i,
                            double m0_j, IntPtr thresholds, int inv_sigma_hi, double* out_i,
                            double* out_j) {
  const int n_blocks = passes<kSteps>(load_thresholds(thresholds), inv_sigma_hi, 2);
  Ptr ptr_i = consts_i + kFirst, ptr_j = consts_j + kFirst;
  double g0, z0;
  const double e =
      fm::erf_gauss_fast(table, kc, (consts_i[0] - log_m_min) * inv_sigma, &g0, &z0);
  const double a = 2.0 * z0 * inv_sigma, b = -2.0 * inv_sigma * inv_sigma;
  double p_prev = 0.0, p = inv_sigma, nb = -b, sum_i = 0.0, sum_j = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int block = 0; block < n_blocks; ++block) {
    const f64x4_t mi = load_four(ptr_i, 0, 0), mj = load_four(ptr_j, 0, 0);
    ptr_i += 4;
    ptr_j += 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
      sum_i = fma(p, mi[k], sum_i);
      sum_j = fma(p, mj[k], sum_j);
      nb += b;
      const double next = fma(a, p, nb * p_prev);
      p_prev = p;
      p = next;
    }
  }
  *out_i = fma(g0, sum_i, m0_i * e);
  *out_j = fma(g0, sum_j, m0_j * e);
}

// ---- satellites ---------------------------------------------------------------------------
//
// <N_sat>(M) = ((M - M0) / M1)^alpha above M0 (Zheng et al. 2007, eq. 3; tabcorr.py:560-563).
// With the bin's reference mass Mc (its geometric centre), y_k = M_k / Mc - 1 and
// eps = Mc / (Mc - M0):
//
//   sum_k W_k (M_k - M0)^alpha = (Mc - M0)^alpha sum_n C(alpha, n) eps^n mu_n,   mu_n = sum_k W_k y_k^n
//
// (binomial series of (1 + eps y)^alpha; the moments mu_n do not depend on the draw): one
// log2 + exp2 for (Mc - M0)^alpha / M1^alpha and three instructions per term -- d_(n+1) = d_n eps
// (alpha - n) with the 1 / n! folded into the moments -- instead of ten log2 + exp2.  For
// 0 <= alpha <= 4 the coefficients stay below 16, so n terms leave a tail below 1e-16 when
// r = eps max|y_k| <= 0.047 (12 terms), 0.097 (16), 0.15 (20), 0.20 (24), 0.25 (28), 0.30 (32):
// bins far enough above M0 (M0 <= Mc (1 - max|y| / r)); the others run their node loop.
namespace sat {

constexpr int kMaxTerms = 32;
constexpr int kSteps = 6;                    // 12, 16, ..., 32 terms
constexpr int kFirst = 4;                    // mu_1 sits at consts[kFirst]: 32-byte blocks of four
constexpr int kStride = kFirst + kMaxTerms;  // per bin: Mc, max|y|, (unused), mu_0, mu_1 / 1! ..
                                             // mu_32 / 32!
constexpr int kThresholds = 8;               // int32 per bin (kSteps used)
// thresholds[kShortest] = the first non-zero threshold: the largest M0 for which the shortest
// expansion that serves the bin at all suffices.  Kernels that evaluate elsewhere what no
// expansion serves (predict_fused_kernel's deferred pairs) take in place only the draws below
// it -- a wave pays for its longest lane: bins 0.09 dex wide, 20 terms in place 38.2 us per
// 10^4 draws, up to 24: 38.4, 28: 38.5, 32: 38.6, the node loops in place 39.2.
constexpr int kShortest = 6;

// Largest r = eps max|y| for which `n_terms` terms leave a tail below kTolerance.
double r_max(int n_terms);

// Per-bin constants from the bin's node masses and normalised weights: consts[0] = Mc,
// consts[3] = mu_0, consts[kFirst + n - 1] = mu_n / n!; thresholds[s] = high dword of the
// largest M0 for which 12 + 4 s terms suffice (0: never).
void bin_consts(int n_gauss, const double* mass, const double* weight, double log_min,
                double log_max, double* consts, int32_t* thresholds);

template <typename IntPtr>
TC_HD int terms_for(IntPtr thresholds, int m0_hi) {
  int n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = kSteps - 1; s >= 0; --s)
    if (m0_hi < thresholds[s]) n = 12 + 4 * s;
  return n;
}

// Per lane, as for the centrals: does the expansion apply to a draw with this key, and does
// the draw add the terms of pass `block` (the first three passes are the fewest any draw takes)?
template <typename IntPtr>
TC_HD bool eligible(IntPtr thresholds, int m0_hi) {
  return m0_hi < thresholds[kSteps - 1];
}

// 1 / x to rounding (hardware estimate + two Newton steps on the device).
TC_HD double reciprocal(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}

// sum_k W_k (1 + eps y_k)^alpha by the terms an ELIGIBLE draw needs + the n = 0 term (the
// unrolled passes of central_sum; the first three are the fewest any draw takes).
template <typename Ptr, typename IntPtr>
TC_HD double binomial_sum(Ptr consts, double eps, double alpha, IntPtr thresholds, int m0_hi) {
  const int n_blocks = series::passes<kSteps>(series::load_thresholds(thresholds), m0_hi, 3);
  Ptr block_ptr = consts + kFirst;
  double d = 1.0, g = eps * alpha, sum = consts[3];         // n = 0: mu_0
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int block = 0; block < n_blocks; ++block) {
    const f64x4_t cur = load_four(block_ptr, 0, 0);
    block_ptr += 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
      d *= g;                       // d_n = d_(n-1) eps (alpha - (n - 1))
      g -= eps;
      sum = fma(d, cur[k], sum);
    }
  }
  return sum;
}

// The same for two bins with the same nodes (same Mc, eps): the coefficients once.
template <typename Ptr, typename IntPtr>
TC_HD void binomial_sum_pair(Ptr consts_i, Ptr consts_j, double eps, double alpha,
                             IntPtr thresholds, int m0_hi, double* out_i, double* out_j) {
  const int n_blocks = series::passes<kSteps>(series::load_thresholds(thresholds), m0_hi, 3);
  Ptr ptr_i = consts_i + kFirst, ptr_j = consts_j + kFirst;
  double d = 1.0, g = eps * alpha, sum_i = consts_i[3], sum_j = consts_j[3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int block = 0; block < n_blocks; ++block) {
    const f64x4_t cur_i = load_four(ptr_i, 0, 0), cur_j = load_four(ptr_j, 0, 0);
    ptr_i += 4;
    ptr_j += 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
      d *= g;
      g -= eps;
      sum_i = fma(d, cur_i[k], sum_i);
      sum_j = fma(d, cur_j[k], sum_j);
    }
  }
  *out_i = sum_i;
  *out_j = sum_j;
}

}  // namespace sat

// ---- one record per group of one or two bins (round 5) ---------------------------------------
//
// The loops above find a group's constants through a dozen arrays: per group of the reference's
// AbacusSummit tables a wave issued ~20 scalar loads on 5 + (number of passes) dependent levels
// (first member -> thresholds -> centre -> mu_0 -> the blocks of moments, each waited for where
// it is used) against ~250 vector instructions, from half a megabyte of constants that a 16 KB
// scalar cache does not hold: the occupations of predict_cross_fused_kernel were latency, not
// arithmetic.  For tables whose groups have at most two members (secondary-percentile halves,
// scripts/tabulate_snapshot.py:193, or no cut at all) everything a group's expansion reads sits in
// ONE record of kStride doubles at records + group x kStride, 64-byte aligned:
//
//   [0, 3)  int32 x 6: the group's thresholds (series.h: five / namespace sat: six)
//   [3]     centrals log10 M of the first node, satellites the larger of the end nodes' masses
//   [4]     centre (log10) / Mc          [5], [6]  m_0 / mu_0 of the two members
//   [7]     centrals log10 M of the last node
//   2 x (index of the first member) + (1 if there is a second): centrals in the sixth int32 of
//   [0, 3), satellites in the low half of [7]
//   [8 + 8 p, 16 + 8 p)  pass p: four moments of the first member, four of the second
//                        (a single member: its own twice)
//
// so that the head and the passes every eligible draw takes (two for centrals, three for
// satellites) are requested together, at addresses that depend on the group's number alone.
namespace record {
constexpr int kHead = 8;
constexpr int kBlock = 8;
constexpr int kStride = kHead + kBlock * (sat::kMaxTerms / 4);      // 72 doubles: nine lines
constexpr int kLow = 3, kCentre = 4, kFirstSum = 5, kHigh = 7;

struct f64x8_t {
  double v[8];
};

// Eight consecutive doubles (64-byte aligned: one 64-byte scalar load on the device).
template <typename Ptr>
TC_HD f64x8_t load_eight(Ptr p) {
  f64x8_t m;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double __attribute__((ext_vector_type(8))) f64x8v;
  typedef const __attribute__((address_space(4))) f64x8v* sc_f64x8v;
  const f64x8v value = *(sc_f64x8v)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) m.v[i] = value[i];
#else
  for (int i = 0; i < 8; ++i) m.v[i] = p[i];
#endif
  return m;
}

// A record's head and first three passes, requested TOGETHER and waited for once.  Left to the
// compiler the loads of the passes sink below the first branch that looks at the head (the
// plateau test), the third pass's below the test that a draw needs it: three dependent round
// trips to the L2 per group instead of one.
template <typename Ptr>
TC_HD void load_record(Ptr rec, f64x8_t& head, f64x8_t& b0, f64x8_t& b1, f64x8_t& b2) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double __attribute__((ext_vector_type(8))) f64x8v;
  f64x8v h, x0, x1, x2;
  asm volatile(
      "s_load_dwordx16 %0, %4, 0x0\n\t"
      "s_load_dwordx16 %1, %4, 0x40\n\t"
      "s_load_dwordx16 %2, %4, 0x80\n\t"
      "s_load_dwordx16 %3, %4, 0xc0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(h), "=&s"(x0), "=&s"(x1), "=&s"(x2)
      : "s"(rec));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    head.v[i] = h[i];
    b0.v[i] = x0[i];
    b1.v[i] = x1[i];
    b2.v[i] = x2[i];
  }
#else
  head = load_eight(rec);
  b0 = load_eight(rec + kHead);
  b1 = load_eight(rec + kHead + kBlock);
  b2 = load_eight(rec + kHead + 2 * kBlock);
#endif
}

// The thresholds out of a record's head.
TC_HD Thresholds thresholds_of(const f64x8_t& head) {
  Thresholds t;
  for (int i = 0; i < 3; ++i) {
    const unsigned long long bits = fm::bits_of(head.v[i]);
    t.v[2 * i] = (int)(unsigned)(bits & 0xffffffffull);
    t.v[2 * i + 1] = (int)(unsigned)(bits >> 32);
  }
  t.v[6] = t.v[7] = 0;
  return t;
}

// A group's record from its members' per-bin constants (series::bin_consts / sat::bin_consts of
// the first and the second member -- the same pointer twice for a single member --, their
// thresholds = the first member's, m_0 of the centrals = the members' sums of weights).
void group_record(bool central, const double* consts_i, const double* consts_j,
                  const int32_t* thresholds, double sum_i, double sum_j, const double* log_m,
                  const double* mass, int n_gauss, int first_member, bool two, double* out);

// 2 x first member + second: see above.
TC_HD int members_of(bool central, const Thresholds& limit, const f64x8_t& head) {
  return central ? limit.v[5] : fm::low_word(head.v[kHigh]);
}
}  // namespace record

// ... per CENTRAL bin, for predict_fused_kernel's instance that defers the centrals no expansion
// serves as well: 32 doubles = four 64-byte lines, ALL of which one batch of loads brings --
//   [0, 3)  int32 x 5: thresholds   [3] log10 M of the first node   [4] centre   [5] m_0
//   [7]     log10 M of the last node          [8 + n)  moment n + 1 of 24
namespace cen_record {
constexpr int kHead = 8;
constexpr int kStride = kHead + series::kMaxTerms;      // 32 doubles
constexpr int kLow = 3, kCentre = 4, kSum = 5, kHigh = 7;

template <typename Ptr>
TC_HD void load_record(Ptr rec, record::f64x8_t& head, record::f64x8_t& m0, record::f64x8_t& m1,
                       record::f64x8_t& m2) {
  record::load_record(rec, head, m0, m1, m2);     // (the same four 64-byte lines)
}

void bin_record(const double* consts, const int32_t* thresholds, double weight_sum,
                const double* log_m, int n_gauss, double* out);
}  // namespace cen_record

// ... and per satellite BIN, for predict_fused_kernel's deferring instance: kStride doubles,
// 64-byte aligned --
//   [0, 3)  int32 x 6: thresholds; [3] as int32 x 2: thresholds[kShortest], passes of the
//   shortest expansion     [4] Mc     [5] mu_0     [6] the larger of the end nodes' masses
//   [8 + 4 p, 12 + 4 p)    the four moments of pass p
// head and the first four passes: three 64-byte loads, one round trip.
namespace sat_record {
constexpr int kHead = 8;
constexpr int kBlock = 4;
constexpr int kStride = kHead + kBlock * (sat::kMaxTerms / 4);      // 40 doubles: five lines
constexpr int kLimit = 3, kCentre = 4, kSum = 5, kLargest = 6;

template <typename Ptr>
TC_HD void load_record(Ptr rec, record::f64x8_t& head, record::f64x8_t& first,
                       record::f64x8_t& second) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double __attribute__((ext_vector_type(8))) f64x8v;
  f64x8v h, x, y;
  asm volatile(
      "s_load_dwordx16 %0, %3, 0x0\n\t"
      "s_load_dwordx16 %1, %3, 0x40\n\t"
      "s_load_dwordx16 %2, %3, 0x80\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(h), "=&s"(x), "=&s"(y)
      : "s"(rec));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    head.v[i] = h[i];
    first.v[i] = x[i];
    second.v[i] = y[i];
  }
#else
  head = record::load_eight(rec);
  first = record::load_eight(rec + kHead);
  second = record::load_eight(rec + kHead + 8);
#endif
}

// A satellite bin's record from sat::bin_consts' output and its node masses.
void bin_record(const double* consts, const int32_t* thresholds, const double* mass, int n_gauss,
                double* out);
}  // namespace sat_record

}  // namespace series
}  // namespace tc
