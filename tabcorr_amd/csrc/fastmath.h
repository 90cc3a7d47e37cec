// Table-driven FP64 erf / log2 / exp2 for the occupation kernel.
//
// The stock device libm spends ~100 FP64 VALU instructions per erf() or pow() and
// its piecewise branches diverge across the 64 draws of a wave.  The occupation
// kernel evaluates G * n_gauss of each per draw, so these functions are replaced by
// branch-free table + short-polynomial forms.  Every function does ONE LDS gather (the
// per-lane row differs, so the gathers are bank-conflict prone: profiles/r01_notes.md
// measured the 4-gather erf of the first version LDS-bound) and rounds to the row index
// with the 1.5 * 2^52 trick (index in the low dword of the sum, no conversions):
//
//   erf(x)  : |x| clamped to 6; rows {erf(c), 2/sqrt(pi) exp(-c^2)} at c = i / 128; the
//             Taylor coefficients around c are Hermite polynomials in c evaluated on the
//             fly: erf(c + h) = E + G h (1 - c h + (2c^2-1)/3 h^2 + c(3-2c^2)/6 h^3 +
//             (4c^4-12c^2+3)/30 h^4), |h| <= 1/256, truncation 2e-16.
//   log2(y) : the top 8 mantissa bits straight from the bit pattern, mantissa and exponent
//             from the frexp instructions; 256 rows {2/c, log2 c - 1}; log2(1 + r) to degree
//             5 with |r| <= 2^-9.
//   exp2(z) : z = k / 256 + r; 256 rows 2^(j/256); 2^r to degree 4, |r| <= 2^-9.
//
// Accurate to a few 1e-16 (erf, exp2 relative; log2 absolute) -- far inside the 1e-10
// parity budget.  The same inline functions run on the host (tc_debug_fastmath) so that
// their accuracy is tested without a GPU.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIP__)
#define TC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TC_HD inline
#endif

namespace tc {
namespace fm {

constexpr int kErfRows = 769;       // c_i = i / 128, i = 0 .. 768
constexpr int kLogRows = 256;
constexpr int kExpRows = 256;
constexpr int kErfOffset = 0;
constexpr int kLogOffset = 2 * kErfRows;                     // 1538 (16-byte aligned)
constexpr int kExpOffset = kLogOffset + 2 * kLogRows;        // 2050
constexpr int kTableDoubles = kExpOffset + kExpRows;         // 2306 doubles = 18.4 KB

constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52: ulp 1, low dword = integer
constexpr double kMagicErf = kMagic / 128.0;    // 1.5 * 2^45: ulp 1 / 128
constexpr double kLog2Of10Hi = 3.321928094887362182;      // fl(log2 10)
constexpr double kLog2Of10Lo = 1.661617516973592e-16;     // log2 10 - kLog2Of10Hi
constexpr double kLn2 = 0.6931471805599453094;

// Fills `table` (kTableDoubles doubles) in extended precision.
void build_tables(double* table);

// Polynomial constants that must sit in VGPRs: an FP64 FMA takes one scalar / literal
// operand at most, so "p = fma(K1, r, K0)" would otherwise rebuild K0 with two v_mov per
// evaluation.  pin() hides the value from the compiler's rematerialisation on the device.
TC_HD double pin(double value) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(value));
#endif
  return value;
}

struct Consts {
  double erf_p5_1, erf_p3_0, log_l4, exp_c3, magic, c256;
};

TC_HD Consts make_consts() {
  Consts k;
  k.erf_p5_1 = pin(-0.4);
  k.erf_p3_0 = pin(-1.0 / 3.0);
  k.log_l4 = pin(-1.4426950408889634074 / 4.0);
  k.exp_c3 = pin(kLn2 * kLn2 * kLn2 / 6.0);
  k.magic = pin(kMagic);
  k.c256 = pin(256.0);
  return k;
}

TC_HD uint64_t bits_of(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, sizeof(u));
  return u;
}
TC_HD double from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, sizeof(x));
  return x;
}
TC_HD int low_word(double x) { return (int)(uint32_t)bits_of(x); }

TC_HD double erf_fast(const double* table, const Consts& k, double x) {
  double t = fabs(x);
  t = t < 6.0 ? t : 6.0;                       // erf(6) = 1 - 2e-17
  // 1.5 * 2^45 has ulp 1 / 128: the sum IS t rounded to the nearest row centre, the row
  // number in its low dword (three additions; scaling by 128 first took four instructions)
  const double u = t + kMagicErf;
  const double c = u - kMagicErf;
  const double h = t - c;                      // exact, |h| <= 1/256
  const double* row = table + kErfOffset + 2 * low_word(u);
  const double c2 = c * c;
  double p5 = fma(c2, 4.0 / 30.0, k.erf_p5_1);
  p5 = fma(p5, c2, 0.1);
  const double p4 = fma(c2, -1.0 / 3.0, 0.5) * c;
  const double p3 = fma(c2, 2.0 / 3.0, k.erf_p3_0);
  double s = fma(p5, h, p4);
  s = fma(s, h, p3);
  s = fma(s, h, -c);
  s = fma(s, h, 1.0);
  return copysign(fma(row[1] * h, s, row[0]), x);
}

// erf(x) (x clamped to [-6, 6] as in erf_fast) and its derivative 2/sqrt(pi) exp(-x^2) (0 from
// |x| = 6 on): the row
// holds the derivative at the row centre c, and exp(-(c + h)^2) = exp(-c^2) exp(-(2 c + h) h)
// with |(2 c + h) h| <= 0.047: Taylor to degree 9 (3e-18).  For the moment expansion of a
// bin's node sum (kernels.hip.h: central_series).
TC_HD double erf_gauss_fast(const double* table, const Consts& k, double x, double* gauss,
                            double* clamped = nullptr) {
  double t = fabs(x);
  t = t < 6.0 ? t : 6.0;
  if (clamped != nullptr) *clamped = copysign(t, x);      // x within [-6, 6] (an infinite x too)
  const double u = t + kMagicErf;
  const double c = u - kMagicErf;
  const double h = t - c;
  const double* row = table + kErfOffset + 2 * low_word(u);
  const double c2 = c * c;
  double p5 = fma(c2, 4.0 / 30.0, k.erf_p5_1);
  p5 = fma(p5, c2, 0.1);
  const double p4 = fma(c2, -1.0 / 3.0, 0.5) * c;
  const double p3 = fma(c2, 2.0 / 3.0, k.erf_p3_0);
  double s = fma(p5, h, p4);
  s = fma(s, h, p3);
  s = fma(s, h, -c);
  s = fma(s, h, 1.0);
  const double g = row[1];
  const double e = -fma(2.0, c, h) * h;        // -(2 c + h) h
  double q = fma(e, 1.0 / 362880.0, 1.0 / 40320.0);
  q = fma(q, e, 1.0 / 5040.0);
  q = fma(q, e, 1.0 / 720.0);
  q = fma(q, e, 1.0 / 120.0);
  q = fma(q, e, 1.0 / 24.0);
  q = fma(q, e, 1.0 / 6.0);
  q = fma(q, e, 0.5);
  q = fma(q, e, 1.0);
  // (beyond the clamp the derivative is below 3e-16 and counts as zero: the expansion around
  // such a point is the constant +-1 its nodes evaluate to)
  *gauss = fabs(x) < 6.0 ? fma(g * e, q, g) : 0.0;      // g exp(e)
  return copysign(fma(g * h, s, row[0]), x);
}

// log2(y) - offset for a positive normal y.
TC_HD double log2_fast_offset(const double* table, const Consts& k, double y,
                              double offset) {
  const uint64_t bits = bits_of(y);
  const uint32_t hi = (uint32_t)(bits >> 32);
  // mantissa in [0.5, 1) and its exponent (one more than the IEEE one: the rows' second
  // entries are log2 c - 1); the rows hold 2 / c for c in [1, 2)
#if defined(__HIP_DEVICE_COMPILE__)
  // (one v_bfe_u32; hidden from the optimiser, which would otherwise fold the field into the
  // address arithmetic as a shift and a mask)
  int idx = (int)__builtin_amdgcn_ubfe(hi, 12u, 8u);
  asm volatile("" : "+v"(idx));
  const double m = __builtin_amdgcn_frexp_mant(y);
  const double e = (double)__builtin_amdgcn_frexp_exp(y);
#else
  const int idx = (int)((hi >> 12) & 0xffu);
  int exponent;
  const double m = frexp(y, &exponent);
  const double e = (double)exponent;
#endif
  const double* row = table + kLogOffset + 2 * idx;
  const double r = fma(m, row[0], -1.0);       // 2 m / c - 1, |r| <= 2^-9
  constexpr double kInvLn2 = 1.4426950408889634074;
  double p = fma(r, kInvLn2 / 5.0, k.log_l4);
  p = fma(p, r, kInvLn2 / 3.0);
  p = fma(p, r, -kInvLn2 / 2.0);
  p = fma(p, r, kInvLn2);
  p = p * r;                                   // log2(1 + r)
  return ((e - offset) + row[1]) + p;
}

TC_HD double log2_fast(const double* table, const Consts& k, double y) {
  return log2_fast_offset(table, k, y, 0.0);
}

// 2^z; with keep == false the result is exactly 0 (the scaling step does the masking).
TC_HD double exp2_fast(const double* table, const Consts& k, double z, bool keep = true) {
  z = z < 1000.0 ? z : 1000.0;
  z = z > -1000.0 ? z : -1000.0;
  const double u = fma(z, k.c256, k.magic);
  const double kf = u - kMagic;
  const double r = fma(kf, -0.00390625, z);    // exact, |r| <= 2^-9
  const int n = low_word(u);
  const double tj = table[kExpOffset + (n & 255)];
  constexpr double c1 = kLn2, c2 = kLn2 * kLn2 / 2.0,
                   c4 = kLn2 * kLn2 * kLn2 * kLn2 / 24.0;
  double p = fma(r, c4, k.exp_c3);
  p = fma(p, r, c2);
  p = fma(p, r, c1);
  p = p * r;                                   // 2^r - 1
  return ldexp(fma(tj, p, tj), keep ? (n >> 8) : -4000);
}

// 10^x with the product x * log2(10) carried in two parts (x up to ~16: the rounding of
// a plain product would cost 4e-15 relative).
TC_HD double exp10_fast(const double* table, const Consts& k, double x) {
  const double zh = x * kLog2Of10Hi;
  const double zl = fma(x, kLog2Of10Hi, -zh) + x * kLog2Of10Lo;
  const double v = exp2_fast(table, k, zh);
  return fma(v, zl * kLn2, v);
}

}  // namespace fm
}  // namespace tc
