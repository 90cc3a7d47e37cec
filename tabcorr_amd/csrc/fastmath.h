// Table-driven FP64 erf / log / exp for the occupation kernel.
//
// The stock device libm spends ~100 FP64 VALU instructions per erf() or pow() and
// its piecewise branches diverge across the 64 draws of a wave.  The occupation
// kernel evaluates G * n_gauss of each per draw (as many VALU instructions as the
// contraction itself), so these three functions are replaced by branch-free
// table + short-polynomial forms (13-15 FP64 instructions each plus one LDS
// gather), accurate to a few 1e-16 -- far inside the 1e-10 parity budget:
//
//   erf(x)  : |x| clamped to 6; local degree-7 Taylor polynomial around the
//             nearest multiple of 1/32 (193 rows of 8 coefficients).
//   log(y)  : y = 2^e m, m in [0.5, 1); 256 rows {1/c, log c}; log(1 + r) to
//             degree 6 with |r| <= 2^-9.
//   exp(z)  : z = (256 k + j) ln2/256 + r; 256 rows 2^(j/256); exp(r) to degree 5.
//
// The same inline functions run on the host (tc_debug_fastmath) so that their
// accuracy is tested without a GPU.
#pragma once

#include <cmath>

#if defined(__HIP__)
#define TC_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TC_HD inline
#endif

namespace tc {
namespace fm {

constexpr int kErfRows = 193;       // c_i = i / 32, i = 0 .. 192
constexpr int kErfStride = 10;      // doubles per row (8 coefficients + pad: spreads
                                    // rows over LDS banks, keeps 16-byte alignment)
constexpr int kLogRows = 256;
constexpr int kExpRows = 256;
constexpr int kErfOffset = 0;
constexpr int kLogOffset = kErfRows * kErfStride;            // 1930
constexpr int kExpOffset = kLogOffset + 2 * kLogRows;        // 2442
constexpr int kTableDoubles = kExpOffset + kExpRows;         // 2698 doubles = 21.6 KB

constexpr double kLn2Hi = 6.93147180369123816490e-01;   // ln 2, upper bits
constexpr double kLn2Lo = 1.90821492927058770002e-10;   // ln 2 - kLn2Hi
constexpr double kInvLn2x256 = 256.0 * 1.44269504088896338700e+00;

// Fills `table` (kTableDoubles doubles) in extended precision.
void build_tables(double* table);

TC_HD double erf_fast(const double* table, double x) {
  double t = fabs(x);
  t = t < 6.0 ? t : 6.0;                       // erf(6) = 1 - 2e-17
  const double ri = rint(t * 32.0);
  const double h = fma(ri, -0.03125, t);       // exact, |h| <= 1/64
  const double* a = table + kErfOffset + (int)ri * kErfStride;
  double p = a[7];
  p = fma(p, h, a[6]);
  p = fma(p, h, a[5]);
  p = fma(p, h, a[4]);
  p = fma(p, h, a[3]);
  p = fma(p, h, a[2]);
  p = fma(p, h, a[1]);
  p = fma(p, h, a[0]);
  return copysign(p, x);
}

// Natural logarithm of a positive normal number.
TC_HD double log_fast(const double* table, double y) {
  int e;
  const double m = frexp(y, &e);               // m in [0.5, 1)
  const int idx = (int)((m - 0.5) * 512.0);    // 0 .. 255
  const double* row = table + kLogOffset + 2 * idx;
  const double r = fma(m, row[0], -1.0);       // m / c - 1, |r| <= 2^-9 (1 + eps)
  double p = -1.0 / 6.0;
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = p * r;
  p = fma(p, r, r);                            // log(1 + r)
  const double ef = (double)e;
  double s = fma(ef, kLn2Lo, p);
  s += row[1];
  return fma(ef, kLn2Hi, s);
}

TC_HD double exp_fast(const double* table, double z) {
  z = z < 700.0 ? z : 700.0;
  z = z > -1000.0 ? z : -1000.0;
  const double kf = rint(z * kInvLn2x256);
  double r = fma(kf, -kLn2Hi / 256.0, z);
  r = fma(kf, -kLn2Lo / 256.0, r);
  const int k = (int)kf;
  const double tj = table[kExpOffset + (k & 255)];
  double p = 1.0 / 120.0;
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = p * r;                                   // exp(r) - 1
  return ldexp(fma(tj, p, tj), k >> 8);
}

}  // namespace fm
}  // namespace tc
