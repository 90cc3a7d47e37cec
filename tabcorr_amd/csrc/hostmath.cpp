// Host-side mathematics and work planning (see hostmath.h).
#include "hostmath.h"
#include "fastmath.h"
#include "series.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <tuple>
#include <thread>

namespace tc {

void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    // Root i (descending) of P_n by Newton iteration on the three-term
    // recurrence, in extended precision.
    long double z = cosl(pi * (i + 0.75L) / (n + 0.5L));
    long double dp = 1.0L;
    for (int iter = 0; iter < 100; ++iter) {
      long double p0 = 1.0L, p1 = z;
      for (int k = 2; k <= n; ++k) {
        long double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = p2;
      }
      if (n == 0) p1 = 1.0L;
      // p1 = P_n(z), p0 = P_{n-1}(z)
      dp = n * (z * p1 - p0) / (z * z - 1.0L);
      long double dz = p1 / dp;
      z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    {
      long double p0 = 1.0L, p1 = z;
      for (int k = 2; k <= n; ++k) {
        long double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = p2;
      }
      dp = n * (z * p1 - p0) / (z * z - 1.0L);
    }
    long double weight = 2.0L / ((1.0L - z * z) * dp * dp);
    // ascending order: root z > 0 is stored at the upper end.
    x[n - 1 - i] = (double)((z + 1.0L) / 2.0L);
    x[i] = (double)((1.0L - z) / 2.0L);
    w[n - 1 - i] = (double)weight;
    w[i] = (double)weight;
  }
  if (n % 2 == 1) x[n / 2] = 0.5;
}

bool spline_interpolation_matrix(int n_points, const double* xp,
                                 std::vector<double>& a) {
  // Same linear system as tabcorr/interpolator.py:243-264, solved by
  // Gauss-Jordan elimination with partial pivoting in extended precision.
  const int n = n_points - 1;
  const int dim = 4 * n;
  std::vector<long double> m((size_t)dim * dim, 0.0L);
  auto at = [&](int r, int c) -> long double& { return m[(size_t)r * dim + c]; };
  auto power = [](long double x, int e) {
    long double v = 1.0L;
    for (int k = 0; k < e; ++k) v *= x;
    return v;
  };
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) {
      at(i, i * 4 + k) = power(xp[i], k);
      at(i + n, i * 4 + k) = power(xp[i + 1], k);
    }
  }
  for (int i = 0; i < n - 1; ++i) {
    for (int k = 0; k < 3; ++k) {
      long double v = (k + 1) * power(xp[i + 1], k);
      at(i + 2 * n, i * 4 + 1 + k) = v;
      at(i + 2 * n, (i + 1) * 4 + 1 + k) = -v;
    }
    const long double c2[2] = {2.0L, 6.0L};
    for (int k = 0; k < 2; ++k) {
      long double v = c2[k] * power(xp[i + 1], k);
      at(i + 3 * n - 1, i * 4 + 2 + k) = v;
      at(i + 3 * n - 1, (i + 1) * 4 + 2 + k) = -v;
    }
  }
  at(dim - 1, 3) = 6.0L * xp[1];
  at(dim - 1, 7) = -6.0L * xp[1];
  at(dim - 2, dim - 5) = 6.0L * xp[n_points - 2];
  at(dim - 2, dim - 1) = -6.0L * xp[n_points - 2];

  std::vector<long double> inv((size_t)dim * dim, 0.0L);
  for (int i = 0; i < dim; ++i) inv[(size_t)i * dim + i] = 1.0L;
  for (int col = 0; col < dim; ++col) {
    int pivot = col;
    long double best = fabsl(at(col, col));
    for (int r = col + 1; r < dim; ++r) {
      if (fabsl(at(r, col)) > best) {
        best = fabsl(at(r, col));
        pivot = r;
      }
    }
    if (best == 0.0L || !std::isfinite((double)best)) return false;
    if (pivot != col) {
      for (int c = 0; c < dim; ++c) {
        std::swap(at(pivot, c), at(col, c));
        std::swap(inv[(size_t)pivot * dim + c], inv[(size_t)col * dim + c]);
      }
    }
    long double scale = 1.0L / at(col, col);
    for (int c = 0; c < dim; ++c) {
      at(col, c) *= scale;
      inv[(size_t)col * dim + c] *= scale;
    }
    for (int r = 0; r < dim; ++r) {
      if (r == col) continue;
      long double f = at(r, col);
      if (f == 0.0L) continue;
      for (int c = 0; c < dim; ++c) {
        at(r, c) -= f * at(col, c);
        inv[(size_t)r * dim + c] -= f * inv[(size_t)col * dim + c];
      }
    }
  }
  // a[:, :-1] = inv[:, :n]; a[:, 1:] += inv[:, n:2n]  (interpolator.py:268-270)
  a.assign((size_t)dim * n_points, 0.0);
  for (int r = 0; r < dim; ++r) {
    for (int j = 0; j < n_points; ++j) {
      long double v = 0.0L;
      if (j < n) v += inv[(size_t)r * dim + j];
      if (j >= 1) v += inv[(size_t)r * dim + n + j - 1];
      a[(size_t)r * n_points + j] = (double)v;
    }
  }
  return true;
}

namespace fm {

void build_tables(double* table) {
  const long double two_over_sqrt_pi = 1.128379167095512573896158903121545172L;
  for (int i = 0; i < kErfRows; ++i) {
    const long double c = i / 128.0L;
    table[kErfOffset + 2 * i] = (double)erfl(c);
    table[kErfOffset + 2 * i + 1] = (double)(two_over_sqrt_pi * expl(-c * c));
  }
  for (int i = 0; i < kLogRows; ++i) {
    const long double c = 1.0L + (i + 0.5L) / 256.0L;
    // 2 / c: the mantissa comes as m in [0.5, 1)
    const double inv = (double)(2.0L / c);
    table[kLogOffset + 2 * i] = inv;
    // log2 of the reciprocal actually stored, so that r = m * inv - 1 is consistent
    // (minus one: the exponent of a mantissa in [0.5, 1) is one more than the IEEE exponent)
    table[kLogOffset + 2 * i + 1] = (double)(-log2l((long double)inv));
  }
  for (int j = 0; j < kExpRows; ++j)
    table[kExpOffset + j] = (double)exp2l(j / 256.0L);
}

}  // namespace fm

namespace {

void add_triangle(std::vector<Segment>& out, int component, int lo, int hi, int budget) {
  // rows [lo, hi), columns [lo, i]: diagonal triangles of `side` rows plus the
  // rectangles left of them
  const int n = hi - lo;
  if (n <= 0) return;
  const int pieces = (n + budget - 1) / budget;           // whole triangle if it fits
  const int side = pieces == 1 ? n : std::max(1, std::min(n, budget / 2));
  for (int a = lo; a < hi; a += side) {
    const int b = std::min(hi, a + side);
    for (int c = lo; c < a; c += side)
      out.push_back({component, 1, a, b, c, std::min(a, c + side), 0, 0});
    out.push_back({component, 0, a, b, a, b, 0, 0});
  }
}

void add_rectangle(std::vector<Segment>& out, int component, int i_lo, int i_hi,
                   int j_lo, int j_hi, int budget) {
  const int n_i = i_hi - i_lo, n_j = j_hi - j_lo;
  if (n_i <= 0 || n_j <= 0) return;
  // equal column blocks of at most budget / 2 bins (the other half of the budget is
  // left for the rows a workgroup covers); rows are cut by the chunking
  const int width_max = std::max(1, budget / 2);
  const int n_blocks = (n_j + width_max - 1) / width_max;
  const int height_max = std::max(1, budget - (n_j + n_blocks - 1) / n_blocks);
  for (int jb = 0; jb < n_blocks; ++jb) {
    const int c0 = j_lo + (int)((int64_t)n_j * jb / n_blocks);
    const int c1 = j_lo + (int)((int64_t)n_j * (jb + 1) / n_blocks);
    // row blocks only when even a single row range would not fit (very many rows
    // are handled by the chunking; a chunk never spans more than its rows)
    (void)height_max;
    out.push_back({component, 1, i_lo, i_hi, c0, c1, 0, 0});
  }
}

}  // namespace

namespace series {

double h_max(int n_terms) {
  // tail of the expansion beyond n_terms: sum_n 1.23 2^((n-1)/2) h^n / sqrt(n! n) (series.h)
  auto tail = [n_terms](double h) {
    long double sum = 0.0L, factorial = 1.0L;
    for (int n = 1; n <= n_terms + 60; ++n) {
      factorial *= n;
      if (n > n_terms)
        sum += 1.23L * powl(2.0L, 0.5L * (n - 1)) * powl((long double)h, n) /
               sqrtl(factorial * n);
    }
    return (double)sum;
  };
  double lo = 0.0, hi = 4.0;
  for (int i = 0; i < 80; ++i) {
    const double mid = 0.5 * (lo + hi);
    (tail(mid) < kTolerance ? lo : hi) = mid;
  }
  return lo;
}

void bin_consts(int n_gauss, const double* log_m, const double* weight, double log_min,
                double log_max, double* consts, int32_t* thresholds) {
  for (int i = 0; i < kStride; ++i) consts[i] = 0.0;
  for (int i = 0; i < kThresholds; ++i) thresholds[i] = 0;
  const double centre = 0.5 * (log_min + log_max), half = 0.5 * std::fabs(log_max - log_min);
  if (!std::isfinite(centre) || !std::isfinite(half)) return;
  consts[0] = centre;
  long double factorial = 1.0L;
  for (int n = 1; n <= kMaxTerms; ++n) {
    factorial *= n;
    long double moment = 0.0L;
    for (int k = 0; k < n_gauss; ++k)
      moment += (long double)weight[k] * powl((long double)log_m[k] - (long double)centre, n);
    consts[kFirst + n - 1] = (double)((n % 2 ? 1.0L : -1.0L) * moment / factorial);
  }
  // (the nodes may lie a rounding error outside the edges: the bound uses the farthest one)
  double reach = half;
  for (int k = 0; k < n_gauss; ++k) reach = std::max(reach, std::fabs(log_m[k] - centre));
  static const double h_of_step[kSteps] = {h_max(8), h_max(12), h_max(16), h_max(20),
                                           h_max(24)};
  for (int s = 0; s < kSteps; ++s) {
    // |1 / sigma| < h / reach; compared through the high dwords (strictly below suffices)
    const double limit = reach > 0.0 ? h_of_step[s] / reach : 1e300;
    uint64_t bits;
    std::memcpy(&bits, &limit, sizeof(bits));
    thresholds[s] = std::isfinite(limit) ? (int32_t)(bits >> 32) : 0x7fe00000;
  }
}

namespace sat {

double r_max(int n_terms) {
  // tail beyond n_terms of sum_n 16 r^n: 16 r^(n_terms + 1) / (1 - r) (series.h)
  double lo = 0.0, hi = 0.9;
  for (int i = 0; i < 80; ++i) {
    const double mid = 0.5 * (lo + hi);
    (16.0 * std::pow(mid, n_terms + 1) / (1.0 - mid) < kTolerance ? lo : hi) = mid;
  }
  return lo;
}

void bin_consts(int n_gauss, const double* mass, const double* weight, double log_min,
                double log_max, double* consts, int32_t* thresholds) {
  for (int i = 0; i < kStride; ++i) consts[i] = 0.0;
  for (int i = 0; i < kThresholds; ++i) thresholds[i] = 0;
  const double centre = std::pow(10.0, 0.5 * (log_min + log_max));
  if (!std::isfinite(centre) || !(centre > 0.0)) return;
  consts[0] = centre;
  double y_max = 0.0;
  for (int k = 0; k < n_gauss; ++k) y_max = std::max(y_max, std::fabs(mass[k] / centre - 1.0));
  consts[1] = y_max;
  long double factorial = 1.0L;
  for (int n = 0; n <= kMaxTerms; ++n) {
    if (n > 0) factorial *= n;
    long double moment = 0.0L;
    for (int k = 0; k < n_gauss; ++k)
      moment += (long double)weight[k] *
                powl((long double)mass[k] / (long double)centre - 1.0L, n);
    consts[n == 0 ? 3 : kFirst + n - 1] = (double)(moment / factorial);
  }
  if (!std::isfinite(y_max)) return;
  for (int s = 0; s < kSteps; ++s) {
    // eps y_max <= r  <=>  M0 <= Mc (1 - y_max / r); through the high dwords, strictly below
    const double r = r_max(12 + 4 * s);
    const double limit = centre * (1.0 - y_max / r);
    if (!(limit > 0.0) || !std::isfinite(limit)) continue;
    uint64_t bits;
    std::memcpy(&bits, &limit, sizeof(bits));
    thresholds[s] = (int32_t)(bits >> 32);
  }
  // (the limit of the SHORTEST expansion that serves the bin at all: kShortest)
  for (int s = kSteps - 1; s >= 0; --s)
    if (thresholds[s] != 0) thresholds[kShortest] = thresholds[s];
}

}  // namespace sat

namespace record {

void group_record(bool central, const double* consts_i, const double* consts_j,
                  const int32_t* thresholds, double sum_i, double sum_j, const double* log_m,
                  const double* mass, int n_gauss, int first_member, bool two, double* out) {
  for (int i = 0; i < kStride; ++i) out[i] = 0.0;
  int32_t head[6] = {0, 0, 0, 0, 0, 0};
  const int n_steps = central ? series::kSteps : sat::kSteps;
  for (int s = 0; s < n_steps; ++s) head[s] = thresholds[s];
  const int32_t members[2] = {2 * first_member + (two ? 1 : 0), 0};
  if (central) head[5] = members[0];
  else std::memcpy(out + kHigh, members, sizeof(members));
  std::memcpy(out, head, sizeof(head));
  out[kCentre] = consts_i[0];
  if (central) {
    out[kLow] = log_m[0];
    out[kHigh] = log_m[n_gauss - 1];
    out[kFirstSum] = sum_i;
    out[kFirstSum + 1] = sum_j;
  } else {
    out[kLow] = mass[0] > mass[n_gauss - 1] ? mass[0] : mass[n_gauss - 1];
    out[kFirstSum] = consts_i[3];
    out[kFirstSum + 1] = consts_j[3];
  }
  const int first = central ? series::kFirst : sat::kFirst;
  const int n_terms = central ? series::kMaxTerms : sat::kMaxTerms;
  for (int n = 0; n < n_terms; ++n) {
    double* block = out + kHead + kBlock * (n / 4);
    block[n % 4] = consts_i[first + n];
    block[4 + n % 4] = consts_j[first + n];
  }
}

}  // namespace record

namespace cen_record {

void bin_record(const double* consts, const int32_t* thresholds, double weight_sum,
                const double* log_m, int n_gauss, double* out) {
  for (int i = 0; i < kStride; ++i) out[i] = 0.0;
  int32_t head[6] = {0, 0, 0, 0, 0, 0};
  for (int s = 0; s < series::kSteps; ++s) head[s] = thresholds[s];
  std::memcpy(out, head, sizeof(head));
  out[kLow] = log_m[0];
  out[kHigh] = log_m[n_gauss - 1];
  out[kCentre] = consts[0];
  out[kSum] = weight_sum;
  for (int n = 0; n < series::kMaxTerms; ++n) out[kHead + n] = consts[series::kFirst + n];
}

}  // namespace cen_record

namespace sat_record {

void bin_record(const double* consts, const int32_t* thresholds, const double* mass, int n_gauss,
                double* out) {
  for (int i = 0; i < kStride; ++i) out[i] = 0.0;
  int32_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int shortest = 0;
  for (int s = sat::kSteps - 1; s >= 0; --s) {
    head[s] = thresholds[s];
    if (thresholds[s] != 0) shortest = s;
  }
  head[6] = thresholds[sat::kShortest];
  head[7] = 3 + shortest;
  std::memcpy(out, head, sizeof(head));
  out[kCentre] = consts[0];
  out[kSum] = consts[3];
  out[kLargest] = mass[0] > mass[n_gauss - 1] ? mass[0] : mass[n_gauss - 1];
  for (int n = 0; n < sat::kMaxTerms; ++n) out[kHead + n] = consts[sat::kFirst + n];
}

}  // namespace sat_record

}  // namespace series

void find_node_groups(int n_bins, int n_central, const double* log_min, const double* log_max,
                      NodeGroups& out) {
  out = NodeGroups();
  // (exact comparison of the bin edges: members of a group must have bit-identical nodes)
  std::map<std::tuple<int, uint64_t, uint64_t>, int> seen;
  std::vector<std::vector<int32_t>> groups;
  auto bits = [](double x) {
    uint64_t u;
    std::memcpy(&u, &x, sizeof(u));
    return u;
  };
  for (int g = 0; g < n_bins; ++g) {
    if (g == n_central) out.n_central_groups = (int)groups.size();
    const auto key = std::make_tuple(g < n_central ? 0 : 1, bits(log_min[g]), bits(log_max[g]));
    auto it = seen.find(key);
    // (a NaN edge never equals anything: such a bin stays alone)
    if (it == seen.end() || log_min[g] != log_min[g] || log_max[g] != log_max[g]) {
      seen[key] = (int)groups.size();
      groups.emplace_back(1, g);
    } else {
      groups[it->second].push_back(g);
    }
  }
  if (n_central >= n_bins) out.n_central_groups = (int)groups.size();
  out.n_groups = (int)groups.size();
  out.begin.push_back(0);
  for (const std::vector<int32_t>& members : groups) {
    out.member.insert(out.member.end(), members.begin(), members.end());
    out.begin.push_back((int32_t)out.member.size());
    out.largest = std::max(out.largest, (int)members.size());
  }
}

void build_plan(int mode, int n_bins, const uint8_t* is_central, int block,
                int row_budget, Plan& plan) {
  plan.mode = mode;
  plan.n_bins = n_bins;
  plan.block = block;
  plan.perm.clear();
  for (int g = 0; g < n_bins; ++g)
    if (is_central[g]) plan.perm.push_back(g);
  plan.n_central = (int)plan.perm.size();
  for (int g = 0; g < n_bins; ++g)
    if (!is_central[g]) plan.perm.push_back(g);
  const int gc = plan.n_central;
  plan.n_components = mode == 0 ? 3 : 2;
  row_budget = std::max(row_budget, 4);

  plan.segments.clear();
  if (mode == 0) {
    add_triangle(plan.segments, 0, 0, gc, row_budget);
    add_rectangle(plan.segments, 1, gc, n_bins, 0, gc, row_budget);
    add_triangle(plan.segments, 2, gc, n_bins, row_budget);
  } else {
    // one-row rectangles without a row bin; the chunking cuts them by columns
    if (gc > 0) plan.segments.push_back({0, 1, -1, 0, 0, gc, 0, 0});
    if (n_bins > gc) plan.segments.push_back({1, 1, -1, 0, gc, n_bins, 0, 0});
  }

  plan.column.clear();
  plan.prefactor.clear();
  plan.pos_i.clear();
  plan.pos_j.clear();
  plan.n_entries = 0;
  for (Segment& seg : plan.segments) {
    seg.q_begin = (int64_t)plan.column.size();
    const int j_last = seg.rectangular ? seg.j_hi - 1 : -1;
    int64_t n_real = 0;
    if (mode != 0) {
      n_real = seg.j_hi - seg.j_lo;
    } else if (seg.rectangular) {
      n_real = (int64_t)(seg.i_hi - seg.i_lo) * (seg.j_hi - seg.j_lo);
    } else {
      for (int i = seg.i_lo; i < seg.i_hi; ++i) n_real += i - seg.j_lo + 1;
    }
    seg.n_real = n_real;
    int i = seg.i_lo, j = seg.j_lo;
    for (int64_t e = 0; e < n_real; ++e) {
      if (mode == 0) {
        const int a = plan.perm[i], b = plan.perm[j];
        plan.column.push_back(packed_index(a, b));
        plan.prefactor.push_back(a == b ? 1 : 2);
      } else {
        plan.column.push_back(plan.perm[j]);
        plan.prefactor.push_back(1);
      }
      plan.pos_i.push_back(i);
      plan.pos_j.push_back(j);
      if (e + 1 < n_real) {
        if (mode != 0) ++j; else advance_pair(seg.j_lo, j_last, i, j);
      }
    }
    // zero padding: frozen on the last real pair (any valid pair would do)
    const int64_t padded = (n_real + block - 1) / block * block;
    for (int64_t e = n_real; e < padded; ++e) {
      plan.column.push_back(-1);
      plan.prefactor.push_back(0);
      plan.pos_i.push_back(i);
      plan.pos_j.push_back(j);
    }
    plan.n_entries += n_real;
  }
  plan.n_positions = (int64_t)plan.column.size();
}

void build_quad_layout(int n_bins, int n_central, bool by_type, QuadLayout& out) {
  out.n_bins = n_bins;
  out.n_central = n_central;
  out.by_type = by_type;
  out.comps.clear();
  out.n_units = 0;
  // (an empty component -- a table without centrals or without satellites -- keeps its
  // slot: the component index is the output component)
  auto add = [&](int component, bool triangular, int i0, int ni, int j0, int nj) {
    if (ni <= 0 || nj <= 0) ni = nj = 0;
    QuadComp comp;
    comp.component = component;
    comp.triangular = triangular ? 1 : 0;
    comp.i_bin0 = i0;
    comp.i_count = ni;
    comp.j_bin0 = j0;
    comp.j_count = nj;
    comp.n_rb = (ni + 3) / 4;
    comp.n_cb = triangular ? comp.n_rb : (nj + 3) / 4;
    comp.unit_base = out.n_units;
    comp.n_units = triangular ? (int64_t)comp.n_rb * (comp.n_rb + 1) / 2
                              : (int64_t)comp.n_rb * comp.n_cb;
    out.n_units += comp.n_units;
    out.comps.push_back(comp);
  };
  if (!by_type) {
    add(0, true, 0, n_bins, 0, n_bins);
  } else {
    const int n_sat = n_bins - n_central;
    add(0, true, 0, n_central, 0, n_central);
    add(1, false, n_central, n_sat, 0, n_central);
    add(2, true, n_central, n_sat, n_central, n_sat);
  }
}

namespace {
// Number the slabs so that those of one output group are consecutive: `slab_group[s]` is the
// group of provisional slab s; returns the new id of every provisional slab (stable: inside a
// group the provisional order is kept) and fills group_begin.
std::vector<int32_t> number_slabs_by_group(const std::vector<int32_t>& slab_group, int n_groups,
                                           std::vector<int32_t>& group_begin) {
  group_begin.assign((size_t)n_groups + 1, 0);
  for (int32_t g : slab_group) ++group_begin[(size_t)g + 1];
  for (int g = 0; g < n_groups; ++g) group_begin[g + 1] += group_begin[g];
  std::vector<int32_t> cursor(group_begin.begin(), group_begin.end() - 1);
  std::vector<int32_t> renumbered(slab_group.size());
  for (size_t s = 0; s < slab_group.size(); ++s) renumbered[s] = cursor[slab_group[s]]++;
  return renumbered;
}
}  // namespace

void build_quad_schedule(const QuadLayout& layout, int n_tiles, int n_rtiles, int n_tables,
                         bool separate, int max_waves, int min_units_per_wave,
                         QuadSchedule& out, int order) {
  bool table_sync = order == kQuadTableSync && std::max(1, n_tables) > 1;
  const bool table_major = order == kQuadTableMajor || order == kQuadTableSync;
  const bool rtile_major = order == kQuadRtileMajor && n_rtiles > 1;
  const bool unit_major = order == kQuadUnitMajor && std::max(1, n_tables) == 1;
  out.runs.clear();
  out.wave_runs.clear();
  out.group_begin.clear();
  n_tables = std::max(1, n_tables);
  const int n_comps = (int)layout.comps.size();
  const int groups_per_rtile = separate ? n_comps : 1;
  out.n_groups = n_tiles * n_rtiles * groups_per_rtile;
  const int64_t per_rtile = layout.n_units * n_tables;          // units of one (tile, r tile)
  const int64_t total = per_rtile * n_rtiles * n_tiles;
  // (r-tile-major: the shares are cut per r tile)
  const int64_t span = rtile_major ? per_rtile * n_tiles : total;
  const int n_passes = rtile_major ? n_rtiles : 1;
  int64_t n_waves = std::min<int64_t>(max_waves, span / std::max(1, min_units_per_wave));
  n_waves = std::max<int64_t>(1, std::min<int64_t>(n_waves, span));
  if (total == 0) n_waves = 0;
  out.n_waves = (int)n_waves;
  out.group_begin.assign((size_t)out.n_groups + 1, 0);
  // table-synchronous: needs whole eighths of the waves (else plain table-major)
  table_sync = table_sync && n_waves >= 8 && n_waves % 8 == 0;
  bool unit_sync = order == kQuadUnitSync && n_tables == 1 && n_waves >= 8 &&
                   n_waves % 8 == 0 && layout.n_units >= 64;
  // kQuadUnitSync: S sub-ranges per (draw tile, r tile) such that the rounds of an XCD fill
  int unit_parts = 1;
  int64_t n_items = 0;
  if (unit_sync) {
    double best = -1.0;
    for (int parts = 1; parts <= 8 && layout.n_units / parts >= 64; ++parts) {
      const double per = (double)n_tiles * n_rtiles * parts / (double)n_waves;
      const double fill = per / std::ceil(per);
      if (fill > best + 1e-9) {
        best = fill;
        unit_parts = parts;
      }
    }
    n_items = (int64_t)n_tiles * n_rtiles * unit_parts;
    // (fewer items than waves: as many waves as whole eighths of the items)
    if (n_items < n_waves) n_waves = n_items / 8 * 8;
    unit_sync = n_waves >= 8;
    if (!unit_sync) n_waves = std::max<int64_t>(1, std::min<int64_t>(max_waves, n_items));
    out.n_waves = (int)n_waves;
  }
  const int64_t per_xcd = table_sync || unit_sync ? n_waves / 8 : 1;
  const int64_t per_table_span = layout.n_units * n_rtiles * (int64_t)n_tiles;

  // position -> (tile, rtile, comp, table, unit in component)
  int slab = 0;
  std::vector<int32_t> slab_group;
  for (int64_t w = 0; w < n_waves; ++w) {
    out.wave_runs.push_back((int32_t)out.runs.size());
    const size_t first_run = out.runs.size();
    // the wave's intervals of the linearised space: one (per r tile in r-tile-major order), or
    // -- table-synchronous -- its slice of every table piece of its XCD's range
    std::vector<std::pair<int64_t, int64_t>> intervals;
    if (table_sync) {
      const int64_t xcd = w / per_xcd, local = w % per_xcd;
      const int64_t lo = (int64_t)((__int128)total * xcd / 8);
      const int64_t hi = (int64_t)((__int128)total * (xcd + 1) / 8);
      for (int64_t a = lo; a < hi;) {
        const int64_t b = std::min(hi, (a / per_table_span + 1) * per_table_span);
        intervals.emplace_back(a + (int64_t)((__int128)(b - a) * local / per_xcd),
                               a + (int64_t)((__int128)(b - a) * (local + 1) / per_xcd));
        a = b;
      }
    } else if (unit_sync) {
      // items in (r tile, sub-range, draw tile) order; XCD x owns items [n x / 8, n (x + 1) / 8),
      // its wave `local` takes every per_xcd-th of them
      const int64_t xcd = w / per_xcd, local = w % per_xcd;
      const int64_t lo = n_items * xcd / 8, hi = n_items * (xcd + 1) / 8;
      for (int64_t item = lo + local; item < hi; item += per_xcd) {
        const int64_t tile = item % n_tiles, rest = item / n_tiles;
        const int64_t part = rest % unit_parts, rtile = rest / unit_parts;
        const int64_t base = (tile * n_rtiles + rtile) * per_rtile;
        intervals.emplace_back(base + layout.n_units * part / unit_parts,
                               base + layout.n_units * (part + 1) / unit_parts);
      }
    } else {
      // (128-bit product: span * w can exceed 63 bits for huge batches of huge tables)
      for (int pass = 0; pass < n_passes; ++pass)
        intervals.emplace_back((int64_t)((__int128)span * w / n_waves),
                               (int64_t)((__int128)span * (w + 1) / n_waves));
    }
    for (size_t pass_index = 0; pass_index < intervals.size(); ++pass_index) {
    const int pass = (int)pass_index;
    int64_t begin = intervals[pass_index].first;
    const int64_t end = intervals[pass_index].second;
    while (begin < end) {
      int64_t tile_rtile, unit;
      int comp = 0, table;
      int64_t part_stop = end;        // (unit-major: a run stays inside its part)
      if (unit_major) {
        // parts of the unit range, then (tile, rtile), then the part's units (across the
        // components in layout order)
        const int64_t n_tr = (int64_t)n_tiles * n_rtiles;
        int part = 0;
        int64_t offset = 0, lo = 0, hi = per_rtile / kQuadUnitParts;
        while (part + 1 < kQuadUnitParts && begin >= offset + (hi - lo) * n_tr) {
          offset += (hi - lo) * n_tr;
          ++part;
          lo = hi;
          hi = per_rtile * (part + 1) / kQuadUnitParts;
        }
        const int64_t size = hi - lo, inside = (begin - offset) % size;
        tile_rtile = (begin - offset) / size;
        part_stop = begin + (size - inside);
        int64_t rest = lo + inside;
        while (rest >= layout.comps[comp].n_units) {
          rest -= layout.comps[comp].n_units;
          ++comp;
        }
        table = 0;
        unit = rest;
      } else if (rtile_major) {
        // inside r tile `pass`: draw tiles, then components, tables, units
        const int64_t tile = begin / per_rtile;
        int64_t rest = begin % per_rtile;
        tile_rtile = tile * n_rtiles + pass;
        while (rest >= layout.comps[comp].n_units * n_tables) {
          rest -= layout.comps[comp].n_units * n_tables;
          ++comp;
        }
        table = (int)(rest / layout.comps[comp].n_units);
        unit = rest % layout.comps[comp].n_units;
      } else if (table_major) {
        // tables, then (tile, rtile), then the components' units
        const int64_t per_table = layout.n_units * n_rtiles * n_tiles;
        table = (int)(begin / per_table);
        int64_t rest = begin % per_table;
        tile_rtile = rest / layout.n_units;
        rest %= layout.n_units;
        while (rest >= layout.comps[comp].n_units) {
          rest -= layout.comps[comp].n_units;
          ++comp;
        }
        unit = rest;
      } else {
        tile_rtile = begin / per_rtile;
        int64_t rest = begin % per_rtile;
        // inside a (tile, rtile): components, then tables, then the component's units
        while (rest >= layout.comps[comp].n_units * n_tables) {
          rest -= layout.comps[comp].n_units * n_tables;
          ++comp;
        }
        table = (int)(rest / layout.comps[comp].n_units);
        unit = rest % layout.comps[comp].n_units;
      }
      const QuadComp& qc = layout.comps[comp];
      const int64_t stop =
          std::min<int64_t>(std::min<int64_t>(end, part_stop), begin + (qc.n_units - unit));
      QuadRun run;
      run.tile = (int32_t)(tile_rtile / n_rtiles);
      run.rtile = (int32_t)(tile_rtile % n_rtiles);
      run.comp = comp;
      run.table = table;
      if (qc.triangular) {
        int64_t rb = (int64_t)((std::sqrt(8.0 * (double)unit + 1.0) - 1.0) / 2.0);
        while ((rb + 1) * (rb + 2) / 2 <= unit) ++rb;
        while (rb * (rb + 1) / 2 > unit) --rb;
        run.rb0 = (int32_t)rb;
        run.cb0 = (int32_t)(unit - rb * (rb + 1) / 2);
      } else {
        run.rb0 = (int32_t)(unit / qc.n_cb);
        run.cb0 = (int32_t)(unit % qc.n_cb);
      }
      run.count = (int32_t)(stop - begin);
      run.slab = -1;
      out.runs.push_back(run);
      begin = stop;
    }
    }
    // flush after the last run of every output group this wave touches
    auto group_of = [&](const QuadRun& run) {
      return ((int64_t)run.tile * n_rtiles + run.rtile) * groups_per_rtile +
             (separate ? run.comp : 0);
    };
    for (size_t k = first_run; k < out.runs.size(); ++k) {
      const bool last = k + 1 == out.runs.size();
      if (last || group_of(out.runs[k + 1]) != group_of(out.runs[k])) {
        out.runs[k].slab = slab++;
        slab_group.push_back((int32_t)group_of(out.runs[k]));
      }
    }
  }
  out.wave_runs.push_back((int32_t)out.runs.size());
  out.n_slabs = slab;
  // slabs of a group consecutive (they already are unless the order is table-major)
  const std::vector<int32_t> renumbered =
      number_slabs_by_group(slab_group, out.n_groups, out.group_begin);
  for (QuadRun& run : out.runs)
    if (run.slab >= 0) run.slab = renumbered[run.slab];
}

void merge_quad_schedule(const QuadLayout& layout, int n_rtiles, bool separate,
                         int waves_per_block, int max_slots, QuadSchedule& schedule,
                         QuadMergePlan& plan) {
  const int groups_per_rtile = separate ? (int)layout.comps.size() : 1;
  auto group_of = [&](const QuadRun& run) {
    return ((int64_t)run.tile * n_rtiles + run.rtile) * groups_per_rtile +
           (separate ? run.comp : 0);
  };
  plan.waves_per_block = waves_per_block;
  plan.n_blocks = (schedule.n_waves + waves_per_block - 1) / waves_per_block;
  plan.lds_slots = 0;
  plan.block_begin.assign((size_t)plan.n_blocks + 1, 0);
  plan.merges.clear();
  std::vector<int32_t> slab_group;
  int next_slab = 0;
  for (int block = 0; block < plan.n_blocks; ++block) {
    const int w_begin = block * waves_per_block;
    const int w_end = std::min(schedule.n_waves, w_begin + waves_per_block);
    std::vector<int> flushes;      // runs of this workgroup that flush, in order
    for (int ri = schedule.wave_runs[w_begin]; ri < schedule.wave_runs[w_end]; ++ri)
      if (schedule.runs[ri].slab >= 0) flushes.push_back(ri);
    if ((int)flushes.size() > max_slots || flushes.size() < 2) {
      for (int ri : flushes) {      // direct form, renumbered
        schedule.runs[ri].slab = next_slab++;
        slab_group.push_back((int32_t)group_of(schedule.runs[ri]));
      }
    } else {
      plan.lds_slots = std::max(plan.lds_slots, (int)flushes.size());
      for (size_t k = 0; k < flushes.size(); ++k) {
        const int64_t group = group_of(schedule.runs[flushes[k]]);
        if (k == 0 || group != group_of(schedule.runs[flushes[k - 1]])) {
          plan.merges.push_back({next_slab++, (int32_t)k, 0, 0});
          slab_group.push_back((int32_t)group);
        }
        ++plan.merges.back().count;
        schedule.runs[flushes[k]].slab = -2 - (int32_t)k;
      }
    }
    plan.block_begin[(size_t)block + 1] = (int32_t)plan.merges.size();
  }
  schedule.n_slabs = next_slab;
  const std::vector<int32_t> renumbered =
      number_slabs_by_group(slab_group, schedule.n_groups, schedule.group_begin);
  for (QuadRun& run : schedule.runs)
    if (run.slab >= 0) run.slab = renumbered[run.slab];
  for (QuadMerge& m : plan.merges) m.slab = renumbered[m.slab];
}

void triangle_parts(int n_rb, int n_parts, int32_t* rb0, int32_t* cb0, int32_t* count) {
  const int64_t n_units = (int64_t)n_rb * (n_rb + 1) / 2;
  for (int part = 0; part < n_parts; ++part) {
    const int64_t begin = n_units * part / n_parts, end = n_units * (part + 1) / n_parts;
    int64_t rb = (int64_t)((std::sqrt(8.0 * (double)begin + 1.0) - 1.0) / 2.0);
    while ((rb + 1) * (rb + 2) / 2 <= begin) ++rb;
    while (rb * (rb + 1) / 2 > begin) --rb;
    rb0[part] = (int32_t)rb;
    cb0[part] = (int32_t)(begin - rb * (rb + 1) / 2);
    count[part] = (int32_t)(end - begin);
  }
}

QuadTiling quad_tiling(int n_r) {
  QuadTiling tiling;
  tiling.n_rtiles = (n_r + 19) / 20;
  tiling.r_per_tile = (n_r + tiling.n_rtiles - 1) / tiling.n_rtiles;
  tiling.n_u = (tiling.r_per_tile + 3) / 4;
  return tiling;
}

void fill_quad_table(const QuadLayout& layout, const std::vector<int32_t>& perm, int n_r,
                     int64_t n_pairs, const void* matrix, bool matrix_is_f32,
                     const QuadTiling& tiling, std::vector<double>& out) {
  const int up = (tiling.n_u + 1) / 2;
  const size_t per_unit = (size_t)up * 128;
  out.assign((size_t)tiling.n_rtiles * layout.n_units * per_unit, 0.0);
  auto source = [&](int r, int64_t column) {
    return matrix_is_f32 ? (double)((const float*)matrix)[(size_t)r * n_pairs + column]
                         : ((const double*)matrix)[(size_t)r * n_pairs + column];
  };
  for (const QuadComp& comp : layout.comps) {
    for (int rb = 0; rb < comp.n_rb; ++rb) {
      for (int cb = 0; cb < quad_row_length(comp, rb); ++cb) {
        const int64_t unit =
            comp.unit_base + (comp.triangular ? (int64_t)rb * (rb + 1) / 2 + cb
                                              : (int64_t)rb * comp.n_cb + cb);
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, k = lane >> 4;
          const int i_local = 4 * rb + (m >> 2), j_local = 4 * cb + k;
          if (i_local >= comp.i_count || j_local >= comp.j_count) continue;
          const int i = comp.i_bin0 + i_local, j = comp.j_bin0 + j_local;
          if (comp.triangular && j > i) continue;
          const int64_t column = packed_index(perm[i], perm[j]);
          const double prefactor = i == j ? 1.0 : 2.0;
          for (int z = 0; z < tiling.n_rtiles; ++z) {
            for (int u = 0; u < tiling.n_u; ++u) {
              const int r_local = 4 * u + (m & 3);
              const int r = z * tiling.r_per_tile + r_local;
              if (r_local >= tiling.r_per_tile || r >= n_r) continue;
              out[((size_t)z * layout.n_units + unit) * per_unit +
                  ((size_t)(u / 2) * 64 + lane) * 2 + (u & 1)] = source(r, column) * prefactor;
            }
          }
        }
      }
    }
  }
}

QuadTiling quad_tiling_f32(int n_r) {
  QuadTiling tiling;
  tiling.n_rtiles = (n_r + 15) / 16;
  tiling.r_per_tile = (n_r + tiling.n_rtiles - 1) / tiling.n_rtiles;
  tiling.n_u = (tiling.r_per_tile + 3) / 4;
  return tiling;
}

void fill_quad_table_f32(const QuadLayout& layout, const std::vector<int32_t>& perm, int n_r,
                         int64_t n_pairs, const void* matrix, bool matrix_is_f32,
                         const QuadTiling& tiling, std::vector<float>& out) {
  const size_t per_unit = 256;       // 64 lanes x 4 sub-tiles
  out.assign((size_t)tiling.n_rtiles * layout.n_units * per_unit, 0.0f);
  auto source = [&](int r, int64_t column) {
    return matrix_is_f32 ? (double)((const float*)matrix)[(size_t)r * n_pairs + column]
                         : ((const double*)matrix)[(size_t)r * n_pairs + column];
  };
  for (const QuadComp& comp : layout.comps) {
    for (int rb = 0; rb < comp.n_rb; ++rb) {
      for (int cb = 0; cb < quad_row_length(comp, rb); ++cb) {
        const int64_t unit =
            comp.unit_base + (comp.triangular ? (int64_t)rb * (rb + 1) / 2 + cb
                                              : (int64_t)rb * comp.n_cb + cb);
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, k = lane >> 4;
          const int i_local = 4 * rb + (m & 3), j_local = 4 * cb + k;
          if (i_local >= comp.i_count || j_local >= comp.j_count) continue;
          const int i = comp.i_bin0 + i_local, j = comp.j_bin0 + j_local;
          if (comp.triangular && j > i) continue;
          const int64_t column = packed_index(perm[i], perm[j]);
          const double prefactor = i == j ? 1.0 : 2.0;
          for (int z = 0; z < tiling.n_rtiles; ++z) {
            for (int u = 0; u < tiling.n_u; ++u) {
              const int r_local = 4 * u + (m >> 2);
              const int r = z * tiling.r_per_tile + r_local;
              if (r_local >= tiling.r_per_tile || r >= n_r) continue;
              out[((size_t)z * layout.n_units + unit) * per_unit + (size_t)lane * 4 + u] =
                  (float)(source(r, column) * prefactor);
            }
          }
        }
      }
    }
  }
}

void quad_emulate(const QuadLayout& layout, const QuadSchedule& schedule,
                  const QuadTiling& tiling, const std::vector<double>& table,
                  const double* densities, int64_t ldb, int64_t n_draws, int n_r,
                  bool separate, double* out, const QuadMergePlan* merge) {
  const int n_u = tiling.n_u, up = (n_u + 1) / 2, rt = 4 * n_u;
  const size_t per_unit = (size_t)up * 128;
  const int n_comp_out = separate ? (int)layout.comps.size() : 1;
  std::vector<double> partial((size_t)schedule.n_slabs * rt * 32, 0.0);
  // densities of bin `bin` (zero beyond the last one, as the buffer resource returns)
  auto density = [&](int bin, int64_t draw) {
    return bin < layout.n_bins ? densities[(size_t)bin * ldb + draw] : 0.0;
  };
  // LDS slots of the workgroup being emulated (merge plan)
  std::vector<double> stage(merge ? (size_t)std::max(1, merge->lds_slots) * rt * 32 : 0, 0.0);
  for (int w = 0; w < schedule.n_waves; ++w) {
    std::vector<double> f((size_t)n_u * 2 * 64, 0.0);      // F[u][set] per lane
    for (int ri = schedule.wave_runs[w]; ri < schedule.wave_runs[w + 1]; ++ri) {
      const QuadRun& run = schedule.runs[ri];
      const QuadComp& comp = layout.comps[run.comp];
      int rb = run.rb0, cb = run.cb0, left = run.count;
      int64_t unit = comp.unit_base + (comp.triangular ? (int64_t)rb * (rb + 1) / 2 + cb
                                                       : (int64_t)rb * comp.n_cb + cb);
      while (left > 0) {
        const int n = std::min(quad_row_length(comp, rb) - cb, left);
        left -= n;
        // D[u][set][v] per lane: r = 4 u + l / 16, i = i0 + v, draw = 2 (l % 16) + set
        std::vector<double> d((size_t)n_u * 2 * 4 * 64, 0.0);
        for (int t = 0; t < n; ++t, ++unit) {
          const double* a_unit =
              table.data() + ((size_t)run.rtile * layout.n_units + unit) * per_unit;
          for (int u = 0; u < n_u; ++u)
            for (int set = 0; set < 2; ++set)
              for (int l = 0; l < 64; ++l)      // output lane
                for (int v = 0; v < 4; ++v) {
                  const int m = (l >> 4) + 4 * v, col = l & 15;
                  double sum = d[(((size_t)u * 2 + set) * 4 + v) * 64 + l];
                  for (int k = 0; k < 4; ++k) {
                    const int lane_a = k * 16 + m, lane_b = k * 16 + col;
                    const double av = a_unit[((size_t)(u / 2) * 64 + lane_a) * 2 + (u & 1)];
                    const double bv = density(comp.j_bin0 + 4 * (cb + t) + (lane_b >> 4),
                                              (int64_t)run.tile * 32 + 2 * (lane_b & 15) + set);
                    sum = std::fma(av, bv, sum);
                  }
                  d[(((size_t)u * 2 + set) * 4 + v) * 64 + l] = sum;
                }
        }
        for (int u = 0; u < n_u; ++u)
          for (int set = 0; set < 2; ++set)
            for (int l = 0; l < 64; ++l)
              for (int v = 0; v < 4; ++v)
                f[((size_t)u * 2 + set) * 64 + l] = std::fma(
                    d[(((size_t)u * 2 + set) * 4 + v) * 64 + l],
                    density(comp.i_bin0 + 4 * rb + v, (int64_t)run.tile * 32 + 2 * (l & 15) + set),
                    f[((size_t)u * 2 + set) * 64 + l]);
        ++rb;
        cb = 0;
      }
      if (run.slab >= 0 || run.slab <= -2) {
        double* target = run.slab >= 0 ? partial.data() + (size_t)run.slab * rt * 32
                                       : stage.data() + (size_t)(-2 - run.slab) * rt * 32;
        for (int u = 0; u < n_u; ++u)
          for (int set = 0; set < 2; ++set)
            for (int l = 0; l < 64; ++l) {
              target[((size_t)4 * u + (l >> 4)) * 32 + 2 * (l & 15) + set] =
                  f[((size_t)u * 2 + set) * 64 + l];
              f[((size_t)u * 2 + set) * 64 + l] = 0.0;
            }
      }
    }
    // end of a workgroup: its merges, LDS slots added in slot order
    if (merge != nullptr &&
        ((w + 1) % merge->waves_per_block == 0 || w + 1 == schedule.n_waves)) {
      const int block = w / merge->waves_per_block;
      for (int e = merge->block_begin[block]; e < merge->block_begin[block + 1]; ++e) {
        const QuadMerge& m = merge->merges[e];
        for (int k = 0; k < rt * 32; ++k) {
          double sum = 0.0;
          for (int slot = m.first_slot; slot < m.first_slot + m.count; ++slot)
            sum += stage[(size_t)slot * rt * 32 + k];
          partial[(size_t)m.slab * rt * 32 + k] = sum;
        }
      }
    }
  }
  // finalize_quad_kernel's grouping
  for (int64_t b = 0; b < n_draws; ++b) {
    const int64_t tile32 = b / 32;
    for (int c = 0; c < n_comp_out; ++c)
      for (int r = 0; r < n_r; ++r) {
        const int z = r / tiling.r_per_tile, r_local = r % tiling.r_per_tile;
        const int64_t group = (tile32 * tiling.n_rtiles + z) * n_comp_out + c;
        double sum = 0.0;
        for (int s = schedule.group_begin[group]; s < schedule.group_begin[group + 1]; ++s)
          sum += partial[((size_t)s * rt + r_local) * 32 + b % 32];
        out[((size_t)b * n_comp_out + c) * n_r + r] = sum;
      }
  }
}

void build_chunking(const Plan& plan, int n_chunks, int waves_per_group,
                    Chunking& out) {
  out.waves_per_group = waves_per_group;
  out.chunks.clear();
  out.groups.clear();
  out.max_rows = 0;
  if (n_chunks < 1) n_chunks = 1;
  const int64_t total = std::max<int64_t>(1, plan.n_entries);
  const int eb = plan.block;

  std::vector<int> chunk_segment;
  for (size_t si = 0; si < plan.segments.size(); ++si) {
    const Segment& seg = plan.segments[si];
    const int64_t n_real = seg.n_real;
    if (n_real == 0) continue;
    const int64_t n_blocks = (n_real + eb - 1) / eb;
    // chunks of this segment, proportional to its share of the entries
    int64_t nc = (n_real * n_chunks + total / 2) / total;
    // full workgroups: a multiple of waves_per_group where there are several
    if (nc > waves_per_group)
      nc = (nc + waves_per_group / 2) / waves_per_group * waves_per_group;
    nc = std::max<int64_t>(1, std::min<int64_t>(nc, n_blocks));
    for (int64_t k = 0; k < nc; ++k) {
      const int64_t b0 = n_blocks * k / nc, b1 = n_blocks * (k + 1) / nc;
      Chunk chunk;
      chunk.q_begin = (int32_t)(seg.q_begin + b0 * eb);
      chunk.q_end = (int32_t)(seg.q_begin + b1 * eb);
      chunk.n_real = (int32_t)(std::min<int64_t>(n_real, b1 * eb) - b0 * eb);
      chunk.component = seg.component;
      out.chunks.push_back(chunk);
      chunk_segment.push_back((int)si);
    }
  }

  // workgroups: waves_per_group consecutive chunks of ONE segment
  size_t begin = 0;
  while (begin < out.chunks.size()) {
    const int seg = chunk_segment[begin];
    size_t end = begin;
    while (end < out.chunks.size() && end - begin < (size_t)waves_per_group &&
           chunk_segment[end] == seg)
      ++end;
    Group group;
    group.chunk_begin = (int32_t)begin;
    group.n_chunks = (int32_t)(end - begin);
    group.component = plan.segments[seg].component;
    int j_lo = plan.n_bins, j_hi = 0, i_lo = plan.n_bins, i_hi = 0;
    for (int k = 0; k < group.n_chunks; ++k) {
      const Chunk& chunk = out.chunks[begin + k];
      for (int32_t q = chunk.q_begin; q < chunk.q_begin + chunk.n_real; ++q) {
        j_lo = std::min(j_lo, plan.pos_j[q]);
        j_hi = std::max(j_hi, plan.pos_j[q] + 1);
        if (plan.pos_i[q] >= 0) {
          i_lo = std::min(i_lo, plan.pos_i[q]);
          i_hi = std::max(i_hi, plan.pos_i[q] + 1);
        }
      }
    }
    if (j_hi < j_lo) { j_lo = 0; j_hi = 0; }
    group.j_lo = j_lo;
    group.j_hi = j_hi;
    int rows = j_hi - j_lo;
    if (i_hi <= i_lo) {                       // mode cross: no row bins
      group.i_lo = group.i_hi = 0;
      group.i_shift = 0;
    } else if (i_lo >= j_lo && i_hi <= j_hi) {  // row bins are among the columns
      group.i_lo = group.i_hi = i_lo;
      group.i_shift = -j_lo;
    } else if (i_lo < j_hi && i_hi > j_lo) {  // overlapping: stage the union once
      const int lo = std::min(i_lo, j_lo), hi = std::max(i_hi, j_hi);
      group.j_lo = lo;
      group.j_hi = hi;
      group.i_lo = group.i_hi = i_lo;
      group.i_shift = -lo;
      rows = hi - lo;
    } else {                                  // disjoint: second region
      group.i_lo = i_lo;
      group.i_hi = i_hi;
      group.i_shift = (j_hi - j_lo) - i_lo;
      rows += i_hi - i_lo;
    }
    out.max_rows = std::max(out.max_rows, rows);
    out.groups.push_back(group);
    begin = end;
  }
}

// ---- pair counting: cell grid ----------------------------------------------------------------

namespace {
// Cells along one dimension and how many neighbour cells to each side hold the partners of a
// point.  Preferred: cells at least reach / 2 wide, two neighbours per side (5 x 5 x 5 cells
// of an eighth of the volume: 15.6 instead of 27 cell volumes of candidates per point); with
// fewer than five such cells: cells at least `reach` wide, one neighbour per side; with
// fewer than three of those: ONE cell (the minimum image does the wrapping).  No more cells
// than the points warrant (about 8 points per cell at least), at most 256.
int cells_along(double box, double reach, int64_t n_points, bool allow_fine, int* neighbours) {
  const int cap = std::max(
      3, (int)std::max<double>(1.0, std::cbrt((double)std::max<int64_t>(n_points, 1) / 8.0)));
  const int fine = std::min(std::min((int)std::floor(2.0 * box / reach), cap), 256);
  if (allow_fine && fine >= 5) {
    *neighbours = 2;
    return fine;
  }
  const int coarse = std::min(std::min((int)std::floor(box / reach), cap), 256);
  if (coarse >= 3) {
    *neighbours = 1;
    return coarse;
  }
  *neighbours = 0;
  return 1;
}
}  // namespace

CellGrid make_cell_grid(const double* boxsize, double reach_xy, double reach_z,
                        int64_t n_points, bool allow_fine) {
  CellGrid grid;
  grid.lx = boxsize[0];
  grid.ly = boxsize[1];
  grid.lz = boxsize[2];
  grid.nx = cells_along(grid.lx, reach_xy, n_points, allow_fine, &grid.reach_x);
  grid.ny = cells_along(grid.ly, reach_xy, n_points, allow_fine, &grid.reach_y);
  grid.nz = cells_along(grid.lz, reach_z, n_points, allow_fine, &grid.reach_z);
  return grid;
}

namespace {
// Host threads for the sorts of large point sets (the sorts are memory-bound scatter passes:
// 10^6 labelled points took as long on one core as a quarter of the pair count on the GPU).
int sort_threads(int64_t n) {
  if (n < 200000) return 1;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(hw == 0 ? 1u : hw, 16u));
}

template <typename Body>
void run_threads(int n_threads, Body body) {   // body(thread index)
  std::vector<std::thread> threads;
  for (int t = 1; t < n_threads; ++t) threads.emplace_back(body, t);
  body(0);
  for (std::thread& thread : threads) thread.join();
}
}  // namespace

// Counting sort by cell, stable (the points of a cell keep their input order) whatever the
// number of threads: thread t counts and later scatters the t-th contiguous chunk of points,
// its cursors start behind the points of the chunks before it.
int64_t sort_into_cells(const CellGrid& grid, const double* pos, const int32_t* label,
                        int64_t n, CellSort& out) {
  const int n_cells = grid.n_cells();
  const int n_threads = sort_threads(n);
  std::vector<int32_t> cell((size_t)n);
  std::vector<std::vector<int32_t>> cursor((size_t)n_threads);
  std::vector<int64_t> outside((size_t)n_threads, -1);
  run_threads(n_threads, [&](int t) {
    std::vector<int32_t>& count = cursor[t];
    count.assign((size_t)n_cells, 0);
    const int64_t begin = n * t / n_threads, end = n * (t + 1) / n_threads;
    for (int64_t p = begin; p < end; ++p) {
      const double x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
      if (!(x >= 0.0 && x <= grid.lx && y >= 0.0 && y <= grid.ly && z >= 0.0 && z <= grid.lz)) {
        outside[t] = p;
        return;
      }
      const int cx = std::min(grid.nx - 1, (int)(x / grid.lx * grid.nx));
      const int cy = std::min(grid.ny - 1, (int)(y / grid.ly * grid.ny));
      const int cz = std::min(grid.nz - 1, (int)(z / grid.lz * grid.nz));
      cell[p] = (cx * grid.ny + cy) * grid.nz + cz;
      ++count[cell[p]];
    }
  });
  for (int t = 0; t < n_threads; ++t)
    if (outside[t] >= 0) return outside[t];      // (the first one: chunks are in order)
  out.cell_start.assign((size_t)n_cells + 1, 0);
  int32_t running = 0;
  for (int c = 0; c < n_cells; ++c) {
    out.cell_start[c] = running;
    for (int t = 0; t < n_threads; ++t) {
      const int32_t here = cursor[t][c];
      cursor[t][c] = running;
      running += here;
    }
  }
  out.cell_start[n_cells] = running;
  out.x.resize(n);
  out.y.resize(n);
  out.z.resize(n);
  out.label.clear();
  if (label != nullptr) out.label.resize(n);
  run_threads(n_threads, [&](int t) {
    std::vector<int32_t>& next = cursor[t];
    const int64_t begin = n * t / n_threads, end = n * (t + 1) / n_threads;
    for (int64_t p = begin; p < end; ++p) {
      const int32_t slot = next[cell[p]]++;
      out.x[slot] = pos[3 * p];
      out.y[slot] = pos[3 * p + 1];
      out.z[slot] = pos[3 * p + 2];
      if (label != nullptr) out.label[slot] = label[p];
    }
  });
  return -1;
}

// Stable counting sort by label inside every cell (a cell's points fit the cache; cells much
// smaller than the label range fall back to a comparison sort), cells dealt out to the threads
// in chunks.
void sort_cells_by_label(CellSort& cells) {
  if (cells.label.empty()) return;
  const size_t n = cells.x.size();
  const size_t n_cells = cells.cell_start.size() - 1;
  int32_t max_label = 0;
  for (int32_t value : cells.label) max_label = std::max(max_label, value);
  CellSort sorted;
  sorted.cell_start = cells.cell_start;
  sorted.x.resize(n);
  sorted.y.resize(n);
  sorted.z.resize(n);
  sorted.label.resize(n);
  const int n_threads = sort_threads((int64_t)n);
  std::atomic<size_t> next_chunk(0);
  const size_t chunk = 64;
  run_threads(n_threads, [&](int) {
    std::vector<int32_t> offset((size_t)max_label + 2);
    std::vector<int32_t> order;
    for (;;) {
      const size_t c0 = next_chunk.fetch_add(chunk);
      if (c0 >= n_cells) break;
      for (size_t c = c0; c < std::min(n_cells, c0 + chunk); ++c) {
        const int32_t begin = cells.cell_start[c], end = cells.cell_start[c + 1];
        if ((int64_t)(end - begin) * 4 < (int64_t)max_label) {
          order.resize((size_t)(end - begin));
          std::iota(order.begin(), order.end(), begin);
          std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return cells.label[a] < cells.label[b];
          });
          for (int32_t k = 0; k < end - begin; ++k) {
            const int32_t from = order[k], to = begin + k;
            sorted.x[to] = cells.x[from];
            sorted.y[to] = cells.y[from];
            sorted.z[to] = cells.z[from];
            sorted.label[to] = cells.label[from];
          }
          continue;
        }
        std::fill(offset.begin(), offset.end(), 0);
        for (int32_t p = begin; p < end; ++p) ++offset[(size_t)cells.label[p] + 1];
        for (size_t l = 0; l + 1 < offset.size(); ++l) offset[l + 1] += offset[l];
        for (int32_t p = begin; p < end; ++p) {
          const int32_t to = begin + offset[cells.label[p]]++;
          sorted.x[to] = cells.x[p];
          sorted.y[to] = cells.y[p];
          sorted.z[to] = cells.z[p];
          sorted.label[to] = cells.label[p];
        }
      }
    }
  });
  cells = std::move(sorted);
}

void build_label_blocks(const CellSort& cells, int n_labels, int block, LabelBlocks& out) {
  out.block = std::max(1, block);
  out.n_blocks = std::max(1, (n_labels + out.block - 1) / out.block);
  const size_t n_cells = cells.cell_start.empty() ? 0 : cells.cell_start.size() - 1;
  const size_t stride = (size_t)out.n_blocks + 1;
  out.start.assign(n_cells * stride, 0);
  for (size_t c = 0; c < n_cells; ++c) {
    int32_t p = cells.cell_start[c];
    const int32_t end = cells.cell_start[c + 1];
    for (int k = 0; k <= out.n_blocks; ++k) {
      // (labels are sorted inside the cell: advance to the first label of block k)
      while (k < out.n_blocks && p < end && cells.label[p] < k * out.block) ++p;
      out.start[c * stride + k] = k < out.n_blocks ? p : end;
    }
  }
}

void build_pair_units(const CellGrid& grid, const CellSort& set1, const CellSort& set2,
                      int n_blocks1, int n_blocks2, int target_units, PairUnits& out) {
  out = PairUnits();
  const int n_cells = grid.n_cells();
  // candidate pairs per cell: its set-1 points x the set-2 points of the cells around it
  std::vector<double> weight((size_t)n_cells, 0.0);
  double total = 0.0;
  for (int cx = 0; cx < grid.nx; ++cx)
    for (int cy = 0; cy < grid.ny; ++cy)
      for (int cz = 0; cz < grid.nz; ++cz) {
        const int c = (cx * grid.ny + cy) * grid.nz + cz;
        const double n1 = set1.cell_start[c + 1] - set1.cell_start[c];
        if (n1 == 0.0) continue;
        double n2 = 0.0;
        for (int ox = -grid.reach_x; ox <= grid.reach_x; ++ox)
          for (int oy = -grid.reach_y; oy <= grid.reach_y; ++oy)
            for (int oz = -grid.reach_z; oz <= grid.reach_z; ++oz) {
              const int other = (((cx + ox + grid.nx) % grid.nx) * grid.ny +
                                 (cy + oy + grid.ny) % grid.ny) * grid.nz +
                                (cz + oz + grid.nz) % grid.nz;
              n2 += set2.cell_start[other + 1] - set2.cell_start[other];
            }
        // (+ a fixed cost per visited cell: the neighbour ranges are set up per cell)
        weight[c] = n1 * n2 + 256.0;
        total += weight[c];
        out.max_cell_candidates = std::max(out.max_cell_candidates, n1 * n2);
      }
  const int pairs = std::max(1, n_blocks1 * n_blocks2);
  const int ranges = std::max(1, std::min(n_cells, (target_units + pairs - 1) / pairs));
  std::vector<int32_t> cut(1, 0);
  double running = 0.0;
  for (int c = 0; c < n_cells; ++c) {
    running += weight[c];
    if ((int)cut.size() < ranges && running >= total * (double)cut.size() / ranges)
      cut.push_back(c + 1);
  }
  if (cut.back() != n_cells) cut.push_back(n_cells);
  for (int a = 0; a < n_blocks1; ++a)
    for (int b = 0; b < n_blocks2; ++b)
      for (size_t r = 0; r + 1 < cut.size(); ++r) {
        if (cut[r] == cut[r + 1]) continue;
        out.block1.push_back(a);
        out.block2.push_back(b);
        out.cell_begin.push_back(cut[r]);
        out.cell_end.push_back(cut[r + 1]);
      }
}

}  // namespace tc
