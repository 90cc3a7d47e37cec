// Host-side mathematics and work planning (see hostmath.h).
#include "hostmath.h"

#include <algorithm>
#include <cmath>
#include <numeric>

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

void build_plan(int mode, int n_bins, const uint8_t* is_central, Plan& plan) {
  plan.mode = mode;
  plan.n_bins = n_bins;
  plan.perm.clear();
  for (int g = 0; g < n_bins; ++g)
    if (is_central[g]) plan.perm.push_back(g);
  plan.n_central = (int)plan.perm.size();
  for (int g = 0; g < n_bins; ++g)
    if (!is_central[g]) plan.perm.push_back(g);
  const int gc = plan.n_central;

  plan.entry_column.clear();
  plan.entry_prefactor.clear();
  plan.entry_component.clear();
  if (mode == 0) {
    plan.n_components = 3;
    // component-major, then row-major: all cen-cen pairs, all cen-sat pairs,
    // all sat-sat pairs.
    for (int comp = 0; comp < 3; ++comp) {
      for (int i = 0; i < n_bins; ++i) {
        int j_lo, j_hi;
        if (comp == 0) {
          if (i >= gc) continue;
          j_lo = 0; j_hi = i + 1;
        } else if (comp == 1) {
          if (i < gc) continue;
          j_lo = 0; j_hi = gc;
        } else {
          if (i < gc) continue;
          j_lo = gc; j_hi = i + 1;
        }
        for (int j = j_lo; j < j_hi; ++j) {
          int a = plan.perm[i], b = plan.perm[j];
          plan.entry_column.push_back(packed_index(a, b));
          plan.entry_prefactor.push_back(a == b ? 1 : 2);
          plan.entry_component.push_back((int8_t)comp);
        }
      }
    }
  } else {
    plan.n_components = 2;
    for (int j = 0; j < n_bins; ++j) {
      plan.entry_column.push_back(plan.perm[j]);
      plan.entry_prefactor.push_back(1);
      plan.entry_component.push_back(j < gc ? 0 : 1);
    }
  }
  plan.n_entries = (int64_t)plan.entry_column.size();
}

namespace {

// Segments of one component in entry order.
void component_segments(const Plan& plan, int comp, std::vector<Segment>& segs,
                        int64_t& e0) {
  const int gc = plan.n_central, g = plan.n_bins;
  if (plan.mode == 0) {
    for (int i = 0; i < g; ++i) {
      int j_lo, j_hi;
      if (comp == 0) {
        if (i >= gc) continue;
        j_lo = 0; j_hi = i + 1;
      } else if (comp == 1) {
        if (i < gc) continue;
        j_lo = 0; j_hi = gc;
      } else {
        if (i < gc) continue;
        j_lo = gc; j_hi = i + 1;
      }
      if (j_hi <= j_lo) continue;
      segs.push_back({i, j_lo, j_hi - j_lo, (int32_t)e0});
      e0 += j_hi - j_lo;
    }
  } else {
    int j_lo = comp == 0 ? 0 : gc;
    int j_hi = comp == 0 ? gc : g;
    if (j_hi > j_lo) {
      segs.push_back({-1, j_lo, j_hi - j_lo, (int32_t)e0});
      e0 += j_hi - j_lo;
    }
  }
}

}  // namespace

void build_chunking(const Plan& plan, int n_chunks, int waves_per_group,
                    Chunking& out) {
  out.waves_per_group = waves_per_group;
  out.segments.clear();
  out.chunks.clear();
  out.groups.clear();
  out.max_rows = 0;
  if (n_chunks < 1) n_chunks = 1;

  std::vector<std::vector<Segment>> comp_segs(plan.n_components);
  std::vector<int64_t> comp_entries(plan.n_components, 0);
  int64_t e0 = 0;
  for (int c = 0; c < plan.n_components; ++c) {
    int64_t begin = e0;
    component_segments(plan, c, comp_segs[c], e0);
    comp_entries[c] = e0 - begin;
  }
  const int64_t total = std::max<int64_t>(1, plan.n_entries);

  for (int c = 0; c < plan.n_components; ++c) {
    if (comp_entries[c] == 0) continue;
    // chunks of this component, proportional to its share of the entries
    int64_t nc = (comp_entries[c] * n_chunks + total / 2) / total;
    nc = std::max<int64_t>(1, std::min<int64_t>(nc, comp_entries[c]));
    int64_t done = 0;       // entries of this component already assigned
    size_t s = 0;           // current segment
    int32_t used = 0;       // entries of segment s already assigned
    for (int64_t k = 0; k < nc; ++k) {
      int64_t target = comp_entries[c] * (k + 1) / nc - done;
      Chunk chunk;
      chunk.seg_begin = (int32_t)out.segments.size();
      chunk.component = c;
      chunk.n_entries = (int32_t)target;
      while (target > 0) {
        const Segment& seg = comp_segs[c][s];
        int32_t take = (int32_t)std::min<int64_t>(target, seg.len - used);
        out.segments.push_back({seg.i, seg.j0 + used, take, seg.e0 + used});
        used += take;
        target -= take;
        done += take;
        if (used == seg.len) {
          ++s;
          used = 0;
        }
      }
      chunk.seg_end = (int32_t)out.segments.size();
      out.chunks.push_back(chunk);
    }
  }

  for (size_t begin = 0; begin < out.chunks.size(); begin += waves_per_group) {
    Group group;
    group.chunk_begin = (int32_t)begin;
    group.n_chunks = (int32_t)std::min<size_t>(waves_per_group,
                                                out.chunks.size() - begin);
    int lo = plan.n_bins, hi = 0;
    for (int k = 0; k < group.n_chunks; ++k) {
      const Chunk& chunk = out.chunks[begin + k];
      for (int s = chunk.seg_begin; s < chunk.seg_end; ++s) {
        const Segment& seg = out.segments[s];
        lo = std::min(lo, seg.j0);
        hi = std::max(hi, seg.j0 + seg.len);
        if (seg.i >= 0) {
          lo = std::min(lo, seg.i);
          hi = std::max(hi, seg.i + 1);
        }
      }
    }
    if (hi < lo) { lo = 0; hi = 0; }
    group.row_lo = lo;
    group.row_hi = hi;
    out.max_rows = std::max(out.max_rows, hi - lo);
    out.groups.push_back(group);
  }
}

}  // namespace tc
