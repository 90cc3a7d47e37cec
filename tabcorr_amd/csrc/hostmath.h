// Host-side mathematics and work planning of libtabcorr_hip.so (no GPU code).
//
// Everything here is what the reference computes once per table or per
// interpolator and caches on the Python object; file:line citations are into
// johannesulf/TabCorr v1.2.0.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace tc {

// Gauss-Legendre nodes on (0, 1) and weights on (-1, 1), ascending, as
// tabcorr/tabcorr.py:543-546.
void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w);

// Packed lower-triangle column of the pair (i, j): tabcorr/tabcorr.py:770-806.
inline int64_t packed_index(int64_t i, int64_t j) {
  return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i;
}

// Not-a-knot cubic spline matrix, tabcorr/interpolator.py:219-272.  Returns
// false if the linear system is singular (repeated abscissae).
bool spline_interpolation_matrix(int n, const double* xp, std::vector<double>& a);

// One run of consecutive entries that share the row weight n_i:
// weights are n_i * n_j for j = j0 .. j0 + len - 1 (mode auto) or n_j alone
// (mode cross, i = -1); the entries start at position e0 of the re-laid-out
// table.  Indices refer to the library's bin order (centrals first).
struct Segment {
  int32_t i;
  int32_t j0;
  int32_t len;
  int32_t e0;
};

// The work of one wavefront: segments [seg_begin, seg_end), all of one
// component (0 cen-cen | cen, 1 cen-sat | sat, 2 sat-sat).
struct Chunk {
  int32_t seg_begin;
  int32_t seg_end;
  int32_t component;
  int32_t n_entries;
};

// The work of one workgroup: chunks [chunk_begin, chunk_begin + n_chunks) and
// the rows [row_lo, row_hi) of the per-draw number densities it stages in LDS.
struct Group {
  int32_t chunk_begin;
  int32_t n_chunks;
  int32_t row_lo;
  int32_t row_hi;
};

struct Plan {
  int mode = 0;
  int n_bins = 0;
  int n_central = 0;
  int n_components = 0;          // 3 (auto) or 2 (cross)
  int64_t n_entries = 0;         // = P
  std::vector<int32_t> perm;     // library bin g' -> reference row
  // Entry order of the re-laid-out table: reference column and prefactor.
  std::vector<int64_t> entry_column;
  std::vector<int8_t> entry_prefactor;
  std::vector<int8_t> entry_component;
};

struct Chunking {
  int waves_per_group = 0;
  std::vector<Segment> segments;
  std::vector<Chunk> chunks;
  std::vector<Group> groups;
  int max_rows = 0;              // max over groups of row_hi - row_lo
};

// Bin permutation (stable sort, centrals first) and entry order.
void build_plan(int mode, int n_bins, const uint8_t* is_central, Plan& plan);

// Cut the entries into about n_chunks wave-sized chunks that never mix
// components, and pack them waves_per_group at a time into workgroups.
void build_chunking(const Plan& plan, int n_chunks, int waves_per_group,
                    Chunking& out);

}  // namespace tc
