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

// The contraction walks the table in "positions".  The packed pairs are cut into
// SEGMENTS, each belonging to one output component (0 cen-cen | cen, 1 cen-sat | sat,
// 2 sat-sat) and shaped either as a triangle (rows i in [i_lo, i_hi), columns j in
// [j_lo, i]) or as a rectangle (rows [i_lo, i_hi) x columns [j_lo, j_hi)); mode cross
// uses one-row "rectangles" without a row bin (i = -1).  Segments are sized so that the
// density rows a workgroup has to stage (its column range plus its row range) fit in
// LDS, which also lifts any limit on the number of bins.  Inside a segment the entries
// are row-major and are followed by zero padding up to a multiple of the block size of
// the table layout (8 positions: two matrix-core steps, kernels.hip.h).
struct Segment {
  int32_t component;
  int32_t rectangular;   // 0: triangle (row i ends at column i), 1: rectangle
  int32_t i_lo, i_hi;    // row bins (library order); i_lo = -1 in mode cross
  int32_t j_lo, j_hi;    // column bins [j_lo, j_hi) (triangle: j_hi unused)
  int64_t q_begin;       // first position
  int64_t n_real;        // entries before the padding
};

// The work of one wavefront: positions [q_begin, q_end) of one segment, of which the
// first n_real are real entries.
struct Chunk {
  int32_t q_begin;
  int32_t q_end;
  int32_t n_real;
  int32_t component;
};

// The work of one workgroup: chunks [chunk_begin, chunk_begin + n_chunks), all of one
// segment, and the rows of the per-draw number densities it stages in LDS: the
// column bins [j_lo, j_hi) first and, unless they lie inside that range, the row bins
// [i_lo, i_hi) behind them; bin i is found at LDS row i + i_shift.
struct Group {
  int32_t chunk_begin;
  int32_t n_chunks;
  int32_t j_lo;
  int32_t j_hi;
  int32_t component;
  int32_t i_lo;
  int32_t i_hi;        // i_hi == i_lo: nothing staged separately
  int32_t i_shift;
};

struct Plan {
  int mode = 0;
  int n_bins = 0;
  int n_central = 0;
  int n_components = 0;          // 3 (auto) or 2 (cross)
  int block = 1;                 // positions per block of the table layout
  int64_t n_entries = 0;         // = P
  int64_t n_positions = 0;       // padded, multiple of `block` per segment
  std::vector<int32_t> perm;     // library bin g' -> reference row
  std::vector<Segment> segments;
  // Position order of the re-laid-out table: reference column (-1 = padding),
  // prefactor and bin pair (library order).
  std::vector<int64_t> column;
  std::vector<int8_t> prefactor;
  std::vector<int32_t> pos_i, pos_j;
};

struct Chunking {
  int waves_per_group = 0;
  std::vector<Chunk> chunks;
  std::vector<Group> groups;
  int max_rows = 0;              // max over groups of the LDS rows staged
};

// Bin permutation (stable sort, centrals first), segmentation (no workgroup will need
// more than about `row_budget` density rows) and position order.
void build_plan(int mode, int n_bins, const uint8_t* is_central, int block,
                int row_budget, Plan& plan);

// Cut the positions into about n_chunks wave-sized chunks (multiples of the
// block size, never crossing a segment) and pack them waves_per_group at a
// time into workgroups.
void build_chunking(const Plan& plan, int n_chunks, int waves_per_group,
                    Chunking& out);

// Position order inside a segment: j -> j + 1, wrapping to (i + 1, j_lo) after column
// `j_last` (rectangle) or after the diagonal (j_last < 0).
inline void advance_pair(int j_lo, int j_last, int& i, int& j) {
  const int last = j_last >= 0 ? j_last : i;
  if (++j > last) {
    ++i;
    j = j_lo;
  }
}

}  // namespace tc
