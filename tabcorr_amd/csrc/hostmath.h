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

// Groups of bins that share their Gauss-Legendre nodes: the same log_prim_haloprop_min / max
// (tabcorr.py:548-549: the nodes depend on nothing else) and the same galaxy type -- the
// secondary-percentile bins of one mass bin (tabcorr.py:186-205).  Bins are in library order
// (centrals first: the first n_central); group i = member[begin[i] .. begin[i + 1]), groups in
// the order of their first member, members ascending; groups of centrals come first.
struct NodeGroups {
  std::vector<int32_t> begin;    // n_groups + 1
  std::vector<int32_t> member;   // n_bins
  int n_groups = 0;
  int n_central_groups = 0;
  int largest = 0;               // members of the largest group
};
void find_node_groups(int n_bins, int n_central, const double* log_min, const double* log_max,
                      NodeGroups& out);

// Cut the positions into about n_chunks wave-sized chunks (multiples of the
// block size, never crossing a segment) and pack them waves_per_group at a
// time into workgroups.
void build_chunking(const Plan& plan, int n_chunks, int waves_per_group,
                    Chunking& out);

// ---- quadratic-form contraction (mode auto, float64) ------------------------------------
//
// sum_p T[r][p] c_p n_i n_j = sum_i n_i (sum_{j <= i} c_ij T_r[i][j] n_j): the inner sum is a
// matrix product over j with the density rows themselves as one operand (no pair weights
// to form), the outer factor n_i is applied once per row of 4 x 4 bin blocks.  The work
// is cut into UNITS: one 4 x 4 block of bin pairs (block row rb, block column cb of one
// component) x all r values of one r tile x one tile of 32 draws = 2 U matrix-core
// instructions (U = r sub-tiles of 4 per r tile, <= 5).  A layout lists the components
// (the whole triangle, or cen-cen triangle / cen-sat rectangle / sat-sat triangle when
// the output is separated by galaxy type: each type is then padded to whole blocks) and
// numbers their units row-major; the table is stored per r tile and unit as
// [u pair][lane = k * 16 + r_local + 4 i_local][2] doubles (kernels.hip.h).
struct QuadComp {
  int32_t component;     // 0 cen-cen (or everything), 1 cen-sat, 2 sat-sat
  int32_t triangular;    // block columns 0 .. rb of block row rb, else all n_cb
  int32_t i_bin0;        // first row bin (library order)
  int32_t i_count;
  int32_t j_bin0;        // first column bin
  int32_t j_count;
  int32_t n_rb;          // block rows
  int32_t n_cb;          // block columns (triangle: n_rb)
  int64_t unit_base;     // first unit of the component in the layout
  int64_t n_units;
};

struct QuadLayout {
  int n_bins = 0;
  int n_central = 0;
  bool by_type = false;
  std::vector<QuadComp> comps;   // 1 (whole triangle) or 3
  int64_t n_units = 0;
};

void build_quad_layout(int n_bins, int n_central, bool by_type, QuadLayout& out);

// Units of block row rb of a component.
inline int quad_row_length(const QuadComp& comp, int rb) {
  return comp.triangular ? rb + 1 : comp.n_cb;
}

// The work of one wavefront is a list of RUNS: `count` consecutive units of one (draw
// tile, r tile, component, table), starting at block (rb0, cb0).  After a run with
// slab >= 0 the wave writes its sums to that slab of the partial buffer and clears them.
struct QuadRun {
  int32_t tile;      // tile of 32 draws
  int32_t rtile;
  int32_t comp;      // index into the layout's components
  int32_t table;     // interpolator: table index, else 0
  int32_t rb0;
  int32_t cb0;
  int32_t count;
  int32_t slab;
};

// The linearised unit space (draw tile, r tile, component, table, unit), cut into equal
// contiguous ranges, one per wave -- every wave gets the same number of matrix-core
// instructions whatever the batch size.  Slabs are numbered in run order, so the slabs of
// one output group (draw tile, r tile[, component when `separate`]) are consecutive:
// group g owns slabs [group_begin[g], group_begin[g + 1]).
struct QuadSchedule {
  int n_waves = 0;
  int n_slabs = 0;
  int n_groups = 0;
  std::vector<QuadRun> runs;
  std::vector<int32_t> wave_runs;     // n_waves + 1
  std::vector<int32_t> group_begin;   // n_groups + 1
};

// `table_major` (interpolators): the linearised space is (table, draw tile, r tile, component,
// unit) instead of (draw tile, r tile, component, table, unit) -- consecutive shares then
// walk the SAME matrix over consecutive draw tiles, so that with an eighth of the shares per
// XCD every L2 holds the ~K / 8 matrices its waves use instead of streaming all K through it
// once per draw tile.  An output group then receives slabs from K tables; slabs are numbered
// so that those of one group stay consecutive (in table order).
// kQuadRtileMajor (matrices beyond one L2: hundreds of r values): the r tiles are walked one
// after the other by ALL waves, each taking an equal share of that r tile's (draw tile,
// component, unit) space -- with an eighth of the shares per XCD every L2 holds the slice of
// the matrix of the r tile in flight (1 / n_rtiles of it) instead of streaming the whole
// matrix once per draw tile.
// kQuadUnitMajor (one table of 2 - 4 MB, e.g. BASELINE configs[2]'s 3.2 MB by-type matrix, which
// does not stay in a 4 MB L2 next to the streaming density rows: 56 % of the passes re-read it
// from the Infinity Cache): the units of a (draw tile, r tile) are cut into eight parts and the
// linearised space is (part, draw tile, r tile, units of the part) -- with an eighth of the
// shares per XCD every L2 keeps ONE eighth of the matrix for all draw tiles.  An output group
// then receives slabs from all eight parts.
// kQuadTableSync (interpolators whose matrices do not fit the L2s next to each other: the
// reference's database grids of up to 4 x 4 x 4 tables): table-major, but the waves of an XCD
// -- an eighth of the shares, `n_waves` a multiple of 8 -- walk the tables of that XCD's range ONE
// AFTER THE OTHER, each wave taking an equal slice of every table's (draw tile, r tile, unit)
// space in turn: at any time the XCD's waves read one matrix instead of all K / 8 of them
// (64 tables of 120 bins: 12 MB per L2 of 4 MB, 23 GB of fabric reads per 10^4 draws).
// kQuadUnitSync (one table whose matrix is far beyond the L2s: BASELINE configs[4], 38 r tiles of
// 4 MB in float64): the (r tile, sub-range of the units, draw tile) items -- S sub-ranges per
// (draw tile, r tile), S chosen so that the rounds fill -- are dealt to the XCDs in that order
// and inside an XCD round-robin to its waves, so that at any time the 256 waves of an XCD walk
// the SAME units of the same r tile for different draw tiles: the matrix streams through every
// L2 once per round instead of once per wave.  Shares differ by one item (a few per cent).
constexpr int kQuadTileMajor = 0, kQuadTableMajor = 1, kQuadRtileMajor = 2, kQuadUnitMajor = 3,
              kQuadTableSync = 4, kQuadUnitSync = 5;
constexpr int kQuadUnitParts = 8;
void build_quad_schedule(const QuadLayout& layout, int n_tiles, int n_rtiles, int n_tables,
                         bool separate, int max_waves, int min_units_per_wave,
                         QuadSchedule& out, int order = kQuadTileMajor);

// Workgroup-level merging of the slabs.  Consecutive waves walk consecutive ranges, so the
// waves of one workgroup mostly end / start inside the same output group and each would
// write its own slab of it.  After this pass a run's `slab` is
//   >= 0   a slab of the partial buffer the wave writes itself (workgroups with more
//          flushes than `max_slots` keep this direct form), or
//   <= -2  LDS slot -2 - slab of its workgroup,
// and per workgroup a list of merges says which consecutive LDS slots are added (in slot
// order) into which slab of the partial buffer.  Slabs stay numbered in group order.
struct QuadMerge {
  int32_t slab;         // destination in the partial buffer
  int32_t first_slot;   // LDS slots first_slot .. first_slot + count - 1
  int32_t count;
  int32_t pad;
};
struct QuadMergePlan {
  int waves_per_block = 0;
  int n_blocks = 0;
  int lds_slots = 0;                       // most LDS slots any workgroup needs
  std::vector<int32_t> block_begin;        // (n_blocks + 1) into merges
  std::vector<QuadMerge> merges;
};
void merge_quad_schedule(const QuadLayout& layout, int n_rtiles, bool separate,
                         int waves_per_block, int max_slots, QuadSchedule& schedule,
                         QuadMergePlan& plan);

// The units of a triangular component (block rows rb = 0 .. n_rb - 1 with rb + 1 units each,
// row-major) cut into `n_parts` equal contiguous parts -- the quarters the waves of
// predict_fused_kernel walk: first block row, block column and number of units of every part.
void triangle_parts(int n_rb, int n_parts, int32_t* rb0, int32_t* cb0, int32_t* count);

// r tiling of the quadratic-form kernel: n_rtiles tiles of r_per_tile values (the last one
// may hold fewer), n_u = ceil(r_per_tile / 4) <= 5 sub-tiles of 4.
struct QuadTiling {
  int n_rtiles = 1;
  int r_per_tile = 0;
  int n_u = 0;
};
QuadTiling quad_tiling(int n_r);

// The re-laid-out matrix of a layout: (n_rtiles, n_units, (n_u + 1) / 2, 64 lanes, 2)
// doubles with the pair prefactor folded in; `matrix` is the reference's (n_r, n_pairs)
// tpcf_matrix in float64 or float32, `perm` the library's bin order.
void fill_quad_table(const QuadLayout& layout, const std::vector<int32_t>& perm, int n_r,
                     int64_t n_pairs, const void* matrix, bool matrix_is_f32,
                     const QuadTiling& tiling, std::vector<double>& out);

// float32 variant (contract_quad_f32_kernel, v_mfma_f32_16x16x4_f32): r tiles of 16 values
// (n_u <= 4 sub-tiles of 4), draw tiles of 64 (four column sets of 16); per r tile and unit
// [lane = k * 16 + i_local + 4 r_local][u] floats -- one 16-byte load per lane carries the A
// operand of all four sub-tiles.
QuadTiling quad_tiling_f32(int n_r);
void fill_quad_table_f32(const QuadLayout& layout, const std::vector<int32_t>& perm, int n_r,
                         int64_t n_pairs, const void* matrix, bool matrix_is_f32,
                         const QuadTiling& tiling, std::vector<float>& out);

// TEST INFRASTRUCTURE (never called by the product): executes a schedule on the host the
// way contract_quad_kernel + finalize_quad_kernel do -- same operand lanes, same unit
// walk, same slab grouping -- so that the layout, the schedule and the grouping can be
// checked without a GPU.  densities: (n_bins, ldb); out: (n_draws, n_comp_out, n_r) sums
// before the normalisation.
void quad_emulate(const QuadLayout& layout, const QuadSchedule& schedule,
                  const QuadTiling& tiling, const std::vector<double>& table,
                  const double* densities, int64_t ldb, int64_t n_draws, int n_r,
                  bool separate, double* out, const QuadMergePlan* merge = nullptr);

// ---- pair counting (paircount.hip): cell grid of a periodic box ----------------------------
//
// Per dimension: cells at least reach / 2 wide with two neighbour cells per side, or (fewer
// than five of those) at least `reach` wide with one, or ONE cell and no neighbour offsets
// (the minimum image does the wrapping there) -- the partners of a point always lie in the
// neighbour cells around its own.
struct CellGrid {
  int nx = 1, ny = 1, nz = 1;
  int reach_x = 0, reach_y = 0, reach_z = 0;   // neighbour cells per side: 2, 1 or 0
  double lx = 0, ly = 0, lz = 0;               // box size
  int n_cells() const { return nx * ny * nz; }
};
CellGrid make_cell_grid(const double* boxsize, double reach_xy, double reach_z,
                        int64_t n_points, bool allow_fine = true);

// Points of one set sorted by cell (counting sort): coordinates, labels (if given) and the
// cell offsets.  Returns -1, or the index of the first point outside [0, box].
struct CellSort {
  std::vector<double> x, y, z;
  std::vector<int32_t> label, cell_start;   // cell_start: n_cells + 1
};
int64_t sort_into_cells(const CellGrid& grid, const double* pos, const int32_t* label,
                        int64_t n, CellSort& out);
// Inside every cell, order the points by label (stable).
void sort_cells_by_label(CellSort& cells);

// Labelled pair count (paircount.hip: pair_count_blocks_kernel): the labels are cut into blocks
// of `block` consecutive values; with the points sorted by (cell, label), start[(cell,
// k)] = first point of `cell` whose label is >= k * block, k = 0 .. n_blocks (the last entry
// is the end of the cell): the points of one cell and one label block are contiguous.
struct LabelBlocks {
  int block = 1;
  int n_blocks = 1;
  std::vector<int32_t> start;   // (n_cells, n_blocks + 1)
};
void build_label_blocks(const CellSort& cells, int n_labels, int block, LabelBlocks& out);

// Work units of that kernel: (set-1 label block, set-2 label block, range of cells), the cell
// ranges cut so that every unit holds about the same number of candidate pairs (points of a
// cell x points in the cells around it) and all blocks together give ~`target_units` units.
struct PairUnits {
  std::vector<int32_t> block1, block2, cell_begin, cell_end;
  double max_cell_candidates = 0.0;   // largest (points of a cell) x (points around it)
};
void build_pair_units(const CellGrid& grid, const CellSort& set1, const CellSort& set2,
                      int n_blocks1, int n_blocks2, int target_units, PairUnits& out);

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
