// Argument blocks and compile-time constants of the kernels in kernels.hip.h, visible to
// the host-only translation units (table.cpp, interp.cpp) that fill them in.
#pragma once

#include <cstdint>

#include "hostmath.h"

namespace tc {

constexpr int kLanes = 64;
constexpr unsigned kFlagSeparate = 1u;
constexpr unsigned kFlagModulate = 2u;
constexpr unsigned kFlagAssembias = 4u;
constexpr int kOccWaves = 4;
constexpr int kF32Block = 8;    // entries per block
constexpr int kF64Block = 8;    // positions per block of the FP64 table layout (two steps)
constexpr int kF32Tile = 32;    // r values per tile
constexpr int kFinalizeRows = 32;   // (component, r) rows per LDS pass
constexpr int kMaxInterpDim = 8;
constexpr int kMaxInterpAxis = 32;

// Constants of the GROUPED occupation kernels (kernels.hip.h: occ_group_zheng07): the nodes per
// group, everything else per member in group order (launch.hip: get_quadrature).
struct GroupArgs {
  const int32_t* begin;      // (n_groups + 1) first member index of every group
  const int32_t* member;     // (n_bins) library bin of every member index
  const double* log_m;       // (n_groups, 10) log10 of the node masses
  const double* m;           // (n_groups, 10) node masses
  const double* weight;      // (n_bins, 10) weights in member order, then (n_bins) their sums
  const double* n_h;         // (n_bins) in member order
  const double* percentile;  // (n_bins) in member order
  // moment expansion of the central bins' node sums (series.h), per member in group order:
  // (n_bins, series::kStride) doubles and (n_bins, series::kThresholds) int32; NULL: off
  const double* series;
  const int32_t* series_thr;
  // ... of the satellite bins (series.h, namespace sat): (n_bins, sat::kStride) doubles and
  // (n_bins, sat::kThresholds) int32
  const double* sat_series;
  const int32_t* sat_series_thr;
  // everything a group's expansions read in one record per group (series.h, namespace record:
  // (n_groups, record::kStride) doubles); NULL: a group with more than two members exists
  const double* records;
};

struct OccArgs {
  const double* theta;     // (n_draws, n_theta) row-major
  int n_theta;
  int64_t n_draws;
  int64_t ldb;             // draws rounded up to a multiple of 64
  int n_bins;
  int n_central;
  int n_gauss;
  int n_tiles;             // draw tiles (ldb / 64)
  int n_splits;            // bin ranges per draw tile
  unsigned flags;
  double split;
  const double* log_m;     // (n_bins, n_gauss) log10 of the node masses
  const double* m;         // (n_bins, n_gauss) node masses
  const double* weight;    // (n_bins, n_gauss) normalised quadrature weights
  const double* n_h;       // (n_bins)
  const double* percentile;  // (n_bins)
  const int32_t* perm;     // library bin -> reference row
  const double* math_table;  // fm::kTableDoubles doubles (fastmath.h)
  double* nbuf;            // (n_bins, ldb) number density per bin and draw
  float* nbuf32;           // optional float copy of it (float32 quadratic-form kernel)
  double* ngal;            // (bin splits, 2, ldb) partial cen / sat densities
  double* occupation;      // optional (n_draws, n_bins) in reference order
  // Groups of bins with the same quadrature nodes (the secondary-percentile bins of one mass
  // bin; table.cpp: find_node_groups): group i = bins group_member[group_begin[i] ..
  // group_begin[i + 1]) in library order, the groups of centrals first.  Used by the GROUPED
  // kernels only, which split the groups -- not the bins -- into n_splits ranges.
  int n_groups;
  int n_central_groups;
  GroupArgs group;
  // moment expansion of the central bins' node sums (series.h), per bin: NULL: off
  const double* series;
  const int32_t* series_thr;
  const double* sat_series;     // ... of the satellite bins (namespace sat)
  const int32_t* sat_series_thr;
};

// Un-batched predict(): one draw through one launch (single_draw_kernel).
struct SingleArgs {
  double theta_value[7];     // the draw, by value
  int n_theta;
  int n_bins;
  int n_central;
  int n_gauss;
  unsigned flags;
  double split;
  const double* log_m;       // quadrature constants as in OccArgs
  const double* m;
  const double* weight;
  const double* n_h;
  const double* percentile;
  const double* math_table;
  const double* table;       // re-laid-out matrix of the single r tile
  const int32_t* pos_off;
  int64_t n_positions;
  int rt;
  int n_r;
  int mode;
  double* partial;           // (blocks, rt) partial sums, page-locked host memory
  double* ngal;              // (2) centrals / satellites number density, host memory
  // Interpolator (un-batched Interpolator.predict): grid = n_tables x blocks_per_table;
  // workgroup b serves table b / blocks_per_table with the bins of that table's class.
  // n_tables == 0: one table, the direct pointers above.  Per table the host memory holds
  // partial (n_tables, blocks_per_table, rt) and ngal (n_tables, 2).
  int n_tables;
  int blocks_per_table;
  const double* const* tables;       // (n_tables) re-laid-out matrices
  const int32_t* table_class;        // (n_tables)
  const double* const* class_log_m;  // (n_classes) quadrature constants and bin columns
  const double* const* class_m;
  const double* const* class_weight;
  const double* const* class_n_h;
  const double* const* class_percentile;
  unsigned long long* stamps;   // developer timeline: 8 x 100 MHz stamps per block, or NULL
  // Several independent draws in one launch (an ensemble sampler proposes many points per
  // step): grid = n_walkers x blocks_per_table, workgroup b serves draw b / blocks_per_table
  // whose parameters are theta_many[draw][n_theta] (page-locked host memory); partial
  // (n_walkers, blocks_per_table, rt) and ngal (n_walkers, 2).  n_walkers == 0: one draw,
  // theta_value.
  const double* theta_many;
  int n_walkers;
  // Completion without a stream synchronisation: every workgroup, after its last store to
  // host memory, sets done[workgroup] = epoch (system-scope release); the host polls.
  unsigned long long* done;
  unsigned long long epoch;
  // Resident form (resident_draw_kernel: the launch itself is the larger part of an un-batched
  // call): the workgroups stay on the chip between calls and take every draw from a mailbox
  // in page-locked host memory -- seven 16-byte entries {parameter, number of the call it
  // belongs to} (the host writes the number behind the parameter; kResidentStop: leave),
  // `epoch` = the first call to serve.  A workgroup leaves when told to, when no call has arrived for
  // idle_ticks, or after life_ticks (100 MHz ticks), and then sets exited[workgroup] =
  // launch_id behind its last store.
  const unsigned long long* mailbox;
  unsigned long long* exited;
  unsigned long long launch_id;
  unsigned long long idle_ticks, life_ticks;
  int poll_waves;            // waves per workgroup that poll the mailbox (1 .. 4)
};
constexpr unsigned long long kResidentStop = ~0ull;
constexpr int kResidentBusyOffset = 64;   // exited[64 + workgroup]: ticks the last call took

// Resident form for ENSEMBLES (resident_ensemble_kernel; launch.hip: ensemble_predict): 2 ..
// kEnsembleMaxWalkers draws per call, host memory to host memory, no launch and no stream
// synchronisation per call.  The grid is 4 * n_slices workgroups of 512 threads that stay on
// the chip (one per CU); workgroup b = 4 * slice + c.  A call runs in three phases that hand
// their data on through device memory (write-through stores, a flag per producer that carries
// the number of the call; nothing is ever reset):
//   A  workgroup w < n_walkers: the occupation of walker w (two quadrature nodes per thread),
//      its G number densities and the two totals -> dens[w], flag_a[w];
//   B  the table's positions are cut into n_slices slices of four quarters, kept in LDS for
//      the life of the launch; lane = walker of a group of 64.  With one group the four
//      workgroups of a slice take a quarter each, with two groups half a slice for a group
//      each, with three or four the whole slice for group c.  Every quarter is summed by itself
//      and the quarters of a slice are added as (q0 + q1) + (q2 + q3) -- by the workgroup or by
//      the reader of its partial sums -- so that a walker's result does not depend on the size
//      of the ensemble.  -> partial[b] (rt, 64), flag_b[b];
//   C  workgroup (row, group): the sum over the slices of one row of one group in fixed
//      order (rows rt, rt + 1: the two totals of the group's walkers) -> 64 contiguous doubles
//      of out (groups, rt + 2, 64) in page-locked memory, then done[group][row].
// The host writes the walkers' parameters (8 doubles each) and then the header (number of the
// call << 10 | walkers) into the mailbox.  On a large-BAR system the mailbox is DEVICE memory
// that the host stores into through the PCIe aperture (tools/micro/bar_write.hip: a host ->
// device -> host round trip 1.8 us against 2.45 us with the mailbox in page-locked memory,
// and polling it costs no PCIe reads): every workgroup polls the header itself.  Otherwise
// the mailbox is page-locked host memory, workgroup 0 alone polls the header and passes it on
// through a word in device memory.  (First version: every workgroup polled a line of its own
// in host memory -- 256 pollers keep the link so busy that a call took 22 - 38 us.)
// Every wait is bounded: idle_ticks / life_ticks
// as for the single draw, call_ticks (or the host's stop) for a wait inside a call; a workgroup
// that leaves sets exited[b] = launch_id << 40 | the call it would have served next, and when
// that is the current one or an earlier one the host serves the call another way.
struct EnsembleArgs {
  int n_theta;
  int n_bins;
  int n_central;
  int n_gauss;
  unsigned flags;
  double split;
  const double* log_m;
  const double* m;
  const double* weight;
  const double* n_h;
  const double* percentile;
  const double* math_table;
  const double* table;          // re-laid-out matrix of the single r tile (as SingleArgs)
  const int32_t* pos_off;
  int64_t n_positions;
  int rt;
  int mode;
  int n_slices;                 // grid = 4 * n_slices
  int per_quarter;              // positions per quarter of a slice
  int dens_stride;              // doubles per walker in dens (n_bins + 2, padded)
  double* dens;                 // device memory
  unsigned long long* flag_a;   // (kEnsembleMaxWalkers)
  double* partial;              // (grid, rt, 64)
  unsigned long long* flag_b;   // (grid)
  unsigned long long* callword; // the header, forwarded by workgroup 0 (direct == 0)
  const unsigned long long* mailbox;   // header (8 words), 8 doubles per walker
  int direct;                   // 1: the mailbox is device memory, every workgroup polls it
  double* out;                  // page-locked: (4, rt + 2, 64)
  unsigned long long* done;     // page-locked: (4, rt + 2) completion words
  unsigned long long* exited;   // page-locked: (grid), then 8 phase stamps of workgroup 0
  unsigned long long epoch;     // first call to serve
  unsigned long long launch_id;
  unsigned long long idle_ticks, life_ticks, call_ticks;
  int lds_area, lds_dens, lds_t, lds_ij;   // byte offsets into the dynamic LDS
  int skip;                     // developer A/B (TC_ENS_SKIP): phases left out, results wrong
};
constexpr int kEnsembleThreads = 512;
constexpr int kEnsembleMaxWalkers = 256;
constexpr int kEnsembleDensPad = 65;      // doubles per bin row of the densities in LDS

struct ContractArgs {
  const double* nbuf;       // (n_bins, ldb)
  int64_t ldb;
  const void* table;        // (n_rtiles, n_positions, RT) re-laid-out matrix
  int64_t n_positions;
  const Chunk* chunks;
  const Group* groups;
  int mode;
  int n_central;
  int r_stride;             // n_rtiles * RT: padded number of r values
  int n_tiles;              // draw tiles (grid.x covers 8 * ceil(n_tiles / 8) * slabs
                            // with xcd_map, else n_tiles * slabs)
  int xcd_map;              // 1: block b runs on XCD b % 8 (whole MI355X, 8 XCDs x 32 CUs)
  int n_slabs;              // groups * table splits per draw tile
  const int32_t* pos_ij;    // float32 kernel: packed bin pairs of every position
  const int32_t* pos_off;   // FP64 matrix kernel: (i, j) * 512 (LDS row bytes) per position
  unsigned long long* trace;  // developer timeline (TC_TRACE): 6 words per block, or NULL
  unsigned long long* wave_trace;  // TC_TRACE: 6 words per wave (progress stamps)
  double* partial;          // (n_groups * k_splits, r_stride, ldb)
  // Interpolator: the block loops over tables [k_begin, k_end) of its k split and
  // accumulates coef[k][draw] * (table k contraction) into the same registers.
  int n_tables;             // 0: single table (fields below unused)
  int k_splits;             // blockIdx.y = group * k_splits + k split
  const double* const* tables;   // (n_tables) re-laid-out matrices
  const double* const* nbufs;    // (n_classes) density buffers
  const int32_t* table_class;    // (n_tables) density class of each table
  const double* coef;       // (n_tables, ldb) spline weight / pair-weight norm
};

// ---- quadratic-form contraction (contract_quad_kernel, mode auto, float64) --------------
constexpr int kQuadMaxU = 5;          // r sub-tiles of 4 per r tile: at most 20 r values
constexpr int kQuadTile = 32;         // draws per tile (two matrix-core column sets of 16)
constexpr int kQuadWavesPerBlock = 4;

// Device copy of a QuadComp: what the kernel needs to walk a component (32 bytes).
struct QuadCompArgs {
  int32_t triangular;
  int32_t i_bin0;
  int32_t j_bin0;
  int32_t n_cb;
  uint32_t unit_base;       // first unit of the component inside an r tile
  int32_t pad[3];
};

constexpr int kQuadTileF32 = 64;       // draws per tile of the float32 kernel (four column sets)
constexpr int kQuadMaxUF32 = 4;        // r sub-tiles per r tile: at most 16 r values

struct QuadArgs {
  const float* nbuf32;             // float32 kernel: (n_bins, ldb) densities in float
  const double* nbuf;              // (n_bins, ldb); interpolator: NULL
  const double* const* nbufs;      // interpolator: density buffer of each class
  const float* const* nbufs32;     // ... float copies (float32 kernel)
  int64_t ldb;
  int n_bins;
  const void* table;               // (n_rtiles, n_units, UP, 64, 2) doubles; interpolator: NULL
  const double* const* tables;     // interpolator: one such matrix per table
  const int32_t* table_class;      // interpolator: density class of each table
  const double* coef;              // interpolator: (n_tables, ldb) spline weight / norm
  uint32_t rtile_bytes;            // bytes of one r tile of a table
  const QuadRun* runs;
  const QuadCompArgs* comps;
  const int32_t* wave_runs;        // (n_waves, 2): first run and end run of every wave
  // (n_waves, 16): the same two numbers, the wave's first QuadRun (8) and the first five
  // words of that run's QuadCompArgs: ONE scalar load at the start of a wave instead of
  // three dependent ones (float64 kernel)
  const int32_t* wave_head;
  int n_waves;
  int priority;                    // wave priority (0..3)
  void* partial;                   // (n_slabs, 4 U, 32) doubles / (n_slabs, 4 U, 64) floats
  const int32_t* merge_range;      // (workgroups, 2): first and end merge of every workgroup
  const int32_t* merges;           // QuadMerge = 4 x int32: slab, first LDS slot, count, -
  unsigned long long* stamps;      // developer timeline: 6 words per wave, or NULL
};

struct FinalizeQuadArgs {
  const void* partial;          // (n_slabs, rt, 32) doubles or (n_slabs, rt, 64) floats
  const int32_t* group_begin;   // (n_groups + 1): slabs of each (draw tile, r tile[, component])
  const double* ngal_part;      // as FinalizeArgs
  int n_ngal_parts;
  int n_rtiles;
  int r_per_tile;               // r values per r tile (the last tile may hold fewer)
  int rt;                       // 4 U: rows of a slab
  int groups_per_rtile;         // 1, or the number of components when separated
  int n_comp;                   // output components
  int n_r;
  int mode;
  int priority;
  int64_t ldb;
  int64_t n_draws;
  double* ngal;
  double* xi;
  // Fused Gaussian likelihood (total prediction, all rows of a draw in one LDS pass, one row
  // block): chi2[b] = (xi_b - data)^T P (xi_b - data) straight from the tile, xi itself is
  // not written.  chi2_data = data (n_r) followed by P (n_r, n_r); NULL: off.
  const double* chi2_data;
  double* chi2;
};

// ---- one launch per batch (predict_fused_kernel): occupation -> quadratic form -> results ---
// A workgroup of kFusedWaves waves owns 64 draws from theta to (ngal, xi[, chi2]): the
// densities live in LDS ((dens_rows, 64) doubles, rows beyond n_bins zero), nothing is handed
// over through memory.  Mode auto, total correlation function, one r tile.
constexpr int kFusedWaves = 8;       // (default; 16 where one workgroup per CU is all that fits)
constexpr int kFusedMaxParts = 8;    // parts of the units per 32-draw tile: waves / 2

struct FusedArgs {
  const double* theta;       // (n_draws, n_theta)
  int n_theta;
  int n_bins;
  int n_central;
  int n_gauss;
  int dens_rows;             // rows of the LDS density array (whole blocks of four)
  // the work of the 4 (8) waves of a 32-draw tile: `count` units of one component from block
  // (rb0, cb0) on.  Total: equal parts of the triangle.  Separated by galaxy type: the cen-cen
  // triangle (one quarter of the waves), the cen-sat rectangle (two quarters), the sat-sat
  // triangle (one quarter).
  int part_rb0[kFusedMaxParts];
  int part_cb0[kFusedMaxParts];
  int part_count[kFusedMaxParts];
  int part_triangular[kFusedMaxParts];    // block columns 0 .. rb of block row rb, else all part_n_cb
  int part_n_cb[kFusedMaxParts];
  int part_i_row0[kFusedMaxParts];        // first density row of the component's rows / columns
  int part_j_row0[kFusedMaxParts];
  int part_unit_base[kFusedMaxParts];     // first unit of the component in the table
  int separate;              // 1: ngal (n_draws, 2), xi (n_draws, 3, n_r): cc, cs, ss
  int n_r;
  int priority;              // wave priorities: phase 2 | phase 1 << 2 | phase 3 << 4; bits 8 - 10:
                             // developer knobs; bit 11: every entry of the matrix is finite
  int64_t n_draws;
  const double* log_m;       // quadrature constants as in OccArgs
  const double* m;
  const double* weight;
  const double* n_h;
  const double* percentile;  // (n_bins) sec_haloprop_percentile (assembly bias)
  double split;
  const double* math_table;
  const void* table;         // re-laid-out matrix of the whole triangle (one r tile)
  uint32_t table_bytes;
  double* ngal;              // (n_draws) or (n_draws, 2)
  double* xi;                // (n_draws, n_r) or (n_draws, 3, n_r); NULL: the likelihood is fused
  const double* chi2_data;   // as FinalizeQuadArgs
  double* chi2;
  // groups of bins with the same quadrature nodes (OccArgs; GROUPED instances only)
  int n_groups;
  int n_central_groups;
  GroupArgs group;
  const double* series;        // moment expansion (OccArgs), 64-draw workgroups only
  const int32_t* series_thr;
  const double* sat_series;
  const int32_t* sat_series_thr;
  int sat_cap;                 // SATDEFER: longest expansion in place = 12 + 4 sat_cap terms
  const double* sat_records;   // SATDEFER: series.h, namespace sat_record, by library bin
  const double* cen_records;   // SATDEFER = 2: series.h, namespace cen_record, per central bin
};

// ---- mode cross, one launch per batch (predict_cross_fused_kernel, kernels.hip.h) ----------
constexpr int kCrossWaves = 8;
constexpr int kCrossChunkBins = 48;  // bins whose mean occupations one LDS buffer holds
constexpr int kCrossMaxRows = 128;   // K (R + 1) of the largest instance (16 rows per wave)
// LDS layout in doubles (launch.hip: run_cross_fused fills the offsets): math table, later the
// spline weights / norms (K, 64) and the results tile (rows out, 65) when they fit there | two
// chunk buffers (kCrossChunkBins, 64), later the row sums (ROWS, 64) | separated by galaxy
// type: the row sums of the centrals (ROWS, 64) | the tile when it does not fit the first part
constexpr int kCrossTableDoubles = 2306;       // = fm::kTableDoubles (fastmath.h)
// predict_cross_small_kernel (up to 16 rows, the sums in every wave's registers): math table |
// stage (waves, 8 rows, 64): the waves' sums of half the rows, later the spline weights and the
// results tile | sums (components, 16, 64)
constexpr int kCrossSmallRows = 16, kCrossSmallChunk = 8;
constexpr int kCrossMaxSplits = 8;
constexpr int cross_small_lds_doubles(int n_comp) {
  return kCrossTableDoubles + kCrossWaves * kCrossSmallChunk * kLanes +
         n_comp * kCrossSmallRows * kLanes;
}
constexpr int cross_buffer_doubles(int rows) {
  return 2 * kCrossChunkBins * kLanes > rows * kLanes ? 2 * kCrossChunkBins * kLanes
                                                      : rows * kLanes;
}

struct CrossFusedArgs {
  const double* theta;       // (n_draws, n_theta)
  int n_theta;
  int64_t n_draws;
  int n_bins;
  int n_groups, n_central_groups;
  GroupArgs group;
  const double* math_table;
  // predict_cross_small_kernel: (n_bins in member order, 16) doubles.  predict_cross_fused_kernel:
  // the A operands of its matrix instructions -- per chunk (the bins padded with zero rows to a
  // multiple of 4), 4-bin step and block of 16 rows 64 doubles, [l] = coefficient of bin
  // 4 step + l / 16, row 16 block + l % 16; chunk c starts at step chunk_block[c]
  const double* rows;
  const int32_t* chunk_block;    // (n_chunks + 1)
  // Deferred (group, draw) pairs of predict_cross_fused_kernel (undecorated, up to 64 rows): 1 =
  // lanes that no shortcut and no expansion serves do not run the node loop in place; their
  // pairs are marked in a bitmap in LDS (one 64-bit word per group, at lds_bitmap doubles) and
  // evaluated after the chunks, 64 pairs per wave at a time, from lists at lds_list (512 words
  // per wave); bin_operand[mi] = where member bin mi's 16-row blocks start in `rows`
  // (doubles: + 64 per block of 16 rows, + row % 16).
  int defer;
  int lds_bitmap, lds_list;
  const int32_t* bin_operand;    // (n_bins)
  // chunks of whole groups with at most kCrossChunkBins members, none across the boundary
  // between centrals and satellites: chunk c = groups chunk_group[c] .. chunk_group[c + 1]
  const int32_t* chunk_group;    // (n_chunks + 1)
  int n_chunks, n_central_chunks;
  int n_tables;              // K
  int n_r;
  int separate;              // 1: ngal (n_draws, 2), xi (n_draws, 2, n_r): centrals, satellites
  int row_stride;            // doubles between the rows of consecutive members (= ROWS)
  int cen_waves;             // predict_cross_small_kernel, separate: waves [0, cen_waves) take
                             // the groups of centrals
  // predict_cross_small_kernel, medium batches: n_splits workgroups per tile of 64 draws, each
  // with a share of the groups (split_*[s] .. split_*[s + 1]: all groups / the centrals' / the
  // satellites'); partial (tiles, n_splits, n_comp, 16, 64) doubles, counters (tiles) ints, zero
  // between launches
  int n_splits;
  int split_all[kCrossMaxSplits + 1], split_cen[kCrossMaxSplits + 1],
      split_sat[kCrossMaxSplits + 1];
  double* partial;
  int* counters;
  int lds_res0;              // offsets (doubles) into the dynamic LDS: sums of the centrals,
  int lds_tile;              // spline weights + results tile
  int interp;                // 1: spline weights from x (else one table, c = 1)
  int n_dim;
  int n_axis[kMaxInterpDim];
  int axis_offset[kMaxInterpDim];
  int a_offset[kMaxInterpDim];
  const double* xp;
  const double* a;
  const int32_t* table_node;   // (K, n_dim)
  const double* x;             // (n_draws, n_dim)
  double split;
  int priority;
  double* ngal;
  double* xi;                // NULL: the likelihood is fused
  const double* chi2_data;   // as FinalizeQuadArgs
  double* chi2;
};

struct FinalizeArgs {
  const double* partial;   // (n_groups, r_stride, ldb)
  const Group* groups;     // component of each group
  const double* ngal_part; // (n_ngal_parts, 2, ldb); NULL: the partials are already
                           // normalised and ngal has been written (interpolator)
  int n_ngal_parts;
  int n_groups;            // partial slabs = n_groups * k_splits
  int k_splits;
  int n_comp;              // 1: sum all components; else per component
  int r_stride;
  int n_r;
  int mode;
  int64_t ldb;
  int64_t n_draws;
  double* ngal;            // (n_draws) or (n_draws, 2)
  double* xi;              // (n_draws, n_comp, n_r)
};

struct InterpArgs {
  int n_dim;
  int n_tables;
  int n_classes;
  int mode;
  int separate;             // write ngal as (n_draws, 2)
  int n_axis[kMaxInterpDim];
  int axis_offset[kMaxInterpDim];   // offset of xp_d in xp
  int a_offset[kMaxInterpDim];      // offset of a_d in a
  const double* xp;         // concatenated abscissae
  const double* a;          // concatenated (n_d - 1, 4, n_d) spline matrices
  const int32_t* table_node;     // (n_tables, n_dim) grid index of each table
  const int32_t* table_class;    // (n_tables)
  const double* x;          // (n_draws, n_dim)
  const double* const* ngal_parts;  // per class: (n_parts, 2, ldb)
  int n_ngal_parts;
  int64_t ldb;
  int64_t n_draws;
  double* coef;             // (n_tables, ldb)
  double* ngal;             // (n_draws) or (n_draws, 2)
};

}  // namespace tc
