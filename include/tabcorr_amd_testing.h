/*
 * tabcorr_amd_testing.h -- test-infrastructure and developer hooks of libtabcorr_hip.so.
 *
 * NOT part of the drop-in boundary (include/tabcorr_amd.h): nothing here is called by the
 * product's host classes.  The CPU tests (tests/test_host_cpu.py) use the tc_debug_* helpers
 * to check host-side planning code and the table-driven math without a GPU; the timeline
 * readers return data only from developer builds (-DTC_DEVELOPER_KNOBS, tools/build_dev.sh).
 */
#ifndef TABCORR_AMD_TESTING_H
#define TABCORR_AMD_TESTING_H

#include "tabcorr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host evaluation of the table-driven FP64 functions the occupation kernel uses in place
 * of the device libm (tabcorr_amd/csrc/fastmath.h): kind 0 erf, 1 log2 (x > 0 normal),
 * 2 exp2, 3 exp10, 4 erf and 5 its derivative 2/sqrt(pi) exp(-x^2) from erf_gauss_fast. */
int tc_debug_fastmath(int kind, int64_t n, const double* x, double* y);

/* The moment expansion of a central bin's node sum (tabcorr_amd/csrc/series.h) on the host,
 * next to the node loop it replaces: for one bin [log_min, log_max] with
 * prim_haloprop_dist_index `dist_index` and n_gauss nodes, and n draws (log_m_min, sigma):
 * series[i] = the expansion with the number of terms the kernel would take for a wave whose
 * smallest |sigma| is that draw's (terms[i]; 0: the expansion does not apply, series[i] is
 * then the node loop's value), nodes[i] = sum_k W_k erf_fast((log M_k - log_m_min) / sigma). */
int tc_debug_central_series(int n_gauss, double log_min, double log_max, double dist_index,
                            int64_t n, const double* log_m_min, const double* sigma,
                            double* series, double* nodes, int32_t* terms);

/* The same for a satellite bin (series.h, namespace sat): the binomial expansion of
 * sum_k W_k ((M_k - M0) / M1)^alpha around the bin's reference mass next to the node loop;
 * terms[i] from that draw's M0 (0: the expansion does not apply -- the bin is too close to M0 or
 * alpha outside [0, 4] --, series[i] is then the node loop's value). */
int tc_debug_satellite_series(int n_gauss, double log_min, double log_max, double dist_index,
                              int64_t n, const double* log_m0, const double* log_m1,
                              const double* alpha, double* series, double* nodes,
                              int32_t* terms);

/* Work decomposition used by the contraction kernel for a table with n_bins rows of which
 * the first n_central (after the library's stable sort by gal_type) are centrals, cut into
 * n_chunks wave-sized pieces.  Outputs one record per packed column, in processing order:
 * entry_pair[e] = packed column p (auto) or bin (cross), entry_chunk[e], entry_class[e]
 * (0 cen-cen / cen, 1 cen-sat, 2 sat-sat / sat).  Lets CPU tests check that the kernel's
 * traversal covers every column exactly once and only gathers density rows its workgroup
 * stages (padding positions included). */
int tc_plan_debug(int mode, int n_bins, const uint8_t* is_central, int n_chunks,
                  int64_t* n_entries, int32_t* entry_pair, int32_t* entry_chunk,
                  int32_t* entry_class);

/* Quadratic-form contraction kernel (mode auto, float64; tabcorr_amd/csrc/hostmath.h):
 * builds the schedule for a table of n_bins bins (the first n_central centrals) and checks
 * that it covers every (draw tile, r tile, component, table, unit) exactly once with
 * consecutive slabs per output group; returns its size and the smallest / largest number
 * of units any wave gets. */
int tc_debug_quad_schedule(int n_bins, int n_central, int by_type, int n_tiles, int n_rtiles,
                           int n_tables, int separate, int max_waves, int min_units,
                           int order /* 0 draw-tile-major, 1 table-major, 2 r-tile-major */,
                           int* n_waves, int* n_runs, int* n_slabs, int64_t* units_min,
                           int64_t* units_max);
/* The equal contiguous parts of a triangle of n_rb block rows that the waves of
 * predict_fused_kernel walk (hostmath.h: triangle_parts): first block row, block column and
 * number of units of each of the n_parts parts. */
int tc_debug_triangle_parts(int n_rb, int n_parts, int32_t* rb0, int32_t* cb0, int32_t* count);
/* Groups of bins that share their Gauss-Legendre nodes (tabcorr_amd/csrc/hostmath.h:
 * find_node_groups; identical log_prim_haloprop_min / max and galaxy type), for n_bins bins in
 * library order of which the first n_central are centrals: group i = member[begin[i] ..
 * begin[i + 1]).  begin holds n_groups + 1 entries (capacity n_bins + 1), member n_bins. */
int tc_debug_node_groups(int n_bins, int n_central, const double* log_min,
                         const double* log_max, int32_t* begin, int32_t* member,
                         int* n_groups, int* n_central_groups);
/* TEST INFRASTRUCTURE, never called by the product: executes the kernel's table layout,
 * schedule and slab grouping on the host, lane by lane, for densities (n_bins, ldb) given
 * in the reference's bin order; out (n_draws, 1 | 3, n_r) = sum_p c_p T[r][p] n_i n_j
 * before the normalisation (tabcorr.py:641-655).  Lets CPU tests check the index logic of
 * contract_quad_kernel / finalize_quad_kernel without a GPU. */
int tc_debug_quad_emulate(int n_bins, int n_r, const double* tpcf_matrix,
                          const uint8_t* is_central, int by_type, int separate,
                          const double* densities, int64_t ldb, int64_t n_draws,
                          int max_waves, int min_units, int order, double* out);

/* Developer timeline (environment TC_TRACE=1): per workgroup of the last contraction
 * launch six words: 100 MHz timestamps at start / after staging / after the main loop /
 * at the end, HW_ID and XCC_ID.  Copies min(capacity, n_blocks) records. */
int tc_debug_trace(tc_table* table, uint64_t* out, int64_t capacity, int64_t* n_blocks);
/* Same run, per wavefront: timestamps at the start of the main loop, after 1/4, 1/2 and
 * 3/4 of its blocks and at the end, and HW_ID (wave slot / SIMD / CU). */
int tc_debug_wave_trace(tc_table* table, uint64_t* out, int64_t capacity,
                        int64_t* n_waves);
/* Resident un-batched path (option "resident"): per workgroup the 100 MHz ticks the last call
 * took from the sight of its parameters to the store of its completion word. */
int tc_debug_resident_ticks(tc_table* table, uint64_t* out, int64_t capacity, int64_t* n_blocks);
/* Resident ensemble kernel: eight words of workgroup 0 in the last call (developer builds;
 * 100 MHz stamps: call seen, occupation stored, the group's occupations seen, densities in LDS,
 * quarters summed, partial sums stored; [6] shader cycles of the quarter sums; [7] finished),
 * then three host times of that call in
 * ns (published, every row combined -- from its begin --, time spent on the rows): 11 words. */
int tc_debug_ensemble_stamps(tc_table* table, uint64_t* out);
#ifdef __cplusplus
}
#endif

#endif /* TABCORR_AMD_TESTING_H */
