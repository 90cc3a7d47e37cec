/*
 * tabcorr_amd.h -- C ABI of libtabcorr_hip.so, the MI355X (gfx950) implementation
 * of TabCorr's predict() path.
 *
 * The reference (johannesulf/TabCorr v1.2.0) is pure Python and has no FFI layer;
 * the seam this library sits behind is the Python class surface.  Each entry point
 * below names the reference code it replaces (paths relative to the reference
 * repository root).  The library is loaded with ctypes by tabcorr_amd/_lib.py;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C symbols, opaque handles, int status return (0 = TC_OK), no exceptions
 *     cross the boundary; tc_last_error() returns a thread-local message.
 *   - all host arrays are caller-owned, C-contiguous, little-endian; outputs are
 *     caller-allocated.  The library never frees caller memory.
 *   - functions with the suffix _device take DEVICE pointers, enqueue work on the
 *     handle's stream and return without synchronising (tc_*_synchronize waits).
 *   - a handle is bound to the HIP device that was current when it was created and
 *     must be used by one host thread at a time.  An interpolator handle works THROUGH the
 *     table handles it was created from (their caches, lanes, timers and fused-likelihood
 *     state): a call on the interpolator must not run concurrently with a call on any of
 *     its tables either (the Python classes take the tables' locks with the
 *     interpolator's).
 *   - there is no CPU fallback: every compute entry point fails with TC_ERR_HIP when
 *     no gfx950 device is usable.
 */
#ifndef TABCORR_AMD_H
#define TABCORR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_OK 0
#define TC_ERR_INVALID 1     /* bad argument (Python side raises ValueError) */
#define TC_ERR_HIP 2         /* HIP runtime / device failure (RuntimeError) */
#define TC_ERR_UNSUPPORTED 3 /* valid request this build cannot serve */
#define TC_ERR_RCCL 4        /* RCCL failure */

/* tabcorr/tabcorr.py:625,648 -- attrs['mode'] */
#define TC_MODE_AUTO 0
#define TC_MODE_CROSS 1

/* element type of a correlation matrix handed to tc_table_create */
#define TC_DTYPE_F64 0
#define TC_DTYPE_F32 1

/* flags of the predict entry points */
#define TC_FLAG_SEPARATE_GAL_TYPE 1u  /* predict(..., separate_gal_type=True), tabcorr.py:652-683 */
#define TC_FLAG_MODULATE_WITH_CENOCC 2u /* <N_sat> *= <N_cen> (halotools Zheng07Sats option) */
#define TC_FLAG_ASSEMBIAS 4u          /* theta carries 2 extra Heaviside assembly-bias strengths */
#define TC_FLAG_LEGACY_NO_DIST_INDEX 8u /* table without prim_haloprop_dist_index, tabcorr.py:571-574 */
/* Occupation family of the theta columns (default: Zheng et al. 2007).  With
 * TC_FLAG_LEAUTHAUD11 every entry point named *_zheng07_* evaluates the Leauthaud et al.
 * (2011) centrals / satellites on the Behroozi et al. (2010) stellar-to-halo mass relation
 * instead; theta then has 14 columns: logm0, logm1, beta, delta, gamma (the relation at the
 * model's redshift), scatter, alphasat, bsat, betasat, bcut, betacut, threshold (log10 of
 * the stellar mass threshold), the Hubble parameter inside the relation (halotools'
 * Behroozi10SmHm: 0.7) and that of the satellite terms (halotools' Leauthaud11Sats: 0.72).  TC_FLAG_MODULATE_WITH_CENOCC applies as for
 * Zheng07; TC_FLAG_ASSEMBIAS is not available for this family. */
#define TC_FLAG_LEAUTHAUD11 16u

typedef struct tc_table tc_table;
typedef struct tc_interp tc_interp;
typedef struct tc_comm tc_comm;

/* ---- runtime ---------------------------------------------------------------------- */

const char* tc_last_error(void);
/* The copies that end a synchronous host call (results from the page-locked staging area into
 * the caller's arrays) are shared with `helpers` threads of the library (default 3, 0 .. 7; 0:
 * the calling thread copies alone), which poll for the next job for `spin_us` microseconds
 * (default 100, 0 .. 100000) before they sleep: set both to 0 where every core runs a rank of
 * its own.  Process-wide; the threads are started by the first large copy, and a forked child
 * starts its own. */
int tc_set_copy_threads(int helpers, int spin_us);
int tc_device_count(int* count);
int tc_set_device(int device);
int tc_get_device(int* device);
/* HIP runtime version the library is bound to (guards against a second HIP runtime in
 * the process) and a short device description. */
int tc_runtime_version(int* version);
int tc_device_name(char* buffer, size_t size);
int tc_device_synchronize(void);

/* Device memory for callers that keep draws / results resident (bench.py, multi-GPU). */
int tc_device_malloc(void** ptr, size_t bytes);
int tc_device_free(void* ptr);
int tc_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
int tc_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);

/* Page-locked host memory for the asynchronous entry points (tc_*_async): caller-owned,
 * released with tc_host_free.  tc_host_register pins memory the caller already owns (e.g.
 * the coordinate array of an ensemble sampler) until tc_host_unregister.  The library keeps
 * a registry of these ranges; tc_host_is_pinned reports whether [ptr, ptr + bytes) lies
 * inside one (no HIP call, usable without a device). */
int tc_host_alloc(void** ptr, size_t bytes);
int tc_host_free(void* ptr);
int tc_host_register(void* ptr, size_t bytes);
int tc_host_unregister(void* ptr);
int tc_host_is_pinned(const void* ptr, size_t bytes, int* pinned);

/* ---- helpers exported for the host-side tests (pure host code, no GPU needed) ------- */

/* Gauss-Legendre nodes mapped to (0, 1) and weights, as tabcorr.py:543-546
 * (np.polynomial.legendre.leggauss(n); x = (x + 1) / 2). */
int tc_gauss_legendre(int n, double* x, double* w);

/* Packed-pair index map of tabcorr.py:770-806 / 626-639: for every packed column p the
 * row index i1, column index i2 (i2 <= i1) and prefactor (1 on the diagonal, else 2). */
int tc_pair_indices(int n_bins, int32_t* index_1, int32_t* index_2, int32_t* prefactor);

/* Not-a-knot cubic spline matrix a[(n-1)][4][n] of interpolator.py:219-272. */
int tc_spline_interpolation_matrix(int n, const double* xp, double* a);

/* ---- one tabulated table (replaces the state of a `TabCorr` instance) ----------------
 *
 * tc_table_create uploads everything predict() needs from a TabCorr object
 * (tabcorr.py:356-368): the correlation matrix `tpcf_matrix` (R x P, row-major, P =
 * G (G + 1) / 2 packed lower triangle in mode auto, P = G in mode cross) and the
 * gal_type columns read at tabcorr.py:537-541, 568-570, 623.  It precomputes what the
 * reference caches lazily on `self` (pair indices, tabcorr.py:626-639; Gauss-Legendre
 * nodes, :543-546).  compute_dtype selects the arithmetic of the contraction: F64
 * (default, parity <= 1e-10) or F32 (table and accumulation in float, for BASELINE
 * configs[4]; stated tolerance 1e-5).  Occupations are always evaluated in F64.
 */
int tc_table_create(int mode, int n_bins, int n_r, int64_t n_pairs,
                    const void* tpcf_matrix, int matrix_dtype,
                    const double* n_h,
                    const double* log_prim_haloprop_min,
                    const double* log_prim_haloprop_max,
                    const double* sec_haloprop_percentile,
                    const double* prim_haloprop_dist_index, /* NULL: legacy table */
                    const uint8_t* is_central,              /* 1 = 'centrals' row */
                    int compute_dtype,
                    tc_table** table);
int tc_table_destroy(tc_table* table);
int tc_table_synchronize(tc_table* table);

/* Shape queries: n_bins, n_r, n_pairs, number of xi components with
 * TC_FLAG_SEPARATE_GAL_TYPE (3 auto / 2 cross), table bytes resident in HBM. */
int tc_table_info(const tc_table* table, int* mode, int* n_bins, int* n_r,
                  int64_t* n_pairs, int* n_components, int64_t* device_bytes);

/* TabCorr.mean_occupation(model) for a batch of Zheng07 parameter vectors
 * (tabcorr.py:465-578 with the halotools callbacks of :556-563 evaluated on device).
 * theta: (n_draws, n_theta) with columns logMmin, sigma_logM, logM0, logM1, alpha
 * [, A_cen, A_sat when TC_FLAG_ASSEMBIAS]; occupation out: (n_draws, n_bins). */
int tc_mean_occupation_zheng07_batch(tc_table* table, const double* theta, int n_theta,
                                     int64_t n_draws, int n_gauss_prim, unsigned flags,
                                     double* occupation);

/* TabCorr.predict(model) for a batch of Zheng07 draws (tabcorr.py:580-683).
 * Outputs, without TC_FLAG_SEPARATE_GAL_TYPE: ngal (n_draws), xi (n_draws, n_r).
 * With it: ngal (n_draws, 2) [centrals, satellites]; xi (n_draws, n_components, n_r) in
 * the reference's dict order (centrals-centrals, centrals-satellites,
 * satellites-satellites | centrals, satellites). */
int tc_predict_zheng07_batch(tc_table* table, const double* theta, int n_theta,
                             int64_t n_draws, int n_gauss_prim, unsigned flags,
                             double* ngal, double* xi);
int tc_predict_zheng07_batch_device(tc_table* table, const double* theta_device,
                                    int n_theta, int64_t n_draws, int n_gauss_prim,
                                    unsigned flags, double* ngal_device,
                                    double* xi_device);

/* Gaussian likelihood fused behind predict() (the step every
 * likelihood built on the reference performs on the host, README.md:7): for each draw
 * chi2 = (xi - data)^T precision (xi - data) over the n_r correlation function bins of
 * the total prediction.  Outputs ngal (n_draws) and chi2 (n_draws): one pair of doubles
 * per draw crosses PCIe instead of 1 + n_r. */
int tc_chi2_zheng07_batch(tc_table* table, const double* theta, int n_theta,
                          int64_t n_draws, int n_gauss_prim, unsigned flags,
                          const double* data, const double* precision, double* ngal,
                          double* chi2);

/* The same with the draws and the results (ngal, chi2: n_draws doubles each) resident on
 * the device; data / precision are host arrays (uploaded when they change).  Enqueues and
 * returns; at most 2^18 draws per call.  Lets a multi-GPU run gather 16 bytes per draw
 * instead of 8 (1 + n_r). */
int tc_chi2_zheng07_batch_device(tc_table* table, const double* theta_device, int n_theta,
                                 int64_t n_draws, int n_gauss_prim, unsigned flags,
                                 const double* data, const double* precision,
                                 double* ngal_device, double* chi2_device);

/* A handful of independent draws in ONE launch (the proposals of an ensemble sampler's step,
 * or the reference's un-batched predict(), README.md:72-75, for n_walkers = 1): every
 * workgroup evaluates the occupations of its draw itself, contracts its share of the table
 * and stores its partial sums and a completion word into page-locked host memory, which the
 * host polls -- no device-side combination, no stream synchronisation.  theta (n_walkers,
 * n_theta), ngal (n_walkers), xi (n_walkers, n_r): host arrays.  Total prediction only;
 * calls the path cannot serve (TC_FLAG_SEPARATE_GAL_TYPE, more than 64 walkers, several r
 * tiles, float32 tables, TC_FLAG_LEAUTHAUD11) are forwarded to tc_predict_zheng07_batch,
 * which in turn routes small eligible batches here. */
int tc_predict_zheng07_many(tc_table* table, const double* theta, int n_theta, int n_walkers,
                            int n_gauss_prim, unsigned flags, double* ngal, double* xi);
/* ONE draw against several tables in one call -- the reference's documented likelihood step
 * evaluates two: halotab_wp.predict(model), then halotab_ds.predict(model)
 * (docs/guides/overview.rst:86-92).  Every table's call is posted before the first answer is
 * waited for (its resident kernel's mailbox, or one launch on the table's own stream), so the
 * tables' round trips overlap; each table is served by the path tc_predict_zheng07_many would
 * take for it, with the same bits.  tables: 1 .. 16 different handles (any mix of modes and
 * shapes; all must accept the same theta and flags); ngal (n_tables); xi[k]: n_r(k) doubles. */
int tc_predict_zheng07_joint(tc_table* const* tables, int n_tables, const double* theta,
                             int n_theta, int n_gauss_prim, unsigned flags, double* ngal,
                             double* const* xi);

/* Asynchronous host-to-host form of the two calls above -- the SURVEY.md section 8d metric
 * (theta on the host -> (ngal, xi) on the host) at the device rate.  What it replaces in the
 * reference is the user's loop of predict() calls (README.md:72-75): an ensemble sampler
 * enqueues the batch of walker positions of step k + 1 while the results of step k are still
 * on their way back.
 *
 * theta, ngal, xi (chi2) must lie in page-locked memory (tc_host_alloc / tc_host_register),
 * else TC_ERR_INVALID.  The call enqueues upload -> occupation -> contraction ->
 * finalisation -> download on the next lane of the handle (stream + workspaces + staging,
 * rotating as the _device entry points do) and returns a ticket without synchronising;
 * copies run on the lane's stream, so the kernels of call k + 1 overlap the download of call
 * k.  The caller must not touch theta / ngal / xi of a ticket before tc_table_wait(ticket)
 * returned (tc_table_query: non-blocking test).  Tickets complete in any order; results of
 * a ticket land only in the buffers passed with it.  At most 64 tickets may be pending. */
int tc_predict_zheng07_batch_async(tc_table* table, const double* theta_pinned, int n_theta,
                                   int64_t n_draws, int n_gauss_prim, unsigned flags,
                                   double* ngal_pinned, double* xi_pinned, int64_t* ticket);
int tc_chi2_zheng07_batch_async(tc_table* table, const double* theta_pinned, int n_theta,
                                int64_t n_draws, int n_gauss_prim, unsigned flags,
                                const double* data, const double* precision,
                                double* ngal_pinned, double* chi2_pinned, int64_t* ticket);
int tc_table_wait(tc_table* table, int64_t ticket);
int tc_table_query(tc_table* table, int64_t ticket, int* done);

/* TabCorr.predict(ndarray): the operator seam of tabcorr.py:616-621 for arbitrary
 * occupation models evaluated by the caller.  occupation: (n_draws, n_bins). */
int tc_predict_occupation_batch(tc_table* table, const double* occupation,
                                int64_t n_draws, unsigned flags, double* ngal,
                                double* xi);

/* ---- interpolation over a grid of tables (replaces `Interpolator`) -------------------
 *
 * tables: K handles in tabcorr_list order.  points: (K, n_dim) extra-parameter values
 * of each table (the rows of param_dict_table, interpolator.py:14-61).  The library
 * validates the grid (TC_ERR_INVALID if it is not one, interpolator.py:45-57), builds
 * the spline matrices (interpolator.py:219-272) and shares occupations between tables
 * whose gal_type tables are identical (interpolator.py:63-70).
 */
int tc_interp_create(tc_table* const* tables, int n_tables, int n_dim,
                     const double* points, tc_interp** interp);
int tc_interp_destroy(tc_interp* interp);
int tc_interp_synchronize(tc_interp* interp);
/* abscissae of dimension d (sorted unique values): returns n and copies min(n, size). */
int tc_interp_axis(const tc_interp* interp, int dim, int* n, double* xp, int size);

/* Interpolator.predict(model) for a batch (interpolator.py:124-216): theta as above,
 * x: (n_draws, n_dim) values of the extra parameters (model.param_dict[key]).
 * Out-of-range x is clamped to the outermost spline segment (extrapolate=True,
 * interpolator.py:327-328); the extrapolate=False ValueError is raised by the host
 * class before the call.  Outputs as tc_predict_zheng07_batch. */
int tc_interp_predict_zheng07_batch(tc_interp* interp, const double* theta, int n_theta,
                                    const double* x, int64_t n_draws, int n_gauss_prim,
                                    unsigned flags, double* ngal, double* xi);
int tc_interp_predict_zheng07_batch_device(tc_interp* interp, const double* theta_device,
                                           int n_theta, const double* x_device,
                                           int64_t n_draws, int n_gauss_prim,
                                           unsigned flags, double* ngal_device,
                                           double* xi_device);

/* The fused Gaussian likelihood (tc_chi2_zheng07_batch) behind Interpolator.predict: what an
 * MCMC over interpolated tables (cosmology / phase-space parameters next to the HOD) needs per
 * draw.  Host arrays in, ngal (n_draws) and chi2 (n_draws) out; the `_device` form keeps the
 * draws, x and the results on the device (data / precision are host arrays, uploaded when
 * they change), enqueues and returns. */
int tc_interp_chi2_zheng07_batch(tc_interp* interp, const double* theta, int n_theta,
                                 const double* x, int64_t n_draws, int n_gauss_prim,
                                 unsigned flags, const double* data, const double* precision,
                                 double* ngal, double* chi2);
int tc_interp_chi2_zheng07_batch_device(tc_interp* interp, const double* theta_device,
                                        int n_theta, const double* x_device, int64_t n_draws,
                                        int n_gauss_prim, unsigned flags, const double* data,
                                        const double* precision, double* ngal_device,
                                        double* chi2_device);

/* Asynchronous host-to-host forms (page-locked theta, x, outputs; see
 * tc_predict_zheng07_batch_async): upload, kernels and download of a call on one of the
 * interpolator's two lanes (stream + workspaces), consecutive calls alternating between them. */
int tc_interp_predict_zheng07_batch_async(tc_interp* interp, const double* theta_pinned,
                                          int n_theta, const double* x_pinned, int64_t n_draws,
                                          int n_gauss_prim, unsigned flags, double* ngal_pinned,
                                          double* xi_pinned, int64_t* ticket);
int tc_interp_chi2_zheng07_batch_async(tc_interp* interp, const double* theta_pinned,
                                       int n_theta, const double* x_pinned, int64_t n_draws,
                                       int n_gauss_prim, unsigned flags, const double* data,
                                       const double* precision, double* ngal_pinned,
                                       double* chi2_pinned, int64_t* ticket);
int tc_interp_wait(tc_interp* interp, int64_t ticket);
int tc_interp_query(tc_interp* interp, int64_t ticket, int* done);

/* Run-time options of a table handle (the library never reads the environment):
 *   "pipeline"    1 (default): consecutive device-pointer calls rotate over the handle's
 *                 lanes (stream + workspaces) so that kernels of neighbouring batches
 *                 overlap; 0: every call on lane 0, kernels strictly serialised.
 *   "lanes"       number of lanes, 1..8 (default 4; more lose 20 %: four hardware queues).
 *   "ordered"     0 (default): device-pointer calls complete in any order -- wait with
 *                 tc_table_synchronize (or gather with tc_comm_gather, which waits for every
 *                 lane); 1: their finalisations are chained so that results appear in call
 *                 order (0.3 - 3 us per 10^4-draw step).
 *   "fused"       1 (default): pipelined device-pointer and asynchronous calls that qualify
 *                 (mode auto, at most 20 r values; 104 bins, or 208 for the Zheng07 family with
 *                 n_gauss_prim = 10; total or separated by galaxy type) run as ONE launch per
 *                 batch, a workgroup carrying 64 (32) draws from the parameters to the results,
 *                 for batches of
 *                 "fused_min_draws" .. "fused_max_draws" draws (default 0 = chosen per table:
 *                 512 for small tables, ~7000 for 100 bins x 19 r values; 30720; asynchronous
 *                 calls: no upper bound); 0: always occupation, contraction, finalisation
 *                 kernels; 2: one launch also for calls that run alone on their lane and for
 *                 tables of 105 .. 248 bins (one 16-wave workgroup per CU: level with the
 *                 three kernels).
 *   "fused_waves" 0 (default): 8 waves per workgroup where two workgroups fit a CU; 8 / 16:
 *                 that many where the table fits.
 *   "fused_draws" 0 (default): workgroups of 32 draws (one tile, eight waves, two per CU) for
 *                 batches below 8192 draws of the Zheng07 family with n_gauss_prim = 10 --
 *                 a third faster there, and the one-launch form then pays from 12 draws per
 *                 bin on -- and for tables of 105 .. 208 bins, of 64 draws otherwise; 32 / 64:
 *                 forced.
 *   "sync_chunks" synchronous host-array calls (tc_predict_zheng07_batch, tc_chi2_zheng07_batch
 *                 beyond the zero-copy size): 0 (default) batches of 2048 draws and more are cut
 *                 into 2 .. 8 chunks of draws (about a megabyte of results each) whose staging,
 *                 kernels, transfers and copies into the caller's arrays overlap; N >= 1: N
 *                 chunks for every batch; -1: the serial path (upload, the kernels alone on one
 *                 lane, download).  "sync_form" (default 32): draws per workgroup of the
 *                 one-launch form the chunks take where it serves the table -- a draw's result
 *                 then does not depend on the number of chunks; 0: whatever a pipelined call of
 *                 the chunk's size takes.  "sync_direct_out": as "async_direct_out" for the
 *                 chunks' staging area (2, default: arrays up to 1 MB are stored by the kernels
 *                 themselves, larger ones travel by copy command).
 *   "sync_stagger"  synchronous host calls on float32 tables with 16 MB of results and more at
 *                 2 KB per draw and more (a (19, 40) table's 10^4 draws are 61 MB): the chunks' kernels run one after the other -- each
 *                 with the chip to itself, the previous chunk's results travelling and being
 *                 copied meanwhile -- and the chunks shrink geometrically to this many per cent
 *                 of the first (default 10, six chunks), so that only a small chunk's transfer
 *                 is left behind the last kernel: 3.1e6 -> 3.6e6 calls/s into arrays the caller
 *                 keeps.  0: equal chunks on all lanes, as for small results.
 *   "autotune"    value = the predict flags to tune for (a combination of TC_FLAG_*; 0 = the
 *                 total prediction of plain Zheng07), n_gauss_prim = 10: MEASURES on this table,
 *                 for batch sizes 256 .. 65536 (x 2 steps), which form serves a batch fastest in
 *                 the pipelined regime -- three kernels, one launch with 64-draw or with 32-draw
 *                 workgroups -- and keeps the choice in the handle (~0.5 s; draws from a wide
 *                 prior box on scratch buffers).  Pipelined device-pointer and asynchronous
 *                 calls with these flags then take the measured form of the nearest batch size
 *                 instead of the built-in estimate (fitted on a handful of table shapes).  -1:
 *                 forget every measurement.  tc_table_autotune_result reads it back.
 *   "autotune_after"  default 0 = never (round 6; it was 256 in round 5).  N > 0: the N-th
 *                 pipelined device-pointer or asynchronous call with one combination of predict
 *                 flags (tables the one-launch forms can serve) runs "autotune" for that
 *                 combination by itself -- that call BLOCKS for about half a second, once, and
 *                 every lane is synchronised -- so that a loop of calls gets the measured form
 *                 without asking; the built-in estimate serves the calls before it.  The
 *                 measured choice depends on wall-clock timings: the last bits of a draw may
 *                 differ before and after that call and from run to run, which is why it is
 *                 opt-in (and refused while "deterministic" is set).
 *   "series"      bit mask; default -1 = by the table: bit 0 for tables whose widest central
 *                 bin is at most 0.1 dex (with wider bins nearly every wavefront of a wide prior
 *                 holds a draw whose sigma_logM is below a bin width and runs the expansion AND
 *                 the node loop), bit 1 off.  Bit 0: the Gauss-Legendre sum of an undecorated
 *                 central bin (tabcorr/tabcorr.py:556-578) by its moment expansion around the
 *                 bin centre -- the same sum re-ordered, one erf and 8 .. 24 short terms instead
 *                 of n_gauss_prim erf evaluations, tabcorr_amd/csrc/series.h -- for every draw
 *                 whose sigma_logM is above about a bin width (truncation below 1e-16;
 *                 otherwise, and for decorated centrals, the node loop); bit 1: the binomial
 *                 expansion of a satellite bin well above the draw's M0.  0: always the node
 *                 loops.  Each draw takes the terms IT needs: with any value of this option a
 *                 draw's result is bit for bit the same wherever it sits in whatever batch
 *                 (round 4 chose the term count per wavefront, which was faster on wide priors
 *                 and made the last bits depend on the neighbouring draws).  Ensembles of
 *                 similar draws (an MCMC's walkers) gain from 3; on a wide prior box a
 *                 wavefront runs the satellites' expansion AND their node loop for most bins.
 *   "grouped"     1 (default): bins with identical log_prim_haloprop_min / max and galaxy type
 *                 -- the secondary-percentile bins of one mass bin (tabcorr/tabcorr.py:186-205)
 *                 -- share their Gauss-Legendre nodes (:548-549); the occupation functions are
 *                 evaluated once per node of such a GROUP and every member bin accumulates
 *                 them with its own weights (Zheng07 family, n_gauss_prim = 10).  No effect on
 *                 tables without such bins.  0: every bin by itself (same occupations per bin
 *                 to the last bit; sums over bins may differ in the last bits).
 *   "fused_defer" 2 (default) / 1: pipelined / asynchronous batches of undecorated Zheng07
 *                 predictions with n_gauss_prim = 10 that take the one-launch form with 64-draw
 *                 workgroups evaluate a satellite bin by its binomial expansion for the draws
 *                 the bin's SHORTEST expansion serves (csrc/series.h; the same number of terms
 *                 for all of them) and every other (bin, draw) pair after the wave's bins, 64
 *                 pairs per pass of the node loop -- tables whose satellite bins all have an
 *                 expansion (bins up to ~0.2 dex wide), unless "series" is 0.  BASELINE
 *                 configs[1] 39.2 -> 37.5 us per 10^4 draws, the reference's example table 18.2 ->
 *                 17.5.  What a draw defers depends on the draw alone: same bits wherever it
 *                 sits in whatever batch.  2: where the centrals take their expansion
 *                 (option "series"), a central bin comes from one record as well and the draws
 *                 no expansion serves (sigma_logM below about a tenth of a bin width, parameters
 *                 to fix up) join the deferred pairs: the node loops of BOTH galaxy types leave
 *                 the loop of bins (configs[1] 37.8 -> 37.2 us).  0: the node loops in place.
 *   "cross_wide_min_draws"  default 4096.  Mode cross (tpcf_matrix with one column per halo
 *                 bin), tables or interpolators of up to 16 result rows whose node groups have
 *                 at most two members: undecorated batches of this many draws take the one-launch
 *                 form that multiplies on the matrix pipe and reads a group's expansion
 *                 constants from one record (csrc/series.h, namespace record) -- 49.1 against
 *                 59.1 us per 10^4 draws of the reference's AbacusSummit table --, smaller ones
 *                 the form that keeps the row sums in registers (faster below ~3500 draws);
 *                 the two differ in the last bits.  0: always the register form.
 *   "single_draw" 1 (default): an un-batched predict() goes through one launch.
 *   "resident"    1: un-batched calls (tc_predict_zheng07_batch with one draw; total
 *                 correlation function, Zheng07 family) are served by ONE resident launch:
 *                 the call writes its parameters into a mailbox in page-locked memory, the
 *                 workgroups answer through page-locked partial sums -- no launch per call.
 *                 The kernel leaves when no call has arrived for "resident_idle_us"
 *                 microseconds (default 2000, 10 .. 10^6; the next call launches it again),
 *                 after 10 s, and before any other kind of call on the handle and
 *                 tc_table_destroy.  Device-wide synchronisations by the caller
 *                 (hipDeviceSynchronize, hipFree) wait for it at most that idle time.
 *                 2 (default): the library moves a loop of un-batched calls to the resident
 *                 kernel by itself -- from the eighth call on that follows its predecessor
 *                 within 300 us, with the idle time "resident_auto_idle_us" (default 250: the
 *                 caller's device-wide synchronisations wait at most that long) -- and goes
 *                 back to one launch per call for 4096 calls when more than a quarter of 32
 *                 such calls found the kernel gone (a caller that pauses or synchronises the
 *                 device between its calls).  Single draws only; ensembles need 1.
 *                 Whichever of the two kernels serves a call, the bits are the same (round 6:
 *                 both are instances of one body compiled with -ffp-contract=on), and an error
 *                 of the resident kernel in this mode never reaches the caller: the kernel is
 *                 stopped, the call served by a launch, the mode suspended for 4096 calls
 *                 and switched off after three such failures.
 *                 0: one launch per call, always.
 *                 Calls with 24 .. 256 draws in host arrays (an ensemble sampler's step) are
 *                 served the same way by a second resident kernel of one workgroup per CU
 *                 (kernel_args.h: EnsembleArgs): no launch, no copy command, no stream
 *                 synchronisation per call; a draw's result does not depend on the number of
 *                 draws in the call or its place among them.  When its workgroups do not
 *                 all find a place on the chip (other work running) the launched path serves
 *                 the call -- after three such calls all of them, until the option is set
 *                 again.  The kernel holds every CU's
 *                 LDS while it waits: the ensemble kernel of another handle starts when this
 *                 one has left (its idle time), so alternate between handles with a short
 *                 "resident_idle_us" or keep the option to the one table of the sampler.
 *   "resident_min_walkers"  smallest number of draws per call the resident ensemble kernel
 *                 takes (default 24, 2 .. 256): for fewer, one launch of the un-batched kernel
 *                 is faster.
 *   "resident_wait_us"  how long a workgroup of the resident ensemble kernel waits for another
 *                 one inside a call before it gives up (default 20 000, 1 .. 10^6).
 *   "resident_aperture"  1 (default): on large-BAR systems the mailbox of the resident
 *                 ensemble kernel lies in device memory that the host stores into through
 *                 the PCIe aperture (no PCIe reads while the kernel polls); 0: in page-locked
 *                 host memory, as on systems without a large BAR.
 *   "deterministic"  which of a call's possible kernel forms runs, and hence the LAST BITS of its
 *                 results (every form agrees with the reference to ~1e-14; the reference itself,
 *                 tabcorr/tabcorr.py:580-683, is deterministic).
 *                 0 (default): the fastest form for the call -- a function of the table, the
 *                 flags, the entry point (un-batched, host arrays, device pointers,
 *                 asynchronous) and the batch size.  Never of timing or of the number of calls
 *                 made so far (round 6: "autotune_after" is off by default and the two
 *                 un-batched kernels give the same bits), so the same sequence of calls
 *                 returns the same bits in every run; the same DRAW in batches of different
 *                 sizes, or through different entry points, may differ in the last bits.
 *                 1: the same, and "autotune" / "autotune_after" are refused (nothing
 *                 measured can enter the choice).
 *                 2: batch-invariant -- ONE form per (table, flags): the one-launch kernel
 *                 (predict_fused_kernel with 64-draw workgroups, or the mode-cross one-launch
 *                 kernel with one workgroup per 64 draws) for every entry point and every
 *                 batch size, one draw included, so that a draw's (ngal, xi) depends on the
 *                 draw alone: wherever it sits in whatever batch, synchronous or asynchronous,
 *                 first call or millionth.  Where no one-launch form serves the table and the
 *                 flags (tc_table_batch_invariant says: float32 tables, several r tiles, more
 *                 than 248 bins, interpolators in mode auto, fused likelihood of separated
 *                 components) the call runs as with 1.  What it costs: un-batched calls
 *                 ~70 us instead of ~10-18 (a whole workgroup's latency for one draw); batches
 *                 below ~8000 draws up to 2x (64-draw workgroups do not fill the chip);
 *                 batches of 10^4 draws and more nothing (it is the default form there).
 *                 Setting the option waits for the handle's work in flight.
 *   "trace"       developer timelines (developer builds only, tabcorr_amd_testing.h). */
int tc_table_set_option(tc_table* table, const char* name, int value);
/* The handle's resident kernels so far: launches (of either kernel), calls that found the
 * single-draw kernel gone and launched it again, calls the automatic mode handed back to a
 * launch after an error (it switches itself off at three), and whether one is running now
 * (0 no, 1 the single-draw kernel, 2 the ensemble kernel).  Any pointer may be NULL. */
int tc_table_resident_stats(const tc_table* table, int64_t* launches, int64_t* relaunches,
                            int64_t* failures, int* running);
/* *out = 1 when option "deterministic" = 2 is set AND a one-launch form serves this table with
 * these predict flags and n_gauss_prim: every batched, un-batched and asynchronous call with
 * them then runs that one form (a draw's bits depend on the draw alone); else 0. */
int tc_table_batch_invariant(tc_table* table, int n_gauss, unsigned flags, int* out);
/* What option "autotune" measured for `flags`: per batch size (`count` of them, at most
 * `capacity`: 9) the form chosen (0 three kernels, 64 / 32: draws per workgroup of the
 * one-launch form) and the microseconds per call of the three forms in that order, us
 * (count, 3) (0: form not available). */
int tc_table_autotune_result(const tc_table* table, unsigned flags, int capacity, int* count,
                             int64_t* sizes, int* forms, float* us);

/* ---- measurement -------------------------------------------------------------------
 * HIP events on the handle's own stream (torch.cuda.Event would only see torch's).
 * tc_table_timer_begin/end bracket a region; with profile = 1 every launch of the
 * contraction kernel -- of this table, or of an interpolator whose first table this is --
 * additionally carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch's
 * own begin and end, the interval rocprofv3 --kernel-trace reports) and
 * tc_table_kernel_time reports the count and mean duration since timer_begin.  profile = n
 * > 1: every n-th launch only, from the first on (a pair of events costs a launch ~1.5 us on
 * its queue; with n = the number of lanes the sampled launches are those of one lane). */
int tc_table_timer_begin(tc_table* table, int profile_kernels);
int tc_table_timer_end(tc_table* table, float* elapsed_ms);
int tc_table_kernel_time(tc_table* table, int* n_launches, float* mean_ms);
/* Launch geometry of the last predict call (for DESIGN.md / bench.py reporting). */
int tc_table_last_launch(const tc_table* table, int* n_workgroups, int* waves_per_workgroup,
                         int* n_splits, int* lds_bytes);

/* ---- tabulation: pair counting ---------------------------------
 *
 * tc_pair_count_rppi replaces one Corrfunc.theory.DDrppi call of tabcorr/corrfunc.py:62-84
 * (the pair count behind `wp` for one pair of halo bins): ORDERED pair counts between the
 * points pos1 (n1, 3) and pos2 (n2, 3) -- pos2 = NULL: of pos1 with itself, every pair then
 * counted twice as Corrfunc's autocorr = 1 does -- in a periodic box `boxsize` (3), per
 * projected-separation bin (rp_bins: n_rp + 1 increasing edges) and line-of-sight bin
 * (n_pi equal bins on [0, pi_max)): npairs (n_rp, n_pi), row-major.  Separations are
 * minimum-image; r_p^2 = dx dx + dy dy is compared with rp_bins^2; a pair with i == j only
 * counts when rp_bins[0] == 0.  Integer counters: the result is exact and independent of
 * the execution order.
 *
 * tc_pair_count_rppi_labelled is the loop of tabcorr/tabcorr.py:846-922
 * (compute_tpcf_matrix: one pair count per pair of halo bins, from a process pool) in ONE
 * pass: label1 / label2 give the halo bin of every point (0 <= label < n_labels) and
 * counts (n_rp, n_labels, n_labels) receives, for every pair of bins (a, b), the ordered
 * pair counts between the points of bin a in set 1 and of bin b in set 2 (pos2 = NULL:
 * set 1 with itself), summed over |pi| < pi_max.  Positions must lie in [0, boxsize]. */
int tc_pair_count_rppi(const double* pos1, int64_t n1, const double* pos2, int64_t n2,
                       const double* boxsize, const double* rp_bins, int n_rp, double pi_max,
                       int n_pi, uint64_t* npairs);
/* tc_pair_count_smu replaces one Corrfunc.theory.DDsmu call of tabcorr/corrfunc.py:141-163
 * (the pair count behind `s_mu_tpcf`): ordered pair counts per bin of the three-dimensional
 * separation s (s_bins: n_s + 1 increasing edges, compared squared) and of mu = |dz| / s
 * (n_mu equal bins on [0, 1); a pair with mu == 1 is not counted): npairs (n_s, n_mu). */
int tc_pair_count_smu(const double* pos1, int64_t n1, const double* pos2, int64_t n2,
                      const double* boxsize, const double* s_bins, int n_s, int n_mu,
                      uint64_t* npairs);
int tc_pair_count_rppi_labelled(const double* pos1, const int32_t* label1, int64_t n1,
                                const double* pos2, const int32_t* label2, int64_t n2,
                                int n_labels, const double* boxsize, const double* rp_bins,
                                int n_rp, double pi_max, uint64_t* counts);
/* The same one-pass form for `s_mu_tpcf` tables (tabcorr/tabcorr.py:846-922 with
 * tpcf = tabcorr/corrfunc.py:98-175): counts (n_s, n_mu, n_labels, n_labels). */
int tc_pair_count_smu_labelled(const double* pos1, const int32_t* label1, int64_t n1,
                               const double* pos2, const int32_t* label2, int64_t n2,
                               int n_labels, const double* boxsize, const double* s_bins,
                               int n_s, int n_mu, uint64_t* counts);

/* tc_mass_in_cylinders is the pair count behind an excess-surface-density table: what
 * TabCorr.tabulate does per halo bin when it is handed halotools' mean_delta_sigma with
 * tpcf_args = (particle positions, particle masses, rp_bins) (tabcorr/tabcorr.py:846-922 in
 * mode 'cross'; scripts/tabulate_snapshot.py:228-237), for the objects of ALL bins in one
 * pass: mass (n_objects, n_edges) receives, per object and per radius rp_bins[k], the summed
 * mass of the particles whose projected separation r = sqrt(dx^2 + dy^2) (periodic in x and y,
 * the line of sight spanning the box) is <= rp_bins[k].  masses = NULL: unit masses (counts).
 * Sums run in a fixed order (reproducible; exact for equal masses). */
int tc_mass_in_cylinders(const double* objects, int64_t n_objects, const double* particles,
                         int64_t n_particles, const double* masses, const double* boxsize,
                         const double* rp_bins, int n_edges, double* mass);

/* ---- multi-GPU: one process per GPU, results collected with one RCCL gather ---------- */

#define TC_UNIQUE_ID_BYTES 128
int tc_comm_unique_id(void* id /* TC_UNIQUE_ID_BYTES */);
int tc_comm_create(const void* id, int n_ranks, int rank, tc_comm** comm);
int tc_comm_destroy(tc_comm* comm);
/* Gather `count` doubles from every rank's send_device into recv_device on `root`
 * (rank-major), on the communicator's own stream, after the work queued so far on
 * `table`'s stream (NULL: no dependency).  `slot` in [0, 4) names the send buffer:
 * tc_comm_release(comm, table, slot) later makes `table`'s stream wait on the device
 * for that gather, so that the buffer can be overwritten -- this is what lets the gather
 * of batch k overlap with the prediction of batch k + 1. */
int tc_comm_gather(tc_comm* comm, tc_table* table, const double* send_device,
                   double* recv_device, int64_t count, int root, int slot);
int tc_comm_release(tc_comm* comm, tc_table* table, int slot);
/* The same for results produced by an interpolator handle (Interpolator.predict sharded
 * over the GPUs: BASELINE configs[3], interpolator.py:124-216). */
int tc_comm_gather_interp(tc_comm* comm, tc_interp* interp, const double* send_device,
                          double* recv_device, int64_t count, int root, int slot);
int tc_comm_release_interp(tc_comm* comm, tc_interp* interp, int slot);
int tc_comm_barrier(tc_comm* comm);
int tc_comm_synchronize(tc_comm* comm);

#ifdef __cplusplus
}
#endif

#endif /* TABCORR_AMD_H */
