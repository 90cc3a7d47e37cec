// Shared between the translation units of libtabcorr_hip.so: error reporting, device and
// pinned buffers, the table handle, and the launch layer (launch.hip) the entry points in
// table.cpp / interp.cpp / comm.cpp call.  Not installed; the public surface is
// include/tabcorr_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tabcorr_amd.h"
#include "../../include/tabcorr_amd_testing.h"
#include "fastmath.h"
#include "hostmath.h"
#include "kernel_args.h"

namespace tc {
namespace host {

// Records the message tc_last_error() returns on this thread; returns `code`.
int fail(int code, const char* format, ...);
const char* last_error();

// roctx ranges around the stages of a call (upload, occupation, contraction, finalisation,
// download, gather), visible with `rocprofv3 --marker-trace`.  The marker library is
// resolved at run time; without it the calls do nothing.
void range_push(const char* name);
void range_pop();
struct Range {
  explicit Range(const char* name) { range_push(name); }
  ~Range() { range_pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};

#define TC_HIP(call)                                                          \
  do {                                                                        \
    hipError_t tc_hip_status = (call);                                        \
    if (tc_hip_status != hipSuccess)                                          \
      return ::tc::host::fail(TC_ERR_HIP, "%s failed: %s (%s:%d)", #call,     \
                              hipGetErrorString(tc_hip_status), __FILE__,     \
                              __LINE__);                                      \
  } while (0)

#define TC_CHECK(condition, ...)                                              \
  do {                                                                        \
    if (!(condition)) return ::tc::host::fail(TC_ERR_INVALID, __VA_ARGS__);   \
  } while (0)

// A device allocation that can only grow (never reallocated while a launch
// that uses it may be in flight: growth synchronises the stream first).
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need, hipStream_t stream) {
    if (need <= bytes) return TC_OK;
    if (ptr != nullptr) {
      TC_HIP(hipStreamSynchronize(stream));
      TC_HIP(hipFree(ptr));
      ptr = nullptr;
      bytes = 0;
    }
    size_t grow = need + need / 4;
    TC_HIP(hipMalloc(&ptr, grow));
    bytes = grow;
    return TC_OK;
  }
  void release() {
    if (ptr != nullptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// Page-locked host staging for small transfers: hipMemcpyAsync from / to pageable
// memory goes through an internal bounce buffer and costs tens of microseconds per
// call, which dominates the latency of un-batched predict() calls.
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return TC_OK;
    if (ptr != nullptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
    size_t grow = std::max<size_t>(need + need / 2, 4096);
    TC_HIP(hipHostMalloc(&ptr, grow, hipHostMallocDefault));
    bytes = grow;
    return TC_OK;
  }
  void release() {
    if (ptr != nullptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};
// Whether [ptr, ptr + bytes) lies in page-locked memory known to the library
// (tc_host_alloc / tc_host_register; runtime.cpp).
// `device_ptr` receives the address the device sees the range at (NULL when it has none).
bool is_pinned(const void* ptr, size_t bytes, void** device_ptr = nullptr);
// memcpy shared with a few helper threads from 256 KB on (runtime.cpp: CopyPool): the results'
// way out of the staging areas at the end of a synchronous host call.
void parallel_copy(void* dst, const void* src, size_t bytes);
// Helper threads of that copy (0: none) and how long they poll for the next job before they
// sleep (tc_set_copy_threads).
void configure_copy_pool(int helpers, int spin_us);

// (developer builds only: environment overrides)
inline int env_int_early(const char* name, int fallback) {
#ifdef TC_DEVELOPER_KNOBS
  const char* value = getenv(name);
  if (value != nullptr && *value != 0) return atoi(value);
#endif
  (void)name;
  return fallback;
}
// Larger transfers go directly.
inline size_t stage_limit() {
  static const size_t limit = (size_t)env_int_early("TC_STAGE_LIMIT_MB", 1) << 20;
  return limit;
}

// Calls moving at most this many bytes skip the copy commands: the kernels address the
// page-locked staging buffers directly.  512 KB (2000-2500 draws of a 19-bin table: 67 us per
// call against 83 with copies); beyond it the finalisation kernel's transposed stores over
// PCIe lose to the copy engines (3000 draws: 125 us against 105, tools/archive/r02_host_limits.sh).
inline size_t zero_copy_limit() {
  static const size_t limit = (size_t)env_int_early("TC_ZERO_COPY_KB", 512) << 10;
  return limit;
}

// The streams of a handle's lanes.  Whether their kernels overlap depends on which hardware
// queues the runtime gives them (four per process): created while one to three other streams
// exist -- the null stream after a first hipMemcpy is enough -- two of four lanes end up on one
// queue and a 10^4-draw step of the AbacusSummit table takes 77 us instead of 60
// (tools/r04_streams.py).  With none or at least four streams present the four lanes land on
// four queues: so four short-lived ballast streams are created first and destroyed afterwards.
int create_lane_streams(int count, hipStream_t* streams);

// Page-locked workspace of the un-batched path (launch.hip: launch_single_draw): the kernel
// stores its results and one completion word per workgroup here, the host polls the words.
struct SingleWorkspace {
  PinnedBuffer buffer;
  unsigned long long epoch = 0;
  int jobs = 0, blocks = 0;
  size_t partial_offset = 0, theta_offset = 0, done_offset = 0;   // in doubles
  int prepare(int n_jobs, int n_blocks, int rt, int extra_doubles);
  double* ngal() const { return (double*)buffer.ptr; }
  double* partial() const { return (double*)buffer.ptr + partial_offset; }
  double* theta() const { return (double*)buffer.ptr + theta_offset; }
  unsigned long long* done() const {
    return (unsigned long long*)((double*)buffer.ptr + done_offset);
  }
};

template <typename T>
int upload(const std::vector<T>& host, void** device) {
  size_t bytes = std::max<size_t>(1, host.size()) * sizeof(T);
  TC_HIP(hipMalloc(device, bytes));
  if (!host.empty())
    TC_HIP(hipMemcpy(*device, host.data(), host.size() * sizeof(T),
                     hipMemcpyHostToDevice));
  return TC_OK;
}

struct DeviceChunking {
  tc::Chunking host;
  void* chunks = nullptr;
  void* groups = nullptr;
};

struct Quadrature {
  int n_gauss = 0;
  void* log_m = nullptr;
  void* m = nullptr;
  void* weight = nullptr;
  // the same constants for the GROUPED kernels (n_gauss = 10, tables with groups of bins):
  // nodes per group, weights (+ their sums) per member in group order (kernel_args.h: GroupArgs)
  void* group_log_m = nullptr;
  void* group_m = nullptr;
  void* group_weight = nullptr;
  // moment expansion of the central bins (series.h): per bin, and per member in group order
  void* series = nullptr;
  void* series_thr = nullptr;
  void* group_series = nullptr;
  void* group_series_thr = nullptr;
  void* sat_series = nullptr;          // satellites (series.h, namespace sat), as above
  void* sat_series_thr = nullptr;
  void* group_sat_series = nullptr;
  void* group_sat_series_thr = nullptr;
  void* cen_records = nullptr;         // series.h, namespace cen_record (central bins)
  void* sat_records = nullptr;         // series.h, namespace sat_record (satellite bins)
  void* group_records = nullptr;       // series.h, namespace record (groups of <= 2 members)
};

// A schedule of the quadratic-form kernel on the device (hostmath.h: QuadSchedule).
struct DeviceQuadSchedule {
  int n_waves = 0;
  int n_slabs = 0;
  int n_groups = 0;
  int n_runs = 0;
  void* runs = nullptr;
  void* wave_runs = nullptr;
  void* wave_head = nullptr;     // (n_waves, 16): QuadArgs::wave_head
  void* group_begin = nullptr;
  void* merge_range = nullptr;   // (workgroups, 2)
  void* merges = nullptr;        // QuadMerge per merge
  int lds_bytes = 0;             // merge slots of the busiest workgroup
};

// One layout of a table for the quadratic-form kernel: the re-laid-out matrix, the
// components, and the schedules built so far, keyed by (draw tiles, separate, tables).
struct QuadTable {
  tc::QuadLayout layout;
  void* d_table = nullptr;      // (n_rtiles, n_units, (n_u + 1) / 2, 64, 2) doubles
  void* d_comps = nullptr;      // QuadCompArgs per component
  bool finite = false;          // float64: every entry is finite and at most 1e20 in size (run_fused)
  size_t rtile_bytes = 0;
  size_t bytes = 0;
  std::map<std::vector<int64_t>, std::unique_ptr<DeviceQuadSchedule>> schedules;
  void release();
  // Frees every cached schedule (the caller makes sure no kernel still reads them).
  void drop_schedules();
};

// Coefficients of predict_cross_fused_kernel (mode cross, one launch per batch) for one table
// or for the K tables of an interpolator: per member bin (group order of the first table) the
// rows T_k[r][g] n_h,k[g] and n_h,k[g]; built at the first call that may use them.
struct CrossFused {
  bool tried = false;
  int rows = 0;              // ROWS of the kernel instance; 0: not available for this handle
  int n_tables = 0;
  void* d_rows = nullptr;
  // chunks of whole groups of at most kCrossChunkBins bins (kernel_args.h: CrossFusedArgs)
  void* d_chunk_group = nullptr;
  void* d_chunk_block = nullptr;     // first 4-bin step of every chunk (matrix-operand layout)
  void* d_bin_operand = nullptr;     // where a member bin's operands start (deferred pairs)
  std::vector<int32_t> chunk_group_host;
  int n_chunks = 0, n_central_chunks = 0;
  void release() {
    if (d_rows != nullptr) (void)hipFree(d_rows);
    if (d_chunk_group != nullptr) (void)hipFree(d_chunk_group);
    if (d_chunk_block != nullptr) (void)hipFree(d_chunk_block);
    if (d_bin_operand != nullptr) (void)hipFree(d_bin_operand);
    d_rows = d_chunk_group = d_chunk_block = d_bin_operand = nullptr;
    rows = 0;
    tried = false;
  }
};

// Developer knobs.  The release build never reads the environment: tuning values are the
// defaults of `Tuning` below (read once per handle in developer builds, -DTC_DEVELOPER_KNOBS,
// so that parameter sweeps stay possible) and the few run-time options a caller may
// legitimately change go through tc_table_set_option().
inline int env_int(const char* name, int fallback) {
#ifdef TC_DEVELOPER_KNOBS
  const char* value = getenv(name);
  if (value == nullptr || *value == 0) return fallback;
  return atoi(value);
#else
  (void)name;
  return fallback;
#endif
}

struct Tuning {
  int lanes = 4;              // pipelining lanes of device-pointer calls (1..4)
  int pipeline = 1;           // 0: every device-pointer call on lane 0 (kernels serialised)
  int single_draw = 1;        // one-launch path for un-batched predict()
  int many_blocks = 256;      // workgroups of a launch that several walkers share: one round
                              // (64 walkers 41 -> 37 us per call against 128, 512: 44)
  int single_round = 1;       // un-batched Interpolator.predict: all tables' workgroups at once
  int poll_done = 1;          // ... completed by polling its completion words in host memory
  int quad_waves = 0;         // resident contraction waves per SIMD (quadratic-form kernel;
                              // 0: chosen by the matrix work per draw tile, launch.hip)
  int quad_waves_f32 = 3;     // ... of the float32 kernel
  int quad_merge = 1;         // workgroup-level merging of the partial slabs (hostmath.h)
  int quad_order = -1;        // schedule order of one table (-1: chosen by matrix size)
  // wave priorities.  The occupation kernel heads each lane's chain (occupation ->
  // contraction -> finalisation); below the contraction's priority its dependent chains of
  // FP64 instructions wait behind 64-cycle matrix instructions and the kernel takes 85-95 us
  // in the pipeline (25 us alone), which the lane's next two kernels then wait for: 44.2 us
  // per step with priority 0, 43.2 with 1, 42.6 with 2, 48.1 with 3 (tools/archive/r02_sweep.sh)
  int prio_occ = 2, prio_contract = 1, prio_finalize = 3;
  int finalize_threads = 0;   // 0: chosen per batch size
  int finalize_row_blocks = 0;
  int occ_splits = 0;         // 0: chosen per batch size
  int occ_per_cu = 4;
  int n_groups = 0, n_waves = 0, lds_min = 0, min_chunk_entries = 32;   // segment kernels
  int k_splits = 0;           // interpolator, segment kernels: table splits (0: chosen)
  int trace = 0;              // developer timelines
  // Asynchronous host calls (10^4 draws per call, four lanes, tools/archive/r03_async_sweep.sh): with
  // copy commands both ways 64-68 us per call; the occupation kernel reading the draws from
  // the caller's page-locked memory itself (one command and one engine hop of ~10 us less in
  // the lane's chain) 58; small results (number densities, chi2) stored by the finalisation
  // kernel directly 57 / the likelihood 47.5; ALL results stored directly 66 (1.5 MB of
  // 64-byte writes over PCIe from the shader cores lose to the copy engine); downloads on a
  // stream of their own 91-105 (cross-stream events), more lanes 72-80 (four hardware queues).
  // synchronous host calls as overlapping chunks of draws (table.cpp: predict_chunked):
  // sync_chunks 0 auto (from 2048 draws on), -1 never, N >= 1 that many; sync_form: draws per
  // workgroup of the one-launch form the chunks take where it serves the table (so that the
  // result does not depend on the number of chunks; 0: whatever a pipelined call of the chunk's
  // size would take); sync_direct_out as async_direct_out, for the staging area of the chunks
  // the dispatch measures itself (option "autotune") at this many pipelined / asynchronous
  // calls with one combination of predict flags; 0 (default since round 6): never by itself --
  // the measurement stalls a call documented as non-blocking for ~0.5 s and its outcome comes
  // from wall-clock timings, so the same draws could return different last bits before and
  // after it and from run to run: TabCorr.autotune() asks for it explicitly
  int autotune_after = 0;
  int sync_chunks = 0;
  int sync_form = 32;
  int sync_stagger = 10;      // large results: chunks one after the other, shrinking to this
                              // many per cent of the first (0: off; table.cpp)
  int sync_direct_out = 2;
  int async_direct_in = 1;    // 1: kernels read the draws from the caller's pinned memory
  int async_direct_out = 2;   // 0: copy commands, 1: kernels store everything, 2: kernels
                              // store arrays up to kDirectOutBytes, copy commands beyond
  // One launch per batch (predict_fused_kernel) for the calls it covers: 0 never, 1 the
  // pipelined device-pointer and asynchronous calls of fused_min_draws .. fused_max_draws
  // draws, 2 every call of that size.  Sustained device-resident rate on four lanes
  // (tools/archive/r03_fused_scan.py, BASELINE configs[1]'s table; three kernels / one launch, us per
  // call): 4096 draws 23.6 / 30.7, 6144 31.4 / 31.5, 7168 34.9 / 32.0, 8192 35.6 / 33.0, 10^4
  // 42.7 / 39.6, 16384 66.7 / 63.5, 28672 111.7 / 111.0, 32768 126.3 / 126.6,
  // 40000 151.8 / 156.3 -- both designs approach the same 38 us per 10^4 draws for huge
  // batches (matrix + vector instructions on one FP64 pipe); the one-launch form gets there
  // with the batch sizes an ensemble sampler has, the three kernels spread small batches
  // over the whole chip.  Asynchronous host calls (us per call, tools/archive/r03_async.py): 6144 draws
  // 38.8 / 41.6, 10^4 58 / 47.6, 20 000 93.7 / 82.6, 40 000 176 / 162: no upper bound there.
  // moment expansions of the node sums (series.h): bit 0 central bins, bit 1 satellite bins;
  // -1 (default): the centrals' for tables with narrow bins (launch.hip: series_mask).  Every
  // lane takes the terms ITS draw needs (a draw's bits must not depend on its neighbours in
  // the batch), and a wave runs the node loop as well wherever one of its draws needs it.
  int series = -1;
  int cross_defer = 1;          // mode cross, one launch: deferred (group, draw) pairs (kernel_args.h)
  int fused_defer = 2;          // predict_fused_kernel: 1 the satellites' expansion + deferred
                                // pairs, 2 the centrals' likewise (where their expansion is on)
  int fused_sat_cap = 6;        // ... in place up to 12 + 4 x this many terms (0 .. 5), or
                                // 6 = series::sat::kShortest: the shortest that serves the bin
  int cross_wide_min_draws = 4096;   // launch.hip: choose_cross_fused (0: never the wide form)
  int cross_target = 160;       // mode cross: workgroups a launch should have at least
                                // (several per tile of 64 draws below that)
  int resident_aperture = 1;    // resident ensemble kernel: the mailbox in device memory that
                                // the host writes through the PCIe aperture (large BAR only)
  int cross_min_draws = 192;    // ... smallest batch of the chunked form (17 - 128 rows;
                                // tools/r04_cross_scan.py, AbacusSummit interpolator, us per call
                                // one launch / three kernels: 256 draws 15.7 / 20.6, 1024 20.6 /
                                // 26.0, 4096 35.1 / 68.1, 10^4 79.6 / 164.6)
  int fused = 1;
  int fused_min_draws = 0;      // 0: chosen per table (launch.hip: fused_eligible)
  int fused_max_draws = 30720;
  int fused_waves = 0;          // 0: 8 where two workgroups fit a CU, else 16; 8 / 16: forced
  int fused_draws = 0;          // draws per workgroup of the one-launch form: 0 = 32 for batches
                                // below 8192 draws where that form applies, else 64
                                // (launch.hip: fused_half_tiles); 32 / 64: forced
  // the latency form of the one-launch kernel (40 draws per workgroup, one workgroup per CU:
  // launch.hip: fused_spread_eligible) for calls that run alone: 1 on; smallest batch; largest
  // batch in rounds of one workgroup per CU
  int fused_spread = 1, fused_spread_min = 8192, fused_spread_rounds = 1;
  int prio_fused = 1, prio_fused_occ = 2, prio_fused_out = 3;   // phases 2, 1, 3
  // Reproducible bits on request (round 6).  0: the fastest form per call -- a function of the
  // table, the flags, the entry point and the batch size (never of timing: autotune_after is
  // off by default); 1: the same, and the measured dispatch ("autotune", "autotune_after") is
  // refused; 2: BATCH-INVARIANT -- one kernel form per (table, flags) for every entry point and
  // every batch size, one draw included, wherever a one-launch form serves the table
  // (launch.hip: batch_invariant_form), so that a draw's (ngal, xi) depends on the draw alone
  int deterministic = 0;
  int skip_occ = 0, skip_finalize = 0;   // diagnosis (developer builds only)
  void load() {
    fused = env_int("TC_FUSED", fused);
    fused_min_draws = env_int("TC_FUSED_MIN_DRAWS", fused_min_draws);
    fused_max_draws = env_int("TC_FUSED_MAX_DRAWS", fused_max_draws);
    prio_fused = env_int("TC_PRIO_FUSED", prio_fused);
    prio_fused_occ = env_int("TC_PRIO_FUSED_OCC", prio_fused_occ);
    prio_fused_out = env_int("TC_PRIO_FUSED_OUT", prio_fused_out);
    lanes = env_int("TC_LANES", lanes);
    pipeline = env_int("TC_PIPELINE", pipeline);
    single_draw = env_int("TC_SINGLE_DRAW", single_draw);
    quad_waves = env_int("TC_QUAD_WAVES", quad_waves);
    quad_waves_f32 = env_int("TC_QUAD_WAVES_F32", quad_waves_f32);
    quad_merge = env_int("TC_QUAD_MERGE", quad_merge);
    prio_occ = env_int("TC_PRIO_O", prio_occ);
    prio_contract = env_int("TC_PRIO_C", prio_contract);
    prio_finalize = env_int("TC_PRIO_F", prio_finalize);
    finalize_threads = env_int("TC_FINALIZE_THREADS", finalize_threads);
    finalize_row_blocks = env_int("TC_FINALIZE_ROW_BLOCKS", finalize_row_blocks);
    occ_splits = env_int("TC_OCC_SPLITS", occ_splits);
    occ_per_cu = env_int("TC_OCC_PER_CU", occ_per_cu);
    n_groups = env_int("TC_NGROUPS", n_groups);
    n_waves = env_int("TC_NWAVES", n_waves);
    lds_min = env_int("TC_LDS_MIN", lds_min);
    min_chunk_entries = env_int("TC_MIN_CHUNK_ENTRIES", min_chunk_entries);
    k_splits = env_int("TC_KSPLITS", k_splits);
    trace = env_int("TC_TRACE", trace);
    skip_occ = env_int("TC_SKIP_OCC", skip_occ);
    skip_finalize = env_int("TC_SKIP_FINALIZE", skip_finalize);
  }
};

}  // namespace host
}  // namespace tc

// Measured choice of the form a batch takes (option "autotune"; table.cpp: autotune): for a
// grid of batch sizes the fastest of {three kernels, one launch with 64-draw workgroups, one
// launch with 32-draw workgroups} in the pipelined regime, per combination of predict flags.
// launch.hip: fused_eligible / fused_half_tiles ask it before their formula.
struct AutoChoice {
  static constexpr int kSizes = 9;
  // batch sizes measured (geometric: a batch takes the choice of the nearest one)
  static constexpr int64_t size(int i) { return (int64_t)256 << i; }      // 256 .. 65536
  int form[kSizes] = {};         // 0 three kernels, 32 / 64 draws per workgroup of one launch
  float us[kSizes][3] = {};      // measured us per call: three kernels, 64 draws, 32 draws
  // The form for a batch of n draws: every form's time interpolated linearly between the two
  // measured sizes around n (a form that is missing at either end is out; beyond the grid the
  // nearest end decides), the fastest wins, a one-launch form only by 2 %.
  int form_for(int64_t n_draws) const {
    if (n_draws <= size(0)) return form[0];
    if (n_draws >= size(kSizes - 1)) return form[kSizes - 1];
    int i = 0;
    while (size(i + 1) < n_draws) ++i;
    const double w = (double)(n_draws - size(i)) / (double)(size(i + 1) - size(i));
    static const int shape[3] = {0, 64, 32};
    int best = 0;
    double best_us = (1.0 - w) * us[i][0] + w * us[i + 1][0];
    for (int k = 1; k < 3; ++k) {
      if (us[i][k] <= 0.0f || us[i + 1][k] <= 0.0f) continue;
      const double value = (1.0 - w) * us[i][k] + w * us[i + 1][k];
      if (value < (best == 0 ? 0.98 : 1.0) * best_us) {
        best_us = value;
        best = shape[k];
      }
    }
    return best;
  }
};

struct tc_table {
  int device = 0;
  int mode = 0;
  int n_bins = 0;
  int n_r = 0;
  int64_t n_pairs = 0;
  int compute_dtype = TC_DTYPE_F64;
  bool legacy = false;
  tc::Plan plan;
  int rt = 0;          // r values per tile (compile-time kernel parameter)
  int n_rtiles = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;

  // host copies of the gal_type columns in library (centrals-first) order
  std::vector<double> n_h, log_min, log_max, percentile, dist_index;

  void* d_table = nullptr;       // (n_rtiles, n_entries, rt)
  size_t table_bytes = 0;
  void* d_n_h = nullptr;
  void* d_percentile = nullptr;
  void* d_perm = nullptr;
  // Groups of bins with the same quadrature nodes (hostmath.h: find_node_groups).  `grouped`:
  // some group has more than one member and the option "grouped" is on -- the occupation
  // kernels then evaluate every group's nodes once (kernels.hip.h: occ_group_zheng07).
  tc::NodeGroups node_groups;
  void* d_group_begin = nullptr;
  void* d_group_member = nullptr;
  void* d_group_n_h = nullptr;         // n_h and sec_haloprop_percentile in member order
  void* d_group_percentile = nullptr;
  bool grouped = false;
  void* d_math_table = nullptr;  // fastmath.h tables
  void* d_pos_ij = nullptr;      // float32 variant: packed bin pairs per position
  void* d_pos_off = nullptr;     // FP64 kernel: LDS row byte offsets per position
  // Mode auto in float64 runs the quadratic-form kernel (kernels.hip.h): the matrix by
  // galaxy type (cen-cen / cen-sat / sat-sat, each type padded to whole 4 x 4 blocks; also
  // serves the total when the number of centrals is a multiple of 4) and, otherwise, the
  // unpadded whole triangle for the total prediction.
  // mode cross in float64: the matrix as (n_bins in library order, n_r), kept on the host for
  // the coefficient rows of predict_cross_fused_kernel (also those of interpolators)
  std::vector<double> cross_host;
  tc::host::CrossFused cross_fused, cross_fused_wide;
  // Set while the chunks of a synchronous host call are queued (0 otherwise): the workgroups a
  // mode-cross launch should have at least, instead of Tuning::cross_target -- 512 / chunks, so
  // that a call that has the chip to itself fills it once with shares of tiles (the
  // AbacusSummit interpolator, one chunk: 242 -> 200 us per 10^4 draws host to host; its single
  // table, two chunks: 149 -> 139; pipelined launches keep 160: they share the chip anyway).
  int sync_cross_target = 0;
  std::vector<hipEvent_t> chunk_events;    // ... between the chunks of a call with large results
  // ... and that the chunks of such a call have the chip to themselves: where the latency form
  // of the one-launch kernel serves the table, every chunk takes it when all chunks together
  // have at most one workgroup of 40 draws per CU (launch.hip: fused_spread_eligible)
  bool sync_spread = false;
  bool quad = false;
  tc::QuadTiling quad_tiling;
  tc::host::QuadTable quad_by_type, quad_total;
  int n_cus = 256;               // compute units of the device
  int n_xcds = 8;                // accelerator complexes (each with its own L2)
  tc::host::Tuning tuning;
  std::map<unsigned, AutoChoice> autotuned;      // by predict flags (n_gauss_prim = 10)
  // pipelined / asynchronous calls seen per predict flags: the autotune_after-th one measures
  // the forms by itself (table.cpp: maybe_autotune)
  std::map<unsigned, int> pipelined_calls;
  bool autotuning = false;
  std::map<int, tc::host::Quadrature> quadrature;
  std::map<std::pair<int, int>, std::unique_ptr<tc::host::DeviceChunking>> chunkings;
  std::map<int64_t, tc::host::DeviceChunking*> choices;   // decomposition chosen per tile count

  // Independent "lanes" (stream + workspaces; four by default, TC_LANES).  Consecutive
  // device-pointer predict calls alternate between them, so that the occupation kernel of batch k + 1
  // overlaps the contraction of batch k (both are FP64-issue bound and the contraction
  // leaves issue slots free at its ramp-down); results are still produced in call order
  // (the finalisation kernels are chained by events).  Host-buffer calls use lane 0.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t finished = nullptr;   // recorded after the lane's last finalisation
    tc::host::DeviceBuffer nbuf, ngal2, partial;
    tc::host::DeviceBuffer nbuf32;   // float copy of nbuf (float32 quadratic-form kernel)
    tc::host::DeviceBuffer xi;       // chi2 device calls: the correlation functions
    tc::host::DeviceBuffer in_theta, out;   // asynchronous host calls: staging of draws / results
    tc::host::DeviceBuffer cross_counters;  // mode cross, several workgroups per tile: arrivals
    int ngal_parts = 1;              // partial sums the occupation step left in ngal2
  };
  static constexpr int kMaxLanes = 8;
  Lane lanes[kMaxLanes];
  int n_lanes = 2;
  int prev = -1;                     // lane of the previous finalisation
  int cur = 0;                       // lane of the current / last predict call
  int force_lane = -1;               // host-buffer entry points pin lane 0
  int async_lane = -1;               // asynchronous host calls: the lane already chosen
  // Option "ordered": the finalisations of consecutive device-pointer calls are chained by
  // events so that their results appear in call order.  Off by default: every call only orders
  // its own kernels, tc_table_synchronize / tc_comm_gather wait for every lane (the chain
  // costs 0.3 us per step on prior draws and 3 us on posterior-like ones, whose short
  // occupation kernels then wait for a neighbour's finalisation: tools/archive/r03_clustered.py).
  bool chain = false;
  // Tickets of the asynchronous host calls (tc_*_async): a ring of events; a ticket whose
  // slot was reused is older than every lane's current work.
  struct Ticket {
    hipEvent_t done = nullptr;
    int64_t id = -1;
  };
  static constexpr int kMaxTickets = 64;
  Ticket tickets[kMaxTickets];
  int64_t next_ticket = 0;
  // chi2 calls: where the finalisation may put the fused likelihood (data | precision on
  // the device, output); run_contraction sets chi2_fused when it did
  const double* fuse_chi2_data = nullptr;
  double* fuse_chi2_out = nullptr;
  bool chi2_fused = false;
  uint64_t device_calls = 0;
  tc::host::DeviceBuffer theta, out_ngal, out_xi, occupation, trace, wave_trace;
  tc::host::DeviceBuffer chi2_data;          // data vector + precision matrix of chi2 calls
  std::vector<double> chi2_host;             // host copy of what chi2_data holds
  size_t wave_trace_count = 0;
  tc::host::PinnedBuffer h_in, h_out;
  tc::host::SingleWorkspace single_ws;       // un-batched path
  // Resident un-batched path (launch.hip: resident_predict; option "resident"): one launch of
  // resident_draw_kernel on a stream of its own serves the calls until it is told to stop or
  // none has arrived for idle_us; mailbox = [call number | 7 parameters | one word per
  // workgroup: the launch it has left] in page-locked memory.
  struct Resident {
    int enabled = 0;
    int idle_us = 2000;
    // Automatic mode (default; option "resident" = 2; table.cpp: resident_auto_*): un-batched
    // calls that follow one another within auto_gap_us -- an MCMC's predict(model) loop -- are
    // served by the resident kernel from the auto_streak_min-th on, with the short idle time
    // auto_idle_us (a device-wide synchronisation by the caller waits for the kernel at most
    // that long); when more than a quarter of 32 such calls find the kernel gone (the caller
    // pauses or synchronises between calls: every call would pay a launch AND a stop), the
    // launched path serves the next auto_backoff_calls calls.
    int auto_mode = 1;
    int auto_idle_us = 250, auto_gap_us = 300, auto_streak_min = 8, auto_backoff_calls = 4096;
    int auto_streak = 0, auto_backoff = 0, auto_window = 0, auto_relaunches = 0;
    int auto_failures = 0;               // calls the automatic mode had to hand back to a launch
    int inject_failures = 0;             // test hook: the next calls of resident_predict fail
    long long auto_last_ns = 0;          // when the last un-batched call returned
    bool auto_serving = false;           // the running launch was started by the automatic mode
    unsigned long long relaunches = 0;   // calls that found the kernel gone (all modes)
    int running_idle_us = 0;             // idle time of the launch that is running
    int poll_waves = 1;
    bool running = false;
    unsigned long long launch_id = 0;
    hipStream_t stream = nullptr;
    tc::host::PinnedBuffer mailbox;
    tc::host::DeviceBuffer single_aperture;   // the seven entries in device memory (large BAR)
    bool single_aperture_decided = false;
    tc::host::SingleWorkspace ws;
    int n_theta = 0, n_gauss = 0, blocks = 0;
    unsigned flags = 0;
    // the ensemble form (launch.hip: ensemble_predict; kernel_args.h: EnsembleArgs): which of
    // the two kernels the launch runs, its call counter, its page-locked results and the
    // device memory the phases hand their data on through
    bool ensemble = false;
    unsigned long long ens_epoch = 0;
    int ens_grid = 0;
    tc::host::PinnedBuffer ens_mailbox, ens_out;
    tc::host::DeviceBuffer ens_device;
    tc::host::DeviceBuffer ens_aperture;    // the mailbox in device memory (large BAR), or none
    unsigned long long ens_host_ns[3] = {0, 0, 0};   // last call: published, all rows combined, time spent on rows
    int wait_us = 20000;         // ensemble kernel: limit of a wait for another workgroup
    int min_walkers = 24;        // ... smallest ensemble it takes (fewer walkers: one launch of
                                 // single_draw_kernel is faster -- G = 60: 2 walkers 17.8 against
                                 // 20.3 us, 32: 22.5 / 22.5, 64: 26.4 / 23.3; G = 100: 16 walkers
                                 // 30.1 / 26.4 -- tools/r04_ensemble.py)
    int ens_failures = 0;        // consecutive calls the kernel left before it answered
    bool ens_disabled = false;   // ... three of them: the launched path until "resident" is set again
  } resident;
  size_t trace_blocks = 0;
  size_t trace_launches = 0;

  // measurement
  bool profile_kernels = false;
  int profile_every = 1;            // (tc_table_timer_begin: every n-th launch carries events)
  size_t profile_launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kernel_events;
  size_t kernel_events_used = 0;
  int last_workgroups = 0, last_waves = 0, last_splits = 0, last_lds = 0;
};

namespace tc {
namespace host {

// Asynchronous host calls: result arrays up to this size are stored by the kernels into the
// caller's page-locked memory, larger ones travel by copy command.
constexpr size_t kDirectOutBytes = 256 * 1024;
constexpr size_t kSyncDirectOutBytes = 1024 * 1024;   // (per array and chunk of a synchronous call)
// Largest dynamic LDS allocation a workgroup may ask for (160 KiB per CU).
constexpr int kMaxLdsBytes = 160 * 1024;
// Draws are processed in slabs so that the workspaces stay bounded.
constexpr int64_t kMaxSlab = 1 << 18;

// ---- launch layer (launch.hip) --------------------------------------------------------
int get_quadrature(tc_table* t, int n_gauss, Quadrature** out);
// The GroupArgs of a table's groups of bins (n_gauss = 10).
tc::GroupArgs group_args(const tc_table* t, const Quadrature& q);
int get_chunking(tc_table* t, int n_chunks, int waves, DeviceChunking** out);
int lds_bytes_for(const Chunking& chunking, int rt, int elem = 8);
int wave_slots(const tc_table* t, bool interp);
int blocks_per_cu(int lds_bytes, int waves, int slots);
int choose_chunking(tc_table* t, int64_t n_draws, int tables_per_block, DeviceChunking** out,
                    int* lds_bytes);
int launch_contract_rt(int rt, dim3 grid, dim3 block, int lds, hipStream_t stream,
                       const ContractArgs& args, hipEvent_t start = nullptr,
                       hipEvent_t stop = nullptr);
int set_lds_limit_rt(int rt, int lds);
int launch_contract_f32(dim3 grid, dim3 block, int lds, hipStream_t stream,
                        const ContractArgs& args, hipEvent_t start = nullptr,
                        hipEvent_t stop = nullptr);
// Occupation kernel for a slab of draws: densities into (nbuf, ngal2) -- the current
// lane's by default -- and optionally the occupations in reference order.
int run_occupation(tc_table* t, const double* theta_device, int n_theta, int64_t n_draws,
                   int64_t ldb, int n_gauss, unsigned flags, double* occupation_device,
                   DeviceBuffer* nbuf = nullptr, DeviceBuffer* ngal2 = nullptr,
                   hipStream_t stream = nullptr, int* ngal_parts = nullptr,
                   DeviceBuffer* nbuf32 = nullptr);
// Contraction + finalisation of draws whose densities are already in the current lane.
int run_contraction(tc_table* t, int64_t n_draws, int64_t ldb, unsigned flags,
                    double* ngal_device, double* xi_device);
// One launch per slab (predict_fused_kernel) for the calls it covers; ngal and xi (or the
// likelihood, t->fuse_chi2_out) as run_contraction leaves them.
bool fused_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags);
// Option "deterministic" = 2: is there a one-launch form for this table and these flags (whose
// bits do not depend on the batch)?  Every entry point then takes it for every batch size.
bool batch_invariant_form(tc_table* t, int n_gauss, unsigned flags);
int fused_dens_rows(const tc_table* t, bool separate);
int fused_lds_bytes(const tc_table* t, bool separate, int waves, int draws);
int fused_waves(const tc_table* t, bool separate, unsigned flags);
int series_mask(const tc_table* t);
bool fused_half_tiles(const tc_table* t, bool separate, int64_t n_draws, int n_gauss,
                      unsigned flags);
bool fused_wide_tables(const tc_table* t, bool separate, int n_gauss, unsigned flags);
bool fused_spread_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags);
int run_fused(tc_table* t, const double* theta_device, int n_theta, int64_t n_draws, int n_gauss,
              unsigned flags, double* ngal_device, double* xi_device);
int check_predict_args(const tc_table* t, const void* theta, int n_theta, int64_t n_draws,
                       int n_gauss, unsigned flags);
// Mode cross, one launch per batch (predict_cross_fused_kernel): the coefficient rows of one
// table / K tables with common mass bins (cf->rows == 0 afterwards: not available), whether a
// call takes that form, and the launch (`interp`: the spline part of the arguments, or NULL).
int build_cross_fused(tc_table* const* tables, int n_tables, CrossFused* cf, bool wide = false);
// ... and which of a handle's two sets a call takes: `narrow` (the instance the row count asks
// for) or, for tables of up to 16 rows, `wide` (32 rows: the chunk form) -- built on first use.
CrossFused* choose_cross_fused(tc_table* const* tables, int n_tables, CrossFused* narrow,
                               CrossFused* wide, int64_t n_draws, unsigned flags, int* status);
bool cross_fused_eligible(const tc_table* t0, const CrossFused& cf, int64_t n_draws, int n_gauss,
                          unsigned flags, bool alone);
int run_cross_fused(tc_table* t0, const CrossFused& cf, const tc::CrossFusedArgs* interp,
                    const double* theta_device, int n_theta, int64_t n_draws, unsigned flags,
                    double* ngal_device, double* xi_device, hipStream_t stream,
                    DeviceBuffer* partial, DeviceBuffer* counters);
int launch_finalize(const FinalizeArgs& args, const Tuning& tuning, hipStream_t stream);
// Quadratic-form path: layout upload, schedules, launches.
int build_quad_table(tc_table* t, bool by_type, const void* matrix, int matrix_dtype,
                     QuadTable* out);
int get_quad_schedule(tc_table* t, QuadTable* q, int64_t n_tiles, int n_tables, bool separate,
                      DeviceQuadSchedule** out);
int launch_contract_quad(int n_u, bool interp, const QuadArgs& args, int lds_bytes,
                         hipStream_t stream, hipEvent_t start, hipEvent_t stop);
int launch_finalize_quad(const FinalizeQuadArgs& args, const Tuning& tuning, hipStream_t stream,
                         bool f32 = false);
int launch_contract_quad_f32_interp(int n_u, const tc::QuadArgs& args, int lds_bytes,
                                    hipStream_t stream, hipEvent_t start, hipEvent_t stop);
int launch_contract_quad_f32(int n_u, const QuadArgs& args, int lds_bytes, hipStream_t stream,
                             hipEvent_t start, hipEvent_t stop);
// Stream an interpolator's work is queued on (interp.cpp).
hipStream_t interp_stream(tc_interp* interp);
int interp_join_lanes(tc_interp* interp, hipStream_t stream);
int interp_lanes_wait(tc_interp* interp, hipEvent_t event);
// Events for hipExtLaunchKernelGGL while the table's kernel timer is on, else NULLs.
int next_kernel_events(tc_table* t, hipEvent_t* start, hipEvent_t* stop);
// Largest number of draws one slab may hold (workspaces bounded; 32-bit scalar offsets of
// the quadratic-form kernel: n_bins * ldb * 8 < 2^32).
int64_t max_slab(const tc_table* t);
bool single_draw_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags);
constexpr int kSingleMaxBlocks = 64;
constexpr int kSingleMaxWalkers = 64;
// Batches of at most this many draws take the one-launch path (tools/archive/r03_latency.py: 1 draw
// 15 us, 16 draws 25 us, 64 draws 35 us against 36-37 us for the three-kernel path).
inline int64_t many_walkers_limit() {
  static const int64_t limit =
      std::min<int64_t>(kSingleMaxWalkers, env_int_early("TC_MANY_WALKERS", 64));
  return limit;
}
int resident_predict(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags,
                     double* ngal, double* xi);
int resident_post(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags);
int resident_collect(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags,
                     double* ngal, double* xi);
int resident_stop(tc_table* t);
bool resident_eligible(const tc_table* t, int n_gauss);
// Is a single-draw resident kernel of ANOTHER handle running on this handle's device?
bool other_resident_running(const tc_table* t);
// 2 .. kEnsembleMaxWalkers draws through the resident ensemble kernel, host to host.
bool ensemble_eligible(const tc_table* t, int64_t n_walkers, int n_gauss, unsigned flags);
int ensemble_predict(tc_table* t, const double* theta, int n_theta, int n_walkers, int n_gauss,
                     unsigned flags, double* ngal, double* xi);
int launch_single_draw(tc_table* t, const double* theta, int n_theta, int n_walkers, int n_gauss,
                       unsigned flags, SingleWorkspace* ws, hipStream_t stream);
int wait_single_done(SingleWorkspace* ws, hipStream_t stream, bool poll = true);
void combine_single_draw(const tc_table* t, const SingleWorkspace& ws, int walker, double* ngal,
                         double* xi);
int single_draw_blocks(const tc_table* t);
int launch_single_draw_tables(tc_table* t0, const SingleArgs& prepared, int n_tables,
                              int blocks_per_table, hipStream_t stream);
int launch_interp_coef(const InterpArgs& args, hipStream_t stream);
int launch_occ_from_array(tc_table* t, const double* occupation_device, int64_t n_draws,
                          int64_t ldb, double* nbuf, double* ngal2, hipStream_t stream);
int launch_chi2(const double* xi, int64_t n_draws, int n_r, const double* data,
                const double* precision, double* chi2, hipStream_t stream);

// ---- kernel instances (inst_quad.hip, inst_fused.hip, inst_cross.hip, inst_single.hip) ----
// The device code lives in these translation units; launch.hip fills the argument blocks and
// says which instance it wants.
int launch_occupation(const tc::OccArgs& oa, unsigned flags, int n_gauss, bool grouped,
                      int64_t grid_blocks, hipStream_t stream);
int launch_occ_from_array_kernel(const double* occupation_device, int64_t n_draws, int64_t ldb,
                                 int n_bins, int n_central, const double* n_h,
                                 const int32_t* perm, double* nbuf, double* ngal2, float* nbuf32,
                                 hipStream_t stream);
// predict_fused_kernel<n_gauss, U, assembias, modulate, leauthaud, waves, draws, grouped, defer>
struct FusedInstance {
  int n_gauss = 10;          // 10, or 0 = any number of nodes
  bool assembias = false, modulate = false, leauthaud = false, grouped = false;
  int waves = 8, draws = 64; // 8 x 64, 16 x 64, 8 x 32, 8 x 40 (the latency form)
  int defer = 0;             // SATDEFER (8 x 64, undecorated Zheng07, ten nodes)
};
int launch_fused_instance(const FusedInstance& instance, int device, int n_u, dim3 grid,
                          dim3 block, int lds, hipStream_t stream, hipEvent_t k0, hipEvent_t k1,
                          const tc::FusedArgs& fa);
int launch_cross_instance(bool assembias, bool modulate, bool defer, int device, int rows,
                          dim3 grid, dim3 block, int lds, hipStream_t stream, hipEvent_t k0,
                          hipEvent_t k1, const tc::CrossFusedArgs& ca);
int launch_single_kernel(int blocks, hipStream_t stream, const tc::SingleArgs& sa);
int launch_resident_kernel(int blocks, hipStream_t stream, const tc::SingleArgs& sa);
int launch_ensemble_kernel(int device, int grid, int lds_bytes, hipStream_t stream,
                           const tc::EnsembleArgs& ea);

// ---- staging (table.cpp) --------------------------------------------------------------
int copy_in(PinnedBuffer* stage, void* device, const void* host, size_t bytes,
            hipStream_t stream);
int copy_out(PinnedBuffer* stage, double* ngal, size_t ngal_count, const void* d_ngal,
             double* xi, size_t xi_count, const void* d_xi, hipStream_t stream);

}  // namespace host
}  // namespace tc
