// C ABI of libtabcorr_hip.so (declared in include/tabcorr_amd.h).
#include <hip/hip_runtime.h>

#include <algorithm>
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
#include "fastmath.h"
#include "hostmath.h"
#include "kernels.hip.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* format, ...) {
  char buffer[1024];
  va_list args;
  va_start(args, format);
  vsnprintf(buffer, sizeof(buffer), format, args);
  va_end(args);
  g_last_error = buffer;
  return code;
}

#define TC_HIP(call)                                                          \
  do {                                                                        \
    hipError_t tc_hip_status = (call);                                        \
    if (tc_hip_status != hipSuccess)                                          \
      return fail(TC_ERR_HIP, "%s failed: %s (%s:%d)", #call,                 \
                  hipGetErrorString(tc_hip_status), __FILE__, __LINE__);      \
  } while (0)

#define TC_CHECK(condition, ...)                                              \
  do {                                                                        \
    if (!(condition)) return fail(TC_ERR_INVALID, __VA_ARGS__);               \
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
// Larger transfers go directly (TC_STAGE_LIMIT_MB overrides).
size_t stage_limit() {
  static const size_t limit = [] {
    const char* value = getenv("TC_STAGE_LIMIT_MB");
    return (size_t)(value && *value ? atoi(value) : 1) << 20;
  }();
  return limit;
}

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
};

int env_int(const char* name, int fallback) {
  const char* value = getenv(name);
  if (value == nullptr || *value == 0) return fallback;
  return atoi(value);
}

}  // namespace

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
  void* d_math_table = nullptr;  // fastmath.h tables
  void* d_pos_ij = nullptr;      // float32 variant: packed bin pairs per position
  std::map<int, Quadrature> quadrature;
  std::map<std::pair<int, int>, std::unique_ptr<DeviceChunking>> chunkings;
  std::map<int64_t, DeviceChunking*> choices;   // decomposition chosen per tile count

  // Two independent "lanes" (stream + workspaces).  Consecutive device-pointer
  // predict calls alternate between them, so that the occupation kernel of batch k + 1
  // overlaps the contraction of batch k (both are FP64-issue bound and the contraction
  // leaves issue slots free at its ramp-down); results are still produced in call order
  // (the finalisation kernels are chained by events).  Host-buffer calls use lane 0.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t finished = nullptr;   // recorded after the lane's last finalisation
    DeviceBuffer nbuf, ngal2, partial;
    int ngal_parts = 1;              // partial sums the occupation step left in ngal2
  };
  static constexpr int kMaxLanes = 4;
  Lane lanes[kMaxLanes];
  int n_lanes = 2;
  int prev = -1;                     // lane of the previous finalisation
  int cur = 0;                       // lane of the current / last predict call
  int force_lane = -1;               // host-buffer entry points pin lane 0
  uint64_t device_calls = 0;
  DeviceBuffer theta, out_ngal, out_xi, occupation, trace, wave_trace;
  size_t wave_trace_count = 0;
  PinnedBuffer h_in, h_out;
  size_t trace_blocks = 0;

  // measurement
  bool profile_kernels = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kernel_events;
  size_t kernel_events_used = 0;
  int last_workgroups = 0, last_waves = 0, last_splits = 0, last_lds = 0;
};

namespace {

int get_quadrature(tc_table* t, int n_gauss, Quadrature** out) {
  auto it = t->quadrature.find(n_gauss);
  if (it != t->quadrature.end()) {
    *out = &it->second;
    return TC_OK;
  }
  // tabcorr/tabcorr.py:543-549, 568-578: nodes, node masses and the weights
  // w_k M^(d + 1) / sum_k w_k M^(d + 1), normalised here once per table.  The
  // power is taken relative to the first node so that M^11 cannot overflow.
  std::vector<double> x, w;
  tc::gauss_legendre(n_gauss, x, w);
  const int g = t->n_bins;
  std::vector<double> log_m((size_t)g * n_gauss), m((size_t)g * n_gauss),
      weight((size_t)g * n_gauss);
  for (int i = 0; i < g; ++i) {
    const double d_log = t->log_max[i] - t->log_min[i];
    const double exponent = t->legacy ? 0.0 : t->dist_index[i] + 1.0;
    long double norm = 0.0L;
    std::vector<long double> raw(n_gauss);
    for (int k = 0; k < n_gauss; ++k) {
      const double mass = std::pow(10.0, t->log_min[i] + d_log * x[k]);
      m[(size_t)i * n_gauss + k] = mass;
      log_m[(size_t)i * n_gauss + k] = std::log10(mass);
      const double m_ref = m[(size_t)i * n_gauss];
      raw[k] = (long double)w[k] *
               powl((long double)mass / (long double)m_ref, (long double)exponent);
      norm += raw[k];
    }
    for (int k = 0; k < n_gauss; ++k)
      weight[(size_t)i * n_gauss + k] = (double)(raw[k] / norm);
  }
  Quadrature q;
  q.n_gauss = n_gauss;
  int status = upload(log_m, &q.log_m);
  if (status == TC_OK) status = upload(m, &q.m);
  if (status == TC_OK) status = upload(weight, &q.weight);
  if (status != TC_OK) return status;
  t->quadrature[n_gauss] = q;
  *out = &t->quadrature[n_gauss];
  return TC_OK;
}

int get_chunking(tc_table* t, int n_chunks, int waves, DeviceChunking** out) {
  auto key = std::make_pair(n_chunks, waves);
  auto it = t->chunkings.find(key);
  if (it != t->chunkings.end()) {
    *out = it->second.get();
    return TC_OK;
  }
  std::unique_ptr<DeviceChunking> c(new DeviceChunking);
  tc::build_chunking(t->plan, n_chunks, waves, c->host);
  int status = upload(c->host.chunks, &c->chunks);
  if (status == TC_OK) status = upload(c->host.groups, &c->groups);
  if (status != TC_OK) return status;
  *out = c.get();
  t->chunkings[key] = std::move(c);
  return TC_OK;
}

// Largest dynamic LDS allocation a workgroup may ask for (160 KiB per CU).
constexpr int kMaxLdsBytes = 160 * 1024;

int lds_bytes_for(const tc::Chunking& chunking, int rt, int elem = 8) {
  int span = 1;
  while (span < chunking.waves_per_group) span <<= 1;
  return std::max(chunking.max_rows, (span / 2) * rt) * 64 * elem;
}

// Workgroups of this kernel that fit on one CU: LDS (160 KiB) and wave slots (the
// kernel needs ~70 VGPRs: 7 waves per SIMD).
int blocks_per_cu(int lds_bytes, int waves) {
  const int by_lds = kMaxLdsBytes / std::max(lds_bytes, 1);
  const int by_waves = 28 / waves;
  return std::max(1, std::min(std::min(by_lds, by_waves), 8));
}

// Pick the decomposition for a batch.  Draw tiles alone rarely fill the chip (10^4
// draws are 157 tiles), so the table is additionally cut into groups of `waves` chunks.
// Candidates (4 or 8 waves per workgroup, 1..32 groups) are ranked by a small cost model
// of the busiest CU -- workgroups per CU x waves x (entries per wave + fixed overhead),
// penalised when fewer than four waves per SIMD are resident or when the workgroups need
// several scheduling rounds -- calibrated on per-workgroup timelines (tools/trace.py):
// the main loop issues one FP64 VALU instruction per 4 cycles per SIMD as long as >= 4
// waves per SIMD are resident, and idle time comes from uneven workgroup counts per CU.
int choose_chunking(tc_table* t, int64_t n_draws, int n_comp_out,
                    DeviceChunking** out, int* lds_bytes) {
  (void)n_comp_out;
  const int64_t n_tiles = (n_draws + 63) / 64 * t->n_rtiles;
  const int elem = t->compute_dtype == TC_DTYPE_F32 ? 4 : 8;
  const int forced_groups = env_int("TC_NGROUPS", 0);
  const int forced_waves = env_int("TC_NWAVES", 0);
  if (forced_groups == 0 && forced_waves == 0) {
    auto cached = t->choices.find(n_tiles);
    if (cached != t->choices.end()) {
      *out = cached->second;
      *lds_bytes = lds_bytes_for((*out)->host, t->rt, elem);
      return TC_OK;
    }
  }
  const int64_t min_entries = env_int("TC_MIN_CHUNK_ENTRIES", 32);
  const double overhead_entries = 24.0;
  const int n_cus = 256;
  double best_cost = 0.0;
  int best_chunks = 0, best_waves = 0;
  tc::Chunking trial;
  for (int waves : {4, 5, 6, 7, 8}) {
    if (forced_waves > 0 && waves != forced_waves) continue;
    if (forced_waves == 0 && waves != 4 && waves != 8) continue;
    if (t->compute_dtype == TC_DTYPE_F32 && waves == 4) continue;
    int last_groups = -1;
    for (int groups = 1; groups <= 32; ++groups) {
      if (forced_groups > 0 && groups != forced_groups) continue;
      int64_t n_chunks = (int64_t)groups * waves;
      n_chunks = std::min<int64_t>(
          n_chunks, std::max<int64_t>(1, t->plan.n_entries / min_entries));
      const int use_waves = (int)std::min<int64_t>(waves, n_chunks);
      tc::build_chunking(t->plan, (int)n_chunks, use_waves, trial);
      const int actual_groups = (int)trial.groups.size();
      if (actual_groups == last_groups && forced_groups == 0) continue;
      last_groups = actual_groups;
      const int bytes = lds_bytes_for(trial, t->rt, elem);
      if (bytes > kMaxLdsBytes) continue;
      int longest = 1;
      for (const tc::Chunk& chunk : trial.chunks)
        longest = std::max(longest, chunk.q_end - chunk.q_begin);
      const int fit = blocks_per_cu(bytes + 1024, use_waves);
      const double blocks = (double)n_tiles * actual_groups;
      const double per_cu = std::ceil(blocks / n_cus);
      const double resident = std::min<double>(per_cu, fit) * use_waves / 4.0;
      const double rounds = std::ceil(per_cu / fit);
      double cost = per_cu * use_waves * (longest + overhead_entries);
      if (resident < 4.0) cost *= 4.0 / resident;
      cost *= 1.0 + 0.1 * (rounds - 1.0);
      if (best_chunks == 0 || cost < best_cost) {
        best_cost = cost;
        best_chunks = (int)n_chunks;
        best_waves = use_waves;
      }
    }
  }
  if (best_chunks == 0)
    return fail(TC_ERR_UNSUPPORTED,
                "table with %d bins needs more than %d bytes of LDS per workgroup",
                t->n_bins, kMaxLdsBytes);
  DeviceChunking* c = nullptr;
  int status = get_chunking(t, best_chunks, best_waves, &c);
  if (status != TC_OK) return status;
  if (forced_groups == 0 && forced_waves == 0) t->choices[n_tiles] = c;
  *out = c;
  *lds_bytes = std::max(lds_bytes_for(c->host, t->rt, elem), env_int("TC_LDS_MIN", 0));
  return TC_OK;
}

#define TC_RT_CASES                                                           \
  TC_CASE(4) TC_CASE(8) TC_CASE(12) TC_CASE(16) TC_CASE(20) TC_CASE(24)       \
  TC_CASE(28) TC_CASE(32)

int launch_contract_rt(int rt, dim3 grid, dim3 block, int lds, hipStream_t stream,
                       const tc::ContractArgs& args) {
  switch (rt) {
#define TC_CASE(N)                                                            \
  case N:                                                                     \
    if (args.n_tables > 0)                                                    \
      hipLaunchKernelGGL((tc::contract_kernel<N, true>), grid, block, lds,    \
                         stream, args);                                       \
    else                                                                      \
      hipLaunchKernelGGL((tc::contract_kernel<N, false>), grid, block, lds,   \
                         stream, args);                                       \
    break;
    TC_RT_CASES
#undef TC_CASE
    default:
      return fail(TC_ERR_UNSUPPORTED, "no kernel for r tile %d", rt);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int set_lds_limit_rt(int rt, int lds) {
  switch (rt) {
#define TC_CASE(N)                                                            \
  case N:                                                                     \
    TC_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&tc::contract_kernel<N, false>),        \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    TC_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&tc::contract_kernel<N, true>),         \
        hipFuncAttributeMaxDynamicSharedMemorySize, lds));                    \
    break;
    TC_RT_CASES
#undef TC_CASE
    default:
      break;
  }
  return TC_OK;
}

// Contraction + finalisation of draws whose densities are already in nbuf / ngal2.
int run_contraction(tc_table* t, int64_t n_draws, int64_t ldb, unsigned flags,
                    double* ngal_device, double* xi_device) {
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  DeviceChunking* c = nullptr;
  int lds = 0;
  int status = choose_chunking(t, n_draws, n_comp, &c, &lds);
  if (status != TC_OK) return status;
  tc_table::Lane& lane = t->lanes[t->cur];
  hipStream_t stream = lane.stream;
  const int n_groups = (int)c->host.groups.size();
  const int r_stride = t->rt * t->n_rtiles;
  status = lane.partial.reserve(
      (size_t)n_groups * r_stride * ldb * sizeof(double), stream);
  if (status != TC_OK) return status;

  tc::ContractArgs ca;
  ca.nbuf = (const double*)lane.nbuf.ptr;
  ca.ldb = ldb;
  ca.table = t->d_table;
  ca.n_positions = t->plan.n_positions;
  ca.chunks = (const tc::Chunk*)c->chunks;
  ca.groups = (const tc::Group*)c->groups;
  ca.mode = t->mode;
  ca.n_central = t->plan.n_central;
  ca.r_stride = r_stride;
  ca.trace = nullptr;
  ca.wave_trace = nullptr;
  ca.pos_ij = nullptr;
  if (env_int("TC_TRACE", 0)) {
    t->trace_blocks = (size_t)(ldb / 64) * n_groups * t->n_rtiles;
    status = t->trace.reserve(t->trace_blocks * 6 * sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    ca.trace = (unsigned long long*)t->trace.ptr;
    t->wave_trace_count = t->trace_blocks * c->host.waves_per_group;
    status = t->wave_trace.reserve(t->wave_trace_count * 6 * sizeof(unsigned long long),
                                   stream);
    if (status != TC_OK) return status;
    TC_HIP(hipMemsetAsync(t->wave_trace.ptr, 0,
                          t->wave_trace_count * 6 * sizeof(unsigned long long), stream));
    ca.wave_trace = (unsigned long long*)t->wave_trace.ptr;
  }
  ca.n_tables = 0;
  ca.k_splits = 1;
  ca.tables = nullptr;
  ca.nbufs = nullptr;
  ca.table_class = nullptr;
  ca.coef = nullptr;
  ca.partial = (double*)lane.partial.ptr;

  const int n_tiles = (int)(ldb / 64);
  ca.n_tiles = n_tiles;
  ca.n_slabs = n_groups;
  dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * n_groups), 1, (unsigned)t->n_rtiles);
  dim3 block(64 * c->host.waves_per_group);
  if (lds > 64 * 1024) {
    status = set_lds_limit_rt(t->rt, lds);
    if (status != TC_OK) return status;
  }
  hipEvent_t k0 = nullptr, k1 = nullptr;
  if (t->profile_kernels) {
    if (t->kernel_events_used == t->kernel_events.size()) {
      hipEvent_t e0, e1;
      TC_HIP(hipEventCreate(&e0));
      TC_HIP(hipEventCreate(&e1));
      t->kernel_events.emplace_back(e0, e1);
    }
    k0 = t->kernel_events[t->kernel_events_used].first;
    k1 = t->kernel_events[t->kernel_events_used].second;
    ++t->kernel_events_used;
    TC_HIP(hipEventRecord(k0, stream));
  }
  if (t->compute_dtype == TC_DTYPE_F32) {
    ca.pos_ij = (const int32_t*)t->d_pos_ij;
    if (lds > 64 * 1024)
      TC_HIP(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&tc::contract_f32_kernel),
          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(tc::contract_f32_kernel, grid, block, lds, stream, ca);
    TC_HIP(hipGetLastError());
  } else {
    status = launch_contract_rt(t->rt, grid, block, lds, stream, ca);
    if (status != TC_OK) return status;
  }
  if (t->profile_kernels) TC_HIP(hipEventRecord(k1, stream));
  t->last_workgroups = n_tiles * n_groups * t->n_rtiles;
  t->last_waves = c->host.waves_per_group;
  t->last_splits = n_groups;
  t->last_lds = lds;

  tc::FinalizeArgs fa;
  fa.partial = (const double*)lane.partial.ptr;
  fa.groups = (const tc::Group*)c->groups;
  fa.ngal_part = (const double*)lane.ngal2.ptr;
  fa.n_ngal_parts = lane.ngal_parts;
  fa.n_groups = n_groups;
  fa.k_splits = 1;
  fa.n_comp = n_comp;
  fa.r_stride = r_stride;
  fa.n_r = t->n_r;
  fa.mode = t->mode;
  fa.ldb = ldb;
  fa.n_draws = n_draws;
  fa.ngal = ngal_device;
  fa.xi = xi_device;
  // results appear in call order: wait for the previous call's finalisation
  // (host-buffer calls synchronise before returning and need no chaining)
  if (t->prev >= 0 && t->prev != t->cur)
    TC_HIP(hipStreamWaitEvent(stream, t->lanes[t->prev].finished, 0));
  hipLaunchKernelGGL(tc::finalize_kernel, dim3((unsigned)(ldb / 64)),
                     dim3(env_int("TC_FINALIZE_THREADS", 256)), 0, stream, fa);
  TC_HIP(hipGetLastError());
  if (t->force_lane >= 0) {
    t->prev = -1;
  } else {
    TC_HIP(hipEventRecord(lane.finished, stream));
    t->prev = t->cur;
  }
  return TC_OK;
}

int run_occupation(tc_table* t, const double* theta_device, int n_theta,
                   int64_t n_draws, int64_t ldb, int n_gauss, unsigned flags,
                   double* occupation_device, DeviceBuffer* nbuf = nullptr,
                   DeviceBuffer* ngal2 = nullptr, hipStream_t stream = nullptr,
                   int* ngal_parts = nullptr) {
  tc_table::Lane& lane = t->lanes[t->cur];
  if (nbuf == nullptr) nbuf = &lane.nbuf;
  if (ngal2 == nullptr) ngal2 = &lane.ngal2;
  if (stream == nullptr) stream = lane.stream;
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  // enough blocks to fill the chip: split the bins when there are few draw tiles
  const int64_t n_tiles = ldb / 64;
  int splits = (int)std::min<int64_t>(
      (t->n_bins + tc::kOccWaves - 1) / tc::kOccWaves,
      std::max<int64_t>(1, env_int("TC_OCC_BLOCKS", 2048) / n_tiles));
  splits = std::max(1, splits);
  status = nbuf->reserve((size_t)t->n_bins * ldb * sizeof(double), stream);
  if (status == TC_OK)
    status = ngal2->reserve((size_t)splits * 2 * ldb * sizeof(double), stream);
  if (status != TC_OK) return status;
  if (ngal_parts != nullptr) *ngal_parts = splits; else lane.ngal_parts = splits;
  tc::OccArgs oa;
  oa.theta = theta_device;
  oa.n_theta = n_theta;
  oa.n_draws = n_draws;
  oa.ldb = ldb;
  oa.n_bins = t->n_bins;
  oa.n_central = t->plan.n_central;
  oa.n_gauss = n_gauss;
  oa.flags = flags;
  oa.split = 0.5;
  oa.log_m = (const double*)q->log_m;
  oa.m = (const double*)q->m;
  oa.weight = (const double*)q->weight;
  oa.n_h = (const double*)t->d_n_h;
  oa.percentile = (const double*)t->d_percentile;
  oa.perm = (const int32_t*)t->d_perm;
  oa.math_table = (const double*)t->d_math_table;
  oa.nbuf = (double*)nbuf->ptr;
  oa.ngal = (double*)ngal2->ptr;
  oa.occupation = occupation_device;
  {
    const dim3 grid((unsigned)(ldb / 64), (unsigned)splits), block(tc::kOccWaves * 64);
    const bool assembias = (flags & TC_FLAG_ASSEMBIAS) != 0;
    if (n_gauss == 10 && !assembias)
      hipLaunchKernelGGL((tc::occ_zheng07_kernel<10, false>), grid, block, 0, stream, oa);
    else if (n_gauss == 10)
      hipLaunchKernelGGL((tc::occ_zheng07_kernel<10, true>), grid, block, 0, stream, oa);
    else if (!assembias)
      hipLaunchKernelGGL((tc::occ_zheng07_kernel<0, false>), grid, block, 0, stream, oa);
    else
      hipLaunchKernelGGL((tc::occ_zheng07_kernel<0, true>), grid, block, 0, stream, oa);
  }
  TC_HIP(hipGetLastError());
  return TC_OK;
}

int check_predict_args(const tc_table* t, const void* theta, int n_theta,
                       int64_t n_draws, int n_gauss, unsigned flags) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_CHECK(n_draws >= 0, "n_draws must be non-negative");
  TC_CHECK(n_draws == 0 || theta != nullptr, "theta is NULL");
  TC_CHECK(n_gauss >= 1 && n_gauss <= 4096, "n_gauss_prim must be in [1, 4096]");
  const int need = (flags & TC_FLAG_ASSEMBIAS) ? 7 : 5;
  TC_CHECK(n_theta == need, "theta must have %d columns, got %d", need, n_theta);
  return TC_OK;
}

// Draws are processed in slabs so that the workspaces stay bounded.
constexpr int64_t kMaxSlab = 1 << 18;

}  // namespace

extern "C" {

const char* tc_last_error(void) { return g_last_error.c_str(); }

int tc_device_count(int* count) {
  TC_CHECK(count != nullptr, "count is NULL");
  *count = 0;
  hipError_t status = hipGetDeviceCount(count);
  if (status != hipSuccess) {
    *count = 0;
    return fail(TC_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(status));
  }
  return TC_OK;
}

int tc_set_device(int device) {
  TC_HIP(hipSetDevice(device));
  return TC_OK;
}

int tc_get_device(int* device) {
  TC_CHECK(device != nullptr, "device is NULL");
  TC_HIP(hipGetDevice(device));
  return TC_OK;
}

int tc_runtime_version(int* version) {
  TC_CHECK(version != nullptr, "version is NULL");
  TC_HIP(hipRuntimeGetVersion(version));
  return TC_OK;
}

int tc_device_name(char* buffer, size_t size) {
  TC_CHECK(buffer != nullptr && size > 0, "buffer is NULL");
  int device = 0;
  TC_HIP(hipGetDevice(&device));
  hipDeviceProp_t prop;
  TC_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(buffer, size, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
           prop.multiProcessorCount);
  return TC_OK;
}

int tc_device_synchronize(void) {
  TC_HIP(hipDeviceSynchronize());
  return TC_OK;
}

int tc_device_malloc(void** ptr, size_t bytes) {
  TC_CHECK(ptr != nullptr, "ptr is NULL");
  TC_HIP(hipMalloc(ptr, std::max<size_t>(bytes, 1)));
  return TC_OK;
}

int tc_device_free(void* ptr) {
  if (ptr != nullptr) TC_HIP(hipFree(ptr));
  return TC_OK;
}

int tc_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  TC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return TC_OK;
}

int tc_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  TC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return TC_OK;
}

int tc_gauss_legendre(int n, double* x, double* w) {
  TC_CHECK(n >= 1 && x != nullptr && w != nullptr, "invalid arguments");
  std::vector<double> xs, ws;
  tc::gauss_legendre(n, xs, ws);
  std::copy(xs.begin(), xs.end(), x);
  std::copy(ws.begin(), ws.end(), w);
  return TC_OK;
}

int tc_debug_fastmath(int kind, int64_t n, const double* x, double* y) {
  TC_CHECK(kind >= 0 && kind <= 2 && n >= 0 && x && y, "invalid arguments");
  static std::vector<double> table;
  if (table.empty()) {
    table.resize(tc::fm::kTableDoubles);
    tc::fm::build_tables(table.data());
  }
  for (int64_t i = 0; i < n; ++i) {
    if (kind == 0) y[i] = tc::fm::erf_fast(table.data(), x[i]);
    if (kind == 1) y[i] = tc::fm::log_fast(table.data(), x[i]);
    if (kind == 2) y[i] = tc::fm::exp_fast(table.data(), x[i]);
  }
  return TC_OK;
}

int tc_pair_indices(int n_bins, int32_t* index_1, int32_t* index_2,
                    int32_t* prefactor) {
  TC_CHECK(n_bins >= 0 && index_1 && index_2 && prefactor, "invalid arguments");
  int64_t p = 0;
  for (int i = 0; i < n_bins; ++i) {
    for (int j = 0; j <= i; ++j, ++p) {
      index_1[p] = i;
      index_2[p] = j;
      prefactor[p] = i == j ? 1 : 2;
    }
  }
  return TC_OK;
}

int tc_spline_interpolation_matrix(int n, const double* xp, double* a) {
  TC_CHECK(xp != nullptr && a != nullptr, "invalid arguments");
  // tabcorr/interpolator.py:239-241
  TC_CHECK(n >= 4, "Cannot perform spline interpolation with less than 4 values.");
  std::vector<double> out;
  TC_CHECK(tc::spline_interpolation_matrix(n, xp, out),
           "singular spline system (repeated abscissae?)");
  std::copy(out.begin(), out.end(), a);
  return TC_OK;
}

int tc_plan_debug(int mode, int n_bins, const uint8_t* is_central, int n_chunks,
                  int64_t* n_entries, int32_t* entry_pair, int32_t* entry_chunk,
                  int32_t* entry_class) {
  TC_CHECK(mode == TC_MODE_AUTO || mode == TC_MODE_CROSS, "invalid mode");
  TC_CHECK(n_bins >= 1 && is_central && n_entries, "invalid arguments");
  tc::Plan plan;
  tc::build_plan(mode, n_bins, is_central, 4, env_int("TC_ROW_BUDGET", 56), plan);
  tc::Chunking chunking;
  tc::build_chunking(plan, n_chunks, 8, chunking);
  *n_entries = plan.n_entries;
  if (entry_pair == nullptr) return TC_OK;
  // Walk every chunk exactly as the kernel does and record what it visits.
  std::vector<int> seen((size_t)plan.n_positions, 0);
  int64_t e = 0;
  for (size_t c = 0; c < chunking.chunks.size(); ++c) {
    const tc::Chunk& chunk = chunking.chunks[c];
    if ((chunk.q_begin % plan.block) != 0 || (chunk.q_end % plan.block) != 0)
      return fail(TC_ERR_INVALID, "chunk %zu is not block aligned", c);
    int i = chunk.i0, j = chunk.j0, remaining = chunk.n_real;
    for (int q = chunk.q_begin; q < chunk.q_end; ++q) {
      if (seen[q]++) return fail(TC_ERR_INVALID, "position %d covered twice", q);
      if (plan.column[q] >= 0) {
        const int64_t column =
            mode == TC_MODE_AUTO ? tc::packed_index(plan.perm[i], plan.perm[j])
                                 : plan.perm[j];
        if (column != plan.column[q] || q >= chunk.q_begin + chunk.n_real)
          return fail(TC_ERR_INVALID, "walk mismatch at position %d", q);
        if (e >= plan.n_entries) return fail(TC_ERR_INVALID, "too many entries");
        entry_pair[e] = (int32_t)column;
        entry_chunk[e] = (int32_t)c;
        entry_class[e] = chunk.component;
        ++e;
      }
      if (--remaining > 0) tc::advance_pair(chunk.j_lo, chunk.j_last, i, j);
    }
  }
  for (int64_t q = 0; q < plan.n_positions; ++q)
    if (!seen[q]) return fail(TC_ERR_INVALID, "position %lld not covered", (long long)q);
  if (e != plan.n_entries) return fail(TC_ERR_INVALID, "entries missing");
  return TC_OK;
}

int tc_table_create(int mode, int n_bins, int n_r, int64_t n_pairs,
                    const void* tpcf_matrix, int matrix_dtype, const double* n_h,
                    const double* log_min, const double* log_max,
                    const double* percentile, const double* dist_index,
                    const uint8_t* is_central, int compute_dtype, tc_table** out) {
  TC_CHECK(out != nullptr, "table output pointer is NULL");
  *out = nullptr;
  TC_CHECK(mode == TC_MODE_AUTO || mode == TC_MODE_CROSS, "invalid mode %d", mode);
  TC_CHECK(n_bins >= 1 && n_bins < (1 << 20), "invalid number of bins %d", n_bins);
  TC_CHECK(n_r >= 1, "invalid number of correlation function bins %d", n_r);
  const int64_t expect =
      mode == TC_MODE_AUTO ? (int64_t)n_bins * (n_bins + 1) / 2 : n_bins;
  TC_CHECK(n_pairs == expect,
           "tpcf_matrix has %lld columns but %d bins in mode '%s' need %lld",
           (long long)n_pairs, n_bins, mode == TC_MODE_AUTO ? "auto" : "cross",
           (long long)expect);
  TC_CHECK(n_pairs < (1LL << 31), "too many pair columns");
  TC_CHECK(tpcf_matrix && n_h && log_min && log_max && percentile && is_central,
           "NULL input array");
  TC_CHECK(matrix_dtype == TC_DTYPE_F64 || matrix_dtype == TC_DTYPE_F32,
           "invalid matrix dtype");
  TC_CHECK(compute_dtype == TC_DTYPE_F64 || compute_dtype == TC_DTYPE_F32,
           "invalid compute dtype");

  std::unique_ptr<tc_table> t(new tc_table);
  TC_HIP(hipGetDevice(&t->device));
  t->mode = mode;
  t->n_bins = n_bins;
  t->n_r = n_r;
  t->n_pairs = n_pairs;
  t->compute_dtype = compute_dtype;
  t->legacy = dist_index == nullptr;
  TC_CHECK(compute_dtype == TC_DTYPE_F64 || true, "unreachable");
  // r tiling: at most 32 accumulators per lane, a multiple of 4 so that a block
  // of at most 4 entries fills whole 128-byte lines (float64); tiles of exactly 32 r
  // values and blocks of 8 entries for the float32 MFMA kernel.
  const int max_rt = 32;
  t->n_rtiles = (n_r + max_rt - 1) / max_rt;
  if (compute_dtype == TC_DTYPE_F32) {
    t->rt = tc::kF32Tile;
    tc::build_plan(mode, n_bins, is_central, tc::kF32Block,
                   env_int("TC_ROW_BUDGET_F32", 128), t->plan);
  } else {
    int rt = (n_r + t->n_rtiles - 1) / t->n_rtiles;
    t->rt = (rt + 3) / 4 * 4;
    // LDS rows a workgroup may stage: half the bins (one triangle or one column block
    // of the cen-sat rectangle) plus a few rows, within 28..66 KB
    const int budget = std::max(56, std::min(128, n_bins / 2 + 6));
    tc::build_plan(mode, n_bins, is_central, tc::block_entries(t->rt),
                   env_int("TC_ROW_BUDGET", budget), t->plan);
  }

  for (int g = 0; g < n_bins; ++g) {
    const int src = t->plan.perm[g];
    t->n_h.push_back(n_h[src]);
    t->log_min.push_back(log_min[src]);
    t->log_max.push_back(log_max[src]);
    t->percentile.push_back(percentile[src]);
    t->dist_index.push_back(dist_index ? dist_index[src] : -1.0);
  }

  const int rt = t->rt;
  const int64_t n_positions = t->plan.n_positions;
  const size_t count = (size_t)t->n_rtiles * n_positions * rt;
  auto source = [&](int r, int64_t column) {
    return matrix_dtype == TC_DTYPE_F64
               ? ((const double*)tpcf_matrix)[(size_t)r * n_pairs + column]
               : (double)((const float*)tpcf_matrix)[(size_t)r * n_pairs + column];
  };
  std::vector<double> tmp64;
  std::vector<float> tmp32;
  std::vector<int32_t> pos_ij;
  if (compute_dtype == TC_DTYPE_F64) {
    // Re-laid-out matrix: [r tile][position][r in tile] with the pair prefactor
    // (tabcorr.py:638-642) folded in (a multiplication by 2 is exact) and zero
    // rows at the padding positions.
    t->table_bytes = count * sizeof(double);
    tmp64.assign(count, 0.0);
    for (int r = 0; r < n_r; ++r) {
      const int tile = r / rt, rr = r % rt;
      for (int64_t q = 0; q < n_positions; ++q) {
        const int64_t column = t->plan.column[q];
        if (column < 0) continue;
        tmp64[((size_t)tile * n_positions + q) * rt + rr] =
            source(r, column) * t->plan.prefactor[q];
      }
    }
  } else {
    // float32 MFMA layout: [r tile][block of 8 positions][k][r][k-step] (kernels.hip.h)
    t->table_bytes = count * sizeof(float);
    tmp32.assign(count, 0.0f);
    for (int r = 0; r < n_r; ++r) {
      const int tile = r / rt, rr = r % rt;
      for (int64_t q = 0; q < n_positions; ++q) {
        const int64_t column = t->plan.column[q];
        if (column < 0) continue;
        const int64_t block = q / 8;
        const int p = (int)(q % 8) / 2, k = (int)(q % 2);
        tmp32[((size_t)tile * n_positions + block * 8) * rt + (k * 32 + rr) * 4 + p] =
            (float)(source(r, column) * t->plan.prefactor[q]);
      }
    }
    TC_CHECK(n_bins < 65535, "too many bins for the float32 variant");
    pos_ij.assign((size_t)n_positions, 0);
    for (int64_t q = 0; q < n_positions; ++q) {
      const int64_t block = q / 8;
      const int p = (int)(q % 8) / 2, k = (int)(q % 2);
      const int i = t->plan.pos_i[q] < 0 ? 0 : t->plan.pos_i[q];
      pos_ij[(size_t)block * 8 + k * 4 + p] = (i << 16) | t->plan.pos_j[q];
    }
  }
  for (tc_table::Lane& lane : t->lanes) {
    TC_HIP(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
    TC_HIP(hipEventCreateWithFlags(&lane.finished, hipEventDisableTiming));
  }
  t->stream = t->lanes[0].stream;
  t->n_lanes = std::max(1, std::min(env_int("TC_LANES", 3), (int)tc_table::kMaxLanes));
  TC_HIP(hipEventCreate(&t->ev_begin));
  TC_HIP(hipEventCreate(&t->ev_end));
  int status = compute_dtype == TC_DTYPE_F64 ? upload(tmp64, &t->d_table)
                                             : upload(tmp32, &t->d_table);
  if (status == TC_OK && compute_dtype == TC_DTYPE_F32)
    status = upload(pos_ij, &t->d_pos_ij);
  if (status == TC_OK) status = upload(t->n_h, &t->d_n_h);
  if (status == TC_OK) status = upload(t->percentile, &t->d_percentile);
  if (status == TC_OK) status = upload(t->plan.perm, &t->d_perm);
  if (status == TC_OK) {
    std::vector<double> math_table(tc::fm::kTableDoubles);
    tc::fm::build_tables(math_table.data());
    status = upload(math_table, &t->d_math_table);
  }
  if (status != TC_OK) {
    tc_table_destroy(t.release());
    return status;
  }
  *out = t.release();
  return TC_OK;
}

int tc_table_destroy(tc_table* t) {
  if (t == nullptr) return TC_OK;
  (void)hipSetDevice(t->device);
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) (void)hipStreamSynchronize(lane.stream);
  for (void* p : {t->d_table, t->d_n_h, t->d_percentile, t->d_perm, t->d_math_table,
                  t->d_pos_ij})
    if (p) (void)hipFree(p);
  for (auto& kv : t->quadrature)
    for (void* p : {kv.second.log_m, kv.second.m, kv.second.weight})
      if (p) (void)hipFree(p);
  for (auto& kv : t->chunkings)
    for (void* p : {kv.second->chunks, kv.second->groups})
      if (p) (void)hipFree(p);
  for (DeviceBuffer* b : {&t->theta, &t->out_ngal, &t->out_xi, &t->occupation,
                          &t->trace, &t->wave_trace})
    b->release();
  for (tc_table::Lane& lane : t->lanes) {
    lane.nbuf.release();
    lane.ngal2.release();
    lane.partial.release();
    if (lane.finished) (void)hipEventDestroy(lane.finished);
  }
  t->h_in.release();
  t->h_out.release();
  for (auto& ev : t->kernel_events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (t->ev_begin) (void)hipEventDestroy(t->ev_begin);
  if (t->ev_end) (void)hipEventDestroy(t->ev_end);
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) (void)hipStreamDestroy(lane.stream);
  delete t;
  return TC_OK;
}

int tc_table_synchronize(tc_table* t) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  for (tc_table::Lane& lane : t->lanes) TC_HIP(hipStreamSynchronize(lane.stream));
  return TC_OK;
}

int tc_table_info(const tc_table* t, int* mode, int* n_bins, int* n_r,
                  int64_t* n_pairs, int* n_components, int64_t* device_bytes) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  if (mode) *mode = t->mode;
  if (n_bins) *n_bins = t->n_bins;
  if (n_r) *n_r = t->n_r;
  if (n_pairs) *n_pairs = t->n_pairs;
  if (n_components) *n_components = t->plan.n_components;
  if (device_bytes) *device_bytes = (int64_t)t->table_bytes;
  return TC_OK;
}

int tc_predict_zheng07_batch_device(tc_table* t, const double* theta_device,
                                    int n_theta, int64_t n_draws, int n_gauss,
                                    unsigned flags, double* ngal_device,
                                    double* xi_device) {
  int status = check_predict_args(t, theta_device, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(ngal_device && xi_device, "output pointer is NULL");
  TC_HIP(hipSetDevice(t->device));
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  if (t->force_lane >= 0)
    t->cur = t->force_lane;
  else
    t->cur = env_int("TC_PIPELINE", 1) ? (int)(t->device_calls++ % t->n_lanes) : 0;
  for (int64_t begin = 0; begin < n_draws; begin += kMaxSlab) {
    const int64_t n = std::min(kMaxSlab, n_draws - begin);
    const int64_t ldb = (n + 63) / 64 * 64;
    status = run_occupation(t, theta_device + begin * n_theta, n_theta, n, ldb,
                            n_gauss, flags, nullptr);
    if (status != TC_OK) return status;
    status = run_contraction(t, n, ldb, flags,
                             ngal_device + begin * (separate ? 2 : 1),
                             xi_device + begin * n_comp * t->n_r);
    if (status != TC_OK) return status;
  }
  return TC_OK;
}

namespace {

// Host -> device copy of a small input through the pinned staging buffer.
int copy_in(PinnedBuffer* stage, void* device, const void* host, size_t bytes,
            hipStream_t stream) {
  if (bytes <= stage_limit() && stage->reserve(bytes) == TC_OK) {
    memcpy(stage->ptr, host, bytes);
    host = stage->ptr;
  }
  TC_HIP(hipMemcpyAsync(device, host, bytes, hipMemcpyHostToDevice, stream));
  return TC_OK;
}

// Device -> host copy of [ngal | xi], synchronising the stream.
int copy_out(PinnedBuffer* stage, double* ngal, size_t ngal_count, const void* d_ngal,
             double* xi, size_t xi_count, const void* d_xi, hipStream_t stream) {
  const size_t bytes = (ngal_count + xi_count) * sizeof(double);
  if (bytes <= stage_limit() && stage->reserve(bytes) == TC_OK) {
    double* h = (double*)stage->ptr;
    TC_HIP(hipMemcpyAsync(h, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost, stream));
    TC_HIP(hipMemcpyAsync(h + ngal_count, d_xi, xi_count * 8, hipMemcpyDeviceToHost,
                          stream));
    TC_HIP(hipStreamSynchronize(stream));
    memcpy(ngal, h, ngal_count * 8);
    memcpy(xi, h + ngal_count, xi_count * 8);
    return TC_OK;
  }
  TC_HIP(hipMemcpyAsync(ngal, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost, stream));
  TC_HIP(hipMemcpyAsync(xi, d_xi, xi_count * 8, hipMemcpyDeviceToHost, stream));
  TC_HIP(hipStreamSynchronize(stream));
  return TC_OK;
}

}  // namespace

int tc_predict_zheng07_batch(tc_table* t, const double* theta, int n_theta,
                             int64_t n_draws, int n_gauss, unsigned flags,
                             double* ngal, double* xi) {
  int status = check_predict_args(t, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(ngal && xi, "output pointer is NULL");
  TC_HIP(hipSetDevice(t->device));
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  const size_t ngal_count = (size_t)n_draws * (separate ? 2 : 1);
  const size_t xi_count = (size_t)n_draws * n_comp * t->n_r;
  status = t->theta.reserve((size_t)n_draws * n_theta * sizeof(double), t->stream);
  if (status == TC_OK) status = t->out_ngal.reserve(ngal_count * 8, t->stream);
  if (status == TC_OK) status = t->out_xi.reserve(xi_count * 8, t->stream);
  if (status != TC_OK) return status;
  status = copy_in(&t->h_in, t->theta.ptr, theta, (size_t)n_draws * n_theta * 8,
                   t->stream);
  if (status != TC_OK) return status;
  t->force_lane = 0;
  status = tc_predict_zheng07_batch_device(
      t, (const double*)t->theta.ptr, n_theta, n_draws, n_gauss, flags,
      (double*)t->out_ngal.ptr, (double*)t->out_xi.ptr);
  t->force_lane = -1;
  if (status != TC_OK) return status;
  return copy_out(&t->h_out, ngal, ngal_count, t->out_ngal.ptr, xi, xi_count,
                  t->out_xi.ptr, t->stream);
}

int tc_chi2_zheng07_batch(tc_table* t, const double* theta, int n_theta,
                          int64_t n_draws, int n_gauss, unsigned flags,
                          const double* data, const double* precision, double* ngal,
                          double* chi2) {
  int status = check_predict_args(t, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  TC_CHECK(!(flags & TC_FLAG_SEPARATE_GAL_TYPE),
           "chi2 is defined for the total correlation function only");
  if (n_draws == 0) return TC_OK;
  TC_CHECK(data && precision && ngal && chi2, "NULL pointer");
  TC_HIP(hipSetDevice(t->device));
  const int n_r = t->n_r;
  const size_t xi_count = (size_t)n_draws * n_r;
  status = t->theta.reserve((size_t)n_draws * n_theta * sizeof(double), t->stream);
  if (status == TC_OK) status = t->out_ngal.reserve((size_t)n_draws * 2 * 8, t->stream);
  if (status == TC_OK) status = t->out_xi.reserve(xi_count * 8, t->stream);
  if (status == TC_OK)
    status = t->occupation.reserve((size_t)(n_r + 1) * n_r * 8, t->stream);
  if (status != TC_OK) return status;
  status = copy_in(&t->h_in, t->theta.ptr, theta, (size_t)n_draws * n_theta * 8,
                   t->stream);
  if (status != TC_OK) return status;
  // data vector and precision matrix behind each other in a scratch buffer
  double* d_data = (double*)t->occupation.ptr;
  double* d_precision = d_data + n_r;
  TC_HIP(hipMemcpyAsync(d_data, data, (size_t)n_r * 8, hipMemcpyHostToDevice, t->stream));
  TC_HIP(hipMemcpyAsync(d_precision, precision, (size_t)n_r * n_r * 8,
                        hipMemcpyHostToDevice, t->stream));
  double* d_ngal = (double*)t->out_ngal.ptr;
  double* d_chi2 = d_ngal + n_draws;
  t->force_lane = 0;
  status = tc_predict_zheng07_batch_device(t, (const double*)t->theta.ptr, n_theta,
                                           n_draws, n_gauss, flags, d_ngal,
                                           (double*)t->out_xi.ptr);
  t->force_lane = -1;
  if (status != TC_OK) return status;
  hipLaunchKernelGGL(tc::chi2_kernel, dim3((unsigned)((n_draws + 255) / 256)), dim3(256),
                     0, t->stream, (const double*)t->out_xi.ptr, n_draws, n_r,
                     (const double*)d_data, (const double*)d_precision, d_chi2);
  TC_HIP(hipGetLastError());
  return copy_out(&t->h_out, ngal, (size_t)n_draws, d_ngal, chi2, (size_t)n_draws,
                  d_chi2, t->stream);
}

int tc_mean_occupation_zheng07_batch(tc_table* t, const double* theta, int n_theta,
                                     int64_t n_draws, int n_gauss, unsigned flags,
                                     double* occupation) {
  int status = check_predict_args(t, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(occupation != nullptr, "output pointer is NULL");
  TC_HIP(hipSetDevice(t->device));
  for (int64_t begin = 0; begin < n_draws; begin += kMaxSlab) {
    const int64_t n = std::min(kMaxSlab, n_draws - begin);
    const int64_t ldb = (n + 63) / 64 * 64;
    const size_t occ_bytes = (size_t)n * t->n_bins * 8;
    status = t->theta.reserve((size_t)n * n_theta * 8, t->stream);
    if (status == TC_OK) status = t->occupation.reserve(occ_bytes, t->stream);
    if (status != TC_OK) return status;
    TC_HIP(hipMemcpyAsync(t->theta.ptr, theta + begin * n_theta,
                          (size_t)n * n_theta * 8, hipMemcpyHostToDevice, t->stream));
    t->cur = 0;
    status = run_occupation(t, (const double*)t->theta.ptr, n_theta, n, ldb, n_gauss,
                            flags, (double*)t->occupation.ptr);
    if (status != TC_OK) return status;
    TC_HIP(hipMemcpyAsync(occupation + begin * t->n_bins, t->occupation.ptr,
                          occ_bytes, hipMemcpyDeviceToHost, t->stream));
    TC_HIP(hipStreamSynchronize(t->stream));
  }
  return TC_OK;
}

int tc_predict_occupation_batch(tc_table* t, const double* occupation,
                                int64_t n_draws, unsigned flags, double* ngal,
                                double* xi) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_CHECK(n_draws >= 0, "n_draws must be non-negative");
  if (n_draws == 0) return TC_OK;
  TC_CHECK(occupation && ngal && xi, "NULL array");
  TC_HIP(hipSetDevice(t->device));
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  for (int64_t begin = 0; begin < n_draws; begin += kMaxSlab) {
    const int64_t n = std::min(kMaxSlab, n_draws - begin);
    const int64_t ldb = (n + 63) / 64 * 64;
    const size_t occ_bytes = (size_t)n * t->n_bins * 8;
    const size_t ngal_count = (size_t)n * (separate ? 2 : 1);
    const size_t xi_count = (size_t)n * n_comp * t->n_r;
    int status = t->occupation.reserve(occ_bytes, t->stream);
    if (status == TC_OK)
      status = t->lanes[0].nbuf.reserve((size_t)t->n_bins * ldb * 8, t->stream);
    if (status == TC_OK)
      status = t->lanes[0].ngal2.reserve(2 * ldb * 8, t->stream);
    if (status == TC_OK) status = t->out_ngal.reserve(ngal_count * 8, t->stream);
    if (status == TC_OK) status = t->out_xi.reserve(xi_count * 8, t->stream);
    if (status != TC_OK) return status;
    status = copy_in(&t->h_in, t->occupation.ptr, occupation + begin * t->n_bins,
                     occ_bytes, t->stream);
    if (status != TC_OK) return status;
    hipLaunchKernelGGL(tc::occ_from_array_kernel,
                       dim3((unsigned)((ldb + 255) / 256)), dim3(256), 0, t->stream,
                       (const double*)t->occupation.ptr, n, ldb, t->n_bins,
                       t->plan.n_central, (const double*)t->d_n_h,
                       (const int32_t*)t->d_perm, (double*)t->lanes[0].nbuf.ptr,
                       (double*)t->lanes[0].ngal2.ptr);
    TC_HIP(hipGetLastError());
    t->cur = 0;
    t->lanes[0].ngal_parts = 1;
    status = run_contraction(t, n, ldb, flags, (double*)t->out_ngal.ptr,
                             (double*)t->out_xi.ptr);
    if (status != TC_OK) return status;
    status = copy_out(&t->h_out, ngal + begin * (separate ? 2 : 1), ngal_count,
                      t->out_ngal.ptr, xi + begin * n_comp * t->n_r, xi_count,
                      t->out_xi.ptr, t->stream);
    if (status != TC_OK) return status;
  }
  return TC_OK;
}

int tc_table_timer_begin(tc_table* t, int profile_kernels) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_HIP(hipSetDevice(t->device));
  t->profile_kernels = profile_kernels != 0;
  t->kernel_events_used = 0;
  TC_HIP(hipEventRecord(t->ev_begin, t->stream));
  return TC_OK;
}

int tc_table_timer_end(tc_table* t, float* elapsed_ms) {
  TC_CHECK(t != nullptr && elapsed_ms != nullptr, "NULL argument");
  for (int l = 1; l < tc_table::kMaxLanes; ++l)
    TC_HIP(hipStreamSynchronize(t->lanes[l].stream));
  TC_HIP(hipEventRecord(t->ev_end, t->stream));
  TC_HIP(hipEventSynchronize(t->ev_end));
  TC_HIP(hipEventElapsedTime(elapsed_ms, t->ev_begin, t->ev_end));
  t->profile_kernels = false;
  return TC_OK;
}

int tc_table_kernel_time(tc_table* t, int* n_launches, float* mean_ms) {
  TC_CHECK(t != nullptr && n_launches && mean_ms, "NULL argument");
  double total = 0.0;
  for (size_t i = 0; i < t->kernel_events_used; ++i) {
    float ms = 0.0f;
    TC_HIP(hipEventSynchronize(t->kernel_events[i].second));
    TC_HIP(hipEventElapsedTime(&ms, t->kernel_events[i].first,
                               t->kernel_events[i].second));
    total += ms;
  }
  *n_launches = (int)t->kernel_events_used;
  *mean_ms = t->kernel_events_used ? (float)(total / t->kernel_events_used) : 0.0f;
  return TC_OK;
}

int tc_debug_trace(tc_table* t, uint64_t* out, int64_t capacity, int64_t* n_blocks) {
  TC_CHECK(t != nullptr && n_blocks != nullptr, "NULL argument");
  *n_blocks = (int64_t)t->trace_blocks;
  if (out == nullptr || t->trace.ptr == nullptr) return TC_OK;
  for (tc_table::Lane& lane : t->lanes) TC_HIP(hipStreamSynchronize(lane.stream));
  const int64_t n = std::min<int64_t>(capacity, (int64_t)t->trace_blocks);
  TC_HIP(hipMemcpy(out, t->trace.ptr, (size_t)n * 6 * sizeof(uint64_t),
                   hipMemcpyDeviceToHost));
  return TC_OK;
}

int tc_debug_wave_trace(tc_table* t, uint64_t* out, int64_t capacity, int64_t* n_waves) {
  TC_CHECK(t != nullptr && n_waves != nullptr, "NULL argument");
  *n_waves = (int64_t)t->wave_trace_count;
  if (out == nullptr || t->wave_trace.ptr == nullptr) return TC_OK;
  for (tc_table::Lane& lane : t->lanes) TC_HIP(hipStreamSynchronize(lane.stream));
  const int64_t n = std::min<int64_t>(capacity, (int64_t)t->wave_trace_count);
  TC_HIP(hipMemcpy(out, t->wave_trace.ptr, (size_t)n * 6 * sizeof(uint64_t),
                   hipMemcpyDeviceToHost));
  return TC_OK;
}

int tc_table_last_launch(const tc_table* t, int* n_workgroups, int* waves,
                         int* n_splits, int* lds_bytes) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  if (n_workgroups) *n_workgroups = t->last_workgroups;
  if (waves) *waves = t->last_waves;
  if (n_splits) *n_splits = t->last_splits;
  if (lds_bytes) *lds_bytes = t->last_lds;
  return TC_OK;
}

}  // extern "C"

// ---- interpolation over a grid of tables ----------------------------------------------

struct tc_interp {
  int device = 0;
  int n_tables = 0;
  int n_dim = 0;
  std::vector<tc_table*> tables;
  std::vector<std::vector<double>> xp;      // per dimension
  std::vector<int32_t> table_node;          // (K, D)
  std::vector<int32_t> table_class;         // (K)
  std::vector<int> class_table;             // representative table of each class
  hipStream_t stream = nullptr;
  void* d_xp = nullptr;
  void* d_a = nullptr;
  void* d_table_node = nullptr;
  void* d_table_class = nullptr;
  void* d_tables = nullptr;                 // (K) device pointers
  void* d_nbufs = nullptr;                  // (V) device pointers
  void* d_ngal_parts = nullptr;             // (V) device pointers
  std::vector<int> axis_offset, a_offset;
  std::vector<DeviceBuffer> nbuf, ngal2;    // per class
  std::vector<void*> nbuf_ptrs, ngal_ptrs;  // last uploaded pointer values
  DeviceBuffer theta, x, coef, partial, out_ngal, out_xi;
  std::map<std::pair<int, int>, std::unique_ptr<DeviceChunking>> chunkings;
};

namespace {

bool same_bins(const tc_table* a, const tc_table* b) {
  return a->n_h == b->n_h && a->log_min == b->log_min && a->log_max == b->log_max &&
         a->percentile == b->percentile && a->dist_index == b->dist_index &&
         a->legacy == b->legacy;
}

int interp_predict_device(tc_interp* it, const double* theta_device, int n_theta,
                          const double* x_device, int64_t n_draws, int n_gauss,
                          unsigned flags, double* ngal_device, double* xi_device) {
  tc_table* t0 = it->tables[0];
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t0->plan.n_components : 1;
  const int n_classes = (int)it->class_table.size();
  const int64_t ldb = (n_draws + 63) / 64 * 64;
  int status = TC_OK;

  // occupations once per class of identical halo tables (interpolator.py:181-184)
  int ngal_parts = 1;
  for (int v = 0; v < n_classes; ++v) {
    status = run_occupation(it->tables[it->class_table[v]], theta_device, n_theta,
                            n_draws, ldb, n_gauss, flags, nullptr, &it->nbuf[v],
                            &it->ngal2[v], it->stream, &ngal_parts);
    if (status != TC_OK) return status;
  }
  bool moved = false;
  for (int v = 0; v < n_classes; ++v) {
    moved = moved || it->nbuf_ptrs[v] != it->nbuf[v].ptr ||
            it->ngal_ptrs[v] != it->ngal2[v].ptr;
    it->nbuf_ptrs[v] = it->nbuf[v].ptr;
    it->ngal_ptrs[v] = it->ngal2[v].ptr;
  }
  if (moved) {
    TC_HIP(hipMemcpyAsync(it->d_nbufs, it->nbuf_ptrs.data(), n_classes * sizeof(void*),
                          hipMemcpyHostToDevice, it->stream));
    TC_HIP(hipMemcpyAsync(it->d_ngal_parts, it->ngal_ptrs.data(),
                          n_classes * sizeof(void*), hipMemcpyHostToDevice, it->stream));
    TC_HIP(hipStreamSynchronize(it->stream));   // the host vectors may change later
  }

  status = it->coef.reserve((size_t)it->n_tables * ldb * sizeof(double), it->stream);
  if (status != TC_OK) return status;
  tc::InterpArgs ia;
  ia.n_dim = it->n_dim;
  ia.n_tables = it->n_tables;
  ia.n_classes = n_classes;
  ia.mode = t0->mode;
  ia.separate = separate ? 1 : 0;
  for (int d = 0; d < it->n_dim; ++d) {
    ia.n_axis[d] = (int)it->xp[d].size();
    ia.axis_offset[d] = it->axis_offset[d];
    ia.a_offset[d] = it->a_offset[d];
  }
  ia.xp = (const double*)it->d_xp;
  ia.a = (const double*)it->d_a;
  ia.table_node = (const int32_t*)it->d_table_node;
  ia.table_class = (const int32_t*)it->d_table_class;
  ia.x = x_device;
  ia.ngal_parts = (const double* const*)it->d_ngal_parts;
  ia.n_ngal_parts = ngal_parts;
  ia.ldb = ldb;
  ia.n_draws = n_draws;
  ia.coef = (double*)it->coef.ptr;
  ia.ngal = ngal_device;
  hipLaunchKernelGGL(tc::interp_coef_kernel, dim3((unsigned)(ldb / 64)), dim3(64), 0,
                     it->stream, ia);
  TC_HIP(hipGetLastError());

  // decomposition: as for one table (choose_chunking), with the tables looped inside
  // the block; the tables are split over blocks only when one pass would leave the chip
  // underfilled
  const int64_t n_tiles = ldb / 64;
  DeviceChunking* c = nullptr;
  int lds = 0;
  {
    // the table handle caches the chunkings; all tables share table 0's plan
    status = choose_chunking(t0, n_draws, n_comp, &c, &lds);
    if (status != TC_OK) return status;
  }
  int k_splits = 1;
  {
    const int64_t blocks = n_tiles * t0->n_rtiles * (int64_t)c->host.groups.size();
    const int64_t want = 256 * (int64_t)blocks_per_cu(lds, c->host.waves_per_group);
    if (blocks * 2 <= want)
      k_splits = (int)std::min<int64_t>(it->n_tables, want / std::max<int64_t>(1, blocks));
    k_splits = std::max(1, env_int("TC_KSPLITS", k_splits));
    k_splits = std::min(k_splits, it->n_tables);
  }
  if (lds > kMaxLdsBytes)
    return fail(TC_ERR_UNSUPPORTED, "table with %d bins needs %d bytes of LDS",
                t0->n_bins, lds);
  const int n_groups = (int)c->host.groups.size();
  const int r_stride = t0->rt * t0->n_rtiles;
  status = it->partial.reserve(
      (size_t)n_groups * k_splits * r_stride * ldb * sizeof(double), it->stream);
  if (status != TC_OK) return status;

  tc::ContractArgs ca;
  ca.nbuf = nullptr;
  ca.ldb = ldb;
  ca.table = nullptr;
  ca.n_positions = t0->plan.n_positions;
  ca.chunks = (const tc::Chunk*)c->chunks;
  ca.groups = (const tc::Group*)c->groups;
  ca.mode = t0->mode;
  ca.n_central = t0->plan.n_central;
  ca.r_stride = r_stride;
  ca.trace = nullptr;
  ca.wave_trace = nullptr;
  ca.pos_ij = nullptr;
  ca.partial = (double*)it->partial.ptr;
  ca.n_tables = it->n_tables;
  ca.k_splits = k_splits;
  ca.tables = (const double* const*)it->d_tables;
  ca.nbufs = (const double* const*)it->d_nbufs;
  ca.table_class = (const int32_t*)it->d_table_class;
  ca.coef = (const double*)it->coef.ptr;
  ca.n_tiles = (int)n_tiles;
  ca.n_slabs = n_groups * k_splits;
  dim3 grid((unsigned)((n_tiles + 7) / 8 * 8 * n_groups * k_splits), 1,
            (unsigned)t0->n_rtiles);
  dim3 block(64 * c->host.waves_per_group);
  if (lds > 64 * 1024) {
    status = set_lds_limit_rt(t0->rt, lds);
    if (status != TC_OK) return status;
  }
  status = launch_contract_rt(t0->rt, grid, block, lds, it->stream, ca);
  if (status != TC_OK) return status;

  tc::FinalizeArgs fa;
  fa.partial = (const double*)it->partial.ptr;
  fa.groups = (const tc::Group*)c->groups;
  fa.ngal_part = nullptr;      // already normalised; ngal written by the coef kernel
  fa.n_ngal_parts = 0;
  fa.n_groups = n_groups;
  fa.k_splits = k_splits;
  fa.n_comp = n_comp;
  fa.r_stride = r_stride;
  fa.n_r = t0->n_r;
  fa.mode = t0->mode;
  fa.ldb = ldb;
  fa.n_draws = n_draws;
  fa.ngal = ngal_device;
  fa.xi = xi_device;
  hipLaunchKernelGGL(tc::finalize_kernel, dim3((unsigned)n_tiles), dim3(256), 0,
                     it->stream, fa);
  TC_HIP(hipGetLastError());
  return TC_OK;
}

}  // namespace

extern "C" {

int tc_interp_create(tc_table* const* tables, int n_tables, int n_dim,
                     const double* points, tc_interp** out) {
  TC_CHECK(out != nullptr, "interp output pointer is NULL");
  *out = nullptr;
  TC_CHECK(tables != nullptr && points != nullptr, "NULL argument");
  TC_CHECK(n_tables >= 1 && n_dim >= 1 && n_dim <= tc::kMaxInterpDim,
           "invalid number of tables or dimensions");
  std::unique_ptr<tc_interp> it(new tc_interp);
  TC_HIP(hipGetDevice(&it->device));
  it->n_tables = n_tables;
  it->n_dim = n_dim;
  it->tables.assign(tables, tables + n_tables);
  tc_table* t0 = tables[0];
  for (int k = 0; k < n_tables; ++k) {
    tc_table* t = tables[k];
    TC_CHECK(t != nullptr, "table %d is NULL", k);
    TC_CHECK(t->device == it->device, "table %d lives on another device", k);
    if (t->compute_dtype != TC_DTYPE_F64)
      return fail(TC_ERR_UNSUPPORTED, "interpolation of float32 tables is not built");
    TC_CHECK(t->mode == t0->mode && t->n_bins == t0->n_bins && t->n_r == t0->n_r &&
                 t->plan.perm == t0->plan.perm,
             "table %d differs from table 0 in mode, shape or gal_type layout", k);
  }
  // abscissae: sorted unique values per dimension (interpolator.py:41-43)
  int64_t grid_size = 1;
  std::vector<double> xp_all, a_all;
  for (int d = 0; d < n_dim; ++d) {
    std::vector<double> xp;
    for (int k = 0; k < n_tables; ++k) xp.push_back(points[(size_t)k * n_dim + d]);
    std::sort(xp.begin(), xp.end());
    xp.erase(std::unique(xp.begin(), xp.end()), xp.end());
    TC_CHECK((int)xp.size() <= tc::kMaxInterpAxis, "more than %d grid values in "
             "dimension %d", tc::kMaxInterpAxis, d);
    // interpolator.py:239-241
    TC_CHECK(xp.size() >= 4,
             "Cannot perform spline interpolation with less than 4 values.");
    std::vector<double> a;
    TC_CHECK(tc::spline_interpolation_matrix((int)xp.size(), xp.data(), a),
             "singular spline system in dimension %d", d);
    it->axis_offset.push_back((int)xp_all.size());
    it->a_offset.push_back((int)a_all.size());
    xp_all.insert(xp_all.end(), xp.begin(), xp.end());
    a_all.insert(a_all.end(), a.begin(), a.end());
    grid_size *= (int64_t)xp.size();
    it->xp.push_back(xp);
  }
  // the points must form the full grid, each node once (interpolator.py:45-57)
  TC_CHECK(grid_size == n_tables, "The 'param_dict_table' does not describe a grid.");
  std::vector<char> taken((size_t)grid_size, 0);
  it->table_node.resize((size_t)n_tables * n_dim);
  for (int k = 0; k < n_tables; ++k) {
    int64_t flat = 0;
    for (int d = 0; d < n_dim; ++d) {
      const std::vector<double>& xp = it->xp[d];
      const int node = (int)(std::lower_bound(xp.begin(), xp.end(),
                                              points[(size_t)k * n_dim + d]) -
                             xp.begin());
      it->table_node[(size_t)k * n_dim + d] = node;
      flat = flat * (int64_t)xp.size() + node;
    }
    TC_CHECK(!taken[flat], "The 'param_dict_table' does not describe a grid.");
    taken[flat] = 1;
  }
  // classes of identical halo tables share their occupations (interpolator.py:63-70)
  it->table_class.assign(n_tables, -1);
  for (int k = 0; k < n_tables; ++k) {
    for (size_t v = 0; v < it->class_table.size(); ++v) {
      if (same_bins(tables[k], tables[it->class_table[v]])) {
        it->table_class[k] = (int32_t)v;
        break;
      }
    }
    if (it->table_class[k] < 0) {
      it->table_class[k] = (int32_t)it->class_table.size();
      it->class_table.push_back(k);
    }
  }
  const size_t n_classes = it->class_table.size();
  it->nbuf.resize(n_classes);
  it->ngal2.resize(n_classes);
  it->nbuf_ptrs.assign(n_classes, nullptr);
  it->ngal_ptrs.assign(n_classes, nullptr);

  TC_HIP(hipStreamCreateWithFlags(&it->stream, hipStreamNonBlocking));
  std::vector<void*> table_ptrs;
  for (int k = 0; k < n_tables; ++k) table_ptrs.push_back(tables[k]->d_table);
  std::vector<void*> zeros(n_classes, nullptr);
  int status = upload(xp_all, &it->d_xp);
  if (status == TC_OK) status = upload(a_all, &it->d_a);
  if (status == TC_OK) status = upload(it->table_node, &it->d_table_node);
  if (status == TC_OK) status = upload(it->table_class, &it->d_table_class);
  if (status == TC_OK) status = upload(table_ptrs, &it->d_tables);
  if (status == TC_OK) status = upload(zeros, &it->d_nbufs);
  if (status == TC_OK) status = upload(zeros, &it->d_ngal_parts);
  if (status != TC_OK) {
    tc_interp_destroy(it.release());
    return status;
  }
  *out = it.release();
  return TC_OK;
}

int tc_interp_destroy(tc_interp* it) {
  if (it == nullptr) return TC_OK;
  (void)hipSetDevice(it->device);
  if (it->stream) (void)hipStreamSynchronize(it->stream);
  for (void* p : {it->d_xp, it->d_a, it->d_table_node, it->d_table_class, it->d_tables,
                  it->d_nbufs, it->d_ngal_parts})
    if (p) (void)hipFree(p);
  for (auto& kv : it->chunkings)
    for (void* p : {kv.second->chunks, kv.second->groups})
      if (p) (void)hipFree(p);
  for (DeviceBuffer& b : it->nbuf) b.release();
  for (DeviceBuffer& b : it->ngal2) b.release();
  for (DeviceBuffer* b : {&it->theta, &it->x, &it->coef, &it->partial, &it->out_ngal,
                          &it->out_xi})
    b->release();
  if (it->stream) (void)hipStreamDestroy(it->stream);
  delete it;
  return TC_OK;
}

int tc_interp_synchronize(tc_interp* it) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  TC_HIP(hipStreamSynchronize(it->stream));
  return TC_OK;
}

int tc_interp_axis(const tc_interp* it, int dim, int* n, double* xp, int size) {
  TC_CHECK(it != nullptr && n != nullptr, "NULL argument");
  TC_CHECK(dim >= 0 && dim < it->n_dim, "invalid dimension %d", dim);
  *n = (int)it->xp[dim].size();
  if (xp != nullptr)
    for (int i = 0; i < std::min(*n, size); ++i) xp[i] = it->xp[dim][i];
  return TC_OK;
}

int tc_interp_predict_zheng07_batch_device(tc_interp* it, const double* theta_device,
                                           int n_theta, const double* x_device,
                                           int64_t n_draws, int n_gauss, unsigned flags,
                                           double* ngal_device, double* xi_device) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  int status = check_predict_args(it->tables[0], theta_device, n_theta, n_draws,
                                  n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(x_device && ngal_device && xi_device, "NULL pointer");
  TC_HIP(hipSetDevice(it->device));
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? it->tables[0]->plan.n_components : 1;
  for (int64_t begin = 0; begin < n_draws; begin += kMaxSlab) {
    const int64_t n = std::min(kMaxSlab, n_draws - begin);
    status = interp_predict_device(
        it, theta_device + begin * n_theta, n_theta, x_device + begin * it->n_dim, n,
        n_gauss, flags, ngal_device + begin * (separate ? 2 : 1),
        xi_device + begin * n_comp * it->tables[0]->n_r);
    if (status != TC_OK) return status;
  }
  return TC_OK;
}

int tc_interp_predict_zheng07_batch(tc_interp* it, const double* theta, int n_theta,
                                    const double* x, int64_t n_draws, int n_gauss,
                                    unsigned flags, double* ngal, double* xi) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  int status = check_predict_args(it->tables[0], theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(x && ngal && xi, "NULL pointer");
  TC_HIP(hipSetDevice(it->device));
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? it->tables[0]->plan.n_components : 1;
  const size_t ngal_count = (size_t)n_draws * (separate ? 2 : 1);
  const size_t xi_count = (size_t)n_draws * n_comp * it->tables[0]->n_r;
  status = it->theta.reserve((size_t)n_draws * n_theta * 8, it->stream);
  if (status == TC_OK) status = it->x.reserve((size_t)n_draws * it->n_dim * 8, it->stream);
  if (status == TC_OK) status = it->out_ngal.reserve(ngal_count * 8, it->stream);
  if (status == TC_OK) status = it->out_xi.reserve(xi_count * 8, it->stream);
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(it->theta.ptr, theta, (size_t)n_draws * n_theta * 8,
                        hipMemcpyHostToDevice, it->stream));
  TC_HIP(hipMemcpyAsync(it->x.ptr, x, (size_t)n_draws * it->n_dim * 8,
                        hipMemcpyHostToDevice, it->stream));
  status = tc_interp_predict_zheng07_batch_device(
      it, (const double*)it->theta.ptr, n_theta, (const double*)it->x.ptr, n_draws,
      n_gauss, flags, (double*)it->out_ngal.ptr, (double*)it->out_xi.ptr);
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(ngal, it->out_ngal.ptr, ngal_count * 8, hipMemcpyDeviceToHost,
                        it->stream));
  TC_HIP(hipMemcpyAsync(xi, it->out_xi.ptr, xi_count * 8, hipMemcpyDeviceToHost,
                        it->stream));
  TC_HIP(hipStreamSynchronize(it->stream));
  return TC_OK;
}

}  // extern "C"

// ---- multi-GPU: RCCL, resolved at run time --------------------------------------------
//
// One process per GPU (torchrun style).  librccl is opened lazily so that single-GPU use
// and CPU-only hosts never load it.  The only collective of the path is the gather of the
// per-rank results on the root; a barrier is provided for timing.
#include <dlfcn.h>

namespace {

typedef struct ncclComm* nccl_comm_t;
typedef struct { char internal[TC_UNIQUE_ID_BYTES]; } nccl_unique_id;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_unique_id*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*Gather)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle != nullptr) return TC_OK;
  const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  void* handle = nullptr;
  for (const char* name : names) {
    handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (handle != nullptr) break;
  }
  if (handle == nullptr) return fail(TC_ERR_RCCL, "cannot load librccl: %s", dlerror());
#define TC_SYM(field, symbol)                                                  \
  *(void**)(&g_rccl.field) = dlsym(handle, symbol);                            \
  if (g_rccl.field == nullptr)                                                 \
    return fail(TC_ERR_RCCL, "librccl lacks %s", symbol);
  TC_SYM(GetUniqueId, "ncclGetUniqueId")
  TC_SYM(CommInitRank, "ncclCommInitRank")
  TC_SYM(CommDestroy, "ncclCommDestroy")
  TC_SYM(Gather, "ncclGather")
  TC_SYM(AllReduce, "ncclAllReduce")
  TC_SYM(GetErrorString, "ncclGetErrorString")
#undef TC_SYM
  g_rccl.handle = handle;
  return TC_OK;
}

#define TC_RCCL(call)                                                          \
  do {                                                                         \
    int tc_rccl_status = (call);                                               \
    if (tc_rccl_status != 0)                                                   \
      return fail(TC_ERR_RCCL, "%s failed: %s", #call,                         \
                  g_rccl.GetErrorString(tc_rccl_status));                      \
  } while (0)

constexpr int kNcclFloat64 = 8;
constexpr int kNcclSum = 0;

}  // namespace

struct tc_comm {
  int device = 0;
  int n_ranks = 1;
  int rank = 0;
  nccl_comm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr;
  hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};  // per send-buffer slot
  double* token = nullptr;   // one double for the barrier all-reduce
};

extern "C" {

int tc_comm_unique_id(void* id) {
  TC_CHECK(id != nullptr, "id is NULL");
  int status = load_rccl();
  if (status != TC_OK) return status;
  nccl_unique_id unique;
  TC_RCCL(g_rccl.GetUniqueId(&unique));
  memcpy(id, &unique, TC_UNIQUE_ID_BYTES);
  return TC_OK;
}

int tc_comm_create(const void* id, int n_ranks, int rank, tc_comm** out) {
  TC_CHECK(out != nullptr && id != nullptr, "NULL argument");
  *out = nullptr;
  TC_CHECK(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "invalid rank %d of %d", rank,
           n_ranks);
  int status = load_rccl();
  if (status != TC_OK) return status;
  std::unique_ptr<tc_comm> c(new tc_comm);
  TC_HIP(hipGetDevice(&c->device));
  c->n_ranks = n_ranks;
  c->rank = rank;
  nccl_unique_id unique;
  memcpy(&unique, id, TC_UNIQUE_ID_BYTES);
  TC_RCCL(g_rccl.CommInitRank(&c->comm, n_ranks, unique, rank));
  // RCCL prints a version banner through C stdio; push it out now so that it cannot
  // trail the caller's own output (bench.py's JSON line) at process exit
  fflush(stdout);
  fflush(stderr);
  TC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  TC_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  for (hipEvent_t& event : c->done)
    TC_HIP(hipEventCreateWithFlags(&event, hipEventDisableTiming));
  TC_HIP(hipMalloc((void**)&c->token, sizeof(double)));
  TC_HIP(hipMemset(c->token, 0, sizeof(double)));
  *out = c.release();
  return TC_OK;
}

int tc_comm_destroy(tc_comm* c) {
  if (c == nullptr) return TC_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->token) (void)hipFree(c->token);
  if (c->ready) (void)hipEventDestroy(c->ready);
  for (hipEvent_t event : c->done)
    if (event) (void)hipEventDestroy(event);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TC_OK;
}

int tc_comm_gather(tc_comm* c, tc_table* t, const double* send_device,
                   double* recv_device, int64_t count, int root, int slot) {
  TC_CHECK(slot >= 0 && slot < 4, "slot must be in [0, 4)");
  TC_CHECK(c != nullptr && send_device != nullptr, "NULL argument");
  TC_CHECK(count >= 0 && root >= 0 && root < c->n_ranks, "invalid count or root");
  TC_CHECK(c->rank != root || recv_device != nullptr, "recv buffer is NULL on the root");
  TC_HIP(hipSetDevice(c->device));
  if (t != nullptr) {
    // the gather starts once the predictions queued on the table's stream are done,
    // without blocking the host: later batches overlap with the transfer
    TC_HIP(hipEventRecord(c->ready, t->lanes[t->cur].stream));
    TC_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  }
  TC_RCCL(g_rccl.Gather(send_device, recv_device, (size_t)count, kNcclFloat64, root,
                        c->comm, c->stream));
  TC_HIP(hipEventRecord(c->done[slot], c->stream));
  return TC_OK;
}

int tc_comm_release(tc_comm* c, tc_table* t, int slot) {
  TC_CHECK(c != nullptr && t != nullptr, "NULL argument");
  TC_CHECK(slot >= 0 && slot < 4, "slot must be in [0, 4)");
  TC_HIP(hipSetDevice(c->device));
  for (tc_table::Lane& lane : t->lanes)
    TC_HIP(hipStreamWaitEvent(lane.stream, c->done[slot], 0));
  return TC_OK;
}

int tc_comm_barrier(tc_comm* c) {
  TC_CHECK(c != nullptr, "comm handle is NULL");
  TC_HIP(hipSetDevice(c->device));
  TC_RCCL(g_rccl.AllReduce(c->token, c->token, 1, kNcclFloat64, kNcclSum, c->comm,
                           c->stream));
  TC_HIP(hipStreamSynchronize(c->stream));
  return TC_OK;
}

int tc_comm_synchronize(tc_comm* c) {
  TC_CHECK(c != nullptr, "comm handle is NULL");
  TC_HIP(hipSetDevice(c->device));
  TC_HIP(hipStreamSynchronize(c->stream));
  return TC_OK;
}

}  // extern "C"
