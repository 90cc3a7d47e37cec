// Table handle of the C ABI (include/tabcorr_amd.h): upload and re-layout of one
// tabulated correlation matrix, the batched predict / chi2 / occupation entry points,
// and the measurement hooks bench.py uses.
#include <cmath>

#include "internal.h"

using namespace tc::host;

extern "C" {

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
  TC_CHECK(compute_dtype != TC_DTYPE_F32 || n_bins < 65535,
           "too many bins for the float32 variant");

  // (tc_table_destroy releases whatever exists when a later step fails)
  struct Destroy {
    void operator()(tc_table* table) const { tc_table_destroy(table); }
  };
  std::unique_ptr<tc_table, Destroy> t(new tc_table);
  TC_HIP(hipGetDevice(&t->device));
  {
    hipDeviceProp_t prop;
    TC_HIP(hipGetDeviceProperties(&prop, t->device));
    t->n_cus = std::max(1, prop.multiProcessorCount);
    // MI355X: 8 accelerator complexes (XCDs) of 32 CUs; a compute partition exposes fewer
    t->n_xcds = std::max(1, t->n_cus / 32);
  }
  t->tuning.load();
  t->mode = mode;
  t->n_bins = n_bins;
  t->n_r = n_r;
  t->n_pairs = n_pairs;
  t->compute_dtype = compute_dtype;
  t->legacy = dist_index == nullptr;
  // r tiling: at most 32 accumulators per lane, a multiple of 4 so that a block
  // of at most 4 entries fills whole 128-byte lines (float64); tiles of exactly 32 r
  // values and blocks of 8 entries for the float32 MFMA kernel.
  const int max_rt = 32;
  t->n_rtiles = (n_r + max_rt - 1) / max_rt;
  if (compute_dtype == TC_DTYPE_F32) {
    t->rt = tc::kF32Tile;
    tc::build_plan(mode, n_bins, is_central, tc::kF32Block,
                   env_int("TC_ROW_BUDGET_F32", 128), t->plan);   // (developer builds)
  } else {
    int rt = (n_r + t->n_rtiles - 1) / t->n_rtiles;
    t->rt = (rt + 3) / 4 * 4;
    // LDS rows a workgroup may stage: half the bins (one triangle or one column block
    // of the cen-sat rectangle) plus a few rows, within 28..66 KB
    const int budget = std::max(56, std::min(128, n_bins / 2 + 6));
    tc::build_plan(mode, n_bins, is_central, tc::kF64Block,
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
    // Re-laid-out matrix, per r tile and position, with the pair prefactor
    // (tabcorr.py:638-642) folded in (a multiplication by 2 is exact) and zero
    // rows at the padding positions.
    t->table_bytes = count * sizeof(double);
    tmp64.assign(count, 0.0);
    for (int r = 0; r < n_r; ++r) {
      const int tile = r / rt, rr = r % rt;
      for (int64_t q = 0; q < n_positions; ++q) {
        const int64_t column = t->plan.column[q];
        if (column < 0) continue;
        // blocks of 8 positions (two steps of 4) as [r sub-tile of 4][position in step]
        // [r in sub-tile][step] (kernels.hip.h)
        const size_t index = ((size_t)tile * n_positions + q / 8 * 8) * rt +
                             (((rr / 4) * 16 + (q % 4) * 4 + rr % 4) * 2 + (q / 4) % 2);
        tmp64[index] = source(r, column) * t->plan.prefactor[q];
      }
    }
    // LDS row offsets per pair of steps and position in step: (i, j of step 0, i, j of
    // step 1) * 512 bytes
    pos_ij.assign((size_t)n_positions * 2, 0);
    for (int64_t q = 0; q < n_positions; ++q) {
      const size_t slot = ((size_t)(q / 8) * 4 + q % 4) * 4 + ((q / 4) % 2) * 2;
      pos_ij[slot] = std::max(t->plan.pos_i[q], 0) * 512;
      pos_ij[slot + 1] = t->plan.pos_j[q] * 512;
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
    pos_ij.assign((size_t)n_positions, 0);
    for (int64_t q = 0; q < n_positions; ++q) {
      const int64_t block = q / 8;
      const int p = (int)(q % 8) / 2, k = (int)(q % 2);
      const int i = t->plan.pos_i[q] < 0 ? 0 : t->plan.pos_i[q];
      pos_ij[(size_t)block * 8 + k * 4 + p] = (i << 16) | t->plan.pos_j[q];
    }
  }
  if (mode == TC_MODE_CROSS && compute_dtype == TC_DTYPE_F64 &&
      (int64_t)n_r + 1 <= tc::kCrossMaxRows) {
    t->cross_host.resize((size_t)n_bins * n_r);
    for (int g = 0; g < n_bins; ++g)
      for (int r = 0; r < n_r; ++r)
        t->cross_host[(size_t)g * n_r + r] = source(r, t->plan.perm[g]);
  }
  // (streams of the lanes in use; tc_table_set_option "lanes" creates further ones)
  t->n_lanes = std::max(1, std::min(t->tuning.lanes, (int)tc_table::kMaxLanes));
  {
    hipStream_t streams[tc_table::kMaxLanes] = {};
    const int created = create_lane_streams(t->n_lanes, streams);
    for (int l = 0; l < t->n_lanes; ++l) t->lanes[l].stream = streams[l];
    if (created != TC_OK) return created;
    for (int l = 0; l < t->n_lanes; ++l)
      TC_HIP(hipEventCreateWithFlags(&t->lanes[l].finished, hipEventDisableTiming));
  }
  t->stream = t->lanes[0].stream;
  TC_HIP(hipEventCreate(&t->ev_begin));
  TC_HIP(hipEventCreate(&t->ev_end));
  int status = compute_dtype == TC_DTYPE_F64 ? upload(tmp64, &t->d_table)
                                             : upload(tmp32, &t->d_table);
  if (status == TC_OK && compute_dtype == TC_DTYPE_F32)
    status = upload(pos_ij, &t->d_pos_ij);
  if (status == TC_OK && compute_dtype == TC_DTYPE_F64)
    status = upload(pos_ij, &t->d_pos_off);
  if (status == TC_OK) status = upload(t->n_h, &t->d_n_h);
  if (status == TC_OK) status = upload(t->percentile, &t->d_percentile);
  if (status == TC_OK) status = upload(t->plan.perm, &t->d_perm);
  tc::find_node_groups(n_bins, t->plan.n_central, t->log_min.data(), t->log_max.data(),
                       t->node_groups);
  t->grouped = t->node_groups.largest > 1;
  if (status == TC_OK) status = upload(t->node_groups.begin, &t->d_group_begin);
  if (status == TC_OK) status = upload(t->node_groups.member, &t->d_group_member);
  {
    std::vector<double> n_h_m, percentile_m;
    for (int32_t g : t->node_groups.member) {
      n_h_m.push_back(t->n_h[g]);
      percentile_m.push_back(t->percentile[g]);
    }
    if (status == TC_OK) status = upload(n_h_m, &t->d_group_n_h);
    if (status == TC_OK) status = upload(percentile_m, &t->d_group_percentile);
  }
  if (status == TC_OK) {
    std::vector<double> math_table(tc::fm::kTableDoubles);
    tc::fm::build_tables(math_table.data());
    status = upload(math_table, &t->d_math_table);
  }
  // The quadratic-form kernel serves mode auto in float64 up to 256 MB of matrix.  While the
  // matrix fits an L2 (cfg2 0.8 MB, cfg3 3.2 MB) the shares walk it once per tile of 32 draws;
  // larger ones are walked r tile by r tile by all waves (hostmath.h: kQuadRtileMajor),
  // interpolators table by table, and matrices far beyond the L2s (BASELINE configs[4] in
  // float64: 38 r tiles of 4 MB) unit-synchronously: the waves of an XCD read the same units at
  // the same time (kQuadUnitSync: 4.40 ms per 10^4 draws against 4.76 for the segment kernel,
  // whose workgroups share their slice through LDS, and 7.0-7.3 in the other orders).
  const tc::QuadTiling quad_tiling = tc::quad_tiling(n_r);
  const double quad_blocks = (n_bins / 4.0 + 1.0) * (n_bins / 4.0 + 2.0) / 2.0;
  const double quad_bytes =
      quad_blocks * ((quad_tiling.n_u + 1) / 2) * 1024.0 * quad_tiling.n_rtiles;
  // float32: r tiles of 16 values, 1 KB per unit; an r tile's slice must stay in an L2
  // (r-tile-major order): up to 3 MB per r tile, i.e. about 300 bins
  const tc::QuadTiling quad_tiling_f32 = tc::quad_tiling_f32(n_r);
  const bool quad_f32 = compute_dtype == TC_DTYPE_F32 && quad_blocks * 1024.0 <= 3.0 * 1024 * 1024;
  if (status == TC_OK && mode == TC_MODE_AUTO &&
      ((compute_dtype == TC_DTYPE_F64 && quad_bytes <= 256.0 * 1024 * 1024) || quad_f32)) {
    // quadratic-form kernel: the matrix by galaxy type and, when the centrals do not fill
    // whole 4 x 4 blocks, the unpadded triangle for the total prediction
    t->quad = true;
    t->quad_tiling = quad_f32 ? quad_tiling_f32 : quad_tiling;
    status = build_quad_table(t.get(), true, tpcf_matrix, matrix_dtype, &t->quad_by_type);
    const int n_central = t->plan.n_central;
    // (... and for every float64 table with a single r tile: predict_fused_kernel walks the
    // whole triangle as one component)
    // (only where that kernel can ever take the table -- launch.hip: fused_eligible --: up to
    // 20 r values, up to 248 bins; a second copy of the matrix otherwise serves nobody)
    const bool fusable = !quad_f32 && quad_tiling.n_rtiles == 1 && n_r <= 20 && n_bins <= 248;
    if (status == TC_OK && ((n_central % 4 != 0 && n_central < n_bins) || fusable))
      status = build_quad_table(t.get(), false, tpcf_matrix, matrix_dtype, &t->quad_total);
  }
  if (status != TC_OK) return status;
  *out = t.release();
  return TC_OK;
}

int tc_table_destroy(tc_table* t) {
  if (t == nullptr) return TC_OK;
  (void)hipSetDevice(t->device);
  (void)resident_stop(t);
  if (t->resident.stream) (void)hipStreamDestroy(t->resident.stream);
  t->resident.mailbox.release();
  t->resident.ens_mailbox.release();
  t->resident.ens_out.release();
  t->resident.ens_device.release();
  t->resident.ens_aperture.release();
  t->resident.single_aperture.release();
  t->resident.ws.buffer.release();
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) (void)hipStreamSynchronize(lane.stream);
  for (void* p : {t->d_table, t->d_n_h, t->d_percentile, t->d_perm, t->d_math_table,
                  t->d_pos_ij, t->d_pos_off, t->d_group_begin, t->d_group_member,
                  t->d_group_n_h, t->d_group_percentile})
    if (p) (void)hipFree(p);
  for (auto& kv : t->quadrature)
    for (void* p : {kv.second.log_m, kv.second.m, kv.second.weight, kv.second.group_log_m,
                    kv.second.group_m, kv.second.group_weight, kv.second.series,
                    kv.second.series_thr, kv.second.group_series, kv.second.group_series_thr,
                    kv.second.sat_series, kv.second.sat_series_thr, kv.second.group_sat_series,
                    kv.second.group_sat_series_thr, kv.second.group_records, kv.second.sat_records, kv.second.cen_records})
      if (p) (void)hipFree(p);
  for (auto& kv : t->chunkings)
    for (void* p : {kv.second->chunks, kv.second->groups})
      if (p) (void)hipFree(p);
  t->quad_by_type.release();
  t->quad_total.release();
  t->cross_fused.release();
  t->cross_fused_wide.release();
  for (DeviceBuffer* b : {&t->theta, &t->out_ngal, &t->out_xi, &t->occupation,
                          &t->trace, &t->wave_trace, &t->chi2_data})
    b->release();
  for (tc_table::Lane& lane : t->lanes) {
    lane.nbuf.release();
    lane.ngal2.release();
    lane.partial.release();
    lane.xi.release();
    lane.nbuf32.release();
    lane.in_theta.release();
    lane.out.release();
    lane.cross_counters.release();
    if (lane.finished) (void)hipEventDestroy(lane.finished);
  }
  for (tc_table::Ticket& ticket : t->tickets)
    if (ticket.done) (void)hipEventDestroy(ticket.done);
  t->h_in.release();
  t->h_out.release();
  t->single_ws.buffer.release();
  for (auto& ev : t->kernel_events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  for (hipEvent_t event : t->chunk_events) (void)hipEventDestroy(event);
  if (t->ev_begin) (void)hipEventDestroy(t->ev_begin);
  if (t->ev_end) (void)hipEventDestroy(t->ev_end);
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) (void)hipStreamDestroy(lane.stream);
  delete t;
  return TC_OK;
}

int tc_table_synchronize(tc_table* t) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
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

}  // extern "C"

namespace {
int autotune(tc_table* t, unsigned flags);

// The built-in estimate of which form serves a batch (launch.hip: fused_eligible) was fitted on
// a handful of table shapes and is up to 1.4 x behind the best form elsewhere (VERDICT r04 item
// 5).  So a handle that sees a LOOP of pipelined or asynchronous calls measures: the
// autotune_after-th such call with one combination of flags runs option "autotune" for it.
int maybe_autotune(tc_table* t, int n_gauss, unsigned flags) {
  if (t->autotuning || t->tuning.autotune_after <= 0 || t->tuning.deterministic != 0 ||
      n_gauss != 10 || t->tuning.fused != 1 ||
      t->tuning.fused_min_draws != 0 || t->tuning.fused_draws != 0 || t->tuning.fused_waves != 0)
    return TC_OK;
  const bool pipelined = t->async_lane >= 0 || (t->force_lane < 0 && t->tuning.pipeline &&
                                                t->n_lanes > 1);
  // (only where the one-launch forms can serve the table at all)
  if (!pipelined || !t->quad || t->compute_dtype != TC_DTYPE_F64 || t->mode != TC_MODE_AUTO ||
      t->quad_tiling.n_rtiles != 1 || t->n_r > 20 || t->chain || t->tuning.trace ||
      t->fuse_chi2_out != nullptr)
    return TC_OK;
  const unsigned key = flags & (TC_FLAG_SEPARATE_GAL_TYPE | TC_FLAG_MODULATE_WITH_CENOCC |
                                TC_FLAG_ASSEMBIAS | TC_FLAG_LEAUTHAUD11);
  if (t->autotuned.count(key) != 0) return TC_OK;
  if (++t->pipelined_calls[key] < t->tuning.autotune_after) return TC_OK;
  t->pipelined_calls[key] = 0;
  // (the measurement calls this entry point itself, with its own lanes and options)
  const int saved_force = t->force_lane, saved_async = t->async_lane, saved_cur = t->cur;
  t->autotuning = true;
  t->force_lane = t->async_lane = -1;
  const int status = autotune(t, key);
  t->autotuning = false;
  t->force_lane = saved_force;
  t->async_lane = saved_async;
  t->cur = saved_cur;
  if (status != TC_OK) {
    // (a table the measurement cannot serve: keep the estimate, do not try again)
    t->tuning.autotune_after = 0;
    (void)tc_table_synchronize(t);
  }
  return TC_OK;
}
}  // namespace

extern "C" {

int tc_predict_zheng07_batch_device(tc_table* t, const double* theta_device,
                                    int n_theta, int64_t n_draws, int n_gauss,
                                    unsigned flags, double* ngal_device,
                                    double* xi_device) {
  int status = check_predict_args(t, theta_device, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(ngal_device && xi_device, "output pointer is NULL");
  TC_HIP(hipSetDevice(t->device));
  if (t->resident.running && (status = resident_stop(t)) != TC_OK) return status;
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  t->chi2_fused = false;     // (set by whichever form writes the likelihood itself)
  if (t->force_lane >= 0)
    t->cur = t->force_lane;
  else if (t->async_lane >= 0)
    t->cur = t->async_lane;
  else
    t->cur = t->tuning.pipeline ? (int)(t->device_calls++ % t->n_lanes) : 0;
  // a loop of pipelined / asynchronous calls: its autotune_after-th call measures which form
  // serves which batch size on this table (about half a second, once per combination of flags)
  if ((status = maybe_autotune(t, n_gauss, flags)) != TC_OK) return status;
  const int64_t slab = max_slab(t);
  for (int64_t begin = 0; begin < n_draws; begin += slab) {
    const int64_t n = std::min(slab, n_draws - begin);
    const int64_t ldb = (n + 63) / 64 * 64;
    if (t->mode == TC_MODE_CROSS && !t->cross_host.empty() && t->tuning.fused != 0) {
      // mode cross: one launch per batch where it pays (launch.hip)
      tc_table* self = t;
      const CrossFused& cf = *choose_cross_fused(&self, 1, &t->cross_fused, &t->cross_fused_wide,
                                                 n, flags, &status);
      if (status != TC_OK) return status;
      const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1;
      if (cross_fused_eligible(t, cf, n, n_gauss, flags, alone)) {
        status = run_cross_fused(t, cf, nullptr, theta_device + begin * n_theta,
                                 n_theta, n, flags, ngal_device + begin * (separate ? 2 : 1),
                                 xi_device + begin * n_comp * t->n_r, t->lanes[t->cur].stream,
                                 &t->lanes[t->cur].partial, &t->lanes[t->cur].cross_counters);
        if (status != TC_OK) return status;
        t->prev = t->force_lane >= 0 ? -1 : t->cur;
        continue;
      }
    }
    if (fused_eligible(t, n, n_gauss, flags)) {
      status = run_fused(t, theta_device + begin * n_theta, n_theta, n, n_gauss, flags,
                         ngal_device + begin * (separate ? 2 : 1),
                         xi_device + begin * n_comp * t->n_r);
      if (status != TC_OK) return status;
      continue;
    }
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

}  // extern "C"

namespace tc {
namespace host {

// Host -> device copy of a small input through the pinned staging buffer.
int copy_in(PinnedBuffer* stage, void* device, const void* host, size_t bytes,
            hipStream_t stream) {
  Range range("upload");
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
  Range range("download");
  const size_t bytes = (ngal_count + xi_count) * sizeof(double);
  if (bytes <= stage_limit() && stage->reserve(bytes) == TC_OK) {
    double* h = (double*)stage->ptr;
    TC_HIP(hipMemcpyAsync(h, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost, stream));
    TC_HIP(hipMemcpyAsync(h + ngal_count, d_xi, xi_count * 8, hipMemcpyDeviceToHost,
                          stream));
    TC_HIP(hipStreamSynchronize(stream));
    memcpy(ngal, h, ngal_count * 8);
    parallel_copy(xi, h + ngal_count, xi_count * 8);
    return TC_OK;
  }
  TC_HIP(hipMemcpyAsync(ngal, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost, stream));
  TC_HIP(hipMemcpyAsync(xi, d_xi, xi_count * 8, hipMemcpyDeviceToHost, stream));
  TC_HIP(hipStreamSynchronize(stream));
  return TC_OK;
}

}  // namespace host
}  // namespace tc

namespace {
long long monotonic_ns() {
  timespec now{};
  clock_gettime(CLOCK_MONOTONIC, &now);
  return (long long)now.tv_sec * 1000000000LL + now.tv_nsec;
}

// Should this un-batched call go to the resident kernel although nobody asked for it?  From the
// auto_streak_min-th call on that follows its predecessor within auto_gap_us, unless the mode
// is backing off.
bool resident_auto_wants(tc_table* t) {
  tc_table::Resident& r = t->resident;
  const long long now = monotonic_ns();
  const bool close = r.auto_last_ns != 0 && now - r.auto_last_ns < (long long)r.auto_gap_us * 1000;
  r.auto_streak = close ? std::min(r.auto_streak + 1, 1 << 20) : 0;
  if (r.auto_backoff > 0) {
    --r.auto_backoff;
    return false;
  }
  // (a neighbour's resident kernel is running: this process is in a loop of un-batched calls,
  // and a launch of this handle might queue behind that kernel -- launch.hip)
  if (other_resident_running(t)) return true;
  return r.auto_streak >= r.auto_streak_min;
}

// Bookkeeping behind a call the automatic mode served: a window of 32 calls of which more than
// a quarter found the kernel gone ends the mode for auto_backoff_calls calls.
void resident_auto_served(tc_table* t, bool relaunched) {
  tc_table::Resident& r = t->resident;
  r.auto_last_ns = monotonic_ns();
  r.auto_relaunches += relaunched ? 1 : 0;
  if (++r.auto_window < 32) return;
  if (r.auto_relaunches > 8) {
    r.auto_backoff = r.auto_backoff_calls;
    r.auto_streak = 0;
    (void)resident_stop(t);
  }
  r.auto_window = r.auto_relaunches = 0;
}
}  // namespace

extern "C" {

int tc_predict_zheng07_many(tc_table* t, const double* theta, int n_theta, int n_walkers,
                            int n_gauss, unsigned flags, double* ngal, double* xi) {
  int status = check_predict_args(t, theta, n_theta, n_walkers, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_walkers == 0) return TC_OK;
  TC_CHECK(ngal && xi, "output pointer is NULL");
  // (option "deterministic" = 2: the batched path's one-launch form, whatever the batch size)
  if (batch_invariant_form(t, n_gauss, flags))
    return tc_predict_zheng07_batch(t, theta, n_theta, n_walkers, n_gauss, flags, ngal, xi);
  if (n_walkers > kSingleMaxWalkers || !single_draw_eligible(t, 1, n_gauss, flags)) {
    // (separated by galaxy type, several r tiles, float32 tables, the Leauthaud11 family, more
    // walkers than one launch takes: the batched path serves them)
    TC_CHECK(n_walkers > many_walkers_limit() || !single_draw_eligible(t, 1, n_gauss, flags),
             "internal: un-batched routing");
    return tc_predict_zheng07_batch(t, theta, n_theta, n_walkers, n_gauss, flags, ngal, xi);
  }
  TC_HIP(hipSetDevice(t->device));
  if (n_walkers == 1 && t->resident.enabled && resident_eligible(t, n_gauss))
    return resident_predict(t, theta, n_theta, n_gauss, flags, ngal, xi);
  // the resident kernel by itself for loops of un-batched calls (internal.h: Resident::auto_*)
  const bool auto_candidate = n_walkers == 1 && t->resident.auto_mode && !t->resident.enabled &&
                              resident_eligible(t, n_gauss);
  if (auto_candidate && resident_auto_wants(t)) {
    const unsigned long long before = t->resident.relaunches;
    status = resident_predict(t, theta, n_theta, n_gauss, flags, ngal, xi);
    if (status == TC_OK) {
      resident_auto_served(t, t->resident.relaunches != before);
      return status;
    }
    // Nobody asked for the resident kernel: whatever went wrong with it (it keeps leaving
    // before it answers, its mailbox or aperture cannot be allocated, ...) is not the caller's
    // business.  Stop it, stay away from it for a while -- for good after three such failures
    // -- and serve THIS call by a launch.
    (void)resident_stop(t);
    (void)hipGetLastError();
    tc_table::Resident& r = t->resident;
    r.auto_backoff = r.auto_backoff_calls;
    r.auto_streak = r.auto_window = r.auto_relaunches = 0;
    if (++r.auto_failures >= 3) r.auto_mode = false;
  }
  if (t->resident.enabled && ensemble_eligible(t, n_walkers, n_gauss, flags)) {
    status = ensemble_predict(t, theta, n_theta, n_walkers, n_gauss, flags, ngal, xi);
    if (status != TC_ERR_UNSUPPORTED) return status;     // (else: the launched path below)
  }
  if (t->resident.running && (status = resident_stop(t)) != TC_OK) return status;
  status = launch_single_draw(t, theta, n_theta, n_walkers, n_gauss, flags, &t->single_ws,
                              t->stream);
  if (status != TC_OK) return status;
  status = wait_single_done(&t->single_ws, t->stream, t->tuning.poll_done != 0);
  if (status != TC_OK) return status;
  for (int w = 0; w < n_walkers; ++w)
    combine_single_draw(t, t->single_ws, w, ngal + w, xi + (size_t)w * t->n_r);
  if (auto_candidate) t->resident.auto_last_ns = monotonic_ns();
  return TC_OK;
}

// One draw against SEVERAL tables -- the reference's documented step evaluates two:
// halotab_wp.predict(model), then halotab_ds.predict(model) (docs/guides/overview.rst:86-92,
// tests/test_database.py:17-18).  Every table's call is posted first -- the parameters into the
// mailbox of its resident kernel, or one launch of single_draw_kernel on the table's own stream
// -- and only then are the answers waited for, in table order: the tables' round trips overlap
// (two tables: 18.6 -> ~11 us per pair with the resident kernels, 33 -> ~20 us with launches).
// Which path serves a table is decided as in tc_predict_zheng07_many, table by table, and the
// results are those calls' bit for bit.
int tc_predict_zheng07_joint(tc_table* const* tables, int n_tables, const double* theta,
                             int n_theta, int n_gauss, unsigned flags, double* ngal,
                             double* const* xi) {
  constexpr int kMaxJoint = 16;
  TC_CHECK(tables != nullptr && n_tables >= 1 && n_tables <= kMaxJoint,
           "n_tables must be in [1, %d]", kMaxJoint);
  TC_CHECK(ngal != nullptr && xi != nullptr, "output pointer is NULL");
  enum Route { kGeneric, kResident, kLaunched };
  Route route[kMaxJoint];
  bool automatic[kMaxJoint];
  unsigned long long relaunches[kMaxJoint];
  for (int k = 0; k < n_tables; ++k) {
    tc_table* t = tables[k];
    int status = check_predict_args(t, theta, n_theta, 1, n_gauss, flags);
    if (status != TC_OK) return status;
    TC_CHECK(xi[k] != nullptr, "output pointer is NULL");
    for (int j = 0; j < k; ++j) TC_CHECK(tables[j] != t, "a table appears twice");
    route[k] = kGeneric;
    automatic[k] = false;
    if (batch_invariant_form(t, n_gauss, flags) || !single_draw_eligible(t, 1, n_gauss, flags))
      continue;
    TC_HIP(hipSetDevice(t->device));
    tc_table::Resident& r = t->resident;
    const bool eligible = resident_eligible(t, n_gauss);
    const bool candidate = r.auto_mode && !r.enabled && eligible;
    if (r.enabled && eligible) {
      route[k] = kResident;
    } else if (candidate && resident_auto_wants(t)) {
      route[k] = kResident;
      automatic[k] = true;
    } else {
      route[k] = kLaunched;
      automatic[k] = candidate;
    }
  }
  // post
  int first_error = TC_OK;
  for (int k = 0; k < n_tables; ++k) {
    tc_table* t = tables[k];
    if (route[k] == kGeneric) continue;
    TC_HIP(hipSetDevice(t->device));
    if (route[k] == kResident) {
      relaunches[k] = t->resident.relaunches;
      int status = resident_post(t, theta, n_theta, n_gauss, flags);
      if (status == TC_OK) continue;
      if (!automatic[k]) {
        first_error = status;
        n_tables = k;             // (collect what was posted, then report)
        break;
      }
      // (nobody asked for the resident kernel: tc_predict_zheng07_many's fallback)
      (void)resident_stop(t);
      (void)hipGetLastError();
      tc_table::Resident& r = t->resident;
      r.auto_backoff = r.auto_backoff_calls;
      r.auto_streak = r.auto_window = r.auto_relaunches = 0;
      if (++r.auto_failures >= 3) r.auto_mode = false;
      route[k] = kLaunched;
    }
    int status = TC_OK;
    if (t->resident.running) status = resident_stop(t);
    if (status == TC_OK)
      status = launch_single_draw(t, theta, n_theta, 1, n_gauss, flags, &t->single_ws, t->stream);
    if (status != TC_OK) {
      first_error = status;
      n_tables = k;
      break;
    }
  }
  // collect, in table order
  for (int k = 0; k < n_tables; ++k) {
    tc_table* t = tables[k];
    TC_HIP(hipSetDevice(t->device));
    int status = TC_OK;
    if (route[k] == kGeneric) {
      status = tc_predict_zheng07_many(t, theta, n_theta, 1, n_gauss, flags, ngal + k, xi[k]);
    } else if (route[k] == kResident) {
      status = resident_collect(t, theta, n_theta, n_gauss, flags, ngal + k, xi[k]);
      if (status == TC_OK && automatic[k])
        resident_auto_served(t, t->resident.relaunches != relaunches[k]);
      if (status != TC_OK && automatic[k]) {
        // (as above: stop, back off, serve this table by a launch)
        (void)resident_stop(t);
        (void)hipGetLastError();
        tc_table::Resident& r = t->resident;
        r.auto_backoff = r.auto_backoff_calls;
        r.auto_streak = r.auto_window = r.auto_relaunches = 0;
        if (++r.auto_failures >= 3) r.auto_mode = false;
        status = launch_single_draw(t, theta, n_theta, 1, n_gauss, flags, &t->single_ws,
                                    t->stream);
        if (status == TC_OK)
          status = wait_single_done(&t->single_ws, t->stream, t->tuning.poll_done != 0);
        if (status == TC_OK) combine_single_draw(t, t->single_ws, 0, ngal + k, xi[k]);
      }
    } else {
      status = wait_single_done(&t->single_ws, t->stream, t->tuning.poll_done != 0);
      if (status == TC_OK) combine_single_draw(t, t->single_ws, 0, ngal + k, xi[k]);
      if (automatic[k]) t->resident.auto_last_ns = monotonic_ns();
    }
    if (status != TC_OK && first_error == TC_OK) first_error = status;
  }
  return first_error;
}

namespace {
int predict_async(tc_table* t, const double* theta, int n_theta, int64_t n_draws, int n_gauss,
                  unsigned flags, const double* data, const double* precision, double* ngal,
                  double* second, bool chi2, int64_t* ticket_out, bool staging,
                  hipEvent_t wait_for = nullptr, hipEvent_t kernels_done = nullptr);

// How many chunks a synchronous host call of n_draws with out_bytes of results is cut into
// (0: not chunked -- the serial path).
int sync_chunks(const tc_table* t, int64_t n_draws, size_t out_bytes) {
  if (t->tuning.sync_chunks < 0 || !t->tuning.pipeline || t->n_lanes < 2 || t->chain) return 0;
  if (t->tuning.sync_chunks >= 1)
    return (int)std::min<int64_t>(t->tuning.sync_chunks, (n_draws + 63) / 64);
  // auto: about a megabyte of results per chunk, 2 .. 8 chunks of at least 1024 draws.  (All
  // chunks' kernels share the chip and finish together, so more chunks buy nothing for small
  // results -- 10^4 draws x 19 r values: 2 chunks 128 us, 3: 130, 4: 134, serial 150 -- while
  // the 61 MB of a (19, 40) table's 10^4 draws travel and are copied under the kernels.)
  if (n_draws < 2048) return 0;
  const int64_t by_bytes = std::max<int64_t>(2, (int64_t)(out_bytes >> 20));
  return (int)std::min<int64_t>(std::min<int64_t>(8, n_draws / 1024), by_bytes);
}

// A synchronous host-to-host call as overlapping chunks of draws: the draws
// of chunk k are staged and queued on lane k % lanes while chunk k - 1 computes; the results of
// chunk k are copied to the caller's arrays (on four host threads: parallel_copy) while later
// chunks compute and travel.  Before: upload -> all kernels alone on one lane -> whole download
// -> synchronise, strictly in series.  Every chunk takes the form the library picks for a
// PIPELINED call -- for tables the one-launch form serves, workgroups of 32 draws (option
// "sync_form"), whose results do not depend on where a draw sits in which batch: there the
// call returns the same bits for any number of chunks (tests/test_gpu_async.py); elsewhere the
// chunks differ from one piece by rounding (the schedule of the three-kernel path cuts a
// draw's sums where the batch size puts them).
int predict_chunked(tc_table* t, const double* theta, int n_theta, int64_t n_draws, int n_gauss,
                    unsigned flags, const double* data, const double* precision, double* ngal,
                    double* second, bool chi2, int n_chunks) {
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  const size_t ngal_cols = separate ? 2 : 1;
  const size_t second_cols = chi2 ? 1 : (size_t)n_comp * t->n_r;
  const size_t theta_bytes = (size_t)n_draws * n_theta * sizeof(double);
  const size_t out_bytes = (size_t)n_draws * (ngal_cols + second_cols) * sizeof(double);
  int status = t->h_in.reserve(theta_bytes);
  if (status == TC_OK) status = t->h_out.reserve(out_bytes);
  if (status != TC_OK) return status;
  const int64_t chunk = ((n_draws + n_chunks - 1) / n_chunks + 63) / 64 * 64;
  n_chunks = (int)((n_draws + chunk - 1) / chunk);
  int64_t tickets[64], begins[65];
  TC_CHECK(n_chunks <= 64, "internal: too many chunks");
  for (int k = 0; k <= n_chunks; ++k) begins[k] = std::min<int64_t>(n_draws, k * chunk);
  // Large results (tens of megabytes: a (19, 40) table's 10^4 draws are 61 MB): equal chunks on
  // four lanes finish in two groups of four, and the second group's downloads -- 145 us each,
  // queued behind one another -- have no kernels left to hide under (0.6 of 3.2 ms).  Instead
  // the chunks' kernels run ONE AFTER THE OTHER (events between the lanes), each with the chip
  // to itself, chunk k - 1 travelling and being copied while chunk k computes, and the chunks
  // SHRINK towards the end, so that what is left exposed behind the last kernel is a few
  // hundred draws' worth of transfer and copy.
  // (results of 2 KB per draw and more only: with a few values per draw the transfers are small
  // beside the kernels, and a chunk's launch alone on the chip runs below the rate of four
  // overlapping ones -- 4 x 10^5 draws of 19 values 2.48 against 2.19 ms, tools/r06_big_sync.py)
  const bool staggered = t->tuning.sync_stagger != 0 && out_bytes >= ((size_t)16 << 20) &&
                         second_cols * sizeof(double) >= 2048 && n_chunks >= 4 && !chi2 &&
                         t->n_lanes >= 2 &&
                         // (float32 tables: there the transfers are half of what the kernels
                         // take; the same table in float64 computes twice as long and loses
                         // more to small launches than the transfers cost: 2.0e6 -> 1.8e6)
                         t->compute_dtype == TC_DTYPE_F32;
  if (staggered) {
    // (tools/r06_stagger.py, 10^4 draws of a (19, 40) float32 table into arrays the caller
    // keeps, 1e6 calls/s: 6 chunks shrinking to 2 / 5 / 10 / 20 % of the first 3.44 / 3.58 / 3.62 /
    // 3.49; 7 chunks 3.52 / 3.58 / 3.49 / 3.45; 8 chunks 3.34 / 3.38 / 3.51 / 3.39; equal chunks
    // on four lanes 3.14)
    if (t->tuning.sync_chunks == 0) n_chunks = std::min(n_chunks, 6);
    double share[64];
    const double ratio = std::pow(0.01 * t->tuning.sync_stagger, 1.0 / (n_chunks - 1));
    double total = 0.0, done = 0.0;
    for (int k = 0; k < n_chunks; ++k) total += share[k] = std::pow(ratio, k);
    begins[0] = 0;
    for (int k = 0; k < n_chunks; ++k) {
      done += share[k];
      const int64_t end = k + 1 == n_chunks
                              ? n_draws
                              : std::min<int64_t>(n_draws, ((int64_t)(n_draws * done / total) + 63) / 64 * 64);
      begins[k + 1] = std::max(end, begins[k]);
    }
    while ((int)t->chunk_events.size() < n_chunks) {
      hipEvent_t event;
      TC_HIP(hipEventCreateWithFlags(&event, hipEventDisableTiming));
      t->chunk_events.push_back(event);
    }
  }
  double* h_theta = (double*)t->h_in.ptr;
  double* h_out = (double*)t->h_out.ptr;
  // (one form for every chunk size: the one-launch form with sync_form draws per workgroup
  // wherever it serves the table at all, not only from the batch size on where it pays)
  // (mode cross: the workgroups of ALL chunks together should fill the chip once -- see
  // tc_table::sync_cross_target)
  t->sync_cross_target = 512 / n_chunks;
  // (the latency form for every chunk when the whole call fits one round of 40-draw workgroups:
  // the chunks' launches then fill the chip together)
  t->sync_spread = n_draws >= t->tuning.fused_spread_min &&
                   (n_draws + 39) / 40 + n_chunks <= (int64_t)t->n_cus;
  const bool spread = t->sync_spread && fused_spread_eligible(t, n_draws, n_gauss, flags);
  t->sync_spread = spread;
  const int saved_draws = t->tuning.fused_draws, saved_min = t->tuning.fused_min_draws;
  if (spread) {
    t->tuning.fused_draws = 40;
  } else if (t->tuning.sync_form != 0 && saved_draws == 0 && t->tuning.fused == 1 &&
      t->tuning.deterministic < 2) {
    t->tuning.fused_draws = t->tuning.sync_form;
    if (saved_min == 0) t->tuning.fused_min_draws = 1;
  }
  for (int k = 0; k < n_chunks && status == TC_OK; ++k) {
    const int64_t begin = begins[k], n = begins[k + 1] - begin;
    memcpy(h_theta + begin * n_theta, theta + begin * n_theta, (size_t)n * n_theta * 8);
    // (the chunk's results side by side in the staging area: [ngal | xi] -- one copy command)
    double* out = h_out + begin * (ngal_cols + second_cols);
    status = predict_async(t, h_theta + begin * n_theta, n_theta, n, n_gauss, flags, data,
                           precision, out, out + n * ngal_cols, chi2, &tickets[k], true,
                           staggered && k > 0 ? t->chunk_events[k - 1] : nullptr,
                           staggered ? t->chunk_events[k] : nullptr);
    if (status != TC_OK) n_chunks = k;       // (wait for what was queued, then report)
  }
  t->tuning.fused_draws = saved_draws;
  t->tuning.fused_min_draws = saved_min;
  t->sync_cross_target = 0;
  t->sync_spread = false;
  for (int k = 0; k < n_chunks; ++k) {
    const int64_t begin = begins[k], n = begins[k + 1] - begin;
    const int waited = tc_table_wait(t, tickets[k]);
    if (waited != TC_OK) {
      (void)tc_table_synchronize(t);
      return waited;
    }
    if (status != TC_OK) continue;
    const double* out = h_out + begin * (ngal_cols + second_cols);
    memcpy(ngal + begin * ngal_cols, out, (size_t)n * ngal_cols * 8);
    parallel_copy(second + begin * second_cols, out + n * ngal_cols, (size_t)n * second_cols * 8);
  }
  return status;
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
  const size_t theta_bytes = (size_t)n_draws * n_theta * sizeof(double);
  const size_t out_bytes = (ngal_count + xi_count) * sizeof(double);
  // Small and medium calls (the un-batched predict() of an MCMC step, an ensemble of a few
  // thousand walkers): no copy commands at all.  The kernels read the draws from and write
  // the results to page-locked host memory, which the device addresses directly; two API
  // calls and two copy-engine round trips less (1 draw 45 -> 40 us, 1000 draws 72 -> 54 us;
  // beyond ~1 MB the copy engines win).
  // (option "deterministic" = 2: neither of the un-batched kernels -- their sums are cut
  // differently from the batched forms')
  const bool invariant = batch_invariant_form(t, n_gauss, flags);
  if (!invariant && t->resident.enabled && ensemble_eligible(t, n_draws, n_gauss, flags)) {
    // an ensemble of up to 256 walkers with option "resident": no launch at all (unless its
    // workgroups do not all find a place on the chip: the launched path below)
    status = ensemble_predict(t, theta, n_theta, (int)n_draws, n_gauss, flags, ngal, xi);
    if (status != TC_ERR_UNSUPPORTED) return status;
  }
  if (!invariant && n_draws <= many_walkers_limit() &&
      single_draw_eligible(t, 1, n_gauss, flags)) {
    // one draw -- or a handful (an ensemble sampler's proposals): ONE launch, the device-side
    // combination replaced by a few hundred additions here (kernels.hip.h: single_draw_kernel)
    return tc_predict_zheng07_many(t, theta, n_theta, (int)n_draws, n_gauss, flags, ngal, xi);
  }
  if (theta_bytes + out_bytes <= zero_copy_limit() && t->h_in.reserve(theta_bytes) == TC_OK &&
      t->h_out.reserve(out_bytes) == TC_OK) {
    memcpy(t->h_in.ptr, theta, theta_bytes);
    double* h = (double*)t->h_out.ptr;
    t->force_lane = 0;
    status = tc_predict_zheng07_batch_device(t, (const double*)t->h_in.ptr, n_theta,
                                             n_draws, n_gauss, flags, h, h + ngal_count);
    t->force_lane = -1;
    if (status != TC_OK) return status;
    TC_HIP(hipStreamSynchronize(t->stream));
    memcpy(ngal, h, ngal_count * 8);
    parallel_copy(xi, h + ngal_count, xi_count * 8);
    return TC_OK;
  }
  if (const int n_chunks = sync_chunks(t, n_draws, out_bytes))
    return predict_chunked(t, theta, n_theta, n_draws, n_gauss, flags, nullptr, nullptr, ngal, xi,
                           false, n_chunks);
  status = t->theta.reserve(theta_bytes, t->stream);
  if (status == TC_OK) status = t->out_ngal.reserve(ngal_count * 8, t->stream);
  if (status == TC_OK) status = t->out_xi.reserve(xi_count * 8, t->stream);
  if (status != TC_OK) return status;
  status = copy_in(&t->h_in, t->theta.ptr, theta, theta_bytes, t->stream);
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

namespace {

// The data vector and the precision matrix of a likelihood do not change between calls:
// they are uploaded when they differ from the host copy of the last upload.
int upload_chi2_data(tc_table* t, const double* data, const double* precision) {
  const int n_r = t->n_r;
  int status = t->chi2_data.reserve((size_t)(n_r + 1) * n_r * 8, t->stream);
  if (status != TC_OK) return status;
  const size_t data_count = (size_t)(n_r + 1) * n_r;
  if (t->chi2_host.size() != data_count ||
      memcmp(t->chi2_host.data(), data, (size_t)n_r * 8) != 0 ||
      memcmp(t->chi2_host.data() + n_r, precision, (size_t)n_r * n_r * 8) != 0) {
    // (kernels of earlier calls may still be reading the old values)
    for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
    t->chi2_host.assign(data, data + n_r);
    t->chi2_host.insert(t->chi2_host.end(), precision, precision + (size_t)n_r * n_r);
    TC_HIP(hipMemcpyAsync(t->chi2_data.ptr, t->chi2_host.data(), data_count * 8,
                          hipMemcpyHostToDevice, t->stream));
    TC_HIP(hipStreamSynchronize(t->stream));
  }
  return TC_OK;
}

}  // namespace

int tc_chi2_zheng07_batch_device(tc_table* t, const double* theta_device, int n_theta,
                                 int64_t n_draws, int n_gauss, unsigned flags,
                                 const double* data, const double* precision,
                                 double* ngal_device, double* chi2_device) {
  int status = check_predict_args(t, theta_device, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  TC_CHECK(!(flags & TC_FLAG_SEPARATE_GAL_TYPE),
           "chi2 is defined for the total correlation function only");
  if (n_draws == 0) return TC_OK;
  TC_CHECK(data && precision && ngal_device && chi2_device, "NULL pointer");
  TC_CHECK(n_draws <= max_slab(t), "at most %lld draws per call",
           (long long)max_slab(t));
  TC_HIP(hipSetDevice(t->device));
  if (t->resident.running) {
    const int stopped = resident_stop(t);
    if (stopped != TC_OK) return stopped;
  }
  status = upload_chi2_data(t, data, precision);
  if (status != TC_OK) return status;
  // xi stays in a workspace of the lane the call runs on; the lane's `finished` event is
  // re-recorded behind the chi2 kernel, so the next call on this lane (whose finalisation
  // waits for that event through the chain of lanes) cannot overwrite it early
  const int lane_index =
      t->force_lane >= 0   ? t->force_lane
      : t->async_lane >= 0 ? t->async_lane
      : t->tuning.pipeline ? (int)(t->device_calls % t->n_lanes)
                           : 0;
  tc_table::Lane& lane = t->lanes[lane_index];
  status = lane.xi.reserve((size_t)n_draws * t->n_r * 8, lane.stream);
  if (status != TC_OK) return status;
  const double* d_data = (const double*)t->chi2_data.ptr;
  t->fuse_chi2_data = d_data;
  t->fuse_chi2_out = chi2_device;
  status = tc_predict_zheng07_batch_device(t, theta_device, n_theta, n_draws, n_gauss, flags,
                                           ngal_device, (double*)lane.xi.ptr);
  t->fuse_chi2_out = nullptr;
  if (status != TC_OK) return status;
  if (t->chi2_fused) return TC_OK;      // (the finalisation kernel wrote chi2 itself)
  status = launch_chi2((const double*)lane.xi.ptr, n_draws, t->n_r, d_data, d_data + t->n_r,
                       chi2_device, lane.stream);
  if (status != TC_OK) return status;
  if (t->force_lane < 0 && t->chain) TC_HIP(hipEventRecord(lane.finished, lane.stream));
  return TC_OK;
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
  const size_t theta_bytes = (size_t)n_draws * n_theta * sizeof(double);
  status = t->out_xi.reserve(xi_count * 8, t->stream);
  if (status == TC_OK) status = upload_chi2_data(t, data, precision);
  if (status != TC_OK) return status;
  double* d_data = (double*)t->chi2_data.ptr;
  double* d_precision = d_data + n_r;
  // small calls: draws read from and results written to page-locked host memory directly
  const bool direct = theta_bytes + (size_t)n_draws * 16 <= zero_copy_limit() &&
                      t->h_in.reserve(theta_bytes) == TC_OK &&
                      t->h_out.reserve((size_t)n_draws * 16) == TC_OK;
  const double* theta_device = nullptr;
  double* d_ngal = nullptr;
  if (direct) {
    memcpy(t->h_in.ptr, theta, theta_bytes);
    theta_device = (const double*)t->h_in.ptr;
    d_ngal = (double*)t->h_out.ptr;
  } else {
    status = t->theta.reserve(theta_bytes, t->stream);
    if (status == TC_OK) status = t->out_ngal.reserve((size_t)n_draws * 2 * 8, t->stream);
    if (status != TC_OK) return status;
    status = copy_in(&t->h_in, t->theta.ptr, theta, theta_bytes, t->stream);
    if (status != TC_OK) return status;
    theta_device = (const double*)t->theta.ptr;
    d_ngal = (double*)t->out_ngal.ptr;
  }
  double* d_chi2 = d_ngal + n_draws;
  // (one slab: the finalisation kernel may write the likelihood itself)
  const bool fuse = n_draws <= max_slab(t);
  t->force_lane = 0;
  t->fuse_chi2_data = d_data;
  t->fuse_chi2_out = fuse ? d_chi2 : nullptr;
  status = tc_predict_zheng07_batch_device(t, theta_device, n_theta, n_draws, n_gauss, flags,
                                           d_ngal, (double*)t->out_xi.ptr);
  t->fuse_chi2_out = nullptr;
  t->force_lane = -1;
  if (status != TC_OK) return status;
  if (!(fuse && t->chi2_fused))
    status = launch_chi2((const double*)t->out_xi.ptr, n_draws, n_r, d_data, d_precision,
                         d_chi2, t->stream);
  if (status != TC_OK) return status;
  if (direct) {
    TC_HIP(hipStreamSynchronize(t->stream));
    memcpy(ngal, d_ngal, (size_t)n_draws * 8);
    memcpy(chi2, d_chi2, (size_t)n_draws * 8);
    return TC_OK;
  }
  return copy_out(&t->h_out, ngal, (size_t)n_draws, d_ngal, chi2, (size_t)n_draws,
                  d_chi2, t->stream);
}

namespace {

// Next ticket of the handle: its event (created on first use of the slot) and its number.
int next_ticket(tc_table* t, tc_table::Ticket** out) {
  tc_table::Ticket& slot = t->tickets[t->next_ticket % tc_table::kMaxTickets];
  if (slot.done == nullptr)
    TC_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
  slot.id = t->next_ticket++;
  *out = &slot;
  return TC_OK;
}

// Asynchronous host-to-host prediction or likelihood on the handle's next lane: everything
// (upload, three kernels, download, the ticket's event) is queued on that lane's stream, so
// lanes overlap each other's copies and kernels and nothing orders one ticket against
// another.
// `staging`: the buffers are the library's own page-locked staging areas (hipHostMalloc: the
// device sees them at the same addresses) -- the chunks of a synchronous call, predict_chunked.
// wait_for / kernels_done (the chunks of a synchronous call with large results): the call's
// kernels start behind `wait_for`, and `kernels_done` is recorded between its kernels and its
// download -- so that the chunks' kernels run one after the other, each with the chip to itself,
// while the previous chunk's results travel.
int predict_async(tc_table* t, const double* theta, int n_theta, int64_t n_draws, int n_gauss,
                  unsigned flags, const double* data, const double* precision, double* ngal,
                  double* second, bool chi2, int64_t* ticket_out, bool staging,
                  hipEvent_t wait_for, hipEvent_t kernels_done) {
  int status = check_predict_args(t, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  TC_CHECK(ticket_out != nullptr, "ticket is NULL");
  TC_CHECK(n_draws == 0 || (ngal && second), "output pointer is NULL");
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  TC_CHECK(!(chi2 && separate), "chi2 is defined for the total correlation function only");
  TC_CHECK(!chi2 || (data && precision), "NULL pointer");
  TC_CHECK(!chi2 || n_draws <= max_slab(t), "at most %lld draws per call",
           (long long)max_slab(t));
  const int n_comp = separate ? t->plan.n_components : 1;
  const size_t ngal_count = (size_t)n_draws * (separate ? 2 : 1);
  const size_t second_count = chi2 ? (size_t)n_draws : (size_t)n_draws * n_comp * t->n_r;
  const size_t theta_bytes = (size_t)n_draws * n_theta * sizeof(double);
  void *theta_seen = nullptr, *ngal_seen = nullptr, *second_seen = nullptr;
  if (staging) {
    theta_seen = (void*)theta;
    ngal_seen = ngal;
    second_seen = second;
  } else {
    TC_CHECK(n_draws == 0 || (is_pinned(theta, theta_bytes, &theta_seen) &&
                              is_pinned(ngal, ngal_count * 8, &ngal_seen) &&
                              is_pinned(second, second_count * 8, &second_seen)),
             "asynchronous calls need page-locked buffers (tc_host_alloc / tc_host_register)");
  }
  TC_HIP(hipSetDevice(t->device));
  if (t->resident.running) {
    const int stopped = resident_stop(t);
    if (stopped != TC_OK) return stopped;
  }
  tc_table::Ticket* ticket = nullptr;
  const int lane_index = t->tuning.pipeline ? (int)(t->device_calls++ % t->n_lanes) : 0;
  tc_table::Lane& lane = t->lanes[lane_index];
  if (n_draws > 0) {
    // Which arrays the kernels address in the caller's page-locked memory themselves and
    // which travel by copy command: see Tuning::async_direct_in / async_direct_out.
    // (the chunks of a synchronous call: the kernels store up to 1 MB per array themselves --
    // one call's 10^4 draws of 19 r values in two chunks 128 us against 151 with copy commands;
    // larger chunks, e.g. of a (19, 40) table, go through the copy engines)
    const int mode = staging ? t->tuning.sync_direct_out : t->tuning.async_direct_out;
    const size_t direct_bytes = staging ? kSyncDirectOutBytes : kDirectOutBytes;
    auto direct = [mode, direct_bytes](void* seen, size_t count) {
      return seen != nullptr && (mode == 1 || (mode == 2 && count * 8 <= direct_bytes));
    };
    const bool direct_in = t->tuning.async_direct_in && theta_seen != nullptr;
    const bool direct_ngal = direct(ngal_seen, ngal_count);
    const bool direct_second = direct(second_seen, second_count);
    if (!direct_in) status = lane.in_theta.reserve(theta_bytes, lane.stream);
    if (status == TC_OK && !(direct_ngal && direct_second))
      status = lane.out.reserve((ngal_count + second_count) * 8, lane.stream);
    if (status != TC_OK) return status;
    const double* d_theta = direct_in ? (const double*)theta_seen
                                      : (const double*)lane.in_theta.ptr;
    double* d_ngal = direct_ngal ? (double*)ngal_seen : (double*)lane.out.ptr;
    double* d_second = direct_second ? (double*)second_seen
                                     : (double*)lane.out.ptr + ngal_count;
    if (!direct_in) {
      Range range("upload");
      TC_HIP(hipMemcpyAsync(lane.in_theta.ptr, theta, theta_bytes, hipMemcpyHostToDevice,
                            lane.stream));
    }
    if (wait_for != nullptr) TC_HIP(hipStreamWaitEvent(lane.stream, wait_for, 0));
    const bool saved_chain = t->chain;
    t->async_lane = lane_index;
    t->chain = false;
    status = chi2 ? tc_chi2_zheng07_batch_device(t, d_theta, n_theta, n_draws, n_gauss, flags,
                                                 data, precision, d_ngal, d_second)
                  : tc_predict_zheng07_batch_device(t, d_theta, n_theta, n_draws, n_gauss,
                                                    flags, d_ngal, d_second);
    t->async_lane = -1;
    t->chain = saved_chain;
    if (status != TC_OK) return status;
    if (kernels_done != nullptr) TC_HIP(hipEventRecord(kernels_done, lane.stream));
    Range range("download");
    if (!direct_ngal && !direct_second && second == ngal + ngal_count) {
      // (adjacent in the caller's memory as in the staging buffer: one command)
      TC_HIP(hipMemcpyAsync(ngal, d_ngal, (ngal_count + second_count) * 8,
                            hipMemcpyDeviceToHost, lane.stream));
    } else {
      if (!direct_ngal)
        TC_HIP(hipMemcpyAsync(ngal, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost,
                              lane.stream));
      if (!direct_second)
        TC_HIP(hipMemcpyAsync(second, d_second, second_count * 8, hipMemcpyDeviceToHost,
                              lane.stream));
    }
  }
  status = next_ticket(t, &ticket);
  if (status != TC_OK) return status;
  TC_HIP(hipEventRecord(ticket->done, lane.stream));
  *ticket_out = ticket->id;
  return TC_OK;
}

}  // namespace

int tc_predict_zheng07_batch_async(tc_table* t, const double* theta, int n_theta,
                                   int64_t n_draws, int n_gauss, unsigned flags, double* ngal,
                                   double* xi, int64_t* ticket) {
  return predict_async(t, theta, n_theta, n_draws, n_gauss, flags, nullptr, nullptr, ngal, xi,
                       false, ticket, false);
}

int tc_chi2_zheng07_batch_async(tc_table* t, const double* theta, int n_theta, int64_t n_draws,
                                int n_gauss, unsigned flags, const double* data,
                                const double* precision, double* ngal, double* chi2,
                                int64_t* ticket) {
  return predict_async(t, theta, n_theta, n_draws, n_gauss, flags, data, precision, ngal, chi2,
                       true, ticket, false);
}

int tc_table_wait(tc_table* t, int64_t ticket) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_CHECK(ticket >= 0 && ticket < t->next_ticket, "unknown ticket %lld", (long long)ticket);
  const tc_table::Ticket& slot = t->tickets[ticket % tc_table::kMaxTickets];
  if (slot.id == ticket) {
    TC_HIP(hipEventSynchronize(slot.done));
    return TC_OK;
  }
  // the slot was reused: the ticket is older than the work now queued on every lane
  return tc_table_synchronize(t);
}

int tc_table_query(tc_table* t, int64_t ticket, int* done) {
  TC_CHECK(t != nullptr && done != nullptr, "NULL argument");
  TC_CHECK(ticket >= 0 && ticket < t->next_ticket, "unknown ticket %lld", (long long)ticket);
  const tc_table::Ticket& slot = t->tickets[ticket % tc_table::kMaxTickets];
  *done = 0;
  if (slot.id == ticket) {
    const hipError_t state = hipEventQuery(slot.done);
    if (state == hipSuccess) *done = 1;
    else if (state != hipErrorNotReady)
      return fail(TC_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(state));
    return TC_OK;
  }
  // a reused slot: done once every lane has passed its own later tickets
  *done = 1;
  for (tc_table::Lane& lane : t->lanes) {
    if (lane.stream == nullptr) continue;
    const hipError_t state = hipStreamQuery(lane.stream);
    if (state == hipErrorNotReady) *done = 0;
    else if (state != hipSuccess)
      return fail(TC_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(state));
  }
  return TC_OK;
}

int tc_mean_occupation_zheng07_batch(tc_table* t, const double* theta, int n_theta,
                                     int64_t n_draws, int n_gauss, unsigned flags,
                                     double* occupation) {
  int status = check_predict_args(t, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(occupation != nullptr, "output pointer is NULL");
  TC_HIP(hipSetDevice(t->device));
  if (t->resident.running) {
    const int stopped = resident_stop(t);
    if (stopped != TC_OK) return stopped;
  }
  const int64_t slab = max_slab(t);
  for (int64_t begin = 0; begin < n_draws; begin += slab) {
    const int64_t n = std::min(slab, n_draws - begin);
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
  if (t->resident.running) {
    const int stopped = resident_stop(t);
    if (stopped != TC_OK) return stopped;
  }
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  const int64_t slab = max_slab(t);
  for (int64_t begin = 0; begin < n_draws; begin += slab) {
    const int64_t n = std::min(slab, n_draws - begin);
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
    status = launch_occ_from_array(t, (const double*)t->occupation.ptr, n, ldb,
                                   (double*)t->lanes[0].nbuf.ptr,
                                   (double*)t->lanes[0].ngal2.ptr, t->stream);
    if (status != TC_OK) return status;
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

namespace {

// Option "autotune": which form serves a batch of n draws fastest on THIS table in the
// pipelined regime -- three kernels, one launch with 64-draw workgroups, one launch with
// 32-draw workgroups --, measured for a grid of batch sizes with draws from a wide prior box and
// kept in the handle (internal.h: AutoChoice).  `flags`: the predict flags to tune for.
int autotune(tc_table* t, unsigned flags) {
  TC_CHECK(!(flags & ~(TC_FLAG_SEPARATE_GAL_TYPE | TC_FLAG_MODULATE_WITH_CENOCC |
                       TC_FLAG_ASSEMBIAS | TC_FLAG_LEAUTHAUD11)),
           "autotune: the value is a combination of predict flags");
  TC_CHECK(!((flags & TC_FLAG_LEAUTHAUD11) && (flags & TC_FLAG_ASSEMBIAS)),
           "autotune: no assembly bias for the Leauthaud11 family");
  int status = tc_table_synchronize(t);
  if (status != TC_OK) return status;
  if (t->resident.running && (status = resident_stop(t)) != TC_OK) return status;
  t->autotuned.erase(flags);
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const bool leauthaud = (flags & TC_FLAG_LEAUTHAUD11) != 0;
  const int n_theta = leauthaud ? 14 : (flags & TC_FLAG_ASSEMBIAS) ? 7 : 5;
  const int n_comp = separate ? t->plan.n_components : 1;
  const int64_t n_max = AutoChoice::size(AutoChoice::kSizes - 1);
  // draws: uniform in the box of tabcorr_amd.synthetic.zheng07_draws (/ the defaults of the
  // Leauthaud11 model with a little scatter), a fixed multiplicative generator
  std::vector<double> theta((size_t)n_max * n_theta);
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto uniform = [&state]() {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(state >> 11) * (1.0 / 9007199254740992.0);
  };
  static const double lo[7] = {11.5, 0.1, 11.0, 12.5, 0.7, -1.0, -1.0};
  static const double hi[7] = {13.5, 0.8, 13.0, 14.5, 1.4, 1.0, 1.0};
  static const double leauthaud_mid[14] = {10.72, 12.35, 0.43, 0.56, 1.54, 0.2,  1.0,
                                           10.62, 0.859, 1.47, -0.13, 10.5, 0.7, 0.72};
  for (int64_t b = 0; b < n_max; ++b)
    for (int i = 0; i < n_theta; ++i)
      theta[(size_t)b * n_theta + i] =
          leauthaud ? leauthaud_mid[i] * (i < 12 ? 1.0 + 0.02 * (uniform() - 0.5) : 1.0)
                    : lo[i] + (hi[i] - lo[i]) * uniform();
  DeviceBuffer d_theta, d_ngal, d_xi;
  struct Release {
    DeviceBuffer *a, *b, *c;
    ~Release() {
      a->release();
      b->release();
      c->release();
    }
  } release{&d_theta, &d_ngal, &d_xi};
  status = d_theta.reserve(theta.size() * 8, t->stream);
  if (status == TC_OK) status = d_ngal.reserve((size_t)n_max * 2 * 8, t->stream);
  if (status == TC_OK) status = d_xi.reserve((size_t)n_max * n_comp * t->n_r * 8, t->stream);
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpy(d_theta.ptr, theta.data(), theta.size() * 8, hipMemcpyHostToDevice));

  const Tuning saved = t->tuning;
  struct Restore {
    tc_table* t;
    Tuning saved;
    ~Restore() { t->tuning = saved; }
  } restore{t, saved};
  t->tuning.fused_min_draws = 1;
  t->tuning.fused_max_draws = 1 << 30;
  auto time_form = [&](int64_t n, float* us) -> int {
    auto call = [&]() {
      return tc_predict_zheng07_batch_device(t, (const double*)d_theta.ptr, n_theta, n, 10, flags,
                                             (double*)d_ngal.ptr, (double*)d_xi.ptr);
    };
    int st = TC_OK;
    for (int k = 0; k < 4 && st == TC_OK; ++k) st = call();          // warm (schedules, lanes)
    if (st == TC_OK) st = tc_table_synchronize(t);
    if (st != TC_OK) return st;
    // at least 8 ms and 8 calls, at most 400
    int calls = 8;
    double seconds = 0.0;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < calls && st == TC_OK; ++k) st = call();
      if (st == TC_OK) st = tc_table_synchronize(t);
      if (st != TC_OK) return st;
      seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (seconds >= 8e-3 || calls >= 400) break;
      calls = (int)std::min<double>(400.0, std::max<double>(calls * 2.0,
                                                            calls * 1e-2 / std::max(seconds, 1e-6)));
    }
    *us = (float)(seconds / calls * 1e6);
    return TC_OK;
  };
  AutoChoice choice;
  for (int i = 0; i < AutoChoice::kSizes; ++i) {
    const int64_t n = AutoChoice::size(i);
    t->tuning.fused = 0;
    status = time_form(n, &choice.us[i][0]);
    if (status != TC_OK) return status;
    choice.form[i] = 0;
    float best = choice.us[i][0];
    const int shapes[2] = {64, 32};
    for (int k = 0; k < 2; ++k) {
      t->tuning.fused = 2;
      t->tuning.fused_draws = shapes[k];
      choice.us[i][1 + k] = 0.0f;
      if (!fused_eligible(t, n, 10, flags)) continue;
      // (the shape asked for is the one that would run?)
      const bool half = fused_wide_tables(t, separate, 10, flags) ||
                        fused_half_tiles(t, separate, n, 10, flags);
      if (half != (shapes[k] == 32)) continue;
      status = time_form(n, &choice.us[i][1 + k]);
      if (status != TC_OK) return status;
      // (a one-launch form must win by 2 %: ties go to the three kernels)
      if (choice.us[i][1 + k] < 0.98f * best) {
        best = choice.us[i][1 + k];
        choice.form[i] = shapes[k];
      }
    }
  }
  t->autotuned[flags] = choice;
  return TC_OK;
}

}  // namespace

int tc_table_autotune_result(const tc_table* t, unsigned flags, int capacity, int* count,
                             int64_t* sizes, int* forms, float* us) {
  TC_CHECK(t != nullptr && count && sizes && forms && us, "NULL argument");
  *count = 0;
  auto it = t->autotuned.find(flags);
  TC_CHECK(it != t->autotuned.end(), "no autotune result for these flags");
  TC_CHECK(capacity >= AutoChoice::kSizes, "capacity below %d", (int)AutoChoice::kSizes);
  for (int i = 0; i < AutoChoice::kSizes; ++i) {
    sizes[i] = AutoChoice::size(i);
    forms[i] = it->second.form[i];
    for (int k = 0; k < 3; ++k) us[3 * i + k] = it->second.us[i][k];
  }
  *count = AutoChoice::kSizes;
  return TC_OK;
}

int tc_table_resident_stats(const tc_table* t, int64_t* launches, int64_t* relaunches,
                            int64_t* failures, int* running) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  const tc_table::Resident& r = t->resident;
  if (launches) *launches = (int64_t)r.launch_id;
  if (relaunches) *relaunches = (int64_t)r.relaunches;
  if (failures) *failures = (int64_t)r.auto_failures;
  if (running) *running = r.running ? (r.ensemble ? 2 : 1) : 0;
  return TC_OK;
}

int tc_table_batch_invariant(tc_table* t, int n_gauss, unsigned flags, int* out) {
  TC_CHECK(t != nullptr && out != nullptr, "NULL argument");
  TC_HIP(hipSetDevice(t->device));
  *out = batch_invariant_form(t, n_gauss, flags) ? 1 : 0;
  return TC_OK;
}

int tc_table_set_option(tc_table* t, const char* name, int value) {
  TC_CHECK(t != nullptr && name != nullptr, "NULL argument");
  // (streams and events created below, and the resident kernel stopped, belong to the table's
  // device whatever device is current in the calling thread)
  TC_HIP(hipSetDevice(t->device));
  const std::string key(name);
  if (key == "pipeline") {
    t->tuning.pipeline = value != 0;
  } else if (key == "deterministic") {
    // 0 (default): the fastest form per call -- chosen from the table, the flags, the entry
    // point and the batch size, never from timing (no self-measurement unless "autotune_after"
    // or "autotune" ask for it); 1: the same, and the measured dispatch is refused; 2:
    // batch-invariant -- ONE kernel form per (table, flags) for every entry point and batch
    // size (internal.h: Tuning::deterministic)
    TC_CHECK(value >= 0 && value <= 2, "deterministic must be 0, 1 or 2");
    int status = resident_stop(t);
    if (status != TC_OK) return status;
    for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
    t->tuning.deterministic = value;
    if (value != 0) {
      t->autotuned.clear();
      t->tuning.autotune_after = 0;
    }
  } else if (key == "autotune_after") {
    TC_CHECK(value == 0 || t->tuning.deterministic == 0,
             "autotune_after: the measured dispatch is not available while \"deterministic\" is set");
    // the N-th pipelined / asynchronous call with one combination of predict flags measures the
    // forms by itself (option "autotune": ~0.5 s, once); 0: never
    TC_CHECK(value >= 0, "autotune_after must be non-negative");
    t->tuning.autotune_after = value;
  } else if (key == "autotune") {
    // value = the predict flags to tune for (TC_FLAG_*; n_gauss_prim = 10), or -1: forget
    // every measured choice (back to the formula of launch.hip: fused_eligible)
    if (value < 0) {
      t->autotuned.clear();
      return TC_OK;
    }
    TC_CHECK(t->tuning.deterministic == 0,
             "autotune: the measured dispatch is not available while \"deterministic\" is set");
    return autotune(t, (unsigned)value);
  } else if (key == "resident_min_walkers") {
    // smallest ensemble the resident ensemble kernel takes (default 24; smaller ones go through
    // one launch of single_draw_kernel, which is faster for them)
    TC_CHECK(value >= 2 && value <= 256, "resident_min_walkers must be in [2, 256]");
    t->resident.min_walkers = value;
  } else if (key == "resident_wait_us") {
    // how long a workgroup of the resident ensemble kernel waits for another one inside a call
    // before it gives up (default 20 000; tests use 1 to exercise that path)
    TC_CHECK(value >= 1 && value <= 1000000, "resident_wait_us must be in [1, 1000000]");
    const int status = resident_stop(t);
    if (status != TC_OK) return status;
    t->resident.wait_us = value;
  } else if (key == "resident_aperture") {
    // 1 (default): the resident ensemble kernel's mailbox lies in device memory (large-BAR
    // systems; the host writes it through the PCIe aperture); 0: in page-locked host memory
    TC_CHECK(value == 0 || value == 1, "resident_aperture must be 0 or 1");
    const int status = resident_stop(t);
    if (status != TC_OK) return status;
    t->tuning.resident_aperture = value;
    t->resident.ens_grid = 0;          // (the buffers are chosen again at the next call)
    t->resident.single_aperture_decided = false;
  } else if (key == "cross_min_draws") {
    TC_CHECK(value >= 1, "cross_min_draws must be positive");
    t->tuning.cross_min_draws = value;
  } else if (key == "cross_defer") {
    // developer A/B: 0 = predict_cross_fused_kernel runs every lane's node loop in place
    t->tuning.cross_defer = value != 0;
  } else if (key == "fused_defer") {
    // developer A/B: 0 = predict_fused_kernel runs the satellites' node loops in place
    TC_CHECK(value >= 0 && value <= 2, "fused_defer must be 0, 1 or 2");
    t->tuning.fused_defer = value;
  } else if (key == "fused_sat_cap") {
    TC_CHECK(value >= 0 && value <= 6, "fused_sat_cap must be in [0, 6]");
    t->tuning.fused_sat_cap = value;
  } else if (key == "cross_wide_min_draws") {
    // mode cross, tables of up to 16 rows: undecorated batches of this many draws take the
    // 32-row chunk form (launch.hip: choose_cross_fused); 0: never
    TC_CHECK(value >= 0, "cross_wide_min_draws must not be negative");
    t->tuning.cross_wide_min_draws = value;
  } else if (key == "cross_target") {
    // developer A/B: workgroups a launch of predict_cross_small_kernel should have at least
    TC_CHECK(value >= 1 && value <= 4096, "cross_target must be in [1, 4096]");
    t->tuning.cross_target = value;
  } else if (key == "series") {
    // bit 0: the node sum of an undecorated central bin by its moment expansion (csrc/series.h)
    // for every draw that allows it; bit 1: the binomial expansion of the satellite bins well
    // above a draw's M0; 0: always the node loops.  Per draw either way (a draw's bits do not
    // depend on its neighbours in the batch).
    TC_CHECK(value >= -1 && value <= 3, "series must be -1 .. 3");
    t->tuning.series = value;
  } else if (key == "grouped") {
    // 1 (default): bins that share their quadrature nodes -- the secondary-percentile bins of a
    // mass bin -- have the nodes' occupations evaluated once per group (kernels.hip.h:
    // occ_group_zheng07); 0: every bin by itself (A/B and tests: rounding-level differences
    // only in the sums over bins, none per bin)
    t->grouped = value != 0 && t->node_groups.largest > 1;
  } else if (key == "lanes") {
    TC_CHECK(value >= 1 && value <= (int)tc_table::kMaxLanes, "lanes must be in [1, %d]",
             (int)tc_table::kMaxLanes);
    for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
    for (int l = 0; l < value; ++l) {
      if (t->lanes[l].stream != nullptr) continue;
      TC_HIP(hipStreamCreateWithFlags(&t->lanes[l].stream, hipStreamNonBlocking));
      TC_HIP(hipEventCreateWithFlags(&t->lanes[l].finished, hipEventDisableTiming));
    }
    t->tuning.lanes = t->n_lanes = value;
    t->prev = -1;
  } else if (key == "ordered") {
    // 0 (default): every call only orders its own kernels (results land in the buffers the
    // caller passed; tc_table_synchronize and tc_comm_gather wait for every lane);
    // 1: the finalisations of consecutive device-pointer calls are chained by events, so
    // that a consumer waiting for the LAST call finds the earlier ones complete as well
    for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
    t->chain = value != 0;
    t->prev = -1;
  } else if (key == "sync_chunks") {
    // synchronous host calls (tc_predict_zheng07_batch, ...): 0 (default) cut batches of 2048
    // draws and more into up to 8 chunks whose uploads, kernels, downloads and host copies
    // overlap; N >= 1: N chunks for every batch (1: the same machinery, one piece); -1: the
    // serial path of rounds 1-4 (upload, kernels alone on one lane, download, copy)
    TC_CHECK(value >= -1 && value <= 64, "sync_chunks must be in [-1, 64]");
    t->tuning.sync_chunks = value;
  } else if (key == "sync_stagger") {
    // synchronous host calls with 16 MB of results and more run their chunks' kernels one after
    // the other, the chunks shrinking geometrically to this many per cent of the first
    // (default 10; 0: equal chunks on all lanes as for small results; predict_chunked)
    TC_CHECK(value >= 0 && value <= 100, "sync_stagger must be in [0, 100]");
    t->tuning.sync_stagger = value;
  } else if (key == "sync_form") {
    TC_CHECK(value == 0 || value == 32 || value == 64, "sync_form must be 0, 32 or 64");
    t->tuning.sync_form = value;
  } else if (key == "sync_direct_out") {
    TC_CHECK(value >= 0 && value <= 2, "sync_direct_out must be 0, 1 or 2");
    t->tuning.sync_direct_out = value;
  } else if (key == "async_direct_in" || key == "async_direct_out") {
    // asynchronous host calls: 1 = the kernels read the draws from / write the results to
    // the caller's page-locked buffers themselves; 0 = copy commands on the lane's stream
    // (async_direct_out 2: only the number densities directly, xi by copy command)
    (key == "async_direct_in" ? t->tuning.async_direct_in : t->tuning.async_direct_out) =
        value;
  } else if (key == "fused") {
    // 0: always the three-kernel path; 1 (default): one launch per batch
    // (predict_fused_kernel) for pipelined device-pointer and asynchronous calls of
    // "fused_min_draws" .. "fused_max_draws" draws that it covers; 2: for every such call, also
    // those that run alone
    TC_CHECK(value >= 0 && value <= 2, "fused must be 0, 1 or 2");
    t->tuning.fused = value;
  } else if (key == "fused_min_draws" || key == "fused_max_draws") {
    TC_CHECK(value >= (key == "fused_min_draws" ? 0 : 1), "%s must be positive", name);
    (key == "fused_min_draws" ? t->tuning.fused_min_draws : t->tuning.fused_max_draws) = value;
  } else if (key == "resident") {
    // 1: un-batched calls (one draw, total correlation function, Zheng07 family) are served by
    // ONE resident launch that takes them from a mailbox in page-locked memory (launch.hip:
    // resident_predict); it leaves when idle for "resident_idle_us" and before any other kind
    // of call on this handle.  0 (default): one launch per call.
    // 2 (default): by itself for loops of un-batched calls -- from the eighth call on that
    // follows its predecessor within 300 us, with an idle time of 250 us, backing off when the
    // caller's pauses or device-wide synchronisations keep ending the launch.
    TC_CHECK(value >= 0 && value <= 2, "resident must be 0, 1 or 2");
    t->resident.enabled = value == 1;
    t->resident.auto_mode = value == 2;
    t->resident.auto_streak = t->resident.auto_backoff = 0;
    t->resident.auto_window = t->resident.auto_relaunches = 0;
    t->resident.auto_failures = 0;
    t->resident.ens_disabled = false;
    t->resident.ens_failures = 0;
    if (value != 1) return resident_stop(t);
  } else if (key == "resident_inject_failures") {
    // test hook: the next `value` calls that reach the resident single-draw kernel fail as if
    // the kernel kept leaving (the automatic mode must then serve them by launches)
    TC_CHECK(value >= 0, "resident_inject_failures must not be negative");
    t->resident.inject_failures = value;
  } else if (key == "resident_auto_idle_us") {
    TC_CHECK(value >= 10 && value <= 100000, "resident_auto_idle_us must be in [10, 100000]");
    const int status = resident_stop(t);
    if (status != TC_OK) return status;
    t->resident.auto_idle_us = value;
  } else if (key == "resident_poll_waves") {
    TC_CHECK(value >= 1 && value <= 4, "resident_poll_waves must be in [1, 4]");
    const int status = resident_stop(t);
    if (status != TC_OK) return status;
    t->resident.poll_waves = value;
  } else if (key == "resident_idle_us") {
    TC_CHECK(value >= 10 && value <= 1000000, "resident_idle_us must be in [10, 1000000]");
    const int status = resident_stop(t);
    if (status != TC_OK) return status;
    t->resident.idle_us = value;
  } else if (key == "fused_draws") {
    // (40: the latency form -- launch.hip: fused_spread_eligible -- for every batch it serves)
    TC_CHECK(value == 0 || value == 32 || value == 40 || value == 64,
             "fused_draws must be 0, 32, 40 or 64");
    t->tuning.fused_draws = value;
  } else if (key == "fused_spread") {
    // 1 (default): calls that have the chip to themselves (host arrays, one lane, pipeline
    // off) take the latency form of the one-launch kernel where it serves them; 0: never
    t->tuning.fused_spread = value != 0;
  } else if (key == "fused_spread_min") {
    TC_CHECK(value >= 1, "fused_spread_min must be positive");
    t->tuning.fused_spread_min = value;
  } else if (key == "fused_spread_rounds") {
    TC_CHECK(value >= 1 && value <= 64, "fused_spread_rounds must be in [1, 64]");
    t->tuning.fused_spread_rounds = value;
  } else if (key == "fused_waves") {
    TC_CHECK(value == 0 || value == 8 || value == 16, "fused_waves must be 0, 8 or 16");
    t->tuning.fused_waves = value;
  } else if (key == "prio_fused") {
    t->tuning.prio_fused = value & 3;
  } else if (key == "prio_fused_occ") {
    t->tuning.prio_fused_occ = value & 3;
  } else if (key == "prio_fused_out") {
    t->tuning.prio_fused_out = value & 3;
  } else if (key == "single_draw") {
    t->tuning.single_draw = value != 0;
  } else if (key == "many_blocks") {
    TC_CHECK(value >= 2 && value <= 4096, "many_blocks must be in [2, 4096]");
    t->tuning.many_blocks = value;
  } else if (key == "single_round") {
    // (first table of an interpolator) 1: its un-batched call is sized so that all tables'
    // workgroups are on the chip at once; 0: one pass over the positions per workgroup
    t->tuning.single_round = value != 0;
  } else if (key == "poll_done") {
    // un-batched calls: 1 (default) poll the kernel's completion words in page-locked host
    // memory, 0 wait with hipStreamSynchronize
    t->tuning.poll_done = value != 0;
  } else if (key == "trace") {
    t->tuning.trace = value;
  } else if (key == "occ_splits" || key == "occ_per_cu" || key == "finalize_threads" ||
             key == "finalize_row_blocks") {
    // developer A/B: launch geometry of the occupation / finalisation kernels (0: chosen)
    TC_CHECK(value >= 0 && value <= 4096, "invalid value");
    (key == "occ_splits" ? t->tuning.occ_splits
     : key == "occ_per_cu" ? t->tuning.occ_per_cu
     : key == "finalize_threads" ? t->tuning.finalize_threads
                                 : t->tuning.finalize_row_blocks) = value;
  } else if (key == "prio_occ" || key == "prio_contract" || key == "prio_finalize") {
    // developer A/B: wave priorities (0..3) of the three kernels
    TC_CHECK(value >= 0 && value <= 3, "priorities are 0..3");
    (key == "prio_occ" ? t->tuning.prio_occ
                       : key == "prio_contract" ? t->tuning.prio_contract
                                                : t->tuning.prio_finalize) = value;
  } else if (key == "quad_merge" || key == "quad_waves" || key == "quad_order") {
    // developer A/B of the quadratic-form schedule: schedules are rebuilt on demand
    for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
    if (key == "quad_merge") t->tuning.quad_merge = value != 0;
    else if (key == "quad_order") t->tuning.quad_order = value;
    else t->tuning.quad_waves = t->tuning.quad_waves_f32 = value;
    for (tc::host::QuadTable* q : {&t->quad_by_type, &t->quad_total}) q->drop_schedules();
  } else {
    return fail(TC_ERR_INVALID, "unknown option '%s'", name);
  }
  return TC_OK;
}

int tc_table_timer_begin(tc_table* t, int profile_kernels) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_HIP(hipSetDevice(t->device));
  t->profile_kernels = profile_kernels != 0;
  t->profile_every = profile_kernels > 1 ? profile_kernels : 1;
  t->profile_launches = 0;
  t->kernel_events_used = 0;
  TC_HIP(hipEventRecord(t->ev_begin, t->stream));
  return TC_OK;
}

int tc_table_timer_end(tc_table* t, float* elapsed_ms) {
  TC_CHECK(t != nullptr && elapsed_ms != nullptr, "NULL argument");
  for (int l = 1; l < tc_table::kMaxLanes; ++l)
    if (t->lanes[l].stream) TC_HIP(hipStreamSynchronize(t->lanes[l].stream));
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
  for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
  const int64_t n = std::min<int64_t>(capacity, (int64_t)t->trace_blocks);
  TC_HIP(hipMemcpy(out, t->trace.ptr, (size_t)n * 6 * sizeof(uint64_t),
                   hipMemcpyDeviceToHost));
  return TC_OK;
}

int tc_debug_wave_trace(tc_table* t, uint64_t* out, int64_t capacity, int64_t* n_waves) {
  TC_CHECK(t != nullptr && n_waves != nullptr, "NULL argument");
  *n_waves = (int64_t)t->wave_trace_count;
  if (out == nullptr || t->wave_trace.ptr == nullptr) return TC_OK;
  for (tc_table::Lane& lane : t->lanes)
      if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
  const int64_t n = std::min<int64_t>(capacity, (int64_t)t->wave_trace_count);
  TC_HIP(hipMemcpy(out, t->wave_trace.ptr, (size_t)n * 6 * sizeof(uint64_t),
                   hipMemcpyDeviceToHost));
  return TC_OK;
}

int tc_debug_resident_ticks(tc_table* t, uint64_t* out, int64_t capacity, int64_t* n_blocks) {
  TC_CHECK(t != nullptr && n_blocks != nullptr, "NULL argument");
  *n_blocks = t->resident.mailbox.ptr != nullptr ? t->resident.blocks : 0;
  if (out == nullptr || t->resident.mailbox.ptr == nullptr) return TC_OK;
  const unsigned long long* words = (const unsigned long long*)t->resident.mailbox.ptr;
  for (int64_t b = 0; b < std::min<int64_t>(capacity, *n_blocks); ++b)
    out[b] = words[16 + kSingleMaxBlocks + b];
  return TC_OK;
}

// Phase stamps of workgroup 0 in the last call of the resident ensemble kernel (developer
// builds; 100 MHz ticks: call seen, occupation stored, the group's occupations seen, densities
// in LDS, quarters summed, partial sums stored, [6] shader cycles of the quarter sums, call
// finished), then three host times of
// that call in ns: published, every row combined (from its begin), time spent on the rows.
int tc_debug_ensemble_stamps(tc_table* t, uint64_t* out) {
  TC_CHECK(t != nullptr && out != nullptr, "NULL argument");
  TC_CHECK(t->resident.ens_mailbox.ptr != nullptr, "no ensemble call has been made");
  const unsigned long long* words = (const unsigned long long*)t->resident.ens_mailbox.ptr;
  const size_t offset = 8 + (size_t)tc::kEnsembleMaxWalkers * 8 + t->resident.ens_grid;
  for (int i = 0; i < 8; ++i) out[i] = words[offset + i];
  for (int i = 0; i < 3; ++i) out[8 + i] = t->resident.ens_host_ns[i];
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
