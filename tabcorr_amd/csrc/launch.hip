// Launch layer: chooses the decomposition of a batch, fills the kernels' argument blocks and
// hands them to the kernel instances, which live in translation units of their own
// (inst_quad.hip: the three-kernel path, inst_fused.hip / inst_cross.hip: the one-launch forms,
// inst_single.hip: un-batched calls; internal.h declares their entry points).
#include <immintrin.h>

#include <atomic>
#include <mutex>

#include "internal.h"
#include "kernels.hip.h"
#include "series.h"

namespace tc {
namespace host {

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
  // (behind the weights: every bin's sum of them, added in node order as the kernels would)
  std::vector<double> log_m((size_t)g * n_gauss), m((size_t)g * n_gauss),
      weight((size_t)g * n_gauss + g);
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
    double sum = 0.0;
    for (int k = 0; k < n_gauss; ++k) {
      weight[(size_t)i * n_gauss + k] = (double)(raw[k] / norm);
      sum += weight[(size_t)i * n_gauss + k];
    }
    weight[(size_t)g * n_gauss + i] = sum;
  }
  Quadrature q;
  q.n_gauss = n_gauss;
  int status = upload(log_m, &q.log_m);
  if (status == TC_OK) status = upload(m, &q.m);
  if (status == TC_OK) status = upload(weight, &q.weight);
  // moment expansion of the central bins' node sums (series.h)
  std::vector<double> series((size_t)g * tc::series::kStride + tc::series::kPad);
  std::vector<int32_t> series_thr((size_t)g * tc::series::kThresholds);
  for (int i = 0; i < g; ++i)
    tc::series::bin_consts(n_gauss, log_m.data() + (size_t)i * n_gauss,
                           weight.data() + (size_t)i * n_gauss, t->log_min[i], t->log_max[i],
                           series.data() + (size_t)i * tc::series::kStride,
                           series_thr.data() + (size_t)i * tc::series::kThresholds);
  if (status == TC_OK) status = upload(series, &q.series);
  if (status == TC_OK) status = upload(series_thr, &q.series_thr);
  if (n_gauss >= 4 && t->plan.n_central > 0) {
    // ... and a central bin's constants in one record (series.h, namespace cen_record)
    namespace rec = tc::series::cen_record;
    std::vector<double> records((size_t)t->plan.n_central * rec::kStride);
    for (int i = 0; i < t->plan.n_central; ++i)
      rec::bin_record(series.data() + (size_t)i * tc::series::kStride,
                      series_thr.data() + (size_t)i * tc::series::kThresholds,
                      weight[(size_t)g * n_gauss + i], log_m.data() + (size_t)i * n_gauss,
                      n_gauss, records.data() + (size_t)i * rec::kStride);
    if (status == TC_OK) status = upload(records, &q.cen_records);
  }
  std::vector<double> sat_series((size_t)g * tc::series::sat::kStride + tc::series::kPad);
  std::vector<int32_t> sat_series_thr((size_t)g * tc::series::sat::kThresholds);
  for (int i = 0; i < g; ++i)
    tc::series::sat::bin_consts(n_gauss, m.data() + (size_t)i * n_gauss,
                                weight.data() + (size_t)i * n_gauss, t->log_min[i],
                                t->log_max[i],
                                sat_series.data() + (size_t)i * tc::series::sat::kStride,
                                sat_series_thr.data() + (size_t)i * tc::series::sat::kThresholds);
  if (status == TC_OK) status = upload(sat_series, &q.sat_series);
  if (status == TC_OK) status = upload(sat_series_thr, &q.sat_series_thr);
  if (n_gauss >= 4) {
    // ... and a satellite bin's constants in one record (series.h, namespace sat_record), by
    // library bin (the centrals' slots stay empty)
    namespace rec = tc::series::sat_record;
    std::vector<double> records((size_t)g * rec::kStride);
    // (only where some expansion serves every bin: with none, all of a bin's pairs would be
    // deferred -- the node loops in place are faster then)
    bool served = g > t->plan.n_central;
    for (int i = t->plan.n_central; i < g; ++i)
      served = served &&
               sat_series_thr[(size_t)i * tc::series::sat::kThresholds + tc::series::sat::kShortest] != 0;
    for (int i = t->plan.n_central; i < g && served; ++i)
      rec::bin_record(sat_series.data() + (size_t)i * tc::series::sat::kStride,
                      sat_series_thr.data() + (size_t)i * tc::series::sat::kThresholds,
                      m.data() + (size_t)i * n_gauss, n_gauss,
                      records.data() + (size_t)i * rec::kStride);
    if (status == TC_OK && served) status = upload(records, &q.sat_records);
  }
  if (status == TC_OK && n_gauss == 10 &&
      (t->node_groups.largest > 1 || t->mode == TC_MODE_CROSS)) {
    // GROUPED kernels: the nodes of every group (= those of its first member), the weights and
    // their sums per member in group order
    const tc::NodeGroups& groups = t->node_groups;
    std::vector<double> g_log_m((size_t)groups.n_groups * n_gauss),
        g_m((size_t)groups.n_groups * n_gauss), g_weight((size_t)g * n_gauss + g);
    for (int i = 0; i < groups.n_groups; ++i) {
      const int first = groups.member[groups.begin[i]];
      for (int k = 0; k < n_gauss; ++k) {
        g_log_m[(size_t)i * n_gauss + k] = log_m[(size_t)first * n_gauss + k];
        g_m[(size_t)i * n_gauss + k] = m[(size_t)first * n_gauss + k];
      }
    }
    for (int mi = 0; mi < g; ++mi) {
      const int bin = groups.member[mi];
      for (int k = 0; k < n_gauss; ++k)
        g_weight[(size_t)mi * n_gauss + k] = weight[(size_t)bin * n_gauss + k];
      g_weight[(size_t)g * n_gauss + mi] = weight[(size_t)g * n_gauss + bin];
    }
    std::vector<double> g_series((size_t)g * tc::series::kStride + tc::series::kPad);
    std::vector<int32_t> g_series_thr((size_t)g * tc::series::kThresholds);
    std::vector<double> g_sat((size_t)g * tc::series::sat::kStride + tc::series::kPad);
    std::vector<int32_t> g_sat_thr((size_t)g * tc::series::sat::kThresholds);
    for (int mi = 0; mi < g; ++mi) {
      const int bin = groups.member[mi];
      std::copy_n(sat_series.data() + (size_t)bin * tc::series::sat::kStride,
                  tc::series::sat::kStride, g_sat.data() + (size_t)mi * tc::series::sat::kStride);
      std::copy_n(sat_series_thr.data() + (size_t)bin * tc::series::sat::kThresholds,
                  tc::series::sat::kThresholds,
                  g_sat_thr.data() + (size_t)mi * tc::series::sat::kThresholds);
      std::copy_n(series.data() + (size_t)bin * tc::series::kStride, tc::series::kStride,
                  g_series.data() + (size_t)mi * tc::series::kStride);
      std::copy_n(series_thr.data() + (size_t)bin * tc::series::kThresholds,
                  tc::series::kThresholds,
                  g_series_thr.data() + (size_t)mi * tc::series::kThresholds);
    }
    if (groups.largest <= 2) {
      // one record per group (series.h, namespace record)
      namespace rec = tc::series::record;
      std::vector<double> records((size_t)groups.n_groups * rec::kStride);
      for (int i = 0; i < groups.n_groups; ++i) {
        const size_t mi = groups.begin[i];
        const size_t mj = groups.begin[i + 1] - groups.begin[i] > 1 ? mi + 1 : mi;
        const bool central = i < groups.n_central_groups;
        rec::group_record(
            central,
            central ? g_series.data() + mi * tc::series::kStride
                    : g_sat.data() + mi * tc::series::sat::kStride,
            central ? g_series.data() + mj * tc::series::kStride
                    : g_sat.data() + mj * tc::series::sat::kStride,
            central ? g_series_thr.data() + mi * tc::series::kThresholds
                    : g_sat_thr.data() + mi * tc::series::sat::kThresholds,
            g_weight[(size_t)g * n_gauss + mi], g_weight[(size_t)g * n_gauss + mj],
            g_log_m.data() + (size_t)i * n_gauss, g_m.data() + (size_t)i * n_gauss, n_gauss,
            (int)mi, mj != mi, records.data() + (size_t)i * rec::kStride);
      }
      status = upload(records, &q.group_records);
    }
    if (status == TC_OK) status = upload(g_log_m, &q.group_log_m);
    if (status == TC_OK) status = upload(g_m, &q.group_m);
    if (status == TC_OK) status = upload(g_weight, &q.group_weight);
    if (status == TC_OK) status = upload(g_series, &q.group_series);
    if (status == TC_OK) status = upload(g_series_thr, &q.group_series_thr);
    if (status == TC_OK) status = upload(g_sat, &q.group_sat_series);
    if (status == TC_OK) status = upload(g_sat_thr, &q.group_sat_series_thr);
  }
  if (status != TC_OK) return status;
  t->quadrature[n_gauss] = q;
  *out = &t->quadrature[n_gauss];
  return TC_OK;
}


// Which expansions of series.h the occupation kernels take (option "series": bit 0 centrals,
// bit 1 satellites; -1, the default: by the table).  Every lane takes the path ITS draw asks
// for, so a wave that holds draws with sigma_logM below about a bin width runs the expansion
// AND the node loop of a central bin: with bins of 0.15 dex (the reference's example table)
// that is nearly every wave of a wide prior and costs 8 % of the step (19.6 against 18.1 us per
// 10^4 draws); with 0.09 dex (BASELINE configs[1]) every draw above sigma_logM = 0.08
// qualifies (39.4 against 40.0 us), with 0.012 dex (AbacusSummit) everything does (80 against
// 85 us).  So: the centrals' expansion for tables whose widest central bin is at most 0.1 dex.
int series_mask(const tc_table* t) {
  if (t->tuning.series >= 0) return t->tuning.series;
  double widest = 0.0;
  for (int g = 0; g < t->plan.n_central; ++g)       // (library order: centrals first)
    widest = std::max(widest, std::fabs(t->log_max[g] - t->log_min[g]));
  return widest <= 0.1 ? 1 : 0;
}

tc::GroupArgs group_args(const tc_table* t, const Quadrature& q) {
  tc::GroupArgs ga{};
  ga.begin = (const int32_t*)t->d_group_begin;
  ga.member = (const int32_t*)t->d_group_member;
  ga.log_m = (const double*)q.group_log_m;
  ga.m = (const double*)q.group_m;
  ga.weight = (const double*)q.group_weight;
  ga.n_h = (const double*)t->d_group_n_h;
  ga.percentile = (const double*)t->d_group_percentile;
  ga.series = (series_mask(t) & 1) ? (const double*)q.group_series : nullptr;
  ga.series_thr = (const int32_t*)q.group_series_thr;
  ga.sat_series = (series_mask(t) & 2) ? (const double*)q.group_sat_series : nullptr;
  ga.sat_series_thr = (const int32_t*)q.group_sat_series_thr;
  ga.records = (const double*)q.group_records;
  return ga;
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

int lds_bytes_for(const tc::Chunking& chunking, int rt, int elem) {
  int span = 1;
  while (span < chunking.waves_per_group) span <<= 1;
  // + 1: the row of ones the matrix kernel uses as "n_i" in mode cross
  return std::max(chunking.max_rows + 1, (span / 2) * rt) * 64 * elem;
}

// Waves of the contraction kernel one SIMD holds, from the kernels' register counts
// (FP64 matrix kernel: 54 ... 128 VGPRs growing with the r tile, 76 ... 160 with the
// interpolator's table loop; float32 kernel ~100).
int wave_slots(const tc_table* t, bool interp) {
  if (t->compute_dtype == TC_DTYPE_F32) return 4;
  const int rt = t->rt;
  if (interp) return rt <= 4 ? 6 : rt <= 8 ? 5 : rt <= 20 ? 4 : 3;
  return rt <= 8 ? 8 : rt <= 12 ? 7 : rt <= 16 ? 6 : rt <= 20 ? 5 : 4;
}

// Workgroups of the contraction kernel that fit on one CU: LDS (160 KiB) and wave slots.
int blocks_per_cu(int lds_bytes, int waves, int slots) {
  const int by_lds = kMaxLdsBytes / std::max(lds_bytes, 1);
  const int by_waves = slots * 4 / waves;
  return std::max(1, std::min(std::min(by_lds, by_waves), 8));
}

// Pick the decomposition for a batch.  Draw tiles alone rarely fill the chip (10^4
// draws are 157 tiles), so the table is additionally cut into groups of `waves` chunks.
// Candidates (4 or 8 waves per workgroup, 1..32 groups) are ranked by a small cost model
// of the busiest CU -- workgroups per CU x waves x (entries per wave + fixed overhead),
// penalised when fewer than four waves per SIMD are resident or when the workgroups need
// several scheduling rounds -- calibrated on per-workgroup timelines (tools/archive/trace.py):
// the main loop issues one FP64 VALU instruction per 4 cycles per SIMD as long as >= 4
// waves per SIMD are resident, and idle time comes from uneven workgroup counts per CU.
int choose_chunking(tc_table* t, int64_t n_draws, int tables_per_block,
                    DeviceChunking** out, int* lds_bytes) {
  const int64_t n_tiles = (n_draws + 63) / 64 * t->n_rtiles;
  const int elem = t->compute_dtype == TC_DTYPE_F32 ? 4 : 8;
  const int forced_groups = t->tuning.n_groups;
  const int forced_waves = t->tuning.n_waves;
  // (the developer overrides are part of the key: sweeps stay cheap on the host)
  const int64_t key =
      (n_tiles * 4096 + forced_groups * 64 + forced_waves) * 1024 + tables_per_block;
  {
    auto cached = t->choices.find(key);
    if (cached != t->choices.end()) {
      *out = cached->second;
      *lds_bytes = std::max(lds_bytes_for((*out)->host, t->rt, elem), t->tuning.lds_min);
      return TC_OK;
    }
  }
  const int64_t min_entries = t->tuning.min_chunk_entries;
  // fixed cost of a workgroup (staging, reduction) in units of table entries
  const double overhead_entries = 24.0;
  const int n_cus = t->n_cus;
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
      const int fit =
          blocks_per_cu(bytes + 1024, use_waves, wave_slots(t, tables_per_block > 1));
      const double blocks = (double)n_tiles * actual_groups;
      const double per_cu = std::ceil(blocks / n_cus);
      const double resident = std::min<double>(per_cu, fit) * use_waves / 4.0;
      const double rounds = std::ceil(per_cu / fit);
      double cost = per_cu * use_waves * (longest + overhead_entries);
      if (resident < 4.0) cost *= 4.0 / resident;
      cost *= 1.0 + 0.1 * (rounds - 1.0);
      if (tables_per_block > 1 && n_tiles >= 64) {
        // interpolator: workgroups are long (all tables of a split), so what counts is
        // the half-empty last scheduling round against the staging of finer groups
        // (measured, 25 tables: 196 tiles best with 16-20 groups, 1563 tiles with 8)
        cost = (rounds + 0.5) * std::min<double>(per_cu, fit) * use_waves *
               (longest + 14.0);
      }
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
  t->choices[key] = c;
  *out = c;
  *lds_bytes = std::max(lds_bytes_for(c->host, t->rt, elem), t->tuning.lds_min);
  return TC_OK;
}

// ---- quadratic-form path -------------------------------------------------------------------

void QuadTable::release() {
  for (void* p : {d_table, d_comps})
    if (p) (void)hipFree(p);
  d_table = d_comps = nullptr;
  drop_schedules();
}

void QuadTable::drop_schedules() {
  for (auto& kv : schedules)
    for (void* p : {kv.second->runs, kv.second->wave_runs, kv.second->wave_head,
                    kv.second->group_begin, kv.second->merge_range, kv.second->merges})
      if (p) (void)hipFree(p);
  schedules.clear();
}

int64_t max_slab(const tc_table* t) {
  // scalar offsets into the density buffer are 32 bits: n_bins * ldb * 8 < 2^32
  const int64_t by_offset = ((int64_t)0xff000000 / (8 * (int64_t)std::max(1, t->n_bins))) / 64 * 64;
  return std::max<int64_t>(64, std::min<int64_t>(kMaxSlab, by_offset));
}

int build_quad_table(tc_table* t, bool by_type, const void* matrix, int matrix_dtype,
                     QuadTable* out) {
  tc::build_quad_layout(t->n_bins, t->plan.n_central, by_type, out->layout);
  int status;
  if (t->compute_dtype == TC_DTYPE_F32) {
    std::vector<float> host;
    tc::fill_quad_table_f32(out->layout, t->plan.perm, t->n_r, t->n_pairs, matrix,
                            matrix_dtype == TC_DTYPE_F32, t->quad_tiling, host);
    out->rtile_bytes = (size_t)out->layout.n_units * 1024;
    out->bytes = host.size() * sizeof(float);
    status = upload(host, &out->d_table);
  } else {
    std::vector<double> host;
    tc::fill_quad_table(out->layout, t->plan.perm, t->n_r, t->n_pairs, matrix,
                        matrix_dtype == TC_DTYPE_F32, t->quad_tiling, host);
    const int up = (t->quad_tiling.n_u + 1) / 2;
    out->rtile_bytes = (size_t)out->layout.n_units * up * 1024;
    out->bytes = host.size() * sizeof(double);
    // (finite, and small enough that no sum of entry x density x density overflows while the
    // draw's pair-weight sum stays below 1e280 -- predict_fused_kernel's latency form)
    out->finite = true;
    for (const double value : host) out->finite = out->finite && std::fabs(value) <= 1e20;
    status = upload(host, &out->d_table);
  }
  if (out->rtile_bytes >= ((size_t)1 << 32) - (1 << 24))
    return fail(TC_ERR_UNSUPPORTED, "table with %d bins is too large for the quadratic-form "
                "kernel", t->n_bins);
  if (status != TC_OK) return status;
  std::vector<tc::QuadCompArgs> comps;
  for (const tc::QuadComp& comp : out->layout.comps) {
    tc::QuadCompArgs c;
    c.triangular = comp.triangular;
    c.i_bin0 = comp.i_bin0;
    c.j_bin0 = comp.j_bin0;
    c.n_cb = comp.n_cb;
    c.unit_base = (uint32_t)comp.unit_base;
    c.pad[0] = c.pad[1] = c.pad[2] = 0;
    comps.push_back(c);
  }
  return upload(comps, &out->d_comps);
}

int get_quad_schedule(tc_table* t, QuadTable* q, int64_t n_tiles, int n_tables, bool separate,
                      DeviceQuadSchedule** out) {
  const std::vector<int64_t> key = {n_tiles, (int64_t)n_tables, separate ? 1 : 0};
  auto it = q->schedules.find(key);
  if (it != q->schedules.end()) {
    *out = it->second.get();
    return TC_OK;
  }
  // One schedule per distinct number of draw tiles (~0.3 MB each): a caller sweeping the
  // batch size must not grow the cache without bound.
  constexpr size_t kMaxCachedSchedules = 64;
  if (q->schedules.size() >= kMaxCachedSchedules) {
    TC_HIP(hipDeviceSynchronize());      // (kernels in flight may still read them)
    q->drop_schedules();
  }
  // every SIMD gets `tuning.quad_waves` waves with equal shares of the matrix-core work;
  // small batches use fewer waves (at least 8 units = 16 n_u instructions each)
  // (float32: three -- the matrix instruction is half as long, a third wave covers more of the
  // issue gaps: configs[4] 2327 -> 2253 us per launch, 0.836 -> 0.864; float64: three waves are
  // no faster alone and 8 % slower in the pipeline)
  // float64, default: two for tables of BASELINE configs[1]'s size, three where a draw tile
  // carries more matrix work -- more bins or an interpolator's tables (configs[2] 125.6 -> 124.7
  // us per launch and +1.5 % per step, configs[3] 936 -> 921 us and +0.4 %)
  const int per_simd = t->compute_dtype == TC_DTYPE_F32 ? t->tuning.quad_waves_f32
                       : t->tuning.quad_waves > 0       ? t->tuning.quad_waves
                       : q->layout.n_units * std::max(1, n_tables) >= 1000 ? 3
                                                                           : 2;
  const int max_waves = t->n_cus * 4 * std::max(1, std::min(3, per_simd));
  tc::QuadSchedule schedule;
  // (interpolators: table-major order; one matrix larger than an L2: r-tile-major; see
  // hostmath.h)
  // (interpolators: the waves of an XCD share its K / 8 matrices -- all at once while those fit
  // its L2 next to the density rows, one after the other beyond ~3 MB: kQuadTableSync)
  // (from eight tables on: every XCD then owns whole tables.  BASELINE configs[3], 25 tables of
  // 0.83 MB: 937 -> 930 us per 12 500 draws and 4.08 -> 3.14 GB of fabric reads; the database's
  // 64 tables of 1.5 MB: 3547 -> 2730 us.  Tried and dropped: tile by tile inside every XCD's
  // tables -- consecutive waves share a tile's density rows but cycle through the XCD's three
  // matrices: 5.9 GB, 948 us; profiles/r04_notes.md)
  const bool many_matrices = n_tables >= 8;
  const int order = n_tables > 1 ? ((t->tuning.quad_order == tc::kQuadTableMajor ||
                                     t->tuning.quad_order == tc::kQuadTableSync)
                                        ? t->tuning.quad_order
                                    : many_matrices ? tc::kQuadTableSync
                                                    : tc::kQuadTableMajor)
                    : t->tuning.quad_order >= 0 ? t->tuning.quad_order
                    // (one matrix far beyond the L2s, BASELINE configs[4]: the waves of an XCD
                    // walk the same units for different draw tiles -- float64 7.3 ms draw-tile-
                    // major / 7.0 r-tile-major / 4.40 this way, against 4.76 for the segment
                    // kernel that served it until round 3; float32 2.23 -> 2.24 ms with 3.2 GB
                    // of fabric reads instead of 9.9)
                    : q->bytes > ((size_t)32 << 20) ? tc::kQuadUnitSync
                    : (t->compute_dtype == TC_DTYPE_F64 && t->quad_tiling.n_rtiles > 1 &&
                       q->bytes > ((size_t)4 << 20))
                        ? tc::kQuadRtileMajor
                        // (float32, BASELINE configs[4]: 2.42 ms draw-tile-major against 2.80 --
                        // half the bytes per flop, and 48 passes cost 13 slabs per group)
                        : tc::kQuadTileMajor;
  tc::build_quad_schedule(q->layout, (int)n_tiles, t->quad_tiling.n_rtiles, n_tables, separate,
                          max_waves, 8, schedule, order);
  // workgroup-level merging of the slabs: at most 12 LDS slots (60 KB for 20 r values) per
  // workgroup of kQuadWavesPerBlock waves, two workgroups per CU
  tc::QuadMergePlan merge;
  tc::merge_quad_schedule(q->layout, t->quad_tiling.n_rtiles, separate, tc::kQuadWavesPerBlock,
                          t->tuning.quad_merge ? 12 : 0, schedule, merge);
  std::unique_ptr<DeviceQuadSchedule> d(new DeviceQuadSchedule);
  d->n_waves = schedule.n_waves;
  d->lds_bytes = merge.lds_slots * 4 * t->quad_tiling.n_u *
                 (t->compute_dtype == TC_DTYPE_F32 ? tc::kQuadTileF32 * (int)sizeof(float)
                                                   : tc::kQuadTile * (int)sizeof(double));
  d->n_slabs = schedule.n_slabs;
  d->n_groups = schedule.n_groups;
  d->n_runs = (int)schedule.runs.size();
  int status = upload(schedule.runs, &d->runs);
  // The k-th share of the unit space goes to a wave on XCD k * 8 / n_waves: workgroup b
  // runs on XCD b % 8 and each XCD has its own L2, so the waves that walk one draw tile
  // (consecutive shares) read its density rows through ONE L2 instead of eight.  Per wave
  // the kernel gets [first run, end run).
  std::vector<int32_t> wave_range((size_t)schedule.n_waves * 2);
  std::vector<int32_t> merge_range((size_t)merge.n_blocks * 2, 0);
  {
    const int per_block = tc::kQuadWavesPerBlock;
    const bool by_xcd = t->n_xcds == 8 && schedule.n_waves % (8 * per_block) == 0;
    for (int share = 0; share < schedule.n_waves; ++share) {
      int wave = share;
      if (by_xcd) {
        const int per_xcd = schedule.n_waves / 8;
        const int xcd = share / per_xcd, local = share % per_xcd;
        const int block = xcd + 8 * (local / per_block);
        wave = block * per_block + local % per_block;
      }
      wave_range[2 * (size_t)wave] = schedule.wave_runs[share];
      wave_range[2 * (size_t)wave + 1] = schedule.wave_runs[share + 1];
      if (share % per_block == 0) {      // the workgroup of these shares
        const int block = wave / per_block, share_block = share / per_block;
        merge_range[2 * (size_t)block] = merge.block_begin[share_block];
        merge_range[2 * (size_t)block + 1] = merge.block_begin[share_block + 1];
      }
    }
  }
  // per wave: its range, its first run and that run's component in one 64-byte record
  std::vector<int32_t> wave_head((size_t)schedule.n_waves * 16, 0);
  for (int wave = 0; wave < schedule.n_waves; ++wave) {
    int32_t* head = wave_head.data() + (size_t)wave * 16;
    head[0] = wave_range[2 * (size_t)wave];
    head[1] = wave_range[2 * (size_t)wave + 1];
    if (head[0] >= head[1]) continue;
    static_assert(sizeof(tc::QuadRun) == 32 && sizeof(tc::QuadCompArgs) == 32, "record sizes");
    const tc::QuadRun& run = schedule.runs[(size_t)head[0]];
    memcpy(head + 2, &run, sizeof(run));
    const tc::QuadComp& comp = q->layout.comps[(size_t)run.comp];
    head[10] = comp.triangular;
    head[11] = comp.i_bin0;
    head[12] = comp.j_bin0;
    head[13] = comp.n_cb;
    head[14] = (int32_t)(uint32_t)comp.unit_base;
  }
  if (status == TC_OK) status = upload(wave_range, &d->wave_runs);
  if (status == TC_OK) status = upload(wave_head, &d->wave_head);
  if (status == TC_OK) status = upload(merge_range, &d->merge_range);
  if (status == TC_OK) status = upload(merge.merges, &d->merges);
  if (status == TC_OK) status = upload(schedule.group_begin, &d->group_begin);
  if (status != TC_OK) {
    for (void* p : {d->runs, d->wave_runs, d->wave_head, d->group_begin, d->merge_range,
                    d->merges})
      if (p) (void)hipFree(p);
    return status;
  }
  *out = d.get();
  q->schedules[key] = std::move(d);
  return TC_OK;
}

// Event pair of the next contraction launch while tc_table_timer_begin(profile = 1) is in
// effect, else NULLs.  The events are handed to hipExtLaunchKernelGGL, which stamps them
// with the dispatch's own begin / end -- the interval rocprofv3 --kernel-trace reports
// (tools/micro/event_timing.hip: 35.8 us against the profiler's 35.95 for a 35 us kernel;
// hipEventRecord before / after the launch reads 38.6).
int next_kernel_events(tc_table* t, hipEvent_t* start, hipEvent_t* stop) {
  *start = *stop = nullptr;
  if (!t->profile_kernels) return TC_OK;
  // (a pair of events costs a launch ~1.5 us on the queue: a short timed region samples)
  if (t->profile_every > 1 && t->profile_launches++ % (size_t)t->profile_every != 0) return TC_OK;
  if (t->kernel_events_used == t->kernel_events.size()) {
    hipEvent_t e0, e1;
    TC_HIP(hipEventCreate(&e0));
    TC_HIP(hipEventCreate(&e1));
    t->kernel_events.emplace_back(e0, e1);
  }
  *start = t->kernel_events[t->kernel_events_used].first;
  *stop = t->kernel_events[t->kernel_events_used].second;
  ++t->kernel_events_used;
  return TC_OK;
}

// Contraction + finalisation through the quadratic-form kernel (mode auto, float64).
static int run_contraction_quad(tc_table* t, int64_t n_draws, int64_t ldb, unsigned flags,
                                double* ngal_device, double* xi_device) {
  Range range("contraction + finalisation");
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  QuadTable* q = separate || t->quad_total.d_table == nullptr ? &t->quad_by_type
                                                              : &t->quad_total;
  const tc::QuadTiling& tiling = t->quad_tiling;
  const bool f32 = t->compute_dtype == TC_DTYPE_F32;
  DeviceQuadSchedule* schedule = nullptr;
  int status = get_quad_schedule(t, q, ldb / (f32 ? tc::kQuadTileF32 : tc::kQuadTile), 1,
                                 separate, &schedule);
  if (status != TC_OK) return status;
  tc_table::Lane& lane = t->lanes[t->cur];
  hipStream_t stream = lane.stream;
  const int rt = 4 * tiling.n_u;
  status = lane.partial.reserve(
      (size_t)schedule->n_slabs * rt *
          (f32 ? tc::kQuadTileF32 * sizeof(float) : tc::kQuadTile * sizeof(double)),
      stream);
  if (status != TC_OK) return status;

  tc::QuadArgs qa{};
  qa.nbuf = (const double*)lane.nbuf.ptr;
  qa.nbuf32 = (const float*)lane.nbuf32.ptr;
  qa.nbufs = nullptr;
  qa.ldb = ldb;
  qa.n_bins = t->n_bins;
  qa.table = q->d_table;
  qa.tables = nullptr;
  qa.table_class = nullptr;
  qa.coef = nullptr;
  qa.rtile_bytes = (uint32_t)q->rtile_bytes;
  qa.runs = (const tc::QuadRun*)schedule->runs;
  qa.comps = (const tc::QuadCompArgs*)q->d_comps;
  qa.wave_runs = (const int32_t*)schedule->wave_runs;
  qa.wave_head = (const int32_t*)schedule->wave_head;
  qa.n_waves = schedule->n_waves;
  qa.partial = lane.partial.ptr;
  qa.priority = t->tuning.prio_contract;
  qa.merge_range = (const int32_t*)schedule->merge_range;
  qa.merges = (const int32_t*)schedule->merges;
  qa.stamps = nullptr;
  if (t->tuning.trace) {
    // developer timeline of the last launch: 6 words per wave (tc_debug_wave_trace)
    t->wave_trace_count = (size_t)schedule->n_waves;
    status = t->wave_trace.reserve(t->wave_trace_count * 6 * sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    qa.stamps = (unsigned long long*)t->wave_trace.ptr;
  }

  hipEvent_t k0 = nullptr, k1 = nullptr;
  status = next_kernel_events(t, &k0, &k1);
  if (status != TC_OK) return status;
  status = f32 ? launch_contract_quad_f32(tiling.n_u, qa, schedule->lds_bytes, stream, k0, k1)
               : launch_contract_quad(tiling.n_u, false, qa, schedule->lds_bytes, stream, k0, k1);
  if (status != TC_OK) return status;
  t->last_workgroups = (schedule->n_waves + tc::kQuadWavesPerBlock - 1) / tc::kQuadWavesPerBlock;
  t->last_waves = tc::kQuadWavesPerBlock;
  t->last_splits = schedule->n_slabs;
  t->last_lds = 0;

  tc::FinalizeQuadArgs fa{};
  fa.partial = lane.partial.ptr;
  fa.group_begin = (const int32_t*)schedule->group_begin;
  fa.ngal_part = (const double*)lane.ngal2.ptr;
  fa.n_ngal_parts = lane.ngal_parts;
  fa.n_rtiles = tiling.n_rtiles;
  fa.r_per_tile = tiling.r_per_tile;
  fa.rt = rt;
  fa.groups_per_rtile = separate ? (int)q->layout.comps.size() : 1;
  fa.priority = t->tuning.prio_finalize;
  fa.n_comp = separate ? t->plan.n_components : 1;
  fa.n_r = t->n_r;
  fa.mode = t->mode;
  fa.ldb = ldb;
  fa.n_draws = n_draws;
  fa.ngal = ngal_device;
  fa.xi = xi_device;
  t->chi2_fused = false;
  if (t->fuse_chi2_out != nullptr && !separate && t->n_r <= tc::kFinalizeRows) {
    // the likelihood straight from the finalisation's LDS tile: no xi round trip, one
    // launch (and one gap in the lane's chain) less
    fa.chi2_data = t->fuse_chi2_data;
    fa.chi2 = t->fuse_chi2_out;
    fa.xi = nullptr;
    t->chi2_fused = true;
  }
  if (t->chain && t->prev >= 0 && t->prev != t->cur)
    TC_HIP(hipStreamWaitEvent(stream, t->lanes[t->prev].finished, 0));
  if (!t->tuning.skip_finalize) status = launch_finalize_quad(fa, t->tuning, stream, f32);
  if (status != TC_OK) return status;
  if (t->force_lane >= 0) {
    t->prev = -1;
  } else {
    // (the lane's `finished` event orders the next lane's finalisation behind this one; with
    // unordered finalisations nobody waits for it -- tc_comm_gather records its own -- and the
    // marker would only put a 7-25 us bubble between this finalisation and the lane's next
    // occupation kernel: profiles/r03_notes.md)
    if (t->chain) TC_HIP(hipEventRecord(lane.finished, stream));
    t->prev = t->cur;
  }
  return TC_OK;
}

// Contraction + finalisation of draws whose densities are already in nbuf / ngal2.
int run_contraction(tc_table* t, int64_t n_draws, int64_t ldb, unsigned flags,
                    double* ngal_device, double* xi_device) {
  if (t->quad) return run_contraction_quad(t, n_draws, ldb, flags, ngal_device, xi_device);
  Range range("contraction + finalisation");
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t->plan.n_components : 1;
  DeviceChunking* c = nullptr;
  int lds = 0;
  int status = choose_chunking(t, n_draws, 1, &c, &lds);
  if (status != TC_OK) return status;
  tc_table::Lane& lane = t->lanes[t->cur];
  hipStream_t stream = lane.stream;
  const int n_groups = (int)c->host.groups.size();
  const int r_stride = t->rt * t->n_rtiles;
  status = lane.partial.reserve(
      (size_t)n_groups * r_stride * ldb * sizeof(double), stream);
  if (status != TC_OK) return status;

  tc::ContractArgs ca{};
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
  ca.pos_off = (const int32_t*)t->d_pos_off;
  if (const int ring = t->tuning.trace) {
    // developer timelines: ring > 1 keeps the block records of the last `ring` launches
    // (tools/archive/occupancy.py) and skips the per-wave stamps
    const size_t blocks = (size_t)(ldb / 64) * n_groups * t->n_rtiles;
    t->trace_blocks = blocks * ring;
    status = t->trace.reserve(t->trace_blocks * 6 * sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    ca.trace = (unsigned long long*)t->trace.ptr + (t->trace_launches++ % ring) * blocks * 6;
    if (ring == 1) {
      t->wave_trace_count = blocks * c->host.waves_per_group;
      status = t->wave_trace.reserve(
          t->wave_trace_count * 6 * sizeof(unsigned long long), stream);
      if (status != TC_OK) return status;
      TC_HIP(hipMemsetAsync(t->wave_trace.ptr, 0,
                            t->wave_trace_count * 6 * sizeof(unsigned long long), stream));
      ca.wave_trace = (unsigned long long*)t->wave_trace.ptr;
    }
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
  ca.xcd_map = t->n_xcds == 8 && n_tiles >= 8;
  const int padded_tiles = ca.xcd_map ? (n_tiles + 7) / 8 * 8 : n_tiles;
  dim3 grid((unsigned)(padded_tiles * n_groups), 1, (unsigned)t->n_rtiles);
  dim3 block(64 * c->host.waves_per_group);
  if (lds > 64 * 1024) {
    status = set_lds_limit_rt(t->rt, lds);
    if (status != TC_OK) return status;
  }
  hipEvent_t k0 = nullptr, k1 = nullptr;
  status = next_kernel_events(t, &k0, &k1);
  if (status != TC_OK) return status;
  if (t->compute_dtype == TC_DTYPE_F32) {
    ca.pos_ij = (const int32_t*)t->d_pos_ij;
    status = launch_contract_f32(grid, block, lds, stream, ca, k0, k1);
    if (status != TC_OK) return status;
  } else {
    status = launch_contract_rt(t->rt, grid, block, lds, stream, ca, k0, k1);
    if (status != TC_OK) return status;
  }
  t->last_workgroups = n_tiles * n_groups * t->n_rtiles;
  t->last_waves = c->host.waves_per_group;
  t->last_splits = n_groups;
  t->last_lds = lds;

  tc::FinalizeArgs fa{};
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
  if (t->chain && t->prev >= 0 && t->prev != t->cur)
    TC_HIP(hipStreamWaitEvent(stream, t->lanes[t->prev].finished, 0));
  if (!t->tuning.skip_finalize) status = launch_finalize(fa, t->tuning, stream);
  if (status != TC_OK) return status;
  if (t->force_lane >= 0) {
    t->prev = -1;
  } else {
    // (the lane's `finished` event orders the next lane's finalisation behind this one; with
    // unordered finalisations nobody waits for it -- tc_comm_gather records its own -- and the
    // marker would only put a 7-25 us bubble between this finalisation and the lane's next
    // occupation kernel: profiles/r03_notes.md)
    if (t->chain) TC_HIP(hipEventRecord(lane.finished, stream));
    t->prev = t->cur;
  }
  return TC_OK;
}

int run_occupation(tc_table* t, const double* theta_device, int n_theta,
                   int64_t n_draws, int64_t ldb, int n_gauss, unsigned flags,
                   double* occupation_device, DeviceBuffer* nbuf, DeviceBuffer* ngal2,
                   hipStream_t stream, int* ngal_parts, DeviceBuffer* nbuf32) {
  Range range("occupation");
  tc_table::Lane& lane = t->lanes[t->cur];
  if (nbuf == nullptr) nbuf = &lane.nbuf;
  if (ngal2 == nullptr) ngal2 = &lane.ngal2;
  if (stream == nullptr) stream = lane.stream;
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  // Work items = (draw tile, range of bins); blocks of kOccWaves waves stride over them,
  // four per CU at most.  The per-draw setup, the table staging and the block-level sums
  // are per item, so batches that cover the chip anyway get the fewest bin ranges that
  // still give every CU an item (sustained rate for 157 tiles x 100 bins: 42.5 us per step
  // with 2 ranges, 43.4 with 5, 45.0 with 13, 44.6 with 1); small batches (latency, not
  // throughput) minimise rounds x (bins per wave per item + per-item overhead).
  const int64_t n_tiles = ldb / 64;
  // groups of bins that share their nodes (Zheng07 family, default n_gauss_prim): the work
  // items are ranges of groups
  const bool grouped = t->grouped && n_gauss == 10 && !(flags & TC_FLAG_LEAUTHAUD11);
  const int n_units = grouped ? t->node_groups.n_groups : t->n_bins;
  int splits = 1, grid_blocks = 1;
  {
    const int n_cus = t->n_cus;
    const int64_t slots = (int64_t)n_cus * 4;
    const int max_splits = (n_units + tc::kOccWaves - 1) / tc::kOccWaves;
    double best = 0.0;
    for (int trial = 1; trial <= max_splits; ++trial) {
      const int per_block = (n_units + trial - 1) / trial;
      if ((n_units + per_block - 1) / per_block != trial) continue;
      const int64_t items = n_tiles * trial;
      if (n_tiles >= 64) {
        // pipelined calls: throughput, the fewest ranges that give every CU an item; calls
        // that run alone on their lane (host-buffer API, pipeline off) wait for this kernel:
        // twice as many, shorter items (10^4 draws: 25 -> 20 us, host-to-host 158 -> 149 us;
        // in the pipeline they cost 0.7 us per step)
        const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1;
        splits = trial;
        grid_blocks = (int)std::min<int64_t>(items, slots);
        if (items >= (alone ? 2 : 1) * (int64_t)n_cus) break;
        continue;
      }
      const int64_t rounds = (items + slots - 1) / slots;
      const int bins_per_wave = (per_block + tc::kOccWaves - 1) / tc::kOccWaves;
      const double cost = (double)rounds * (bins_per_wave + 1.5);
      if (best == 0.0 || cost < best) {
        best = cost;
        splits = trial;
        grid_blocks = (int)std::min<int64_t>(items, slots);
      }
    }
    const int forced = t->tuning.occ_splits;
    if (forced > 0) {
      const int per_block = (n_units + forced - 1) / forced;
      splits = (n_units + per_block - 1) / per_block;
      grid_blocks = (int)std::min<int64_t>(
          n_tiles * splits, (int64_t)n_cus * std::max(1, t->tuning.occ_per_cu));
    }
  }
  // float copy of the densities for the float32 quadratic-form kernel: the lane's own, or
  // the caller's (interpolators: one per class of halo tables)
  if (nbuf32 == nullptr && t->quad && t->compute_dtype == TC_DTYPE_F32 && nbuf == &lane.nbuf)
    nbuf32 = &lane.nbuf32;
  const bool want_f32 = nbuf32 != nullptr;
  if (want_f32) {
    status = nbuf32->reserve((size_t)t->n_bins * ldb * sizeof(float), stream);
    if (status != TC_OK) return status;
  }
  status = nbuf->reserve((size_t)t->n_bins * ldb * sizeof(double), stream);
  if (status == TC_OK)
    status = ngal2->reserve((size_t)splits * 2 * ldb * sizeof(double), stream);
  if (status != TC_OK) return status;
  if (ngal_parts != nullptr) *ngal_parts = splits; else lane.ngal_parts = splits;
  tc::OccArgs oa{};
  oa.theta = theta_device;
  oa.n_theta = n_theta;
  oa.n_draws = n_draws;
  oa.ldb = ldb;
  oa.n_bins = t->n_bins;
  oa.n_central = t->plan.n_central;
  oa.n_gauss = n_gauss;
  oa.n_tiles = (int)n_tiles;
  oa.n_splits = splits;
  oa.flags = flags | ((unsigned)(t->tuning.prio_occ & 3) << 8);
  oa.split = 0.5;
  oa.log_m = (const double*)q->log_m;
  oa.m = (const double*)q->m;
  oa.weight = (const double*)q->weight;
  oa.n_h = (const double*)t->d_n_h;
  oa.percentile = (const double*)t->d_percentile;
  oa.perm = (const int32_t*)t->d_perm;
  oa.math_table = (const double*)t->d_math_table;
  oa.nbuf = (double*)nbuf->ptr;
  oa.nbuf32 = want_f32 ? (float*)nbuf32->ptr : nullptr;
  oa.ngal = (double*)ngal2->ptr;
  oa.occupation = occupation_device;
  oa.n_groups = t->node_groups.n_groups;
  oa.n_central_groups = t->node_groups.n_central_groups;
  oa.group = group_args(t, *q);
  oa.series = (series_mask(t) & 1) ? (const double*)q->series : nullptr;
  oa.series_thr = (const int32_t*)q->series_thr;
  oa.sat_series = (series_mask(t) & 2) ? (const double*)q->sat_series : nullptr;
  oa.sat_series_thr = (const int32_t*)q->sat_series_thr;
  // diagnosis only (developer builds, tools/archive/ab.sh): reuse the densities of the previous call
  static int occ_calls = 0;
  if (t->tuning.skip_occ && ++occ_calls > 8) return TC_OK;
  return launch_occupation(oa, flags, n_gauss, grouped, grid_blocks, stream);
}

// One launch per slab of draws (predict_fused_kernel): float64 quadratic form with one r tile,
// densities of 64 draws within the LDS.
bool fused_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags) {
  // (asynchronous host calls: one command per call in the lane's chain pays for any size from
  // the lower bound on -- 20 000 draws 93.7 -> 82.6 us, 40 000 176 -> 162 us per call)
  if (t->tuning.fused == 0) return false;
  // (a measured choice -- option "autotune" -- knows where the one-launch form stops paying)
  const bool tuned_flags = n_gauss == 10 && t->tuning.fused == 1 &&
                           t->tuning.fused_min_draws == 0 && t->autotuned.count(flags) != 0;
  const bool invariant = t->tuning.deterministic >= 2;     // (one form whatever the batch)
  if (n_draws > t->tuning.fused_max_draws && t->async_lane < 0 && !tuned_flags && !invariant)
    return false;
  if (!t->quad || t->compute_dtype != TC_DTYPE_F64 || t->quad_total.d_table == nullptr)
    return false;
  if (t->quad_tiling.n_rtiles != 1 || t->n_r > 20 || t->chain || t->tuning.trace) return false;
  const bool leauthaud = (flags & TC_FLAG_LEAUTHAUD11) != 0;
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  if (separate) {
    // cen-cen | two halves of cen-sat | sat-sat on the four waves of a 32-draw tile: both
    // galaxy types present (equal numbers of bins, as a TabCorr table has them, balance the
    // waves to within a block row); no likelihood of separated components
    const tc::QuadLayout& layout = t->quad_by_type.layout;
    if (t->quad_by_type.d_table == nullptr || layout.comps.size() != 3 ||
        t->fuse_chi2_out != nullptr)
      return false;
    for (const tc::QuadComp& comp : layout.comps)
      if (comp.n_units <= 0) return false;
  }
  // (the decorated variants are compiled for the reference's default n_gauss_prim only)
  if (!leauthaud && (flags & (TC_FLAG_ASSEMBIAS | TC_FLAG_MODULATE_WITH_CENOCC)) && n_gauss != 10)
    return false;
  if (t->quad_total.layout.comps.size() != 1 || !t->quad_total.layout.comps[0].triangular)
    return false;
  // Two workgroups of 8 waves per CU (up to 80 KB of LDS each: 104 bins) or not at all: larger
  // tables fit ONE workgroup per CU, whose phases no neighbour covers.  With 8 waves the three
  // kernels are then 8-13 % ahead, with 16 waves (eight parts of the units per tile, four waves
  // per SIMD again) level -- tools/archive/r03_fused_waves.py, 10^4 draws, us per step, three kernels /
  // 8 waves / 16 waves: G = 112 51.8 / 55.6 / 51.5, 128 64.2 / 69.2 / 64.3, 200 137.0 / 150.9 /
  // 138.7, 240 189.6 / 215.0 / 192.3; G = 100: 43.5 / 39.4 / 43.2 -- so beyond 104 bins the
  // one-launch form (16 waves, up to 160 KB: 248 bins) is taken only when forced.
  // the latency form for calls that have the chip to themselves (fused_spread_eligible)
  if (!invariant && fused_spread_eligible(t, n_draws, n_gauss, flags)) return true;
  const bool wide = fused_wide_tables(t, separate, n_gauss, flags);
  if (n_gauss < 1 || (!wide && fused_waves(t, separate, flags) == 0)) return false;
  // option "deterministic" = 2: every call the form covers takes it, alone on its lane or not,
  // one draw or a million (Leauthaud11 with modulate_with_cenocc included)
  if (invariant) return true;
  // a measured choice for this table and these flags (option "autotune") replaces the formula
  // below for the calls it was measured on: pipelined device-pointer and asynchronous calls
  if (n_gauss == 10 && t->tuning.fused == 1 && t->tuning.fused_min_draws == 0) {
    auto tuned = t->autotuned.find(flags);
    const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1;
    if (tuned != t->autotuned.end() && !alone) return tuned->second.form_for(n_draws) != 0;
  }
  // Smallest batch: a launch lasts as long as one workgroup does, whatever the batch, so the
  // one-launch form pays from the batch size on at which four lanes of such launches beat the
  // three kernels (which spread any batch over the whole chip).  Estimated duration of a
  // workgroup alone on its CU (us): 5 + 10 (G / 100) (n_gauss / 10) [occupations] + 60 (units /
  // 325) (U / 5) [matrix work]: 75 for BASELINE configs[1]'s table, 33 for the reference's
  // example table (G = 60), 27 for G = 100 with three r values.  Measured crossovers
  // (tools/archive/r03_fused_scan.py with N_PRIM / N_R): estimates up to 28 us win from 512 draws on
  // (G = 40: 5.8 against 11.0 us per call at 512 draws; G = 100 with three r values: 8.5 /
  // 12.0), longer workgroups from ~90 draws per estimated microsecond (G = 60: 3000-4000;
  // G = 100, R = 8: 4096; G = 80: 5000; G = 100, R = 19: 6500-7000) -- below that both forms
  // are bound by the host thread that queues them (10-13 us per call) and differ by noise.
  // calls that run alone on their lane (host-buffer API, one lane, pipeline off)
  const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1;
  {
    const tc::QuadLayout& layout = separate ? t->quad_by_type.layout : t->quad_total.layout;
    const double estimate = (5.0 + 10.0 * (t->n_bins / 100.0) * (n_gauss / 10.0) +
                             60.0 * ((double)layout.n_units / 325.0) * (t->quad_tiling.n_u / 5.0)) *
                            8.0 / (wide ? 16 : fused_waves(t, separate, flags));
    // Leauthaud11 (a Newton inverse of the stellar-to-halo mass relation per central node: the
    // occupations outweigh the matrix work and spread better over the chip as a kernel of their
    // own): tools/archive/r03_fused_leauthaud.py, us per step, three kernels / one launch of 64-draw
    // workgroups: G = 100: 4000 draws 57.3 / 88.7, 10^4 126.0 / 114.5; G = 60: 4000 38.4 / 49.9,
    // 10^4 73.0 / 65.2 (32-draw workgroups below 8192 draws: further down); with
    // modulate_with_cenocc (the inverse at the satellites' nodes too) never clearly ahead: 10^4
    // draws 183.4 / 188.1 and 106.0 / 106.0, 4000 draws in 32-draw workgroups 80.2 / 80.6 and
    // 51.5 / 45.9 -- only when forced.
    if (leauthaud && (flags & TC_FLAG_MODULATE_WITH_CENOCC) && t->tuning.fused < 2) return false;
    // Wide tables (eight waves x 32 draws; tools/archive/r03_fused_wide.py, us per step, three kernels /
    // one launch: G = 112: 1024 draws 14.9 / 18.6, 2048 21.4 / 19.1, 4096 27.2 / 20.3, 10^4 52.3 /
    // 50.2, 20 000 98.6 / 99.0; G = 200: 2048 38.6 / 49.1, 4096 61.2 / 54.6, 6144 90.1 / 81.6,
    // 10^4 137.3 / 134.8; separated + assembly bias: 4096 66.5 / 57.0, 10^4 146.9 / 140.8): from
    // 15 draws per bin on.
    const int64_t min_draws = t->tuning.fused_min_draws > 0 ? t->tuning.fused_min_draws
                              : wide                        ? 15 * (int64_t)t->n_bins
                              // (Leauthaud11 in 32-draw workgroups, tools/archive/r03_fused_leauthaud.py,
                              // three kernels / one launch: G = 60: 2000 draws 20.7 / 25.6, 4000
                              // 37.4 / 28.7; G = 100: 2000 33.9 / 39.6, 4000 56.5 / 50.3)
                              : leauthaud && fused_half_tiles(t, separate, n_draws, n_gauss, flags)
                                  ? 3000
                              : fused_half_tiles(t, separate, n_draws, n_gauss, flags)
                                  // (tools/archive/r03_fused_low.py, us per step, three kernels / one
                                  // launch: a step of small batches costs a fifth of the
                                  // estimate -- G = 40: 256 draws 14.0 / 5.6; G = 60: 256 8.7 /
                                  // 5.5, 1024 8.6 / 7.6; G = 100, R = 3: 256 13.4 / 6.3; G = 100,
                                  // R = 19: 1024 15.0 / 15.4, 1280 16.3 / 15.6, 2048 19.3 / 16.4)
                                  ? (estimate <= 50.0 ? 256 : 12 * (int64_t)t->n_bins)
                              : leauthaud                   ? 8192
                              : estimate <= 28.0            ? 512
                                                            : (int64_t)(90.0 * estimate);
    if (alone && t->tuning.fused < 2) {
      // A call that has the chip to itself (round 6).  The three kernels spread any batch over
      // the whole chip -- on BASELINE configs[1]'s table 26 us for 1024 draws, 66 us for 10^4:
      // ~22 us + 4.4 us per 1000 draws, scaled with the table's work per draw --, a one-launch
      // form lasts as long as ONE workgroup whatever the batch (0.56 / 0.95 of the estimate for
      // 32 / 64 draws: 42 and 71 us there) for as long as one round of workgroups covers it:
      // 32-draw workgroups up to 32 draws per CU, 64-draw ones from 40 to 64 draws per CU
      // (in between: the latency form above; tools/r06_latency.py: 4096 draws 36.8 / 41.7,
      // 6144: 46.0 / 41.8, 8192: 55.5 / 41.8; 12288: 73.9 / 71.5, 16384: 90.7 / 71.7 us).
      if (leauthaud) return false;      // (its occupations spread better as a kernel of their own)
      // (measured from 1024 draws on; below, both ways are a few launches' worth of latency)
      if (n_draws < 2048) return false;
      const double three = 22.0 + 0.0044 * (double)n_draws * estimate / 75.0;
      if (n_draws <= (int64_t)32 * t->n_cus && t->tuning.fused_draws == 0 && !wide &&
          fused_half_tiles(t, separate, n_draws, n_gauss, flags))
        return 0.56 * estimate < three;
      if (n_draws > (int64_t)40 * t->n_cus && n_draws <= (int64_t)64 * t->n_cus && !wide &&
          t->tuning.fused_draws == 0 && fused_waves(t, separate, flags) == 8)
        return 0.95 * estimate < three;
      return false;
    }
    if (n_draws < min_draws) return false;
  }
  return true;
}

bool batch_invariant_form(tc_table* t, int n_gauss, unsigned flags) {
  if (t->tuning.deterministic < 2) return false;
  if (t->mode == TC_MODE_CROSS && !t->cross_host.empty() && t->tuning.fused != 0) {
    int status = TC_OK;
    tc_table* self = t;
    const CrossFused& cf =
        *choose_cross_fused(&self, 1, &t->cross_fused, &t->cross_fused_wide, 1, flags, &status);
    return status == TC_OK && cross_fused_eligible(t, cf, 1, n_gauss, flags, true);
  }
  return fused_eligible(t, 1, n_gauss, flags);
}

// Rows of the LDS density array: whole blocks of four covering every row a component reads.
int fused_dens_rows(const tc_table* t, bool separate) {
  if (!separate) return 4 * t->quad_total.layout.comps[0].n_rb;
  int rows = 0;
  for (const tc::QuadComp& comp : t->quad_by_type.layout.comps)
    rows = std::max(rows, std::max(comp.i_bin0 + 4 * comp.n_rb, comp.j_bin0 + 4 * comp.n_cb));
  return (rows + 3) / 4 * 4;
}

int fused_lds_bytes(const tc_table* t, bool separate, int waves, int draws) {
  const int dens_rows = fused_dens_rows(t, separate);
  return (std::max(dens_rows * draws, tc::fused_slot_doubles(waves, draws)) +
          tc::fused_scratch_doubles(waves)) * 8;
}

// The latency form (predict_fused_kernel with 40 draws per workgroup, one workgroup per CU,
// v_mfma_f64_4x4x4): for a call that has the chip to itself.  The 64-draw workgroups of the
// throughput form put 10^4 draws on 157 of the 256 CUs (71 us alone on the chip, 0.37 of the
// FP64 peak -- four such launches in flight are what fills it); 250 workgroups of 40 draws reach
// every CU.  Undecorated Zheng07 with ten nodes, total correlation function (or its likelihood),
// bins evaluated one by one; batches of up to one workgroup per CU.
bool fused_spread_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags) {
  if (t->tuning.fused_spread == 0 || n_gauss != 10 || t->tuning.deterministic >= 2) return false;
  // (what every one-launch form of mode auto needs: the float64 quadratic-form layout of the
  // whole triangle, one r tile, no chained finalisations, no developer timeline)
  if (t->tuning.fused == 0 || !t->quad || t->compute_dtype != TC_DTYPE_F64 ||
      t->quad_total.d_table == nullptr || t->quad_tiling.n_rtiles != 1 || t->n_r > 20 ||
      t->chain || t->tuning.trace || t->quad_total.layout.comps.size() != 1 ||
      !t->quad_total.layout.comps[0].triangular)
    return false;
  if (flags & (TC_FLAG_SEPARATE_GAL_TYPE | TC_FLAG_ASSEMBIAS | TC_FLAG_MODULATE_WITH_CENOCC |
               TC_FLAG_LEAUTHAUD11))
    return false;
  if (t->grouped) return false;
  if (t->tuning.fused_draws != 0 && t->tuning.fused_draws != 40) return false;
  if (fused_lds_bytes(t, false, 8, 40) > kMaxLdsBytes) return false;
  if (t->tuning.fused_draws == 40) return true;          // (forced: any batch)
  // alone on the chip, between fused_spread_min and one workgroup per CU (the three kernels
  // spread smaller batches over the chip in less than a 40-draw workgroup's lifetime)
  const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1 ||
                     t->sync_spread;
  // (up to 32 draws per CU the 32-draw workgroups are shorter-lived)
  if (alone && !t->sync_spread && n_draws <= (int64_t)32 * t->n_cus &&
      t->tuning.fused_draws == 0 && fused_half_tiles(t, false, n_draws, n_gauss, flags))
    return false;
  return alone && n_draws >= t->tuning.fused_spread_min &&
         n_draws <= (int64_t)40 * t->n_cus * std::max(1, t->tuning.fused_spread_rounds);
}

// Workgroups of ONE 32-draw tile and eight waves (eight parts of the units; up to 80 KB of LDS,
// two per CU), Zheng07 family with the default n_gauss_prim.
// * Batches below 8192 draws of tables up to 104 bins: a workgroup lasts a quarter as long as
//   the 64-draw one and a batch has twice as many -- what batches need that do not fill the
//   chip's 512 places with four launches of 64-draw workgroups (tools/archive/r03_fused_half.py, us per
//   step, three kernels / 64 draws x 8 waves / 32 x 4 / 32 x 8: G = 100: 1024 draws 12.8 / 23.4 /
//   19.3 / 15.1, 2048 19.0 / 29.0 / 21.7 / 16.0, 4096 23.9 / 30.2 / 22.3 / 16.9, 6144 31.6 / 31.1 /
//   26.2 / 25.1, 10^4 43.5 / 39.4 / 41.0 / 41.4; G = 60: 1024 10.2 / 12.3 / 9.7 / 7.4, 4096 15.3 /
//   13.4 / 11.0 / 8.2, 6144 18.6 / 14.2 / 12.2 / 11.9, 10^4 21.5 / 17.7 / 19.0 / 19.5): once the
//   chip is full the 64-draw form's fixed costs per draw win by 4-7 %.
// * Tables of 105-208 bins, whose 64-draw workgroup does not fit half a CU: any batch size.
bool fused_half_tiles(const tc_table* t, bool separate, int64_t n_draws, int n_gauss,
                      unsigned flags) {
  // (option "deterministic" = 2: the shape must not depend on the batch size -- 64 draws per
  // workgroup unless 32 are forced)
  if (t->tuning.deterministic >= 2)
    return t->tuning.fused_draws == 32 && n_gauss == 10 &&
           (t->tuning.fused_waves == 0 || t->tuning.fused_waves == 8) &&
           fused_lds_bytes(t, separate, 8, 32) <= 80 * 1024;
  if (t->tuning.fused_draws == 0 && n_gauss == 10 && t->tuning.fused == 1 &&
      t->tuning.fused_min_draws == 0) {
    auto tuned = t->autotuned.find(flags);      // (measured: option "autotune")
    if (tuned != t->autotuned.end() && tuned->second.form_for(n_draws) != 0)
      return tuned->second.form_for(n_draws) == 32 &&
             fused_lds_bytes(t, separate, 8, 32) <= 80 * 1024;
  }
  // (a call that has the chip to itself: up to one 32-draw workgroup per CU INCLUSIVE -- 8192
  // draws on 256 CUs take 41.9 us this way, 49.5 in the latency form, 55 as three kernels)
  const bool alone = t->force_lane >= 0 || !t->tuning.pipeline || t->n_lanes == 1;
  if (t->tuning.fused_draws == 64 ||
      (t->tuning.fused_draws == 0 && n_draws >= 8192 && !(alone && n_draws <= (int64_t)32 * t->n_cus)))
    return false;
  if (n_gauss != 10) return false;
  if (t->tuning.fused_waves != 0 && t->tuning.fused_waves != 8) return false;
  return fused_lds_bytes(t, separate, 8, 32) <= 80 * 1024;
}

bool fused_wide_tables(const tc_table* t, bool separate, int n_gauss, unsigned flags) {
  if (n_gauss != 10) return false;
  if (t->tuning.fused_draws == 64 || t->tuning.fused_waves != 0) return false;
  return fused_lds_bytes(t, separate, 8, 64) > 80 * 1024 &&
         fused_lds_bytes(t, separate, 8, 32) <= 80 * 1024;
}

// Waves per workgroup of the one-launch form for this table: 8, 16, or 0 (does not fit).
int fused_waves(const tc_table* t, bool separate, unsigned flags) {
  const bool fits8 = fused_lds_bytes(t, separate, 8, 64) <= 80 * 1024;
  const bool fits16 = fused_lds_bytes(t, separate, 16, 64) <= 160 * 1024;
  if (t->tuning.fused_waves == 8 && fused_lds_bytes(t, separate, 8, 64) <= 160 * 1024) return 8;
  if (t->tuning.fused_waves == 16 && fits16) return 16;
  // (16 waves: level with the three kernels on the shapes swept by hand, so only when forced --
  // or when the table has been measured with THESE flags (option "autotune"): that choice then
  // decides; a measurement with other flags says nothing about this LDS footprint)
  return fits8 ? 8
               : (fits16 && (t->tuning.fused >= 2 || t->autotuned.count(flags) != 0 ||
                             t->tuning.deterministic >= 2)) ? 16 : 0;
}

int run_fused(tc_table* t, const double* theta_device, int n_theta, int64_t n_draws,
              int n_gauss, unsigned flags, double* ngal_device, double* xi_device) {
  Range range("occupation + contraction + finalisation (one launch)");
  tc_table::Lane& lane = t->lanes[t->cur];
  hipStream_t stream = lane.stream;
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const QuadTable& q_table = separate ? t->quad_by_type : t->quad_total;
  tc::FusedArgs fa{};
  fa.theta = theta_device;
  fa.n_theta = n_theta;
  fa.n_bins = t->n_bins;
  fa.n_central = t->plan.n_central;
  fa.n_gauss = n_gauss;
  fa.dens_rows = fused_dens_rows(t, separate);
  fa.separate = separate ? 1 : 0;
  const bool spread = fused_spread_eligible(t, n_draws, n_gauss, flags);
  const bool wide = !spread && fused_wide_tables(t, separate, n_gauss, flags);
  const bool half_tiles = !spread && !wide && fused_half_tiles(t, separate, n_draws, n_gauss, flags);
  const int waves = spread || wide || half_tiles ? 8 : fused_waves(t, separate, flags);
  const int draws = spread ? 40 : wide || half_tiles ? 32 : 64;
  const int n_parts = spread ? 8 : waves * 32 / draws;
  if (!separate) {
    const tc::QuadComp& comp = q_table.layout.comps[0];
    tc::triangle_parts(comp.n_rb, n_parts, fa.part_rb0, fa.part_cb0, fa.part_count);
    for (int part = 0; part < n_parts; ++part) {
      fa.part_triangular[part] = 1;
      fa.part_n_cb[part] = comp.n_cb;
      fa.part_i_row0[part] = comp.i_bin0;
      fa.part_j_row0[part] = comp.j_bin0;
      fa.part_unit_base[part] = (int)comp.unit_base;
    }
  } else {
    // first quarter of a tile's waves: cen-cen, the two middle quarters: cen-sat (whole units,
    // row-major), last quarter: sat-sat; equal shares of a component's units within its waves
    const int quarter = n_parts / 4;
    for (int part = 0; part < n_parts; ++part) {
      const int c = part < quarter ? 0 : part < 3 * quarter ? 1 : 2;
      const int first = c == 0 ? 0 : c == 1 ? quarter : 3 * quarter;
      const int pieces = c == 1 ? 2 * quarter : quarter, piece = part - first;
      const tc::QuadComp& comp = q_table.layout.comps[c];
      fa.part_triangular[part] = comp.triangular;
      fa.part_n_cb[part] = comp.n_cb;
      fa.part_i_row0[part] = comp.i_bin0;
      fa.part_j_row0[part] = comp.j_bin0;
      fa.part_unit_base[part] = (int)comp.unit_base;
      if (comp.triangular) {
        int32_t rb0[tc::kFusedMaxParts], cb0[tc::kFusedMaxParts], count[tc::kFusedMaxParts];
        tc::triangle_parts(comp.n_rb, pieces, rb0, cb0, count);
        fa.part_rb0[part] = rb0[piece];
        fa.part_cb0[part] = cb0[piece];
        fa.part_count[part] = count[piece];
      } else {
        const int64_t begin = comp.n_units * piece / pieces;
        const int64_t end = comp.n_units * (piece + 1) / pieces;
        fa.part_rb0[part] = (int)(begin / comp.n_cb);
        fa.part_cb0[part] = (int)(begin % comp.n_cb);
        fa.part_count[part] = (int)(end - begin);
      }
    }
  }
  fa.n_r = t->n_r;
  fa.priority = (t->tuning.prio_fused & 3) | ((t->tuning.prio_fused_occ & 3) << 2) |
                ((t->tuning.prio_fused_out & 3) << 4) |
                ((env_int_early("TC_FUSED_SKIP", 0) & 3) << 8) |  // (developer builds only)
                // (the latency form may add its row shares four sums at a time: FusedArgs)
                (q_table.finite ? 1 << 11 : 0);
  fa.n_draws = n_draws;
  fa.n_groups = t->node_groups.n_groups;
  fa.n_central_groups = t->node_groups.n_central_groups;
  fa.group = group_args(t, *q);
  fa.series = (series_mask(t) & 1) ? (const double*)q->series : nullptr;
  fa.series_thr = (const int32_t*)q->series_thr;
  fa.sat_series = (series_mask(t) & 2) ? (const double*)q->sat_series : nullptr;
  fa.sat_series_thr = (const int32_t*)q->sat_series_thr;
  fa.log_m = (const double*)q->log_m;
  fa.m = (const double*)q->m;
  fa.weight = (const double*)q->weight;
  fa.n_h = (const double*)t->d_n_h;
  fa.percentile = (const double*)t->d_percentile;
  fa.split = 0.5;
  fa.math_table = (const double*)t->d_math_table;
  fa.table = q_table.d_table;
  fa.table_bytes =
      (uint32_t)((size_t)q_table.layout.n_units * (size_t)((t->quad_tiling.n_u + 1) / 2) * 1024);
  fa.ngal = ngal_device;
  fa.xi = xi_device;
  t->chi2_fused = false;
  if (t->fuse_chi2_out != nullptr) {
    fa.chi2_data = t->fuse_chi2_data;
    fa.chi2 = t->fuse_chi2_out;
    fa.xi = nullptr;
    t->chi2_fused = true;
  }
  const int lds = fused_lds_bytes(t, separate, waves, draws);
  const dim3 grid((unsigned)((n_draws + draws - 1) / draws)), block(64 * waves);
#ifdef TC_DEVELOPER_KNOBS
  if (fa.chi2 == nullptr && env_int_early("TC_FUSED_STAMPS", 0) != 0) {
    // (32 slots of stamps per workgroup of the last launch: tc_debug_trace hands them out as 6-word
    // records; tools/r06_stamps.py)
    t->trace_blocks = ((size_t)grid.x * 32 + 5) / 6;
    status = t->trace.reserve(t->trace_blocks * 6 * sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    fa.chi2_data = (const double*)t->trace.ptr;
    fa.priority |= 1 << 10;
  }
  if (env_int_early("TC_FUSED_SAME_UNIT", 0) != 0) fa.priority |= 1 << 12;
#endif
  hipEvent_t k0 = nullptr, k1 = nullptr;
  status = next_kernel_events(t, &k0, &k1);
  if (status != TC_OK) return status;
  const bool assembias = (flags & TC_FLAG_ASSEMBIAS) != 0;
  const bool modulate = (flags & TC_FLAG_MODULATE_WITH_CENOCC) != 0;
  // the satellites' expansion + deferred pairs (predict_fused_kernel, SATDEFER): undecorated
  // Zheng07, ten nodes, eight waves x 64 draws, every satellite bin with an expansion -- whether
  // or not the centrals take theirs (the reference's example table, bins 0.15 dex wide: 18.2 ->
  // 17.5 us per 10^4 draws); option "series" = 0 switches every expansion off
  const bool sat_defer = t->tuning.fused_defer != 0 && !assembias && !modulate &&
                         !(flags & TC_FLAG_LEAUTHAUD11) && n_gauss == 10 && !wide &&
                         !half_tiles && waves == 8 && t->n_bins <= 256 &&
                         !(t->grouped && n_gauss == 10) && t->tuning.series != 0 &&
                         q->sat_series != nullptr && q->sat_records != nullptr;
  if (sat_defer) fa.sat_series = (const double*)q->sat_series;
  fa.sat_records = (const double*)q->sat_records;
  fa.cen_records = (const double*)q->cen_records;
  // (both galaxy types by their records, the centrals no expansion serves deferred as well,
  // where the centrals' expansion is on)
  const bool cen_defer = sat_defer && t->tuning.fused_defer >= 2 && (series_mask(t) & 1) != 0 &&
                         q->cen_records != nullptr;
  fa.sat_cap = t->tuning.fused_sat_cap;
  tc::host::FusedInstance instance{};
  instance.n_gauss = n_gauss == 10 ? 10 : 0;
  instance.assembias = assembias;
  instance.modulate = modulate;
  instance.leauthaud = (flags & TC_FLAG_LEAUTHAUD11) != 0;
  instance.waves = waves;
  instance.draws = draws;
  // bins that share their nodes (Zheng07 family, ten nodes): the GROUPED instances
  instance.grouped = t->grouped && n_gauss == 10 && !(flags & TC_FLAG_LEAUTHAUD11);
  instance.defer = cen_defer ? 2 : sat_defer ? 1 : 0;
  status = launch_fused_instance(instance, t->device, t->quad_tiling.n_u, grid, block, lds, stream,
                                 k0, k1, fa);
  if (status != TC_OK) return status;
  t->last_workgroups = (int)grid.x;
  t->last_waves = waves;
  t->last_splits = 0;
  t->last_lds = lds;
  t->prev = t->force_lane >= 0 ? -1 : t->cur;
  return TC_OK;
}

// ---- mode cross, one launch per batch -----------------------------------------------------

CrossFused* choose_cross_fused(tc_table* const* tables, int n_tables, CrossFused* narrow,
                               CrossFused* wide, int64_t n_draws, unsigned flags, int* status) {
  const tc_table* t0 = tables[0];
  *status = TC_OK;
  if (!narrow->tried) *status = build_cross_fused(tables, n_tables, narrow);
  if (*status != TC_OK || narrow->rows == 0 || narrow->rows > tc::kCrossSmallRows) return narrow;
  // Up to 16 rows.  Undecorated batches whose groups have records (series.h) take the chunk
  // form with 32 rows -- products on the matrix pipe, expansions from the records, deferred
  // pairs: 52.4 against 58.8 us per 10^4 draws of the reference's AbacusSummit table (13 rows)
  // -- from cross_wide_min_draws draws on; below, the register form's several workgroups per
  // tile win.
  const bool undecorated =
      !(flags & (TC_FLAG_ASSEMBIAS | TC_FLAG_MODULATE_WITH_CENOCC | TC_FLAG_LEAUTHAUD11));
  // (option "deterministic" = 2: the form must not depend on the batch size)
  if (t0->tuning.deterministic >= 2) return narrow;
  if (!undecorated || t0->tuning.cross_defer == 0 || (series_mask(t0) & 1) == 0 ||
      t0->node_groups.largest > 2 || t0->tuning.cross_wide_min_draws <= 0 ||
      n_draws < t0->tuning.cross_wide_min_draws)
    return narrow;
  if (!wide->tried) *status = build_cross_fused(tables, n_tables, wide, true);
  return *status == TC_OK && wide->rows > 0 ? wide : narrow;
}

int build_cross_fused(tc_table* const* tables, int n_tables, CrossFused* cf, bool wide) {
  cf->tried = true;
  cf->rows = 0;
  cf->n_tables = n_tables;
  const tc_table* t0 = tables[0];
  if (t0->mode != TC_MODE_CROSS || t0->compute_dtype != TC_DTYPE_F64) return TC_OK;
  const int per_table = t0->n_r + 1;
  if ((int64_t)n_tables * per_table > tc::kCrossMaxRows) return TC_OK;
  for (int k = 0; k < n_tables; ++k) {
    const tc_table* t = tables[k];
    // the tables share the nodes and the weights of every bin (n_h may differ: it is folded
    // into the rows)
    if (t->cross_host.empty() || t->n_bins != t0->n_bins || t->n_r != t0->n_r ||
        t->plan.perm != t0->plan.perm || t->log_min != t0->log_min ||
        t->log_max != t0->log_max || t->percentile != t0->percentile ||
        t->dist_index != t0->dist_index || t->legacy != t0->legacy)
      return TC_OK;
  }
  // (instances: 2, 4, 8, 16 rows per wave)
  const int rows = n_tables * per_table;
  const int instance = rows <= 16 && !wide ? 16 : rows <= 32 ? 32 : rows <= 64 ? 64 : 128;
  const tc::NodeGroups& groups = t0->node_groups;
  if (groups.largest > tc::kCrossChunkBins) return TC_OK;
  // chunks: whole groups, at most kCrossChunkBins bins, centrals and satellites apart
  std::vector<int32_t> chunk_group{0};
  cf->n_central_chunks = 0;
  for (int g = 0, members = 0; g < groups.n_groups; ++g) {
    const int size = groups.begin[g + 1] - groups.begin[g];
    if (g > chunk_group.back() &&
        (members + size > tc::kCrossChunkBins || g == groups.n_central_groups)) {
      chunk_group.push_back(g);
      members = 0;
    }
    if (g == groups.n_central_groups) cf->n_central_chunks = (int)chunk_group.size() - 1;
    members += size;
  }
  if (groups.n_central_groups == groups.n_groups) cf->n_central_chunks = (int)chunk_group.size();
  chunk_group.push_back(groups.n_groups);
  cf->n_chunks = (int)chunk_group.size() - 1;
  cf->chunk_group_host = chunk_group;
  std::vector<double> host((size_t)t0->n_bins * instance, 0.0);
  for (int mi = 0; mi < t0->n_bins; ++mi) {
    const int g = groups.member[mi];
    for (int k = 0; k < n_tables; ++k) {
      const tc_table* t = tables[k];
      double* row = host.data() + (size_t)mi * instance + (size_t)k * per_table;
      for (int r = 0; r < t->n_r; ++r) row[r] = t->cross_host[(size_t)g * t->n_r + r] * t->n_h[g];
      row[t->n_r] = t->n_h[g];
    }
  }
  // predict_cross_fused_kernel (more than 16 rows) multiplies on the matrix pipe: its A operands
  // per (chunk, 4-bin step, block of 16 rows) -- kernel_args.h: CrossFusedArgs::rows
  std::vector<int32_t> chunk_block(cf->n_chunks + 1, 0);
  if (instance > tc::kCrossSmallRows) {
    const int row_blocks = instance / 16;
    for (int c = 0; c < cf->n_chunks; ++c) {
      const int bins = groups.begin[chunk_group[c + 1]] - groups.begin[chunk_group[c]];
      chunk_block[c + 1] = chunk_block[c] + (bins + 3) / 4;
    }
    std::vector<double> operands((size_t)chunk_block[cf->n_chunks] * row_blocks * 64, 0.0);
    for (int c = 0; c < cf->n_chunks; ++c) {
      const int m0 = groups.begin[chunk_group[c]], m1 = groups.begin[chunk_group[c + 1]];
      for (int mi = m0; mi < m1; ++mi) {
        const int step = chunk_block[c] + (mi - m0) / 4, k = (mi - m0) % 4;
        for (int row = 0; row < instance; ++row)
          operands[((size_t)step * row_blocks + row / 16) * 64 + k * 16 + row % 16] =
              host[(size_t)mi * instance + row];
      }
    }
    host.swap(operands);
    // (where a member bin's blocks of 16 rows start: the deferred pairs' coefficients)
    std::vector<int32_t> bin_operand(t0->n_bins, 0);
    for (int c = 0; c < cf->n_chunks; ++c) {
      const int m0 = groups.begin[chunk_group[c]], m1 = groups.begin[chunk_group[c + 1]];
      for (int mi = m0; mi < m1; ++mi)
        bin_operand[mi] = ((chunk_block[c] + (mi - m0) / 4) * row_blocks) * 64 + ((mi - m0) % 4) * 16;
    }
    const int uploaded = upload(bin_operand, &cf->d_bin_operand);
    if (uploaded != TC_OK) return uploaded;
  }
  int status = upload(host, &cf->d_rows);
  if (status == TC_OK) status = upload(chunk_group, &cf->d_chunk_group);
  if (status == TC_OK) status = upload(chunk_block, &cf->d_chunk_block);
  if (status != TC_OK) return status;
  cf->rows = instance;
  return TC_OK;
}

namespace {
// LDS layout of predict_cross_fused_kernel (kernel_args.h): offsets in doubles, total in bytes.
struct CrossLds {
  int res0 = 0, tile = 0, bitmap = 0, list = 0, bytes = 0;
};
// `defer_groups` > 0: the deferred pairs' bitmap (one word per group) behind the buffers; their
// lists (8 waves x 512 words = 2048 doubles) lie in the buffers' area behind the row sums, which
// take rows x 64 of its 2 x 48 x 64 doubles once the chunks are done (up to 64 rows).
CrossLds cross_lds_layout(const CrossFused& cf, int n_r, bool separate, int defer_groups = 0) {
  CrossLds lds;
  int end = tc::kCrossTableDoubles + tc::cross_buffer_doubles(cf.rows);
  if (defer_groups > 0) {
    lds.list = tc::kCrossTableDoubles + cf.rows * 64;
    lds.bitmap = end;
    end += (defer_groups + 1) / 2 * 2;
  }
  if (separate) {
    lds.res0 = end;
    end += cf.rows * 64;
  }
  const int tile = cf.n_tables * 64 + (separate ? 2 : 1) * n_r * 65;
  if (tile <= tc::kCrossTableDoubles) {
    lds.tile = 0;
  } else {
    lds.tile = end;
    end += tile;
  }
  lds.bytes = end * 8;
  return lds;
}
}  // namespace

bool cross_fused_eligible(const tc_table* t0, const CrossFused& cf, int64_t n_draws, int n_gauss,
                          unsigned flags, bool alone) {
  if (cf.rows == 0 || t0->tuning.fused == 0 || n_gauss != 10 || t0->chain || t0->tuning.trace)
    return false;
  if (flags & TC_FLAG_LEAUTHAUD11) return false;
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const tc::NodeGroups& groups = t0->node_groups;
  if (separate && (groups.n_central_groups == 0 || groups.n_central_groups == groups.n_groups ||
                   t0->fuse_chi2_out != nullptr))
    return false;
  if (cf.rows > tc::kCrossSmallRows &&
      cross_lds_layout(cf, t0->n_r, separate).bytes > kMaxLdsBytes - 256)
    return false;
  if (t0->fuse_chi2_out != nullptr && t0->n_r > 32) return false;
  // A workgroup carries 64 draws through the bins; batches of few tiles get several workgroups
  // per tile (run_cross_fused), so that the one-launch forms pay from a few hundred draws on.
  // (tools/r04_cross_scan.py, the reference's AbacusSummit table, us per step one launch / three
  // kernels: 4096 draws 48.6 / 44.0, 6144 47.4 / 59.0, 10^4 57.9 / 90.7, 32768 185 / 290)
  // The register form (<= 16 rows) gives a tile several workgroups below ~120 tiles and wins
  // from the smallest batches on (256 draws 8.7 us against 13.4 for the three kernels, 1024: 9.3 /
  // 19.4, 4096: 25.6 / 55.7).
  if (t0->tuning.deterministic >= 2) return true;        // (one form whatever the batch)
  const int64_t min_draws = t0->tuning.fused_min_draws > 0 ? t0->tuning.fused_min_draws
                            : cf.rows <= tc::kCrossSmallRows ? 192
                                                             : t0->tuning.cross_min_draws;
  if (n_draws < min_draws) return false;
  return t0->tuning.fused >= 2 || !alone;
}

int run_cross_fused(tc_table* t0, const CrossFused& cf, const tc::CrossFusedArgs* interp,
                    const double* theta_device, int n_theta, int64_t n_draws, unsigned flags,
                    double* ngal_device, double* xi_device, hipStream_t stream,
                    DeviceBuffer* partial, DeviceBuffer* counters) {
  Range range("occupation + contraction + finalisation (mode cross, one launch)");
  Quadrature* q = nullptr;
  int status = get_quadrature(t0, 10, &q);
  if (status != TC_OK) return status;
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  tc::CrossFusedArgs ca{};
  if (interp != nullptr) ca = *interp;
  ca.theta = theta_device;
  ca.n_theta = n_theta;
  ca.n_draws = n_draws;
  ca.n_bins = t0->n_bins;
  ca.n_groups = t0->node_groups.n_groups;
  ca.n_central_groups = t0->node_groups.n_central_groups;
  ca.group = group_args(t0, *q);
  ca.math_table = (const double*)t0->d_math_table;
  ca.rows = (const double*)cf.d_rows;
  ca.n_tables = cf.n_tables;
  ca.n_r = t0->n_r;
  ca.separate = separate ? 1 : 0;
  ca.chunk_group = (const int32_t*)cf.d_chunk_group;
  ca.chunk_block = (const int32_t*)cf.d_chunk_block;
  ca.n_chunks = cf.n_chunks;
  ca.n_central_chunks = cf.n_central_chunks;
  // The deferred pairs (kernel_args.h): undecorated, 17 .. 64 rows, the centrals' expansion on
  // -- the satellites' then comes with it whatever bit 1 of "series" says: its cost on wide
  // priors was the node loop a wave ran NEXT to it for a single draw, which is what goes away.
  const bool small = cf.rows <= tc::kCrossSmallRows;
  const bool defer = t0->tuning.cross_defer != 0 && !small && cf.rows <= 64 &&
                     cf.d_bin_operand != nullptr &&
                     !(flags & (TC_FLAG_ASSEMBIAS | TC_FLAG_MODULATE_WITH_CENOCC)) &&
                     (series_mask(t0) & 1) != 0 && q->group_sat_series != nullptr &&
                     q->group_records != nullptr &&
                     tc::cross_buffer_doubles(cf.rows) >= cf.rows * 64 + 2048;
  CrossLds layout = cross_lds_layout(cf, t0->n_r, separate, defer ? ca.n_groups : 0);
  if (defer && layout.bytes > kMaxLdsBytes / 2 - 256) {
    // (not at the price of the second workgroup per CU)
    layout = cross_lds_layout(cf, t0->n_r, separate);
    ca.defer = 0;
  } else {
    ca.defer = defer ? 1 : 0;
  }
  if (ca.defer) {
    ca.group.sat_series = (const double*)q->group_sat_series;
    ca.lds_bitmap = layout.bitmap;
    ca.lds_list = layout.list;
    ca.bin_operand = (const int32_t*)cf.d_bin_operand;
  }
  ca.lds_res0 = layout.res0;
  ca.lds_tile = layout.tile;
  ca.row_stride = cf.rows;
  if (cf.rows <= tc::kCrossSmallRows) {
    layout.bytes = tc::cross_small_lds_doubles(separate ? 2 : 1) * 8;
    if (ca.defer) {
      // (the bitmap behind the sums; the lists lie in the stage, which is free by then)
      ca.lds_bitmap = tc::cross_small_lds_doubles(separate ? 2 : 1);
      layout.bytes += (ca.n_groups + 1) / 2 * 2 * 8;
    }
    if (separate) {
      // waves for the groups of centrals in proportion to their cost (a node of the centrals
      // takes ~21 instructions, one of the satellites ~33)
      const double cen = 21.0 * ca.n_central_groups,
                   sat = 33.0 * (ca.n_groups - ca.n_central_groups);
      ca.cen_waves = std::max(1, std::min(tc::kCrossWaves - 1,
                                          (int)std::lround(tc::kCrossWaves * cen / (cen + sat))));
    }
  }
  ca.interp = interp != nullptr ? 1 : 0;
  ca.split = 0.5;
  ca.priority = (t0->tuning.prio_fused_occ & 3) | ((t0->tuning.prio_fused_out & 3) << 4);
  ca.ngal = ngal_device;
  ca.xi = xi_device;
  t0->chi2_fused = false;
  if (t0->fuse_chi2_out != nullptr) {
    ca.chi2_data = t0->fuse_chi2_data;
    ca.chi2 = t0->fuse_chi2_out;
    ca.xi = nullptr;
    t0->chi2_fused = true;
  }
  const int lds = layout.bytes;
  const int64_t n_tiles = (n_draws + 63) / 64;
  // Medium batches of the register form (<= 16 rows): a workgroup carries 64 draws through ALL
  // bins, so few tiles leave most CUs idle -- several workgroups per tile, each with a share of
  // the groups (by cost: a node of the centrals ~21 instructions, of the satellites ~33), the
  // last to arrive adds the shares.  About two workgroups per CU over four lanes in flight.
  // (option "deterministic" = 2: ONE workgroup per tile whatever the batch -- the shares of a
  // tile's groups, and with them the order of a draw's sums, follow the number of tiles)
  const int cross_target = t0->tuning.deterministic >= 2 ? 0
                           : t0->sync_cross_target > 0   ? t0->sync_cross_target
                                                         : t0->tuning.cross_target;
  int n_splits = 1;
  if (cf.rows <= tc::kCrossSmallRows) {
    // (tools/r04_cross_splits.py, AbacusSummit table, us per call, one workgroup per tile /
    // this rule: 256 draws 36.9 / 8.7, 1024 43.3 / 9.3, 4096 49.4 / 25.6, 6144 48.1 / 37.1,
    // 8192 47.7, 10^4 59.0 -- from ~120 tiles on a second workgroup per tile only costs)
    n_splits = (int)std::min<int64_t>(
        tc::kCrossMaxSplits,
        std::max<int64_t>(1, (2 * cross_target + n_tiles) / (2 * n_tiles)));
    const int n_cen = ca.n_central_groups, n_sat = ca.n_groups - ca.n_central_groups;
    n_splits = std::max(1, std::min(n_splits, std::min(std::max(n_cen, 1), std::max(n_sat, 1))));
    const double cost_cen = 21.0, cost_sat = 33.0;
    const double total_cost = cost_cen * n_cen + cost_sat * n_sat;
    for (int k = 0; k <= n_splits; ++k) {
      ca.split_cen[k] = (int)((int64_t)n_cen * k / n_splits);
      ca.split_sat[k] = n_cen + (int)((int64_t)n_sat * k / n_splits);
      // all groups, equal cost per split: centrals first
      const double target = total_cost * k / n_splits;
      ca.split_all[k] = target <= cost_cen * n_cen
                            ? (int)std::lround(target / cost_cen)
                            : n_cen + (int)std::lround((target - cost_cen * n_cen) / cost_sat);
    }
    ca.split_all[0] = 0;
    ca.split_all[n_splits] = ca.n_groups;
    ca.n_splits = n_splits;
    if (n_splits > 1) {
      const size_t count = (size_t)(separate ? 2 : 1) * tc::kCrossSmallRows * 64;
      status = partial->reserve((size_t)n_tiles * n_splits * count * sizeof(double), stream);
      if (status != TC_OK) return status;
      const size_t had = counters->bytes;
      status = counters->reserve((size_t)n_tiles * sizeof(int), stream);
      if (status != TC_OK) return status;
      // (zero once: the last workgroup of a tile resets its counter)
      if (counters->bytes != had) TC_HIP(hipMemsetAsync(counters->ptr, 0, counters->bytes, stream));
      ca.partial = (double*)partial->ptr;
      ca.counters = (int*)counters->ptr;
    }
  } else {
    // the chunked form likewise: ranges of chunks of about equal cost (per group a node
    // evaluation, per member bin its row FMAs)
    n_splits = (int)std::min<int64_t>(
        tc::kCrossMaxSplits,
        std::max<int64_t>(1, (2 * cross_target + n_tiles) / (2 * n_tiles)));
    n_splits = std::max(1, std::min(n_splits, cf.n_chunks));
    if (n_splits > 1) {
      const tc::NodeGroups& groups = t0->node_groups;
      std::vector<double> cost(cf.n_chunks + 1, 0.0);
      for (int c = 0; c < cf.n_chunks; ++c) {
        double value = 0.0;
        for (int g = cf.chunk_group_host[c]; g < cf.chunk_group_host[c + 1]; ++g)
          value += (g < groups.n_central_groups ? 210.0 : 330.0) +
                   (groups.begin[g + 1] - groups.begin[g]) * (10.0 + cf.rows);
        cost[c + 1] = cost[c] + value;
      }
      ca.split_all[0] = 0;
      for (int k = 1; k < n_splits; ++k) {
        const double target = cost[cf.n_chunks] * k / n_splits;
        int c = ca.split_all[k - 1] + 1;
        while (c < cf.n_chunks - (n_splits - k) && cost[c] < target) ++c;
        ca.split_all[k] = c;
      }
      ca.split_all[n_splits] = cf.n_chunks;
      ca.n_splits = n_splits;
      const size_t count = (size_t)(separate ? 2 : 1) * cf.rows * 64;
      status = partial->reserve((size_t)n_tiles * n_splits * count * sizeof(double), stream);
      if (status != TC_OK) return status;
      const size_t had = counters->bytes;
      status = counters->reserve((size_t)n_tiles * sizeof(int), stream);
      if (status != TC_OK) return status;
      if (counters->bytes != had) TC_HIP(hipMemsetAsync(counters->ptr, 0, counters->bytes, stream));
      ca.partial = (double*)partial->ptr;
      ca.counters = (int*)counters->ptr;
    }
  }
  const dim3 grid((unsigned)(n_tiles * n_splits)), block(64 * tc::kCrossWaves);
#ifdef TC_DEVELOPER_KNOBS
  if (ca.chi2 == nullptr && env_int_early("TC_FUSED_STAMPS", 0) != 0) {
    // (16 slots of stamps per workgroup of the last launch: tools/r06_stamps_cross.py)
    t0->trace_blocks = ((size_t)grid.x * 16 + 5) / 6;
    status = t0->trace.reserve(t0->trace_blocks * 6 * sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    TC_HIP(hipMemsetAsync(t0->trace.ptr, 0, t0->trace_blocks * 6 * sizeof(unsigned long long),
                          stream));
    ca.chi2_data = (const double*)t0->trace.ptr;
    ca.priority |= 1 << 10;
  }
#endif
  hipEvent_t k0 = nullptr, k1 = nullptr;
  status = next_kernel_events(t0, &k0, &k1);
  if (status != TC_OK) return status;
  const bool assembias = (flags & TC_FLAG_ASSEMBIAS) != 0;
  const bool modulate = (flags & TC_FLAG_MODULATE_WITH_CENOCC) != 0;
  // (ca.defer: undecorated, up to 64 rows -- the instances with the deferred pairs)
  status = launch_cross_instance(assembias, modulate, ca.defer != 0, t0->device, cf.rows, grid,
                                 block, lds, stream, k0, k1, ca);
  if (status != TC_OK) return status;
  t0->last_workgroups = (int)grid.x;
  t0->last_waves = tc::kCrossWaves;
  t0->last_splits = 0;
  t0->last_lds = lds;
  return TC_OK;
}

int check_predict_args(const tc_table* t, const void* theta, int n_theta,
                       int64_t n_draws, int n_gauss, unsigned flags) {
  TC_CHECK(t != nullptr, "table handle is NULL");
  TC_CHECK(n_draws >= 0, "n_draws must be non-negative");
  TC_CHECK(n_draws == 0 || theta != nullptr, "theta is NULL");
  TC_CHECK(n_gauss >= 1 && n_gauss <= 4096, "n_gauss_prim must be in [1, 4096]");
  TC_CHECK(!(flags & TC_FLAG_LEAUTHAUD11) || !(flags & TC_FLAG_ASSEMBIAS),
           "assembly bias is implemented for the Zheng07 family only");
  const int need = (flags & TC_FLAG_LEAUTHAUD11) ? tc::kLeauthaudTheta
                                                 : (flags & TC_FLAG_ASSEMBIAS) ? 7 : 5;
  TC_CHECK(n_theta == need, "theta must have %d columns, got %d", need, n_theta);
  return TC_OK;
}

// Un-batched predict(): one draw through single_draw_kernel, or TC_ERR_UNSUPPORTED when
// the call does not qualify (the caller then takes the batched path).
bool single_draw_eligible(const tc_table* t, int64_t n_draws, int n_gauss, unsigned flags) {
  return n_draws == 1 && t->compute_dtype == TC_DTYPE_F64 && t->n_rtiles == 1 &&
         !(flags & (TC_FLAG_SEPARATE_GAL_TYPE | TC_FLAG_LEAUTHAUD11)) && t->n_bins <= tc::kSingleMaxBins &&
         (int64_t)t->n_bins * n_gauss <= tc::kSingleMaxNodes && t->tuning.single_draw;
}

// Workspace of the un-batched path in page-locked host memory (SingleWorkspace, internal.h):
// per draw [0] centrals, [1] satellites number density and (workgroups, rt) partial sums of
// the contraction, then one completion word per workgroup.  n_walkers draws (1 ..
// kSingleMaxWalkers) go through ONE launch; combine_single_draw() turns a draw's part into
// (ngal, xi) once wait_single_done() has seen every workgroup's word.
void fill_single_args(const tc_table* t, const Quadrature* q, const double* theta, int n_theta,
                      int n_gauss, unsigned flags, int blocks, const SingleWorkspace& ws,
                      tc::SingleArgs* out);

int launch_single_draw(tc_table* t, const double* theta, int n_theta, int n_walkers, int n_gauss,
                       unsigned flags, SingleWorkspace* ws, hipStream_t stream) {
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  // workgroups per draw: a single pass over the table positions for a few draws; fewer (each
  // evaluates all occupation nodes of its draw, and the host polls one word per workgroup)
  // when many draws share the launch
  const int blocks = std::max(1, std::min(single_draw_blocks(t),
                                          std::max(2, t->tuning.many_blocks / n_walkers)));
  status = ws->prepare(n_walkers, blocks, t->rt, n_walkers > 1 ? n_walkers * n_theta : 0);
  if (status != TC_OK) return status;
  tc::SingleArgs sa{};
  fill_single_args(t, q, theta, n_theta, n_gauss, flags, blocks, *ws, &sa);
  if (n_walkers > 1) {
    // the draws travel through page-locked memory the kernel reads itself
    memcpy(ws->theta(), theta, (size_t)n_walkers * n_theta * sizeof(double));
    sa.theta_many = ws->theta();
    sa.n_walkers = n_walkers;
  }
  if (t->tuning.trace) {
    // developer timeline (tc_table_set_option "trace"): 8 stamps per workgroup
    status = t->trace.reserve((size_t)n_walkers * kSingleMaxBlocks * 8 *
                                  sizeof(unsigned long long), stream);
    if (status != TC_OK) return status;
    t->trace_blocks = (size_t)blocks;
    sa.stamps = (unsigned long long*)t->trace.ptr;
  }
  return launch_single_kernel(blocks * n_walkers, stream, sa);
}

// The argument block of one draw against one table (results and completion words in `ws`).
void fill_single_args(const tc_table* t, const Quadrature* q, const double* theta, int n_theta,
                      int n_gauss, unsigned flags, int blocks, const SingleWorkspace& ws,
                      tc::SingleArgs* out) {
  tc::SingleArgs& sa = *out;
  for (int i = 0; i < 7; ++i) sa.theta_value[i] = i < n_theta ? theta[i] : 0.0;
  sa.n_theta = n_theta;
  sa.n_bins = t->n_bins;
  sa.n_central = t->plan.n_central;
  sa.n_gauss = n_gauss;
  sa.flags = flags;
  sa.split = 0.5;
  sa.log_m = (const double*)q->log_m;
  sa.m = (const double*)q->m;
  sa.weight = (const double*)q->weight;
  sa.n_h = (const double*)t->d_n_h;
  sa.percentile = (const double*)t->d_percentile;
  sa.math_table = (const double*)t->d_math_table;
  sa.table = (const double*)t->d_table;
  sa.pos_off = (const int32_t*)t->d_pos_off;
  sa.n_positions = t->plan.n_positions;
  sa.rt = t->rt;
  sa.n_r = t->n_r;
  sa.mode = t->mode;
  sa.ngal = ws.ngal();
  sa.partial = ws.partial();
  sa.n_tables = 0;
  sa.blocks_per_table = blocks;
  sa.tables = nullptr;
  sa.table_class = nullptr;
  sa.class_log_m = sa.class_m = sa.class_weight = sa.class_n_h = sa.class_percentile = nullptr;
  sa.theta_many = nullptr;
  sa.n_walkers = 0;
  sa.done = ws.done();
  sa.epoch = ws.epoch;
  sa.stamps = nullptr;
}

// ---- resident un-batched path ---------------------------------------------------------------
//
// Of the 15 us of an un-batched call 10 are the launch path (an empty kernel whose completion
// word the host polls: tools/micro/graph_launch.hip).  With option "resident" the workgroups
// of ONE launch of resident_draw_kernel stay on the chip: every call writes its parameters and
// its number into a mailbox in page-locked memory, the workgroups (polling it over PCIe)
// evaluate the draw and answer through the same page-locked partial sums and completion words
// as single_draw_kernel.  The kernel leaves by itself when no call has arrived for
// resident.idle_us (or after 10 s), so that nothing ever waits for it longer than that; a call
// that finds it gone (a workgroup's `exited` word carries the launch's number) launches it
// again.  Every other entry point of the handle stops it first (it would otherwise hold a
// hardware queue that a lane's kernels may share).
namespace {
constexpr size_t kMailboxEntryWords = 16;     // seven entries {parameter, call number}, padded
constexpr size_t kMailboxWords = kMailboxEntryWords + 2 * kSingleMaxBlocks;
static_assert(tc::kResidentBusyOffset == kSingleMaxBlocks, "busy ticks behind the exited words");

// (the parameter before the call number of every entry: x86 keeps the order of the stores)
void publish(unsigned long long* mailbox, const double* theta, int n_theta,
             unsigned long long call) {
  for (int i = 0; i < 7; ++i) {
    const double value = i < n_theta ? theta[i] : 0.0;
    unsigned long long bits;
    memcpy(&bits, &value, 8);
    __atomic_store_n(mailbox + 2 * i, bits, __ATOMIC_RELAXED);
    __atomic_store_n(mailbox + 2 * i + 1, call, __ATOMIC_RELEASE);
  }
}

// 0: every workgroup has answered call `epoch`; 1: one has left before it did.
int resident_wait(tc_table* t, int* left) {
  tc_table::Resident& r = t->resident;
  const volatile unsigned long long* done = r.ws.done();
  const volatile unsigned long long* exited =
      (unsigned long long*)r.mailbox.ptr + kMailboxEntryWords;
  const unsigned long long epoch = r.ws.epoch;
  *left = 0;
  timespec start{};
  for (int b = 0; b < r.blocks; ++b) {
    unsigned spins = 0;
    while (done[b] != epoch) {
      if (exited[b] == r.launch_id) {
        *left = 1;
        return TC_OK;
      }
      __builtin_ia32_pause();
      if ((++spins & 0x3ff) != 0) continue;
      timespec now{};
      clock_gettime(CLOCK_MONOTONIC, &now);
      if (start.tv_sec == 0 && start.tv_nsec == 0) start = now;
      if ((now.tv_sec - start.tv_sec) * 1000000000LL + (now.tv_nsec - start.tv_nsec) >
          500000000LL) {
        (void)resident_stop(t);
        return fail(TC_ERR_HIP, "the resident kernel did not answer (workgroup %d)", b);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return TC_OK;
}
}  // namespace

// Single-draw resident kernels running per device, over all handles of the process (as far as
// the host knows: a kernel that has left by itself is noticed at its handle's next call).  A
// handle whose un-batched call would be a LAUNCH while another handle's resident kernel runs
// risks a place behind that kernel in a shared hardware queue -- it would wait out the other
// kernel's idle time (250 us), the other handle's next call would find its kernel gone, and the
// two would take turns like that until both streaks are long enough (the reference's two tables
// per step: tools/r06_two_tables.py saw 8 relaunches and a 4.5 ms call on the way in).  With a
// neighbour's kernel running, the automatic mode therefore engages at once.
std::atomic<int> g_resident_single[64];

bool other_resident_running(const tc_table* t) {
  if (t->device < 0 || t->device >= 64) return false;
  const int mine = t->resident.running && !t->resident.ensemble ? 1 : 0;
  return g_resident_single[t->device].load(std::memory_order_relaxed) > mine;
}

int resident_stop(tc_table* t) {
  tc_table::Resident& r = t->resident;
  if (!r.running) return TC_OK;
  if (r.ensemble) {
    // (the header; without the aperture workgroup 0 passes the word on)
    unsigned long long* header = r.ens_aperture.ptr != nullptr
                                     ? (unsigned long long*)r.ens_aperture.ptr
                                     : (unsigned long long*)r.ens_mailbox.ptr;
    __atomic_store_n(header, tc::kResidentStop, __ATOMIC_RELEASE);
    _mm_sfence();
  } else {
    unsigned long long* entries = r.single_aperture.ptr != nullptr
                                      ? (unsigned long long*)r.single_aperture.ptr
                                      : (unsigned long long*)r.mailbox.ptr;
    for (int i = 0; i < 7; ++i)
      __atomic_store_n(entries + 2 * i + 1, tc::kResidentStop, __ATOMIC_RELEASE);
    _mm_sfence();
  }
  if (!r.ensemble && t->device >= 0 && t->device < 64) g_resident_single[t->device].fetch_sub(1);
  r.running = false;
  r.ensemble = false;
  TC_HIP(hipStreamSynchronize(r.stream));
  return TC_OK;
}

// One node per thread and a single pass over the workgroup's positions (what the resident
// kernel keeps in registers between the calls).
bool resident_eligible(const tc_table* t, int n_gauss) {
  const int n_slices = tc::kSingleThreads / t->rt;
  return (int64_t)t->n_bins * n_gauss <= tc::kSingleThreads &&
         t->plan.n_positions <= (int64_t)single_draw_blocks(t) * 8 * n_slices;
}

// One call of the resident single-draw kernel in two halves, so that a joint call over several
// tables (tc_predict_zheng07_joint) can post every table's parameters before it waits for the
// first answer: resident_post prepares the handle's buffers, publishes the parameters under a new
// call number and launches the kernel unless it is running; resident_collect waits for the
// workgroups' answers -- relaunching (and publishing again) when one of them had left before
// it saw the call -- and combines them.  resident_predict = both.
namespace {
int resident_launch_and_publish(tc_table* t, Quadrature* q, const double* theta, int n_theta,
                                int n_gauss, unsigned flags) {
  tc_table::Resident& r = t->resident;
  const int blocks = single_draw_blocks(t);
  const int idle_us = r.enabled ? r.idle_us : std::min(r.idle_us, r.auto_idle_us);
  unsigned long long* mailbox = (unsigned long long*)r.mailbox.ptr;
  const bool direct = r.single_aperture.ptr != nullptr;
  unsigned long long* entries = direct ? (unsigned long long*)r.single_aperture.ptr : mailbox;
  // (the aperture is write-combining memory: an entry's two words share a line and leave in
  // program order, the fence sends them off at once)
  publish(entries, theta, n_theta, r.ws.epoch);
  if (direct) _mm_sfence();
  if (!r.running) {
    tc::SingleArgs sa{};
    fill_single_args(t, q, theta, n_theta, n_gauss, flags, blocks, r.ws, &sa);
    sa.mailbox = entries;
    sa.exited = mailbox + kMailboxEntryWords;
    sa.launch_id = ++r.launch_id;
    sa.idle_ticks = (unsigned long long)std::max(1, idle_us) * 100ull;     // 100 MHz
    r.running_idle_us = idle_us;
    r.auto_serving = !r.enabled;
    sa.life_ticks = 1000000000ull;                                          // 10 s
    sa.poll_waves = std::max(1, std::min(4, r.poll_waves));
    const int status = launch_resident_kernel(blocks, r.stream, sa);
    if (status != TC_OK) return status;
    r.running = true;
    if (t->device >= 0 && t->device < 64) g_resident_single[t->device].fetch_add(1);
    r.n_theta = n_theta;
    r.n_gauss = n_gauss;
    r.flags = flags;
    r.blocks = blocks;
  }
  return TC_OK;
}
}  // namespace

int resident_post(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags) {
  tc_table::Resident& r = t->resident;
  if (r.inject_failures > 0) {      // (option "resident_inject_failures": tests of the fallback)
    --r.inject_failures;
    return fail(TC_ERR_HIP, "the resident kernel keeps leaving before it answers (injected)");
  }
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  const int blocks = single_draw_blocks(t);
  const int idle_us = r.enabled ? r.idle_us : std::min(r.idle_us, r.auto_idle_us);
  if (r.running && (r.ensemble || r.n_theta != n_theta || r.n_gauss != n_gauss ||
                    r.flags != flags || r.blocks != blocks || r.running_idle_us != idle_us)) {
    status = resident_stop(t);        // (another kind of call: its own launch)
    if (status != TC_OK) return status;
  }
  if (r.stream == nullptr) TC_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
  if (r.mailbox.ptr == nullptr) {
    status = r.mailbox.reserve(kMailboxWords * 8);
    if (status != TC_OK) return status;
    memset(r.mailbox.ptr, 0, r.mailbox.bytes);
  }
  if (!r.single_aperture_decided) {
    // On a large-BAR system the seven entries live in device memory that the host stores into
    // through the PCIe aperture (tools/micro/bar_write.hip): the polling wave reads local
    // memory instead of crossing the link.  The exited words stay in page-locked memory.
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, t->device) != hipSuccess) {
      (void)hipGetLastError();
      large_bar = 0;
    }
    if (large_bar && t->tuning.resident_aperture) {
      status = r.single_aperture.reserve(256, r.stream);
      if (status != TC_OK) return status;
      TC_HIP(hipMemset(r.single_aperture.ptr, 0, r.single_aperture.bytes));
      TC_HIP(hipDeviceSynchronize());
    } else {
      r.single_aperture.release();
    }
    r.single_aperture_decided = true;
  }
  status = r.ws.prepare(1, blocks, t->rt, 0);       // (a new call number: r.ws.epoch)
  if (status != TC_OK) return status;
  return resident_launch_and_publish(t, q, theta, n_theta, n_gauss, flags);
}

int resident_collect(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags,
                     double* ngal, double* xi) {
  tc_table::Resident& r = t->resident;
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (attempt > 0) {
      status = resident_launch_and_publish(t, q, theta, n_theta, n_gauss, flags);
      if (status != TC_OK) return status;
    }
    int left = 0;
    status = resident_wait(t, &left);
    if (status != TC_OK) return status;
    if (!left) {
      combine_single_draw(t, r.ws, 0, ngal, xi);
      return TC_OK;
    }
    // a workgroup has left (idle or life time) before it saw this call: all of them out, then
    // a new launch serves it
    ++r.relaunches;
    status = resident_stop(t);
    if (status != TC_OK) return status;
  }
  return fail(TC_ERR_HIP, "the resident kernel keeps leaving before it answers");
}

int resident_predict(tc_table* t, const double* theta, int n_theta, int n_gauss, unsigned flags,
                     double* ngal, double* xi) {
  const int status = resident_post(t, theta, n_theta, n_gauss, flags);
  if (status != TC_OK) return status;
  return resident_collect(t, theta, n_theta, n_gauss, flags, ngal, xi);
}

// ---- resident ensemble path -------------------------------------------------------------------
//
// kernel_args.h (EnsembleArgs) has the protocol.  Page-locked: mailbox = the header {walkers,
// call number} on a line of its own, 8 doubles per walker, (grid) exited words and 8 stamps; out = (4, rt + 2, 64) doubles, then (grid)
// completion words.  Device: dens | flag_a | partial | flag_b | callword, zeroed once.
namespace {
struct EnsembleLayout {
  int grid = 0, n_slices = 0, per_quarter = 0, dens_stride = 0;
  int lds_area = 0, lds_dens = 0, lds_t = 0, lds_ij = 0, lds_bytes = 0;
  size_t dev_flag_a = 0, dev_partial = 0, dev_flag_b = 0, dev_callword = 0, dev_bytes = 0;
};

bool ensemble_layout(const tc_table* t, int n_gauss, EnsembleLayout* out) {
  const int n_cus = t->n_cus;       // (of the handle's device, not of whichever is current)
  EnsembleLayout l;
  l.n_slices = std::min(64, n_cus / 4);
  l.grid = 4 * l.n_slices;
  const int rt = t->rt;
  if (l.n_slices < rt + 2 || rt > 32) return false;
  if ((int64_t)t->n_bins * n_gauss > 1024 || t->n_bins > 1022) return false;
  l.per_quarter = (int)((t->plan.n_positions + l.grid - 1) / l.grid);
  if (l.per_quarter > 32) return false;        // (sixteen positions per wave and quarter)
  l.dens_stride = (t->n_bins + 2 + 7) / 8 * 8;
  auto align = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t at = align((size_t)tc::fm::kTableDoubles * 8);
  l.lds_area = (int)at;
  at = align(at + (size_t)2 * 1024 * 8);         // (phase A; phases B and C need as much)
  l.lds_dens = (int)at;
  at = align(at + (size_t)(t->n_bins + 2) * tc::kEnsembleDensPad * 8);
  l.lds_t = (int)at;
  at = align(at + (size_t)4 * l.per_quarter * 32 * 8);
  l.lds_ij = (int)at;
  at = align(at + (size_t)4 * l.per_quarter * 4);
  l.lds_bytes = (int)at;
  if (at > 150 * 1024) return false;
  size_t dev = align((size_t)tc::kEnsembleMaxWalkers * l.dens_stride * 8);
  l.dev_flag_a = dev;
  dev = align(dev + (size_t)tc::kEnsembleMaxWalkers * 8);
  l.dev_partial = dev;
  dev = align(dev + (size_t)l.grid * rt * 64 * 8);
  l.dev_flag_b = dev;
  dev = align(dev + (size_t)l.grid * 8);
  l.dev_callword = dev;
  l.dev_bytes = dev + 256;
  *out = l;
  return true;
}
}  // namespace

bool ensemble_eligible(const tc_table* t, int64_t n_walkers, int n_gauss, unsigned flags) {
  // (walker b is served by workgroup b: on a device with fewer than 256 CUs the grid -- 4 x
  // min(64, CUs / 4) workgroups -- bounds the ensemble, or walkers beyond it would never be
  // computed and every such call would wait out its time limits before falling back)
  EnsembleLayout l;
  return !t->resident.ens_disabled && n_walkers >= std::max(2, t->resident.min_walkers) &&
         n_walkers <= tc::kEnsembleMaxWalkers && single_draw_eligible(t, 1, n_gauss, flags) &&
         ensemble_layout(t, n_gauss, &l) && n_walkers <= l.grid;
}

int ensemble_predict(tc_table* t, const double* theta, int n_theta, int n_walkers, int n_gauss,
                     unsigned flags, double* ngal, double* xi) {
  tc_table::Resident& r = t->resident;
  Quadrature* q = nullptr;
  int status = get_quadrature(t, n_gauss, &q);
  if (status != TC_OK) return status;
  EnsembleLayout l;
  TC_CHECK(ensemble_layout(t, n_gauss, &l), "internal: ensemble routing");
  if (r.running && (!r.ensemble || r.n_theta != n_theta || r.n_gauss != n_gauss ||
                    r.flags != flags)) {
    status = resident_stop(t);        // (another kind of call: its own launch)
    if (status != TC_OK) return status;
  }
  if (r.stream == nullptr) TC_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
  const int rt = t->rt;
  const size_t exited_offset = 8 + (size_t)tc::kEnsembleMaxWalkers * 8;
  const size_t mailbox_words = exited_offset + l.grid + 8;
  const size_t out_doubles = (size_t)4 * (rt + 2) * 64;      // (then 4 x (rt + 2) words)
  if (r.ens_mailbox.ptr == nullptr || r.ens_grid != l.grid) {
    status = r.ens_mailbox.reserve(mailbox_words * 8);
    if (status != TC_OK) return status;
    memset(r.ens_mailbox.ptr, 0, r.ens_mailbox.bytes);
    status = r.ens_out.reserve((out_doubles + l.grid) * 8);
    if (status != TC_OK) return status;
    memset(r.ens_out.ptr, 0, r.ens_out.bytes);
    status = r.ens_device.reserve(l.dev_bytes, r.stream);
    if (status != TC_OK) return status;
    TC_HIP(hipMemset(r.ens_device.ptr, 0, r.ens_device.bytes));
    // With a large BAR the runtime maps device memory into the host's address space: the
    // mailbox goes there (option "resident_aperture" = 0: page-locked host memory instead).
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, t->device) != hipSuccess) {
      (void)hipGetLastError();
      large_bar = 0;
    }
    if (large_bar && t->tuning.resident_aperture) {
      status = r.ens_aperture.reserve(exited_offset * 8, r.stream);
      if (status != TC_OK) return status;
      TC_HIP(hipMemset(r.ens_aperture.ptr, 0, r.ens_aperture.bytes));
      TC_HIP(hipDeviceSynchronize());
    } else {
      r.ens_aperture.release();
    }
    r.ens_grid = l.grid;
  }
  const bool direct = r.ens_aperture.ptr != nullptr;
  unsigned long long* lines =
      direct ? (unsigned long long*)r.ens_aperture.ptr : (unsigned long long*)r.ens_mailbox.ptr;
  volatile unsigned long long* exited = (unsigned long long*)r.ens_mailbox.ptr + exited_offset;
  double* out = (double*)r.ens_out.ptr;
  volatile unsigned long long* done = (unsigned long long*)(out + out_doubles);
  const unsigned long long epoch = ++r.ens_epoch;
  const int n_wg = (n_walkers + 63) / 64;
  auto now_ns = [] {
    timespec ts{};
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec;
  };
  const unsigned long long host_begin = now_ns();
  for (int attempt = 0; attempt < 2; ++attempt) {
    // the parameters (8 doubles per walker), then the header (call number << 10 | walkers): a
    // workgroup reads its parameters only after the header has been seen
    {
      double* slots = (double*)(lines + 8);
      for (int w = 0; w < n_walkers; ++w)
        for (int i = 0; i < 7; ++i)
          slots[(size_t)w * 8 + i] = i < n_theta ? theta[(size_t)w * n_theta + i] : 0.0;
      // (the aperture is write-combining memory: the parameters leave the buffers before
      // the header, and the header at once)
      if (direct) _mm_sfence();
      __atomic_store_n(lines + 0, (epoch << 10) | (unsigned long long)n_walkers, __ATOMIC_RELEASE);
      if (direct) _mm_sfence();
    }
    r.ens_host_ns[0] = now_ns() - host_begin;       // (published)
    if (!r.running) {
      tc::EnsembleArgs ea{};
      ea.n_theta = n_theta;
      ea.n_bins = t->n_bins;
      ea.n_central = t->plan.n_central;
      ea.n_gauss = n_gauss;
      ea.flags = flags;
      ea.split = 0.5;
      ea.log_m = (const double*)q->log_m;
      ea.m = (const double*)q->m;
      ea.weight = (const double*)q->weight;
      ea.n_h = (const double*)t->d_n_h;
      ea.percentile = (const double*)t->d_percentile;
      ea.math_table = (const double*)t->d_math_table;
      ea.table = (const double*)t->d_table;
      ea.pos_off = (const int32_t*)t->d_pos_off;
      ea.n_positions = t->plan.n_positions;
      ea.rt = rt;
      ea.mode = t->mode;
      ea.n_slices = l.n_slices;
      ea.per_quarter = l.per_quarter;
      ea.dens_stride = l.dens_stride;
      char* device = (char*)r.ens_device.ptr;
      ea.dens = (double*)device;
      ea.flag_a = (unsigned long long*)(device + l.dev_flag_a);
      ea.partial = (double*)(device + l.dev_partial);
      ea.flag_b = (unsigned long long*)(device + l.dev_flag_b);
      ea.callword = (unsigned long long*)(device + l.dev_callword);
      ea.mailbox = lines;
      ea.direct = direct ? 1 : 0;
      ea.skip = env_int_early("TC_ENS_SKIP", 0);
      ea.out = out;
      ea.done = (unsigned long long*)done;
      ea.exited = (unsigned long long*)exited;
      ea.epoch = epoch;
      ea.launch_id = ++r.launch_id;
      ea.idle_ticks = (unsigned long long)std::max(1, r.idle_us) * 100ull;   // 100 MHz
      ea.life_ticks = 1000000000ull;                                          // 10 s
      ea.call_ticks = (unsigned long long)std::max(1, r.wait_us) * 100ull;
      ea.lds_area = l.lds_area;
      ea.lds_dens = l.lds_dens;
      ea.lds_t = l.lds_t;
      ea.lds_ij = l.lds_ij;
      // (the word a previous launch's workgroup 0 left behind)
      TC_HIP(hipMemsetAsync(ea.callword, 0, 8, r.stream));
      status = launch_ensemble_kernel(t->device, l.grid, l.lds_bytes, r.stream, ea);
      if (status != TC_OK) return status;
      r.running = true;
      r.ensemble = true;
      r.n_theta = n_theta;
      r.n_gauss = n_gauss;
      r.flags = flags;
    }
    // The completion words of the (group, row) workgroups, the two rows of totals first; every
    // row goes to the caller's arrays as soon as it is there (tabcorr.py:646-650), the lines of
    // the next rows on their way from memory meanwhile (the device's writes land in DRAM: a
    // first read of a line costs ~100 ns).
    bool left = false;
    {
      timespec start{};
      unsigned spins = 0;
      double norm[64];
      for (int grp = 0; grp < n_wg && !left; ++grp) {
        const double* group = out + (size_t)grp * (rt + 2) * 64;
        const int count = std::min(64, n_walkers - 64 * grp);
        for (int step = 0; step < rt + 2 && !left; ++step) {
          auto row_of = [rt](int s) { return s < 2 ? rt + s : s - 2; };
          const int row = row_of(step);
          for (int s = step == 0 ? 0 : step + 3; s <= step + 3 && s < rt + 2; ++s)
            for (int line = 0; line < count * 8; line += 64)
              __builtin_prefetch((const char*)(group + (size_t)row_of(s) * 64) + line, 0, 0);
          volatile unsigned long long* word = done + (size_t)grp * (rt + 2) + row;
          while (*word != epoch) {
            __builtin_ia32_pause();
            if ((++spins & 0x3ff) != 0) continue;
            // (a workgroup that has left says which call it would have served next)
            for (int other = 0; other < l.grid; ++other) {
              const unsigned long long word = exited[other];
              if ((word >> 40) == r.launch_id && (word & ((1ull << 40) - 1)) <= epoch) left = true;
            }
            if (left) break;
            timespec now{};
            clock_gettime(CLOCK_MONOTONIC, &now);
            if (start.tv_sec == 0 && start.tv_nsec == 0) start = now;
            if ((now.tv_sec - start.tv_sec) * 1000000000LL + (now.tv_nsec - start.tv_nsec) >
                500000000LL) {
              (void)resident_stop(t);
              return fail(TC_ERR_HIP,
                          "the resident ensemble kernel did not answer (group %d, row %d)", grp, row);
            }
          }
          if (left) break;
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          const unsigned long long seen_ns = now_ns();
          if (grp == 0 && step == 0) r.ens_host_ns[2] = 0;
          const double* values = group + (size_t)row * 64;
          if (row == rt) {
            for (int w = 0; w < count; ++w) norm[w] = values[w];
          } else if (row == rt + 1) {
            for (int w = 0; w < count; ++w) {
              const double total = norm[w] + values[w];
              ngal[64 * grp + w] = total;
              norm[w] = t->mode == TC_MODE_AUTO ? total * total : total;
            }
          } else if (row < t->n_r) {
            for (int w = 0; w < count; ++w)
              xi[(size_t)(64 * grp + w) * t->n_r + row] = values[w] / norm[w];
          }
          r.ens_host_ns[2] += now_ns() - seen_ns;      // (time spent on the rows)
        }
      }
    }
    r.ens_host_ns[1] = now_ns() - host_begin;       // (every row seen and combined)
    if (!left) {
      r.ens_failures = 0;
      return TC_OK;
    }
    // a workgroup has left (idle, life time or a wait that ran out): all of them out, then a
    // new launch serves the call
    status = resident_stop(t);
    if (status != TC_OK) return status;
  }
  // (its workgroups do not all find a place -- the chip is shared with other work: the launched
  // path serves this call, and after three such calls every call until the option is set again)
  if (++r.ens_failures >= 3) r.ens_disabled = true;
  return TC_ERR_UNSUPPORTED;
}

int single_draw_blocks(const tc_table* t) {
  const int n_slices = tc::kSingleThreads / t->rt;
  return (int)std::max<int64_t>(
      1, std::min<int64_t>(kSingleMaxBlocks,
                           (t->plan.n_positions + 8 * n_slices - 1) / (8 * n_slices)));
}

// Host half of the un-batched path: the partial sums of draw `walker` in workgroup order,
// divided by the total pair weight (tabcorr.py:646-650).
void combine_single_draw(const tc_table* t, const SingleWorkspace& ws, int walker, double* ngal,
                         double* xi) {
  const double* densities = ws.ngal() + 2 * (size_t)walker;
  const double total = densities[0] + densities[1];
  const double norm = t->mode == TC_MODE_AUTO ? total * total : total;
  const double* partial = ws.partial() + (size_t)walker * ws.blocks * t->rt;
  for (int r = 0; r < t->n_r; ++r) {
    double sum = 0.0;
    for (int b = 0; b < ws.blocks; ++b) sum += partial[(size_t)b * t->rt + r];
    xi[r] = sum / norm;
  }
  ngal[0] = total;
}

// Page-locked workspace of the un-batched path: [ngal (jobs, 2) | partial (jobs, blocks, rt) |
// theta (extra doubles) | done (jobs x blocks words)].  Every call gets a new epoch; a
// workgroup's completion word carries the epoch of the call it belongs to.
int SingleWorkspace::prepare(int n_jobs, int n_blocks, int rt, int extra_doubles) {
  const size_t doubles = (size_t)n_jobs * (2 + (size_t)n_blocks * rt) + extra_doubles;
  const size_t bytes = (doubles + (size_t)n_jobs * n_blocks) * sizeof(double);
  if (bytes > buffer.bytes) {
    int status = buffer.reserve(bytes);
    if (status != TC_OK) return status;
    memset(buffer.ptr, 0, buffer.bytes);     // (no stale completion words)
    epoch = 0;
  }
  jobs = n_jobs;
  blocks = n_blocks;
  partial_offset = 2 * (size_t)n_jobs;
  theta_offset = partial_offset + (size_t)n_jobs * n_blocks * rt;
  done_offset = theta_offset + extra_doubles;
  ++epoch;
  return TC_OK;
}

// Wait until every workgroup of the launch has stored its completion word (written behind
// its results with a system-scope release): polling host memory costs the PCIe write latency
// where hipStreamSynchronize adds the runtime's signal path (~5 us per call).  Falls back to
// the stream synchronisation -- which also reports a failed launch -- after 2 ms, and joins the
// stream every 256 calls so that the runtime retires its finished commands.
int wait_single_done(SingleWorkspace* ws, hipStream_t stream, bool poll) {
  const volatile unsigned long long* done = ws->done();
  const int n = ws->jobs * ws->blocks;
  const unsigned long long epoch = ws->epoch;
  bool joined = false;
  if (!poll) {
    TC_HIP(hipStreamSynchronize(stream));
    joined = true;
  }
  for (int b = 0; b < n && !joined; ++b) {
    unsigned spins = 0;
    timespec start{};
    while (done[b] != epoch) {
      __builtin_ia32_pause();
      if ((++spins & 0x3ff) != 0) continue;
      timespec now{};
      clock_gettime(CLOCK_MONOTONIC, &now);
      if (start.tv_sec == 0 && start.tv_nsec == 0) start = now;
      if ((now.tv_sec - start.tv_sec) * 1000000000LL + (now.tv_nsec - start.tv_nsec) > 2000000LL) {
        TC_HIP(hipStreamSynchronize(stream));
        joined = true;
        break;
      }
    }
  }
  if (joined) {
    for (int b = 0; b < n; ++b)
      if (done[b] != epoch)
        return fail(TC_ERR_HIP, "the un-batched kernel finished without reporting workgroup %d",
                    b);
  } else if ((ws->epoch & 0xff) == 0) {
    TC_HIP(hipStreamSynchronize(stream));
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return TC_OK;
}

int launch_occ_from_array(tc_table* t, const double* occupation_device, int64_t n_draws,
                          int64_t ldb, double* nbuf, double* ngal2, hipStream_t stream) {
  // (the float32 quadratic-form kernel reads a float copy of the densities of lane 0)
  float* nbuf32 = nullptr;
  if (t->quad && t->compute_dtype == TC_DTYPE_F32 && nbuf == (double*)t->lanes[0].nbuf.ptr) {
    int status = t->lanes[0].nbuf32.reserve((size_t)t->n_bins * ldb * sizeof(float), stream);
    if (status != TC_OK) return status;
    nbuf32 = (float*)t->lanes[0].nbuf32.ptr;
  }
  return launch_occ_from_array_kernel(occupation_device, n_draws, ldb, t->n_bins,
                                      t->plan.n_central, (const double*)t->d_n_h,
                                      (const int32_t*)t->d_perm, nbuf, ngal2, nbuf32, stream);
}

}  // namespace host
}  // namespace tc
