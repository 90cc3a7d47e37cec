// Interpolator handle of the C ABI (include/tabcorr_amd.h): tensor-product cubic-spline
// interpolation over a grid of tables, evaluated as one contraction over all tables
// (tabcorr/interpolator.py:136-219).
#include "internal.h"

using namespace tc::host;

struct tc_interp {
  int device = 0;
  int n_tables = 0;
  int n_dim = 0;
  std::vector<tc_table*> tables;
  std::vector<std::vector<double>> xp;      // per dimension
  std::vector<int32_t> table_node;          // (K, D)
  std::vector<int32_t> table_class;         // (K)
  std::vector<int> class_table;             // representative table of each class
  void* d_xp = nullptr;
  void* d_a = nullptr;
  void* d_table_node = nullptr;
  void* d_table_class = nullptr;
  void* d_tables = nullptr;                 // (K) device pointers
  void* d_quad_by_type = nullptr;           // (K) matrices of the quadratic-form kernel
  void* d_quad_total = nullptr;             // (K) ... unpadded triangle, if the tables have it
  std::vector<int> axis_offset, a_offset;
  std::vector<double> a_host;               // spline matrices of all dimensions
  // un-batched calls: per n_gauss the per-class pointer arrays of single_draw_kernel
  struct SinglePointers {
    void* log_m = nullptr;
    void* m = nullptr;
    void* weight = nullptr;
    void* n_h = nullptr;
    void* percentile = nullptr;
  };
  std::map<int, SinglePointers> single_pointers;
  // Lanes (stream + workspaces), as for a table handle: consecutive device-pointer and
  // asynchronous calls alternate between them, so that the occupation / spline-weight /
  // finalisation kernels and the ramps of one call's contraction hide behind the
  // neighbour's (BASELINE configs[3], 12 500 draws: 992 -> 929 us per call).  Host-buffer
  // calls use lane 0.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t finished = nullptr;            // recorded by the gather, not per call
    std::vector<DeviceBuffer> nbuf, ngal2;    // per class
    std::vector<DeviceBuffer> nbuf32;         // per class: float copies (float32 quadratic form)
    std::vector<void*> nbuf_ptrs, ngal_ptrs, nbuf32_ptrs;   // last uploaded pointer values
    void* d_nbufs = nullptr;                  // (V) device pointers
    void* d_ngal_parts = nullptr;             // (V) device pointers
    void* d_nbufs32 = nullptr;                // (V) ... float copies of the densities
    DeviceBuffer coef, partial, chi2_xi;
    DeviceBuffer cross_counters;              // mode cross, several workgroups per tile
    DeviceBuffer stage_in, stage_out;         // asynchronous host calls
  };
  static constexpr int kLanes = 4;
  Lane lanes[kLanes];
  // Lanes in use: two for the three-kernel forms (their contraction fills the chip by itself),
  // four in mode cross, where a launch of predict_cross_fused_kernel is one workgroup per 64
  // draws and only several launches in flight fill the CUs (the reference's AbacusSummit
  // interpolator, 10^4 draws per call: 143 us per call with two lanes)
  int n_lanes = 2;
  int force_lane = -1;                        // host-buffer entry points pin lane 0
  int cur = 0;                                // lane of the current / last call
  uint64_t device_calls = 0;
  hipStream_t stream = nullptr;               // = lanes[0].stream
  DeviceBuffer theta, x, out_ngal, out_xi;  // host-buffer calls
  DeviceBuffer chi2_data;                   // fused likelihood: data + precision
  std::vector<double> chi2_host;            // host copy of what chi2_data holds
  PinnedBuffer h_in, h_out;
  SingleWorkspace single_ws;                // un-batched calls
  std::map<std::pair<int, int>, std::unique_ptr<DeviceChunking>> chunkings;
  CrossFused cross_fused, cross_fused_wide; // mode cross: one launch per batch (launch.hip)
  // asynchronous host calls (tc_interp_*_async): tickets as for a table handle
  tc_table::Ticket tickets[tc_table::kMaxTickets];
  int64_t next_ticket = 0;
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
  tc_interp::Lane& L = it->lanes[it->cur];
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? t0->plan.n_components : 1;
  const int n_classes = (int)it->class_table.size();
  const int64_t ldb = (n_draws + 63) / 64 * 64;
  int status = TC_OK;

  // mode cross: everything in one launch where it pays (predict_cross_fused_kernel)
  if (t0->mode == TC_MODE_CROSS && !t0->cross_host.empty() && t0->tuning.fused != 0) {
    const CrossFused& cf =
        *choose_cross_fused(it->tables.data(), it->n_tables, &it->cross_fused,
                            &it->cross_fused_wide, n_draws, flags, &status);
    if (status != TC_OK) return status;
    const bool alone = it->force_lane >= 0 || !t0->tuning.pipeline;
    if (cross_fused_eligible(t0, cf, n_draws, n_gauss, flags, alone)) {
      tc::CrossFusedArgs ca{};
      ca.n_dim = it->n_dim;
      for (int d = 0; d < it->n_dim; ++d) {
        ca.n_axis[d] = (int)it->xp[d].size();
        ca.axis_offset[d] = it->axis_offset[d];
        ca.a_offset[d] = it->a_offset[d];
      }
      ca.xp = (const double*)it->d_xp;
      ca.a = (const double*)it->d_a;
      ca.table_node = (const int32_t*)it->d_table_node;
      ca.x = x_device;
      return run_cross_fused(t0, cf, &ca, theta_device, n_theta, n_draws, flags,
                             ngal_device, xi_device, L.stream, &L.partial, &L.cross_counters);
    }
  }

  // occupations once per class of identical halo tables (interpolator.py:181-184)
  int ngal_parts = 1;
  const bool quad_f32 = t0->quad && t0->compute_dtype == TC_DTYPE_F32;
  for (int v = 0; v < n_classes; ++v) {
    status = run_occupation(it->tables[it->class_table[v]], theta_device, n_theta,
                            n_draws, ldb, n_gauss, flags, nullptr, &L.nbuf[v],
                            &L.ngal2[v], L.stream, &ngal_parts,
                            quad_f32 ? &L.nbuf32[v] : nullptr);
    if (status != TC_OK) return status;
  }
  bool moved = false;
  for (int v = 0; v < n_classes; ++v) {
    moved = moved || L.nbuf_ptrs[v] != L.nbuf[v].ptr ||
            L.ngal_ptrs[v] != L.ngal2[v].ptr || L.nbuf32_ptrs[v] != L.nbuf32[v].ptr;
    L.nbuf_ptrs[v] = L.nbuf[v].ptr;
    L.ngal_ptrs[v] = L.ngal2[v].ptr;
    L.nbuf32_ptrs[v] = L.nbuf32[v].ptr;
  }
  if (moved) {
    TC_HIP(hipMemcpyAsync(L.d_nbufs32, L.nbuf32_ptrs.data(), n_classes * sizeof(void*),
                          hipMemcpyHostToDevice, L.stream));
    TC_HIP(hipMemcpyAsync(L.d_nbufs, L.nbuf_ptrs.data(), n_classes * sizeof(void*),
                          hipMemcpyHostToDevice, L.stream));
    TC_HIP(hipMemcpyAsync(L.d_ngal_parts, L.ngal_ptrs.data(),
                          n_classes * sizeof(void*), hipMemcpyHostToDevice, L.stream));
    TC_HIP(hipStreamSynchronize(L.stream));   // the host vectors may change later
  }

  status = L.coef.reserve((size_t)it->n_tables * ldb * sizeof(double), L.stream);
  if (status != TC_OK) return status;
  tc::InterpArgs ia{};
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
  ia.ngal_parts = (const double* const*)L.d_ngal_parts;
  ia.n_ngal_parts = ngal_parts;
  ia.ldb = ldb;
  ia.n_draws = n_draws;
  ia.coef = (double*)L.coef.ptr;
  ia.ngal = ngal_device;
  {
    Range range("spline weights");
    status = launch_interp_coef(ia, L.stream);
  }
  if (status != TC_OK) return status;
  Range range("contraction + finalisation (all tables)");

  if (t0->quad) {
    // quadratic-form kernel: the unit space (draw tile, r tile, component, TABLE, unit) in
    // equal shares per wave; a table's spline weight scales the outer factor n_i
    const bool by_type = separate || t0->quad_total.d_table == nullptr;
    QuadTable* q = by_type ? &t0->quad_by_type : &t0->quad_total;
    const tc::QuadTiling& tiling = t0->quad_tiling;
    DeviceQuadSchedule* schedule = nullptr;
    const int tile_draws = quad_f32 ? tc::kQuadTileF32 : tc::kQuadTile;
    status = get_quad_schedule(t0, q, ldb / tile_draws, it->n_tables, separate, &schedule);
    if (status != TC_OK) return status;
    const int rt = 4 * tiling.n_u;
    status = L.partial.reserve((size_t)schedule->n_slabs * rt * tile_draws *
                                     (quad_f32 ? sizeof(float) : sizeof(double)),
                                 L.stream);
    if (status != TC_OK) return status;
    tc::QuadArgs qa{};
    qa.nbuf = nullptr;
    qa.nbufs = (const double* const*)L.d_nbufs;
    qa.nbufs32 = (const float* const*)L.d_nbufs32;
    qa.ldb = ldb;
    qa.n_bins = t0->n_bins;
    qa.table = nullptr;
    qa.tables = (const double* const*)(by_type ? it->d_quad_by_type : it->d_quad_total);
    qa.table_class = (const int32_t*)it->d_table_class;
    qa.coef = (const double*)L.coef.ptr;
    qa.rtile_bytes = (uint32_t)q->rtile_bytes;
    qa.runs = (const tc::QuadRun*)schedule->runs;
    qa.comps = (const tc::QuadCompArgs*)q->d_comps;
    qa.wave_runs = (const int32_t*)schedule->wave_runs;
    qa.wave_head = (const int32_t*)schedule->wave_head;
    qa.n_waves = schedule->n_waves;
    qa.partial = L.partial.ptr;
    qa.priority = t0->tuning.prio_contract;
    qa.merge_range = (const int32_t*)schedule->merge_range;
    qa.merges = (const int32_t*)schedule->merges;
    qa.stamps = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;     // timed through the first table's timer
    status = next_kernel_events(t0, &k0, &k1);
    if (status != TC_OK) return status;
    status = quad_f32 ? launch_contract_quad_f32_interp(tiling.n_u, qa, schedule->lds_bytes,
                                                        L.stream, k0, k1)
                      : launch_contract_quad(tiling.n_u, true, qa, schedule->lds_bytes,
                                             L.stream, k0, k1);
    if (status != TC_OK) return status;
    t0->last_workgroups = (schedule->n_waves + tc::kQuadWavesPerBlock - 1) / tc::kQuadWavesPerBlock;
    t0->last_waves = tc::kQuadWavesPerBlock;
    t0->last_splits = schedule->n_slabs;
    t0->last_lds = schedule->lds_bytes;
    tc::FinalizeQuadArgs fq{};
    fq.partial = L.partial.ptr;
    fq.group_begin = (const int32_t*)schedule->group_begin;
    fq.ngal_part = nullptr;      // already normalised; ngal written by the coef kernel
    fq.n_ngal_parts = 0;
    fq.n_rtiles = tiling.n_rtiles;
    fq.r_per_tile = tiling.r_per_tile;
    fq.rt = rt;
    fq.groups_per_rtile = separate ? (int)q->layout.comps.size() : 1;
    fq.priority = t0->tuning.prio_finalize;
    fq.n_comp = n_comp;
    fq.n_r = t0->n_r;
    fq.mode = t0->mode;
    fq.ldb = ldb;
    fq.n_draws = n_draws;
    fq.ngal = ngal_device;
    fq.xi = xi_device;
    t0->chi2_fused = false;
    if (t0->fuse_chi2_out != nullptr && !separate && t0->n_r <= tc::kFinalizeRows) {
      // (launch.hip, run_contraction_quad: the likelihood from the finalisation's LDS tile)
      fq.chi2_data = t0->fuse_chi2_data;
      fq.chi2 = t0->fuse_chi2_out;
      fq.xi = nullptr;
      t0->chi2_fused = true;
    }
    return launch_finalize_quad(fq, t0->tuning, L.stream, quad_f32);
  }

  // decomposition: as for one table (choose_chunking), with the tables looped inside
  // the block; the tables are split over blocks only when one pass would leave the chip
  // underfilled
  const int64_t n_tiles = ldb / 64;
  DeviceChunking* c = nullptr;
  int lds = 0;
  {
    // the table handle caches the chunkings; all tables share table 0's plan
    status = choose_chunking(t0, n_draws, std::min(it->n_tables, 1000), &c, &lds);
    if (status != TC_OK) return status;
  }
  int k_splits = 1;
  {
    const int64_t blocks = n_tiles * t0->n_rtiles * (int64_t)c->host.groups.size();
    const int64_t want = t0->n_cus * (int64_t)blocks_per_cu(lds, c->host.waves_per_group,
                                                        wave_slots(t0, true));
    // about two scheduling rounds of blocks (measured optimum for 16 tiles x 20 groups x
    // 25 tables: 3 splits); every split adds a slab the finalisation has to sum
    if (blocks < want)
      k_splits = (int)std::min<int64_t>(
          it->n_tables, std::max<int64_t>(1, 2 * want / std::max<int64_t>(1, blocks)));
    if (t0->tuning.k_splits > 0) k_splits = t0->tuning.k_splits;
    k_splits = std::min(k_splits, it->n_tables);
  }
  if (lds > kMaxLdsBytes)
    return fail(TC_ERR_UNSUPPORTED, "table with %d bins needs %d bytes of LDS",
                t0->n_bins, lds);
  const int n_groups = (int)c->host.groups.size();
  const int r_stride = t0->rt * t0->n_rtiles;
  status = L.partial.reserve(
      (size_t)n_groups * k_splits * r_stride * ldb * sizeof(double), L.stream);
  if (status != TC_OK) return status;

  tc::ContractArgs ca{};
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
  ca.pos_off = (const int32_t*)t0->d_pos_off;
  ca.partial = (double*)L.partial.ptr;
  ca.n_tables = it->n_tables;
  ca.k_splits = k_splits;
  ca.tables = (const double* const*)it->d_tables;
  ca.nbufs = (const double* const*)L.d_nbufs;
  ca.table_class = (const int32_t*)it->d_table_class;
  ca.coef = (const double*)L.coef.ptr;
  ca.n_tiles = (int)n_tiles;
  ca.n_slabs = n_groups * k_splits;
  ca.xcd_map = t0->n_xcds == 8 && n_tiles >= 8;
  const int64_t padded_tiles = ca.xcd_map ? (n_tiles + 7) / 8 * 8 : n_tiles;
  dim3 grid((unsigned)(padded_tiles * n_groups * k_splits), 1,
            (unsigned)t0->n_rtiles);
  dim3 block(64 * c->host.waves_per_group);
  hipEvent_t k0 = nullptr, k1 = nullptr;       // timed through the first table's timer
  status = next_kernel_events(t0, &k0, &k1);
  if (status != TC_OK) return status;
  if (t0->compute_dtype == TC_DTYPE_F32) {
    ca.pos_ij = (const int32_t*)t0->d_pos_ij;
    status = launch_contract_f32(grid, block, lds, L.stream, ca, k0, k1);
  } else {
    if (lds > 64 * 1024) {
      status = set_lds_limit_rt(t0->rt, lds);
      if (status != TC_OK) return status;
    }
    status = launch_contract_rt(t0->rt, grid, block, lds, L.stream, ca, k0, k1);
  }
  if (status != TC_OK) return status;
  t0->last_workgroups = (int)(grid.x * grid.z);
  t0->last_waves = c->host.waves_per_group;
  t0->last_splits = n_groups * k_splits;
  t0->last_lds = lds;

  tc::FinalizeArgs fa{};
  fa.partial = (const double*)L.partial.ptr;
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
  return launch_finalize(fa, t0->tuning, L.stream);
}

}  // namespace

namespace tc {
namespace host {
hipStream_t interp_stream(tc_interp* interp) { return interp->lanes[interp->cur].stream; }
// Everything queued so far on every lane of the interpolator precedes what is queued on
// `stream` afterwards (tc_comm_gather_interp) / every lane waits for `event`
// (tc_comm_release_interp).
int interp_join_lanes(tc_interp* interp, hipStream_t stream) {
  for (tc_interp::Lane& lane : interp->lanes) {
    if (lane.stream == nullptr) continue;
    TC_HIP(hipEventRecord(lane.finished, lane.stream));
    TC_HIP(hipStreamWaitEvent(stream, lane.finished, 0));
  }
  return TC_OK;
}
int interp_lanes_wait(tc_interp* interp, hipEvent_t event) {
  for (tc_interp::Lane& lane : interp->lanes)
    if (lane.stream != nullptr) TC_HIP(hipStreamWaitEvent(lane.stream, event, 0));
  return TC_OK;
}
}  // namespace host
}  // namespace tc

extern "C" {

int tc_interp_create(tc_table* const* tables, int n_tables, int n_dim,
                     const double* points, tc_interp** out) {
  TC_CHECK(out != nullptr, "interp output pointer is NULL");
  *out = nullptr;
  TC_CHECK(tables != nullptr && points != nullptr, "NULL argument");
  TC_CHECK(n_tables >= 1 && n_dim >= 1 && n_dim <= tc::kMaxInterpDim,
           "invalid number of tables or dimensions");
  struct Destroy {
    void operator()(tc_interp* interp) const { tc_interp_destroy(interp); }
  };
  std::unique_ptr<tc_interp, Destroy> it(new tc_interp);
  TC_HIP(hipGetDevice(&it->device));
  it->n_tables = n_tables;
  it->n_dim = n_dim;
  it->tables.assign(tables, tables + n_tables);
  tc_table* t0 = tables[0];
  for (int k = 0; k < n_tables; ++k) {
    tc_table* t = tables[k];
    TC_CHECK(t != nullptr, "table %d is NULL", k);
    TC_CHECK(t->device == it->device, "table %d lives on another device", k);
    TC_CHECK(t->compute_dtype == t0->compute_dtype,
             "table %d differs from table 0 in compute dtype", k);
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
  for (tc_interp::Lane& lane : it->lanes) {
    lane.nbuf.resize(n_classes);
    lane.ngal2.resize(n_classes);
    lane.nbuf32.resize(n_classes);
    lane.nbuf_ptrs.assign(n_classes, nullptr);
    lane.ngal_ptrs.assign(n_classes, nullptr);
    lane.nbuf32_ptrs.assign(n_classes, nullptr);
    TC_HIP(hipEventCreateWithFlags(&lane.finished, hipEventDisableTiming));
  }
  {
    hipStream_t streams[tc_interp::kLanes] = {};
    const int created = create_lane_streams(tc_interp::kLanes, streams);
    for (int l = 0; l < tc_interp::kLanes; ++l) it->lanes[l].stream = streams[l];
    if (created != TC_OK) return created;
  }
  it->stream = it->lanes[0].stream;
  it->n_lanes = t0->mode == TC_MODE_CROSS ? tc_interp::kLanes : 2;
  std::vector<void*> table_ptrs;
  for (int k = 0; k < n_tables; ++k) table_ptrs.push_back(tables[k]->d_table);
  std::vector<void*> zeros(n_classes, nullptr);
  std::vector<void*> quad_by_type, quad_total;
  for (int k = 0; k < n_tables; ++k) {
    quad_by_type.push_back(tables[k]->quad_by_type.d_table);
    quad_total.push_back(tables[k]->quad_total.d_table);
  }
  int status = upload(xp_all, &it->d_xp);
  if (status == TC_OK) status = upload(quad_by_type, &it->d_quad_by_type);
  if (status == TC_OK) status = upload(quad_total, &it->d_quad_total);
  it->a_host = a_all;
  if (status == TC_OK) status = upload(a_all, &it->d_a);
  if (status == TC_OK) status = upload(it->table_node, &it->d_table_node);
  if (status == TC_OK) status = upload(it->table_class, &it->d_table_class);
  if (status == TC_OK) status = upload(table_ptrs, &it->d_tables);
  for (tc_interp::Lane& lane : it->lanes) {
    if (status == TC_OK) status = upload(zeros, &lane.d_nbufs);
    if (status == TC_OK) status = upload(zeros, &lane.d_nbufs32);
    if (status == TC_OK) status = upload(zeros, &lane.d_ngal_parts);
  }
  if (status != TC_OK) return status;
  *out = it.release();
  return TC_OK;
}

int tc_interp_destroy(tc_interp* it) {
  if (it == nullptr) return TC_OK;
  (void)hipSetDevice(it->device);
  for (tc_interp::Lane& lane : it->lanes)
    if (lane.stream) (void)hipStreamSynchronize(lane.stream);
  for (void* p : {it->d_xp, it->d_a, it->d_table_node, it->d_table_class, it->d_tables,
                  it->d_quad_by_type, it->d_quad_total})
    if (p) (void)hipFree(p);
  for (auto& kv : it->chunkings)
    for (void* p : {kv.second->chunks, kv.second->groups})
      if (p) (void)hipFree(p);
  for (auto& kv : it->single_pointers)
    for (void* p : {kv.second.log_m, kv.second.m, kv.second.weight, kv.second.n_h,
                    kv.second.percentile})
      if (p) (void)hipFree(p);
  for (tc_interp::Lane& lane : it->lanes) {
    for (void* p : {lane.d_nbufs, lane.d_nbufs32, lane.d_ngal_parts})
      if (p) (void)hipFree(p);
    for (DeviceBuffer& b : lane.nbuf) b.release();
    for (DeviceBuffer& b : lane.ngal2) b.release();
    for (DeviceBuffer& b : lane.nbuf32) b.release();
    for (DeviceBuffer* b : {&lane.coef, &lane.partial, &lane.chi2_xi, &lane.stage_in,
                            &lane.stage_out, &lane.cross_counters})
      b->release();
    if (lane.finished) (void)hipEventDestroy(lane.finished);
  }
  for (DeviceBuffer* b : {&it->theta, &it->x, &it->out_ngal, &it->out_xi, &it->chi2_data})
    b->release();
  it->cross_fused.release();
  it->cross_fused_wide.release();
  it->h_in.release();
  it->h_out.release();
  it->single_ws.buffer.release();
  for (tc_table::Ticket& ticket : it->tickets)
    if (ticket.done) (void)hipEventDestroy(ticket.done);
  for (tc_interp::Lane& lane : it->lanes)
    if (lane.stream) (void)hipStreamDestroy(lane.stream);
  delete it;
  return TC_OK;
}

int tc_interp_synchronize(tc_interp* it) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  for (tc_interp::Lane& lane : it->lanes)
    if (lane.stream) TC_HIP(hipStreamSynchronize(lane.stream));
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
  it->cur = it->force_lane >= 0 ? it->force_lane
            : it->tables[0]->tuning.pipeline ? (int)(it->device_calls++ % it->n_lanes)
                                             : 0;
  const int64_t slab = max_slab(it->tables[0]);
  for (int64_t begin = 0; begin < n_draws; begin += slab) {
    const int64_t n = std::min(slab, n_draws - begin);
    status = interp_predict_device(
        it, theta_device + begin * n_theta, n_theta, x_device + begin * it->n_dim, n,
        n_gauss, flags, ngal_device + begin * (separate ? 2 : 1),
        xi_device + begin * n_comp * it->tables[0]->n_r);
    if (status != TC_OK) return status;
  }
  return TC_OK;
}

namespace {

// Un-batched Interpolator.predict: ONE launch evaluates every table (single_draw_kernel
// with grid = tables x workgroups per table, each workgroup computing the occupations of
// its table's class itself), the partial sums land in page-locked host memory, and the
// host normalises each table, forms the tensor-product spline weights
// (interpolator.py:275-331) and adds the tables in list order.  No device-side
// combination, no second launch: ~25 us against ~70 us for the four-launch batched path.
int interp_predict_one(tc_interp* it, const double* theta, int n_theta, const double* x,
                       int n_gauss, unsigned flags, double* ngal, double* xi) {
  tc_table* t0 = it->tables[0];
  const int n_classes = (int)it->class_table.size();
  auto found = it->single_pointers.find(n_gauss);
  if (found == it->single_pointers.end()) {
    std::vector<void*> log_m, m, weight, n_h, percentile;
    for (int v = 0; v < n_classes; ++v) {
      tc_table* t = it->tables[it->class_table[v]];
      Quadrature* q = nullptr;
      int status = get_quadrature(t, n_gauss, &q);
      if (status != TC_OK) return status;
      log_m.push_back(q->log_m);
      m.push_back(q->m);
      weight.push_back(q->weight);
      n_h.push_back(t->d_n_h);
      percentile.push_back(t->d_percentile);
    }
    tc_interp::SinglePointers p;
    int status = upload(log_m, &p.log_m);
    if (status == TC_OK) status = upload(m, &p.m);
    if (status == TC_OK) status = upload(weight, &p.weight);
    if (status == TC_OK) status = upload(n_h, &p.n_h);
    if (status == TC_OK) status = upload(percentile, &p.percentile);
    if (status != TC_OK) return status;
    found = it->single_pointers.emplace(n_gauss, p).first;
  }
  const tc_interp::SinglePointers& p = found->second;
  const int rt = t0->rt, n_tables = it->n_tables;
  // One workgroup of 1024 threads per CU at a time (16 of the 24 wave slots its registers
  // allow): all tables' workgroups in ONE round where that leaves each of them at most four
  // passes over its positions -- a workgroup that waits for a CU costs more than more passes
  // (tools/archive/r03_interp_one.py, un-batched predict(model), us per call, one pass per workgroup /
  // one round: 5 x 5 tables of 100 bins 35.0 / 31.3 (325 / 250 workgroups), 6 x 6 37.5 / 31.9,
  // 4 x 4 x 4 58.8 / 41.0 (832 / 256 workgroups); 4 x 4 fits anyway: 28.8)
  int blocks = single_draw_blocks(t0);
  const int one_round = t0->n_cus / std::max(1, n_tables);
  if (t0->tuning.single_round && one_round >= 1 && one_round < blocks && 4 * one_round >= blocks)
    blocks = one_round;
  SingleWorkspace& workspace = it->single_ws;
  int status = workspace.prepare(n_tables, blocks, rt, 0);
  if (status != TC_OK) return status;
  double* ws = workspace.ngal();
  tc::SingleArgs sa{};
  for (int i = 0; i < 7; ++i) sa.theta_value[i] = i < n_theta ? theta[i] : 0.0;
  sa.n_theta = n_theta;
  sa.n_bins = t0->n_bins;
  sa.n_central = t0->plan.n_central;
  sa.n_gauss = n_gauss;
  sa.flags = flags;
  sa.split = 0.5;
  sa.log_m = sa.m = sa.weight = sa.n_h = sa.percentile = nullptr;
  sa.math_table = (const double*)t0->d_math_table;
  sa.table = nullptr;
  sa.pos_off = (const int32_t*)t0->d_pos_off;
  sa.n_positions = t0->plan.n_positions;
  sa.rt = rt;
  sa.n_r = t0->n_r;
  sa.mode = t0->mode;
  sa.ngal = ws;
  sa.partial = workspace.partial();
  sa.done = workspace.done();
  sa.epoch = workspace.epoch;
  sa.tables = (const double* const*)it->d_tables;
  sa.table_class = (const int32_t*)it->d_table_class;
  sa.class_log_m = (const double* const*)p.log_m;
  sa.class_m = (const double* const*)p.m;
  sa.class_weight = (const double* const*)p.weight;
  sa.class_n_h = (const double* const*)p.n_h;
  sa.class_percentile = (const double* const*)p.percentile;
  status = launch_single_draw_tables(t0, sa, n_tables, blocks, it->stream);
  if (status != TC_OK) return status;

  // spline weights while the kernel runs
  double weight[tc::kMaxInterpDim][tc::kMaxInterpAxis];
  for (int d = 0; d < it->n_dim; ++d) {
    const std::vector<double>& xp = it->xp[d];
    const int n = (int)xp.size();
    const double xv = x[d];
    int seg = -1;
    for (int i = 0; i < n; ++i) seg += xp[i] <= xv ? 1 : 0;   // np.digitize(x, xp) - 1
    if (xv == xp[n - 1]) seg = n - 2;
    seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
    const double* m = it->a_host.data() + it->a_offset[d] + (size_t)seg * 4 * n;
    const double x2 = xv * xv, x3 = x2 * xv;
    for (int j = 0; j < n; ++j)
      weight[d][j] = m[j] + m[n + j] * xv + m[2 * n + j] * x2 + m[3 * n + j] * x3;
  }
  status = wait_single_done(&workspace, it->stream, t0->tuning.poll_done != 0);
  if (status != TC_OK) return status;
  const double* partial = workspace.partial();
  double n_cen = 0.0, n_sat = 0.0;
  for (int r = 0; r < t0->n_r; ++r) xi[r] = 0.0;
  for (int k = 0; k < n_tables; ++k) {
    double c = 1.0;
    for (int d = 0; d < it->n_dim; ++d) c *= weight[d][it->table_node[(size_t)k * it->n_dim + d]];
    const double cen = ws[2 * k], sat = ws[2 * k + 1];
    const double total = cen + sat;
    const double coef = c / (t0->mode == TC_MODE_AUTO ? total * total : total);
    n_cen += c * cen;
    n_sat += c * sat;
    const double* rows = partial + (size_t)k * blocks * rt;
    for (int r = 0; r < t0->n_r; ++r) {
      double sum = 0.0;
      for (int b = 0; b < blocks; ++b) sum += rows[(size_t)b * rt + r];
      xi[r] += coef * sum;
    }
  }
  ngal[0] = n_cen + n_sat;
  return TC_OK;
}

}  // namespace

namespace {
int interp_async(tc_interp* it, const double* theta, int n_theta, const double* x,
                 int64_t n_draws, int n_gauss, unsigned flags, const double* data,
                 const double* precision, double* ngal, double* second, bool chi2,
                 int64_t* ticket_out, bool staging);

// A synchronous host-array call as overlapping chunks of draws (table.cpp: predict_chunked has
// the reasoning): chunk k's draws and extra parameters are staged and queued on lane k % lanes
// while chunk k - 1 computes, its results are copied to the caller's arrays (on four host
// threads) while later chunks compute and travel.  The interpolator's kernels cut a draw's
// sums where the batch size puts them: chunks agree with one piece to rounding.
int interp_chunked(tc_interp* it, const double* theta, int n_theta, const double* x,
                   int64_t n_draws, int n_gauss, unsigned flags, double* ngal, double* xi,
                   int n_chunks) {
  tc_table* t0 = it->tables[0];
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const size_t ngal_cols = separate ? 2 : 1;
  const size_t xi_cols = (size_t)(separate ? t0->plan.n_components : 1) * t0->n_r;
  const size_t in_cols = (size_t)n_theta + it->n_dim;
  int status = it->h_in.reserve((size_t)n_draws * in_cols * 8);
  if (status == TC_OK) status = it->h_out.reserve((size_t)n_draws * (ngal_cols + xi_cols) * 8);
  if (status != TC_OK) return status;
  const int64_t chunk = ((n_draws + n_chunks - 1) / n_chunks + 63) / 64 * 64;
  n_chunks = (int)((n_draws + chunk - 1) / chunk);
  TC_CHECK(n_chunks <= 64, "internal: too many chunks");
  int64_t tickets[64];
  double* h_in = (double*)it->h_in.ptr;
  double* h_out = (double*)it->h_out.ptr;
  it->tables[0]->sync_cross_target = 512 / n_chunks;       // (internal.h)
  for (int k = 0; k < n_chunks && status == TC_OK; ++k) {
    const int64_t begin = k * chunk, n = std::min(chunk, n_draws - begin);
    double* in = h_in + begin * in_cols;          // [theta | x] of the chunk
    memcpy(in, theta + begin * n_theta, (size_t)n * n_theta * 8);
    memcpy(in + n * n_theta, x + begin * it->n_dim, (size_t)n * it->n_dim * 8);
    double* out = h_out + begin * (ngal_cols + xi_cols);
    status = interp_async(it, in, n_theta, in + n * n_theta, n, n_gauss, flags, nullptr, nullptr,
                          out, out + n * ngal_cols, false, &tickets[k], true);
    if (status != TC_OK) n_chunks = k;
  }
  it->tables[0]->sync_cross_target = 0;
  for (int k = 0; k < n_chunks; ++k) {
    const int64_t begin = k * chunk, n = std::min(chunk, n_draws - begin);
    const int waited = tc_interp_wait(it, tickets[k]);
    if (waited != TC_OK) {
      (void)tc_interp_synchronize(it);
      return waited;
    }
    if (status != TC_OK) continue;
    const double* out = h_out + begin * (ngal_cols + xi_cols);
    memcpy(ngal + begin * ngal_cols, out, (size_t)n * ngal_cols * 8);
    parallel_copy(xi + begin * xi_cols, out + n * ngal_cols, (size_t)n * xi_cols * 8);
  }
  return status;
}
}  // namespace

int tc_interp_predict_zheng07_batch(tc_interp* it, const double* theta, int n_theta,
                                    const double* x, int64_t n_draws, int n_gauss,
                                    unsigned flags, double* ngal, double* xi) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  int status = check_predict_args(it->tables[0], theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(x && ngal && xi, "NULL pointer");
  TC_HIP(hipSetDevice(it->device));
  // (option "deterministic" = 2 of the first table, mode cross: the one-launch form for every
  // batch size, one draw included)
  bool invariant = false;
  if (it->tables[0]->tuning.deterministic >= 2 && it->tables[0]->mode == TC_MODE_CROSS &&
      !it->tables[0]->cross_host.empty() && it->tables[0]->tuning.fused != 0) {
    const CrossFused& cf = *choose_cross_fused(it->tables.data(), it->n_tables, &it->cross_fused,
                                               &it->cross_fused_wide, n_draws, flags, &status);
    if (status != TC_OK) return status;
    invariant = cross_fused_eligible(it->tables[0], cf, n_draws, n_gauss, flags, true);
  }
  if (!invariant && single_draw_eligible(it->tables[0], n_draws, n_gauss, flags) &&
      (int64_t)it->n_tables * single_draw_blocks(it->tables[0]) <= 8192)
    return interp_predict_one(it, theta, n_theta, x, n_gauss, flags, ngal, xi);
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  const int n_comp = separate ? it->tables[0]->plan.n_components : 1;
  const size_t ngal_count = (size_t)n_draws * (separate ? 2 : 1);
  const size_t xi_count = (size_t)n_draws * n_comp * it->tables[0]->n_r;
  const size_t theta_count = (size_t)n_draws * n_theta, x_count = (size_t)n_draws * it->n_dim;
  // small and medium calls: the kernels address page-locked host buffers directly
  // (table.cpp: tc_predict_zheng07_batch)
  if ((theta_count + x_count + ngal_count + xi_count) * 8 <= zero_copy_limit() &&
      it->h_in.reserve((theta_count + x_count) * 8) == TC_OK &&
      it->h_out.reserve((ngal_count + xi_count) * 8) == TC_OK) {
    double* in = (double*)it->h_in.ptr;
    double* out = (double*)it->h_out.ptr;
    memcpy(in, theta, theta_count * 8);
    memcpy(in + theta_count, x, x_count * 8);
    it->force_lane = 0;
    status = tc_interp_predict_zheng07_batch_device(it, in, n_theta, in + theta_count,
                                                    n_draws, n_gauss, flags, out,
                                                    out + ngal_count);
    it->force_lane = -1;
    if (status != TC_OK) return status;
    TC_HIP(hipStreamSynchronize(it->stream));
    memcpy(ngal, out, ngal_count * 8);
    memcpy(xi, out + ngal_count, xi_count * 8);
    return TC_OK;
  }
  {
    // (option "sync_chunks" of the first table: 0 = 2 .. 8 chunks of about a megabyte of results
    // from 2048 draws on, N >= 1 that many, -1 the serial path below)
    const tc_table* t0 = it->tables[0];
    int n_chunks = 0;
    if (t0->tuning.sync_chunks >= 0 && t0->tuning.pipeline && it->n_lanes >= 2) {
      if (t0->tuning.sync_chunks >= 1)
        n_chunks = (int)std::min<int64_t>(t0->tuning.sync_chunks, (n_draws + 63) / 64);
      else if (n_draws >= 2048 && t0->mode == TC_MODE_CROSS)
        // (one piece through the asynchronous path: staging, kernels that store the results
        // themselves, the copy on four threads.  Two and more chunks of an interpolator's
        // one-launch form take 217-470 us per 10^4 draws of the AbacusSummit fixture from one
        // batch of calls to the next -- the runtime's mapping of the lanes' streams to hardware
        // queues is the suspect --, one chunk a steady 245; a single table's chunks are steady:
        // 168 -> 143, tools/r05_sync_chunks.py)
        n_chunks = 1;
      else if (n_draws >= 2048)
        n_chunks = (int)std::min<int64_t>(
            std::min<int64_t>(8, n_draws / 1024),
            std::max<int64_t>(2, (int64_t)((ngal_count + xi_count) * 8 >> 20)));
    }
    if (n_chunks > 0)
      return interp_chunked(it, theta, n_theta, x, n_draws, n_gauss, flags, ngal, xi, n_chunks);
  }
  status = it->theta.reserve((size_t)n_draws * n_theta * 8, it->stream);
  if (status == TC_OK) status = it->x.reserve((size_t)n_draws * it->n_dim * 8, it->stream);
  if (status == TC_OK) status = it->out_ngal.reserve(ngal_count * 8, it->stream);
  if (status == TC_OK) status = it->out_xi.reserve(xi_count * 8, it->stream);
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(it->theta.ptr, theta, (size_t)n_draws * n_theta * 8,
                        hipMemcpyHostToDevice, it->stream));
  TC_HIP(hipMemcpyAsync(it->x.ptr, x, (size_t)n_draws * it->n_dim * 8,
                        hipMemcpyHostToDevice, it->stream));
  it->force_lane = 0;
  status = tc_interp_predict_zheng07_batch_device(
      it, (const double*)it->theta.ptr, n_theta, (const double*)it->x.ptr, n_draws,
      n_gauss, flags, (double*)it->out_ngal.ptr, (double*)it->out_xi.ptr);
  it->force_lane = -1;
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(ngal, it->out_ngal.ptr, ngal_count * 8, hipMemcpyDeviceToHost,
                        it->stream));
  TC_HIP(hipMemcpyAsync(xi, it->out_xi.ptr, xi_count * 8, hipMemcpyDeviceToHost,
                        it->stream));
  TC_HIP(hipStreamSynchronize(it->stream));
  return TC_OK;
}

int tc_interp_chi2_zheng07_batch_device(tc_interp* it, const double* theta_device,
                                        int n_theta, const double* x_device, int64_t n_draws,
                                        int n_gauss, unsigned flags, const double* data,
                                        const double* precision, double* ngal_device,
                                        double* chi2_device) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  TC_CHECK(!(flags & TC_FLAG_SEPARATE_GAL_TYPE),
           "chi2 is defined for the total correlation function only");
  if (n_draws == 0) return TC_OK;
  TC_CHECK(data && precision && ngal_device && chi2_device, "NULL pointer");
  TC_HIP(hipSetDevice(it->device));
  const int n_r = it->tables[0]->n_r;
  tc_table* t0 = it->tables[0];
  // the lane tc_interp_predict_zheng07_batch_device is about to pick
  const int lane_index = it->force_lane >= 0 ? it->force_lane
                         : t0->tuning.pipeline ? (int)(it->device_calls % it->n_lanes)
                                               : 0;
  tc_interp::Lane& L = it->lanes[lane_index];
  // data vector and precision matrix: uploaded when they differ from the last upload
  const size_t data_count = (size_t)(n_r + 1) * n_r;
  int status = it->chi2_data.reserve(data_count * 8, it->stream);
  if (status != TC_OK) return status;
  if (it->chi2_host.size() != data_count ||
      memcmp(it->chi2_host.data(), data, (size_t)n_r * 8) != 0 ||
      memcmp(it->chi2_host.data() + n_r, precision, (size_t)n_r * n_r * 8) != 0) {
    // (earlier kernels of any lane may still read the old ones)
    status = tc_interp_synchronize(it);
    if (status != TC_OK) return status;
    it->chi2_host.assign(data, data + n_r);
    it->chi2_host.insert(it->chi2_host.end(), precision, precision + (size_t)n_r * n_r);
    TC_HIP(hipMemcpyAsync(it->chi2_data.ptr, it->chi2_host.data(), data_count * 8,
                          hipMemcpyHostToDevice, it->stream));
    TC_HIP(hipStreamSynchronize(it->stream));
  }
  status = L.chi2_xi.reserve((size_t)n_draws * n_r * 8, L.stream);
  if (status != TC_OK) return status;
  const double* d_data = (const double*)it->chi2_data.ptr;
  t0->chi2_fused = false;
  t0->fuse_chi2_data = d_data;
  t0->fuse_chi2_out = n_draws <= max_slab(t0) ? chi2_device : nullptr;   // (one slab)
  status = tc_interp_predict_zheng07_batch_device(it, theta_device, n_theta, x_device, n_draws,
                                                  n_gauss, flags, ngal_device,
                                                  (double*)L.chi2_xi.ptr);
  const bool fused = t0->fuse_chi2_out != nullptr && t0->chi2_fused;
  t0->fuse_chi2_out = nullptr;
  if (status != TC_OK) return status;
  if (fused) return TC_OK;
  return launch_chi2((const double*)L.chi2_xi.ptr, n_draws, n_r, d_data, d_data + n_r,
                     chi2_device, L.stream);
}

int tc_interp_chi2_zheng07_batch(tc_interp* it, const double* theta, int n_theta,
                                 const double* x, int64_t n_draws, int n_gauss,
                                 unsigned flags, const double* data, const double* precision,
                                 double* ngal, double* chi2) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  int status = check_predict_args(it->tables[0], theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  if (n_draws == 0) return TC_OK;
  TC_CHECK(x && data && precision && ngal && chi2, "NULL pointer");
  TC_HIP(hipSetDevice(it->device));
  const size_t theta_bytes = (size_t)n_draws * n_theta * 8;
  const size_t x_bytes = (size_t)n_draws * it->n_dim * 8;
  status = it->theta.reserve(theta_bytes, it->stream);
  if (status == TC_OK) status = it->x.reserve(x_bytes, it->stream);
  if (status == TC_OK) status = it->out_ngal.reserve((size_t)n_draws * 2 * 8, it->stream);
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(it->theta.ptr, theta, theta_bytes, hipMemcpyHostToDevice, it->stream));
  TC_HIP(hipMemcpyAsync(it->x.ptr, x, x_bytes, hipMemcpyHostToDevice, it->stream));
  double* d_ngal = (double*)it->out_ngal.ptr;
  double* d_chi2 = d_ngal + n_draws;
  it->force_lane = 0;
  status = tc_interp_chi2_zheng07_batch_device(it, (const double*)it->theta.ptr, n_theta,
                                               (const double*)it->x.ptr, n_draws, n_gauss,
                                               flags, data, precision, d_ngal, d_chi2);
  it->force_lane = -1;
  if (status != TC_OK) return status;
  TC_HIP(hipMemcpyAsync(ngal, d_ngal, (size_t)n_draws * 8, hipMemcpyDeviceToHost, it->stream));
  TC_HIP(hipMemcpyAsync(chi2, d_chi2, (size_t)n_draws * 8, hipMemcpyDeviceToHost, it->stream));
  TC_HIP(hipStreamSynchronize(it->stream));
  return TC_OK;
}

namespace {

int interp_async(tc_interp* it, const double* theta, int n_theta, const double* x,
                 int64_t n_draws, int n_gauss, unsigned flags, const double* data,
                 const double* precision, double* ngal, double* second, bool chi2,
                 int64_t* ticket_out, bool staging) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  tc_table* t0 = it->tables[0];
  int status = check_predict_args(t0, theta, n_theta, n_draws, n_gauss, flags);
  if (status != TC_OK) return status;
  TC_CHECK(ticket_out != nullptr, "ticket is NULL");
  TC_CHECK(n_draws == 0 || (x && ngal && second), "NULL pointer");
  const bool separate = (flags & TC_FLAG_SEPARATE_GAL_TYPE) != 0;
  TC_CHECK(!(chi2 && separate), "chi2 is defined for the total correlation function only");
  TC_CHECK(!chi2 || (data && precision), "NULL pointer");
  const int n_comp = separate ? t0->plan.n_components : 1;
  const size_t ngal_count = (size_t)n_draws * (separate ? 2 : 1);
  const size_t second_count = chi2 ? (size_t)n_draws : (size_t)n_draws * n_comp * t0->n_r;
  const size_t theta_count = (size_t)n_draws * n_theta, x_count = (size_t)n_draws * it->n_dim;
  // (staging: the library's own page-locked staging areas -- the chunks of a synchronous call)
  TC_CHECK(n_draws == 0 || staging ||
               (is_pinned(theta, theta_count * 8) && is_pinned(x, x_count * 8) &&
                is_pinned(ngal, ngal_count * 8) && is_pinned(second, second_count * 8)),
           "asynchronous calls need page-locked buffers (tc_host_alloc / tc_host_register)");
  TC_HIP(hipSetDevice(it->device));
  // everything of a call -- upload, kernels, download, the ticket's event -- on the next lane's
  // stream: the lanes overlap each other's transfers and kernels
  const int lane_index =
      t0->tuning.pipeline ? (int)(it->device_calls % it->n_lanes) : 0;
  tc_interp::Lane& L = it->lanes[lane_index];
  hipStream_t last = L.stream;
  if (n_draws > 0) {
    status = L.stage_out.reserve((ngal_count + second_count) * 8, L.stream);
    if (status == TC_OK) status = L.stage_in.reserve((theta_count + x_count) * 8, L.stream);
    if (status != TC_OK) return status;
    double* d_theta = (double*)L.stage_in.ptr;
    double* d_x = d_theta + theta_count;
    double* d_ngal = (double*)L.stage_out.ptr;
    double* d_second = d_ngal + ngal_count;
    {
      Range range("upload");
      TC_HIP(hipMemcpyAsync(d_theta, theta, theta_count * 8, hipMemcpyHostToDevice, L.stream));
      TC_HIP(hipMemcpyAsync(d_x, x, x_count * 8, hipMemcpyHostToDevice, L.stream));
    }
    status = chi2 ? tc_interp_chi2_zheng07_batch_device(it, d_theta, n_theta, d_x, n_draws,
                                                        n_gauss, flags, data, precision, d_ngal,
                                                        d_second)
                  : tc_interp_predict_zheng07_batch_device(it, d_theta, n_theta, d_x, n_draws,
                                                           n_gauss, flags, d_ngal, d_second);
    if (status != TC_OK) return status;
    Range range("download");
    TC_HIP(hipMemcpyAsync(ngal, d_ngal, ngal_count * 8, hipMemcpyDeviceToHost, L.stream));
    TC_HIP(hipMemcpyAsync(second, d_second, second_count * 8, hipMemcpyDeviceToHost, L.stream));
  } else if (t0->tuning.pipeline) {
    ++it->device_calls;
  }
  tc_table::Ticket& slot = it->tickets[it->next_ticket % tc_table::kMaxTickets];
  if (slot.done == nullptr)
    TC_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
  slot.id = it->next_ticket++;
  TC_HIP(hipEventRecord(slot.done, last));
  *ticket_out = slot.id;
  return TC_OK;
}

}  // namespace

int tc_interp_predict_zheng07_batch_async(tc_interp* it, const double* theta, int n_theta,
                                          const double* x, int64_t n_draws, int n_gauss,
                                          unsigned flags, double* ngal, double* xi,
                                          int64_t* ticket) {
  return interp_async(it, theta, n_theta, x, n_draws, n_gauss, flags, nullptr, nullptr, ngal, xi,
                      false, ticket, false);
}

int tc_interp_chi2_zheng07_batch_async(tc_interp* it, const double* theta, int n_theta,
                                       const double* x, int64_t n_draws, int n_gauss,
                                       unsigned flags, const double* data,
                                       const double* precision, double* ngal, double* chi2,
                                       int64_t* ticket) {
  return interp_async(it, theta, n_theta, x, n_draws, n_gauss, flags, data, precision, ngal,
                      chi2, true, ticket, false);
}

int tc_interp_wait(tc_interp* it, int64_t ticket) {
  TC_CHECK(it != nullptr, "interp handle is NULL");
  TC_CHECK(ticket >= 0 && ticket < it->next_ticket, "unknown ticket %lld", (long long)ticket);
  const tc_table::Ticket& slot = it->tickets[ticket % tc_table::kMaxTickets];
  if (slot.id == ticket) {
    TC_HIP(hipEventSynchronize(slot.done));
    return TC_OK;
  }
  // the slot was reused: the ticket is older than everything queued now
  return tc_interp_synchronize(it);
}

int tc_interp_query(tc_interp* it, int64_t ticket, int* done) {
  TC_CHECK(it != nullptr && done != nullptr, "NULL argument");
  TC_CHECK(ticket >= 0 && ticket < it->next_ticket, "unknown ticket %lld", (long long)ticket);
  const tc_table::Ticket& slot = it->tickets[ticket % tc_table::kMaxTickets];
  *done = 0;
  // (a reused slot: the ticket is done once both lanes have passed their later work)
  hipError_t state = hipSuccess;
  if (slot.id == ticket) {
    state = hipEventQuery(slot.done);
  } else {
    for (tc_interp::Lane& lane : it->lanes) {
      const hipError_t lane_state = hipStreamQuery(lane.stream);
      if (lane_state != hipSuccess) state = lane_state;
    }
  }
  if (state == hipSuccess) *done = 1;
  else if (state != hipErrorNotReady)
    return fail(TC_ERR_HIP, "query failed: %s", hipGetErrorString(state));
  return TC_OK;
}

}  // extern "C"
