// Pair counting for the tabulation step: DD(r_p, pi) in a periodic
// box on the GPU.
//
// In the reference, TabCorr.tabulate fills the correlation matrix by calling the two-point
// function once per pair of halo bins from a multiprocessing pool (tabcorr/tabcorr.py:846-922)
// and, with the Corrfunc backend, every such call is one Corrfunc.theory.DDrppi pair count
// (tabcorr/corrfunc.py:62-84).  Here:
//   * tc_pair_count_rppi is that one call: ordered pair counts per (r_p, pi) bin between two
//     point sets (or of one set with itself);
//   * tc_pair_count_rppi_labelled is the whole loop in one launch: every point carries the
//     index of its halo bin and ONE pass over the box fills count[r_p bin][label 1][label 2]
//     for all bin pairs at once -- the points of all bins share one cell grid and every
//     close pair is visited once, instead of G (G + 1) / 2 separate pair counts.
//
// Pair definition (Corrfunc 2.x DDrppi, restated from its documentation -- Corrfunc is a
// third-party dependency that is not vendored in the reference; oracle/paircount_oracle.py
// is the executable statement of these rules and the kernel is bit-exact against it):
//   dx, dy, dz = minimum-image separations in the periodic box (each coordinate difference
//   d -> d - L if d > L / 2, d + L if d < -L / 2); r_p^2 = dx dx + dy dy with each product and
//   the sum rounded separately (no fused multiply-add); a pair counts if |dz| < pi_max and
//   rp_bins[0]^2 <= r_p^2 < rp_bins[-1]^2; its r_p bin is the last k with rp_bins[k]^2 <=
//   r_p^2, its pi bin int(|dz| * (n_pi / pi_max)).  All ORDERED pairs (i, j) are counted, so
//   an auto-count holds every pair twice; i == j is a pair like any other (it only counts
//   when rp_bins[0] == 0).
//
// Layout: the host sorts the points into a grid of cells at least r_p,max (x, y) and pi_max
// (z) wide, so that the partners of a point lie in the 27 cells around it (a dimension with
// fewer than three cells has one cell and no neighbour offsets: the minimum image does the
// wrapping).  A workgroup owns up to 256 consecutive points of one cell -- one per lane --
// and streams the points of the neighbouring cells through LDS in tiles of 256 (the classic
// all-pairs tiling: one global load per point and tile, 256 distance tests per load).
// Counters are integers: per-workgroup LDS histograms flushed with 64-bit global atomics --
// (r_p, pi) bins, or, for the labelled matrix, (r_p bin, label slot, partner label): the points
// are sorted by label inside the cells and a workgroup takes at most 8 distinct labels, so its
// private counters fit 56 KB (64-bit global atomics per pair only when not even one label
// slot fits: 4x slower).
// The result does not depend on the order of the atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "internal.h"

namespace tc {

constexpr int kPairThreads = 256;
constexpr int kMaxRpBins = 64;

struct PairArgs {
  // set 1 ("i", one lane each) and set 2 ("j", streamed): positions sorted by cell
  const double* x1;
  const double* y1;
  const double* z1;
  const int32_t* label1;
  const double* x2;
  const double* y2;
  const double* z2;
  const int32_t* label2;
  const int32_t* cell_start2;   // (n_cells + 1) offsets into set 2
  const int32_t* item_cell;     // per workgroup: its cell ...
  const int32_t* item_begin;    // ... and its range of set-1 points
  const int32_t* item_end;
  int nx, ny, nz;               // cells per dimension
  int reach_x, reach_y, reach_z;   // neighbour offsets per dimension: 1, or 0 for one cell
  double lx, ly, lz;            // box size
  double edge_sqr[kMaxRpBins + 1];
  int n_rp;
  int n_pi;
  double pi_max;
  double inv_dpi;               // n_pi / pi_max
  int n_labels;
  // labelled counts with private counters: the workgroup's points carry at most label_slots
  // distinct labels (slot1: the slot of every set-1 point, item_labels: (items, label_slots))
  // and LDS holds (n_rp [* n_mu], label_slots, n_labels) counters; 0: 64-bit global atomics
  int label_slots;
  const int32_t* slot1;
  const int32_t* item_labels;
  unsigned long long* counts;   // (n_rp, n_pi) or (n_rp [, n_mu], n_labels, n_labels)
};

__device__ inline double min_image(double d, double box, double half) {
  d = d > half ? d - box : d;
  return d < -half ? d + box : d;
}

// SMU: bins in the three-dimensional separation s and mu = |dz| / s (Corrfunc DDsmu, the pair
// count behind s_mu_tpcf, tabcorr/corrfunc.py:141-163): s^2 = (dx dx + dy dy) + dz dz, a pair
// counts if s_bins[0]^2 <= s^2 < s_bins[-1]^2 and mu < 1, its mu bin is int(mu n_mu) with
// mu = |dz| / sqrt(s^2) (correctly rounded square root and division, as NumPy's); s = 0
// (i == j with s_bins[0] == 0) goes to mu bin 0.
template <bool LABELLED, bool SMU>
__global__ __launch_bounds__(kPairThreads) void pair_count_kernel(PairArgs a) {
  __shared__ double sx[kPairThreads], sy[kPairThreads], sz[kPairThreads];
  __shared__ int32_t sl[kPairThreads];
  extern __shared__ unsigned hist[];   // n_rp * n_pi, or n_rp * label_slots * n_labels counters
  const int tid = threadIdx.x;
  const int n_hist = (LABELLED ? a.label_slots * a.n_labels : 1) * a.n_rp * a.n_pi;
  for (int k = tid; k < n_hist; k += kPairThreads) hist[k] = 0u;

  const int cell = a.item_cell[blockIdx.x];
  const int begin = a.item_begin[blockIdx.x], end = a.item_end[blockIdx.x];
  const int i = begin + tid;
  const bool active = i < end;
  const double xi = active ? a.x1[i] : 0.0, yi = active ? a.y1[i] : 0.0;
  const double zi = active ? a.z1[i] : 0.0;
  const int li = LABELLED && active ? a.label1[i] : 0;
  const int slot = LABELLED && active && a.label_slots > 0 ? a.slot1[i] : 0;
  const int cz = cell % a.nz, cy = (cell / a.nz) % a.ny, cx = cell / (a.nz * a.ny);
  const double hx = 0.5 * a.lx, hy = 0.5 * a.ly, hz = 0.5 * a.lz;
  const double lo_sqr = a.edge_sqr[0], hi_sqr = a.edge_sqr[a.n_rp];
  __syncthreads();

  for (int ox = -a.reach_x; ox <= a.reach_x; ++ox)
    for (int oy = -a.reach_y; oy <= a.reach_y; ++oy)
      for (int oz = -a.reach_z; oz <= a.reach_z; ++oz) {
        const int nxc = (cx + ox + a.nx) % a.nx, nyc = (cy + oy + a.ny) % a.ny;
        const int nzc = (cz + oz + a.nz) % a.nz;
        const int other = (nxc * a.ny + nyc) * a.nz + nzc;
        const int j_begin = a.cell_start2[other], j_end = a.cell_start2[other + 1];
        for (int j0 = j_begin; j0 < j_end; j0 += kPairThreads) {
          const int n_tile = j_end - j0 < kPairThreads ? j_end - j0 : kPairThreads;
          __syncthreads();
          if (tid < n_tile) {
            sx[tid] = a.x2[j0 + tid];
            sy[tid] = a.y2[j0 + tid];
            sz[tid] = a.z2[j0 + tid];
            if (LABELLED) sl[tid] = a.label2[j0 + tid];
          }
          __syncthreads();
          if (!active) continue;
          for (int t = 0; t < n_tile; ++t) {
            const double dz = fabs(min_image(zi - sz[t], a.lz, hz));
            if (!(dz < a.pi_max)) continue;
            const double dx = min_image(xi - sx[t], a.lx, hx);
            const double dy = min_image(yi - sy[t], a.ly, hy);
            // (separately rounded products and sums: the oracle's arithmetic)
            double r_sqr = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if (SMU) r_sqr = __dadd_rn(r_sqr, __dmul_rn(dz, dz));
            if (!(r_sqr >= lo_sqr && r_sqr < hi_sqr)) continue;
            int bin = 0;
            for (int k = 1; k < a.n_rp; ++k) bin += r_sqr >= a.edge_sqr[k] ? 1 : 0;
            int mu_bin = 0;
            if (SMU) {
              const double mu = r_sqr > 0.0 ? __ddiv_rn(dz, __dsqrt_rn(r_sqr)) : 0.0;
              mu_bin = (int)(mu * a.inv_dpi);
              if (!(mu < 1.0 && mu_bin < a.n_pi)) continue;
            }
            if (LABELLED) {
              // (labelled r_p counts are summed over pi: n_pi == 1 there)
              const int cell_bin = SMU ? bin * a.n_pi + mu_bin : bin;
              if (a.label_slots > 0)
                atomicAdd(&hist[(cell_bin * a.label_slots + slot) * a.n_labels + sl[t]], 1u);
              else
                atomicAdd(a.counts + ((size_t)cell_bin * a.n_labels + li) * a.n_labels + sl[t],
                          1ull);
            } else if (SMU) {
              atomicAdd(&hist[bin * a.n_pi + mu_bin], 1u);
            } else {
              const int pi_bin = (int)(dz * a.inv_dpi);
              if (pi_bin < a.n_pi) atomicAdd(&hist[bin * a.n_pi + pi_bin], 1u);
            }
          }
        }
      }
  __syncthreads();
  for (int k = tid; k < n_hist; k += kPairThreads) {
    const unsigned value = hist[k];
    if (value == 0u) continue;
    if (LABELLED) {
      const int lj = k % a.n_labels, s = (k / a.n_labels) % a.label_slots;
      const int bin = k / (a.n_labels * a.label_slots);
      const int label = a.item_labels[blockIdx.x * a.label_slots + s];
      atomicAdd(a.counts + ((size_t)bin * a.n_labels + label) * a.n_labels + lj,
                (unsigned long long)value);
    } else {
      atomicAdd(a.counts + k, (unsigned long long)value);
    }
  }
}

namespace host {

namespace {

struct DeviceArrays {
  std::vector<void*> pointers;
  ~DeviceArrays() {
    for (void* p : pointers)
      if (p) (void)hipFree(p);
  }
  template <typename T>
  int put(const std::vector<T>& host, const T** out) {
    void* device = nullptr;
    int status = upload(host, &device);
    if (status != TC_OK) return status;
    pointers.push_back(device);
    *out = (const T*)device;
    return TC_OK;
  }
};

// smu: rp_bins are the s bins, n_pi the number of mu bins on [0, 1), pi_max is ignored (the
// line-of-sight reach is the largest s).
int pair_count(const double* pos1, const int32_t* label1, int64_t n1, const double* pos2,
               const int32_t* label2, int64_t n2, int n_labels, const double* boxsize,
               const double* rp_bins, int n_rp, double pi_max, int n_pi, uint64_t* counts,
               bool smu = false) {
  TC_CHECK(pos1 != nullptr && boxsize != nullptr && rp_bins != nullptr && counts != nullptr,
           "NULL argument");
  TC_CHECK(n1 >= 0 && n2 >= 0 && n1 < (1LL << 31) && n2 < (1LL << 31), "invalid point count");
  TC_CHECK(n_rp >= 1 && n_rp <= kMaxRpBins, "between 1 and %d r_p bins are supported",
           kMaxRpBins);
  if (smu) pi_max = rp_bins[n_rp];
  TC_CHECK(pi_max > 0.0 && n_pi >= 1, "pi_max and the number of pi / mu bins must be positive");
  for (int k = 0; k <= n_rp; ++k)
    TC_CHECK(rp_bins[k] >= 0.0 && (k == 0 || rp_bins[k] > rp_bins[k - 1]),
             "rp_bins must be non-negative and increasing");
  const bool labelled = n_labels > 0;
  const bool autocorr = pos2 == nullptr;
  if (autocorr) {
    pos2 = pos1;
    label2 = label1;
    n2 = n1;
  }
  const double rp_max = rp_bins[n_rp];
  TC_CHECK(boxsize[0] > 0 && boxsize[1] > 0 && boxsize[2] > 0, "box size must be positive");
  // the minimum image is only the nearest image below half a box
  TC_CHECK(rp_max < 0.5 * std::min(boxsize[0], boxsize[1]) && pi_max < 0.5 * boxsize[2],
           "the largest separation must be smaller than half the box size");
  // (labelled r_p counts arrive with n_pi == 1: summed over the line of sight)
  const size_t n_counts = (size_t)n_rp * n_pi * (labelled ? (size_t)n_labels * n_labels : 1);
  std::fill(counts, counts + n_counts, (uint64_t)0);
  if (n1 == 0 || n2 == 0) return TC_OK;
  if (labelled) {
    TC_CHECK(label1 != nullptr && label2 != nullptr, "labels are NULL");
    for (int64_t p = 0; p < n1; ++p)
      TC_CHECK(label1[p] >= 0 && label1[p] < n_labels, "label of point %lld out of range",
               (long long)p);
    if (!autocorr)
      for (int64_t p = 0; p < n2; ++p)
        TC_CHECK(label2[p] >= 0 && label2[p] < n_labels, "label of point %lld out of range",
                 (long long)p);
  }
  // Labelled counts with enough points per (cell, label) keep private counters per workgroup
  // (below); they need full workgroups, i.e. the coarse grid.  Everything else takes the fine
  // grid (an eighth of the cell volume, 5 x 5 x 5 neighbours: 42 % fewer candidate pairs).
  int label_slots = 0;
  CellGrid grid = make_cell_grid(boxsize, rp_max, pi_max, std::max(n1, n2), !labelled);
  if (labelled) {
    label_slots =
        (int)std::min<size_t>(8, (56 * 1024) / ((size_t)n_rp * n_pi * n_labels * sizeof(unsigned)));
    // with few points per (cell, label) such workgroups would hold a handful of points each,
    // every one of them streaming all neighbour tiles (10^5 points in 100 bins: 48 ms against
    // 13 ms with global atomics; 10^6 points: 645 ms against 790 ms)
    const double per_cell_label = (double)n1 / ((double)grid.n_cells() * n_labels);
    if (label_slots * per_cell_label < 128.0) {
      label_slots = 0;
      grid = make_cell_grid(boxsize, rp_max, pi_max, std::max(n1, n2), true);
    }
  }

  Range range("pair count");
  CellSort set1, set2;
  int64_t outside = sort_into_cells(grid, pos1, labelled ? label1 : nullptr, n1, set1);
  TC_CHECK(outside < 0, "point %lld of the first sample lies outside of the periodic box",
           (long long)outside);
  if (!autocorr) {
    outside = sort_into_cells(grid, pos2, labelled ? label2 : nullptr, n2, set2);
    TC_CHECK(outside < 0, "point %lld of the second sample lies outside of the periodic box",
             (long long)outside);
  }
  // work items: up to 256 consecutive set-1 points of one cell; labelled: sorted by label
  // inside the cells and cut so that a workgroup sees at most `label_slots` labels, as many
  // as keep its private counters within 56 KB of LDS (none fit: global atomics per pair)
  std::vector<int32_t> item_cell, item_begin, item_end;
  LabelItems label_items;
  if (label_slots > 0) {
    sort_cells_by_label(set1);
    if (!autocorr) sort_cells_by_label(set2);
  }
  if (label_slots > 0) {
    build_label_items(set1, kPairThreads, label_slots, label_items);
    item_cell = label_items.cell;
    item_begin = label_items.begin;
    item_end = label_items.end;
  } else {
    for (int c = 0; c < grid.n_cells(); ++c)
      for (int32_t b = set1.cell_start[c]; b < set1.cell_start[c + 1]; b += kPairThreads) {
        item_cell.push_back(c);
        item_begin.push_back(b);
        item_end.push_back(std::min<int32_t>(b + kPairThreads, set1.cell_start[c + 1]));
      }
  }

  DeviceArrays device;
  PairArgs a{};
  int status = device.put(set1.x, &a.x1);
  if (status == TC_OK) status = device.put(set1.y, &a.y1);
  if (status == TC_OK) status = device.put(set1.z, &a.z1);
  a.label1 = nullptr;
  if (status == TC_OK && labelled) status = device.put(set1.label, &a.label1);
  const CellSort& second = autocorr ? set1 : set2;
  if (autocorr) {
    a.x2 = a.x1;
    a.y2 = a.y1;
    a.z2 = a.z1;
    a.label2 = a.label1;
  } else {
    if (status == TC_OK) status = device.put(set2.x, &a.x2);
    if (status == TC_OK) status = device.put(set2.y, &a.y2);
    if (status == TC_OK) status = device.put(set2.z, &a.z2);
    a.label2 = nullptr;
    if (status == TC_OK && labelled) status = device.put(set2.label, &a.label2);
  }
  if (status == TC_OK) status = device.put(second.cell_start, &a.cell_start2);
  if (status == TC_OK) status = device.put(item_cell, &a.item_cell);
  if (status == TC_OK) status = device.put(item_begin, &a.item_begin);
  if (status == TC_OK) status = device.put(item_end, &a.item_end);
  a.label_slots = label_slots;
  a.slot1 = nullptr;
  a.item_labels = nullptr;
  if (status == TC_OK && label_slots > 0) status = device.put(label_items.slot, &a.slot1);
  if (status == TC_OK && label_slots > 0)
    status = device.put(label_items.item_labels, &a.item_labels);
  if (status != TC_OK) return status;
  a.nx = grid.nx;
  a.ny = grid.ny;
  a.nz = grid.nz;
  a.reach_x = grid.reach_x;
  a.reach_y = grid.reach_y;
  a.reach_z = grid.reach_z;
  a.lx = grid.lx;
  a.ly = grid.ly;
  a.lz = grid.lz;
  for (int k = 0; k <= n_rp; ++k) a.edge_sqr[k] = rp_bins[k] * rp_bins[k];
  a.n_rp = n_rp;
  a.n_pi = n_pi;
  a.pi_max = pi_max;
  a.inv_dpi = smu ? (double)n_pi : (double)n_pi / pi_max;
  a.n_labels = n_labels;
  void* d_counts = nullptr;
  TC_HIP(hipMalloc(&d_counts, n_counts * sizeof(uint64_t)));
  device.pointers.push_back(d_counts);
  TC_HIP(hipMemset(d_counts, 0, n_counts * sizeof(uint64_t)));
  a.counts = (unsigned long long*)d_counts;

  const dim3 grid_dim((unsigned)item_cell.size()), block(kPairThreads);
  if (labelled) {
    const size_t lds = (size_t)n_rp * n_pi * label_slots * n_labels * sizeof(unsigned);
    if (smu)
      hipLaunchKernelGGL((pair_count_kernel<true, true>), grid_dim, block, lds, nullptr, a);
    else
      hipLaunchKernelGGL((pair_count_kernel<true, false>), grid_dim, block, lds, nullptr, a);
  } else {
    const size_t lds = (size_t)n_rp * n_pi * sizeof(unsigned);
    TC_CHECK(lds <= 48 * 1024, "at most %d two-dimensional bins are supported", 48 * 1024 / 4);
    if (smu)
      hipLaunchKernelGGL((pair_count_kernel<false, true>), grid_dim, block, lds, nullptr, a);
    else
      hipLaunchKernelGGL((pair_count_kernel<false, false>), grid_dim, block, lds, nullptr, a);
  }
  TC_HIP(hipGetLastError());
  TC_HIP(hipMemcpy(counts, d_counts, n_counts * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return TC_OK;
}

}  // namespace
}  // namespace host
}  // namespace tc

extern "C" {

int tc_pair_count_rppi(const double* pos1, int64_t n1, const double* pos2, int64_t n2,
                       const double* boxsize, const double* rp_bins, int n_rp, double pi_max,
                       int n_pi, uint64_t* npairs) {
  return tc::host::pair_count(pos1, nullptr, n1, pos2, nullptr, n2, 0, boxsize, rp_bins, n_rp,
                              pi_max, n_pi, npairs);
}

int tc_pair_count_smu(const double* pos1, int64_t n1, const double* pos2, int64_t n2,
                      const double* boxsize, const double* s_bins, int n_s, int n_mu,
                      uint64_t* npairs) {
  return tc::host::pair_count(pos1, nullptr, n1, pos2, nullptr, n2, 0, boxsize, s_bins, n_s, 1.0,
                              n_mu, npairs, true);
}

int tc_pair_count_rppi_labelled(const double* pos1, const int32_t* label1, int64_t n1,
                                const double* pos2, const int32_t* label2, int64_t n2,
                                int n_labels, const double* boxsize, const double* rp_bins,
                                int n_rp, double pi_max, uint64_t* counts) {
  using tc::host::fail;
  TC_CHECK(n_labels >= 1 && n_labels <= 4096, "between 1 and 4096 labels are supported");
  return tc::host::pair_count(pos1, label1, n1, pos2, label2, n2, n_labels, boxsize, rp_bins,
                              n_rp, pi_max, 1, counts);
}

int tc_pair_count_smu_labelled(const double* pos1, const int32_t* label1, int64_t n1,
                               const double* pos2, const int32_t* label2, int64_t n2,
                               int n_labels, const double* boxsize, const double* s_bins,
                               int n_s, int n_mu, uint64_t* counts) {
  using tc::host::fail;
  TC_CHECK(n_labels >= 1 && n_labels <= 4096, "between 1 and 4096 labels are supported");
  TC_CHECK(n_s >= 1 && n_mu >= 1 &&
               (double)n_s * n_mu * n_labels * n_labels * sizeof(uint64_t) <= 8e9,
           "the (n_s, n_mu, n_labels, n_labels) counters exceed 8 GB");
  return tc::host::pair_count(pos1, label1, n1, pos2, label2, n2, n_labels, boxsize, s_bins, n_s,
                              1.0, n_mu, counts, true);
}

}  // extern "C"
