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
// Layout: the host sorts the points into a grid of cells at least half of r_p,max (x, y) and
// pi_max (z) wide, so that the partners of a point lie in the 5 x 5 x 5 cells around it (or, in
// small boxes, cells a full reach wide and 27 neighbours; a dimension with fewer than three cells
// has one cell and no neighbour offsets: the minimum image does the wrapping).
//   * pair_count_kernel (one pair of samples): a workgroup owns up to 256 consecutive points of
//     one cell -- one per lane -- and streams the points of the neighbouring cells through LDS in
//     tiles of 256 (one global load per point and tile, 256 distance tests per load); (r_p, pi)
//     counters in LDS, flushed once per workgroup.
//   * pair_count_blocks_kernel (all bin pairs): counters for every (r_p bin, label 1, label 2)
//     would be 19 x 100 x 100 words, and with a handful of points per cell and label a
//     workgroup's pairs scatter over all of them -- one 64-bit global atomic per pair, each a
//     fabric transaction (round 2: 2.3e9 atomics = 73 GB of WRITE_SIZE per 10^6 points).  So the
//     labels are cut into blocks (all labels of sample 1 x 4 labels of sample 2 while that fits
//     30 KB), the points are sorted by (cell, label), and a workgroup owns ONE pair of label
//     blocks over a range of cells: for every cell it gathers the block-2 points of the
//     surrounding cells -- 125 short ranges compacted through a prefix sum, one point per lane --
//     and walks the cell's block-1 points (uniform addresses: they arrive through the scalar
//     cache as SGPR operands, no staging, no barrier).  Its private LDS histogram (r bin, label
//     1, label 2 in block) collects hundreds of pairs per counter before ONE flush: ~1e7 global
//     atomics instead of one per pair.  10^6 points, 100 x 100 bin pairs: 642 -> 141 ms, 1.6x
//     the count of the same points without labels.
// Counters are integers: the result does not depend on the order of the atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "internal.h"

// The separations are formed exactly as the oracle (NumPy) forms them: every product and every
// sum rounded on its own.  hipcc contracts a * b + c * d into a fused multiply-add by default
// (the toolchain's __dmul_rn / __dadd_rn are plain operators compiled with contraction
// allowed), which changes r^2 in the last bit and with it the bin of a pair that sits within
// an ulp of a bin edge (tests/test_gpu_paircount.py constructs such pairs).
#pragma clang fp contract(off)

namespace tc {

// (operators of THIS file, i.e. without the contraction flag the toolchain's own __dmul_rn /
// __dadd_rn inline functions carry)
__device__ __forceinline__ double mul_rn(double a, double b) { return a * b; }
__device__ __forceinline__ double add_rn(double a, double b) { return a + b; }

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
  unsigned long long* counts;   // (n_rp, n_pi) or (n_rp [, n_mu], n_labels, n_labels)
  // pair_count_blocks_kernel: label blocks of both sets (hostmath.h: LabelBlocks, PairUnits)
  const int32_t* block_start1;  // (n_cells, n_blocks1 + 1)
  const int32_t* block_start2;  // (n_cells, n_blocks2 + 1)
  int block1, block2;           // labels per block
  int n_blocks1, n_blocks2;
  const int32_t* unit_block1;   // per workgroup: its pair of label blocks ...
  const int32_t* unit_block2;
  const int32_t* unit_cell_begin;   // ... and its range of cells
  const int32_t* unit_cell_end;
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
// Candidates that pass the distance cuts are COMPACTED before they are binned: with one point
// i per lane and the same point j for all lanes, ~15 % of the lanes pass a test, but a wave
// pays for the r_p bin search and the histogram update whenever any lane does.  So the passing
// lanes append (r^2, |dz|) to a queue of their wave in LDS (ballot + mbcnt), and whenever 64
// entries are there the wave bins 64 real pairs at once: the search and the atomics run at full
// lane occupancy, ~7 tests per binning pass.  Neighbour cells that do not wrap around the box
// skip the minimum image (uniform per cell pair; with at least seven cells per dimension the
// separations inside the 5 x 5 x 5 neighbourhood stay below half the box, so the oracle's
// conditional shifts never fire there).  Counters are integers: the order does not matter.
// (Point j through scalar loads instead of the LDS tiles -- the labelled kernel's way for its
// block-1 points -- is slower here: 122 against 101 ms for 10^6 points.)
template <bool SMU>
__global__ __launch_bounds__(kPairThreads) void pair_count_kernel(PairArgs a) {
  __shared__ double sx[kPairThreads], sy[kPairThreads], sz[kPairThreads];
  __shared__ double queue_r[kPairThreads / 64][128], queue_z[kPairThreads / 64][128];
  extern __shared__ unsigned hist[];   // n_rp * n_pi counters
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  double* qr = queue_r[tid >> 6];
  double* qz = queue_z[tid >> 6];
  const int n_hist = a.n_rp * a.n_pi;
  for (int k = tid; k < n_hist; k += kPairThreads) hist[k] = 0u;

  const int cell = a.item_cell[blockIdx.x];
  const int begin = a.item_begin[blockIdx.x], end = a.item_end[blockIdx.x];
  const int i = begin + tid;
  const bool active = i < end;
  const double xi = active ? a.x1[i] : 0.0, yi = active ? a.y1[i] : 0.0;
  const double zi = active ? a.z1[i] : 0.0;
  const int cz = cell % a.nz, cy = (cell / a.nz) % a.ny, cx = cell / (a.nz * a.ny);
  const double hx = 0.5 * a.lx, hy = 0.5 * a.ly, hz = 0.5 * a.lz;
  const double lo_sqr = a.edge_sqr[0], hi_sqr = a.edge_sqr[a.n_rp];
  __syncthreads();

  // bins the first n entries of the wave's queue (n <= 64)
  auto bin_entries = [&](int n) {
    if (lane < n) {
      const double r_sqr = qr[lane], dz = qz[lane];
      int bin = 0;
      for (int k = 1; k < a.n_rp; ++k) bin += r_sqr >= a.edge_sqr[k] ? 1 : 0;
      if (SMU) {
        const double mu = r_sqr > 0.0 ? __ddiv_rn(dz, __dsqrt_rn(r_sqr)) : 0.0;
        const int mu_bin = (int)(mu * a.inv_dpi);
        if (mu < 1.0 && mu_bin < a.n_pi) atomicAdd(&hist[bin * a.n_pi + mu_bin], 1u);
      } else {
        const int pi_bin = (int)(dz * a.inv_dpi);
        if (pi_bin < a.n_pi) atomicAdd(&hist[bin * a.n_pi + pi_bin], 1u);
      }
    }
  };
  int queued = 0;      // entries in the wave's queue (uniform over the wave)

  for (int ox = -a.reach_x; ox <= a.reach_x; ++ox)
    for (int oy = -a.reach_y; oy <= a.reach_y; ++oy)
      for (int oz = -a.reach_z; oz <= a.reach_z; ++oz) {
        const int nxc = (cx + ox + a.nx) % a.nx, nyc = (cy + oy + a.ny) % a.ny;
        const int nzc = (cz + oz + a.nz) % a.nz;
        const int other = (nxc * a.ny + nyc) * a.nz + nzc;
        const bool wrap_x = a.nx < 7 || cx + ox < 0 || cx + ox >= a.nx;
        const bool wrap_y = a.ny < 7 || cy + oy < 0 || cy + oy >= a.ny;
        const bool wrap_z = a.nz < 7 || cz + oz < 0 || cz + oz >= a.nz;
        const int j_begin = a.cell_start2[other], j_end = a.cell_start2[other + 1];
        auto test = [&](double xj, double yj, double zj, bool in_range) {
          double dz = zi - zj;
          if (wrap_z) dz = min_image(dz, a.lz, hz);
          dz = fabs(dz);
          bool pass = active && in_range && dz < a.pi_max;
          if (__builtin_amdgcn_ballot_w64(pass) == 0) return;
          double dx = xi - xj, dy = yi - yj;
          if (wrap_x) dx = min_image(dx, a.lx, hx);
          if (wrap_y) dy = min_image(dy, a.ly, hy);
          // (separately rounded products and sums: the oracle's arithmetic)
          double r_sqr = add_rn(mul_rn(dx, dx), mul_rn(dy, dy));
          if (SMU) r_sqr = add_rn(r_sqr, mul_rn(dz, dz));
          pass = pass && r_sqr >= lo_sqr && r_sqr < hi_sqr;
          const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
          if (mask == 0) return;
          if (pass) {
            const int slot =
                queued + (int)__builtin_amdgcn_mbcnt_hi(
                             (unsigned)(mask >> 32),
                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            qr[slot] = r_sqr;
            qz[slot] = dz;
          }
          queued += __builtin_popcountll(mask);
          if (queued >= 64) {
            __builtin_amdgcn_wave_barrier();
            bin_entries(64);
            // the rest (fewer than 64 entries) moves to the front
            queued -= 64;
            double rest_r = 0.0, rest_z = 0.0;
            if (lane < queued) {
              rest_r = qr[64 + lane];
              rest_z = qz[64 + lane];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < queued) {
              qr[lane] = rest_r;
              qz[lane] = rest_z;
            }
            __builtin_amdgcn_wave_barrier();
          }
        };
        for (int j0 = j_begin; j0 < j_end; j0 += kPairThreads) {
          const int n_tile = j_end - j0 < kPairThreads ? j_end - j0 : kPairThreads;
          __syncthreads();
          if (tid < n_tile) {
            sx[tid] = a.x2[j0 + tid];
            sy[tid] = a.y2[j0 + tid];
            sz[tid] = a.z2[j0 + tid];
          }
          __syncthreads();
          for (int t = 0; t < n_tile; ++t) test(sx[t], sy[t], sz[t], true);
        }
      }
  __builtin_amdgcn_wave_barrier();
  bin_entries(queued);
  __syncthreads();
  for (int k = tid; k < n_hist; k += kPairThreads) {
    const unsigned value = hist[k];
    if (value != 0u) atomicAdd(a.counts + k, (unsigned long long)value);
  }
}

// All pairs of labels in one pass (see the top of this file).  Workgroup = (label block of set
// 1, label block of set 2, range of cells); LDS: (n_rp [* n_mu], block1, block2) counters.

// (x1 ... label1 = a.x1 ... a.label1 once more as restrict-qualified kernel arguments: nothing
// the kernel writes aliases them, which lets the compiler fetch the block-1 point of an
// iteration with scalar loads)
template <bool SMU, int UNROLL>
__global__ __launch_bounds__(kPairThreads) void pair_count_blocks_kernel(
    PairArgs a, const double* __restrict__ x1, const double* __restrict__ y1,
    const double* __restrict__ z1, const int32_t* __restrict__ label1) {
  extern __shared__ unsigned hist[];
  __shared__ int32_t nb_start[128];
  __shared__ int32_t nb_prefix[129];     // [128]: all block-2 points around the cell
  const int tid = threadIdx.x;
  const int n_bin = a.n_rp * a.n_pi;             // (labelled r_p counts: n_pi == 1)
  const int n_hist = n_bin * a.block1 * a.block2;
  const int ba = a.unit_block1[blockIdx.x], bb = a.unit_block2[blockIdx.x];
  const int c_begin = a.unit_cell_begin[blockIdx.x], c_end = a.unit_cell_end[blockIdx.x];
  const int label_lo1 = ba * a.block1, label_lo2 = bb * a.block2;
  const int stride1 = a.n_blocks1 + 1, stride2 = a.n_blocks2 + 1;
  const int wx = 2 * a.reach_x + 1, wy = 2 * a.reach_y + 1, wz = 2 * a.reach_z + 1;
  const int n_nb = wx * wy * wz;
  const double hx = 0.5 * a.lx, hy = 0.5 * a.ly, hz = 0.5 * a.lz;
  const double lo_sqr = a.edge_sqr[0], hi_sqr = a.edge_sqr[a.n_rp];

  // passing pairs wait in a queue of their wave and are binned 64 at a time (see
  // pair_count_kernel): (r^2 [, |dz|], (label 1 in block) * block2 + label 2 in block)
  __shared__ double queue_r[kPairThreads / 64][128];
  __shared__ double queue_z[SMU ? kPairThreads / 64 : 1][128];
  __shared__ int32_t queue_c[kPairThreads / 64][128];
  const int lane = tid & 63;
  double* qr = queue_r[tid >> 6];
  double* qz = queue_z[SMU ? tid >> 6 : 0];
  int32_t* qc = queue_c[tid >> 6];
  int queued = 0;      // (uniform over the wave)
  auto bin_entries = [&](int n) {
    if (lane < n) {
      const double r_sqr = qr[lane];
      int bin = 0;
      for (int e = 1; e < a.n_rp; ++e) bin += r_sqr >= a.edge_sqr[e] ? 1 : 0;
      bool counts = true;
      if (SMU) {
        const double dz = qz[lane];
        const double mu = r_sqr > 0.0 ? __ddiv_rn(dz, __dsqrt_rn(r_sqr)) : 0.0;
        const int mu_bin = (int)(mu * a.inv_dpi);
        counts = mu < 1.0 && mu_bin < a.n_pi;
        bin = bin * a.n_pi + mu_bin;
      }
      if (counts) atomicAdd(&hist[bin * a.block1 * a.block2 + qc[lane]], 1u);
    }
  };
  auto flush = [&]() {
    __syncthreads();
    for (int k = tid; k < n_hist; k += kPairThreads) {
      const unsigned value = hist[k];
      hist[k] = 0u;
      if (value == 0u) continue;
      const int lj = k % a.block2, li = (k / a.block2) % a.block1, bin = k / (a.block1 * a.block2);
      atomicAdd(a.counts + ((size_t)bin * a.n_labels + label_lo1 + li) * a.n_labels +
                    label_lo2 + lj,
                (unsigned long long)value);
    }
    __syncthreads();
  };
  for (int k = tid; k < n_hist; k += kPairThreads) hist[k] = 0u;
  // candidate pairs since the last flush: the 32-bit counters cannot overflow below 2^32
  unsigned long long pending = 0;

  for (int cell = c_begin; cell < c_end; ++cell) {
    const int p_begin = a.block_start1[cell * stride1 + ba];
    const int p_end = a.block_start1[cell * stride1 + ba + 1];
    if (p_begin == p_end) continue;              // (uniform over the workgroup)
    const int cz = cell % a.nz, cy = (cell / a.nz) % a.ny, cx = cell / (a.nz * a.ny);
    __syncthreads();                             // the previous cell's ranges are no longer read
    // ranges of the block-2 points in the cells around this one, compacted by a prefix sum
    // (wave 0: two ranges per lane, inclusive scan across the lanes)
    if (tid < 64) {
      int len[2], start[2];
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * tid + h;
        len[h] = 0;
        start[h] = 0;
        if (k < n_nb) {
          const int oz = k % wz - a.reach_z, oy = (k / wz) % wy - a.reach_y;
          const int ox = k / (wz * wy) - a.reach_x;
          const int other = (((cx + ox + a.nx) % a.nx) * a.ny + (cy + oy + a.ny) % a.ny) * a.nz +
                            (cz + oz + a.nz) % a.nz;
          start[h] = a.block_start2[other * stride2 + bb];
          len[h] = a.block_start2[other * stride2 + bb + 1] - start[h];
        }
      }
      int sum = len[0] + len[1];
      for (int shift = 1; shift < 64; shift <<= 1) {
        const int up = __shfl_up(sum, shift, 64);
        if (tid >= shift) sum += up;
      }
      // nb_prefix[k] = points before range k; nb_prefix[n_nb ...] = total
      const int before = sum - len[0] - len[1];
      nb_prefix[2 * tid] = before;
      nb_prefix[2 * tid + 1] = before + len[0];
      nb_start[2 * tid] = start[0];
      nb_start[2 * tid + 1] = start[1];
      if (tid == 63) nb_prefix[128] = sum;
    }
    __syncthreads();
    const int total = nb_prefix[128];
    pending += (unsigned long long)(p_end - p_begin) * (unsigned long long)total;
    if (pending >= (1ull << 31)) {
      flush();
      pending = (unsigned long long)(p_end - p_begin) * (unsigned long long)total;
    }
    for (int g0 = 0; g0 < total; g0 += kPairThreads) {
      // (no lane leaves the loop early: the wave's queue is drained by all of its lanes)
      const bool valid = g0 + tid < total;
      const int g = valid ? g0 + tid : total - 1;
      // range of point g: the last k with nb_prefix[k] <= g (ranges of length 0 are skipped
      // because their successor has the same prefix)
      int k = 0;
      for (int step = 64; step >= 1; step >>= 1)
        if (k + step < 128 && nb_prefix[k + step] <= g) k += step;
      const int j = nb_start[k] + (g - nb_prefix[k]);
      const double xj = a.x2[j], yj = a.y2[j], zj = a.z2[j];
      const int lj = a.label2[j] - label_lo2;
      // does any lane's neighbour cell wrap around the box?  (uniform over the wave; without a
      // wrap the minimum image is the identity: see pair_count_kernel)
      const int oz = k % wz - a.reach_z, oy = (k / wz) % wy - a.reach_y;
      const int ox = k / (wz * wy) - a.reach_x;
      const bool wrap_x =
          __builtin_amdgcn_ballot_w64(a.nx < 7 || cx + ox < 0 || cx + ox >= a.nx) != 0;
      const bool wrap_y =
          __builtin_amdgcn_ballot_w64(a.ny < 7 || cy + oy < 0 || cy + oy >= a.ny) != 0;
      const bool wrap_z =
          __builtin_amdgcn_ballot_w64(a.nz < 7 || cz + oz < 0 || cz + oz >= a.nz) != 0;
      // block-1 points several at a time: p is uniform over the wave, so the coordinates come
      // by scalar loads, issued together ahead of the tests (staged through LDS in tiles, as
      // pair_count_kernel does with its points j: 169 against 157 ms for 10^6 points); passing
      // pairs are queued and binned 64 at a time
      auto test = [&](double xi, double yi, double zi, int li, bool in_range) {
        double dz = zi - zj;
        if (wrap_z) dz = min_image(dz, a.lz, hz);
        dz = fabs(dz);
        bool pass = valid && in_range && dz < a.pi_max;
        if (__builtin_amdgcn_ballot_w64(pass) == 0) return;
        double dx = xi - xj, dy = yi - yj;
        if (wrap_x) dx = min_image(dx, a.lx, hx);
        if (wrap_y) dy = min_image(dy, a.ly, hy);
        double r_sqr = add_rn(mul_rn(dx, dx), mul_rn(dy, dy));
        if (SMU) r_sqr = add_rn(r_sqr, mul_rn(dz, dz));
        pass = pass && r_sqr >= lo_sqr && r_sqr < hi_sqr;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask == 0) return;
        if (pass) {
          const int slot = queued + (int)__builtin_amdgcn_mbcnt_hi(
                                        (unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          qr[slot] = r_sqr;
          if (SMU) qz[slot] = dz;
          qc[slot] = (li - label_lo1) * a.block2 + lj;
        }
        queued += __builtin_popcountll(mask);
        if (queued >= 64) {
          __builtin_amdgcn_wave_barrier();
          bin_entries(64);
          queued -= 64;
          double rest_r = 0.0, rest_z = 0.0;
          int rest_c = 0;
          if (lane < queued) {
            rest_r = qr[64 + lane];
            if (SMU) rest_z = qz[64 + lane];
            rest_c = qc[64 + lane];
          }
          __builtin_amdgcn_wave_barrier();
          if (lane < queued) {
            qr[lane] = rest_r;
            if (SMU) qz[lane] = rest_z;
            qc[lane] = rest_c;
          }
          __builtin_amdgcn_wave_barrier();
        }
      };
      for (int p = p_begin; p < p_end; p += UNROLL) {
        double xs[UNROLL], ys[UNROLL], zs[UNROLL];
        int ls[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int q = p + u < p_end ? p + u : p_end - 1;
          xs[u] = x1[q];
          ys[u] = y1[q];
          zs[u] = z1[q];
          ls[u] = label1[q];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) test(xs[u], ys[u], zs[u], ls[u], p + u < p_end);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  bin_entries(queued);
  flush();
}

// Mass in cylinders around every object (the per-object weighted pair count behind an excess
// surface density, halotools' mean_delta_sigma: tabcorr/tabcorr.py:846-922 calls it with
// tpcf_args = (particle positions, particle masses, rp_bins), scripts/tabulate_snapshot.py:
// 228-237): for object g and every annulus d between consecutive edges the summed mass of the
// particles whose projected separation r = sqrt(dx^2 + dy^2) (minimum image in x and y; the
// line of sight spans the whole box) satisfies edge[d - 1] < r <= edge[d] (annulus 0: r <=
// edge[0]), compared squared.  One lane per object, the particles of the surrounding cell
// columns streamed through LDS in tiles of 256 in cell order, every lane adding into its own
// LDS column of annuli -- the order of the additions is fixed by the cell sort, so the sums are
// reproducible (and exact for equal masses: they are counts).  The host accumulates the annuli
// into cylinders.
struct CylinderArgs {
  const double* x1;             // objects, sorted by cell column
  const double* y1;
  const double* x2;             // particles, sorted by cell column
  const double* y2;
  const double* mass2;          // NULL: unit masses
  const int32_t* cell_start2;
  const int32_t* item_cell;
  const int32_t* item_begin;
  const int32_t* item_end;
  int nx, ny;
  int reach_x, reach_y;
  double lx, ly;
  double edge_sqr[kMaxRpBins + 1];
  int n_edges;
  double* annuli;               // (n_objects, n_edges), objects in sorted order
};

__global__ __launch_bounds__(kPairThreads) void mass_in_cylinders_kernel(CylinderArgs a) {
  __shared__ double sx[kPairThreads], sy[kPairThreads], sm[kPairThreads];
  extern __shared__ double column[];   // (n_edges, 256): annulus sums of every lane
  const int tid = threadIdx.x;
  for (int d = 0; d < a.n_edges; ++d) column[d * kPairThreads + tid] = 0.0;
  const int cell = a.item_cell[blockIdx.x];
  const int begin = a.item_begin[blockIdx.x], end = a.item_end[blockIdx.x];
  const int i = begin + tid;
  const bool active = i < end;
  const double xi = active ? a.x1[i] : 0.0, yi = active ? a.y1[i] : 0.0;
  const int cy = cell % a.ny, cx = cell / a.ny;
  const double hx = 0.5 * a.lx, hy = 0.5 * a.ly;
  const double hi_sqr = a.edge_sqr[a.n_edges - 1];
  for (int ox = -a.reach_x; ox <= a.reach_x; ++ox)
    for (int oy = -a.reach_y; oy <= a.reach_y; ++oy) {
      const int other = ((cx + ox + a.nx) % a.nx) * a.ny + (cy + oy + a.ny) % a.ny;
      const int j_begin = a.cell_start2[other], j_end = a.cell_start2[other + 1];
      for (int j0 = j_begin; j0 < j_end; j0 += kPairThreads) {
        const int n_tile = j_end - j0 < kPairThreads ? j_end - j0 : kPairThreads;
        __syncthreads();
        if (tid < n_tile) {
          sx[tid] = a.x2[j0 + tid];
          sy[tid] = a.y2[j0 + tid];
          sm[tid] = a.mass2 != nullptr ? a.mass2[j0 + tid] : 1.0;
        }
        __syncthreads();
        if (!active) continue;
        for (int t = 0; t < n_tile; ++t) {
          const double dx = min_image(xi - sx[t], a.lx, hx);
          const double dy = min_image(yi - sy[t], a.ly, hy);
          const double r_sqr = add_rn(mul_rn(dx, dx), mul_rn(dy, dy));
          if (!(r_sqr <= hi_sqr)) continue;
          int d = 0;
          for (int k = 0; k + 1 < a.n_edges; ++k) d += r_sqr > a.edge_sqr[k] ? 1 : 0;
          column[d * kPairThreads + tid] += sm[t];
        }
      }
    }
  if (active)
    for (int d = 0; d < a.n_edges; ++d)
      a.annuli[(size_t)i * a.n_edges + d] = column[d * kPairThreads + tid];
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
  // Cells half the reach wide with 5 x 5 x 5 neighbours where the box allows (an eighth of
  // the cell volume: 42 % fewer candidate pairs than 27 full-size cells).
  CellGrid grid = make_cell_grid(boxsize, rp_max, pi_max, std::max(n1, n2), true);

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
  const CellSort& second = autocorr ? set1 : set2;
  // work items of the unlabelled kernel: up to 256 consecutive set-1 points of one cell;
  // labelled: label blocks and (block, block, cell range) units of the blocks kernel
  std::vector<int32_t> item_cell, item_begin, item_end;
  LabelBlocks blocks1, blocks2;
  PairUnits units;
  const int n_bin = n_rp * n_pi;
  int lds_counters = n_bin;
  if (labelled) {
    // LDS counters (bins, labels of block 1, labels of block 2): about 30 KB per workgroup
    // keep five workgroups on a CU, which the scalar-load latency of the inner loop needs
    // (19 bins x 100 labels, 10^6 points: 8 labels of set 2 = 60 KB 276 ms, 4 = 30 KB 180 ms,
    // 2 = 15 KB 190 ms: tools/archive/r03_pc_knobs.sh); set 1 keeps all its labels while that fits
    const int limit = 60 * 1024 / (int)sizeof(unsigned);
    const int budget = env_int("TC_PAIR_LDS_KB", 30) * 1024 / (int)sizeof(unsigned);
    TC_CHECK(n_bin <= limit, "at most %d (separation, mu) bins are supported", limit);
    int b1 = n_labels, b2 = std::min(n_labels, 8);
    while (n_bin * b1 * b2 > budget && b2 > 2) --b2;
    while (n_bin * b1 * b2 > budget && b1 > 1) b1 = (b1 + 1) / 2;
    while (n_bin * b1 * b2 > limit && b2 > 1) --b2;
    sort_cells_by_label(set1);
    if (!autocorr) sort_cells_by_label(set2);
    build_label_blocks(set1, n_labels, b1, blocks1);
    build_label_blocks(second, n_labels, b2, blocks2);
    // ~16 units per CU: the cell ranges differ in weight by the clustering of the points
    build_pair_units(grid, set1, second, blocks1.n_blocks, blocks2.n_blocks,
                     env_int("TC_PAIR_UNITS", 4096), units);
    lds_counters = n_bin * b1 * b2;
    // (the workgroup's 32-bit LDS counters are flushed between cells, not inside one)
    if (units.max_cell_candidates >= 4.0e9)
      return fail(TC_ERR_UNSUPPORTED, "a cell and its surroundings hold %.3g candidate pairs; "
                  "more than the labelled pair counter's 32-bit counters take",
                  units.max_cell_candidates);
  } else {
    // heaviest items first: the candidates of an item are its points x the points of the
    // cells around its cell, and clustered points make that vary by orders of magnitude -- in
    // cell order the launch ended with a few workgroups of dense cells on an otherwise idle
    // chip (1.4 resident waves per SIMD on average)
    std::vector<int64_t> around((size_t)grid.n_cells(), 0);
    for (int cx = 0; cx < grid.nx; ++cx)
      for (int cy = 0; cy < grid.ny; ++cy)
        for (int cz = 0; cz < grid.nz; ++cz) {
          int64_t sum = 0;
          for (int ox = -grid.reach_x; ox <= grid.reach_x; ++ox)
            for (int oy = -grid.reach_y; oy <= grid.reach_y; ++oy)
              for (int oz = -grid.reach_z; oz <= grid.reach_z; ++oz) {
                const int other = (((cx + ox + grid.nx) % grid.nx) * grid.ny +
                                   (cy + oy + grid.ny) % grid.ny) * grid.nz +
                                  (cz + oz + grid.nz) % grid.nz;
                sum += second.cell_start[other + 1] - second.cell_start[other];
              }
          around[(size_t)(cx * grid.ny + cy) * grid.nz + cz] = sum;
        }
    struct Item { int32_t cell, begin, end; int64_t cost; };
    std::vector<Item> items;
    for (int c = 0; c < grid.n_cells(); ++c)
      for (int32_t b = set1.cell_start[c]; b < set1.cell_start[c + 1]; b += kPairThreads) {
        const int32_t e = std::min<int32_t>(b + kPairThreads, set1.cell_start[c + 1]);
        // (a wave tests 64 points at once: a partial item costs what its waves cost)
        items.push_back({c, b, e, (int64_t)((e - b + 63) / 64) * around[c]});
      }
    std::stable_sort(items.begin(), items.end(),
                     [](const Item& u, const Item& v) { return u.cost > v.cost; });
    for (const Item& item : items) {
      item_cell.push_back(item.cell);
      item_begin.push_back(item.begin);
      item_end.push_back(item.end);
    }
  }

  DeviceArrays device;
  PairArgs a{};
  int status = device.put(set1.x, &a.x1);
  if (status == TC_OK) status = device.put(set1.y, &a.y1);
  if (status == TC_OK) status = device.put(set1.z, &a.z1);
  a.label1 = nullptr;
  if (status == TC_OK && labelled) status = device.put(set1.label, &a.label1);
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
  if (labelled) {
    if (status == TC_OK) status = device.put(blocks1.start, &a.block_start1);
    if (status == TC_OK) status = device.put(blocks2.start, &a.block_start2);
    if (status == TC_OK) status = device.put(units.block1, &a.unit_block1);
    if (status == TC_OK) status = device.put(units.block2, &a.unit_block2);
    if (status == TC_OK) status = device.put(units.cell_begin, &a.unit_cell_begin);
    if (status == TC_OK) status = device.put(units.cell_end, &a.unit_cell_end);
    a.block1 = blocks1.block;
    a.block2 = blocks2.block;
    a.n_blocks1 = blocks1.n_blocks;
    a.n_blocks2 = blocks2.n_blocks;
  } else {
    if (status == TC_OK) status = device.put(item_cell, &a.item_cell);
    if (status == TC_OK) status = device.put(item_begin, &a.item_begin);
    if (status == TC_OK) status = device.put(item_end, &a.item_end);
  }
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

  const dim3 block(kPairThreads);
  const size_t lds = (size_t)lds_counters * sizeof(unsigned);
  if (labelled) {
    const dim3 grid_dim((unsigned)units.block1.size());
    if (grid_dim.x == 0) return TC_OK;
    const int unroll = env_int("TC_PAIR_UNROLL", 8);                  // (developer builds)
#define TC_LAUNCH(SMU, U)                                                                     \
  hipLaunchKernelGGL((pair_count_blocks_kernel<SMU, U>), grid_dim, block, lds, nullptr, a,  \
                     a.x1, a.y1, a.z1, a.label1)
    if (smu) {
      if (unroll == 4) TC_LAUNCH(true, 4); else if (unroll == 2) TC_LAUNCH(true, 2); else TC_LAUNCH(true, 8);
    } else {
      if (unroll == 4) TC_LAUNCH(false, 4); else if (unroll == 2) TC_LAUNCH(false, 2); else TC_LAUNCH(false, 8);
    }
#undef TC_LAUNCH
  } else {
    const dim3 grid_dim((unsigned)item_cell.size());
    TC_CHECK(lds <= 48 * 1024, "at most %d two-dimensional bins are supported", 48 * 1024 / 4);
    if (smu)
      hipLaunchKernelGGL(pair_count_kernel<true>, grid_dim, block, lds, nullptr, a);
    else
      hipLaunchKernelGGL(pair_count_kernel<false>, grid_dim, block, lds, nullptr, a);
  }
  TC_HIP(hipGetLastError());
  TC_HIP(hipMemcpy(counts, d_counts, n_counts * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return TC_OK;
}

// tc_mass_in_cylinders: cell columns in (x, y) at least half the largest radius wide, objects
// and particles sorted into them (the original index rides along as the label), one launch,
// annuli accumulated into cylinders in the caller's object order.
int mass_in_cylinders(const double* objects, int64_t n_obj, const double* particles,
                      int64_t n_ptcl, const double* masses, const double* boxsize,
                      const double* rp_bins, int n_edges, double* mass) {
  TC_CHECK(boxsize != nullptr && rp_bins != nullptr, "NULL argument");
  TC_CHECK(n_obj >= 0 && n_ptcl >= 0 && n_obj < (1LL << 31) && n_ptcl < (1LL << 31),
           "invalid point count");
  TC_CHECK(n_edges >= 1 && n_edges <= kMaxRpBins + 1, "between 1 and %d radii are supported",
           kMaxRpBins + 1);
  for (int k = 0; k < n_edges; ++k)
    TC_CHECK(rp_bins[k] >= 0.0 && (k == 0 || rp_bins[k] > rp_bins[k - 1]),
             "rp_bins must be non-negative and increasing");
  TC_CHECK(boxsize[0] > 0 && boxsize[1] > 0 && boxsize[2] > 0, "box size must be positive");
  const double rp_max = rp_bins[n_edges - 1];
  TC_CHECK(rp_max < 0.5 * std::min(boxsize[0], boxsize[1]),
           "the largest radius must be smaller than half the box size");
  if (n_obj == 0) return TC_OK;
  TC_CHECK(objects != nullptr && mass != nullptr && (n_ptcl == 0 || particles != nullptr),
           "NULL argument");
  std::fill(mass, mass + (size_t)n_obj * n_edges, 0.0);
  if (n_ptcl == 0) return TC_OK;
  // one cell along z: its reach is the box
  const CellGrid grid = make_cell_grid(boxsize, rp_max, boxsize[2], std::max(n_obj, n_ptcl), true);
  Range range("mass in cylinders");
  std::vector<int32_t> index1((size_t)n_obj), index2((size_t)n_ptcl);
  std::iota(index1.begin(), index1.end(), 0);
  std::iota(index2.begin(), index2.end(), 0);
  CellSort set1, set2;
  int64_t outside = sort_into_cells(grid, objects, index1.data(), n_obj, set1);
  TC_CHECK(outside < 0, "object %lld lies outside of the periodic box", (long long)outside);
  outside = sort_into_cells(grid, particles, index2.data(), n_ptcl, set2);
  TC_CHECK(outside < 0, "particle %lld lies outside of the periodic box", (long long)outside);
  std::vector<double> mass_sorted;
  if (masses != nullptr) {
    mass_sorted.resize((size_t)n_ptcl);
    for (int64_t s = 0; s < n_ptcl; ++s) mass_sorted[s] = masses[set2.label[s]];
  }
  std::vector<int32_t> item_cell, item_begin, item_end;
  for (int c = 0; c < grid.n_cells(); ++c)
    for (int32_t b = set1.cell_start[c]; b < set1.cell_start[c + 1]; b += kPairThreads) {
      item_cell.push_back(c);
      item_begin.push_back(b);
      item_end.push_back(std::min<int32_t>(b + kPairThreads, set1.cell_start[c + 1]));
    }
  DeviceArrays device;
  CylinderArgs a{};
  int status = device.put(set1.x, &a.x1);
  if (status == TC_OK) status = device.put(set1.y, &a.y1);
  if (status == TC_OK) status = device.put(set2.x, &a.x2);
  if (status == TC_OK) status = device.put(set2.y, &a.y2);
  a.mass2 = nullptr;
  if (status == TC_OK && masses != nullptr) status = device.put(mass_sorted, &a.mass2);
  if (status == TC_OK) status = device.put(set2.cell_start, &a.cell_start2);
  if (status == TC_OK) status = device.put(item_cell, &a.item_cell);
  if (status == TC_OK) status = device.put(item_begin, &a.item_begin);
  if (status == TC_OK) status = device.put(item_end, &a.item_end);
  if (status != TC_OK) return status;
  a.nx = grid.nx;
  a.ny = grid.ny;
  a.reach_x = grid.reach_x;
  a.reach_y = grid.reach_y;
  a.lx = grid.lx;
  a.ly = grid.ly;
  for (int k = 0; k < n_edges; ++k) a.edge_sqr[k] = rp_bins[k] * rp_bins[k];
  a.n_edges = n_edges;
  void* d_annuli = nullptr;
  const size_t out_bytes = (size_t)n_obj * n_edges * sizeof(double);
  TC_HIP(hipMalloc(&d_annuli, out_bytes));
  device.pointers.push_back(d_annuli);
  a.annuli = (double*)d_annuli;
  const size_t lds = (size_t)n_edges * kPairThreads * sizeof(double);
  if (lds > 48 * 1024)
    TC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mass_in_cylinders_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mass_in_cylinders_kernel, dim3((unsigned)item_cell.size()),
                     dim3(kPairThreads), lds, nullptr, a);
  TC_HIP(hipGetLastError());
  std::vector<double> annuli((size_t)n_obj * n_edges);
  TC_HIP(hipMemcpy(annuli.data(), d_annuli, out_bytes, hipMemcpyDeviceToHost));
  for (int64_t s = 0; s < n_obj; ++s) {
    double* row = mass + (size_t)set1.label[s] * n_edges;
    double running = 0.0;
    for (int k = 0; k < n_edges; ++k) {
      running += annuli[(size_t)s * n_edges + k];
      row[k] = running;
    }
  }
  return TC_OK;
}

}  // namespace
}  // namespace host
}  // namespace tc

extern "C" {

int tc_mass_in_cylinders(const double* objects, int64_t n_objects, const double* particles,
                         int64_t n_particles, const double* masses, const double* boxsize,
                         const double* rp_bins, int n_edges, double* mass) {
  return tc::host::mass_in_cylinders(objects, n_objects, particles, n_particles, masses, boxsize,
                                     rp_bins, n_edges, mass);
}

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
