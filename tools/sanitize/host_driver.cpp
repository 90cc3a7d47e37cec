// Host-only exerciser of the library's planning / layout code for the sanitizer build
// (SURVEY.md section 5): g++ -fsanitize=address,undefined on hostmath.cpp + this file, no
// HIP.  Walks the same entry points the C ABI uses (bin permutation and segmentation,
// chunking, the quadratic-form layout / schedule / table fill / emulation, spline
// matrices, quadrature nodes, the fast-math tables, the pair counter's cell sort) over a
// sweep of shapes and checks the results against direct evaluations.  Exit status 0 = all
// checks passed and no sanitizer report (-fno-sanitize-recover aborts on the first).
//
//   make -C tools/sanitize && tools/sanitize/host_driver
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <numeric>
#include <vector>

#include "../../tabcorr_amd/csrc/fastmath.h"
#include "../../tabcorr_amd/csrc/hostmath.h"
#include "../../tabcorr_amd/csrc/kernel_args.h"

static int g_failures = 0;
#define EXPECT(condition, ...)                 \
  do {                                         \
    if (!(condition)) {                        \
      ++g_failures;                            \
      printf("FAILED %s:%d: ", __FILE__, __LINE__); \
      printf(__VA_ARGS__);                     \
      printf("\n");                            \
    }                                          \
  } while (0)

static void check_quadrature() {
  for (int n = 1; n <= 64; ++n) {
    std::vector<double> x, w;
    tc::gauss_legendre(n, x, w);
    double sum = 0.0, moment = 0.0;
    for (int k = 0; k < n; ++k) {
      sum += w[k];
      moment += w[k] * x[k];
      EXPECT(x[k] > 0.0 && x[k] < 1.0, "node outside (0, 1) for n = %d", n);
    }
    // numpy's leggauss weights (sum 2) with the nodes mapped to (0, 1): tabcorr.py:543-546
    EXPECT(std::fabs(sum - 2.0) < 1e-13, "weights of n = %d sum to %.17g", n, sum);
    EXPECT(std::fabs(moment - 1.0) < 1e-13, "first moment of n = %d", n);
  }
}

static void check_splines() {
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> uniform(0.1, 1.0);
  for (int n = 4; n <= 14; ++n) {
    std::vector<double> xp(n), y(n), a;
    xp[0] = -1.0;
    for (int i = 1; i < n; ++i) xp[i] = xp[i - 1] + uniform(rng);
    for (double& v : y) v = uniform(rng);
    EXPECT(tc::spline_interpolation_matrix(n, xp.data(), a), "singular spline system n = %d", n);
    EXPECT((int)a.size() == (n - 1) * 4 * n, "spline matrix size n = %d", n);
    // the spline passes through the nodes: segment s at xp[s] and xp[s + 1]
    for (int s = 0; s + 1 < n; ++s)
      for (int end = 0; end < 2; ++end) {
        const double x = xp[s + end];
        double value = 0.0;
        for (int p = 0; p < 4; ++p)
          for (int j = 0; j < n; ++j)
            value += a[((size_t)s * 4 + p) * n + j] * y[j] * std::pow(x, p);
        EXPECT(std::fabs(value - y[s + end]) < 1e-8, "spline n = %d segment %d: %g vs %g", n, s,
               value, y[s + end]);
      }
  }
  std::vector<double> a, three = {0.0, 1.0, 2.0};
  (void)tc::spline_interpolation_matrix(3, three.data(), a);   // too few nodes: must not crash
}

static void check_plans() {
  std::mt19937_64 rng(2);
  for (int mode = 0; mode < 2; ++mode)
    for (int n_bins : {1, 2, 3, 7, 16, 60, 100, 137, 200, 440}) {
      std::vector<uint8_t> central(n_bins);
      for (int g = 0; g < n_bins; ++g) central[g] = (rng() % 3) != 0;
      if (n_bins > 5) std::shuffle(central.begin(), central.end(), rng);
      for (int block : {tc::kF64Block, tc::kF32Block})
        for (int budget : {56, 128}) {
          tc::Plan plan;
          tc::build_plan(mode, n_bins, central.data(), block, budget, plan);
          const int64_t expect = mode == 0 ? (int64_t)n_bins * (n_bins + 1) / 2 : n_bins;
          EXPECT(plan.n_entries == expect, "plan entries %lld vs %lld", (long long)plan.n_entries,
                 (long long)expect);
          std::vector<int> seen((size_t)expect, 0);
          for (int64_t q = 0; q < plan.n_positions; ++q)
            if (plan.column[q] >= 0) {
              EXPECT(plan.column[q] < expect, "column out of range");
              ++seen[plan.column[q]];
            }
          for (int64_t p = 0; p < expect; ++p)
            EXPECT(seen[p] == 1, "mode %d bins %d: column %lld covered %d times", mode, n_bins,
                   (long long)p, seen[p]);
          for (int n_chunks : {1, 4, 13, 32, 100, 256}) {
            for (int waves : {4, 8}) {
              tc::Chunking chunking;
              tc::build_chunking(plan, n_chunks, waves, chunking);
              int64_t covered = 0;
              for (const tc::Chunk& chunk : chunking.chunks) {
                EXPECT(chunk.q_begin % block == 0, "chunk not block aligned");
                covered += chunk.q_end - chunk.q_begin;
              }
              EXPECT(covered == plan.n_positions, "chunks cover %lld of %lld positions",
                     (long long)covered, (long long)plan.n_positions);
            }
          }
        }
    }
}

static void check_quad() {
  std::mt19937_64 rng(3);
  std::normal_distribution<double> normal(0.0, 1.0);
  for (int n_bins : {1, 3, 4, 8, 13, 30, 100})
    for (int n_r : {1, 4, 19, 23, 45})
      for (int separate = 0; separate < 2; ++separate) {
        const int64_t n_pairs = (int64_t)n_bins * (n_bins + 1) / 2;
        std::vector<double> matrix((size_t)n_r * n_pairs);
        for (double& v : matrix) v = (float)std::exp(normal(rng));
        std::vector<uint8_t> central(n_bins);
        for (int g = 0; g < n_bins; ++g) central[g] = g % 3 != 1;
        tc::Plan plan;
        tc::build_plan(0, n_bins, central.data(), tc::kF64Block, 56, plan);
        const bool by_type = separate || plan.n_central % 4 == 0 || plan.n_central == n_bins;
        tc::QuadLayout layout;
        tc::build_quad_layout(n_bins, plan.n_central, by_type, layout);
        const tc::QuadTiling tiling = tc::quad_tiling(n_r);
        std::vector<double> table;
        tc::fill_quad_table(layout, plan.perm, n_r, n_pairs, matrix.data(), false, tiling, table);
        const int64_t n_draws = 70, ldb = 128;
        std::vector<double> densities((size_t)n_bins * ldb, 0.0);   // library bin order
        for (int g = 0; g < n_bins; ++g)
          for (int64_t b = 0; b < n_draws; ++b) densities[(size_t)g * ldb + b] = std::exp(normal(rng));
        for (int max_waves : {1, 7, 64, 2048}) {
          tc::QuadSchedule schedule;
          tc::build_quad_schedule(layout, (int)(ldb / 32), tiling.n_rtiles, 1, separate != 0,
                                  max_waves, 8, schedule);
          const int n_comp = separate ? 3 : 1;
          std::vector<double> out((size_t)n_draws * n_comp * n_r, 0.0);
          // with and without the workgroup-level merge of the slabs
          tc::QuadMergePlan merge;
          if (max_waves != 7)
            tc::merge_quad_schedule(layout, tiling.n_rtiles, separate != 0, tc::kQuadWavesPerBlock,
                                    max_waves == 64 ? 3 : 12, schedule, merge);
          tc::quad_emulate(layout, schedule, tiling, table, densities.data(), ldb, n_draws, n_r,
                           separate != 0, out.data(), max_waves != 7 ? &merge : nullptr);
          EXPECT(schedule.group_begin.back() == schedule.n_slabs, "slab count after the merge");
          // direct evaluation: sum_p c_p T[r][p] n_i n_j over the packed columns
          for (int64_t b = 0; b < n_draws; b += 23)
            for (int r = 0; r < n_r; ++r) {
              double expect[3] = {0.0, 0.0, 0.0};
              for (int i = 0; i < n_bins; ++i)
                for (int j = 0; j <= i; ++j) {
                  const int ri = plan.perm[i], rj = plan.perm[j];
                  const int hi = ri > rj ? ri : rj, lo = ri > rj ? rj : ri;
                  const double t = matrix[(size_t)r * n_pairs + tc::packed_index(hi, lo)];
                  const double w = (i == j ? 1.0 : 2.0) * densities[(size_t)i * ldb + b] *
                                   densities[(size_t)j * ldb + b];
                  const int ci = i < plan.n_central ? 0 : 1, cj = j < plan.n_central ? 0 : 1;
                  expect[separate ? ci + cj : 0] += t * w;
                }
              for (int c = 0; c < n_comp; ++c) {
                const double got = out[((size_t)b * n_comp + c) * n_r + r];
                EXPECT(std::fabs(got - expect[c]) <= 1e-11 * std::fabs(expect[c]) + 1e-300,
                       "quad emulate bins %d r %d sep %d waves %d: %g vs %g", n_bins, n_r, separate,
                       max_waves, got, expect[c]);
              }
            }
        }
      }
}

static void check_fastmath() {
  std::vector<double> table(tc::fm::kTableDoubles);
  tc::fm::build_tables(table.data());
  const tc::fm::Consts k = tc::fm::make_consts();
  double worst = 0.0;
  for (int i = -80000; i <= 80000; ++i) {
    const double x = i * 1e-4;
    worst = std::max(worst, std::fabs(tc::fm::erf_fast(table.data(), k, x) - std::erf(x)));
  }
  EXPECT(worst < 1e-15, "erf_fast deviates by %g", worst);
  for (int i = -3000; i <= 3000; ++i) {
    const double z = i * 0.1;
    const double got = tc::fm::exp2_fast(table.data(), k, z), expect = std::exp2(z);
    EXPECT(std::fabs(got / expect - 1.0) < 1e-15, "exp2_fast(%g)", z);
  }
  for (double y = 1e-300; y < 1e300; y *= 1.7) {
    const double got = tc::fm::log2_fast(table.data(), k, y), expect = std::log2(y);
    EXPECT(std::fabs(got - expect) <= 1e-15 * std::max(1.0, std::fabs(expect)), "log2_fast(%g)", y);
  }
  // out-of-range and non-finite arguments must stay inside the tables
  for (double x : {1e308, -1e308, HUGE_VAL, -HUGE_VAL, std::nan(""), 0.0, -0.0, 5e-324}) {
    (void)tc::fm::erf_fast(table.data(), k, x);
    (void)tc::fm::exp2_fast(table.data(), k, x);
    (void)tc::fm::exp10_fast(table.data(), k, x);
  }
}

static void check_cells() {
  std::mt19937_64 rng(4);
  std::uniform_real_distribution<double> uniform(0.0, 1.0);
  // (250 000 points: the sorts run on several host threads)
  for (int64_t n : {0, 1, 17, 5000, 250000})
    for (int shape = 0; shape < 3; ++shape) {
      const double box[3] = {100.0, shape == 1 ? 45.0 : 100.0, shape == 2 ? 70.0 : 130.0};
      std::vector<double> pos((size_t)n * 3);
      std::vector<int32_t> label((size_t)n);
      for (int64_t p = 0; p < n; ++p) {
        for (int d = 0; d < 3; ++d) pos[3 * p + d] = uniform(rng) * box[d];
        label[p] = (int32_t)(rng() % 7);
      }
      if (n > 3) {
        pos[0] = 0.0;
        pos[4] = box[1];      // exactly on the upper face
        pos[8] = box[2];
      }
      const tc::CellGrid grid = tc::make_cell_grid(box, 20.0, 40.0, n);
      EXPECT(grid.nx >= 1 && grid.ny >= 1 && grid.nz >= 1, "empty grid");
      EXPECT((grid.ny == 1) == (grid.reach_y == 0), "reach does not match the cell count");
      // neighbour cells to each side are distinct cells and cover the reach
      EXPECT(grid.nx >= 2 * grid.reach_x + 1 && grid.ny >= 2 * grid.reach_y + 1 &&
                 grid.nz >= 2 * grid.reach_z + 1, "neighbour cells alias");
      EXPECT(grid.reach_x == 0 || grid.lx / grid.nx * grid.reach_x >= 20.0, "x cells too narrow");
      EXPECT(grid.reach_z == 0 || grid.lz / grid.nz * grid.reach_z >= 40.0, "z cells too narrow");
      tc::CellSort sorted;
      EXPECT(tc::sort_into_cells(grid, pos.data(), label.data(), n, sorted) == -1,
             "points reported outside the box");
      EXPECT((int64_t)sorted.x.size() == n && sorted.cell_start.back() == n, "cell sort lost points");
      for (int c = 0; c < grid.n_cells(); ++c)
        for (int32_t s = sorted.cell_start[c]; s < sorted.cell_start[c + 1]; ++s) {
          const int cx = std::min(grid.nx - 1, (int)(sorted.x[s] / grid.lx * grid.nx));
          EXPECT(cx == c / (grid.ny * grid.nz), "point in the wrong cell");
        }
      {
        // stable: with the input index as label, the points of a cell keep their input order
        // (whatever the number of threads), and sorting by that label changes nothing
        std::vector<int32_t> index((size_t)n);
        std::iota(index.begin(), index.end(), 0);
        tc::CellSort by_index;
        EXPECT(tc::sort_into_cells(grid, pos.data(), index.data(), n, by_index) == -1,
               "points reported outside the box");
        EXPECT(by_index.cell_start == sorted.cell_start && by_index.x == sorted.x,
               "cell sort depends on the labels");
        for (int c = 0; c < grid.n_cells(); ++c)
          for (int32_t s = by_index.cell_start[c] + 1; s < by_index.cell_start[c + 1]; ++s)
            EXPECT(by_index.label[s - 1] < by_index.label[s], "cell sort is not stable");
        const std::vector<double> before = by_index.z;
        tc::sort_cells_by_label(by_index);
        EXPECT(by_index.z == before, "label sort moved points that were in order");
      }
      // labelled work items: sorted by label inside the cells, bounded points and labels
      const tc::CellSort unsorted = sorted;
      tc::sort_cells_by_label(sorted);
      for (int c = 0; c < grid.n_cells(); ++c) {
        // (stable inside a label: equal labels keep the order they had in the cell)
        std::vector<int32_t> order((size_t)(sorted.cell_start[c + 1] - sorted.cell_start[c]));
        std::iota(order.begin(), order.end(), sorted.cell_start[c]);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
          return unsorted.label[a] < unsorted.label[b];
        });
        for (size_t k = 0; k < order.size(); ++k)
          EXPECT(sorted.x[sorted.cell_start[c] + k] == unsorted.x[order[k]] &&
                     sorted.label[sorted.cell_start[c] + k] == unsorted.label[order[k]],
                 "label sort differs from a stable sort");
      }
      for (int c = 0; c < grid.n_cells(); ++c)
        for (int32_t s = sorted.cell_start[c] + 1; s < sorted.cell_start[c + 1]; ++s)
          EXPECT(sorted.label[s - 1] <= sorted.label[s], "labels not sorted inside a cell");
      for (int block : {1, 3, 8}) {
        tc::LabelBlocks blocks;
        tc::build_label_blocks(sorted, 7, block, blocks);
        EXPECT(blocks.n_blocks == (7 + block - 1) / block, "number of label blocks");
        const size_t stride = (size_t)blocks.n_blocks + 1;
        int64_t covered = 0;
        for (int c = 0; c < grid.n_cells(); ++c) {
          EXPECT(blocks.start[c * stride] == sorted.cell_start[c] &&
                     blocks.start[c * stride + blocks.n_blocks] == sorted.cell_start[c + 1],
                 "label blocks do not span the cell");
          for (int k = 0; k < blocks.n_blocks; ++k) {
            EXPECT(blocks.start[c * stride + k] <= blocks.start[c * stride + k + 1],
                   "label block offsets decrease");
            for (int32_t p = blocks.start[c * stride + k]; p < blocks.start[c * stride + k + 1];
                 ++p) {
              EXPECT(sorted.label[p] / block == k, "point in the wrong label block");
              ++covered;
            }
          }
        }
        EXPECT(covered == n, "label blocks cover %lld of %lld points", (long long)covered,
               (long long)n);
        // work units: every (block, block, cell) exactly once
        for (int target : {1, 40, 4096}) {
          tc::PairUnits units;
          tc::build_pair_units(grid, sorted, sorted, blocks.n_blocks, blocks.n_blocks, target,
                               units);
          std::vector<int> seen((size_t)blocks.n_blocks * blocks.n_blocks * grid.n_cells(), 0);
          for (size_t u = 0; u < units.block1.size(); ++u) {
            EXPECT(units.cell_begin[u] < units.cell_end[u] && units.cell_end[u] <= grid.n_cells(),
                   "unit cell range");
            for (int c = units.cell_begin[u]; c < units.cell_end[u]; ++c)
              ++seen[((size_t)units.block1[u] * blocks.n_blocks + units.block2[u]) *
                         grid.n_cells() + c];
          }
          for (int v : seen) EXPECT(v == 1, "a (block, block, cell) is covered %d times", v);
        }
      }
      if (n > 0) {
        pos[2] = -1.0;
        EXPECT(tc::sort_into_cells(grid, pos.data(), nullptr, n, sorted) == 0,
               "point outside the box not reported");
      }
    }
}

int main() {
  check_quadrature();
  check_splines();
  check_plans();
  check_quad();
  check_fastmath();
  check_cells();
  if (g_failures != 0) {
    printf("%d check(s) failed\n", g_failures);
    return 1;
  }
  printf("host sanitizer driver: all checks passed\n");
  return 0;
}
