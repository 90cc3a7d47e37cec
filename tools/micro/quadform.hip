// Developer prototype: the contraction as a quadratic form on v_mfma_f64_16x16x4_f64.
//
//   xi_r[b] = sum_i n_i[b] * ( sum_{j <= i} c_ij T_r[i][j] n_j[b] )
//
// The inner sum is a GEMM over j (K), rows = (r, i) pairs, columns = draws; the B operand is
// the density row itself (no pair weights to form), the outer product with n_i is one FMA
// per (r, i, draw) after the K loop of a row of 4x4 bin blocks.  Work = "units" (one 4x4
// bin block x all r sub-tiles x 32 draws); the linearised (tile, unit) space is cut into
// equal contiguous ranges, one per resident wave (no LDS, no barriers, no workgroup
// structure).  Checks against a CPU evaluation of the packed sum and times the launch.
//
// hipcc -O3 --offload-arch=gfx950 tools/micro/quadform.hip -o tools/micro/quadform
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) d2* gl_d2;

struct Run {
  int tile;      // 32-draw tile
  int row0;      // first block row
  int col0;      // first block column in that row
  int count;     // units
  int slab;      // partial slab written after this run (-1: keep accumulating)
  int pad[3];
};

struct Args {
  const double* nbuf;   // [G][ldb]
  long ldb;
  int n_bins;
  const d2* table;      // [unit][UP][64] x 16 bytes
  unsigned table_bytes;
  const Run* runs;
  const int* wave_runs; // [n_waves + 1]
  int n_waves;
  int skip;            // ablation bits: 1 no A loads, 2 no B loads, 4 no E loads
  double* partial;      // [slab][4 U][32]
  unsigned long long* stamps;   // per wave: start, after first row, end (100 MHz), or NULL
};

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane through a buffer resource: address = base + voffset (per lane) +
// soffset (wave-uniform, an SGPR) + imm; reads beyond the resource's size return 0
template <int IMM = 0>
__device__ inline d2 bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset + IMM, soffset, 0));
}

template <int U>
__global__ __launch_bounds__(256, 2) void quad_kernel(Args a) {
  constexpr int UP = (U + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (wave >= a.n_waves) return;
  unsigned long long s0 = 0, s1 = 0;
  if (a.stamps) s0 = __builtin_amdgcn_s_memrealtime();
  typedef const __attribute__((address_space(4))) int* sc_int;
  typedef const __attribute__((address_space(4))) Run* sc_run;
  const int run_begin = ((sc_int)a.wave_runs)[wave], run_end = ((sc_int)a.wave_runs)[wave + 1];
  const int c = lane & 15, kq = lane >> 4;
  const unsigned row_bytes = (unsigned)(a.ldb * 8);   // one density row
  const unsigned off_a = lane * 16;             // table: [unit][UP][lane] x 16 bytes
  const unsigned off_e = c * 16;                // densities of draws (2c, 2c + 1)
  const unsigned off_b = kq * row_bytes + c * 16;
  const __amdgpu_buffer_rsrc_t rs_table =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.table, 0, a.table_bytes, 0x00020000);
  double F[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) F[u][0] = F[u][1] = 0.0;

  for (int ri = run_begin; ri < run_end; ++ri) {
    Run run;
    run.tile = ((sc_run)a.runs)[ri].tile;
    run.row0 = ((sc_run)a.runs)[ri].row0;
    run.col0 = ((sc_run)a.runs)[ri].col0;
    run.count = ((sc_run)a.runs)[ri].count;
    run.slab = ((sc_run)a.runs)[ri].slab;
    // densities of the tile's 32 draws: rows beyond the last bin read as zero
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.nbuf + (long)run.tile * 256), 0,
        (unsigned)(a.n_bins * row_bytes - run.tile * 256), 0x00020000);
    int row = run.row0, cj = run.col0, left = run.count;
    unsigned ua = (unsigned)(((long)row * (row + 1) / 2 + cj) * (UP * 1024));   // table offset
    d2 t0[UP], t1[UP], b0, b1;
    d4 D[U][2];
    auto fetch = [&](d2 (&t)[UP], d2& b, int col) {
      t[0] = bload<0>(rs_table, off_a, ua);
      if (UP > 1) t[1] = bload<1024>(rs_table, off_a, ua);
      if (UP > 2) t[2] = bload<2048>(rs_table, off_a, ua);
      b = bload(rs_n, off_b, 4 * col * row_bytes);
      ua += UP * 1024;
    };
    auto mma = [&](const d2 (&t)[UP], const d2& b, bool first) {
      if (first) {
        const d4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double av = (u & 1) ? t[u >> 1].y : t[u >> 1].x;
          D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, zero, 0, 0, 0);
          D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, zero, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double av = (u & 1) ? t[u >> 1].y : t[u >> 1].x;
          D[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.x, D[u][0], 0, 0, 0);
          D[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b.y, D[u][1], 0, 0, 0);
        }
      }
    };
    fetch(t0, b0, cj);
    while (left > 0) {
      // the units of one row of blocks inside this run: n >= 1
      const int n = row + 1 - cj < left ? row + 1 - cj : left;
      left -= n;
      d2 e[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) e[v] = bload(rs_n, off_e, (4 * row + v) * row_bytes);
      // unit 0 of the row from buffer 0; afterwards the next unit is in buffer 1
      // (every prefetch is unconditional -- one basic block per phase -- and at worst
      // reads a unit nobody uses; the resources bound every address)
      fetch(t1, b1, n > 1 ? cj + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(t0, b0, true);
      __builtin_amdgcn_sched_barrier(0);
      int t = 1;
      for (; t + 1 < n; t += 2) {
        // (the barriers keep the scheduler from sinking a prefetch next to its use)
        fetch(t0, b0, cj + t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
        fetch(t1, b1, t + 2 < n ? cj + t + 2 : 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t0, b0, false);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t < n) {
        // one unit left, in buffer 1; the next row's first unit goes to buffer 0
        fetch(t0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(t1, b1, false);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // the next row's first unit sits in buffer 1
#pragma unroll
        for (int p = 0; p < UP; ++p) t0[p] = t1[p];
        b0 = b1;
      }
      // epilogue of the row: F += D * n_i
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          F[u][0] = fma(D[u][0][v], e[v].x, F[u][0]);
          F[u][1] = fma(D[u][1][v], e[v].y, F[u][1]);
        }
      ++row;
      cj = 0;
      if (a.stamps && s1 == 0) s1 = __builtin_amdgcn_s_memrealtime();
    }
    if (run.slab >= 0) {
      double* out = a.partial + ((long)run.slab * (4 * U) + kq) * 32 + 2 * c;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        d2 v = {F[u][0], F[u][1]};
        *(d2*)(out + (long)(4 * u) * 32) = v;
        F[u][0] = F[u][1] = 0.0;
      }
    }
  }
  if (a.stamps && lane == 0) {
    a.stamps[wave * 3] = s0;
    a.stamps[wave * 3 + 1] = s1;
    a.stamps[wave * 3 + 2] = __builtin_amdgcn_s_memrealtime();
    a.stamps[a.n_waves * 3 + wave] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
        ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf) << 32);
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 100;
  const int R = argc > 2 ? atoi(argv[2]) : 19;
  const long n_draws = argc > 3 ? atol(argv[3]) : 10000;
  const int waves_per_simd = argc > 4 ? atoi(argv[4]) : 2;
  const int skip = argc > 5 ? atoi(argv[5]) : 0;
  const int zero_data = argc > 6 ? atoi(argv[6]) : 0;
  constexpr int U = 5;
  if ((R + 3) / 4 != U) { printf("this build has U = %d\n", U); return 1; }
  constexpr int UP = (U + 1) / 2;
  const long ldb = (n_draws + 63) / 64 * 64;
  const int n_tiles = (int)(ldb / 32);
  const int NB = (G + 3) / 4;
  const long n_units = (long)NB * (NB + 1) / 2;
  const long P = (long)G * (G + 1) / 2;

  std::mt19937_64 rng(1);
  std::normal_distribution<double> normal(2.0, 1.5);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::vector<double> T((size_t)R * P);
  for (double& v : T) v = (double)(float)std::exp(normal(rng));
  const int NBpad = (G + 3) / 4 * 4;
  std::vector<double> nbuf((size_t)NBpad * ldb, 0.0);
  for (size_t k = 0; k < (size_t)G * ldb; ++k) nbuf[k] = 1e-4 * uni(rng) * uni(rng);

  // table layout [unit][UP][lane] x 2 doubles: lane = k * 16 + m, m = r_local + 4 i_local
  std::vector<double> table((size_t)n_units * UP * 64 * 2, 0.0);
  for (int bi = 0; bi < NB; ++bi)
    for (int bj = 0; bj <= bi; ++bj) {
      const long unit = (long)bi * (bi + 1) / 2 + bj;
      for (int u = 0; u < U; ++u)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, k = lane >> 4;
          const int r = 4 * u + (m & 3), i = 4 * bi + (m >> 2), j = 4 * bj + k;
          double value = 0.0;
          if (r < R && i < G && j <= i) value = T[(size_t)r * P + (long)i * (i + 1) / 2 + j] * (i == j ? 1.0 : 2.0);
          table[((size_t)(unit * UP + u / 2) * 64 + lane) * 2 + (u & 1)] = value;
        }
    }

  // runs: equal contiguous ranges of the (tile, unit) space
  const int n_waves_max = 256 * 4 * waves_per_simd;
  const long total = (long)n_tiles * n_units;
  const int n_waves = (int)std::min<long>(n_waves_max, std::max<long>(1, total / 8));
  std::vector<Run> runs;
  std::vector<int> wave_runs(n_waves + 1, 0);
  std::vector<int> tile_slab_begin(n_tiles + 1, 0);
  int n_slabs = 0;
  {
    std::vector<std::vector<int>> tile_slabs(n_tiles);
    for (int w = 0; w < n_waves; ++w) {
      wave_runs[w] = (int)runs.size();
      long begin = total * w / n_waves, end = total * (w + 1) / n_waves;
      while (begin < end) {
        const int tile = (int)(begin / n_units);
        const long u0 = begin % n_units;
        const long stop = std::min<long>(end, (long)(tile + 1) * n_units);
        int row = (int)((std::sqrt(8.0 * u0 + 1.0) - 1.0) / 2.0);
        while ((long)(row + 1) * (row + 2) / 2 <= u0) ++row;
        while ((long)row * (row + 1) / 2 > u0) --row;
        Run run;
        run.tile = tile;
        run.row0 = row;
        run.col0 = (int)(u0 - (long)row * (row + 1) / 2);
        run.count = (int)(stop - begin);
        run.slab = n_slabs++;
        tile_slabs[tile].push_back(run.slab);
        runs.push_back(run);
        begin = stop;
      }
    }
    wave_runs[n_waves] = (int)runs.size();
  }
  printf("G %d R %d draws %ld: %d tiles x %ld units, %d waves, %zu runs, %d slabs\n", G, R,
         n_draws, n_tiles, n_units, n_waves, runs.size(), n_slabs);

  double *d_nbuf, *d_table, *d_partial;
  Run* d_runs;
  int* d_wave_runs;
  hipMalloc(&d_nbuf, nbuf.size() * 8);
  hipMalloc(&d_table, table.size() * 8);
  hipMalloc(&d_partial, (size_t)n_slabs * 4 * U * 32 * 8);
  hipMalloc(&d_runs, runs.size() * sizeof(Run));
  hipMalloc(&d_wave_runs, wave_runs.size() * 4);
  hipMemcpy(d_nbuf, nbuf.data(), nbuf.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_table, table.data(), table.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_runs, runs.data(), runs.size() * sizeof(Run), hipMemcpyHostToDevice);
  hipMemcpy(d_wave_runs, wave_runs.data(), wave_runs.size() * 4, hipMemcpyHostToDevice);
  hipMemset(d_partial, 0, (size_t)n_slabs * 4 * U * 32 * 8);
  if (zero_data) { hipMemset(d_nbuf, 0, nbuf.size() * 8); hipMemset(d_table, 0, table.size() * 8); }

  Args args;
  args.nbuf = d_nbuf;
  args.ldb = ldb;
  args.n_bins = G;
  args.table = (const d2*)d_table;
  args.table_bytes = (unsigned)(table.size() * 8);
  args.runs = d_runs;
  args.wave_runs = d_wave_runs;
  args.n_waves = n_waves;
  args.partial = d_partial;
  args.skip = skip;
  args.stamps = nullptr;
  const int blocks = (n_waves + 3) / 4;
  quad_kernel<U><<<blocks, 256>>>(args);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }

  // check a sample of draws
  std::vector<double> partial((size_t)n_slabs * 4 * U * 32);
  hipMemcpy(partial.data(), d_partial, partial.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (long b : {0L, 1L, 31L, 32L, 63L, 4097L, n_draws - 1}) {
    const int tile = (int)(b / 32), cc = (int)(b % 32);
    for (int r = 0; r < R; ++r) {
      double expect = 0.0;
      for (int i = 0; i < G; ++i)
        for (int j = 0; j <= i; ++j)
          expect += T[(size_t)r * P + (long)i * (i + 1) / 2 + j] * (i == j ? 1.0 : 2.0) *
                    nbuf[(size_t)i * ldb + b] * nbuf[(size_t)j * ldb + b];
      double got = 0.0;
      for (size_t k = 0; k < runs.size(); ++k)
        if (runs[k].tile == tile && runs[k].slab >= 0)
          got += partial[((size_t)runs[k].slab * 4 * U + r) * 32 + cc];
      worst = std::max(worst, std::fabs(got / expect - 1.0));
    }
  }
  printf("max rel deviation from the packed sum: %.3g\n", worst);

  {
    // per-wave timeline of one launch in the middle of a train of launches
    unsigned long long* d_stamps;
    hipMalloc(&d_stamps, (size_t)n_waves * 32);
    for (int it = 0; it < 200; ++it) quad_kernel<U><<<blocks, 256>>>(args);
    args.stamps = d_stamps;
    quad_kernel<U><<<blocks, 256>>>(args);
    args.stamps = nullptr;
    for (int it = 0; it < 20; ++it) quad_kernel<U><<<blocks, 256>>>(args);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st((size_t)n_waves * 4);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < n_waves; ++w) t0 = std::min(t0, st[w * 3]);
    std::vector<double> start, first, end;
    for (int w = 0; w < n_waves; ++w) {
      start.push_back((st[w * 3] - t0) * 0.01);
      first.push_back((st[w * 3 + 1] - st[w * 3]) * 0.01);
      end.push_back((st[w * 3 + 2] - t0) * 0.01);
    }
    auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("wave start (us after the first): p10 %.2f p50 %.2f p90 %.2f max %.2f\n", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1));
    printf("start -> end of first block row: p10 %.2f p50 %.2f p90 %.2f max %.2f\n", pct(first, .1), pct(first, .5), pct(first, .9), pct(first, 1));
    printf("wave end: p0 %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f\n", pct(end, 0), pct(end, .1), pct(end, .5), pct(end, .9), pct(end, 1));
    // waves per SIMD
    std::vector<int> per_simd;
    {
      std::vector<long> keys;
      for (int w = 0; w < n_waves; ++w) {
        const unsigned long long h = st[(size_t)n_waves * 3 + w];
        const long hw = (long)(h & 0xffffffff), xcc = (long)(h >> 32);
        keys.push_back(xcc * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 4) & 3));
      }
      std::sort(keys.begin(), keys.end());
      int hist[16] = {0}, simds = 0;
      for (size_t k = 0; k < keys.size();) {
        size_t e = k;
        while (e < keys.size() && keys[e] == keys[k]) ++e;
        hist[std::min<size_t>(15, e - k)]++;
        ++simds;
        k = e;
      }
      printf("SIMDs used %d; waves per SIMD histogram:", simds);
      for (int k = 1; k < 10; ++k) printf(" %d:%d", k, hist[k]);
      printf("\n");
    }
    hipFree(d_stamps);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    const int n_launch = rep == 0 ? 3000 : 2000;
    hipEventRecord(e0);
    for (int it = 0; it < n_launch; ++it) quad_kernel<U><<<blocks, 256>>>(args);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)n_draws * (2.0 * R * P + 3.0 * P);
    printf("%d launches: %.2f us per launch, %.1f TFLOP/s algorithmic (%.3f of 78.6)\n", n_launch,
           ms * 1e3 / n_launch, flop / (ms * 1e-3 / n_launch) / 1e12,
           flop / (ms * 1e-3 / n_launch) / 1e12 / 78.6);
  }
  return 0;
}
