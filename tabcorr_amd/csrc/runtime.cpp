// Runtime and host-side helper entry points of the C ABI (include/tabcorr_amd.h):
// error reporting, device management, and the pure host functions (quadrature nodes,
// pair indices, spline matrices, plan self-check) that are testable without a GPU.
#include "internal.h"

namespace tc {
namespace host {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const char* format, ...) {
  char buffer[1024];
  va_list args;
  va_start(args, format);
  vsnprintf(buffer, sizeof(buffer), format, args);
  va_end(args);
  g_last_error = buffer;
  return code;
}

const char* last_error() { return g_last_error.c_str(); }

}  // namespace host
}  // namespace tc

using namespace tc::host;

extern "C" {

const char* tc_last_error(void) { return last_error(); }

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
  TC_CHECK(kind >= 0 && kind <= 3 && n >= 0 && x && y, "invalid arguments");
  static std::vector<double> table;
  if (table.empty()) {
    table.resize(tc::fm::kTableDoubles);
    tc::fm::build_tables(table.data());
  }
  const tc::fm::Consts k = tc::fm::make_consts();
  for (int64_t i = 0; i < n; ++i) {
    if (kind == 0) y[i] = tc::fm::erf_fast(table.data(), k, x[i]);
    if (kind == 1) y[i] = tc::fm::log2_fast(table.data(), k, x[i]);
    if (kind == 2) y[i] = tc::fm::exp2_fast(table.data(), k, x[i]);
    if (kind == 3) y[i] = tc::fm::exp10_fast(table.data(), k, x[i]);
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
  tc::build_plan(mode, n_bins, is_central, tc::kF64Block, env_int("TC_ROW_BUDGET", 56),
                 plan);
  tc::Chunking chunking;
  tc::build_chunking(plan, n_chunks, 4, chunking);
  *n_entries = plan.n_entries;
  if (entry_pair == nullptr) return TC_OK;
  // Visit every chunk as the kernel does: all positions of the chunk, bin pair from the
  // position tables, density rows from the rows its workgroup stages.
  std::vector<int> seen((size_t)plan.n_positions, 0);
  int64_t e = 0;
  for (size_t g = 0; g < chunking.groups.size(); ++g) {
    const tc::Group& group = chunking.groups[g];
    const int n_rows_j = group.j_hi - group.j_lo;
    const int n_rows = n_rows_j + (group.i_hi - group.i_lo);
    if (n_rows > chunking.max_rows) return fail(TC_ERR_INVALID, "group %zu: max_rows", g);
    for (int c = group.chunk_begin; c < group.chunk_begin + group.n_chunks; ++c) {
      const tc::Chunk& chunk = chunking.chunks[c];
      if ((chunk.q_begin % plan.block) != 0 || (chunk.q_end % plan.block) != 0)
        return fail(TC_ERR_INVALID, "chunk %d is not block aligned", c);
      for (int q = chunk.q_begin; q < chunk.q_end; ++q) {
        if (seen[q]++) return fail(TC_ERR_INVALID, "position %d covered twice", q);
        // the rows gathered for this position (padding included) must be staged
        const int row_j = plan.pos_j[q] - group.j_lo;
        if (row_j < 0 || row_j >= n_rows_j)
          return fail(TC_ERR_INVALID, "position %d: column bin not staged", q);
        if (mode == TC_MODE_AUTO) {
          const int row_i = plan.pos_i[q] + group.i_shift;
          if (row_i < 0 || row_i >= n_rows)
            return fail(TC_ERR_INVALID, "position %d: row bin not staged", q);
        }
        if (plan.column[q] < 0) continue;
        const int64_t column =
            mode == TC_MODE_AUTO
                ? tc::packed_index(plan.perm[plan.pos_i[q]], plan.perm[plan.pos_j[q]])
                : plan.perm[plan.pos_j[q]];
        if (column != plan.column[q])
          return fail(TC_ERR_INVALID, "bin pair mismatch at position %d", q);
        if (e >= plan.n_entries) return fail(TC_ERR_INVALID, "too many entries");
        entry_pair[e] = (int32_t)column;
        entry_chunk[e] = c;
        entry_class[e] = chunk.component;
        ++e;
      }
    }
  }
  for (int64_t q = 0; q < plan.n_positions; ++q)
    if (!seen[q]) return fail(TC_ERR_INVALID, "position %lld not covered", (long long)q);
  if (e != plan.n_entries) return fail(TC_ERR_INVALID, "entries missing");
  return TC_OK;
}

}  // extern "C"
