// Runtime and host-side helper entry points of the C ABI (include/tabcorr_amd.h):
// error reporting, device management, and the pure host functions (quadrature nodes,
// pair indices, spline matrices, plan self-check) that are testable without a GPU.
#include <dlfcn.h>
#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "internal.h"
#include "series.h"

namespace tc {
namespace host {

namespace {
thread_local std::string g_last_error;
}

// roctx: librocprofiler-sdk-roctx is what rocprofv3 --marker-trace listens to; the older
// roctracer library is the fallback.  Resolved on first use, never required.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
const Roctx& roctx() {
  static const Roctx table = [] {
    Roctx r;
    // only when a profiler is in the process (rocprofv3 preloads its tool library) or the
    // application itself already uses roctx: a plain run never loads anything
    bool profiled = false;
    for (const char* name : {"librocprofiler-sdk-tool.so", "librocprofiler-sdk-tool.so.1",
                             "librocprofiler-sdk-roctx.so.1", "libroctx64.so.4"})
      if (void* handle = dlopen(name, RTLD_NOLOAD | RTLD_NOW)) {
        profiled = true;
        (void)handle;
      }
    if (!profiled) return r;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                             "libroctx64.so.4", "libroctx64.so"}) {
      void* handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle == nullptr) continue;
      *(void**)(&r.push) = dlsym(handle, "roctxRangePushA");
      *(void**)(&r.pop) = dlsym(handle, "roctxRangePop");
      if (r.push != nullptr && r.pop != nullptr) break;
      r.push = nullptr;
      r.pop = nullptr;
    }
    return r;
  }();
  return table;
}
}  // namespace

void range_push(const char* name) {
  const Roctx& r = roctx();
  if (r.push != nullptr) (void)r.push(name);
}

void range_pop() {
  const Roctx& r = roctx();
  if (r.pop != nullptr) (void)r.pop();
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

int create_lane_streams(int count, hipStream_t* streams) {
  hipStream_t ballast[4] = {nullptr, nullptr, nullptr, nullptr};
  int status = TC_OK;
  for (hipStream_t& stream : ballast)
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) stream = nullptr;
  for (int l = 0; l < count && status == TC_OK; ++l) {
    const hipError_t error = hipStreamCreateWithFlags(&streams[l], hipStreamNonBlocking);
    if (error != hipSuccess)
      status = fail(TC_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(error));
  }
  for (hipStream_t stream : ballast)
    if (stream != nullptr) (void)hipStreamDestroy(stream);
  return status;
}

// ---- copies out of the staging areas on a few host threads ------------------------------------
//
// A synchronous host call ends with the results' way from the page-locked staging area to the
// caller's arrays: 1.6 MB for 10^4 draws of 19 r values, 55-75 us on one core (the source
// comes from memory, not from a cache: the device wrote it) -- more than the kernels take.
// Three helper threads share that copy with the calling thread (slices of 64 KB handed out by
// an atomic counter, so whoever arrives takes what is left: the caller never waits for a
// sleeping helper to wake up).  The helpers spin for a while after a job -- an MCMC loop's next
// call finds them awake -- and then sleep on a condition variable; they are started by the
// first large copy and joined when the library is unloaded.
namespace {
class CopyPool {
 public:
  static constexpr int kMaxHelpers = 7;
  static constexpr size_t kSlice = 64 * 1024;
  // Everything the helper threads touch lives in one heap block: a forked child, in which those
  // threads do not exist (and may have died holding the mutex), abandons the block and starts
  // over -- joining or destroying what fork() did not copy would hang or abort.
  struct State {
    // A job's description.  Two slots, taken by the job number's parity: a helper that comes
    // late for job J reads slot J & 1 while copy() fills the other one for job J + 1, and its
    // ticket (which carries J) no longer matches by the time the slot is filled again.  The
    // fields are atomics: a stale reader may see any mix of old and new values -- it is turned
    // away by the ticket -- but never reads a field that is being written non-atomically.
    struct Job {
      std::atomic<size_t> n_slices{0}, bytes{0};
      std::atomic<char*> dst{nullptr};
      std::atomic<const char*> src{nullptr};
    };
    std::mutex mutex;
    std::condition_variable wake;
    std::vector<std::thread> threads;
    std::atomic<uint64_t> job{0}, next{0};
    std::atomic<size_t> done{0};
    std::atomic<int> sleeping{0};
    Job jobs[2];
    bool stop = false;
  };
  CopyPool() {
    static bool registered = false;
    if (!registered) {
      registered = true;
      pthread_atfork(nullptr, nullptr, [] { instance().abandon(); });
    }
  }
  ~CopyPool() { stop_and_join(); }
  static CopyPool& instance() {
    static CopyPool pool;
    return pool;
  }
  // helpers: 0 .. kMaxHelpers threads beside the caller; spin_us: how long a helper polls for
  // the next job before it sleeps (0: it sleeps at once -- one rank per core)
  void configure(int helpers, int spin_us) {
    std::lock_guard<std::mutex> call(call_mutex_);
    stop_and_join();
    helpers_ = std::max(0, std::min(kMaxHelpers, helpers));
    spin_us_ = std::max(0, spin_us);
  }
  void copy(void* dst, const void* src, size_t bytes) {
    std::lock_guard<std::mutex> call(call_mutex_);       // one job at a time
    if (helpers_ == 0) {
      memcpy(dst, src, bytes);
      return;
    }
    if (state_ == nullptr) state_ = new State;
    State& s = *state_;
    if (s.threads.empty()) {
      const int spins = spin_us_ * 200;                  // (~5 ns per pause)
      for (int i = 0; i < helpers_; ++i) s.threads.emplace_back([&s, spins] { run(s, spins); });
    }
    const uint64_t job = s.job.load(std::memory_order_relaxed) + 1;
    State::Job& slot = s.jobs[job & 1];
    const size_t n_slices = (bytes + kSlice - 1) / kSlice;
    slot.dst.store((char*)dst, std::memory_order_relaxed);
    slot.src.store((const char*)src, std::memory_order_relaxed);
    slot.bytes.store(bytes, std::memory_order_relaxed);
    slot.n_slices.store(n_slices, std::memory_order_relaxed);
    s.done.store(0, std::memory_order_relaxed);
    // the ticket counter carries the job's number in its high half: a helper that comes late
    // for an earlier job finds another number and leaves; the slot is published with it
    s.next.store(job << 32, std::memory_order_release);
    // (sequentially consistent on both sides of the wake-up handshake: with release / acquire
    // alone the store below may pass the load of `sleeping` -- x86 reorders a store with a later
    // load -- while a helper increments `sleeping` and still reads the old job number: it
    // would sleep through the job)
    s.job.store(job, std::memory_order_seq_cst);
    if (s.sleeping.load(std::memory_order_seq_cst) > 0) {
      std::lock_guard<std::mutex> lock(s.mutex);
      s.wake.notify_all();
    }
    work(s, job);
    // (every slice that was handed out has been copied when this returns: nobody touches the
    // job's buffers afterwards)
    while (s.done.load(std::memory_order_acquire) < n_slices) __builtin_ia32_pause();
  }

 private:
  static void work(State& s, uint64_t job) {
    const State::Job& slot = s.jobs[job & 1];
    for (;;) {
      uint64_t ticket = s.next.load(std::memory_order_acquire);
      if ((ticket >> 32) != (job & 0xffffffffu)) return;
      const size_t slice = (size_t)(ticket & 0xffffffffu);
      const size_t n_slices = slot.n_slices.load(std::memory_order_relaxed);
      const size_t bytes = slot.bytes.load(std::memory_order_relaxed);
      char* dst = slot.dst.load(std::memory_order_relaxed);
      const char* src = slot.src.load(std::memory_order_relaxed);
      if (slice >= n_slices) return;
      // (the exchange succeeds only while the ticket still carries this job's number: what was
      // read from the slot above then belongs to it)
      if (!s.next.compare_exchange_weak(ticket, ticket + 1, std::memory_order_acq_rel)) continue;
      const size_t begin = slice * kSlice, n = std::min(kSlice, bytes - begin);
      memcpy(dst + begin, src + begin, n);
      s.done.fetch_add(1, std::memory_order_release);
    }
  }
  static void run(State& s, int spins) {
    uint64_t seen = 0;
    for (;;) {
      // poll for the next job for a while, then sleep
      bool have = false;
      for (int spin = 0; spin < spins && !have; ++spin) {
        have = s.job.load(std::memory_order_acquire) != seen;
        if (!have) __builtin_ia32_pause();
      }
      if (!have) {
        std::unique_lock<std::mutex> lock(s.mutex);
        s.sleeping.fetch_add(1, std::memory_order_seq_cst);
        s.wake.wait(lock, [&] { return s.stop || s.job.load(std::memory_order_seq_cst) != seen; });
        s.sleeping.fetch_sub(1, std::memory_order_seq_cst);
        if (s.stop) return;
      }
      seen = s.job.load(std::memory_order_acquire);
      work(s, seen);
    }
  }
  void stop_and_join() {
    if (state_ == nullptr) return;
    {
      std::lock_guard<std::mutex> lock(state_->mutex);
      state_->stop = true;
    }
    state_->wake.notify_all();
    for (std::thread& thread : state_->threads)
      if (thread.joinable()) thread.join();
    delete state_;
    state_ = nullptr;
  }
  // In the child of a fork(): the helper threads were not copied.  The block they lived in is
  // left alone (its std::thread objects cannot be joined, its mutex may be locked for ever);
  // the next large copy starts new ones.
  void abandon() {
    state_ = nullptr;
    new (&call_mutex_) std::mutex;
  }
  std::mutex call_mutex_;
  State* state_ = nullptr;
  int helpers_ = 3, spin_us_ = 100;
};
}  // namespace

void configure_copy_pool(int helpers, int spin_us) {
  CopyPool::instance().configure(helpers, spin_us);
}

void parallel_copy(void* dst, const void* src, size_t bytes) {
  if (bytes < 4 * CopyPool::kSlice) {
    memcpy(dst, src, bytes);
    return;
  }
  CopyPool::instance().copy(dst, src, bytes);
}

// Page-locked host ranges handed out by tc_host_alloc or pinned by tc_host_register:
// begin -> (bytes, allocated by us).  The asynchronous entry points look their buffers up
// here (a map lookup under a mutex; hipPointerGetAttributes costs microseconds per call).
namespace {
struct PinnedRange {
  size_t bytes = 0;
  bool owned = false;
  uintptr_t device = 0;      // address of the range as the device sees it
};
std::mutex g_pinned_mutex;
std::map<uintptr_t, PinnedRange> g_pinned;
}  // namespace

bool is_pinned(const void* ptr, size_t bytes, void** device_ptr) {
  if (device_ptr != nullptr) *device_ptr = nullptr;
  if (ptr == nullptr) return false;
  const uintptr_t begin = (uintptr_t)ptr;
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  auto it = g_pinned.upper_bound(begin);
  if (it == g_pinned.begin()) return false;
  --it;
  if (begin < it->first || begin + bytes > it->first + it->second.bytes) return false;
  // (a range without a device address -- hipHostGetDevicePointer failed at registration -- has
  // none anywhere inside it: the callers then fall back to copy commands)
  if (device_ptr != nullptr && it->second.device != 0)
    *device_ptr = (void*)(it->second.device + (begin - it->first));
  return true;
}

}  // namespace host
}  // namespace tc

using namespace tc::host;

extern "C" {

const char* tc_last_error(void) { return last_error(); }

int tc_set_copy_threads(int helpers, int spin_us) {
  TC_CHECK(helpers >= 0 && helpers <= 7, "helpers must be in [0, 7]");
  TC_CHECK(spin_us >= 0 && spin_us <= 100000, "spin_us must be in [0, 100000]");
  configure_copy_pool(helpers, spin_us);
  return TC_OK;
}

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

int tc_host_alloc(void** ptr, size_t bytes) {
  TC_CHECK(ptr != nullptr, "ptr is NULL");
  *ptr = nullptr;
  const size_t size = std::max<size_t>(bytes, 1);
  TC_HIP(hipHostMalloc(ptr, size, hipHostMallocDefault));
  void* device = nullptr;
  if (hipHostGetDevicePointer(&device, *ptr, 0) != hipSuccess) device = *ptr;
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  g_pinned[(uintptr_t)*ptr] = PinnedRange{size, true, (uintptr_t)device};
  return TC_OK;
}

int tc_host_free(void* ptr) {
  if (ptr == nullptr) return TC_OK;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned.find((uintptr_t)ptr);
    TC_CHECK(it != g_pinned.end() && it->second.owned,
             "tc_host_free: not a pointer returned by tc_host_alloc");
    g_pinned.erase(it);
  }
  TC_HIP(hipHostFree(ptr));
  return TC_OK;
}

int tc_host_register(void* ptr, size_t bytes) {
  TC_CHECK(ptr != nullptr && bytes > 0, "invalid range");
  {
    // ranges must not overlap one another: a second registration of the same memory would
    // silently replace the first entry (and its unregistration strand the other)
    const uintptr_t begin = (uintptr_t)ptr;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto next = g_pinned.lower_bound(begin);
    if (next != g_pinned.end() && next->first < begin + bytes)
      return fail(TC_ERR_INVALID, "tc_host_register: the range overlaps a page-locked range "
                  "the library already knows");
    if (next != g_pinned.begin()) {
      auto previous = std::prev(next);
      if (previous->first + previous->second.bytes > begin)
        return fail(TC_ERR_INVALID, "tc_host_register: the range overlaps a page-locked range "
                    "the library already knows");
    }
  }
  TC_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  void* device = nullptr;
  if (hipHostGetDevicePointer(&device, ptr, 0) != hipSuccess) device = nullptr;
  bool raced = false;
  {
    // (the lock was dropped around hipHostRegister: another thread may have registered an
    // overlapping range meanwhile -- check again where the entry goes in)
    const uintptr_t begin = (uintptr_t)ptr;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto next = g_pinned.lower_bound(begin);
    raced = next != g_pinned.end() && next->first < begin + bytes;
    if (!raced && next != g_pinned.begin()) {
      auto previous = std::prev(next);
      raced = previous->first + previous->second.bytes > begin;
    }
    if (!raced) g_pinned[begin] = PinnedRange{bytes, false, (uintptr_t)device};
  }
  if (raced) {
    (void)hipHostUnregister(ptr);
    return fail(TC_ERR_INVALID, "tc_host_register: the range overlaps a page-locked range "
                "the library already knows");
  }
  return TC_OK;
}

int tc_host_unregister(void* ptr) {
  if (ptr == nullptr) return TC_OK;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned.find((uintptr_t)ptr);
    TC_CHECK(it != g_pinned.end() && !it->second.owned,
             "tc_host_unregister: not a range pinned by tc_host_register");
    g_pinned.erase(it);
  }
  TC_HIP(hipHostUnregister(ptr));
  return TC_OK;
}

int tc_host_is_pinned(const void* ptr, size_t bytes, int* pinned) {
  TC_CHECK(pinned != nullptr, "pinned is NULL");
  *pinned = is_pinned(ptr, bytes) ? 1 : 0;
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
  TC_CHECK(kind >= 0 && kind <= 5 && n >= 0 && x && y, "invalid arguments");
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
    if (kind >= 4) {
      double gauss;
      const double value = tc::fm::erf_gauss_fast(table.data(), k, x[i], &gauss);
      y[i] = kind == 4 ? value : gauss;
    }
  }
  return TC_OK;
}

int tc_debug_central_series(int n_gauss, double log_min, double log_max, double dist_index,
                            int64_t n, const double* log_m_min, const double* sigma,
                            double* series, double* nodes, int32_t* terms) {
  TC_CHECK(n_gauss >= 1 && n_gauss <= 4096 && n >= 0, "invalid arguments");
  TC_CHECK(log_m_min && sigma && series && nodes && terms, "NULL argument");
  static std::vector<double> table;
  if (table.empty()) {
    table.resize(tc::fm::kTableDoubles);
    tc::fm::build_tables(table.data());
  }
  const tc::fm::Consts kc = tc::fm::make_consts();
  // nodes and normalised weights as launch.hip: get_quadrature
  std::vector<double> x, w;
  tc::gauss_legendre(n_gauss, x, w);
  std::vector<double> log_m(n_gauss), weight(n_gauss);
  std::vector<long double> raw(n_gauss);
  long double norm = 0.0L;
  double m_ref = 0.0;
  for (int k = 0; k < n_gauss; ++k) {
    const double mass = std::pow(10.0, log_min + (log_max - log_min) * x[k]);
    if (k == 0) m_ref = mass;
    log_m[k] = std::log10(mass);
    raw[k] = (long double)w[k] * powl((long double)mass / (long double)m_ref,
                                      (long double)(dist_index + 1.0));
    norm += raw[k];
  }
  double m0 = 0.0;
  for (int k = 0; k < n_gauss; ++k) {
    weight[k] = (double)(raw[k] / norm);
    m0 += weight[k];
  }
  std::vector<double> consts(tc::series::kStride + tc::series::kPad);
  std::vector<int32_t> thresholds(tc::series::kThresholds);
  tc::series::bin_consts(n_gauss, log_m.data(), weight.data(), log_min, log_max, consts.data(),
                         thresholds.data());
  for (int64_t i = 0; i < n; ++i) {
    const double inv_sigma = 1.0 / sigma[i];
    double sum = 0.0;
    for (int k = 0; k < n_gauss; ++k)
      sum = fma(weight[k], tc::fm::erf_fast(table.data(), kc, (log_m[k] - log_m_min[i]) * inv_sigma),
                sum);
    nodes[i] = sum;
    const double magnitude = std::fabs(inv_sigma);
    uint64_t bits;
    memcpy(&bits, &magnitude, sizeof(bits));
    terms[i] = tc::series::terms_for(thresholds.data(), (int)(bits >> 32));
    series[i] = terms[i] > 0 ? tc::series::central_sum(table.data(), kc, log_m_min[i], inv_sigma,
                                                       consts.data(), m0, thresholds.data(),
                                                       (int)(bits >> 32))
                             : sum;
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

int tc_debug_node_groups(int n_bins, int n_central, const double* log_min,
                         const double* log_max, int32_t* begin, int32_t* member,
                         int* n_groups, int* n_central_groups) {
  TC_CHECK(n_bins >= 1 && n_central >= 0 && n_central <= n_bins, "invalid bin counts");
  TC_CHECK(log_min && log_max && begin && member && n_groups && n_central_groups,
           "NULL argument");
  tc::NodeGroups groups;
  tc::find_node_groups(n_bins, n_central, log_min, log_max, groups);
  std::copy(groups.begin.begin(), groups.begin.end(), begin);
  std::copy(groups.member.begin(), groups.member.end(), member);
  *n_groups = groups.n_groups;
  *n_central_groups = groups.n_central_groups;
  return TC_OK;
}

int tc_debug_satellite_series(int n_gauss, double log_min, double log_max, double dist_index,
                              int64_t n, const double* log_m0, const double* log_m1,
                              const double* alpha, double* series, double* nodes,
                              int32_t* terms) {
  TC_CHECK(n_gauss >= 1 && n_gauss <= 4096 && n >= 0, "invalid arguments");
  TC_CHECK(log_m0 && log_m1 && alpha && series && nodes && terms, "NULL argument");
  static std::vector<double> table;
  if (table.empty()) {
    table.resize(tc::fm::kTableDoubles);
    tc::fm::build_tables(table.data());
  }
  const tc::fm::Consts kc = tc::fm::make_consts();
  std::vector<double> x, w;
  tc::gauss_legendre(n_gauss, x, w);
  std::vector<double> mass(n_gauss), weight(n_gauss);
  std::vector<long double> raw(n_gauss);
  long double norm = 0.0L;
  for (int k = 0; k < n_gauss; ++k) {
    mass[k] = std::pow(10.0, log_min + (log_max - log_min) * x[k]);
    raw[k] = (long double)w[k] * powl((long double)mass[k] / (long double)mass[0],
                                      (long double)(dist_index + 1.0));
    norm += raw[k];
  }
  for (int k = 0; k < n_gauss; ++k) weight[k] = (double)(raw[k] / norm);
  std::vector<double> consts(tc::series::sat::kStride + tc::series::kPad);
  std::vector<int32_t> thresholds(tc::series::sat::kThresholds);
  tc::series::sat::bin_consts(n_gauss, mass.data(), weight.data(), log_min, log_max,
                              consts.data(), thresholds.data());
  for (int64_t i = 0; i < n; ++i) {
    // (kernels.hip.h: prepare_draw)
    const double m0 = tc::fm::exp10_fast(table.data(), kc, log_m0[i]);
    const double hi = log_m1[i] * tc::fm::kLog2Of10Hi;
    const double lo = fma(log_m1[i], tc::fm::kLog2Of10Hi, -hi) + log_m1[i] * tc::fm::kLog2Of10Lo;
    const double sat_scale = fma(-alpha[i] * tc::fm::kLn2, lo, 1.0);
    double sum = 0.0;
    for (int k = 0; k < n_gauss; ++k) {
      const double xk = mass[k] - m0;
      sum = fma(weight[k],
                tc::fm::exp2_fast(table.data(), kc,
                                  alpha[i] * tc::fm::log2_fast_offset(
                                                 table.data(), kc, xk > 1e-300 ? xk : 1e-300, hi),
                                  xk > 0.0),
                sum);
    }
    nodes[i] = sum * sat_scale;
    uint64_t bits;
    memcpy(&bits, &m0, sizeof(bits));
    terms[i] = alpha[i] >= 0.0 && alpha[i] <= 4.0
                   ? tc::series::sat::terms_for(thresholds.data(), (int)(bits >> 32))
                   : 0;
    if (terms[i] > 0) {
      const double base = consts[0] - m0;
      const double eps = consts[0] * tc::series::sat::reciprocal(base);
      const double s = tc::series::sat::binomial_sum(consts.data(), eps, alpha[i],
                                                     thresholds.data(), (int)(bits >> 32));
      series[i] = s * tc::fm::exp2_fast(table.data(), kc,
                                        alpha[i] * tc::fm::log2_fast_offset(table.data(), kc, base,
                                                                            hi)) * sat_scale;
    } else {
      series[i] = nodes[i];
    }
  }
  return TC_OK;
}

int tc_plan_debug(int mode, int n_bins, const uint8_t* is_central, int n_chunks,
                  int64_t* n_entries, int32_t* entry_pair, int32_t* entry_chunk,
                  int32_t* entry_class) {
  TC_CHECK(mode == TC_MODE_AUTO || mode == TC_MODE_CROSS, "invalid mode");
  TC_CHECK(n_bins >= 1 && is_central && n_entries, "invalid arguments");
  tc::Plan plan;
  tc::build_plan(mode, n_bins, is_central, tc::kF64Block, 56,
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

int tc_debug_triangle_parts(int n_rb, int n_parts, int32_t* rb0, int32_t* cb0, int32_t* count) {
  TC_CHECK(n_rb >= 1 && n_parts >= 1 && rb0 && cb0 && count, "invalid argument");
  tc::triangle_parts(n_rb, n_parts, rb0, cb0, count);
  return TC_OK;
}

int tc_debug_quad_schedule(int n_bins, int n_central, int by_type, int n_tiles, int n_rtiles,
                           int n_tables, int separate, int max_waves, int min_units, int order,
                           int* n_waves, int* n_runs, int* n_slabs, int64_t* units_min,
                           int64_t* units_max) {
  TC_CHECK(n_bins >= 1 && n_central >= 0 && n_central <= n_bins && n_tiles >= 1 &&
               n_rtiles >= 1 && max_waves >= 1,
           "invalid arguments");
  tc::QuadLayout layout;
  tc::build_quad_layout(n_bins, n_central, by_type != 0, layout);
  tc::QuadSchedule schedule;
  tc::build_quad_schedule(layout, n_tiles, n_rtiles, n_tables, separate != 0, max_waves,
                          min_units, schedule, order);
  const int tables = std::max(1, n_tables);
  // every (tile, r tile, component, table, unit) exactly once; runs stay inside one block
  // row sequence of their component; slabs of a group consecutive and in order
  std::vector<uint8_t> seen((size_t)n_tiles * n_rtiles * tables * layout.n_units, 0);
  const int groups_per_rtile = separate ? (int)layout.comps.size() : 1;
  int64_t lo = -1, hi = 0;
  int next_slab = 0;
  std::vector<uint8_t> slab_seen((size_t)std::max(schedule.n_slabs, 0), 0);
  for (int w = 0; w < schedule.n_waves; ++w) {
    int64_t units = 0;
    for (int ri = schedule.wave_runs[w]; ri < schedule.wave_runs[w + 1]; ++ri) {
      const tc::QuadRun& run = schedule.runs[ri];
      if (run.comp < 0 || run.comp >= (int)layout.comps.size())
        return fail(TC_ERR_INVALID, "run %d: component", ri);
      const tc::QuadComp& comp = layout.comps[run.comp];
      int rb = run.rb0, cb = run.cb0;
      for (int k = 0; k < run.count; ++k) {
        if (rb >= comp.n_rb || cb >= tc::quad_row_length(comp, rb))
          return fail(TC_ERR_INVALID, "run %d leaves its component", ri);
        const int64_t unit = comp.unit_base + (comp.triangular ? (int64_t)rb * (rb + 1) / 2 + cb
                                                                : (int64_t)rb * comp.n_cb + cb);
        const size_t index =
            ((((size_t)run.tile * n_rtiles + run.rtile) * tables + run.table) * layout.n_units) +
            unit;
        if (seen[index]++) return fail(TC_ERR_INVALID, "unit covered twice (run %d)", ri);
        if (++cb == tc::quad_row_length(comp, rb)) {
          ++rb;
          cb = 0;
        }
      }
      units += run.count;
      if (run.slab >= 0) {
        // every slab written exactly once (in run order unless the order is table-major)
        if (run.slab >= schedule.n_slabs || slab_seen[run.slab]++)
          return fail(TC_ERR_INVALID, "slab %d written twice or out of range (run %d)",
                      run.slab, ri);
        if (order == tc::kQuadTileMajor && run.slab != next_slab)
          return fail(TC_ERR_INVALID, "slab order (run %d)", ri);
        const int64_t group = ((int64_t)run.tile * n_rtiles + run.rtile) * groups_per_rtile +
                              (separate ? run.comp : 0);
        if (run.slab < schedule.group_begin[group] || run.slab >= schedule.group_begin[group + 1])
          return fail(TC_ERR_INVALID, "slab %d outside its group", run.slab);
        ++next_slab;
      }
    }
    if (schedule.wave_runs[w + 1] > schedule.wave_runs[w] &&
        schedule.runs[schedule.wave_runs[w + 1] - 1].slab < 0)
      return fail(TC_ERR_INVALID, "wave %d does not flush its sums", w);
    lo = lo < 0 ? units : std::min(lo, units);
    hi = std::max(hi, units);
  }
  for (size_t k = 0; k < seen.size(); ++k)
    if (!seen[k]) return fail(TC_ERR_INVALID, "unit %zu not covered", k);
  if (next_slab != schedule.n_slabs) return fail(TC_ERR_INVALID, "slab count");
  if (n_waves) *n_waves = schedule.n_waves;
  if (n_runs) *n_runs = (int)schedule.runs.size();
  if (n_slabs) *n_slabs = schedule.n_slabs;
  if (units_min) *units_min = lo;
  if (units_max) *units_max = hi;
  return TC_OK;
}

int tc_debug_quad_emulate(int n_bins, int n_r, const double* tpcf_matrix,
                          const uint8_t* is_central, int by_type, int separate,
                          const double* densities, int64_t ldb, int64_t n_draws,
                          int max_waves, int min_units, int order, double* out) {
  TC_CHECK(n_bins >= 1 && n_r >= 1 && tpcf_matrix && is_central && densities && out &&
               ldb % 64 == 0 && n_draws >= 1 && n_draws <= ldb,
           "invalid arguments");
  tc::Plan plan;
  tc::build_plan(TC_MODE_AUTO, n_bins, is_central, tc::kF64Block, 56, plan);
  tc::QuadLayout layout;
  tc::build_quad_layout(n_bins, plan.n_central, by_type != 0, layout);
  const tc::QuadTiling tiling = tc::quad_tiling(n_r);
  std::vector<double> table;
  tc::fill_quad_table(layout, plan.perm, n_r, (int64_t)n_bins * (n_bins + 1) / 2, tpcf_matrix,
                      false, tiling, table);
  tc::QuadSchedule schedule;
  tc::build_quad_schedule(layout, (int)(ldb / 32), tiling.n_rtiles, 1, separate != 0,
                          max_waves, min_units, schedule, order);
  // densities arrive in the reference's bin order; the kernel sees library order
  std::vector<double> ordered((size_t)n_bins * ldb);
  for (int g = 0; g < n_bins; ++g)
    std::copy(densities + (size_t)plan.perm[g] * ldb, densities + (size_t)(plan.perm[g] + 1) * ldb,
              ordered.begin() + (size_t)g * ldb);
  // (the workgroup-level merge of the slabs as the launch layer applies it)
  tc::QuadMergePlan merge;
  tc::merge_quad_schedule(layout, tiling.n_rtiles, separate != 0, tc::kQuadWavesPerBlock, 12,
                          schedule, merge);
  tc::quad_emulate(layout, schedule, tiling, table, ordered.data(), ldb, n_draws, n_r,
                   separate != 0, out, &merge);
  return TC_OK;
}

}  // extern "C"
