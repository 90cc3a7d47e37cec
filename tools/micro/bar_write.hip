// Can the host store straight into device memory (large BAR), and what does a host -> device
// -> host round trip cost with the mailbox there instead of in page-locked host memory?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/bar_write.hip -o /tmp/bar_write && /tmp/bar_write
// One thread of a resident kernel waits for call k in the mailbox and answers k in page-locked
// memory; every wait is bounded (2 s).
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <immintrin.h>

static sigjmp_buf jump;
static void on_fault(int) { siglongjmp(jump, 1); }

__global__ void echo(const unsigned long long* mailbox, unsigned long long* reply,
                     unsigned long long n_calls) {
  const unsigned long long begin = __builtin_amdgcn_s_memrealtime();
  for (unsigned long long k = 1; k <= n_calls; ++k) {
    for (;;) {
      const unsigned long long seen =
          __hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (seen >= k) break;
      if (__builtin_amdgcn_s_memrealtime() - begin > 200000000ull) return;   // 2 s
    }
    __hip_atomic_store(reply, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double round_trips(unsigned long long* mailbox_host_view, const unsigned long long* mailbox_device,
                          unsigned long long* reply, int n_calls) {
  *reply = 0;
  __atomic_store_n(mailbox_host_view, 0ull, __ATOMIC_RELEASE);
  hipStream_t stream;
  hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
  hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, stream, mailbox_device, reply,
                     (unsigned long long)n_calls);
  volatile unsigned long long* answer = reply;
  const auto t0 = std::chrono::steady_clock::now();
  bool lost = false;
  for (int k = 1; k <= n_calls && !lost; ++k) {
    __atomic_store_n(mailbox_host_view, (unsigned long long)k, __ATOMIC_RELEASE);
    _mm_sfence();      // (a write-combining mapping keeps the store in its buffer otherwise)
    unsigned long long spins = 0;
    while (*answer != (unsigned long long)k)
      if (++spins > 300000000ull) { lost = true; break; }
  }
  const auto t1 = std::chrono::steady_clock::now();
  __atomic_store_n(mailbox_host_view, ~0ull, __ATOMIC_RELEASE);
  hipStreamSynchronize(stream);
  hipStreamDestroy(stream);
  if (lost) return -1.0;
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n_calls;
}

int main() {
  signal(SIGSEGV, on_fault);
  signal(SIGBUS, on_fault);
  unsigned long long* reply = nullptr;
  hipHostMalloc((void**)&reply, 4096, hipHostMallocDefault);
  {
    unsigned long long* mailbox = nullptr;
    hipHostMalloc((void**)&mailbox, 4096, hipHostMallocDefault);
    printf("mailbox in page-locked host memory: %.2f us per round trip\n",
           round_trips(mailbox, mailbox, reply, 20000));
    hipHostFree(mailbox);
  }
  struct Kind { const char* name; int flags; };
  const Kind kinds[] = {{"hipMalloc", -1},
                        {"hipExtMallocWithFlags(hipDeviceMallocFinegrained)", (int)hipDeviceMallocFinegrained},
                        {"hipExtMallocWithFlags(hipDeviceMallocUncached)", (int)hipDeviceMallocUncached}};
  for (const Kind& kind : kinds) {
    unsigned long long* device = nullptr;
    hipError_t status = kind.flags < 0 ? hipMalloc((void**)&device, 4096)
                                       : hipExtMallocWithFlags((void**)&device, 4096, kind.flags);
    if (status != hipSuccess) {
      printf("%s: allocation failed (%s)\n", kind.name, hipGetErrorString(status));
      (void)hipGetLastError();
      continue;
    }
    hipMemset(device, 0, 4096);
    hipDeviceSynchronize();
    if (sigsetjmp(jump, 1) != 0) {
      printf("%s: a host store to the device address faults\n", kind.name);
      continue;
    }
    __atomic_store_n(device, 777ull, __ATOMIC_RELEASE);
    _mm_sfence();
    const unsigned long long back = __atomic_load_n(device, __ATOMIC_ACQUIRE);
    unsigned long long copied = 0;
    hipMemcpy(&copied, device, 8, hipMemcpyDeviceToHost);
    printf("%s: host store and load work (host reads back %llu, hipMemcpy from the device "
           "address reads %llu); ", kind.name, back, copied);
    fflush(stdout);
    printf("%.2f us per round trip\n", round_trips(device, device, reply, 20000));
    hipFree(device);
  }
  return 0;
}
