// Multi-GPU gather of the C ABI (include/tabcorr_amd.h): RCCL, resolved at run time.
//
// One process per GPU (torchrun style).  librccl is opened lazily so that single-GPU use
// and CPU-only hosts never load it.  The only collective of the path is the gather of the
// per-rank results on the root; a barrier is provided for timing.
#include <dlfcn.h>

#include "internal.h"

using namespace tc::host;

namespace {

typedef struct ncclComm* nccl_comm_t;
typedef struct { char internal[TC_UNIQUE_ID_BYTES]; } nccl_unique_id;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_unique_id*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*Gather)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle != nullptr) return TC_OK;
  const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  void* handle = nullptr;
  for (const char* name : names) {
    handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (handle != nullptr) break;
  }
  if (handle == nullptr) return fail(TC_ERR_RCCL, "cannot load librccl: %s", dlerror());
#define TC_SYM(field, symbol)                                                  \
  *(void**)(&g_rccl.field) = dlsym(handle, symbol);                            \
  if (g_rccl.field == nullptr)                                                 \
    return fail(TC_ERR_RCCL, "librccl lacks %s", symbol);
  TC_SYM(GetUniqueId, "ncclGetUniqueId")
  TC_SYM(CommInitRank, "ncclCommInitRank")
  TC_SYM(CommDestroy, "ncclCommDestroy")
  TC_SYM(Gather, "ncclGather")
  TC_SYM(AllReduce, "ncclAllReduce")
  TC_SYM(GetErrorString, "ncclGetErrorString")
#undef TC_SYM
  g_rccl.handle = handle;
  return TC_OK;
}

#define TC_RCCL(call)                                                          \
  do {                                                                         \
    int tc_rccl_status = (call);                                               \
    if (tc_rccl_status != 0)                                                   \
      return fail(TC_ERR_RCCL, "%s failed: %s", #call,                         \
                  g_rccl.GetErrorString(tc_rccl_status));                      \
  } while (0)

constexpr int kNcclFloat64 = 8;
constexpr int kNcclSum = 0;

}  // namespace

struct tc_comm {
  int device = 0;
  int n_ranks = 1;
  int rank = 0;
  nccl_comm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr;
  hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};  // per send-buffer slot
  double* token = nullptr;   // one double for the barrier all-reduce
};

extern "C" {

int tc_comm_unique_id(void* id) {
  TC_CHECK(id != nullptr, "id is NULL");
  int status = load_rccl();
  if (status != TC_OK) return status;
  nccl_unique_id unique;
  TC_RCCL(g_rccl.GetUniqueId(&unique));
  memcpy(id, &unique, TC_UNIQUE_ID_BYTES);
  return TC_OK;
}

int tc_comm_create(const void* id, int n_ranks, int rank, tc_comm** out) {
  TC_CHECK(out != nullptr && id != nullptr, "NULL argument");
  *out = nullptr;
  TC_CHECK(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "invalid rank %d of %d", rank,
           n_ranks);
  int status = load_rccl();
  if (status != TC_OK) return status;
  std::unique_ptr<tc_comm> c(new tc_comm);
  TC_HIP(hipGetDevice(&c->device));
  c->n_ranks = n_ranks;
  c->rank = rank;
  nccl_unique_id unique;
  memcpy(&unique, id, TC_UNIQUE_ID_BYTES);
  TC_RCCL(g_rccl.CommInitRank(&c->comm, n_ranks, unique, rank));
  // RCCL prints a version banner through C stdio; push it out now so that it cannot
  // trail the caller's own output (bench.py's JSON line) at process exit
  fflush(stdout);
  fflush(stderr);
  TC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  TC_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  for (hipEvent_t& event : c->done)
    TC_HIP(hipEventCreateWithFlags(&event, hipEventDisableTiming));
  TC_HIP(hipMalloc((void**)&c->token, sizeof(double)));
  TC_HIP(hipMemset(c->token, 0, sizeof(double)));
  *out = c.release();
  return TC_OK;
}

int tc_comm_destroy(tc_comm* c) {
  if (c == nullptr) return TC_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->token) (void)hipFree(c->token);
  if (c->ready) (void)hipEventDestroy(c->ready);
  for (hipEvent_t event : c->done)
    if (event) (void)hipEventDestroy(event);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TC_OK;
}

namespace {

// Gather on the communicator's stream once the work queued so far on `producer` is done,
// without blocking the host: later batches overlap with the transfer.
int gather_after(tc_comm* c, hipStream_t producer, const double* send_device,
                 double* recv_device, int64_t count, int root, int slot) {
  TC_CHECK(slot >= 0 && slot < 4, "slot must be in [0, 4)");
  TC_CHECK(c != nullptr && send_device != nullptr, "NULL argument");
  TC_CHECK(count >= 0 && root >= 0 && root < c->n_ranks, "invalid count or root");
  TC_CHECK(c->rank != root || recv_device != nullptr, "recv buffer is NULL on the root");
  TC_HIP(hipSetDevice(c->device));
  if (producer != nullptr) {
    TC_HIP(hipEventRecord(c->ready, producer));
    TC_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  }
  tc::host::range_push("rccl gather");
  const int status = g_rccl.Gather(send_device, recv_device, (size_t)count, kNcclFloat64,
                                   root, c->comm, c->stream);
  tc::host::range_pop();
  TC_RCCL(status);
  TC_HIP(hipEventRecord(c->done[slot], c->stream));
  return TC_OK;
}

}  // namespace

int tc_comm_gather(tc_comm* c, tc_table* t, const double* send_device,
                   double* recv_device, int64_t count, int root, int slot) {
  if (c != nullptr && t != nullptr && !t->chain) {
    // unordered finalisations (the default; tc_table_set_option "ordered"): the results of earlier
    // calls are not implied by the current lane's stream, wait for every lane's last one
    // (the events are recorded here, behind everything queued so far on each lane, not per
    // call: a marker after every finalisation costs the lane a bubble)
    TC_HIP(hipSetDevice(c->device));
    for (int l = 0; l < t->n_lanes; ++l) {
      if (l == t->cur) continue;
      TC_HIP(hipEventRecord(t->lanes[l].finished, t->lanes[l].stream));
      TC_HIP(hipStreamWaitEvent(c->stream, t->lanes[l].finished, 0));
    }
  }
  return gather_after(c, t != nullptr ? t->lanes[t->cur].stream : nullptr, send_device,
                      recv_device, count, root, slot);
}

int tc_comm_gather_interp(tc_comm* c, tc_interp* interp, const double* send_device,
                          double* recv_device, int64_t count, int root, int slot) {
  if (c != nullptr && interp != nullptr) {
    // results of earlier calls sit behind the interpolator's other lane
    TC_HIP(hipSetDevice(c->device));
    const int status = tc::host::interp_join_lanes(interp, c->stream);
    if (status != TC_OK) return status;
  }
  return gather_after(c, nullptr, send_device, recv_device, count, root, slot);
}

int tc_comm_release(tc_comm* c, tc_table* t, int slot) {
  TC_CHECK(c != nullptr && t != nullptr, "NULL argument");
  TC_CHECK(slot >= 0 && slot < 4, "slot must be in [0, 4)");
  TC_HIP(hipSetDevice(c->device));
  for (tc_table::Lane& lane : t->lanes)
    if (lane.stream) TC_HIP(hipStreamWaitEvent(lane.stream, c->done[slot], 0));
  return TC_OK;
}

int tc_comm_release_interp(tc_comm* c, tc_interp* interp, int slot) {
  TC_CHECK(c != nullptr && interp != nullptr, "NULL argument");
  TC_CHECK(slot >= 0 && slot < 4, "slot must be in [0, 4)");
  TC_HIP(hipSetDevice(c->device));
  return tc::host::interp_lanes_wait(interp, c->done[slot]);
}

int tc_comm_barrier(tc_comm* c) {
  TC_CHECK(c != nullptr, "comm handle is NULL");
  TC_HIP(hipSetDevice(c->device));
  TC_RCCL(g_rccl.AllReduce(c->token, c->token, 1, kNcclFloat64, kNcclSum, c->comm,
                           c->stream));
  TC_HIP(hipStreamSynchronize(c->stream));
  return TC_OK;
}

int tc_comm_synchronize(tc_comm* c) {
  TC_CHECK(c != nullptr, "comm handle is NULL");
  TC_HIP(hipSetDevice(c->device));
  TC_HIP(hipStreamSynchronize(c->stream));
  return TC_OK;
}

}  // extern "C"

