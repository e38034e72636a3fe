// libccz C ABI: the ONE exchange step of the row-sharded path as an RCCL collective.
//
// One process per GPU (the deployment north_star names):   rank 0: ccz_comm_unique_id -> ship the 128 bytes to the
// other ranks by any means (a file, an environment variable, a socket) -> every rank: ccz_comm_init_rank(h, id, world,
// rank) -> ccz_allreduce_sum_f64(h, packed_dev, count) on the handle's stream -> ccz_comm_destroy(h).
// One process driving several GPUs:   ccz_comm_init_all(handles, n) -> ccz_allreduce_sum_f64_multi(handles, bufs, n,
// count) (one grouped call for all devices).
//
// librccl is NOT a link-time dependency: it is dlopen'ed at the first communicator call (an already mapped copy --
// PyTorch ships its own librccl.so -- is reused), so a process that never shards never loads it and libccz.so keeps
// loading on a box without RCCL.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "hip_common.h"

using namespace ccz;

namespace {

// the few RCCL declarations this file needs (rccl.h: ncclUniqueId :43, ncclCommInitRank :220, ncclCommInitAll :236,
// ncclCommDestroy :260, ncclGetErrorString :339, ncclSum :448, ncclFloat64 :467, ncclAllReduce :611, ncclGroupStart/End :923/:933)
struct UniqueId { char internal[128]; };
typedef void* Comm;
constexpr int kSum = 0, kFloat64 = 8;

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommInitAll)(Comm*, int, const int*) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // CCZ_RCCL_LIB: the one library to use (a deployment that pins its RCCL; the tests force the not-found path with it)
    if (const char* forced = getenv("CCZ_RCCL_LIB")) {
      r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    } else {
      const char* names[] = {"librccl.so.1", "librccl.so"};
      for (const char* n : names)                                   // a copy that is already mapped (torch's) wins
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : paths)
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) {
      const char* e = dlerror();                                    // ONE call: dlerror() clears the message it returns
      r.why = std::string("librccl not found: ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(r.lib, name);
      if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + name;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

Rccl& need_rccl() {
  Rccl& r = rccl();
  if (!r.lib || !r.why.empty()) fail(CCZ_ERCCL, "%s", r.why.empty() ? "librccl unavailable" : r.why.c_str());
  return r;
}

void check(Rccl& r, int rc, const char* what) {
  if (rc != 0) fail(CCZ_ERCCL, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "rccl error");
}

}  // namespace

#define CCZ_GUARD(h, ...)                   \
  if (!(h)) return CCZ_EINVAL;              \
  try {                                     \
    ::ccz::DeviceScope ccz_scope_(h);       \
    __VA_ARGS__;                            \
    return CCZ_OK;                          \
  } catch (const ccz::Error& e) {           \
    (h)->err = e.msg;                       \
    return e.code;                          \
  } catch (...) {                           \
    (h)->err = "unknown internal error";    \
    return CCZ_EHIP;                        \
  }

extern "C" {

int ccz_comm_unique_id(ccz_handle h, void* id_out_128) {
  CCZ_GUARD(h, {
    if (!id_out_128) fail(CCZ_EINVAL, "null argument");
    Rccl& r = need_rccl();
    UniqueId id;
    check(r, r.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_out_128, id.internal, sizeof id.internal);
  })
}

int ccz_comm_init_rank(ccz_handle h, const void* id_128, int world, int rank) {
  CCZ_GUARD(h, {
    if (!id_128 || world < 1 || rank < 0 || rank >= world) fail(CCZ_EINVAL, "bad rank/world: %d/%d", rank, world);
    Impl* im = impl(h);
    if (im->comm) fail(CCZ_EINVAL, "the handle already has a communicator (ccz_comm_destroy first)");
    Rccl& r = need_rccl();
    UniqueId id;
    std::memcpy(id.internal, id_128, sizeof id.internal);
    Comm comm = nullptr;
    check(r, r.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    im->comm = comm; im->comm_world = world; im->comm_rank = rank;
  })
}

int ccz_comm_init_all(ccz_handle* handles, int n) {
  if (!handles || n < 1 || n > 64) return CCZ_EINVAL;
  for (int i = 0; i < n; ++i) if (!handles[i]) return CCZ_EINVAL;
  ccz_handle h = handles[0];
  CCZ_GUARD(h, {
    int devs[64];
    for (int i = 0; i < n; ++i) {
      if (impl(handles[i])->comm) fail(CCZ_EINVAL, "handle %d already has a communicator", i);
      devs[i] = handles[i]->device;
      for (int j = 0; j < i; ++j) if (devs[j] == devs[i]) fail(CCZ_EINVAL, "handles %d and %d share device %d", j, i, devs[i]);
    }
    Rccl& r = need_rccl();
    Comm comms[64];
    check(r, r.CommInitAll(comms, n, devs), "ncclCommInitAll");
    for (int i = 0; i < n; ++i) { Impl* im = impl(handles[i]); im->comm = comms[i]; im->comm_world = n; im->comm_rank = i; }
  })
}

int ccz_comm_info(ccz_handle h, int* world_out, int* rank_out) {
  CCZ_GUARD(h, {
    Impl* im = impl(h);
    if (world_out) *world_out = im->comm ? im->comm_world : 0;
    if (rank_out) *rank_out = im->comm ? im->comm_rank : -1;
  })
}

int ccz_comm_destroy(ccz_handle h) {
  CCZ_GUARD(h, {
    Impl* im = impl(h);
    if (im->comm) {
      CCZ_HIP(hipStreamSynchronize(stream(h)));
      Rccl& r = need_rccl();
      Comm c = im->comm;
      im->comm = nullptr; im->comm_world = 0; im->comm_rank = -1;
      check(r, r.CommDestroy(c), "ncclCommDestroy");
    }
  })
}

/* in place, on the handle's stream; enqueue-only (the result is ready in stream order) */
int ccz_allreduce_sum_f64(ccz_handle h, double* buf_dev, int64_t count) {
  CCZ_GUARD(h, {
    if (!buf_dev || count < 1) fail(CCZ_EINVAL, "bad argument");
    Impl* im = impl(h);
    if (!im->comm) fail(CCZ_EINVAL, "the handle has no communicator (ccz_comm_init_rank / ccz_comm_init_all)");
    Rccl& r = need_rccl();
    check(r, r.AllReduce(buf_dev, buf_dev, size_t(count), kFloat64, kSum, im->comm, stream(h)), "ncclAllReduce");
  })
}

/* The whole exchange step of a row-sharded fit: pack (blocks layout) into the handle's own buffer, the local row count into the
 * head's spare slot ON THE DEVICE, all-reduce head and tail on the handle's exchange stream, unpack the head on the handle's
 * stream and the tail behind the collective -- its completion is the event the next ccz_*_solve waits for right before its
 * first off-diagonal read.  One 8-byte host read (the global row count), no allocation after the first fit of a size. */
int ccz_moments_exchange(ccz_handle h, double* moments_dev, int64_t D, const int64_t* dims, int n_views, int64_t n_local,
                         int64_t* n_total_out) {
  CCZ_GUARD(h, {
    if (!moments_dev || !dims || !n_total_out || D < 1 || n_views < 1 || n_local < 0) fail(CCZ_EINVAL, "moments_exchange: bad argument");
    Impl* im = impl(h);
    if (!im->comm) fail(CCZ_EINVAL, "the handle has no communicator (ccz_comm_init_rank / ccz_comm_init_all)");
    Rccl& r = need_rccl();
    int64_t n_head = D + 1;
    for (int i = 0; i < n_views; ++i) {
      if (dims[i] < 1) fail(CCZ_EINVAL, "moments_exchange: view %d has no features", i);
      n_head += dims[i] * (dims[i] + 1) / 2;
    }
    const int64_t count = D * (D + 1) / 2 + D + 1, n_tail = count - n_head;
    if (n_tail < 0) fail(CCZ_EINVAL, "moments_exchange: dims do not sum to D");
    hipStream_t s0 = stream(h);
    if (!im->xchg_stream) {
      CCZ_HIP(hipStreamCreateWithFlags(&im->xchg_stream, hipStreamNonBlocking));
      for (auto& e : im->xchg_ev) CCZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t sx = im->xchg_stream;
    // the buffer may still be read by the previous fit's tail unpack (a caller that exchanged and never solved)
    CCZ_HIP(hipStreamWaitEvent(s0, im->xchg_ev[2], 0));
    if (im->xchg_cap < size_t(count) * 8) {
      CCZ_HIP(hipStreamSynchronize(sx));
      CCZ_HIP(hipStreamSynchronize(s0));
      if (im->xchg_buf) CCZ_HIP(hipFree(im->xchg_buf));
      im->xchg_buf = nullptr;
      im->xchg_cap = 0;
      CCZ_HIP(hipMalloc(&im->xchg_buf, size_t(count) * 8));
      im->xchg_cap = size_t(count) * 8;
    }
    double* packed = static_cast<double*>(im->xchg_buf);
    moments_blocks(h, true, moments_dev, D, dims, n_views, packed, 3, nullptr);
    fill2d(h, 1, 1, packed + n_head - 1, 1, double(n_local));
    CCZ_HIP(hipEventRecord(im->xchg_ev[0], s0));
    CCZ_HIP(hipStreamWaitEvent(sx, im->xchg_ev[0], 0));
    check(r, r.AllReduce(packed, packed, size_t(n_head), kFloat64, kSum, im->comm, sx), "ncclAllReduce (head)");
    CCZ_HIP(hipEventRecord(im->xchg_ev[1], sx));
    if (n_tail > 0) {
      check(r, r.AllReduce(packed + n_head, packed + n_head, size_t(n_tail), kFloat64, kSum, im->comm, sx), "ncclAllReduce (tail)");
      moments_blocks(h, false, moments_dev, D, dims, n_views, packed, 2, sx);
    }
    CCZ_HIP(hipEventRecord(im->xchg_ev[2], sx));
    CCZ_HIP(hipStreamWaitEvent(s0, im->xchg_ev[1], 0));
    moments_blocks(h, false, moments_dev, D, dims, n_views, packed, 1, nullptr);
    double nt = 0.0;
    d2h(h, &nt, packed + n_head - 1, 8);                    // waits for the head only; the tail is still in flight
    *n_total_out = int64_t(nt + 0.5);
    if (n_tail > 0) im->deferred_event = im->xchg_ev[2];    // consumed by the next solve (ops_hip.hip: wait_deferred)
  })
}

int ccz_allreduce_sum_f64_multi(ccz_handle* handles, double* const* bufs_dev, int n, int64_t count) {
  if (!handles || !bufs_dev || n < 1 || count < 1) return CCZ_EINVAL;
  for (int i = 0; i < n; ++i) if (!handles[i] || !bufs_dev[i]) return CCZ_EINVAL;
  ccz_handle h = handles[0];
  CCZ_GUARD(h, {
    for (int i = 0; i < n; ++i) if (!impl(handles[i])->comm) fail(CCZ_EINVAL, "handle %d has no communicator", i);
    Rccl& r = need_rccl();
    check(r, r.GroupStart(), "ncclGroupStart");
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i)
      rc = r.AllReduce(bufs_dev[i], bufs_dev[i], size_t(count), kFloat64, kSum, impl(handles[i])->comm, stream(handles[i]));
    const int rc2 = r.GroupEnd();
    check(r, rc, "ncclAllReduce");
    check(r, rc2, "ncclGroupEnd");
  })
}

}  // extern "C"
