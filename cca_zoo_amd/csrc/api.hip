// libccz C ABI: lifecycle, memory, moments, DCCA loss, transform (HIP build).
// The solver entry points live in solve.cpp.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "hip_common.h"

using namespace ccz;

namespace ccz {
// loss.hip
void cca_loss_impl(ccz_ctx* c, int dtype, const void* z1, const void* z2, int64_t n, int64_t d1, int64_t d2, int64_t ld1,
                   int64_t ld2, double eps, void* loss_dev, void* g1, void* g2, int64_t ldg1, int64_t ldg2);
void pair_loss_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, double eps, void* loss_dev, void* const* g,
                    const int64_t* ldg);
int64_t pair_loss_state_bytes_impl(int dtype, const int64_t* dims, int m);
void pair_loss_forward_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, double eps, void* loss_dev, void* state);
void pair_loss_backward_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, const void* state, const void* grad_out,
                             void* const* g, const int64_t* ldg);
void cca_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, int64_t d1, int64_t d2, double eps, double* loss_host,
                           double* gamma_dev, double* mean_dev);
void pair_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps,
                            double* loss_host, double* gamma_dev, double* mean_dev);
void loss_status_take(ccz_ctx* c, bool synchronise, int* view, int* pivot);
void randn_fill_impl(ccz_ctx* c, int dtype, void* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                     int64_t row0, int64_t row_stride, double scale, bool accumulate);
}  // namespace ccz

#define CCZ_GUARD(h, ...)                   \
  if (!(h)) return CCZ_EINVAL;              \
  try {                                     \
    ::ccz::DeviceScope ccz_scope_(h);                   \
    __VA_ARGS__;                            \
    return CCZ_OK;                          \
  } catch (const ccz::Error& e) {           \
    (h)->err = e.msg;                       \
    return e.code;                          \
  } catch (const std::bad_alloc&) {         \
    (h)->err = "host allocation failed";    \
    return CCZ_ENOMEM;                      \
  } catch (...) {                           \
    (h)->err = "unknown internal error";    \
    return CCZ_EHIP;                        \
  }

namespace ccz {

// out = (X - mean) W     reference: cca_zoo/_base.py:108-123
static void transform_impl(ccz_ctx* c, int dtype, const void* X, int64_t n, int64_t d, int64_t ld, const double* mean,
                           const double* W, int64_t k, void* out, int64_t ldo) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "transform: dtype must be CCZ_F32 or CCZ_F64");
  if (!X || !W || !out || n < 1 || d < 1 || k < 1 || ld < d || ldo < k) fail(CCZ_EINVAL, "transform: bad argument");
  if (dtype == CCZ_F32 && project_split_eligible(c, n, d, k, ld, X, ldo)) {
    // large fp32 projections: the split arithmetic of K1 with the conversion in registers (project_split.hip) -- HBM-bound
    project_split(c, static_cast<const float*>(X), n, d, ld, mean, W, k, static_cast<float*>(out), ldo);
    return;
  }
  DBuf bias(c, k);
  if (mean) gemm(c, false, false, 1, k, d, 1.0, mean, d, W, k, 0.0, bias, k);
  gemm_mixed(c, dtype, n, k, d, 1.0, X, ld, W, k, 0.0, out, ldo, mean ? bias.get() : nullptr);
  // enqueue-only: `out` is ready in stream order (ccz_sync / ccz_stream_release / ccz_memcpy_d2h for a host consumer)
}

}  // namespace ccz

extern "C" {

int ccz_version(void) { return CCZ_VERSION; }

int ccz_create(ccz_handle* out, int device) {
  if (!out) return CCZ_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return CCZ_EHIP; }
  if (device < 0 || device >= count) return CCZ_EINVAL;
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return CCZ_EHIP; }
  ccz_ctx* c = new (std::nothrow) ccz_ctx();
  Impl* im = new (std::nothrow) Impl();
  if (!c || !im) { delete c; delete im; return CCZ_ENOMEM; }
  c->device = device;
  c->impl = im;
  bool ok = hipGetDeviceProperties(&im->props, device) == hipSuccess;
  for (int i = 0; ok && i < 4; ++i) ok = hipEventCreate(&im->ev[i]) == hipSuccess;
  // a real (blocking) stream instead of the legacy null stream: it keeps the implicit ordering with
  // null-stream work (PyTorch's default stream) and, unlike the null stream, can be captured into graphs
  ok = ok && hipStreamCreate(&im->own_stream) == hipSuccess;
  if (ok) c->stream = im->own_stream;
  if (const char* e = getenv("CCZ_GRAPHS")) im->graphs_on = atoi(e);
  ok = ok && hipMalloc(reinterpret_cast<void**>(&im->d_flag), 64 * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&im->d_small), im->small_cap * sizeof(double)) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); delete im; delete c; return CCZ_EHIP; }
  *out = c;
  return CCZ_OK;
}

int ccz_destroy(ccz_handle h) {
  if (!h) return CCZ_OK;
  Impl* im = impl(h);
  if (im) {
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (im->comm) (void)ccz_comm_destroy(h);
    for (auto& g : im->graphs) (void)hipGraphExecDestroy(g.exec);
    if (im->own_stream) (void)hipStreamDestroy(im->own_stream);
    for (auto& b : im->pool) (void)hipFree(b.p);
    for (auto& t : im->tile_tabs) if (t.dev) (void)hipFree(t.dev);
    for (auto& e : im->chain_sync) if (e.second) (void)hipFree(e.second);
    for (auto& e : im->colsum_sync) if (e.second) (void)hipFree(e.second);
    for (auto& e : im->k1_plans) if (e.dev) (void)hipFree(e.dev);
    if (im->chain_dbg) (void)hipFree(im->chain_dbg);
    if (im->xchg_buf) (void)hipFree(im->xchg_buf);
    if (im->xchg_stream) (void)hipStreamDestroy(im->xchg_stream);
    for (auto& e : im->xchg_ev) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(im->ev[i]);
    for (int i = 0; i < 4; ++i) if (im->pipe_ev[i]) (void)hipEventDestroy(im->pipe_ev[i]);
    for (auto& e : im->sp_ev) if (e) (void)hipEventDestroy(e);
    if (im->split_stream) (void)hipStreamDestroy(im->split_stream);
    for (auto& e : im->split_tabs) { (void)hipFree(e.panels); (void)hipFree(e.tiles); (void)hipFree(e.gtiles); }
    for (int i = 0; i < 2; ++i) if (im->pin_buf[i]) (void)hipHostFree(im->pin_buf[i]);
    for (int i = 0; i < Impl::kSmallSlots; ++i) {
      if (im->small_ev[i]) (void)hipEventDestroy(im->small_ev[i]);
      if (im->small_pin[i]) (void)hipHostFree(im->small_pin[i]);
    }
    if (im->copy_stream) (void)hipStreamDestroy(im->copy_stream);
    for (int i = 0; i < 2; ++i) if (im->aux_ev[i]) (void)hipEventDestroy(im->aux_ev[i]);
    for (auto& e : im->bj_ev) if (e) (void)hipEventDestroy(e);
    if (im->aux_stream) (void)hipStreamDestroy(im->aux_stream);
    for (int i = 0; i < 2; ++i) if (im->xs_ev[i]) (void)hipEventDestroy(im->xs_ev[i]);
    if (im->loss_status) (void)hipHostFree(im->loss_status);
    if (im->wait_ev) (void)hipEventDestroy(im->wait_ev);
    if (im->defer_own_ev) (void)hipEventDestroy(im->defer_own_ev);
    if (im->d2h_pin) (void)hipHostFree(im->d2h_pin);
    for (hipEvent_t& e : im->d2h_tev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(im->d_flag);
    (void)hipFree(im->d_small);
    delete im;
  }
  delete h;
  return CCZ_OK;
}

const char* ccz_last_error(ccz_handle h) { return h ? h->err.c_str() : "null handle"; }

int ccz_set_stream(ccz_handle h, void* s) {
  CCZ_GUARD(h, {
    CCZ_HIP(hipStreamSynchronize(stream(h)));
    h->stream = s ? s : static_cast<void*>(impl(h)->own_stream);       // NULL: back to the handle's own stream
    impl(h)->adopted = false;
  })
}

int ccz_sync(ccz_handle h) { CCZ_GUARD(h, sync(h)) }

// Hand-over between a caller's stream and the handle's stream WITHOUT blocking the host.  The handle's own stream is
// a blocking stream: it is already ordered with the legacy null stream (PyTorch's default stream) in both directions,
// so nothing is enqueued between those two; any other pair is joined through an event.  A stream that has been
// destroyed since (a torch side stream) has no pending work: the failed record is ignored.
static void stream_join(ccz_ctx* c, hipStream_t from, hipStream_t to, int slot) {
  if (from == to) return;
  Impl* im = impl(c);
  const bool implicit = (from == im->own_stream && to == nullptr) || (from == nullptr && to == im->own_stream);
  if (implicit) return;                                                // legacy null-stream ordering
  if (!im->xs_ev[slot]) CCZ_HIP(hipEventCreateWithFlags(&im->xs_ev[slot], hipEventDisableTiming));
  const hipError_t rec = hipEventRecord(im->xs_ev[slot], from);
  if (rec != hipSuccess) {
    (void)hipGetLastError();
    // a DESTROYED source stream has nothing pending: no dependency to establish.  Anything else (a capturing stream, a
    // caller's bug) must not silently drop the ordering -- fall back to a full wait
    if (rec == hipErrorInvalidHandle || rec == hipErrorInvalidResourceHandle || rec == hipErrorContextIsDestroyed) return;
    CCZ_HIP(hipDeviceSynchronize());
    return;
  }
  CCZ_HIP(hipStreamWaitEvent(to, im->xs_ev[slot], 0));
}

// the handle goes back to its own stream (after ccz_stream_adopt), ordered after what it enqueued on the adopted one
static void stream_home(ccz_ctx* c) {
  Impl* im = impl(c);
  if (stream(c) == im->own_stream || !im->adopted) return;
  stream_join(c, stream(c), im->own_stream, 0);
  c->stream = im->own_stream;
  im->adopted = false;
}

int ccz_stream_acquire(ccz_handle h, void* ext) {
  CCZ_GUARD(h, {
    stream_home(h);
    stream_join(h, static_cast<hipStream_t>(ext), stream(h), 0);
  })
}

int ccz_stream_release(ccz_handle h, void* ext) {
  CCZ_GUARD(h, {
    if (!(impl(h)->adopted && stream(h) == static_cast<hipStream_t>(ext)))     // adopted: the work already sits in `ext`
      stream_join(h, stream(h), static_cast<hipStream_t>(ext), 1);
  })
}

int ccz_stream_adopt(ccz_handle h, void* ext) {
  CCZ_GUARD(h, {
    hipStream_t e = static_cast<hipStream_t>(ext);
    if (stream(h) != e) {
      stream_join(h, stream(h), e, 0);
      h->stream = e;
    }
    impl(h)->adopted = true;
  })
}

int ccz_loss_status(ccz_handle h, int synchronise, int* view, int* pivot) {
  CCZ_GUARD(h, ccz::loss_status_take(h, synchronise != 0, view, pivot))
}

int ccz_device_info(ccz_handle h, ccz_devinfo* out) {
  CCZ_GUARD(h, {
    if (!out) fail(CCZ_EINVAL, "null argument");
    Impl* im = impl(h);
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->name, im->props.name, sizeof(out->name) - 1);
    std::strncpy(out->arch, im->props.gcnArchName, sizeof(out->arch) - 1);
    out->compute_units = im->props.multiProcessorCount;
    out->wavefront = im->props.warpSize;
    out->hbm_bytes = int64_t(im->props.totalGlobalMem);
    out->lds_bytes_per_cu = int64_t(im->props.maxSharedMemoryPerMultiProcessor);
  })
}

int ccz_dev_alloc(ccz_handle h, void** out, size_t bytes) {
  CCZ_GUARD(h, {
    if (!out) fail(CCZ_EINVAL, "null argument");
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
    if (e != hipSuccess) { (void)hipGetLastError(); fail(CCZ_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    *out = p;
  })
}

int ccz_dev_free(ccz_handle h, void* p) {
  CCZ_GUARD(h, {
    if (p) {
      CCZ_HIP(hipStreamSynchronize(stream(h)));
      CCZ_HIP(hipFree(p));
    }
  })
}

int ccz_memcpy_h2d(ccz_handle h, void* dst, const void* src, size_t bytes) { CCZ_GUARD(h, h2d(h, dst, src, bytes)) }
int ccz_memcpy_d2h(ccz_handle h, void* dst, const void* src, size_t bytes) { CCZ_GUARD(h, d2h(h, dst, src, bytes)) }
int ccz_memset0(ccz_handle h, void* dst, size_t bytes) { CCZ_GUARD(h, zero(h, dst, bytes)) }

int ccz_moments(ccz_handle h, int dtype, const ccz_view* views, int n_views, int64_t n_rows, int views_on_device,
                double* moments_dev, int accumulate) {
  CCZ_GUARD(h, {
    moments_impl(h, dtype, views, n_views, n_rows, views_on_device != 0, moments_dev, accumulate != 0);
  })
}

int ccz_moments_opts(ccz_handle h, int dtype, const ccz_view* views, int n_views, int64_t n_rows, int views_on_device,
                     double* moments_dev, int accumulate, int pilot_mode, int timed) {
  CCZ_GUARD(h, {
    if (pilot_mode < 0 || pilot_mode > 2) fail(CCZ_EINVAL, "moments: pilot_mode must be 0 (never), 1 (automatic) or 2 (always)");
    moments_impl(h, dtype, views, n_views, n_rows, views_on_device != 0, moments_dev, accumulate != 0, pilot_mode, timed != 0);
  })
}

int ccz_moments_symmetrize(ccz_handle h, double* moments_dev, int64_t D) {
  CCZ_GUARD(h, {
    if (!moments_dev || D < 1) fail(CCZ_EINVAL, "bad argument");
    mirror_upper(h, D, moments_dev, D);
  })
}

int ccz_moments_pack(ccz_handle h, const double* moments_dev, int64_t D, double* packed_dev) {
  CCZ_GUARD(h, {
    if (!moments_dev || !packed_dev || D < 1) fail(CCZ_EINVAL, "bad argument");
    pack_upper(h, D, moments_dev, D, packed_dev);
    d2d(h, packed_dev + D * (D + 1) / 2, moments_dev + D * D, size_t(D) * 8);
  })
}

int ccz_moments_unpack(ccz_handle h, const double* packed_dev, int64_t D, double* moments_dev) {
  CCZ_GUARD(h, {
    if (!moments_dev || !packed_dev || D < 1) fail(CCZ_EINVAL, "bad argument");
    unpack_upper(h, D, packed_dev, moments_dev, D);
    d2d(h, moments_dev + D * D, packed_dev + D * (D + 1) / 2, size_t(D) * 8);
  })
}

// blocks layout of the sharded exchange:  [ upper triangles of the diagonal blocks C_11 .. C_mm | colsum (D) | 1 spare
// slot (the caller's row count) ]  then  [ the off-diagonal blocks (i < j), each d_i x d_j row-major ]
}  // extern "C"
namespace ccz {
void moments_blocks(ccz_ctx* c, bool pack, double* mom, int64_t D, const int64_t* dims, int m, double* packed, int which,
                    void* on_stream) {
  if (!mom || !packed || !dims || m < 1 || D < 1) fail(CCZ_EINVAL, "moments blocks: bad argument");
  if (which < 1 || which > 3) fail(CCZ_EINVAL, "moments blocks: which must be 1 (head), 2 (tail) or 3 (both)");
  std::vector<int64_t> off(m + 1, 0);
  for (int i = 0; i < m; ++i) {
    if (dims[i] < 1) fail(CCZ_EINVAL, "moments blocks: view %d has no features", i);
    off[i + 1] = off[i] + dims[i];
  }
  if (off[m] != D) fail(CCZ_EINVAL, "moments blocks: dims do not sum to D");
  void* const prev = c->stream;
  if (on_stream) c->stream = on_stream;              // plain kernels and copies only: no pooled scratch is involved
  try {
    int64_t pos = 0;
    for (int i = 0; i < m; ++i) {
      double* blk = mom + off[i] * D + off[i];
      if (which & 1) { if (pack) pack_upper(c, dims[i], blk, D, packed + pos); else unpack_upper(c, dims[i], packed + pos, blk, D); }
      pos += dims[i] * (dims[i] + 1) / 2;
    }
    if (which & 1) { if (pack) d2d(c, packed + pos, mom + D * D, size_t(D) * 8); else d2d(c, mom + D * D, packed + pos, size_t(D) * 8); }
    pos += D + 1;
    for (int i = 0; i < m; ++i)
      for (int j = i + 1; j < m; ++j) {
        double* blk = mom + off[i] * D + off[j];
        if (which & 2) { if (pack) copy2d(c, dims[i], dims[j], blk, D, packed + pos, dims[j]); else copy2d(c, dims[i], dims[j], packed + pos, dims[j], blk, D); }
        pos += dims[i] * dims[j];
      }
  } catch (...) {
    c->stream = prev;
    throw;
  }
  c->stream = prev;
}
}  // namespace ccz
extern "C" {

int ccz_moments_pack_blocks(ccz_handle h, const double* moments_dev, int64_t D, const int64_t* dims, int n_views, double* packed_dev,
                            int which) {
  CCZ_GUARD(h, moments_blocks(h, true, const_cast<double*>(moments_dev), D, dims, n_views, packed_dev, which, nullptr))
}

int ccz_moments_unpack_blocks(ccz_handle h, const double* packed_dev, int64_t D, const int64_t* dims, int n_views, double* moments_dev,
                              int which, void* on_stream) {
  CCZ_GUARD(h, {
    moments_blocks(h, false, moments_dev, D, dims, n_views, const_cast<double*>(packed_dev), which, on_stream);
    if (on_stream && on_stream != h->stream) {
      // an unpack on a FOREIGN stream is by construction the deferred half of an exchange: record an event the handle
      // OWNS behind it and make the next solve wait for it on the device.  (A borrowed event -- ccz_solve_defer --
      // can be destroyed by its owner while still registered; this one lives as long as the handle.)
      Impl* im = impl(h);
      if (!im->defer_own_ev) CCZ_HIP(hipEventCreateWithFlags(&im->defer_own_ev, hipEventDisableTiming));
      CCZ_HIP(hipEventRecord(im->defer_own_ev, static_cast<hipStream_t>(on_stream)));
      im->deferred_event = im->defer_own_ev;
    }
  })
}

int ccz_solve_defer(ccz_handle h, void* event) {
  CCZ_GUARD(h, {
    // event == NULL: a pending deferral (ccz_moments_exchange's tail, a foreign-stream unpack) is awaited NOW on the handle's
    // stream -- a device-side wait, the host does not block -- for callers that go on to read off-diagonal blocks through
    // entry points other than the solves (score, grid search, the partial / group estimators)
    if (event) impl(h)->deferred_event = event;
    else wait_deferred(h);
  })
}

int ccz_moments_last_ms(ccz_handle h, double* gram_ms, double* colsum_ms) {
  CCZ_GUARD(h, {
    if (gram_ms) *gram_ms = h->last_gram_ms;
    if (colsum_ms) *colsum_ms = h->last_colsum_ms;
  })
}

int ccz_pool_trim(ccz_handle h, size_t* released_bytes) {
  CCZ_GUARD(h, {
    Impl* im = impl(h);
    CCZ_HIP(hipStreamSynchronize(stream(h)));       // pooled blocks are recycled in stream order: nothing may still use them
    size_t freed = 0;
    for (auto it = im->pool.begin(); it != im->pool.end();) {
      if (!it->used) { freed += it->bytes; (void)hipFree(it->p); it = im->pool.erase(it); } else ++it;
    }
    if (released_bytes) *released_bytes = freed;
  })
}

int ccz_k1_route(ccz_handle h, int route, int* previous) {
  CCZ_GUARD(h, {
    if (route < -1 || route > CCZ_K1_BF16X2) fail(CCZ_EINVAL, "k1_route: route must be CCZ_K1_AUTO, CCZ_K1_FP32 or CCZ_K1_BF16X2 (or -1 to query)");
    if (previous) *previous = h->k1_route;
    if (route >= 0) h->k1_route = route;
  })
}

int ccz_moments_last_route(ccz_handle h, int* route, double* split_ms, double* mfma_ms, double* reduce_ms) {
  CCZ_GUARD(h, {
    if (route) *route = h->last_route;
    if (split_ms) *split_ms = h->last_split_ms;
    if (mfma_ms) *mfma_ms = h->last_mfma_ms;
    if (reduce_ms) *reduce_ms = h->last_reduce_ms;
  })
}

int ccz_loss_last_route(ccz_handle h, int* forward_route, int* backward_route) {
  CCZ_GUARD(h, {
    if (forward_route) *forward_route = h->last_route;
    if (backward_route) *backward_route = h->last_bwd_route;
  })
}

int ccz_moments_last_pilot(ccz_handle h, int* used) {
  CCZ_GUARD(h, {
    if (used) *used = h->last_pilot;
  })
}

int ccz_cca_loss(ccz_handle h, int dtype, const void* z1_dev, const void* z2_dev, int64_t n, int64_t d1, int64_t d2,
                 int64_t ld1, int64_t ld2, double eps, void* loss_dev, void* g1_dev, void* g2_dev, int64_t ldg1,
                 int64_t ldg2) {
  CCZ_GUARD(h, {
    cca_loss_impl(h, dtype, z1_dev, z2_dev, n, d1, d2, ld1, ld2, eps, loss_dev, g1_dev, g2_dev, ldg1, ldg2);
  })
}

int ccz_pair_loss(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n, double eps, void* loss_dev,
                  void* const* g_dev, const int64_t* ldg) {
  CCZ_GUARD(h, ccz::pair_loss_impl(h, dtype, z_dev, n_views, n, eps, loss_dev, g_dev, ldg));
}

int64_t ccz_pair_loss_state_bytes(int dtype, const int64_t* dims, int n_views) {
  return ccz::pair_loss_state_bytes_impl(dtype, dims, n_views);
}

int ccz_pair_loss_forward(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n, double eps, void* loss_dev,
                          void* state_dev) {
  CCZ_GUARD(h, ccz::pair_loss_forward_impl(h, dtype, z_dev, n_views, n, eps, loss_dev, state_dev));
}

int ccz_pair_loss_backward(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n, const void* state_dev,
                           const void* grad_out_dev, void* const* g_dev, const int64_t* ldg) {
  CCZ_GUARD(h, ccz::pair_loss_backward_impl(h, dtype, z_dev, n_views, n, state_dev, grad_out_dev, g_dev, ldg));
}

int ccz_cca_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, int64_t d1, int64_t d2, double eps,
                         double* loss_host, double* gamma_dev, double* mean_dev) {
  CCZ_GUARD(h, ccz::cca_loss_moments_impl(h, moments_dev, n_rows, d1, d2, eps, loss_host, gamma_dev, mean_dev));
}

int ccz_pair_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, const int64_t* dims, int n_views, double eps,
                          double* loss_host, double* gamma_dev, double* mean_dev) {
  CCZ_GUARD(h, ccz::pair_loss_moments_impl(h, moments_dev, n_rows, dims, n_views, eps, loss_host, gamma_dev, mean_dev));
}

int ccz_randn_fill(ccz_handle h, int dtype, void* out_dev, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, int64_t row0,
                   int64_t row_stride, double scale, int accumulate) {
  CCZ_GUARD(h, ccz::randn_fill_impl(h, dtype, out_dev, rows, cols, ld, seed, row0, row_stride, scale, accumulate != 0));
}

int ccz_cholinv(ccz_handle h, int count, double* const* A_dev, const int64_t* d, double* const* L_dev, double* const* X_dev) {
  CCZ_GUARD(h, {
    if (count < 1 || count > 8 || !A_dev || !d || !L_dev) fail(CCZ_EINVAL, "cholinv: 1..8 matrices, non-null arrays");
    std::vector<DBuf> T(count);
    std::vector<double*> Tp(count);
    for (int b = 0; b < count; ++b) {
      if (d[b] < 1 || !A_dev[b] || !L_dev[b] || (X_dev && !X_dev[b])) fail(CCZ_EINVAL, "cholinv: bad matrix %d", b);
      T[b] = DBuf(h, (d[b] + 63) / 64 * 4096);
      Tp[b] = T[b].get();
    }
    int* info_dev = static_cast<int*>(dev_alloc(h, 8 * sizeof(int)));
    int got[8];
    try {
      cholinv_batched(h, count, A_dev, d, d, L_dev, d, X_dev, d, Tp.data(), info_dev);
      d2h(h, got, info_dev, size_t(count) * sizeof(int));
    } catch (...) {
      dev_free(h, info_dev);
      throw;
    }
    dev_free(h, info_dev);
    for (int b = 0; b < count; ++b)
      if (got[b] != 0x7fffffff) fail(CCZ_ENOTSPD, "cholinv: matrix %d is not positive definite (pivot %d)", b, got[b] - 1);
  })
}

int ccz_transform(ccz_handle h, int dtype, const void* X_dev, int64_t n, int64_t d, int64_t ld, const double* mean_dev,
                  const double* W_dev, int64_t k, void* out_dev, int64_t ldo) {
  CCZ_GUARD(h, {
    transform_impl(h, dtype, X_dev, n, d, ld, mean_dev, W_dev, k, out_dev, ldo);
  })
}

}  // extern "C"
