// libccz C ABI: lifecycle, memory, moments, DCCA loss, transform (HIP build).
// The solver entry points live in solve.cpp.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "hip_common.h"

using namespace ccz;

#define CCZ_GUARD(h, ...)                   \
  if (!(h)) return CCZ_EINVAL;              \
  try {                                     \
    ::ccz::activate(h);                     \
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

// ---------------------------------------------------------------------------
// DCCA correlation loss: value + closed-form input gradients
// reference: cca_zoo/deep/objectives.py:61-102 (forward) + autograd backward;
// maths: oracle/losses.py::cca_loss_closed_form
// ---------------------------------------------------------------------------
// Everything between the batch moments and the sample-side GEMMs: loss value and the four d x d gradient
// matrices (dz1 = ((z1 - mu1) G11s + (z2 - mu2) G12') / (n-1), dz2 = ((z1 - mu1) G12 + (z2 - mu2) G22s) / (n-1)).
struct LossCore {
  DBuf G11s, G22s, G12, G12t, mu;
  double loss = 0.0;
};

static LossCore cca_loss_core(ccz_ctx* c, const double* G, const double* s, int64_t n, int64_t d1, int64_t d2, double eps,
                              bool want1, bool want2) {
  const int64_t D = d1 + d2;
  const double inv = 1.0 / double(n - 1);
  LossCore out;
  DBuf L1(c, d1 * d1), L2(c, d2 * d2), S12(c, d1 * d2);
  cov_block(c, G, D, s, n, true, inv, 0, d1, 0, d1, L1, d1);
  add_diag(c, d1, L1, d1, eps);
  cov_block(c, G, D, s, n, true, inv, d1, d2, d1, d2, L2, d2);
  add_diag(c, d2, L2, d2, eps);
  cov_block(c, G, D, s, n, true, inv, 0, d1, d1, d2, S12, d2);
  {
    double* Lp[2] = {L1.get(), L2.get()};
    const int64_t dd[2] = {d1, d2};
    int info[2] = {0, 0};
    potrf_lower_batched(c, 2, Lp, dd, dd, info);
    if (info[0] != 0) fail(CCZ_ENOTSPD, "cca_loss: S11 + eps I is not positive definite");
    if (info[1] != 0) fail(CCZ_ENOTSPD, "cca_loss: S22 + eps I is not positive definite");
  }

  // Explicit triangular inverses Li = L^-1 (one blocked TRSM on the identity per factor): every
  // S^-1 product below is then two MFMA GEMMs instead of two blocked triangular solves -- the loss is
  // launch-latency bound at DCCA batch shapes (d ~ 512), and S + eps I keeps this well conditioned in fp64.
  auto tri_inverse = [&](const double* L, int64_t d) {
    DBuf Li(c, d * d);
    fill2d(c, d, d, Li, d, 0.0);
    add_diag(c, d, Li, d, 1.0);
    trsm_right_lower(c, false, d, d, L, d, Li, d);            // I L^-1
    return Li;
  };
  DBuf Li1 = tri_inverse(L1, d1), Li2 = tri_inverse(L2, d2);
  // left  solve: S^-1 M = Li' (Li M)      right solve: M S^-1 = (M Li') Li
  auto solve_left = [&](const double* Li, int64_t d, bool transM, const double* M, int64_t ldm, int64_t r, double alpha, double* o) {
    DBuf t(c, d * r);
    gemm(c, false, transM, d, r, d, 1.0, Li, d, M, ldm, 0.0, t, r);
    gemm(c, true, false, d, r, d, alpha, Li, d, t, r, 0.0, o, r);
  };
  auto solve_right = [&](const double* Li, int64_t d, const double* M, int64_t ldm, int64_t r, double alpha, double* o) {
    DBuf t(c, r * d);
    gemm(c, false, true, r, d, d, 1.0, M, ldm, Li, d, 0.0, t, d);
    gemm(c, false, false, r, d, d, alpha, t, d, Li, d, 0.0, o, d);
  };

  DBuf A(c, d1 * d2), Bmt(c, d1 * d2);
  solve_left(Li1, d1, false, S12, d2, d2, 1.0, A);            // A   = S11^-1 S12           (d1 x d2)
  solve_right(Li2, d2, S12, d2, d1, 1.0, Bmt);                // Bm' = S12 S22^-1           (d1 x d2)
  DBuf rd(c, d1);
  row_dots(c, d1, d2, A, d2, Bmt, d2, rd);                    // tr(A Bm) = sum A o Bm'
  std::vector<double> rh(d1);
  d2h(c, rh.data(), rd, size_t(d1) * 8);
  for (double v : rh) out.loss -= v;
  if (!want1 && !want2) return out;

  // G12 = -2 S11^-1 S12 S22^-1 = -2 A S22^-1 (d1 x d2) and its transpose
  out.G12 = DBuf(c, d1 * d2);
  out.G12t = DBuf(c, d2 * d1);
  solve_right(Li2, d2, A, d2, d1, -2.0, out.G12);
  transpose(c, d1, d2, out.G12, d2, out.G12t, d1);
  out.mu = DBuf(c, D);
  d2d(c, out.mu, s, size_t(D) * 8);
  axpby2d(c, 1, D, 1.0 / double(n), out.mu, D, 0.0, nullptr, 0);
  if (want1) {
    // G11 = A Bm S11^-1 = S11^-1 (S12 S22^-1 S21) S11^-1 is symmetric, so G11 + G11' = 2 G11
    DBuf P(c, d1 * d1);
    out.G11s = DBuf(c, d1 * d1);
    gemm(c, false, true, d1, d1, d2, 1.0, A, d2, Bmt, d2, 0.0, P, d1);          // A Bm
    solve_right(Li1, d1, P, d1, d1, 2.0, out.G11s);
  }
  if (want2) {
    DBuf P(c, d2 * d2);
    out.G22s = DBuf(c, d2 * d2);
    gemm(c, true, false, d2, d2, d1, 1.0, Bmt, d2, A, d2, 0.0, P, d2);          // Bm A
    solve_right(Li2, d2, P, d2, d2, 2.0, out.G22s);
  }
  return out;
}

static void cca_loss_impl(ccz_ctx* c, int dtype, const void* z1, const void* z2, int64_t n, int64_t d1, int64_t d2,
                          int64_t ld1, int64_t ld2, double eps, void* loss_dev, void* g1, void* g2, int64_t ldg1,
                          int64_t ldg2) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "cca_loss: dtype must be CCZ_F32 or CCZ_F64");
  if (!z1 || !z2 || !loss_dev) fail(CCZ_EINVAL, "cca_loss: null argument");
  if (n < 2 || d1 < 1 || d2 < 1 || ld1 < d1 || ld2 < d2) fail(CCZ_EINVAL, "cca_loss: bad shape");
  if ((g1 && ldg1 < d1) || (g2 && ldg2 < d2)) fail(CCZ_EINVAL, "cca_loss: bad gradient stride");
  const int64_t D = d1 + d2;
  DBuf mom(c, D * D + D);
  ccz_view views[2] = {{z1, d1, ld1}, {z2, d2, ld2}};
  moments_impl(c, dtype, views, 2, n, true, mom, false);
  const double inv = 1.0 / double(n - 1);
  LossCore k = cca_loss_core(c, mom, mom.get() + D * D, n, d1, d2, eps, g1 != nullptr, g2 != nullptr);
  if (dtype == CCZ_F32) { const float lf = float(k.loss); h2d(c, loss_dev, &lf, 4); }
  else h2d(c, loss_dev, &k.loss, 8);
  if (!g1 && !g2) return;
  if (g1) {
    DBuf bias(c, d1);
    gemm(c, false, false, 1, d1, d1, 1.0, k.mu, D, k.G11s, d1, 0.0, bias, d1);
    gemm(c, false, false, 1, d1, d2, 1.0, k.mu.get() + d1, D, k.G12t, d1, 1.0, bias, d1);
    gemm_mixed(c, dtype, n, d1, d1, inv, z1, ld1, k.G11s, d1, 0.0, g1, ldg1, bias);
    gemm_mixed(c, dtype, n, d1, d2, inv, z2, ld2, k.G12t, d1, 1.0, g1, ldg1, nullptr);
  }
  if (g2) {
    DBuf bias(c, d2);
    gemm(c, false, false, 1, d2, d2, 1.0, k.mu.get() + d1, D, k.G22s, d2, 0.0, bias, d2);
    gemm(c, false, false, 1, d2, d1, 1.0, k.mu, D, k.G12, d2, 1.0, bias, d2);
    gemm_mixed(c, dtype, n, d2, d2, inv, z2, ld2, k.G22s, d2, 0.0, g2, ldg2, bias);
    gemm_mixed(c, dtype, n, d2, d1, inv, z1, ld1, k.G12, d2, 1.0, g2, ldg2, nullptr);
  }
  sync(c);
}

// Row-sharded batches: the moments have been summed over all ranks (ccz_moments + one all-reduce); this gives
// the loss and ONE (d1 + d2)^2 matrix Gamma with  [dz1 | dz2] = ([z1 | z2] - 1 mean') Gamma  for any subset of
// the rows (each rank applies it to its own shard with ccz_transform).
static void cca_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, int64_t d1, int64_t d2, double eps,
                                  double* loss_host, double* gamma_dev, double* mean_dev) {
  if (!mom || !loss_host) fail(CCZ_EINVAL, "cca_loss_moments: null argument");
  if (n < 2 || d1 < 1 || d2 < 1) fail(CCZ_EINVAL, "cca_loss_moments: bad shape");
  const int64_t D = d1 + d2;
  const bool want = gamma_dev != nullptr;
  if (want && !mean_dev) fail(CCZ_EINVAL, "cca_loss_moments: mean_dev is required with gamma_dev");
  LossCore k = cca_loss_core(c, mom, mom + D * D, n, d1, d2, eps, want, want);
  *loss_host = k.loss;
  if (want) {
    const double inv = 1.0 / double(n - 1);
    copy2d(c, d1, d1, k.G11s, d1, gamma_dev, D);
    copy2d(c, d1, d2, k.G12, d2, gamma_dev + d1, D);
    copy2d(c, d2, d1, k.G12t, d1, gamma_dev + d1 * D, D);
    copy2d(c, d2, d2, k.G22s, d2, gamma_dev + d1 * D + d1, D);
    axpby2d(c, D, D, inv, gamma_dev, D, 0.0, nullptr, 0);
    d2d(c, mean_dev, k.mu, size_t(D) * 8);
  }
  sync(c);
}

// out = (X - mean) W     reference: cca_zoo/_base.py:108-123
static void transform_impl(ccz_ctx* c, int dtype, const void* X, int64_t n, int64_t d, int64_t ld, const double* mean,
                           const double* W, int64_t k, void* out, int64_t ldo) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "transform: dtype must be CCZ_F32 or CCZ_F64");
  if (!X || !W || !out || n < 1 || d < 1 || k < 1 || ld < d || ldo < k) fail(CCZ_EINVAL, "transform: bad argument");
  DBuf bias(c, k);
  if (mean) gemm(c, false, false, 1, k, d, 1.0, mean, d, W, k, 0.0, bias, k);
  gemm_mixed(c, dtype, n, k, d, 1.0, X, ld, W, k, 0.0, out, ldo, mean ? bias.get() : nullptr);
  sync(c);
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
    for (auto& g : im->graphs) (void)hipGraphExecDestroy(g.exec);
    if (im->own_stream) (void)hipStreamDestroy(im->own_stream);
    for (auto& b : im->pool) (void)hipFree(b.p);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(im->ev[i]);
    for (int i = 0; i < 4; ++i) if (im->pipe_ev[i]) (void)hipEventDestroy(im->pipe_ev[i]);
    for (int i = 0; i < 2; ++i) if (im->pin_buf[i]) (void)hipHostFree(im->pin_buf[i]);
    if (im->copy_stream) (void)hipStreamDestroy(im->copy_stream);
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
    h->stream = s;
  })
}

int ccz_sync(ccz_handle h) { CCZ_GUARD(h, sync(h)) }

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

int ccz_moments_last_ms(ccz_handle h, double* gram_ms, double* colsum_ms) {
  CCZ_GUARD(h, {
    if (gram_ms) *gram_ms = h->last_gram_ms;
    if (colsum_ms) *colsum_ms = h->last_colsum_ms;
  })
}

int ccz_cca_loss(ccz_handle h, int dtype, const void* z1_dev, const void* z2_dev, int64_t n, int64_t d1, int64_t d2,
                 int64_t ld1, int64_t ld2, double eps, void* loss_dev, void* g1_dev, void* g2_dev, int64_t ldg1,
                 int64_t ldg2) {
  CCZ_GUARD(h, {
    cca_loss_impl(h, dtype, z1_dev, z2_dev, n, d1, d2, ld1, ld2, eps, loss_dev, g1_dev, g2_dev, ldg1, ldg2);
  })
}

int ccz_cca_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, int64_t d1, int64_t d2, double eps,
                         double* loss_host, double* gamma_dev, double* mean_dev) {
  CCZ_GUARD(h, ccz::cca_loss_moments_impl(h, moments_dev, n_rows, d1, d2, eps, loss_host, gamma_dev, mean_dev));
}

int ccz_transform(ccz_handle h, int dtype, const void* X_dev, int64_t n, int64_t d, int64_t ld, const double* mean_dev,
                  const double* W_dev, int64_t k, void* out_dev, int64_t ldo) {
  CCZ_GUARD(h, {
    transform_impl(h, dtype, X_dev, n, d, ld, mean_dev, W_dev, k, out_dev, ldo);
  })
}

}  // extern "C"
