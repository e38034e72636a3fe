// DCCA correlation losses: value + closed-form input gradients, fused on the device.
//
// reference: cca_zoo/deep/objectives.py:61-102 (CCALoss.forward) + its autograd backward, :138-153 (MCCALoss =
// sum of the pairwise losses); maths: oracle/losses.py::cca_loss_closed_form.
//
// For m views z_1 .. z_m (batch n, widths d_a, D = sum d_a) everything follows from ONE matrix.  With
//   Ce   = centred batch covariance of [z_1 .. z_m] + eps I              (D x D, from the K1 moments)
//   Sinv = blockdiag((Ce_aa)^-1)                                          (one Cholesky + inverse per VIEW)
//   A    = Sinv Ce          (block (a, b) = S_aa^-1 S_ab,  diagonal blocks = I)
//   M    = A Sinv           (block (a, b) = S_aa^-1 S_ab S_bb^-1, symmetric)
// the sum over all pairs a < b of  -tr(S_aa^-1 S_ab S_bb^-1 S_ba)  is
//   loss  = -1/2 sum_{a != b} tr(A_ab A_ba)
// and its gradient with respect to the stacked batch is  [dz_1 .. dz_m] = ([z_1 .. z_m] - 1 mean') Gamma  with
//   Gamma_ab = -2 M_ab / (n-1)   (a != b),      Gamma_aa = (2 (A M)_aa - 2 M_aa) / (n-1)
// (for m = 2 these are the G12, G11 + G11', G22 + G22' of the two-view closed form).  Launch sequence, all on the
// handle's stream with no host synchronisation until the very end:
//   K1 (pilot-shifted for fp32) -> prep (Ce, mean, per-view copies) -> batched Cholesky + inverse (cholinv.hip,
//   d/64 + 1 launches for all views together) -> 4 batched 64-tile GEMM launches (Sinv_a; A_ab; Gamma_ab;
//   Gamma_aa = -sum_b A_ab Gamma_ba) -> loss reduction -> ONE sample-side GEMM (Z - mean) Gamma on the fp32 MFMA pipe.
// The previous formulation (blocked potrf + two triangular solves against the identity + 14 GEMMs + a host round
// trip for the loss value) was ~70 dependent launches and 3.1 ms at batch 8192, 2 x 512.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip_common.h"

namespace ccz {

namespace {

constexpr int LMAXV = 8;
struct PrepArgs {
  int64_t off[LMAXV + 1];
  double* work[LMAXV];     // per-view copy of the diagonal block (consumed by the factorization)
  double* zero_v[LMAXV];   // optional per-view d_a x d_a buffers to clear (split-K destinations), same indexing as work
  double* zero_a;          // optional D x D buffers to clear
  double* zero_b;
  int m;
};

// Ce = (G - s s'/n)/(n-1) + eps I from the upper triangle of G; mean = s / n; diagonal blocks also to work[a]
__global__ void k_loss_prep(const double* __restrict__ G, const double* __restrict__ s, int64_t D, double inv_n, double inv_nm1,
                            double eps, double* __restrict__ Ce, double* __restrict__ mean, PrepArgs pa,
                            double* __restrict__ acc, double* __restrict__ bias) {
  const int64_t total = D * D;
  if (blockIdx.x == 0) {                    // accumulators of later kernels: the loss sum and the bias row mean' Gamma
    if (threadIdx.x == 0) acc[0] = 0.0;
    if (bias) for (int64_t j = threadIdx.x; j < D; j += blockDim.x) bias[j] = 0.0;
  }
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = e / D, j = e - i * D;
    double v = i <= j ? G[i * D + j] : G[j * D + i];
    v = (v - s[i] * s[j] * inv_n) * inv_nm1;
    if (i == j) { v += eps; mean[i] = s[i] * inv_n; }
    Ce[e] = v;
    if (pa.zero_a) pa.zero_a[e] = 0.0;
    if (pa.zero_b) pa.zero_b[e] = 0.0;
    int a = 0;
    while (a + 1 < pa.m && i >= pa.off[a + 1]) ++a;
    if (j >= pa.off[a] && j < pa.off[a + 1]) {
      const int64_t da = pa.off[a + 1] - pa.off[a];
      const int64_t q = (i - pa.off[a]) * da + (j - pa.off[a]);
      pa.work[a][q] = v;
      if (pa.zero_v[a]) pa.zero_v[a][q] = 0.0;
    }
  }
}

// acc += sum over (i, j) in DIFFERENT view blocks of A_ij A_ji  = 2 sum_{a<b} tr(A_ab A_ba)
// (the diagonal blocks of A are the identity up to cond * eps rounding: leaving them out keeps that noise out of the loss)
__global__ void k_trace_sq(const double* __restrict__ A, int64_t D, PrepArgs pa, double* __restrict__ acc) {
  __shared__ double red[4];
  double v = 0.0;
  const int64_t total = D * D;
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = e / D, j = e - i * D;
    int a = 0;
    while (a + 1 < pa.m && i >= pa.off[a + 1]) ++a;
    if (j >= pa.off[a] && j < pa.off[a + 1]) continue;
    v += A[e] * A[j * D + i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

// loss = -1/2 acc -> one element of dtype (and/or a double)
__global__ void k_loss_finish(const double* __restrict__ acc, int dtype, void* __restrict__ out, double* __restrict__ out64) {
  const double l = -0.5 * acc[0];
  if (out) { if (dtype == CCZ_F32) *static_cast<float*>(out) = float(l); else *static_cast<double*>(out) = l; }
  if (out64) *out64 = l;
}

// bias[j] += sum over a 64-row slab of mean_i Gamma_ij  (bias zeroed by k_loss_prep); grid (D / 64, D / 64)
__global__ __launch_bounds__(256) void k_bias_row(const double* __restrict__ Gm, const double* __restrict__ mean, int64_t D,
                                                  double* __restrict__ bias) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t j = int64_t(blockIdx.x) * 64 + c, i0 = int64_t(blockIdx.y) * 64;
  double a = 0.0;
  if (j < D)
    for (int64_t i = i0 + rg; i < min(D, i0 + 64); i += 4) a += mean[i] * Gm[i * D + j];
  red[rg][c] = a;
  __syncthreads();
  if (rg == 0 && j < D) unsafeAtomicAdd(bias + j, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// fp64 -> fp32, elementwise (Gamma and, appended as one more row, the bias mean' Gamma)
__global__ void k_cvt_f32(const double* __restrict__ in, int64_t total, float* __restrict__ out) {
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) out[e] = float(in[e]);
}

__global__ void k_neg_sum(const double* __restrict__ v, int64_t n, int dtype, void* __restrict__ out) {
  __shared__ double red[4];
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double l = -(red[0] + red[1] + red[2] + red[3]);
    if (dtype == CCZ_F32) *static_cast<float*>(out) = float(l); else *static_cast<double*>(out) = l;
  }
}

constexpr int64_t kFusedMaxD = 2048;     // per-view width served by the fused (cholinv) core

bool fused_ok(const int64_t* dims, int m) {
  if (m < 2 || m > LMAXV) return false;
  for (int a = 0; a < m; ++a)
    if (dims[a] > kFusedMaxD) return false;
  static const int on = [] { const char* e = getenv("CCZ_LOSS_FUSED"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// Gamma (D x D), mean (D) and the loss accumulator tr(A A) (one double, zeroed here) from the moments; info_dev: m ints.
// Everything is enqueued on the handle's stream; nothing is read back.
void fused_core(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps, bool want_grad,
                double* acc_dev, double* gamma_dev, double* mean_dev, int* info_dev, double* bias_dev = nullptr) {
  hipStream_t st = stream(c);
  std::vector<int64_t> off(m + 1, 0);
  for (int a = 0; a < m; ++a) off[a + 1] = off[a] + dims[a];
  const int64_t D = off[m];
  const double inv = 1.0 / double(n - 1);
  DBuf Ce(c, D * D), Am(c, D * D);
  std::vector<DBuf> work(m), Lf(m), X(m), T(m), Sinv(m);
  PrepArgs pa{};
  pa.m = m;
  for (int a = 0; a <= m; ++a) pa.off[a] = off[a];
  for (int a = 0; a < m; ++a) {
    const int64_t d = dims[a], nblk = (d + 63) / 64;
    work[a] = DBuf(c, d * d); Lf[a] = DBuf(c, d * d); X[a] = DBuf(c, d * d); Sinv[a] = DBuf(c, d * d);
    T[a] = DBuf(c, nblk * 4096);
    pa.work[a] = work[a].get();
  }
  // (split-K destinations of the product stages below are cleared by the same pass -- see there)
  int64_t dmin = dims[0];
  for (int a = 1; a < m; ++a) dmin = std::min(dmin, dims[a]);
  static const int split_env = [] { const char* e = getenv("CCZ_LOSS_SPLITK"); return e ? atoi(e) : 4; }();
  const int ks = (dmin >= 256 && split_env > 1) ? split_env : 1;
  if (ks > 1) {
    for (int a = 0; a < m; ++a) pa.zero_v[a] = Sinv[a].get();
    pa.zero_a = Am.get();
    pa.zero_b = want_grad ? gamma_dev : nullptr;
  }
  hipLaunchKernelGGL(k_loss_prep, dim3((unsigned)std::min<int64_t>((D * D + 255) / 256, 4096)), dim3(256), 0, st, mom, mom + D * D, D,
                     1.0 / double(n), inv, eps, Ce.get(), mean_dev, pa, acc_dev, want_grad ? bias_dev : nullptr);
  CCZ_LAUNCH_CHECK();
  {
    std::vector<double*> Ap(m), Lp(m), Xp(m), Tp(m);
    std::vector<int64_t> ld(dims, dims + m);
    for (int a = 0; a < m; ++a) { Ap[a] = work[a].get(); Lp[a] = Lf[a].get(); Xp[a] = X[a].get(); Tp[a] = T[a].get(); }
    cholinv_batched(c, m, Ap.data(), ld.data(), ld.data(), Lp.data(), ld.data(), Xp.data(), ld.data(), Tp.data(), info_dev);
  }
  // Only the blocks that are not trivially known are formed: A_aa = I is never computed, Gamma's diagonal blocks come
  // straight from  Gamma_aa = -sum_{b != a} A_ab Gamma_ba  ( = 2/(n-1) ((A M)_aa - M_aa) ).
  // The four product stages below are dependent launches of at most a few hundred 64 x 64 tiles with K = d: on
  // their own they leave most of the chip idle for ~40 us each.  From d = 256 on the K range of every tile is cut
  // into four slices on separate workgroups that accumulate atomically into zeroed destinations (split-K).
  // (ks, and the clearing of Sinv / Am / Gamma: in k_loss_prep above -- three fills fewer per call)
  auto launch = [&](std::vector<MultiGemmArgs>& v) {
    if (ks > 1)
      for (auto& g : v) { g.ksplit = ks; g.beta = 1.0; }       // destinations are zero (or hold the earlier terms of a sum)
    for (size_t i0 = 0; i0 < v.size(); i0 += 8) gemm_f64_multi(c, int(std::min<size_t>(8, v.size() - i0)), v.data() + i0);
  };
  std::vector<MultiGemmArgs> pr;
  // Sinv_a = X_a' X_a   (X lower triangular: the K loop starts at the diagonal; the blocks of X above it -- never
  // written by cholinv -- are never read)
  for (int a = 0; a < m; ++a)
    pr.push_back(MultiGemmArgs{X[a], X[a], Sinv[a], nullptr, dims[a], dims[a], dims[a], 0, dims[a], dims[a], dims[a], true, false, false, 1.0, 0.0, true});
  launch(pr);
  // A_ab = Sinv_a Ce_ab   (a != b)
  pr.clear();
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b)
      if (a != b)
        pr.push_back(MultiGemmArgs{Sinv[a], Ce.get() + off[a] * D + off[b], Am.get() + off[a] * D + off[b], nullptr, dims[a], D, D, 0,
                                   dims[a], dims[b], dims[a], false, false, false, 1.0, 0.0});
  launch(pr);
  hipLaunchKernelGGL(k_trace_sq, dim3((unsigned)std::min<int64_t>((D * D + 255) / 256, 1024)), dim3(256), 0, st, Am.get(), D, pa, acc_dev);
  CCZ_LAUNCH_CHECK();
  if (!want_grad) return;
  // Gamma_ab = -2/(n-1) A_ab Sinv_b   (a != b)
  pr.clear();
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b)
      if (a != b)
        pr.push_back(MultiGemmArgs{Am.get() + off[a] * D + off[b], Sinv[b], gamma_dev + off[a] * D + off[b], nullptr, D, dims[b], D, 0,
                                   dims[a], dims[b], dims[b], false, false, false, -2.0 * inv, 0.0});
  launch(pr);
  // Gamma_aa = -sum_{b != a} A_ab Gamma_ba : one launch per offset s (b = a + s mod m), so that no two problems of a
  // launch write the same block; the first one overwrites, the others accumulate
  for (int sft = 1; sft < m; ++sft) {
    pr.clear();
    for (int a = 0; a < m; ++a) {
      const int b = (a + sft) % m;
      pr.push_back(MultiGemmArgs{Am.get() + off[a] * D + off[b], gamma_dev + off[b] * D + off[a], gamma_dev + off[a] * D + off[a], nullptr,
                                 D, D, D, 0, dims[a], dims[a], dims[b], false, false, false, -1.0, sft > 1 ? 1.0 : 0.0});
    }
    launch(pr);
  }
  if (bias_dev)
    hipLaunchKernelGGL(k_bias_row, dim3((unsigned)((D + 63) / 64), (unsigned)((D + 63) / 64)), dim3(256), 0, st, gamma_dev, mean_dev, D, bias_dev);
  CCZ_LAUNCH_CHECK();
}

void check_info(ccz_ctx* c, const int* info_dev, int m, const char* what) {
  int got[LMAXV];
  d2h(c, got, info_dev, size_t(m) * sizeof(int));      // the one synchronisation of the fused path
  for (int a = 0; a < m; ++a)
    if (got[a] != 0x7fffffff) fail(CCZ_ENOTSPD, "%s: S_%d%d + eps I is not positive definite (pivot %d)", what, a + 1, a + 1, got[a] - 1);
}

// ---------------------------------------------------------------------------
// wide blocks (d > 2048; the metric shape d = 4096): blocked factorizations + explicit triangular inverses, the
// sample-side GEMMs dominate there (4 x n d^2 flops at 95 % of the fp32 MFMA peak).  Two views only.
// ---------------------------------------------------------------------------
struct LossCore {
  DBuf G11s, G22s, G12, G12t, mu, rd;
};

LossCore wide_core(ccz_ctx* c, const double* G, const double* s, int64_t n, int64_t d1, int64_t d2, double eps, bool want1, bool want2) {
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
  auto tri_inverse = [&](const double* L, int64_t d) {
    DBuf Li(c, d * d);
    fill2d(c, d, d, Li, d, 0.0);
    add_diag(c, d, Li, d, 1.0);
    trsm_right_lower(c, false, d, d, L, d, Li, d);            // I L^-1
    return Li;
  };
  DBuf Li1 = tri_inverse(L1, d1), Li2 = tri_inverse(L2, d2);
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
  out.rd = DBuf(c, d1);
  row_dots(c, d1, d2, A, d2, Bmt, d2, out.rd);                // tr(A Bm) = sum A o Bm'   (summed on the device)
  if (!want1 && !want2) return out;
  out.G12 = DBuf(c, d1 * d2);
  out.G12t = DBuf(c, d2 * d1);
  solve_right(Li2, d2, A, d2, d1, -2.0, out.G12);
  transpose(c, d1, d2, out.G12, d2, out.G12t, d1);
  out.mu = DBuf(c, D);
  d2d(c, out.mu, s, size_t(D) * 8);
  axpby2d(c, 1, D, 1.0 / double(n), out.mu, D, 0.0, nullptr, 0);
  if (want1) {
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

}  // namespace

void cca_loss_impl(ccz_ctx* c, int dtype, const void* z1, const void* z2, int64_t n, int64_t d1, int64_t d2, int64_t ld1,
                   int64_t ld2, double eps, void* loss_dev, void* g1, void* g2, int64_t ldg1, int64_t ldg2) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "cca_loss: dtype must be CCZ_F32 or CCZ_F64");
  if (!z1 || !z2 || !loss_dev) fail(CCZ_EINVAL, "cca_loss: null argument");
  if (n < 2 || d1 < 1 || d2 < 1 || ld1 < d1 || ld2 < d2) fail(CCZ_EINVAL, "cca_loss: bad shape");
  if ((g1 && ldg1 < d1) || (g2 && ldg2 < d2)) fail(CCZ_EINVAL, "cca_loss: bad gradient stride");
  const int64_t D = d1 + d2;
  const double inv = 1.0 / double(n - 1);
  const int64_t dims[2] = {d1, d2};
  const bool want = g1 || g2;
  hipStream_t st = stream(c);

  if (fused_ok(dims, 2)) {
    // fp32 batches whose widths suit the 256-column tiles of the FIFO GEMM are gathered into one n x D matrix: the
    // gradient is then ONE product (Z - mean) Gamma whose column ranges land in g1 / g2
    const size_t es = dtype == CCZ_F32 ? 4 : 8;
    const bool fifo = dtype == CCZ_F32 && want && g1 && g2 &&
                      gemm_f32_fifo_split_eligible(n, D, D, d1, g1, ldg1, g2, ldg2);
    void* zcat = nullptr;
    ccz_view views[2] = {{z1, d1, ld1}, {z2, d2, ld2}};
    int nviews = 2;
    if (fifo) {
      zcat = dev_alloc(c, size_t(n) * D * es);
      CCZ_HIP(hipMemcpy2DAsync(zcat, size_t(D) * es, z1, size_t(ld1) * es, size_t(d1) * es, size_t(n), hipMemcpyDeviceToDevice, st));
      CCZ_HIP(hipMemcpy2DAsync(static_cast<char*>(zcat) + size_t(d1) * es, size_t(D) * es, z2, size_t(ld2) * es, size_t(d2) * es, size_t(n),
                               hipMemcpyDeviceToDevice, st));
      views[0] = ccz_view{zcat, D, D};                        // one n x D view: one column-sum launch, same tiles
      nviews = 1;
    }
    // Gamma and, as row D of the same buffer, the bias row mean' Gamma (the centring of the batch)
    DBuf mom(c, D * D + D), gamma(c, want ? (D + 1) * D : 0), mean(c, D), acc(c, 1);
    int* info_dev = static_cast<int*>(dev_alloc(c, LMAXV * sizeof(int)));
    // embeddings (post-ReLU, un-normalised) routinely sit far from zero: always take the pilot-shifted Gram for
    // fp32 -- no host read-back, and at batch sizes the staged kernel costs the same as the FIFO one
    moments_impl(c, dtype, views, nviews, n, true, mom, false, dtype == CCZ_F32 ? 2 : 0, false);
    fused_core(c, mom, n, dims, 2, eps, want, acc, gamma, mean, info_dev, want ? gamma.get() + D * D : nullptr);
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1), 0, st, acc.get(), dtype, loss_dev, static_cast<double*>(nullptr));
    CCZ_LAUNCH_CHECK();
    if (want) {
      double* bias = gamma.get() + D * D;
      if (fifo) {
        float* G32 = static_cast<float*>(dev_alloc(c, size_t(D + 1) * D * 4));
        hipLaunchKernelGGL(k_cvt_f32, dim3((unsigned)std::min<int64_t>(((D + 1) * D + 255) / 256, 2048)), dim3(256), 0, st, gamma.get(),
                           (D + 1) * D, G32);
        CCZ_LAUNCH_CHECK();
        gemm_f32_fifo_split(c, n, D, D, 1.0f, static_cast<const float*>(zcat), D, G32, G32 + D * D, static_cast<float*>(g1), ldg1,
                            static_cast<float*>(g2), ldg2, d1);
        dev_free(c, G32);
      } else {
        if (g1) {
          gemm_mixed(c, dtype, n, d1, d1, 1.0, z1, ld1, gamma, D, 0.0, g1, ldg1, bias);
          gemm_mixed(c, dtype, n, d1, d2, 1.0, z2, ld2, gamma.get() + d1 * D, D, 1.0, g1, ldg1, nullptr);
        }
        if (g2) {
          gemm_mixed(c, dtype, n, d2, d2, 1.0, z2, ld2, gamma.get() + d1 * D + d1, D, 0.0, g2, ldg2, bias + d1);
          gemm_mixed(c, dtype, n, d2, d1, 1.0, z1, ld1, gamma.get() + d1, D, 1.0, g2, ldg2, nullptr);
        }
      }
    }
    try {
      check_info(c, info_dev, 2, "cca_loss");
    } catch (...) {
      dev_free(c, info_dev);
      if (zcat) dev_free(c, zcat);
      throw;
    }
    dev_free(c, info_dev);
    if (zcat) dev_free(c, zcat);
    return;
  }

  DBuf mom(c, D * D + D);
  ccz_view views[2] = {{z1, d1, ld1}, {z2, d2, ld2}};
  moments_impl(c, dtype, views, 2, n, true, mom, false, 1, false);
  LossCore k = wide_core(c, mom, mom.get() + D * D, n, d1, d2, eps, g1 != nullptr, g2 != nullptr);
  hipLaunchKernelGGL(k_neg_sum, dim3(1), dim3(256), 0, st, k.rd.get(), d1, dtype, loss_dev);
  CCZ_LAUNCH_CHECK();
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

// Sum of the pairwise CCA losses of m views from the (all-reduced) batch moments: loss (host) and, if gamma_dev != NULL,
// Gamma (D x D) and the batch mean (D) with  [dz_1 .. dz_m] = ([z_1 .. z_m] - 1 mean') Gamma  for ANY subset of the rows.
void pair_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps,
                            double* loss_host, double* gamma_dev, double* mean_dev) {
  if (!mom || !dims || !loss_host) fail(CCZ_EINVAL, "pair_loss_moments: null argument");
  if (m < 2 || n < 2) fail(CCZ_EINVAL, "pair_loss_moments: at least 2 views and 2 rows are required");
  for (int a = 0; a < m; ++a)
    if (dims[a] < 1) fail(CCZ_EINVAL, "pair_loss_moments: view %d has no features", a);
  const bool want = gamma_dev != nullptr;
  if (want && !mean_dev) fail(CCZ_EINVAL, "pair_loss_moments: mean_dev is required with gamma_dev");
  int64_t D = 0;
  for (int a = 0; a < m; ++a) D += dims[a];
  if (fused_ok(dims, m)) {
    DBuf acc(c, 2), mean_tmp(c, want ? 0 : D);
    int* info_dev = static_cast<int*>(dev_alloc(c, LMAXV * sizeof(int)));
    fused_core(c, mom, n, dims, m, eps, want, acc, gamma_dev, want ? mean_dev : mean_tmp.get(), info_dev);
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1), 0, stream(c), acc.get(), CCZ_F64, static_cast<void*>(nullptr), acc.get() + 1);
    CCZ_LAUNCH_CHECK();
    try {
      check_info(c, info_dev, m, "pair_loss_moments");
    } catch (...) {
      dev_free(c, info_dev);
      throw;
    }
    dev_free(c, info_dev);
    d2h(c, loss_host, acc.get() + 1, 8);
    return;
  }
  if (m != 2) fail(CCZ_EUNSUP, "pair_loss_moments: views wider than %lld features are supported for 2 views only", (long long)kFusedMaxD);
  const int64_t d1 = dims[0], d2 = dims[1];
  LossCore k = wide_core(c, mom, mom + D * D, n, d1, d2, eps, want, want);
  std::vector<double> rh(d1);
  d2h(c, rh.data(), k.rd, size_t(d1) * 8);
  double l = 0.0;
  for (double v : rh) l -= v;
  *loss_host = l;
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

void cca_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, int64_t d1, int64_t d2, double eps, double* loss_host,
                           double* gamma_dev, double* mean_dev) {
  const int64_t dims[2] = {d1, d2};
  pair_loss_moments_impl(c, mom, n, dims, 2, eps, loss_host, gamma_dev, mean_dev);
}

}  // namespace ccz
