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
//   general path: K1 (pilot-shifted for fp32) -> prep (Ce, mean, per-view copies) -> batched Cholesky + inverse
//   (cholinv.hip: one persistent launch, or d/64 + 1 launches) -> 4 batched 64-tile GEMM launches (Sinv_a; A_ab;
//   Gamma_ab; Gamma_aa = -sum_b A_ab Gamma_ba) -> loss reduction -> sample-side GEMM(s) (Z - mean) Gamma;
//   fast path (round 5; two aligned fp32 views <= 2048 columns, n % 32 == 0 -- a DCCA batch; CCZ_LOSS_FAST=0 turns it
//   off): k_colsum_pilot -> K1 writing partial tiles (k_gram_f32_fifo_small) -> k_loss_prep_partials (reduce + Ce in
//   the per-view layout + zeroed destinations) -> k_cholinv_chain -> the 4 GEMM stages, the loss riding on the third
//   -> k_loss_tail -> [backward] ONE fp32 product over the two views where they lie, grad_output read on the device:
//   10 dispatches per forward + backward (profiles/r05_loss_c4.md).  The two phases are separate entry
//   points (pair_loss_forward_impl / pair_loss_backward_impl) joined by a caller-owned state buffer.
// Views wider than 2048 columns (the metric shape, d = 4096) run the SAME formulas through the super-blocked
// factorization, explicit triangular inverses and the 128-tile fp64 GEMM (pair_core, `narrow == false`); there the four
// sample-side n x d x d products dominate.  Nothing on the narrow path is read back by the host: a non-positive pivot
// makes the loss NaN and sets the handle's sticky status (ccz_loss_status), checked by the NEXT call.
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

// The same preparation straight from K1's per-(row chunk, tile) fp32 partial sums (gram.hip: gram_partials_f32) -- the
// loss fast path never forms the moments [G | s]: the chunks are added up in fp64, the pilot shift is undone in the
// centred form
//   sum_r (x_ri - m_i)(x_rj - m_j) = sum_r (x_ri - p_i)(x_rj - p_j) - n (m_i - p_i)(m_j - p_j)       (m = s / n)
// and Ce, the views' diagonal blocks, the mean, the cleared split-K destinations and accumulators all leave this ONE
// launch (rounds 2-4: k_gram_reduce, k_pilot_fixup, k_vec_add, two fills and k_loss_prep).  grid (64, ntiles) x 256: one
// workgroup per 32 x 32 sub-block of a 256 x 256 tile (four elements per thread: the 64 MB of partial sums of a DCCA batch
// want many loads in flight).  Tiles cover the upper triangle; the mirrored half of every matrix is written from an LDS
// transpose, so both halves go out as full 256-byte rows.
constexpr int PSB = 32;
__global__ __launch_bounds__(256) void k_loss_prep_partials(const float* __restrict__ partial, const GramTile* __restrict__ tiles, int ntiles,
                                                            int64_t ksplit, const int* __restrict__ plan, const double* __restrict__ s,
                                                            const float* __restrict__ pilot,
                                                            int64_t D, double n_rows, double inv_nm1, double eps, double* __restrict__ Ce,
                                                            double* __restrict__ mean, PrepArgs pa, double* __restrict__ acc,
                                                            double* __restrict__ bias, const double* __restrict__ msq) {
  // msq (split route, gram_split.hip): sum_k mid_k^2 per column, the one dropped product of the split arithmetic that does not
  // average out -- added on the diagonal before the shift is undone
  __shared__ double tr[PSB][PSB + 1];
  const int tile = blockIdx.y;
  if (tile == 0) {
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < D; e += 64 * 256) {
      mean[e] = s[e] / n_rows;
      if (bias) bias[e] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[0] = 0.0;
  }
  const GramTile t = tiles[tile];
  const int si = blockIdx.x >> 3, sj = blockIdx.x & 7;
  if (t.diag && si > sj) return;                           // diagonal tiles hold both halves: the upper sub-blocks are used
  if (si * PSB >= t.wa || sj * PSB >= t.wb) return;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const int64_t gi0 = t.out_row + si * PSB, gj0 = t.out_col + sj * PSB;
  const int vi = t.wa - si * PSB, vj = t.wb - sj * PSB;    // valid rows / columns of this sub-block
  int a = 0;                                               // the view of this sub-block's rows (a sub-block never straddles views:
  while (a + 1 < pa.m && gi0 >= pa.off[a + 1]) ++a;        // panels are cut per view)
  const bool same_view = gj0 < pa.off[a + 1];
  const int64_t da = pa.off[a + 1] - pa.off[a], li0 = gi0 - pa.off[a], lj0 = gj0 - pa.off[a];
  const double dj = c < vj ? s[gj0 + c] / n_rows - double(pilot[gj0 + c]) : 0.0;
  double g[4] = {0.0, 0.0, 0.0, 0.0};
  // slots of this tile: (chunk, tile) interleaved (k_gram_f32) or contiguous per tile (plan: k_gram_f32_fifo_small)
  const int64_t slot0 = plan ? plan[3 * tile] : tile, nslots = plan ? plan[3 * tile + 1] : ksplit, sstride = plan ? 1 : ntiles;
  // FIFO layout of a diagonal tile: the symmetric quadrants hold their upper triangles only (lower elements come from the
  // transposed position), Q01 is the sum of its slot and of the Q10 slot
  const bool fifo_diag = plan != nullptr && t.diag;
  const bool q01 = fifo_diag && si < 4 && sj >= 4;
  const bool qsym_diag_block = fifo_diag && si == sj;      // a 32 x 32 block on the tile's diagonal
  {
    const int rr0 = si * PSB + r0, cc = sj * PSB + c;
    const float* p = partial + slot0 * 65536;
    for (int64_t ch = 0; ch < nslots; ++ch) {
      const float* pc = p + ch * sstride * 65536;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = rr0 + 8 * k;
        if (r0 + 8 * k < vi && c < vj) {
          float x;
          if (qsym_diag_block && rr > cc) x = pc[cc * 256 + rr];      // lower element of a symmetric block: stored transposed
          else x = pc[rr * 256 + cc];
          if (q01) x += pc[(rr + 128) * 256 + cc - 128];
          g[k] += double(x);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 8 * k;
    double v = 0.0;
    if (r < vi && c < vj) {
      const double di = s[gi0 + r] / n_rows - double(pilot[gi0 + r]);
      const double gk = (msq && gi0 + r == gj0 + c) ? g[k] + msq[gi0 + r] : g[k];
      v = (gk - n_rows * di * dj) * inv_nm1;
      if (gi0 + r == gj0 + c) v += eps;
      const int64_t e1 = (gi0 + r) * D + gj0 + c;
      Ce[e1] = v;
      if (pa.zero_a) pa.zero_a[e1] = 0.0;
      if (pa.zero_b) pa.zero_b[e1] = 0.0;
      if (same_view) {
        const int64_t q1 = (li0 + r) * da + lj0 + c;
        pa.work[a][q1] = v;
        if (pa.zero_v[a]) pa.zero_v[a][q1] = 0.0;
      }
    }
    tr[r][c] = v;
  }
  if (t.diag && si == sj) return;                          // symmetric sub-block: the direct pass wrote both halves
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 8 * k;                              // row of the MIRRORED sub-block = column r of this one
    if (r < vj && c < vi) {
      const double v = tr[c][r];
      const int64_t e2 = (gj0 + r) * D + gi0 + c;
      Ce[e2] = v;
      if (pa.zero_a) pa.zero_a[e2] = 0.0;
      if (pa.zero_b) pa.zero_b[e2] = 0.0;
      if (same_view) {
        const int64_t q2 = (lj0 + r) * da + li0 + c;
        pa.work[a][q2] = v;
        if (pa.zero_v[a]) pa.zero_v[a][q2] = 0.0;
      }
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

// loss = -1/2 acc -> one element of dtype (and/or a double).  info (optional, m ints from the factorization): a failed
// pivot turns the loss into NaN and is recorded in the handle's sticky status words (pinned host memory) -- the
// stream-native replacement of a blocking read-back of the pivot flags after every call.
__global__ void k_loss_finish(const double* __restrict__ acc, int dtype, void* __restrict__ out, double* __restrict__ out64,
                              const int* __restrict__ info, int m, int* __restrict__ status) {
  double l = -0.5 * acc[0];
  for (int a = 0; a < m; ++a)
    if (info[a] != 0x7fffffff) {
      l = __builtin_nan("");
      if (status && status[0] == 0) { status[1] = info[a] - 1; __threadfence_system(); status[0] = a + 1; }
      break;
    }
  if (out) { if (dtype == CCZ_F32) *static_cast<float*>(out) = float(l); else *static_cast<double*>(out) = l; }
  if (out64) *out64 = l;
}

// Everything between the last product stage and the sample-side GEMM in ONE launch (rounds 2-4: k_bias_row, k_cvt_f32 and
// k_loss_finish): bias[j] += sum over a 64-row slab of mean_i Gamma_ij (bias zeroed by the prep kernel), Gamma -> fp32
// (G32 != null), and -- workgroup (0, 0) -- the loss value -1/2 acc with the pivot check of k_loss_finish.
// grid (D / 64, D / 64) x 256.
__global__ __launch_bounds__(256) void k_loss_tail(const double* __restrict__ Gm, const double* __restrict__ mean, int64_t D,
                                                   double* __restrict__ bias, float* __restrict__ G32, const double* __restrict__ acc, int dtype,
                                                   void* __restrict__ out, const int* __restrict__ info, int m, int* __restrict__ status,
                                                   double* __restrict__ corr) {
  // corr (optional, zeroed by k_state_rows): sum_i (mean_i - fl32(mean_i)) Gamma_ij -- what a backward on operands shifted by the
  // fp32 pilot fl32(mean) still owes the exact centring (gemm_split.hip)
  __shared__ double red[4][64];
  __shared__ double redc[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t j = int64_t(blockIdx.x) * 64 + c, i0 = int64_t(blockIdx.y) * 64;
  double a = 0.0, ac = 0.0;
  if (j < D)
    for (int64_t i = i0 + rg; i < min(D, i0 + 64); i += 4) {
      const double g = Gm[i * D + j];
      const double mi = mean[i];
      a += mi * g;
      ac += (mi - double(float(mi))) * g;
      if (G32) G32[i * D + j] = float(g);
    }
  red[rg][c] = a;
  redc[rg][c] = ac;
  __syncthreads();
  if (rg == 0 && j < D) unsafeAtomicAdd(bias + j, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  if (corr && rg == 1 && j < D) unsafeAtomicAdd(corr + j, redc[0][c] + redc[1][c] + redc[2][c] + redc[3][c]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    double l = -0.5 * acc[0];
    for (int v = 0; v < m; ++v)
      if (info[v] != 0x7fffffff) {
        l = __builtin_nan("");
        if (status && status[0] == 0) { status[1] = info[v] - 1; __threadfence_system(); status[0] = v + 1; }
        break;
      }
    if (dtype == CCZ_F32) *static_cast<float*>(out) = float(l); else *static_cast<double*>(out) = l;
  }
}

// the state's batch-mean row <- mean, its pilot-correction row <- 0 (k_loss_tail accumulates into it)
__global__ void k_state_rows(const double* __restrict__ mean, int64_t D, double* __restrict__ mean_out, double* __restrict__ corr_out) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < D) { mean_out[j] = mean[j]; corr_out[j] = 0.0; }
}

// dst = (*scale) * src, elementwise; scale: one element of `dtype` on the device (the upstream gradient of the loss)
__global__ void k_scale_by_dev(const double* __restrict__ src, int64_t total, int dtype, const void* __restrict__ scale, double* __restrict__ dst) {
  const double f = dtype == CCZ_F32 ? double(*static_cast<const float*>(scale)) : *static_cast<const double*>(scale);
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) dst[e] = f * src[e];
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

constexpr int64_t kFusedMaxD = 2048;     // per-view width served by the batched step kernels (cholinv.hip)

// true: every view goes through the batched Cholesky + inverse step kernels and the 64-tile batched products;
// false (a view wider than kFusedMaxD -- the metric shape d = 4096 -- or CCZ_LOSS_FUSED=0): the SAME formulas on the
// super-blocked factorization, explicit triangular inverses and the 128-tile fp64 GEMM
bool narrow_ok(const int64_t* dims, int m) {
  for (int a = 0; a < m; ++a)
    if (dims[a] > kFusedMaxD) return false;
  static const int on = [] { const char* e = getenv("CCZ_LOSS_FUSED"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// Gamma (D x D), mean (D) and the loss accumulator tr(A A) (one double, zeroed here) from the moments, for ANY number
// of views (<= LMAXV) of ANY width; info_dev: LMAXV ints (0x7fffffff = factorization succeeded).
// Narrow views: everything is enqueued on the handle's stream and nothing is read back (the pivot flags stay on the
// device).  Wide views: the super-blocked factorization reads its pivot flags on the host and throws CCZ_ENOTSPD
// itself; info_dev is then set to "succeeded" for the caller's device-side check.
void pair_core(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps, bool want_grad,
               double* acc_dev, double* gamma_dev, double* mean_dev, int* info_dev, double* bias_dev = nullptr,
               const GramPartials* gp = nullptr, bool ride_loss = false) {
  // gp != null (mom unused): Ce comes straight from K1's partial sums (k_loss_prep_partials).
  // ride_loss (with want_grad): the loss accumulator is filled by the Gamma_ab stage (tr(A_ab A_ba) = sum_ij M_ab_ij Ce_ab_ij,
  // M_ab = -(n-1)/2 Gamma_ab) instead of a pass of its own over A.
  if (m < 2 || m > LMAXV) fail(CCZ_EUNSUP, "pairwise CCA loss: 2 .. %d views are supported, got %d", LMAXV, m);
  hipStream_t st = stream(c);
  std::vector<int64_t> off(m + 1, 0);
  for (int a = 0; a < m; ++a) off[a + 1] = off[a] + dims[a];
  const int64_t D = off[m];
  const double inv = 1.0 / double(n - 1);
  const bool narrow = narrow_ok(dims, m);
  DBuf Ce(c, D * D), Am(c, D * D);
  std::vector<DBuf> work(m), Lf(m), X(m), T(m), Sinv(m);
  PrepArgs pa{};
  pa.m = m;
  for (int a = 0; a <= m; ++a) pa.off[a] = off[a];
  for (int a = 0; a < m; ++a) {
    const int64_t d = dims[a], nblk = (d + 63) / 64;
    work[a] = DBuf(c, d * d); X[a] = DBuf(c, d * d); Sinv[a] = DBuf(c, d * d);
    if (narrow) { Lf[a] = DBuf(c, d * d); T[a] = DBuf(c, nblk * 4096); }
    pa.work[a] = work[a].get();
  }
  // (split-K destinations of the product stages below are cleared by the same pass -- see there)
  int64_t dmin = dims[0];
  for (int a = 1; a < m; ++a) dmin = std::min(dmin, dims[a]);
  static const int split_env = [] { const char* e = getenv("CCZ_LOSS_SPLITK"); return e ? atoi(e) : 2; }();   // 2: 256 workgroups per two-view stage, half the atomics of 4 (profiles/r05_loss_c4.md)
  const int ks = (narrow && dmin >= 256 && split_env > 1) ? split_env : 1;
  if (ks > 1) {
    for (int a = 0; a < m; ++a) pa.zero_v[a] = Sinv[a].get();
    pa.zero_a = Am.get();
    pa.zero_b = want_grad ? gamma_dev : nullptr;
  }
  if (gp)
    hipLaunchKernelGGL(k_loss_prep_partials, dim3(64, (unsigned)gp->ntiles), dim3(256), 0, st, gp->partial, gp->tiles, gp->ntiles, gp->ksplit,
                       gp->tile_plan, gp->colsum, gp->pilot, D, double(n), inv, eps, Ce.get(), mean_dev, pa, acc_dev, want_grad ? bias_dev : nullptr,
                       gp->msq);
  else
    hipLaunchKernelGGL(k_loss_prep, dim3((unsigned)std::min<int64_t>((D * D + 255) / 256, 4096)), dim3(256), 0, st, mom, mom + D * D, D,
                       1.0 / double(n), inv, eps, Ce.get(), mean_dev, pa, acc_dev, want_grad ? bias_dev : nullptr);
  CCZ_LAUNCH_CHECK();
  if (narrow) {
    std::vector<double*> Ap(m), Lp(m), Xp(m), Tp(m);
    std::vector<int64_t> ld(dims, dims + m);
    for (int a = 0; a < m; ++a) { Ap[a] = work[a].get(); Lp[a] = Lf[a].get(); Xp[a] = X[a].get(); Tp[a] = T[a].get(); }
    cholinv_batched(c, m, Ap.data(), ld.data(), ld.data(), Lp.data(), ld.data(), Xp.data(), ld.data(), Tp.data(), info_dev);
  } else {
    // factor in place (work[a] <- L_a), then X_a = I L_a^-1 by the super-blocked triangular solve
    std::vector<double*> Ap(m);
    std::vector<int64_t> ld(dims, dims + m);
    std::vector<int> info(m, 0);
    for (int a = 0; a < m; ++a) Ap[a] = work[a].get();
    potrf_lower_batched(c, m, Ap.data(), ld.data(), ld.data(), info.data());
    for (int a = 0; a < m; ++a)
      if (info[a] != 0) fail(CCZ_ENOTSPD, "pairwise CCA loss: S_%d%d + eps I is not positive definite (pivot %d)", a + 1, a + 1, info[a] - 1);
    int ok[LMAXV];
    for (int a = 0; a < LMAXV; ++a) ok[a] = 0x7fffffff;
    h2d_small(c, info_dev, ok, sizeof(ok));
    for (int a = 0; a < m; ++a) {
      fill2d(c, dims[a], dims[a], X[a], dims[a], 0.0);
      add_diag(c, dims[a], X[a], dims[a], 1.0);
      trsm_right_lower(c, false, dims[a], dims[a], work[a], dims[a], X[a], dims[a]);
    }
  }
  // Only the blocks that are not trivially known are formed: A_aa = I is never computed, Gamma's diagonal blocks come
  // straight from  Gamma_aa = -sum_{b != a} A_ab Gamma_ba  ( = 2/(n-1) ((A M)_aa - M_aa) ).
  // Narrow views: the four product stages below are dependent launches of at most a few hundred 64 x 64 tiles with
  // K = d: on their own they leave most of the chip idle for ~40 us each.  From d = 256 on the K range of every tile
  // is cut into four slices on separate workgroups that accumulate atomically into zeroed destinations (split-K).
  // (ks, and the clearing of Sinv / Am / Gamma: in k_loss_prep above -- three fills fewer per call.)
  // Wide views: every product is a 128-tile fp64 GEMM of its own (d^3 work fills the chip).
  auto launch = [&](std::vector<MultiGemmArgs>& v) {
    if (!narrow) {
      for (auto& g : v)
        gemm(c, g.tA, g.tB, g.M, g.N, g.K, g.alpha, g.A, g.lda, g.B, g.ldb, g.beta, g.C, g.ldc);
      return;
    }
    if (ks > 1)
      for (auto& g : v) { g.ksplit = ks; g.beta = 1.0; }       // destinations are zero (or hold the earlier terms of a sum)
    for (size_t i0 = 0; i0 < v.size(); i0 += 8) gemm_f64_multi(c, int(std::min<size_t>(8, v.size() - i0)), v.data() + i0);
  };
  std::vector<MultiGemmArgs> pr;
  // Sinv_a = X_a' X_a   (X lower triangular: the K loop starts at the diagonal; the blocks of X above it -- never
  // written by cholinv -- are never read.  Wide: X was built from the identity, its upper part is exact zeros)
  for (int a = 0; a < m; ++a)
    pr.push_back(MultiGemmArgs{X[a], X[a], Sinv[a], nullptr, dims[a], dims[a], dims[a], 0, dims[a], dims[a], dims[a], true, false, false, 1.0, 0.0, narrow});
  launch(pr);
  // A_ab = Sinv_a Ce_ab   (a != b)
  pr.clear();
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b)
      if (a != b)
        pr.push_back(MultiGemmArgs{Sinv[a], Ce.get() + off[a] * D + off[b], Am.get() + off[a] * D + off[b], nullptr, dims[a], D, D, 0,
                                   dims[a], dims[b], dims[a], false, false, false, 1.0, 0.0});
  launch(pr);
  const bool ride = ride_loss && want_grad && narrow;
  if (!ride) {
    hipLaunchKernelGGL(k_trace_sq, dim3((unsigned)std::min<int64_t>((D * D + 255) / 256, 1024)), dim3(256), 0, st, Am.get(), D, pa, acc_dev);
    CCZ_LAUNCH_CHECK();
  }
  if (!want_grad) return;
  // Gamma_ab = -2/(n-1) A_ab Sinv_b   (a != b); with `ride` every tile also adds its share of
  // sum_ij M_ab_ij Ce_ab_ij = tr(A_ab A_ba) to the loss accumulator (M_ab = -(n-1)/2 Gamma_ab)
  pr.clear();
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b)
      if (a != b) {
        pr.push_back(MultiGemmArgs{Am.get() + off[a] * D + off[b], Sinv[b], gamma_dev + off[a] * D + off[b], nullptr, D, dims[b], D, 0,
                                   dims[a], dims[b], dims[b], false, false, false, -2.0 * inv, 0.0});
        if (ride) {
          MultiGemmArgs& g = pr.back();
          g.dotB = Ce.get() + off[a] * D + off[b];
          g.lddot = D;
          g.dot_scale = -0.5 * double(n - 1);
          g.dot_acc = acc_dev;
        }
      }
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
  // The centring row mean' Gamma is NOT formed here: bias_dev is only zeroed by the preparation kernel (it is an accumulation
  // target); every caller that passes bias_dev launches k_loss_tail next, which fills it (there is no k_bias_row
  // any more -- a caller without that tail must compute the row itself).
}

void check_info(ccz_ctx* c, const int* info_dev, int m, const char* what) {
  int got[LMAXV];
  d2h(c, got, info_dev, size_t(m) * sizeof(int));      // synchronises: only the entries that return HOST values use it
  for (int a = 0; a < m; ++a)
    if (got[a] != 0x7fffffff) fail(CCZ_ENOTSPD, "%s: S_%d%d + eps I is not positive definite (pivot %d)", what, a + 1, a + 1, got[a] - 1);
}

// pooled allocation of raw bytes that goes back to the pool on every exit path
struct PoolPtr {
  ccz_ctx* c;
  void* p;
  PoolPtr(ccz_ctx* c_, size_t bytes) : c(c_), p(bytes ? dev_alloc(c_, bytes) : nullptr) {}
  ~PoolPtr() { if (p) dev_free(c, p); }
  PoolPtr(const PoolPtr&) = delete;
  PoolPtr& operator=(const PoolPtr&) = delete;
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

// The sticky failure record of the stream-native loss (hip_common.h::Impl::loss_status): two ints in pinned host
// memory that k_loss_finish writes when a factorization of the batch covariances met a non-positive pivot.
// status[0] = 1 + view index (0: no failure since the last take), status[1] = pivot.
void loss_status_take(ccz_ctx* c, bool synchronise, int* view, int* pivot) {
  Impl* im = impl(c);
  if (view) *view = 0;
  if (pivot) *pivot = 0;
  if (!im->loss_status) return;
  if (synchronise) sync(c);
  volatile int* st = im->loss_status;
  const int v = st[0];
  if (v != 0) {
    if (view) *view = v;
    if (pivot) *pivot = st[1];
    st[0] = 0;
    st[1] = 0;
  }
}

static int* loss_status_dev(ccz_ctx* c) {
  Impl* im = impl(c);
  if (!im->loss_status) {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::memset(hp, 0, 64);
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(hp); return nullptr; }
    im->loss_status = static_cast<int*>(hp);
    im->loss_status_dev = static_cast<int*>(dp);
  }
  return im->loss_status_dev;
}

// ---------------------------------------------------------------------------------------------------------------------
// The pairwise loss in two phases (ccz_pair_loss_forward / ccz_pair_loss_backward; ccz_pair_loss = both with a unit
// upstream gradient).  m = 2 is CCALoss (deep/objectives.py:61-102), m > 2 MCCALoss (:138-153).
//
// forward:  K1 on [z_1 .. z_m] -> pair_core -> loss on the device; with `state`: Gamma (D x D, fp64), the centring row
//           mean' Gamma (row D of the same buffer) and, for fp32 views, Gamma rounded to fp32 behind it.
// backward: dz_a = (*grad_out) sum_b (z_b - mean_b) Gamma_ba from the SAME views -- for two aligned fp32 views ONE product
//           on the fp32 MFMA pipe that reads the views where they lie (gemm_f32_fifo_pair), else one gemm_mixed per block.
// Why two phases: an autograd caller only learns the upstream gradient in its backward; with the gradient formed in the
// forward it had to multiply both n x d gradients by it afterwards (two more passes over them per step).
//
// Launches of a fp32 DCCA batch (the fast path; rounds 2-4 took 32 dispatches for the same work):
//   k_colsum_pilot, k_gram_f32 (partial sums), k_loss_prep_partials, k_cholinv_chain, 4 x k_gemm_f64_multi (the loss value
//   rides on the third), k_loss_tail | k_gemm_f32_nn_fifo2.
// Everything is enqueued on the handle's stream; on the narrow path the host never waits.
// ---------------------------------------------------------------------------------------------------------------------
struct LossShape {
  int64_t dims[LMAXV], off[LMAXV + 1], D;
};

static LossShape loss_shape(const ccz_view* z, int m, int64_t n, int dtype, const char* what) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "%s: dtype must be CCZ_F32 or CCZ_F64", what);
  if (!z) fail(CCZ_EINVAL, "%s: null argument", what);
  if (m < 2 || m > LMAXV) fail(CCZ_EUNSUP, "%s: 2 .. %d views are supported, got %d", what, LMAXV, m);
  if (n < 2) fail(CCZ_EINVAL, "%s: bad shape (at least 2 rows are required)", what);
  LossShape sh{};
  for (int a = 0; a < m; ++a) {
    if (!z[a].data || z[a].cols < 1 || z[a].ld < z[a].cols) fail(CCZ_EINVAL, "%s: bad shape (view %d)", what, a);
    sh.dims[a] = z[a].cols;
    sh.off[a + 1] = sh.off[a] + sh.dims[a];
  }
  sh.D = sh.off[m];
  return sh;
}

// state: [Gamma fp64 (D x D) | centring row mean' Gamma (D) | batch mean (D) | pilot correction (mean - fl32(mean))' Gamma (D) | Gamma fp32 (D x D)]
int64_t pair_loss_state_bytes_impl(int dtype, const int64_t* dims, int m) {
  if (!dims || m < 2 || m > LMAXV) return -1;
  int64_t D = 0;
  for (int a = 0; a < m; ++a) {
    if (dims[a] < 1) return -1;
    D += dims[a];
  }
  (void)dtype;
  return (D + 3) * D * 8 + D * D * 4;
}

void pair_loss_forward_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, double eps, void* loss_dev, void* state) {
  const LossShape sh = loss_shape(z, m, n, dtype, "cca_loss");
  if (!loss_dev) fail(CCZ_EINVAL, "cca_loss: null argument");
  const int64_t D = sh.D;
  hipStream_t st = stream(c);
  const bool narrow = narrow_ok(sh.dims, m);
  const bool want = state != nullptr;
  double* gamma = static_cast<double*>(state);
  double* bias = want ? gamma + D * D : nullptr;
  float* g32 = (want && dtype == CCZ_F32) ? reinterpret_cast<float*>(gamma + (D + 3) * D) : nullptr;
  DBuf mean(c, D), acc(c, 1);
  PoolPtr info(c, LMAXV * sizeof(int));
  // fp32 DCCA batch: K1's partial sums feed the preparation directly (no moments, no gather, no fills, no atomics).
  // Embeddings (post-ReLU, un-normalised) routinely sit far from zero, so the Gram is always pilot-shifted here.
  static const int fast_env = [] { const char* e = getenv("CCZ_LOSS_FAST"); return e ? atoi(e) : 1; }();
  GramPartials gp;
  bool fast = false;
  if (fast_env && narrow && dtype == CCZ_F32) fast = gram_partials_f32(c, z, m, n, &gp);
  struct Release {
    ccz_ctx* c; GramPartials* gp; bool on;
    ~Release() { if (on) gram_partials_release(c, gp); }
  } rel{c, &gp, fast};
  if (fast) {
    pair_core(c, nullptr, n, sh.dims, m, eps, want, acc, gamma, mean, info.as<int>(), bias, &gp, true);
  } else {
    // general route: the moments [G | s] through ccz_moments' machinery.  Wide views (n ~ 1e6 rows x 8192): the automatic
    // pilot choice (one 2 D-double read-back) keeps centred data on the faster FIFO kernel.
    DBuf mom(c, D * D + D);
    moments_impl(c, dtype, z, m, n, true, mom, false, dtype == CCZ_F32 ? (narrow ? 2 : 1) : 0, false);
    pair_core(c, mom, n, sh.dims, m, eps, want, acc, gamma, mean, info.as<int>(), bias, nullptr, narrow);
  }
  if (want) {
    // the batch mean travels with the state (backward: pilot of the split route), and the pilot-correction row is formed with the centring row
    hipLaunchKernelGGL(k_state_rows, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, mean.get(), D, gamma + (D + 1) * D, gamma + (D + 2) * D);
    hipLaunchKernelGGL(k_loss_tail, dim3((unsigned)((D + 63) / 64), (unsigned)((D + 63) / 64)), dim3(256), 0, st, gamma, mean.get(), D, bias, g32,
                       acc.get(), dtype, loss_dev, info.as<int>(), m, loss_status_dev(c), gamma + (D + 2) * D);
  } else {
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1), 0, st, acc.get(), dtype, loss_dev, static_cast<double*>(nullptr), info.as<int>(), m,
                       loss_status_dev(c));
  }
  CCZ_LAUNCH_CHECK();
}

void pair_loss_backward_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, const void* state, const void* grad_out,
                             void* const* g, const int64_t* ldg) {
  const LossShape sh = loss_shape(z, m, n, dtype, "cca_loss backward");
  if (!state || !g || !ldg) fail(CCZ_EINVAL, "cca_loss backward: null argument");
  const int64_t D = sh.D;
  bool any = false, all = true;
  for (int a = 0; a < m; ++a) {
    if (g[a]) {
      if (ldg[a] < sh.dims[a]) fail(CCZ_EINVAL, "cca_loss: bad gradient stride (view %d)", a);
      any = true;
    } else {
      all = false;
    }
  }
  if (!any) return;
  hipStream_t st = stream(c);
  const double* gamma = static_cast<const double*>(state);
  const double* bias = gamma + D * D;
  const float* g32 = reinterpret_cast<const float*>(gamma + (D + 3) * D);
  const double* mean = gamma + (D + 1) * D;
  const double* corr = gamma + (D + 2) * D;
  // a backward that is a large product (the metric shape: n = 1e6, D = 8192) runs on the bf16 pipe with the split arithmetic of K1
  if (dtype == CCZ_F32 && m == 2 && all && c->k1_route != CCZ_K1_FP32 &&
      gemm_split_pair_eligible(n, D, D, sh.dims[0], sh.dims[0], z[0].data, z[0].ld, z[1].data, z[1].ld, g[0], ldg[0], g[1], ldg[1])) {
    try {
      c->last_bwd_route = CCZ_K1_BF16X2;
      gemm_split_pair(c, n, D, D, sh.dims[0], 1.0f, static_cast<const float*>(grad_out), static_cast<const float*>(z[0].data), z[0].ld,
                      static_cast<const float*>(z[1].data), z[1].ld, g32, corr, mean, static_cast<float*>(g[0]), ldg[0], static_cast<float*>(g[1]),
                      ldg[1], sh.dims[0]);
      return;
    } catch (const Error& e) {
      if (e.code != CCZ_ENOMEM) throw;       // no room for the operand planes (what it enqueued so far only wrote scratch): the fp32 product below
    }
  }
  c->last_bwd_route = dtype == CCZ_F32 ? CCZ_K1_FP32 : CCZ_K1_FP64;
  if (dtype == CCZ_F32 && m == 2 && all && narrow_ok(sh.dims, m) &&
      gemm_f32_fifo_pair_eligible(n, D, D, sh.dims[0], sh.dims[0], z[0].data, z[0].ld, z[1].data, z[1].ld, g[0], ldg[0], g[1], ldg[1])) {
    gemm_f32_fifo_pair(c, n, D, D, sh.dims[0], 1.0f, static_cast<const float*>(grad_out), static_cast<const float*>(z[0].data), z[0].ld,
                       static_cast<const float*>(z[1].data), z[1].ld, g32, bias, static_cast<float*>(g[0]), ldg[0], static_cast<float*>(g[1]),
                       ldg[1], sh.dims[0]);
    return;
  }
  // general route: Gamma (and its centring row) scaled by the upstream gradient on the device, one product per block
  DBuf scaled(c, grad_out ? (D + 1) * D : 0);
  if (grad_out) {
    hipLaunchKernelGGL(k_scale_by_dev, dim3((unsigned)std::min<int64_t>(((D + 1) * D + 255) / 256, 2048)), dim3(256), 0, st, gamma, (D + 1) * D,
                       dtype, grad_out, scaled.get());
    CCZ_LAUNCH_CHECK();
    gamma = scaled.get();
    bias = gamma + D * D;
  }
  // dz_a = sum_b (z_b - mean_b) Gamma_ba : the own block first (it carries the bias row of ALL blocks), then the others
  for (int a = 0; a < m; ++a) {
    if (!g[a]) continue;
    gemm_mixed(c, dtype, n, sh.dims[a], sh.dims[a], 1.0, z[a].data, z[a].ld, gamma + sh.off[a] * D + sh.off[a], D, 0.0, g[a], ldg[a],
               bias + sh.off[a]);
    for (int b = 0; b < m; ++b)
      if (b != a)
        gemm_mixed(c, dtype, n, sh.dims[a], sh.dims[b], 1.0, z[b].data, z[b].ld, gamma + sh.off[b] * D + sh.off[a], D, 1.0, g[a], ldg[a], nullptr);
  }
}

// one-shot form: forward + backward with a unit upstream gradient, the state in pooled scratch
void pair_loss_impl(ccz_ctx* c, int dtype, const ccz_view* z, int m, int64_t n, double eps, void* loss_dev, void* const* g,
                    const int64_t* ldg) {
  const LossShape sh = loss_shape(z, m, n, dtype, "cca_loss");
  if (!loss_dev) fail(CCZ_EINVAL, "cca_loss: null argument");
  bool want = false;
  for (int a = 0; a < m; ++a)
    if (g && g[a]) {
      if (!ldg || ldg[a] < sh.dims[a]) fail(CCZ_EINVAL, "cca_loss: bad gradient stride (view %d)", a);
      want = true;
    }
  PoolPtr state(c, want ? size_t(pair_loss_state_bytes_impl(dtype, sh.dims, m)) : 0);
  pair_loss_forward_impl(c, dtype, z, m, n, eps, loss_dev, state.p);
  if (want) pair_loss_backward_impl(c, dtype, z, m, n, state.p, nullptr, g, ldg);
}

void cca_loss_impl(ccz_ctx* c, int dtype, const void* z1, const void* z2, int64_t n, int64_t d1, int64_t d2, int64_t ld1,
                   int64_t ld2, double eps, void* loss_dev, void* g1, void* g2, int64_t ldg1, int64_t ldg2) {
  if (!z1 || !z2 || !loss_dev) fail(CCZ_EINVAL, "cca_loss: null argument");
  if (n < 2 || d1 < 1 || d2 < 1 || ld1 < d1 || ld2 < d2) fail(CCZ_EINVAL, "cca_loss: bad shape");
  if ((g1 && ldg1 < d1) || (g2 && ldg2 < d2)) fail(CCZ_EINVAL, "cca_loss: bad gradient stride");
  const ccz_view z[2] = {{z1, d1, ld1}, {z2, d2, ld2}};
  void* const g[2] = {g1, g2};
  const int64_t ldg[2] = {ldg1, ldg2};
  pair_loss_impl(c, dtype, z, 2, n, eps, loss_dev, (g1 || g2) ? g : nullptr, ldg);
}

// Sum of the pairwise CCA losses of m views from the (all-reduced) batch moments: loss (host) and, if gamma_dev != NULL,
// Gamma (D x D) and the batch mean (D) with  [dz_1 .. dz_m] = ([z_1 .. z_m] - 1 mean') Gamma  for ANY subset of the rows.
void pair_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps,
                            double* loss_host, double* gamma_dev, double* mean_dev) {
  if (!mom || !dims || !loss_host) fail(CCZ_EINVAL, "pair_loss_moments: null argument");
  if (m < 2 || n < 2) fail(CCZ_EINVAL, "pair_loss_moments: at least 2 views and 2 rows are required");
  if (m > LMAXV) fail(CCZ_EUNSUP, "pair_loss_moments: at most %d views", LMAXV);
  for (int a = 0; a < m; ++a)
    if (dims[a] < 1) fail(CCZ_EINVAL, "pair_loss_moments: view %d has no features", a);
  const bool want = gamma_dev != nullptr;
  if (want && !mean_dev) fail(CCZ_EINVAL, "pair_loss_moments: mean_dev is required with gamma_dev");
  int64_t D = 0;
  for (int a = 0; a < m; ++a) D += dims[a];
  DBuf acc(c, 2), mean_tmp(c, want ? 0 : D);
  PoolPtr info(c, LMAXV * sizeof(int));
  pair_core(c, mom, n, dims, m, eps, want, acc, gamma_dev, want ? mean_dev : mean_tmp.get(), info.as<int>());
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(1), 0, stream(c), acc.get(), CCZ_F64, static_cast<void*>(nullptr), acc.get() + 1,
                     static_cast<const int*>(nullptr), 0, static_cast<int*>(nullptr));
  CCZ_LAUNCH_CHECK();
  check_info(c, info.as<int>(), m, "pair_loss_moments");   // the loss goes back to the HOST here: this entry synchronises anyway
  d2h(c, loss_host, acc.get() + 1, 8);
}

void cca_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, int64_t d1, int64_t d2, double eps, double* loss_host,
                           double* gamma_dev, double* mean_dev) {
  const int64_t dims[2] = {d1, d2};
  pair_loss_moments_impl(c, mom, n, dims, 2, eps, loss_host, gamma_dev, mean_dev);
}

}  // namespace ccz
