// TEST DOUBLE for cca_zoo_amd/csrc/ops.h -- host loops, no GPU.
//
// Builds (with g++) into tests/hostsim/libccz_hostsim.so together with the
// unmodified product driver source cca_zoo_amd/csrc/solve.cpp, so that the
// driver LOGIC (Cholesky whitening, Chebyshev subspace iteration, rCCA / MCCA /
// GCCA assembly, error paths) is exercised by the CPU test-suite.  "Device"
// pointers are host pointers here.  Never loaded by the cca_zoo_amd package.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cca_zoo_amd/csrc/ops.h"
#include "../../cca_zoo_amd/csrc/rng_hash.h"

namespace ccz {

void copy2d(ccz_ctx*, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out, int64_t ldo);

void* dev_alloc(ccz_ctx*, size_t bytes) {
  void* p = std::malloc(bytes ? bytes : 8);
  if (!p) fail(CCZ_ENOMEM, "malloc(%zu) failed", bytes);
  return p;
}
void dev_free(ccz_ctx*, void* p) { std::free(p); }
void h2d(ccz_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
void d2h(ccz_ctx*, void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
void d2d(ccz_ctx*, void* d, const void* s, size_t n) { std::memmove(d, s, n); }
void zero(ccz_ctx*, void* d, size_t n) { std::memset(d, 0, n); }
void sync(ccz_ctx*) {}
void wait_deferred(ccz_ctx*) {}
void trace_mark(ccz_ctx*, const char*) {}
void trace_flush(ccz_ctx*, const char*) {}
void activate(ccz_ctx*) {}
int device_current() { return 0; }
void device_set(int) {}

void gemm(ccz_ctx*, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
          int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  std::vector<double> acc(size_t(M) * N, 0.0);
  for (int64_t i = 0; i < M; ++i)
    for (int64_t k = 0; k < K; ++k) {
      const double a = tA ? A[k * lda + i] : A[i * lda + k];
      if (a == 0.0) continue;
      for (int64_t j = 0; j < N; ++j) acc[i * N + j] += a * (tB ? B[j * ldb + k] : B[k * ldb + j]);
    }
  for (int64_t i = 0; i < M; ++i)
    for (int64_t j = 0; j < N; ++j)
      C[i * ldc + j] = alpha * acc[i * N + j] + (beta == 0.0 ? 0.0 : beta * C[i * ldc + j]);
}

void gemm_ex(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* C2,
             int64_t ldc2, bool) {
  gemm(c, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  if (C2) copy2d(c, M, N, C, ldc, C2, ldc2);
}

int potrf_lower(ccz_ctx*, double* A, int64_t d, int64_t lda) {
  for (int64_t j = 0; j < d; ++j) {
    double v = A[j * lda + j];
    for (int64_t t = 0; t < j; ++t) v -= A[j * lda + t] * A[j * lda + t];
    if (!(v > 0.0)) return int(j) + 1;
    const double piv = std::sqrt(v);
    A[j * lda + j] = piv;
    for (int64_t i = j + 1; i < d; ++i) {
      double w = A[i * lda + j];
      for (int64_t t = 0; t < j; ++t) w -= A[i * lda + t] * A[j * lda + t];
      A[i * lda + j] = w / piv;
    }
  }
  return 0;
}

void potrf_lower_batched(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info) {
  for (int b = 0; b < count; ++b) info[b] = potrf_lower(c, A[b], d[b], lda[b]);
}

int potrf_lower_inv(ccz_ctx* c, double* A, int64_t d, int64_t lda, double* Linv, int64_t ldi) {
  const int info = potrf_lower(c, A, d, lda);
  if (info != 0) return info;
  for (int64_t j = 0; j < d; ++j) {              // column j of L^-1 by forward substitution on e_j
    for (int64_t i = 0; i < d; ++i) {
      if (i < j) { Linv[i * ldi + j] = 0.0; continue; }
      double v = i == j ? 1.0 : 0.0;
      for (int64_t t = j; t < i; ++t) v -= A[i * lda + t] * Linv[t * ldi + j];
      Linv[i * ldi + j] = v / A[i * lda + i];
    }
  }
  return 0;
}

void trsm_right_lower(ccz_ctx*, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl,
                      double* X, int64_t ldx) {
  for (int64_t row = 0; row < r; ++row) {
    double* x = X + row * ldx;
    if (trans) {  // x L' = y  <=>  L x' = y'
      for (int64_t i = 0; i < d; ++i) {
        double v = x[i];
        for (int64_t t = 0; t < i; ++t) v -= L[i * ldl + t] * x[t];
        x[i] = v / L[i * ldl + i];
      }
    } else {      // x L = y
      for (int64_t i = d - 1; i >= 0; --i) {
        double v = x[i];
        for (int64_t t = i + 1; t < d; ++t) v -= x[t] * L[t * ldl + i];
        x[i] = v / L[i * ldl + i];
      }
    }
  }
}

void transpose(ccz_ctx*, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out, int64_t ldo) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) out[j * ldo + i] = in[i * ldi + j];
}
void copy2d(ccz_ctx*, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out, int64_t ldo) {
  if (in == out && ldi == ldo) return;
  for (int64_t i = 0; i < rows; ++i) std::memmove(out + i * ldo, in + i * ldi, size_t(cols) * 8);
}
void axpby2d(ccz_ctx*, int64_t rows, int64_t cols, double alpha, double* A, int64_t lda, double beta,
             const double* B, int64_t ldb) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j)
      A[i * lda + j] = alpha * A[i * lda + j] + (beta == 0.0 ? 0.0 : beta * B[i * ldb + j]);
}
void fill2d(ccz_ctx*, int64_t rows, int64_t cols, double* A, int64_t lda, double v) {
  for (int64_t i = 0; i < rows; ++i) std::fill(A + i * lda, A + i * lda + cols, v);
}
void add_diag(ccz_ctx*, int64_t d, double* A, int64_t lda, double v) {
  for (int64_t i = 0; i < d; ++i) A[i * lda + i] += v;
}
void scale_cols(ccz_ctx*, int64_t rows, int64_t cols, double* A, int64_t lda, const double* v, int mode) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) {
      const double f = mode == 0 ? v[j] : (mode == 1 ? 1.0 / v[j] : 1.0 / std::sqrt(v[j]));
      A[i * lda + j] *= f;
    }
}
void mirror_upper(ccz_ctx*, int64_t d, double* A, int64_t lda) {
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = 0; j < i; ++j) A[i * lda + j] = A[j * lda + i];
}
void pack_upper(ccz_ctx*, int64_t d, const double* A, int64_t lda, double* packed) {
  int64_t o = 0;
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = i; j < d; ++j) packed[o++] = A[i * lda + j];
}
void unpack_upper(ccz_ctx*, int64_t d, const double* packed, double* A, int64_t lda) {
  int64_t o = 0;
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = i; j < d; ++j) A[i * lda + j] = packed[o++];
}
void cov_block(ccz_ctx*, const double* G, int64_t D, const double* s, int64_t n, bool centre, double alpha,
               int64_t r0, int64_t rows, int64_t c0, int64_t cols, double* out, int64_t ldo) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) {
      const int64_t gr = r0 + i, gc = c0 + j;
      double v = gr <= gc ? G[gr * D + gc] : G[gc * D + gr];   // upper triangle is authoritative
      if (centre) v -= s[r0 + i] * s[c0 + j] / double(n);
      out[i * ldo + j] = alpha * v;
    }
}
void randn_fill(ccz_ctx*, int64_t rows, int64_t cols, double* A, int64_t lda, uint64_t seed) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) A[i * lda + j] = hash_normal(seed, uint64_t(i * cols + j));
}
void col_sqnorms(ccz_ctx*, int64_t rows, int64_t cols, const double* A, int64_t lda, double* out) {
  std::fill(out, out + cols, 0.0);
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) out[j] += A[i * lda + j] * A[i * lda + j];
}
double norm_inf(ccz_ctx*, int64_t rows, int64_t cols, const double* A, int64_t lda) {
  double best = 0.0;
  for (int64_t i = 0; i < rows; ++i) {
    double s = 0.0;
    for (int64_t j = 0; j < cols; ++j) s += std::fabs(A[i * lda + j]);
    if (!(s <= best)) best = s;   // propagates NaN
  }
  return best;
}

int jacobi_rows(ccz_ctx*, int64_t p, int64_t q, double* W, int64_t ldw, double* Q, int64_t qc, int64_t ldq,
                int max_sweeps) {
  const double tol = 2.220446049250313e-16 * std::sqrt(double(q)) * 4.0;
  double floor2 = 0.0;
  for (int64_t a = 0; a < p; ++a) {
    double al = 0;
    for (int64_t t = 0; t < q; ++t) al += W[a * ldw + t] * W[a * ldw + t];
    floor2 = std::max(floor2, al);
  }
  floor2 *= 1e-28;
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    int64_t rotations = 0;
    for (int64_t a = 0; a < p - 1; ++a)
      for (int64_t b = a + 1; b < p; ++b) {
        double* wa = W + a * ldw;
        double* wb = W + b * ldw;
        double al = 0, be = 0, ga = 0;
        for (int64_t t = 0; t < q; ++t) { al += wa[t] * wa[t]; be += wb[t] * wb[t]; ga += wa[t] * wb[t]; }
        if (!(std::fabs(ga) > tol * std::sqrt(al * be)) || al * be == 0.0 || !(std::min(al, be) > floor2)) continue;
        ++rotations;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int64_t u = 0; u < q; ++u) {
          const double x = wa[u], y = wb[u];
          wa[u] = cs * x - sn * y;
          wb[u] = sn * x + cs * y;
        }
        if (Q) {
          double* qa = Q + a * ldq;
          double* qb = Q + b * ldq;
          for (int64_t u = 0; u < qc; ++u) {
            const double x = qa[u], y = qb[u];
            qa[u] = cs * x - sn * y;
            qb[u] = sn * x + cs * y;
          }
        }
      }
    if (rotations == 0) return sweep;
  }
  fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps", max_sweeps);
}

int64_t trsm_aux_size(ccz_ctx*, int64_t) { return 0; }
void potrf_lower_batched_aux(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                             double* const*) {
  potrf_lower_batched(c, count, A, d, lda, info);
}
bool potrf_lower_batched_aux_rider(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                                   double* const* aux, const TrsmRider*) {
  potrf_lower_batched_aux(c, count, A, d, lda, info, aux);
  return false;                          // the host double never interleaves: the caller solves afterwards
}
void trsm_right_lower_aux(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                          int64_t ldx, const double*) {
  trsm_right_lower(c, trans, r, d, L, ldl, X, ldx);
}

void trsm_right_lower_aux_multi(ccz_ctx* c, int count, bool trans, const int64_t* r, const int64_t* d,
                                const double* const* L, const int64_t* ldl, double* const* X, const int64_t* ldx,
                                const double* const*) {
  for (int b = 0; b < count; ++b) trsm_right_lower(c, trans, r[b], d[b], L[b], ldl[b], X[b], ldx[b]);
}

// Two-sided Jacobi in the device kernel's own formulation (k_syev_small, ops_hip.hip): round-robin tournament,
// all rotation parameters of a round from the current H, then every 2 x 2 block R_a' M R_b and the rows of V'.
int syev_small_max(ccz_ctx*) { return 160; }
int syev_small(ccz_ctx*, const double* A, int64_t d, int64_t lda, double* w, double* Vt, int64_t ldv, int max_sweeps, double tol) {
  if (d < 1 || d > 160) fail(CCZ_EINVAL, "syev_small: 1 <= d <= 160 required, got %lld", (long long)d);
  const int64_t pe = (d + 1) & ~int64_t(1), m1 = pe - 1, np = pe / 2;
  std::vector<double> H(size_t(pe) * pe, 0.0), V(size_t(pe) * pe, 0.0), cs(np), sn(np), tn(np);
  std::vector<int64_t> pa(np), qa(np);
  double hmax = 0.0;
  for (int64_t r = 0; r < d; ++r) {
    V[r * pe + r] = 1.0;
    for (int64_t c = 0; c < d; ++c) {
      const double h = 0.5 * (A[r * lda + c] + A[c * lda + r]);
      if (!(std::fabs(h) <= 1.79769313486231570e308)) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
      H[r * pe + c] = h;
      hmax = std::max(hmax, std::fabs(h));
    }
  }
  const double thr = tol * hmax;
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    int64_t rot = 0;
    for (int64_t round = 0; round < m1; ++round) {
      for (int64_t k = 0; k < np; ++k) {
        int64_t a, b;
        if (k == 0) { a = m1; b = round; } else { a = (round + k) % m1; b = (round - k + m1) % m1; }
        pa[k] = a; qa[k] = b;
        const double hpq = H[a * pe + b];
        double c = 1.0, s = 0.0, t = 0.0;
        if (std::fabs(hpq) > thr) {
          const double al = 0.5 * (H[b * pe + b] - H[a * pe + a]);
          const double r = std::sqrt(al * al + hpq * hpq);
          t = (al >= 0.0 ? hpq : -hpq) / (std::fabs(al) + r);
          c = 1.0 / std::sqrt(1.0 + t * t);
          s = t * c;
          ++rot;
        }
        cs[k] = c; sn[k] = s; tn[k] = t;
      }
      for (int64_t ka = 0; ka < np; ++ka)
        for (int64_t kb = 0; kb < np; ++kb) {
          const double sa = sn[ka], sb = sn[kb];
          if (sa == 0.0 && sb == 0.0) continue;
          double& h00 = H[pa[ka] * pe + pa[kb]];
          double& h01 = H[pa[ka] * pe + qa[kb]];
          double& h10 = H[qa[ka] * pe + pa[kb]];
          double& h11 = H[qa[ka] * pe + qa[kb]];
          const double m00 = h00, m01 = h01, m10 = h10, m11 = h11;
          if (ka == kb) {
            h00 = m00 - tn[ka] * m01; h11 = m11 + tn[ka] * m01; h01 = 0.0; h10 = 0.0;
          } else {
            const double ca = cs[ka], cb = cs[kb];
            const double n00 = cb * m00 - sb * m01, n01 = sb * m00 + cb * m01;
            const double n10 = cb * m10 - sb * m11, n11 = sb * m10 + cb * m11;
            h00 = ca * n00 - sa * n10; h10 = sa * n00 + ca * n10;
            h01 = ca * n01 - sa * n11; h11 = sa * n01 + ca * n11;
          }
        }
      for (int64_t k = 0; k < np; ++k) {
        if (sn[k] == 0.0) continue;
        for (int64_t r = 0; r < d; ++r) {
          double& vp = V[pa[k] * pe + r];
          double& vq = V[qa[k] * pe + r];
          const double x = vp, y = vq;
          vp = cs[k] * x - sn[k] * y;
          vq = sn[k] * x + cs[k] * y;
        }
      }
    }
    if (rot == 0) {
      for (int64_t i = 0; i < d; ++i) {
        w[i] = H[i * pe + i];
        for (int64_t j = 0; j < d; ++j) Vt[i * ldv + j] = V[i * pe + j];
      }
      return sweep;
    }
  }
  fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps (d=%lld)", max_sweeps, (long long)d);
}

// Two-sided BLOCK Jacobi in the device algorithm's own structure (cca_zoo_amd/csrc/evd_block.hip): 32-wide blocks,
// round-robin tournament over the blocks, per pair a 64 x 64 sub-problem rotated CROSS-block only (a full 63-round
// tournament in the first round of every sweep), the accumulated 64 x 64 rotation applied to the pair's rows and
// columns of A and to the rows of V'.  Host loops over a full (uncompressed) matrix.
namespace {
void blk_pair_of(int round, int k, int m1, int& a, int& b) {
  if (k == 0) { a = m1; b = round; return; }
  a = (round + k) % m1;
  b = ((round - k) % m1 + m1) % m1;
}
void blk_elem_pair(bool full, int r, int k, int& p, int& q) {
  if (full) { blk_pair_of(r, k, 63, p, q); return; }
  p = k;
  q = 32 + ((k + r) & 31);
}
}  // namespace
int syev_block(ccz_ctx*, const double* A, int64_t d, int64_t lda, double* w, double* Vrows, int64_t ldv, int max_sweeps) {
  if (d < 1) fail(CCZ_EINVAL, "syev_block: d >= 1 required");
  const int64_t dp = (d + 63) / 64 * 64;
  const int nb = int(dp / 32), np = nb / 2;
  std::vector<double> Aw(size_t(dp) * dp, 0.0), Vt(size_t(dp) * dp, 0.0);
  double hmax = 0.0;
  for (int64_t i = 0; i < dp; ++i) Vt[i * dp + i] = 1.0;
  for (int64_t i = 0; i < d; ++i)
    for (int64_t j = 0; j < d; ++j) {
      const double h = 0.5 * (A[i * lda + j] + A[j * lda + i]);
      if (!(std::fabs(h) <= 1.79769313486231570e308)) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
      Aw[i * dp + j] = h;
      hmax = std::max(hmax, std::fabs(h));
    }
  const double thr = 2.220446049250313e-16 * hmax;
  std::vector<double> S(64 * 64), R(size_t(np) * 64 * 64), T(64 * size_t(dp));
  std::vector<int64_t> idx(size_t(np) * 64);
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    int64_t rot = 0;
    for (int round = 0; round < nb - 1 && hmax > 0.0; ++round) {
      const bool full = round == 0;
      for (int K = 0; K < np; ++K) {
        int ba, bb;
        blk_pair_of(round, K, nb - 1, ba, bb);
        int64_t* ix = idx.data() + size_t(K) * 64;
        for (int i = 0; i < 32; ++i) { ix[i] = int64_t(ba) * 32 + i; ix[32 + i] = int64_t(bb) * 32 + i; }
        double* Rk = R.data() + size_t(K) * 4096;
        for (int i = 0; i < 64; ++i)
          for (int j = 0; j < 64; ++j) { S[i * 64 + j] = Aw[ix[i] * dp + ix[j]]; Rk[i * 64 + j] = i == j ? 1.0 : 0.0; }
        const int nr = full ? 63 : 32;
        for (int r = 0; r < nr; ++r) {
          double cs[32], sn[32];
          int pp[32], qq[32];
          for (int k = 0; k < 32; ++k) {
            blk_elem_pair(full, r, k, pp[k], qq[k]);
            const double hpq = S[pp[k] * 64 + qq[k]];
            cs[k] = 1.0; sn[k] = 0.0;
            if (std::fabs(hpq) > thr) {
              const double al = 0.5 * (S[qq[k] * 64 + qq[k]] - S[pp[k] * 64 + pp[k]]);
              const double rr = std::sqrt(al * al + hpq * hpq);
              const double t = (al >= 0.0 ? hpq : -hpq) / (std::fabs(al) + rr);
              cs[k] = 1.0 / std::sqrt(1.0 + t * t);
              sn[k] = t * cs[k];
              ++rot;
            }
          }
          for (int k = 0; k < 32; ++k) {            // columns of S and of R
            if (sn[k] == 0.0) continue;
            for (int i = 0; i < 64; ++i) {
              double& x = S[i * 64 + pp[k]];
              double& y = S[i * 64 + qq[k]];
              const double a = x, b = y;
              x = cs[k] * a - sn[k] * b; y = sn[k] * a + cs[k] * b;
              double& u = Rk[i * 64 + pp[k]];
              double& v = Rk[i * 64 + qq[k]];
              const double e = u, f = v;
              u = cs[k] * e - sn[k] * f; v = sn[k] * e + cs[k] * f;
            }
          }
          for (int k = 0; k < 32; ++k) {            // rows of S
            if (sn[k] == 0.0) continue;
            for (int j = 0; j < 64; ++j) {
              double& x = S[pp[k] * 64 + j];
              double& y = S[qq[k] * 64 + j];
              const double a = x, b = y;
              x = cs[k] * a - sn[k] * b; y = sn[k] * a + cs[k] * b;
            }
            S[pp[k] * 64 + qq[k]] = 0.0; S[qq[k] * 64 + pp[k]] = 0.0;
          }
        }
        for (int i = 0; i < 64; ++i)                // the pair's own tile: the rotated sub-problem itself
          for (int j = 0; j < 64; ++j) Aw[ix[i] * dp + ix[j]] = 0.5 * (S[i * 64 + j] + S[j * 64 + i]);
      }
      // A <- Rh' A Rh off the pair tiles, V' <- Rh' V'
      for (int K = 0; K < np; ++K) {                // columns of every other pair's rows
        const int64_t* ix = idx.data() + size_t(K) * 64;
        const double* Rk = R.data() + size_t(K) * 4096;
        for (int64_t i = 0; i < dp; ++i) {
          bool own = false;
          for (int t = 0; t < 64; ++t) own = own || ix[t] == i;
          if (own) continue;
          double tmp[64];
          for (int j = 0; j < 64; ++j) {
            double s = 0.0;
            for (int t = 0; t < 64; ++t) s += Aw[i * dp + ix[t]] * Rk[t * 64 + j];
            tmp[j] = s;
          }
          for (int j = 0; j < 64; ++j) Aw[i * dp + ix[j]] = tmp[j];
        }
      }
      for (int K = 0; K < np; ++K) {                // rows (the pair's own tile keeps the rotated sub-problem), then V'
        const int64_t* ix = idx.data() + size_t(K) * 64;
        const double* Rk = R.data() + size_t(K) * 4096;
        for (int i = 0; i < 64; ++i)
          for (int64_t j = 0; j < dp; ++j) {
            double s = 0.0;
            for (int t = 0; t < 64; ++t) s += Rk[t * 64 + i] * Aw[ix[t] * dp + j];
            T[i * dp + j] = s;
          }
        for (int i = 0; i < 64; ++i)
          for (int64_t j = 0; j < dp; ++j) {
            bool own = false;
            for (int t = 0; t < 64; ++t) own = own || ix[t] == j;
            if (!own) Aw[ix[i] * dp + j] = T[i * dp + j];
          }
        for (int i = 0; i < 64; ++i)
          for (int64_t j = 0; j < dp; ++j) {
            double v = 0.0;
            for (int t = 0; t < 64; ++t) v += Rk[t * 64 + i] * Vt[ix[t] * dp + j];
            T[i * dp + j] = v;
          }
        for (int i = 0; i < 64; ++i)
          for (int64_t j = 0; j < dp; ++j) Vt[ix[i] * dp + j] = T[i * dp + j];
      }
    }
    if (rot == 0) {
      for (int64_t i = 0; i < d; ++i) {
        w[i] = Aw[i * dp + i];
        if (Vrows) for (int64_t j = 0; j < d; ++j) Vrows[i * ldv + j] = Vt[i * dp + j];
      }
      return sweep;
    }
  }
  fail(CCZ_ENOCONV, "block Jacobi did not converge in %d sweeps (d=%lld)", max_sweeps, (long long)d);
}

void row_dots(ccz_ctx*, int64_t rows, int64_t cols, const double* A, int64_t lda, const double* B, int64_t ldb,
              double* out) {
  for (int64_t i = 0; i < rows; ++i) {
    double s = 0.0;
    for (int64_t j = 0; j < cols; ++j) s += A[i * lda + j] * B[i * ldb + j];
    out[i] = s;
  }
}
void gather_rows(ccz_ctx*, int64_t rows, int64_t cols, const double* in, int64_t ldi, const int64_t* perm,
                 const double* scale, double* out, int64_t ldo) {
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < cols; ++j) out[i * ldo + j] = in[perm[i] * ldi + j] * (scale ? scale[i] : 1.0);
}

}  // namespace ccz

extern "C" {
int ccz_version(void) { return CCZ_VERSION; }
int ccz_create(ccz_handle* out, int device) {
  if (!out) return CCZ_EINVAL;
  *out = new ccz_ctx();
  (*out)->device = device;
  return CCZ_OK;
}
int ccz_destroy(ccz_handle h) { delete h; return CCZ_OK; }
const char* ccz_last_error(ccz_handle h) { return h ? h->err.c_str() : "null handle"; }

// ---- test doubles of the HIP-side entry points (api.hip / gram.hip): "device" memory is host memory, K1 is a
// plain double loop.  They exist so that the package's HOST LOGIC (estimators, grid search, partial / group CCA)
// can run end to end in the CPU test suite; the package itself never loads this library.
int ccz_set_stream(ccz_handle, void*) { return CCZ_OK; }
int ccz_sync(ccz_handle) { return CCZ_OK; }
int ccz_dev_alloc(ccz_handle h, void** out, size_t bytes) {
  if (!h || !out) return CCZ_EINVAL;
  *out = std::malloc(bytes ? bytes : 8);
  return *out ? CCZ_OK : CCZ_ENOMEM;
}
int ccz_dev_free(ccz_handle, void* p) { std::free(p); return CCZ_OK; }
int ccz_memcpy_h2d(ccz_handle, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); return CCZ_OK; }
int ccz_memcpy_d2h(ccz_handle, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); return CCZ_OK; }
int ccz_memset0(ccz_handle, void* dst, size_t bytes) { std::memset(dst, 0, bytes); return CCZ_OK; }
int ccz_moments_last_ms(ccz_handle, double* g, double* s) { if (g) *g = 0.0; if (s) *s = 0.0; return CCZ_OK; }
// the generator of JointData.sample_device, on the host (rng_hash.h is shared with the device kernel)
int ccz_randn_fill(ccz_handle, int dtype, void* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, int64_t row0,
                   int64_t row_stride, double scale, int accumulate) {
  if (!out || cols < 1 || ld < cols || row_stride < cols || (row_stride & 1)) return CCZ_EINVAL;
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t q = 0; 2 * q < cols; ++q) {
      double n0, n1;
      ccz::hash_normal_pair(seed, (uint64_t(row0 + r) * uint64_t(row_stride)) / 2 + uint64_t(q), n0, n1);
      for (int e = 0; e < 2 && 2 * q + e < cols; ++e) {
        const double v = scale * (e ? n1 : n0);
        if (dtype == CCZ_F32) { float* p = static_cast<float*>(out) + r * ld + 2 * q + e; *p = float((accumulate ? double(*p) : 0.0) + v); }
        else { double* p = static_cast<double*>(out) + r * ld + 2 * q + e; *p = (accumulate ? *p : 0.0) + v; }
      }
    }
  return CCZ_OK;
}
int ccz_moments_last_pilot(ccz_handle, int* used) { if (used) *used = 0; return CCZ_OK; }
// the double multiplies in float64 on the host: one "route"; the setting is stored and echoed like the product's
int ccz_k1_route(ccz_handle h, int route, int* previous) {
  if (!h || route < -1 || route > CCZ_K1_BF16X2) return CCZ_EINVAL;
  if (previous) *previous = h->k1_route;
  if (route >= 0) h->k1_route = route;
  return CCZ_OK;
}
int ccz_moments_last_route(ccz_handle h, int* route, double* split_ms, double* mfma_ms, double* reduce_ms) {
  if (!h) return CCZ_EINVAL;
  if (route) *route = h->last_route;
  if (split_ms) *split_ms = 0.0;
  if (mfma_ms) *mfma_ms = 0.0;
  if (reduce_ms) *reduce_ms = 0.0;
  return CCZ_OK;
}
int ccz_loss_last_route(ccz_handle h, int* f, int* b) { if (!h) return CCZ_EINVAL; if (f) *f = h->last_route; if (b) *b = h->last_bwd_route; return CCZ_OK; }
int ccz_pool_trim(ccz_handle h, size_t* released_bytes) { if (!h) return CCZ_EINVAL; if (released_bytes) *released_bytes = 0; return CCZ_OK; }

int ccz_moments(ccz_handle h, int dtype, const ccz_view* views, int n_views, int64_t n_rows, int /*on_device*/,
                double* mom, int accumulate) {
  if (!h || !views || !mom || n_views < 1 || n_rows < 0) return CCZ_EINVAL;
  if (dtype != CCZ_F32 && dtype != CCZ_F64) { h->err = "dtype must be CCZ_F32 or CCZ_F64"; return CCZ_EUNSUP; }
  int64_t D = 0;
  for (int v = 0; v < n_views; ++v) D += views[v].cols;
  if (!accumulate) std::fill(mom, mom + D * D + D, 0.0);
  std::vector<double> row(static_cast<size_t>(D));
  for (int64_t r = 0; r < n_rows; ++r) {
    int64_t o = 0;
    for (int v = 0; v < n_views; ++v) {
      for (int64_t j = 0; j < views[v].cols; ++j)
        row[o + j] = dtype == CCZ_F32 ? double(static_cast<const float*>(views[v].data)[r * views[v].ld + j])
                                      : static_cast<const double*>(views[v].data)[r * views[v].ld + j];
      o += views[v].cols;
    }
    for (int64_t i = 0; i < D; ++i) {
      const double a = row[i];
      double* g = mom + i * D;
      for (int64_t j = i; j < D; ++j) g[j] += a * row[j];      // upper triangle only, as the device kernel
      mom[D * D + i] += a;
    }
  }
  return CCZ_OK;
}
int ccz_moments_symmetrize(ccz_handle, double* mom, int64_t D) {
  for (int64_t i = 0; i < D; ++i)
    for (int64_t j = 0; j < i; ++j) mom[i * D + j] = mom[j * D + i];
  return CCZ_OK;
}
int ccz_moments_pack(ccz_handle, const double* mom, int64_t D, double* packed) {
  int64_t o = 0;
  for (int64_t i = 0; i < D; ++i)
    for (int64_t j = i; j < D; ++j) packed[o++] = mom[i * D + j];
  std::memcpy(packed + o, mom + D * D, size_t(D) * 8);
  return CCZ_OK;
}
int ccz_moments_unpack(ccz_handle, const double* packed, int64_t D, double* mom) {
  int64_t o = 0;
  for (int64_t i = 0; i < D; ++i)
    for (int64_t j = i; j < D; ++j) mom[i * D + j] = packed[o++];
  std::memcpy(mom + D * D, packed + o, size_t(D) * 8);
  return CCZ_OK;
}
// blocks layout (ccz.h): [diag-block upper triangles | colsum | 1 spare slot || off-diagonal blocks]
static int blocks_copy(bool pack, double* mom, int64_t D, const int64_t* dims, int m, double* packed, int which) {
  if (!mom || !packed || !dims || which < 1 || which > 3) return CCZ_EINVAL;
  std::vector<int64_t> off(m + 1, 0);
  for (int i = 0; i < m; ++i) off[i + 1] = off[i] + dims[i];
  if (off[m] != D) return CCZ_EINVAL;
  int64_t pos = 0;
  for (int i = 0; i < m; ++i)
    for (int64_t r = 0; r < dims[i]; ++r)
      for (int64_t q = r; q < dims[i]; ++q, ++pos)
        if (which & 1) { double& g = mom[(off[i] + r) * D + off[i] + q]; if (pack) packed[pos] = g; else g = packed[pos]; }
  if (which & 1) { if (pack) std::memcpy(packed + pos, mom + D * D, size_t(D) * 8); else std::memcpy(mom + D * D, packed + pos, size_t(D) * 8); }
  pos += D + 1;
  for (int i = 0; i < m; ++i)
    for (int j = i + 1; j < m; ++j)
      for (int64_t r = 0; r < dims[i]; ++r)
        for (int64_t q = 0; q < dims[j]; ++q, ++pos)
          if (which & 2) { double& g = mom[(off[i] + r) * D + off[j] + q]; if (pack) packed[pos] = g; else g = packed[pos]; }
  return CCZ_OK;
}
int ccz_moments_pack_blocks(ccz_handle, const double* mom, int64_t D, const int64_t* dims, int m, double* packed, int which) {
  return blocks_copy(true, const_cast<double*>(mom), D, dims, m, packed, which);
}
int ccz_moments_unpack_blocks(ccz_handle, const double* packed, int64_t D, const int64_t* dims, int m, double* mom, int which, void*) {
  return blocks_copy(false, mom, D, dims, m, const_cast<double*>(packed), which);
}
int ccz_solve_defer(ccz_handle, void*) { return CCZ_OK; }
// the exchange behind the ABI: the double has no transport -- a world of one is a no-op, anything larger is refused
int ccz_comm_unique_id(ccz_handle h, void* id) { if (!h || !id) return CCZ_EINVAL; std::memset(id, 0x5a, 128); return CCZ_OK; }
int ccz_comm_init_rank(ccz_handle h, const void* id, int world, int rank) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return CCZ_EINVAL;
  if (world != 1) { h->err = "the host double has no collective transport (world size 1 only)"; return CCZ_ERCCL; }
  h->last_pilot = 1000;   // marker: communicator present (the double has no Impl)
  return CCZ_OK;
}
int ccz_comm_init_all(ccz_handle* hs, int n) { return (hs && n == 1 && hs[0]) ? ccz_comm_init_rank(hs[0], "", 1, 0) : CCZ_ERCCL; }
int ccz_comm_info(ccz_handle h, int* w, int* r) {
  if (!h) return CCZ_EINVAL;
  const bool on = h->last_pilot == 1000;
  if (w) *w = on ? 1 : 0;
  if (r) *r = on ? 0 : -1;
  return CCZ_OK;
}
int ccz_comm_destroy(ccz_handle h) { if (!h) return CCZ_EINVAL; h->last_pilot = 0; return CCZ_OK; }
int ccz_allreduce_sum_f64(ccz_handle h, double* buf, int64_t count) {
  if (!h || !buf || count < 1) return CCZ_EINVAL;
  if (h->last_pilot != 1000) { h->err = "the handle has no communicator (ccz_comm_init_rank / ccz_comm_init_all)"; return CCZ_EINVAL; }
  return CCZ_OK;
}
int ccz_allreduce_sum_f64_multi(ccz_handle* hs, double* const* bufs, int n, int64_t count) {
  return (hs && bufs && n == 1) ? ccz_allreduce_sum_f64(hs[0], bufs[0], count) : CCZ_ERCCL;
}
// the whole exchange step: pack (blocks layout) -> all-reduce (a world of one) -> unpack, the row count through the head's slot
int ccz_moments_exchange(ccz_handle h, double* mom, int64_t D, const int64_t* dims, int m, int64_t n_local, int64_t* n_total) {
  if (!h || !mom || !dims || !n_total || D < 1 || m < 1 || n_local < 0) return CCZ_EINVAL;
  if (h->last_pilot != 1000) { h->err = "the handle has no communicator (ccz_comm_init_rank / ccz_comm_init_all)"; return CCZ_EINVAL; }
  const int64_t count = D * (D + 1) / 2 + D + 1;
  int64_t n_head = D + 1;
  for (int i = 0; i < m; ++i) n_head += dims[i] * (dims[i] + 1) / 2;
  std::vector<double> packed(size_t(count), 0.0);
  int rc = blocks_copy(true, mom, D, dims, m, packed.data(), 3);
  if (rc != CCZ_OK) return rc;
  packed[size_t(n_head - 1)] = double(n_local);
  rc = ccz_allreduce_sum_f64(h, packed.data(), count);
  if (rc != CCZ_OK) return rc;
  rc = blocks_copy(false, mom, D, dims, m, packed.data(), 3);
  *n_total = int64_t(packed[size_t(n_head - 1)] + 0.5);
  return rc;
}
int ccz_transform(ccz_handle h, int dtype, const void* X, int64_t n, int64_t d, int64_t ld, const double* mean,
                  const double* W, int64_t k, void* out, int64_t ldo) {
  if (!h || !X || !W || !out) return CCZ_EINVAL;
  for (int64_t r = 0; r < n; ++r)
    for (int64_t j = 0; j < k; ++j) {
      double acc = 0.0;
      for (int64_t i = 0; i < d; ++i) {
        const double x = dtype == CCZ_F32 ? double(static_cast<const float*>(X)[r * ld + i]) : static_cast<const double*>(X)[r * ld + i];
        acc += (x - (mean ? mean[i] : 0.0)) * W[i * k + j];
      }
      if (dtype == CCZ_F32) static_cast<float*>(out)[r * ldo + j] = float(acc);
      else static_cast<double*>(out)[r * ldo + j] = acc;
    }
  return CCZ_OK;
}
}
