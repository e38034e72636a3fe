// Internal device-op interface of libccz.
//
// The solver drivers (solve.cpp: Cholesky whitening, Chebyshev-filtered
// subspace iteration, rCCA/MCCA/GCCA assembly) are plain C++ written against
// this header only.  The product implementation is ops_hip.hip (HIP kernels on
// gfx950).  tests/hostsim/ops_host.cpp implements the same interface with
// host loops so that the driver LOGIC can be unit-tested in a container
// without a GPU; that library is test infrastructure and is never loaded by
// the cca_zoo_amd package.
//
// All matrices: row-major float64 in device memory, leading dimension in
// elements.  Functions throw ccz::Error; the C ABI layer converts to codes.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/ccz.h"

struct ccz_ctx {
  int device = 0;
  void* stream = nullptr;  // hipStream_t
  std::string err;
  double last_gram_ms = 0.0;
  double last_colsum_ms = 0.0;
  int last_pilot = 0;      // 1 if the last ccz_moments launch used the pilot-mean (shifted) Gram kernel
  int k1_route = 0;        // CCZ_K1_AUTO / CCZ_K1_FP32 / CCZ_K1_BF16X2 (ccz_k1_route): arithmetic route of fp32 views through K1
  int last_route = 0;      // route the last ccz_moments launch took (CCZ_K1_FP32, CCZ_K1_BF16X2; CCZ_K1_FP64 for fp64 views)
  int last_bwd_route = 0;  // route of the last loss backward product on this handle (CCZ_K1_FP32 / CCZ_K1_BF16X2 / CCZ_K1_FP64; 0: none yet)
  double last_split_ms = 0.0, last_mfma_ms = 0.0, last_reduce_ms = 0.0;   // stages of the last split-route launch (timed calls)
  void* impl = nullptr;    // backend-private (memory pool, events, device props)
};

namespace ccz {

struct Error {
  int code;
  std::string msg;
};
[[noreturn]] void fail(int code, const char* fmt, ...);

// ---- memory (pooled in the handle) ----------------------------------------
void* dev_alloc(ccz_ctx* c, size_t bytes);
void dev_free(ccz_ctx* c, void* p);
void h2d(ccz_ctx* c, void* dst, const void* src, size_t bytes);
void d2h(ccz_ctx* c, void* dst, const void* src, size_t bytes);  // synchronises the stream
void d2d(ccz_ctx* c, void* dst, const void* src, size_t bytes);
void zero(ccz_ctx* c, void* dst, size_t bytes);
void sync(ccz_ctx* c);
void activate(ccz_ctx* c);  // make the handle's device current on the calling thread (every ABI entry)
int device_current();       // the calling thread's current device (-1 when there is none)
void device_set(int dev);

// Every ABI entry runs inside one of these: the handle's device is made current for the call and the
// caller's device is restored afterwards, so a handle on cuda:1 never leaks "current device = 1" into the
// caller's runtime state (PyTorch keeps its own notion of the current device per thread).
struct DeviceScope {
  int prev, mine;
  explicit DeviceScope(ccz_ctx* c) : prev(device_current()), mine(c->device) { activate(c); }
  ~DeviceScope() { if (prev >= 0 && prev != mine) device_set(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

// RAII device buffer of doubles
class DBuf {
 public:
  DBuf() = default;
  DBuf(ccz_ctx* c, int64_t n) : c_(c), n_(n) { p_ = n > 0 ? static_cast<double*>(dev_alloc(c, size_t(n) * 8)) : nullptr; }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) { reset(); c_ = o.c_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  ~DBuf() { reset(); }
  void reset() { if (p_) dev_free(c_, p_); p_ = nullptr; n_ = 0; }
  double* get() const { return p_; }
  operator double*() const { return p_; }
  int64_t size() const { return n_; }
 private:
  ccz_ctx* c_ = nullptr;
  double* p_ = nullptr;
  int64_t n_ = 0;
};

// ---- dense float64 ops ------------------------------------------------------
// C (M x N) = alpha * op(A) * op(B) + beta * C
void gemm(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha,
          const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
          int64_t ldc);
// gemm with extras used by the blocked factorisations:
//   C2 (optional, ldc2): second destination receiving the same values as C
//   lower_only: skip 64x64 output tiles strictly above the diagonal (C square, symmetric update)
void gemm_ex(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha,
             const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
             int64_t ldc, double* C2, int64_t ldc2, bool lower_only);
// in-place lower Cholesky of the d x d leading block (upper part left untouched).
// returns 0, or j+1 if pivot j was not positive (matrix content then undefined).
int potrf_lower(ccz_ctx* c, double* A, int64_t d, int64_t lda);
// Cholesky factor AND its inverse in one go (small d: the Gram of a Cholesky-QR pass):  A = L L' (lower triangle of A authoritative;
// A is destroyed), Linv (d x d, ld ldi) <- L^-1 as a FULL matrix (zeros above the diagonal).  Returns 0, or 1 + the index of
// the first non-positive pivot (Linv undefined then).
int potrf_lower_inv(ccz_ctx* c, double* A, int64_t d, int64_t lda, double* Linv, int64_t ldi);
// the same for `count` independent matrices at once: the 64-column panel factorisations (the
// sequential critical path) of all matrices share one launch per panel step.  info[b] as above.
void potrf_lower_batched(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda,
                         int* info);
// X (r x d) <- X L^-T (trans) or X L^-1 (!trans); L lower triangular d x d
void trsm_right_lower(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl,
                      double* X, int64_t ldx);
// Factor-time by-products for later triangular solves with the same factor.  trsm_aux_size(d): doubles of auxiliary
// storage per matrix (0: this backend / size keeps none).  potrf_lower_batched_aux fills aux[b] (when non-null) next
// to the factor; trsm_right_lower_aux(aux != null) then skips recomputing them.  (HIP backend: the explicit inverses
// of the 512-column diagonal super-blocks, which the super-blocked factorization forms anyway.)
int64_t trsm_aux_size(ccz_ctx* c, int64_t d);
void potrf_lower_batched_aux(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                             double* const* aux);
void trsm_right_lower_aux(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                          int64_t ldx, const double* aux);
// A triangular solve that RIDES ALONG a factorization:  X (r x d[matrix], ld ldx) <- X L^-T  with L the factor of matrix
// `matrix` of the batch, advanced one column block at a time as soon as that block of the factor is final -- the solve's
// large products fill the chip while the factorization's latency chain crawls on a few workgroups.  `prepare` (optional)
// runs once, right before the rider first reads X: the caller's last chance to fill it (and to wait for its source).
// potrf_lower_batched_aux_rider always factors; it returns true if it also performed the rider's solve (then valid iff
// info[matrix] == 0), false if the backend / shape cannot interleave (X untouched, prepare not called).
struct TrsmRider {
  int matrix = -1;
  int64_t r = 0;
  double* X = nullptr;
  int64_t ldx = 0;
  std::function<void()> prepare;
};
bool potrf_lower_batched_aux_rider(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                                   double* const* aux, const TrsmRider* rider);
// The same solve for `count` independent problems with FEW rows each (the back-projections of the k wanted
// directions, r_b = k): their dependent steps advance together in batched launches.  aux[b] as above (a backend may
// fall back to a loop over trsm_right_lower_aux when one is null or the shapes do not suit it).
void trsm_right_lower_aux_multi(ccz_ctx* c, int count, bool trans, const int64_t* r, const int64_t* d,
                                const double* const* L, const int64_t* ldl, double* const* X, const int64_t* ldx,
                                const double* const* aux);
// out (cols x rows) = in (rows x cols)'
void transpose(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out,
               int64_t ldo);
void copy2d(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out,
            int64_t ldo);
// A = alpha * A + beta * B   (elementwise, rows x cols); B may be null when beta == 0
void axpby2d(ccz_ctx* c, int64_t rows, int64_t cols, double alpha, double* A, int64_t lda,
             double beta, const double* B, int64_t ldb);
void fill2d(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, double value);
void add_diag(ccz_ctx* c, int64_t d, double* A, int64_t lda, double value);
// A[:, j] *= v[j]  (mode 0) ;  A[:, j] /= v[j] (mode 1) ; A[:, j] /= sqrt(v[j]) (mode 2)
void scale_cols(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, const double* v,
                int mode);
// make the lower triangle equal to the upper one: A[i][j] = A[j][i] for i > j
void mirror_upper(ccz_ctx* c, int64_t d, double* A, int64_t lda);
// packed[off(i) + j - i] <-> A[i][j] for j >= i, off(i) = i d - i (i - 1) / 2   (row-major upper triangle)
void pack_upper(ccz_ctx* c, int64_t d, const double* A, int64_t lda, double* packed);
void unpack_upper(ccz_ctx* c, int64_t d, const double* packed, double* A, int64_t lda);
// out (rows x cols) = alpha * ( G[r0+i][c0+j] - (centre ? s[r0+i] s[c0+j] / n : 0) );
// G is D x D (ld D); only its UPPER triangle is read (element (i, j), i > j, comes from (j, i)),
// so the moments need not be symmetrised first; s has D entries
void cov_block(ccz_ctx* c, const double* G, int64_t D, const double* s, int64_t n, bool centre,
               double alpha, int64_t r0, int64_t rows, int64_t c0, int64_t cols, double* out,
               int64_t ldo);
// deterministic pseudo-normal fill (counter-based hash; identical on every backend)
void randn_fill(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, uint64_t seed);
// out[j] = sum_i A[i][j]^2   (column squared norms), out has `cols` entries
void col_sqnorms(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda, double* out);
// max_i sum_j |A[i][j]|  (infinity norm) -> host value
double norm_inf(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda);

// One-sided (Hestenes) Jacobi on the ROWS of W (p x q): finds the orthogonal
// rotation sequence J with J W having mutually orthogonal rows, applies the
// same rotations to the rows of Q (p x qc) when Q != null.  Returns sweeps
// used; throws ENOCONV beyond max_sweeps.
int jacobi_rows(ccz_ctx* c, int64_t p, int64_t q, double* W, int64_t ldw, double* Q, int64_t qc,
                int64_t ldq, int max_sweeps);
// Two-sided Jacobi EVD of a small symmetric A (d x d, d <= syev_small_max(c) -- 160 on the HIP backend; only read,
// symmetrised on load):
// w_dev[i] = eigenvalue i (unsorted), row i of Vrows (d x d, ld ldv) = its eigenvector.  Needs no definiteness and no
// shift.  Returns sweeps; throws EINVAL on non-finite input, ENOCONV beyond max_sweeps.  syev_small_max: largest
// supported d (0: the backend has no such kernel).
int syev_small_max(ccz_ctx* c);
// tol: a rotation is applied while |h_pq| > tol max|A| (the default is a full-accuracy solve; a caller that only needs an orthogonal
// basis and Ritz VALUES -- the first Rayleigh-Ritz of the subspace iteration -- passes a larger one and saves the last sweeps;
// V stays orthogonal to rounding whatever tol is: it is a product of exact rotations).
int syev_small(ccz_ctx* c, const double* A, int64_t d, int64_t lda, double* w_dev, double* Vrows, int64_t ldv,
               int max_sweeps, double tol = 2.220446049250313e-16);
// Two-sided BLOCK Jacobi EVD for the sizes above syev_small_max (HIP backend: evd_block.hip -- 32-wide column blocks,
// the pair sub-problems in LDS, every O(d^3) update as 64-wide tiles on the fp64 matrix pipe).  A (d x d) is only read
// and symmetrised on load; w_dev[i] = eigenvalue i (unsorted), row i of Vrows (ld ldv) = its eigenvector.  Returns
// sweeps; EINVAL on non-finite input, ENOCONV beyond max_sweeps.
int syev_block(ccz_ctx* c, const double* A, int64_t d, int64_t lda, double* w_dev, double* Vrows, int64_t ldv,
               int max_sweeps);
// out[i] = dot(A[i,:], B[i,:]) for i < rows
void row_dots(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda,
              const double* B, int64_t ldb, double* out);
// out[i,:] = in[perm[i],:] * scale[i]   (perm, scale are HOST arrays; scale may be null)
void gather_rows(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi,
                 const int64_t* perm_host, const double* scale_host, double* out, int64_t ldo);

// The off-diagonal blocks of the moments may still be in flight (the second half of the sharded exchange): every
// solve driver calls this right before its FIRST read of an off-diagonal block -- after the per-view factorizations,
// which only need the diagonal blocks.  HIP backend: the handle's stream waits for the event registered with
// ccz_solve_defer (once); host backend: nothing to wait for.
void wait_deferred(ccz_ctx* c);

// ---- measurement aid (CCZ_TRACE_PHASES=2) ------------------------------------------
// trace_mark: note a phase boundary on the handle's stream WITHOUT synchronising (HIP backend: an event, a host
// time stamp and a 20 us single-wave kernel that measures the shader clock); trace_flush: wait for the stream and
// print, per phase, device time, host time and the shader clock at its end.  No-ops on the host backend.
void trace_mark(ccz_ctx* c, const char* name);
void trace_flush(ccz_ctx* c, const char* what);

// ---- solver drivers (solve.cpp) used by other translation units -------------
// full symmetric EVD of A (d x d, destroyed): w_host (d, descending), V rows = eigenvectors
int syev_full(ccz_ctx* c, double* A, int64_t d, std::vector<double>& w_host, double* Vrows,
              int64_t ldv);
// SPD solve helpers: given lower Cholesky factor L (d x d), X (d x r) <- (L L')^-1 X
void chol_solve_inplace(ccz_ctx* c, int64_t d, int64_t r, const double* L, int64_t ldl, double* X,
                        int64_t ldx);

}  // namespace ccz
