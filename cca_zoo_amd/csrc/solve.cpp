// libccz solver drivers: everything after the second moments exist.
//
//   moments (G | s)  ->  covariance blocks  ->  Cholesky whitening (ridge / eps floor)
//                    ->  top-k eigen / singular problem in whitened coordinates
//                        (Chebyshev-filtered subspace iteration + Jacobi Rayleigh-Ritz,
//                         full one-sided Jacobi as the robust fallback)
//                    ->  back-projection to weights
//
// Plain C++ over the device-op interface in ops.h (HIP kernels in the product
// build).  Reference semantics (file:line relative to the reference root) are
// cited per driver; the dense NumPy statement of the same maths is
// oracle/gram_form.py.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <vector>

#include "ops.h"

namespace ccz {

void fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error{code, std::string(buf)};
}

namespace {

constexpr double kRankTol = 64.0 * 2.220446049250313e-16;  // relative eigenvalue floor (fallbacks)
constexpr int64_t kDirectMax = 192;                        // full Jacobi below this size
constexpr int kMaxSweeps = 60;

std::vector<int64_t> offsets(const int64_t* dims, int m) {
  std::vector<int64_t> off(m + 1, 0);
  for (int i = 0; i < m; ++i) off[i + 1] = off[i] + dims[i];
  return off;
}

// ---------------------------------------------------------------------------
// full decompositions by one-sided Jacobi
// ---------------------------------------------------------------------------
}  // namespace

// Symmetric EVD.  Three routes by size and definiteness:
//   * d <= syev_small_max (160) and not psd: two-sided Jacobi in one workgroup (Rayleigh-Ritz problems);
//   * psd and 2 d (d | 1) 8 <= 144 KB (d <= 95): the one-workgroup ONE-sided kernel on the rows of A -- the only route that
//     keeps the RELATIVE accuracy of small eigenvalues of a covariance / Gram matrix;
//   * everything else, psd or not: the two-sided block Jacobi (evd_block.hip).  Its eigenvalues carry an ABSOLUTE error of
//     ~0.6e-15 d ||A|| (the refresh of d >= 1536 brings the residual to 1e-14, not the small eigenvalues to relative
//     accuracy): a psd caller that separates "zero" from "small" must do so with an absolute floor well above that --
//     make_whiteners' rank floor is kRankTol d lam_0 = 64 eps d lam_0 = 1.4e-14 d lam_0, a factor ~24 over the noise, and
//     tests/test_gpu_seams_r5.py::test_rank_detection_between_the_kernels holds the detected rank at d = 320 / 256 / 700.
// The legacy route below (CCZ_EVD_LEGACY=1): rows of (A + shift I) orthogonalised, shift = ||A||_inf so that +lam / -lam
// pairs (MCCA with two views has them exactly) cannot mix; psd skips the shift.
static bool legacy_evd() {   // CCZ_EVD_LEGACY=1: rounds 1-3's launch-per-round one-sided Jacobi above d = 160 (A/B)
  static const bool v = [] { const char* e = getenv("CCZ_EVD_LEGACY"); return e && atoi(e) != 0; }();
  return v;
}
static int syev_full_impl(ccz_ctx* c, double* A, int64_t d, bool psd, std::vector<double>& w,
                          double* Vrows, int64_t ldv, double small_tol = 2.220446049250313e-16) {
  if (!psd && d >= 2 && d <= syev_small_max(c)) {
    // Rayleigh-Ritz sized problems: two-sided Jacobi in one workgroup (no shift needed, no definiteness assumed)
    DBuf wd(c, d), V(c, d * d);
    const int sweeps = syev_small(c, A, d, d, wd, V, d, kMaxSweeps, small_tol);
    std::vector<double> lh(d);
    d2h(c, lh.data(), wd, size_t(d) * 8);
    std::vector<int64_t> perm(d);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return lh[a] > lh[b]; });
    w.resize(d);
    for (int64_t i = 0; i < d; ++i) w[i] = lh[perm[i]];
    if (Vrows) gather_rows(c, d, d, V, d, perm.data(), nullptr, Vrows, ldv);
    return sweeps;
  }
  // PSD matrices small enough for the one-workgroup ONE-sided kernel keep it (relative accuracy of tiny eigenvalues)
  const bool lds_one_sided = psd && size_t(2) * d * (d | 1) * 8 <= size_t(144) * 1024;
  if (d >= 2 && !legacy_evd() && !lds_one_sided) {
    // everything wider: two-sided block Jacobi (no shift, no definiteness assumed; A is only read)
    DBuf wd(c, d), V(c, d * d);
    const int sweeps = syev_block(c, A, d, d, wd, V, d, kMaxSweeps);
    std::vector<double> lh(d);
    d2h(c, lh.data(), wd, size_t(d) * 8);
    std::vector<int64_t> perm(d);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return lh[a] > lh[b]; });
    w.resize(d);
    for (int64_t i = 0; i < d; ++i) w[i] = lh[perm[i]];
    if (Vrows) gather_rows(c, d, d, V, d, perm.data(), nullptr, Vrows, ldv);
    return sweeps;
  }
  double shift = 0.0;
  if (!psd) {
    shift = norm_inf(c, d, d, A, d);
    if (!(shift >= 0.0) || !std::isfinite(shift)) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
    shift *= 1.0 + 1e-3;
    if (shift > 0.0) add_diag(c, d, A, d, shift);
  }
  DBuf Q(c, d * d);
  fill2d(c, d, d, Q, d, 0.0);
  add_diag(c, d, Q, d, 1.0);
  int sweeps = jacobi_rows(c, d, d, A, d, Q, d, d, kMaxSweeps);
  DBuf lam(c, d);
  row_dots(c, d, d, A, d, Q, d, lam);
  std::vector<double> lh(d);
  d2h(c, lh.data(), lam, size_t(d) * 8);
  for (auto& v : lh) v -= shift;
  std::vector<int64_t> perm(d);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return lh[a] > lh[b]; });
  w.resize(d);
  for (int64_t i = 0; i < d; ++i) w[i] = lh[perm[i]];
  if (Vrows) gather_rows(c, d, d, Q, d, perm.data(), nullptr, Vrows, ldv);
  return sweeps;
}

int syev_full(ccz_ctx* c, double* A, int64_t d, std::vector<double>& w, double* Vrows, int64_t ldv) {
  return syev_full_impl(c, A, d, false, w, Vrows, ldv);
}

namespace {

// Thin SVD of A (p x q) by one-sided Jacobi.  Returns row-form factors:
// Ut (r x p) rows = left vectors, s (r, descending, host), Vt (r x q).
int gesvj_rows(ccz_ctx* c, const double* A, int64_t p, int64_t q, int64_t lda, double* Ut,
               std::vector<double>& s, double* Vt) {
  const bool tall = p > q;
  const int64_t r = tall ? q : p, l = tall ? p : q;
  DBuf W(c, r * l);
  if (tall) transpose(c, p, q, A, lda, W, l); else copy2d(c, p, q, A, lda, W, l);
  DBuf Q(c, r * r);
  fill2d(c, r, r, Q, r, 0.0);
  add_diag(c, r, Q, r, 1.0);
  int sweeps = jacobi_rows(c, r, l, W, l, Q, r, r, kMaxSweeps);
  DBuf nn(c, r);
  row_dots(c, r, l, W, l, W, l, nn);
  std::vector<double> nh(r);
  d2h(c, nh.data(), nn, size_t(r) * 8);
  std::vector<int64_t> perm(r);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return nh[a] > nh[b]; });
  s.resize(r);
  std::vector<double> inv(r);
  const double tiny = nh.empty() ? 0.0 : std::max(nh[perm[0]], 0.0) * 1e-300;
  for (int64_t i = 0; i < r; ++i) {
    double v = std::sqrt(std::max(nh[perm[i]], 0.0));
    s[i] = v;
    inv[i] = (v > tiny && v > 0.0) ? 1.0 / v : 0.0;
  }
  // long side: normalised rows of W; short side: rows of Q
  double* longf = tall ? Ut : Vt;
  double* shortf = tall ? Vt : Ut;
  if (longf) gather_rows(c, r, l, W, l, perm.data(), inv.data(), longf, l);
  if (shortf) gather_rows(c, r, r, Q, r, perm.data(), nullptr, shortf, r);
  return sweeps;
}

// ---------------------------------------------------------------------------
// orthonormalisation of the columns of X (p x b): Cholesky-QR, two passes
// ---------------------------------------------------------------------------
bool cholqr_pass(ccz_ctx* c, int64_t p, int64_t b, double* X, int64_t ldx, double rel_shift) {
  DBuf Gm(c, b * b);
  gemm(c, true, false, b, b, p, 1.0, X, ldx, X, ldx, 0.0, Gm, b);
  if (rel_shift > 0.0) {
    std::vector<double> gh(size_t(b) * b);
    d2h(c, gh.data(), Gm, gh.size() * 8);
    double tr = 0.0;
    for (int64_t i = 0; i < b; ++i) tr += gh[i * b + i];
    add_diag(c, b, Gm, b, rel_shift * tr / double(b));
  }
  // factor and inverse from one launch (the chain kernel forms L^-1 next to L), then X <- X L^-T as ONE product: the
  // triangular solve of a b x b factor was an inverse kernel + three small products (170 -> 110 us per pass at b = 80)
  DBuf Li(c, b * b), T(c, p * b);
  if (potrf_lower_inv(c, Gm, b, b, Li, b) != 0) return false;
  gemm(c, false, true, p, b, b, 1.0, X, ldx, Li, b, 0.0, T, b);
  copy2d(c, p, b, T, b, X, ldx);
  return true;
}

void orthonormalize(ccz_ctx* c, int64_t p, int64_t b, double* X, int64_t ldx, bool refine = true) {
  // pass 1 may meet a numerically singular Gram (filtered block nearly rank deficient):
  // retry with a growing diagonal shift; pass 2/3 restore orthogonality.
  double shift = 0.0;
  for (int attempt = 0; attempt < 6; ++attempt) {
    if (cholqr_pass(c, p, b, X, ldx, shift)) {
      if (!refine && shift == 0.0) return;            // a block that is only being kept from collapsing (power steps)
      if (!cholqr_pass(c, p, b, X, ldx, 0.0)) { shift = shift > 0 ? shift * 100 : 1e-14; continue; }
      if (shift > 0.0 && !cholqr_pass(c, p, b, X, ldx, 0.0)) { shift *= 100; continue; }
      return;
    }
    shift = shift > 0 ? shift * 100 : 1e-14;
  }
  fail(CCZ_ENOCONV, "orthonormalisation failed (block of %lld vectors is rank deficient)", (long long)b);
}

// ---------------------------------------------------------------------------
// top-k eigenpairs (largest algebraic) of a symmetric operator
// ---------------------------------------------------------------------------
struct SymOp {
  int64_t p;
  double lower;  // a proven lower bound of the spectrum
  std::function<void(const double* X, int64_t ldx, int64_t b, double* Y, int64_t ldy)> apply;
};

struct RitzState {
  std::vector<double> theta;   // b Ritz values, descending
  std::vector<double> resid;   // b residual norms
};

// Rayleigh-Ritz on span(X): X <- X C, Y <- (S X) C, theta, residual norms.  The three block pointers rotate (the
// products land in scratch blocks that then become X / Y): no block copies.
// jac_tol: the Jacobi threshold of the b x b eigen-solve (see syev_small).  The FIRST Rayleigh-Ritz of a subspace iteration only
// sizes the filter (Ritz values: their error is the SQUARE of what is left off the diagonal) and rotates the basis (orthogonal
// whatever the threshold); the residuals below are those of the pairs actually formed, so acceptance never rests on it.
void rayleigh_ritz(ccz_ctx* c, const SymOp& op, int64_t b, double*& X, double*& Y, double*& tmp,
                   RitzState& st, double jac_tol = 2.220446049250313e-16) {
  const int64_t p = op.p;
  op.apply(X, b, b, Y, b);
  DBuf H(c, b * b), Hs(c, b * b), Vr(c, b * b);
  gemm(c, true, false, b, b, p, 1.0, X, b, Y, b, 0.0, H, b);
  transpose(c, b, b, H, b, Hs, b);
  axpby2d(c, b, b, 0.5, H, b, 0.5, Hs, b);            // symmetrise
  const int jsw = syev_full_impl(c, H, b, false, st.theta, Vr, b, jac_tol);      // rows of Vr = Ritz coefficient vectors
  static const bool trace_rr = getenv("CCZ_TRACE_SOLVER") != nullptr;
  if (trace_rr) fprintf(stderr, "[ccz] rayleigh_ritz b=%lld: %d Jacobi sweeps at threshold %.1e\n", (long long)b, jsw, jac_tol);
  gemm(c, false, true, p, b, b, 1.0, X, b, Vr, b, 0.0, tmp, b);
  std::swap(X, tmp);                                    // X = X C ; tmp = old X (free)
  gemm(c, false, true, p, b, b, 1.0, Y, b, Vr, b, 0.0, tmp, b);
  std::swap(Y, tmp);                                    // Y = Y C ; tmp free
  // residual R = Y - X diag(theta)
  DBuf th(c, b), rn(c, b);
  h2d(c, th, st.theta.data(), size_t(b) * 8);
  d2d(c, tmp, X, size_t(p) * b * 8);
  scale_cols(c, p, b, tmp, b, th, 0);
  axpby2d(c, p, b, -1.0, tmp, b, 1.0, Y, b);          // tmp = Y - X theta
  col_sqnorms(c, p, b, tmp, b, rn);
  st.resid.resize(b);
  d2h(c, st.resid.data(), rn, size_t(b) * 8);
  for (auto& v : st.resid) v = std::sqrt(std::max(v, 0.0));
}

// Degree-m Chebyshev filter damping [a, cut], scaled at aL (Zhou & Saad).
// In: X (p x b).  Out: result left in X.  Y, Z are work blocks.
void chebyshev_filter(ccz_ctx* c, const SymOp& op, int64_t b, int m, double a, double cut,
                      double aL, double*& X, double*& Y, double*& Z) {
  // three-term recurrence on rotating block pointers (the result ends up in X; Y, Z hold scratch): no block copies
  const int64_t p = op.p;
  const double e = 0.5 * (cut - a), c0 = 0.5 * (cut + a);
  double sigma1 = e / (aL - c0);
  double sigma = sigma1;
  const double tau = 2.0 / sigma1;
  op.apply(X, b, b, Y, b);
  axpby2d(c, p, b, sigma1 / e, Y, b, -c0 * sigma1 / e, X, b);   // Y = (S X - c0 X) sigma1/e
  for (int i = 2; i <= m; ++i) {
    const double sn = 1.0 / (tau - sigma);
    op.apply(Y, b, b, Z, b);
    axpby2d(c, p, b, 2.0 * sn / e, Z, b, -c0 * 2.0 * sn / e, Y, b);
    axpby2d(c, p, b, 1.0, Z, b, -sigma * sn, X, b);              // Z = 2 sn/e (S Y - c0 Y) - sigma sn X
    double* t = X; X = Y; Y = Z; Z = t;                          // (X, Y) <- (Y, Z)
    sigma = sn;
  }
  std::swap(X, Y);                                               // the last iterate is the result
}

// Returns theta (k, descending) on the host and the eigenvectors as the first k
// COLUMNS of Xout (p x k, ld k).
void topk_symmetric(ccz_ctx* c, const SymOp& op, int k, std::vector<double>& theta, double* Xout) {
  const int64_t p = op.p;
  if (k > p) k = int(p);
  int64_t b = std::min<int64_t>(p, k + std::max(8, k / 4));
  DBuf Xb(c, p * b), Yb(c, p * b), Zb(c, p * b);
  double *X = Xb, *Y = Yb, *Z = Zb;                  // rotating block pointers (rayleigh_ritz, chebyshev_filter)
  randn_fill(c, p, b, X, b, 0x9E3779B97F4A7C15ull);
  // warm start: two power steps on the shifted operator S - lower I (positive semi-definite, so the
  // top of the spectrum dominates even for indefinite S); a random block has Ritz values that all sit
  // at the spectral mean and would tell the first filter nothing about the gap.  A Gaussian block is
  // well conditioned as it is (singular values within 1 +- sqrt(b/p) of sqrt(p)): it is not orthonormalised first.
  if (b < p) {
    for (int it = 0; it < 2; ++it) {
      op.apply(X, b, b, Y, b);
      if (op.lower != 0.0) axpby2d(c, p, b, 1.0, Y, b, -op.lower, X, b);
      std::swap(X, Y);
      orthonormalize(c, p, b, X, b, /*refine=*/it == 1);   // only the block that enters Rayleigh-Ritz must be orthonormal
    }
  } else {
    orthonormalize(c, p, b, X, b);
  }
  RitzState st;
  // the first Rayleigh-Ritz feeds the filter design, not the answer (unless the block spans everything): a Jacobi threshold of
  // 1e-9 leaves Ritz values good to 1e-18 of the scale and saves the last one or two of its ~8 sweeps (CCZ_RR1_TOL; 0 = full)
  static const double rr1_tol = [] { const char* e = getenv("CCZ_RR1_TOL"); return e ? atof(e) : 1e-9; }();
  rayleigh_ritz(c, op, b, X, Y, Z, st, (b < p && rr1_tol > 2.220446049250313e-16) ? rr1_tol : 2.220446049250313e-16);
  const double tol = 1e-11;
  const int max_cycles = 200;
  double prev_worst = 1e300;
  int boost = 0;
  for (int cycle = 0; cycle < max_cycles; ++cycle) {
    const double scale = std::max({std::fabs(st.theta.front()), std::fabs(st.theta.back()), 1e-300});
    double worst = 0.0;
    for (int i = 0; i < k; ++i) worst = std::max(worst, st.resid[i]);
    if (worst <= tol * scale) {
      static const bool trace_done = getenv("CCZ_TRACE_SOLVER") != nullptr;
      if (trace_done)
        fprintf(stderr, "[ccz] topk p=%lld k=%d b=%lld done after %d cycle(s): worst/scale %.3e\n", (long long)p, k, (long long)b, cycle, worst / scale);
      theta.assign(st.theta.begin(), st.theta.begin() + k);
      copy2d(c, p, k, X, b, Xout, k);
      return;
    }
    if (b == p) break;  // the block spans everything: Rayleigh-Ritz was already exact
    if (worst > 0.1 * prev_worst) boost += 4;   // slow progress -> raise the filter degree
    prev_worst = worst;
    double aL = st.theta.front();
    double cut = st.theta.back();
    double a = std::min(op.lower, cut - 1e-3 * std::max(aL - cut, 1e-12 * scale));
    if (!(aL > cut)) aL = cut + 1e-8 * scale;
    if (!(cut > a)) cut = a + 1e-8 * scale;
    // degree from the gap: a degree-m filter damps the unwanted part relative to the k-th wanted
    // value by ~ 1 / cosh(m acosh(x)), x = (theta_k - centre) / half-width of the damped interval
    const double e = 0.5 * (cut - a), c0 = 0.5 * (cut + a);
    const double x = (st.theta[k - 1] - c0) / e;
    int degree = 30;
    if (x > 1.0 + 1e-12) {
      // x1000 margin: the cosh estimate is optimistic by ~10x in practice, and landing a hair above the tolerance
      // costs a whole extra filter + orthonormalise + Rayleigh-Ritz cycle (one more degree costs two GEMMs)
      static const double margin = [] { const char* e = getenv("CCZ_CHEB_MARGIN"); return e ? atof(e) : 1e3; }();
      const double need = std::max(worst / (tol * scale), 2.0) * margin;
      degree = int(std::ceil(std::acosh(need) / std::acosh(x)));
    }
    static const int max_degree = [] { const char* e = getenv("CCZ_CHEB_MAXDEG"); return e ? atoi(e) : 40; }();
    // The FIRST filter is sized from the Ritz values of two power steps: where the wanted end of the spectrum is dense (x close to
    // 1: MCCA 4 x 2048 x = 1.13, GCCA D = 16384 x = 1.09) those underestimate the gap badly -- the estimate asked for 63 / 70 degrees,
    // the cap of 40 converged both to 4e-14 / 9e-14 in one cycle, and so did 22 (MCCA) and 26 (GCCA: 9e-12), measured in round 6
    // (tools/solve_probe.py with CCZ_TRACE_SOLVER=1 and CCZ_CHEB_MAXDEG).  A filter that falls short costs one more, cheap cycle
    // (accurate Ritz values then: degree 7 at GCCA's shape, 157 ms against 166 with the old cap) -- so the first cycle is capped lower.
    static const int first_cap = [] { const char* e = getenv("CCZ_CHEB_FIRSTCAP"); return e ? atoi(e) : 28; }();
    const int cap = cycle == 0 ? std::min(max_degree, std::max(2, first_cap)) : max_degree;
    degree = std::min(cap, std::max(2, degree) + boost);
    static const bool trace = getenv("CCZ_TRACE_SOLVER") != nullptr;
    if (trace)
      fprintf(stderr, "[ccz] topk p=%lld k=%d b=%lld cycle %d: worst/scale %.3e  x %.6f  degree %d\n", (long long)p, k,
              (long long)b, cycle, worst / scale, x, degree);
    chebyshev_filter(c, op, b, degree, a, cut, aL, X, Y, Z);
    orthonormalize(c, p, b, X, b);
    rayleigh_ritz(c, op, b, X, Y, Z, st);
  }
  double worst = 0.0;
  for (int i = 0; i < k; ++i) worst = std::max(worst, st.resid[i]);
  const double scale = std::max({std::fabs(st.theta.front()), std::fabs(st.theta.back()), 1e-300});
  if (b == p || worst <= 1e-8 * scale) {
    theta.assign(st.theta.begin(), st.theta.begin() + k);
    copy2d(c, p, k, X, b, Xout, k);
    return;
  }
  fail(CCZ_ENOCONV, "subspace iteration stalled (residual %.3e, scale %.3e)", worst, scale);
}

// top-k eigenpairs of a dense symmetric S (p x p): rows-form output Vt (k x p).
// lower_hint: a PROVEN lower bound of the spectrum if the caller has one (the filter damps [lower, cut]: the generic
// bound -||S||_inf is often 10-100x too pessimistic, and the filter degree grows like 1 / sqrt(gap / interval)).
void eig_topk_dense(ccz_ctx* c, const double* S, int64_t p, int k, std::vector<double>& lam,
                    double* Vt, int64_t ldvt, double lower_hint = -HUGE_VAL) {
  k = int(std::min<int64_t>(k, p));
  bool direct = p <= kDirectMax || int64_t(k) * 3 >= p;
  if (!direct) {
    SymOp op;
    op.p = p;
    op.lower = std::isfinite(lower_hint) ? lower_hint : -norm_inf(c, p, p, S, p);
    op.apply = [c, S, p](const double* X, int64_t ldx, int64_t b, double* Y, int64_t ldy) {
      gemm(c, false, false, p, b, p, 1.0, S, p, X, ldx, 0.0, Y, ldy);
    };
    try {
      DBuf Xk(c, p * k);
      topk_symmetric(c, op, k, lam, Xk);
      transpose(c, p, k, Xk, k, Vt, ldvt);
      return;
    } catch (const Error& e) {
      if (e.code != CCZ_ENOCONV) throw;
      direct = true;  // robust fallback below
    }
  }
  DBuf A(c, p * p), V(c, p * p);
  d2d(c, A, S, size_t(p) * p * 8);
  std::vector<double> w;
  syev_full_impl(c, A, p, false, w, V, p);
  lam.assign(w.begin(), w.begin() + k);
  copy2d(c, k, p, V, p, Vt, ldvt);
}

// top-k singular triplets of T given as Tt = T' (q x p, row-major, ld p) -- i.e. T is p x q.
// Row-form outputs: Ut (k x p), Vt (k x q), s (k, host).
void svd_topk_dense(ccz_ctx* c, const double* Tt, int64_t p, int64_t q, int k,
                    std::vector<double>& s, double* Ut, double* Vt) {
  const int64_t r = std::min(p, q);
  k = int(std::min<int64_t>(k, r));
  bool direct = r <= kDirectMax || int64_t(k) * 3 >= r;
  if (!direct) {
    // eigenproblem of T T' (p x p) if p <= q else T' T (q x q); M = matrix whose rows index the
    // small side: small side p: S = T T' = Tt' Tt ; small side q: S = T' T = Tt Tt'
    const bool small_p = p <= q;
    const int64_t ns = small_p ? p : q, nl = small_p ? q : p;
    SymOp op;
    op.p = ns;
    op.lower = 0.0;
    DBuf work(c, nl * (k + std::max(8, k / 4)));
    double* wk = work;
    op.apply = [c, Tt, p, q, small_p, nl, ns, wk](const double* X, int64_t ldx, int64_t b, double* Y, int64_t ldy) {
      if (small_p) {  // Y = Tt' (Tt X):  Tt is q x p
        gemm(c, false, false, q, b, p, 1.0, Tt, p, X, ldx, 0.0, wk, b);
        gemm(c, true, false, p, b, q, 1.0, Tt, p, wk, b, 0.0, Y, ldy);
      } else {        // Y = Tt (Tt' X)
        gemm(c, true, false, p, b, q, 1.0, Tt, p, X, ldx, 0.0, wk, b);
        gemm(c, false, false, q, b, p, 1.0, Tt, p, wk, b, 0.0, Y, ldy);
      }
      (void)nl; (void)ns;
    };
    try {
      std::vector<double> lam;
      DBuf Xk(c, ns * k);
      topk_symmetric(c, op, k, lam, Xk);
      s.resize(k);
      std::vector<double> inv(k);
      for (int i = 0; i < k; ++i) {
        s[i] = std::sqrt(std::max(lam[i], 0.0));
        inv[i] = s[i] > 0.0 ? 1.0 / s[i] : 0.0;
      }
      DBuf other(c, nl * k), invd(c, k);
      h2d(c, invd, inv.data(), size_t(k) * 8);
      if (small_p) {  // Xk = U (p x k); V = T' U / s = Tt U / s  (q x k)
        gemm(c, false, false, q, k, p, 1.0, Tt, p, Xk, k, 0.0, other, k);
        scale_cols(c, q, k, other, k, invd, 0);
        transpose(c, p, k, Xk, k, Ut, p);
        transpose(c, q, k, other, k, Vt, q);
      } else {        // Xk = V (q x k); U = T V / s = Tt' V / s (p x k)
        gemm(c, true, false, p, k, q, 1.0, Tt, p, Xk, k, 0.0, other, k);
        scale_cols(c, p, k, other, k, invd, 0);
        transpose(c, q, k, Xk, k, Vt, q);
        transpose(c, p, k, other, k, Ut, p);
      }
      return;
    } catch (const Error& e) {
      if (e.code != CCZ_ENOCONV) throw;
      direct = true;
    }
  }
  // full Jacobi SVD of Tt (q x p):  Tt = A B' with left (q-side) = V of T, right (p-side) = U of T
  const int64_t rr = std::min(p, q);
  DBuf Lq(c, rr * q), Rp(c, rr * p);
  std::vector<double> sv;
  gesvj_rows(c, Tt, q, p, p, Lq, sv, Rp);
  s.assign(sv.begin(), sv.begin() + k);
  copy2d(c, k, q, Lq, q, Vt, q);
  copy2d(c, k, p, Rp, p, Ut, p);
}

// ---------------------------------------------------------------------------
// whitening factors  F (d x r),  F' R F = I
// ---------------------------------------------------------------------------
struct Whitener {
  int64_t d = 0, r = 0;
  bool chol = true;
  DBuf L;    // chol: lower factor (d x d) ; else explicit F (d x r)
  DBuf aux;  // chol: by-products of the factorization that later triangular solves reuse (trsm_aux_size; may be empty)

  // X (m x d, ld ldx)  ->  X F   in place for chol (r == d); explicit: into out (m x r)
  void right_apply(ccz_ctx* c, int64_t m, double* X, int64_t ldx, double* out, int64_t ldo) const {
    if (chol) {
      trsm_right_lower_aux(c, true, m, d, L, d, X, ldx, aux.get());
      if (out != X) copy2d(c, m, d, X, ldx, out, ldo);
    } else {
      gemm(c, false, false, m, r, d, 1.0, X, ldx, L, r, 0.0, out, ldo);
    }
  }
  // Ut (k x r) -> (F U)' = Ut F'  (k x d)
  void back_project_rows(ccz_ctx* c, int64_t k, double* Ut, int64_t ldu, double* out, int64_t ldo) const {
    if (chol) {
      trsm_right_lower_aux(c, false, k, d, L, d, Ut, ldu, aux.get());
      if (out != Ut) copy2d(c, k, d, Ut, ldu, out, ldo);
    } else {
      gemm(c, false, true, k, d, r, 1.0, Ut, ldu, L, r, 0.0, out, ldo);
    }
  }
};

// Ut_i (k x r_i, ld ldu_i) <- Ut_i F_i'  for several whiteners at once, in place (Cholesky whiteners: r_i = d_i; the
// dependent super-block steps of all views share their launches).  Eigen-floored whiteners have an explicit F of
// another shape and go one by one into out_i.
void back_project_rows_multi(ccz_ctx* c, const std::vector<const Whitener*>& F, int64_t k, const std::vector<double*>& Ut,
                             const std::vector<int64_t>& ldu) {
  std::vector<int64_t> rr, dd, ll, lx;
  std::vector<const double*> Lp, Ap;
  std::vector<double*> Xp;
  for (size_t i = 0; i < F.size(); ++i) {
    if (!F[i]->chol) fail(CCZ_EINVAL, "back_project_rows_multi: Cholesky whiteners only");
    rr.push_back(k); dd.push_back(F[i]->d); ll.push_back(F[i]->d); lx.push_back(ldu[i]);
    Lp.push_back(F[i]->L.get()); Ap.push_back(F[i]->aux.get()); Xp.push_back(Ut[i]);
  }
  for (size_t b0 = 0; b0 < F.size(); b0 += 8) {
    const int cnt = int(std::min<size_t>(8, F.size() - b0));
    trsm_right_lower_aux_multi(c, cnt, false, rr.data() + b0, dd.data() + b0, Lp.data() + b0, ll.data() + b0, Xp.data() + b0,
                               lx.data() + b0, Ap.data() + b0);
  }
}

// R_i (d_i x d_i, consumed) -> Whiteners; the Cholesky panel chains of all blocks run batched.
// allow_floor: a block whose Cholesky fails falls back to the eigen-floored explicit factor
// (rCCA c = 0 on rank-deficient data); otherwise ENOTSPD.
// rider (optional): a triangular solve with one of the factors that the backend may interleave with the factorization
// (ops.h::TrsmRider); *rode tells whether it did.
std::vector<Whitener> make_whiteners(ccz_ctx* c, std::vector<DBuf>& R, const std::vector<int64_t>& dims, bool allow_floor,
                                     const TrsmRider* rider = nullptr, bool* rode = nullptr) {
  const int m = int(R.size());
  std::vector<DBuf> keep(m);
  std::vector<double*> ptr(m);
  std::vector<int64_t> ld(dims);
  std::vector<int> info(m, 0);
  std::vector<DBuf> aux(m);
  std::vector<double*> auxp(m, nullptr);
  for (int i = 0; i < m; ++i) {
    if (allow_floor) { keep[i] = DBuf(c, dims[i] * dims[i]); d2d(c, keep[i], R[i], size_t(dims[i]) * dims[i] * 8); }
    ptr[i] = R[i].get();
    const int64_t na = trsm_aux_size(c, dims[i]);
    if (na > 0) { aux[i] = DBuf(c, na); auxp[i] = aux[i].get(); }
  }
  const bool did = potrf_lower_batched_aux_rider(c, m, ptr.data(), dims.data(), ld.data(), info.data(), auxp.data(), rider);
  if (rode) *rode = did;
  std::vector<Whitener> out(m);
  for (int i = 0; i < m; ++i) {
    Whitener& w = out[i];
    const int64_t d = dims[i];
    w.d = d;
    if (info[i] == 0) {
      w.chol = true;
      w.r = d;
      w.L = std::move(R[i]);
      w.aux = std::move(aux[i]);
      continue;
    }
    if (!allow_floor) fail(CCZ_ENOTSPD, "regularised covariance block %d (%lld x %lld) is not positive definite", i, (long long)d, (long long)d);
    std::vector<double> lam;
    DBuf V(c, d * d);
    syev_full_impl(c, keep[i], d, true, lam, V, d);
    const double floor_ = kRankTol * double(d) * std::max(lam.empty() ? 0.0 : lam[0], 0.0);
    int64_t r = 0;
    while (r < d && lam[r] > floor_) ++r;
    if (r == 0) fail(CCZ_ENOTSPD, "covariance block %d is numerically zero", i);
    std::vector<int64_t> perm(r);
    std::vector<double> sc(r);
    for (int64_t t = 0; t < r; ++t) { perm[t] = t; sc[t] = 1.0 / std::sqrt(lam[t]); }
    DBuf Ft(c, r * d);
    gather_rows(c, r, d, V, d, perm.data(), sc.data(), Ft, d);
    w.chol = false;
    w.r = r;
    w.L = DBuf(c, d * r);
    transpose(c, r, d, Ft, d, w.L, r);
    R[i].reset();
  }
  return out;
}

// eps-floor rule of the reference (linear/_mcca.py:170-172, linear/_gcca.py:102-104): the minimum
// eigenvalue of each block if it is below eps, else eps.  Cheap certificates first:
// c >= eps (ridge) or "R - eps I is positive definite" (one batched Cholesky) => min eig >= eps.
std::vector<double> min_eigs_if_below(ccz_ctx* c, const std::vector<DBuf>& R, const std::vector<int64_t>& dims,
                                      const double* cvals, double eps) {
  const int m = int(R.size());
  std::vector<double> out(m, eps);
  std::vector<int> todo;
  for (int i = 0; i < m; ++i)
    if (!(cvals[i] >= eps)) todo.push_back(i);      // (1-c) C + c I with C >= 0  =>  min eig >= c
  if (todo.empty()) return out;
  std::vector<DBuf> T(todo.size());
  std::vector<double*> ptr(todo.size());
  std::vector<int64_t> dd(todo.size());
  std::vector<int> info(todo.size(), 0);
  for (size_t t = 0; t < todo.size(); ++t) {
    const int64_t d = dims[todo[t]];
    T[t] = DBuf(c, d * d);
    d2d(c, T[t], R[todo[t]], size_t(d) * d * 8);
    add_diag(c, d, T[t], d, -eps);
    ptr[t] = T[t].get();
    dd[t] = d;
  }
  potrf_lower_batched(c, int(todo.size()), ptr.data(), dd.data(), dd.data(), info.data());
  for (size_t t = 0; t < todo.size(); ++t) {
    if (info[t] == 0) continue;                       // certified >= eps
    const int64_t d = dd[t];
    d2d(c, T[t], R[todo[t]], size_t(d) * d * 8);
    std::vector<double> lam;
    syev_full_impl(c, T[t], d, true, lam, nullptr, 0);
    out[todo[t]] = lam.back();
  }
  return out;
}

void means_out(ccz_ctx* c, const double* s_dev, int64_t D, int64_t n, bool center, double* means_host) {
  if (!means_host) return;
  if (center) {
    d2h(c, means_host, s_dev, size_t(D) * 8);
    for (int64_t i = 0; i < D; ++i) means_host[i] /= double(n);
  } else {
    std::fill(means_host, means_host + D, 0.0);
  }
}

// rows-form device block Wt (k x d, ld ldw) -> host (d x k) row-major.  The transposition happens on the DEVICE: the host
// loop that used to do it walked a 2 MB buffer with a stride of d * 8 bytes (32 KB at d = 4096: every access of a column in
// the same cache set), which cost 0.5 ms in some fits and 8 - 13 ms in others, depending on where the allocator put the
// buffer -- the whole of the "outlier" solves of rounds 3 - 5 (tools/d2h_probe.py, profiles/r05_solve_outliers.md).
void rows_to_host_cols(ccz_ctx* c, const double* Wt, int64_t k, int64_t d, int64_t ldw, double scale,
                       double* out_host) {
  DBuf tmp(c, k * d);
  transpose(c, k, d, Wt, ldw, tmp, k);
  d2h(c, out_host, tmp, size_t(k) * size_t(d) * 8);
  if (scale != 1.0)
    for (int64_t i = 0; i < d * k; ++i) out_host[i] *= scale;
}

void check_common(const double* moments, int64_t n, const int64_t* dims, int m, int k) {
  if (!moments || !dims) fail(CCZ_EINVAL, "null argument");
  if (m < 2) fail(CCZ_EINVAL, "at least 2 views are required, got %d", m);
  if (n < 2) fail(CCZ_EINVAL, "at least 2 samples are required, got %lld", (long long)n);
  if (k < 1) fail(CCZ_EINVAL, "latent dimensions must be >= 1, got %d", k);
  for (int i = 0; i < m; ++i)
    if (dims[i] < 1) fail(CCZ_EINVAL, "view %d has no features", i);
}

}  // namespace

void chol_solve_inplace(ccz_ctx* c, int64_t d, int64_t r, const double* L, int64_t ldl, double* X, int64_t ldx) {
  DBuf Xt(c, r * d);
  transpose(c, d, r, X, ldx, Xt, d);
  trsm_right_lower(c, true, r, d, L, ldl, Xt, d);    // X' L^-T
  trsm_right_lower(c, false, r, d, L, ldl, Xt, d);   // X' L^-T L^-1 = ((L L')^-1 X)'
  transpose(c, r, d, Xt, d, X, ldx);
}

// ===========================================================================
// rCCA / CCA / PLS      reference: cca_zoo/linear/_rcca.py:69-101
// ===========================================================================
// CCZ_TRACE_PHASES=1: synchronise at the phase boundaries of the rCCA solve and print the wall time of each phase;
// CCZ_TRACE_PHASES=2: no synchronisation -- events, host time stamps and shader-clock probes (ops.h::trace_mark).
// (Measurement aids for the launch-bound chain; mode 1 changes the overlap between host and device.)
struct PhaseTrace {
  ccz_ctx* c;
  int mode;
  std::chrono::steady_clock::time_point t;
  std::string line;
  explicit PhaseTrace(ccz_ctx* c_) : c(c_) {
    static const int env = [] { const char* e = getenv("CCZ_TRACE_PHASES"); return e ? atoi(e) : 0; }();
    mode = env;
    if (mode == 1) { sync(c); t = std::chrono::steady_clock::now(); }
    if (mode == 2) trace_mark(c, "start");
  }
  void mark(const char* name) {
    if (mode == 2) { trace_mark(c, name); return; }
    if (mode != 1) return;
    sync(c);
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof(buf), " %s %.2f", name, std::chrono::duration<double, std::milli>(now - t).count());
    line += buf;
    t = now;
  }
  ~PhaseTrace() {
    if (mode == 1) fprintf(stderr, "[ccz] rcca phases (ms):%s\n", line.c_str());
    if (mode == 2) { try { trace_flush(c, "rcca"); } catch (...) {} }
  }
};

static void rcca_solve_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t dims[2],
                            const double cc[2], int center, int k, double* W_host, double* means_host,
                            double* vals_host, int* k_out) {
  check_common(mom, n, dims, 2, k);
  for (int i = 0; i < 2; ++i)
    if (!(cc[i] >= 0.0 && cc[i] <= 1.0)) fail(CCZ_EINVAL, "ridge parameter c[%d]=%g outside [0, 1]", i, cc[i]);
  const int64_t d1 = dims[0], d2 = dims[1], D = d1 + d2;
  const double* G = mom;
  const double* s = mom + D * D;
  const double inv = 1.0 / double(n - 1);
  const bool ctr = center != 0;
  // reference: k = min(latent, rank1, rank2); ranks are at most min(n, d)
  int kk = int(std::min<int64_t>({int64_t(k), d1, d2, n}));

  PhaseTrace pt(c);
  std::vector<DBuf> Rv(2);
  Rv[0] = DBuf(c, d1 * d1);
  Rv[1] = DBuf(c, d2 * d2);
  DBuf M12(c, d1 * d2);
  cov_block(c, G, D, s, n, ctr, (1.0 - cc[0]) * inv, 0, d1, 0, d1, Rv[0], d1);
  add_diag(c, d1, Rv[0], d1, cc[0]);
  cov_block(c, G, D, s, n, ctr, (1.0 - cc[1]) * inv, d1, d2, d1, d2, Rv[1], d2);
  add_diag(c, d2, Rv[1], d2, cc[1]);
  pt.mark("cov");

  // The factorizations need the diagonal blocks only: under the sharded exchange the cross block may still be in flight.
  // The first whitening solve  M12 <- M12 L_2^-T  rides along the factorization (its products fill the chip under the
  // Cholesky chain); M12 is formed -- and the exchange's second part awaited -- right before the rider's first step.
  auto form_m12 = [&] {
    wait_deferred(c);
    cov_block(c, G, D, s, n, ctr, inv, 0, d1, d1, d2, M12, d2);
  };
  TrsmRider rider;
  rider.matrix = 1;
  rider.r = d1;
  rider.X = M12.get();
  rider.ldx = d2;
  rider.prepare = form_m12;
  bool rode = false;
  std::vector<Whitener> Fv = make_whiteners(c, Rv, {d1, d2}, true, &rider, &rode);
  pt.mark("factor");
  if (rode && !Fv[1].chol) rode = false;            // the factor it rode on was rejected: M12 holds a partial solve
  if (!rode) form_m12();
  Whitener& F1 = Fv[0];
  Whitener& F2 = Fv[1];
  const int64_t r1 = F1.r, r2 = F2.r;
  kk = int(std::min<int64_t>({int64_t(kk), r1, r2}));

  // Y = M12 F2 (d1 x r2);  Tt = Y' F1 = (F1' M12 F2)'  (r2 x r1)
  // (Cholesky whiteners work in place: r_i = d_i, no copies; the eigen-floored fallback has an explicit F)
  DBuf Y;
  if (F2.chol) { if (!rode) F2.right_apply(c, d1, M12, d2, M12, d2); Y = std::move(M12); }
  else { Y = DBuf(c, d1 * r2); F2.right_apply(c, d1, M12, d2, Y, r2); M12.reset(); }
  DBuf Yt(c, r2 * d1);
  transpose(c, d1, r2, Y, r2, Yt, d1);
  Y.reset();
  DBuf Tt;
  if (F1.chol) { F1.right_apply(c, r2, Yt, d1, Yt, d1); Tt = std::move(Yt); }
  else { Tt = DBuf(c, r2 * r1); F1.right_apply(c, r2, Yt, d1, Tt, r1); Yt.reset(); }

  pt.mark("whiten");
  std::vector<double> sv;
  DBuf Ut(c, int64_t(kk) * r1), Vt(c, int64_t(kk) * r2);
  svd_topk_dense(c, Tt, r1, r2, kk, sv, Ut, Vt);
  pt.mark("topk");

  if (F1.chol && F2.chol) {          // r_i = d_i: in place, both views' dependent steps in shared launches
    back_project_rows_multi(c, {&F1, &F2}, kk, {Ut.get(), Vt.get()}, {r1, r2});
    rows_to_host_cols(c, Ut, kk, d1, d1, 1.0, W_host);
    rows_to_host_cols(c, Vt, kk, d2, d2, 1.0, W_host + d1 * kk);
  } else {
    DBuf W1t(c, int64_t(kk) * d1), W2t(c, int64_t(kk) * d2);
    F1.back_project_rows(c, kk, Ut, r1, W1t, d1);
    F2.back_project_rows(c, kk, Vt, r2, W2t, d2);
    rows_to_host_cols(c, W1t, kk, d1, d1, 1.0, W_host);
    rows_to_host_cols(c, W2t, kk, d2, d2, 1.0, W_host + d1 * kk);
  }
  means_out(c, s, D, n, ctr, means_host);
  pt.mark("backproject");
  if (vals_host) std::copy(sv.begin(), sv.begin() + kk, vals_host);
  if (k_out) *k_out = kk;
}

// ===========================================================================
// MCCA      reference: cca_zoo/linear/_mcca.py:99-197
// ===========================================================================
static void mcca_solve_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m,
                            const double* cc, double eps, int center, int k, double* W_host,
                            double* means_host, double* vals_host, int* k_out) {
  check_common(mom, n, dims, m, k);
  if (!(eps > 0.0)) fail(CCZ_EINVAL, "eps must be > 0, got %g", eps);
  auto off = offsets(dims, m);
  const int64_t D = off[m];
  const double* G = mom;
  const double* s = mom + D * D;
  const double inv = 1.0 / double(n - 1);
  for (int i = 0; i < m; ++i)
    if (!(cc[i] >= 0.0 && cc[i] <= 1.0)) fail(CCZ_EINVAL, "ridge parameter c[%d]=%g outside [0, 1]", i, cc[i]);

  // B blocks (always from the CENTRED covariance: np.cov / PCA re-centre)
  std::vector<DBuf> R(m);
  const std::vector<int64_t> dimv(dims, dims + m);
  for (int i = 0; i < m; ++i) {
    R[i] = DBuf(c, dims[i] * dims[i]);
    cov_block(c, G, D, s, n, true, (1.0 - cc[i]) * inv, off[i], dims[i], off[i], dims[i], R[i], dims[i]);
    add_diag(c, dims[i], R[i], dims[i], cc[i]);
  }
  double min_eig = eps;
  for (double v : min_eigs_if_below(c, R, dimv, cc, eps)) min_eig = std::min(min_eig, v);
  const double shift = min_eig < eps ? eps - min_eig : 0.0;
  if (shift > 0.0)
    for (int i = 0; i < m; ++i) add_diag(c, dims[i], R[i], dims[i], shift);
  std::vector<Whitener> F = make_whiteners(c, R, dimv, false);
  wait_deferred(c);
  // S = L^-1 (C - blockdiag C) L^-T,  zero diagonal blocks
  // Whole block columns / block rows at a time (round 6): the m (m - 1) / 2 off-diagonal blocks used to be whitened one by one
  // -- two triangular solves of d_i x d_j each, 84 launches of 40 - 78 us at 4 x 2048 that filled a quarter of the chip
  // (profiles/r06_solve_timeline_mcca.md: 4.7 ms of a 21 ms solve).  All blocks above the diagonal of block column j share L_j, all
  // blocks right of the diagonal of block row i share L_i: m - 1 solves with off[j] rows in place in S, then m - 1 with D - off[i + 1].
  DBuf S(c, D * D);
  fill2d(c, D, D, S, D, 0.0);
  for (int j = 1; j < m; ++j) {
    double* Uj = S.get() + off[j];                               // rows [0, off[j]) x columns of view j
    cov_block(c, G, D, s, n, true, inv, 0, off[j], off[j], dims[j], Uj, D);
    F[j].right_apply(c, off[j], Uj, D, Uj, D);                   // C_ij L_j^-T for every i < j
  }
  for (int i = 0; i + 1 < m; ++i) {
    const int64_t di = dims[i], rest = D - off[i + 1];
    double* Ri = S.get() + off[i] * D + off[i + 1];              // block row i right of the diagonal: d_i x rest
    DBuf Ct(c, rest * di);
    transpose(c, di, rest, Ri, D, Ct, di);
    F[i].right_apply(c, rest, Ct, di, Ct, di);                   // (L_i^-1 [C_ij L_j^-T]_{j > i})' = S_ji for every j > i
    copy2d(c, rest, di, Ct, di, S.get() + off[i + 1] * D + off[i], D);
    transpose(c, rest, di, Ct, di, Ri, D);
  }
  const int kk = int(std::min<int64_t>(k, D));
  std::vector<double> lam;
  DBuf Yt(c, int64_t(kk) * D);
  // S = L^-1 C L^-T - blockdiag(L_i^-1 C_ii L_i^-T): the first term is positive semi-definite, and with
  // R_i = (1 - c_i) C_ii + (c_i + shift) I the eigenvalues of block i of the second are mu / ((1 - c_i) mu + c_i + shift)
  // < 1 / (1 - c_i) for the eigenvalues mu >= 0 of C_ii.  So S >= -max_i 1 / (1 - c_i)  (c_i = 1: no such bound).
  double lower = 0.0;
  for (int i = 0; i < m; ++i) lower = std::max(lower, cc[i] < 1.0 ? 1.0 / (1.0 - cc[i]) : HUGE_VAL);
  eig_topk_dense(c, S, D, kk, lam, Yt, D, std::isfinite(lower) ? -(1.0 + 1e-6) * lower : -HUGE_VAL);
  S.reset();
  // v_i = sqrt(m) L_i^-T y_i   (normalisation v'(B/m)v = 1 of LAPACK sygvx on (A/m, B/m))
  int64_t wofs = 0;
  {
    std::vector<const Whitener*> Fp;
    std::vector<double*> Yp;
    std::vector<int64_t> ldy;
    for (int i = 0; i < m; ++i) { Fp.push_back(&F[i]); Yp.push_back(Yt.get() + off[i]); ldy.push_back(D); }
    back_project_rows_multi(c, Fp, kk, Yp, ldy);       // make_whiteners(allow_floor = false): all Cholesky
  }
  for (int i = 0; i < m; ++i) {
    rows_to_host_cols(c, Yt.get() + off[i], kk, dims[i], D, std::sqrt(double(m)), W_host + wofs);
    wofs += dims[i] * kk;
  }
  means_out(c, s, D, n, center != 0, means_host);
  if (vals_host) std::copy(lam.begin(), lam.begin() + kk, vals_host);
  if (k_out) *k_out = kk;
}

// ===========================================================================
// GCCA in D x D Gram form      reference: cca_zoo/linear/_gcca.py:80-110
// ===========================================================================
static void gcca_solve_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m,
                            const double* cc, const double* mu, double eps, int center, int k,
                            double* W_host, double* means_host, double* vals_host, int* k_out) {
  check_common(mom, n, dims, m, k);
  if (!(eps > 0.0)) fail(CCZ_EINVAL, "eps must be > 0, got %g", eps);
  auto off = offsets(dims, m);
  const int64_t D = off[m];
  const double* G = mom;
  const double* s = mom + D * D;
  const double inv = 1.0 / double(n - 1);
  const bool ctr = center != 0;
  std::vector<double> rmu(m);
  for (int i = 0; i < m; ++i) {
    if (!(cc[i] >= 0.0 && cc[i] <= 1.0)) fail(CCZ_EINVAL, "ridge parameter c[%d]=%g outside [0, 1]", i, cc[i]);
    const double w = mu ? mu[i] : 1.0;
    if (!(w >= 0.0)) fail(CCZ_EINVAL, "view weight %d must be non-negative, got %g", i, w);
    rmu[i] = std::sqrt(w);
  }
  std::vector<DBuf> R(m);
  const std::vector<int64_t> dimv(dims, dims + m);
  for (int i = 0; i < m; ++i) {
    R[i] = DBuf(c, dims[i] * dims[i]);
    cov_block(c, G, D, s, n, true, (1.0 - cc[i]) * inv, off[i], dims[i], off[i], dims[i], R[i], dims[i]);
    add_diag(c, dims[i], R[i], dims[i], cc[i]);
  }
  {
    const std::vector<double> lo = min_eigs_if_below(c, R, dimv, cc, eps);
    for (int i = 0; i < m; ++i)
      if (lo[i] < eps) add_diag(c, dims[i], R[i], dims[i], eps - lo[i]);   // per-view floor (_gcca.py:102-104)
  }
  std::vector<Whitener> F = make_whiteners(c, R, dimv, false);
  wait_deferred(c);
  // K = F' Gx F,  F = blockdiag(sqrt(mu_i) L_i^-T),  Gx: second moments of the data as fitted.  K is symmetric, so only its blocks on
  // and above the diagonal are formed (round 6: the two whitening passes were 3.3e12 flops of a 16384 x 16384 problem with all D rows
  // in every triangular solve -- 42 ms of a 162 ms solve; the block-upper form is 2.3e12) and mirrored:
  //   pass 1:  Z_ij = sqrt(mu_j) Gx_ij L_j^-T  for i <= j   -- rows [0, off[j + 1]) of block column j, in place
  //   pass 2:  K_ji = sqrt(mu_i) (L_i^-1 Z_ij)' for j >= i  -- block row i from the diagonal on, transposed, whitened by L_i;
  //            written as block column i below the diagonal and mirrored into block row i; the diagonal block is averaged with its
  //            transpose (symmetric up to round-off)
  DBuf Z(c, D * D), K(c, D * D);
  for (int j = 0; j < m; ++j) {
    cov_block(c, G, D, s, n, ctr, rmu[j], 0, off[j + 1], off[j], dims[j], Z.get() + off[j], D);
    F[j].right_apply(c, off[j + 1], Z.get() + off[j], D, Z.get() + off[j], D);
  }
  for (int i = 0; i < m; ++i) {
    const int64_t di = dims[i], rest = D - off[i];
    double* Ri = Z.get() + off[i] * D + off[i];                  // d_i x rest
    DBuf T(c, rest * di);
    transpose(c, di, rest, Ri, D, T, di);
    F[i].right_apply(c, rest, T, di, T, di);
    axpby2d(c, rest, di, rmu[i], T, di, 0.0, nullptr, 0);
    double* Kii = K.get() + off[i] * D + off[i];
    copy2d(c, rest, di, T, di, Kii, D);                          // block column i from the diagonal down
    transpose(c, rest, di, T, di, Kii, D);                       // block row i from the diagonal on (the diagonal block: T_ii')
    axpby2d(c, di, di, 0.5, Kii, D, 0.5, T, di);                 // K_ii = (T_ii' + T_ii) / 2
  }
  const int kk = int(std::min<int64_t>({int64_t(k), D, n}));
  std::vector<double> lam;
  DBuf Ut(c, int64_t(kk) * D);
  // K = F' Gx F with F = blockdiag(sqrt(mu_i) L_i^-T) and Gx a second-moment matrix: positive semi-definite up to
  // round-off (a few ulp of its norm).  The generic bound -||K||_inf would triple the width of the damped interval.
  eig_topk_dense(c, K, D, kk, lam, Ut, D, -1e-9 * norm_inf(c, D, D, K, D));
  K.reset();
  for (int i = 0; i < kk; ++i)
    if (!(lam[i] > 0.0)) fail(CCZ_ENOCONV, "GCCA eigenvalue %d is not positive (%g); fewer than k shared directions", i, lam[i]);
  // rhs = Gx F U / sqrt(lam)   (D x k)  == X' T stacked by view.  (Z = Gx F is only half formed now: F U first -- k rows through
  // every view's factor, one batched back-projection -- then ONE skinny product with Gx, which takes Z's place.)
  DBuf U(c, D * kk), rhs(c, D * kk), lamd(c, kk);
  {
    std::vector<const Whitener*> Fp;
    std::vector<double*> Yp;
    std::vector<int64_t> ldy;
    for (int i = 0; i < m; ++i) { Fp.push_back(&F[i]); Yp.push_back(Ut.get() + off[i]); ldy.push_back(D); }
    back_project_rows_multi(c, Fp, kk, Yp, ldy);                 // Ut_i <- (L_i^-T U_i)'
    for (int i = 0; i < m; ++i) axpby2d(c, kk, dims[i], rmu[i], Ut.get() + off[i], D, 0.0, nullptr, 0);
  }
  transpose(c, kk, D, Ut, D, U, kk);
  cov_block(c, G, D, s, n, ctr, 1.0, 0, D, 0, D, Z, D);          // Gx (symmetric, full)
  gemm(c, false, false, D, kk, D, 1.0, Z, D, U, kk, 0.0, rhs, kk);
  h2d(c, lamd, lam.data(), size_t(kk) * 8);
  scale_cols(c, D, kk, rhs, kk, lamd, 2);
  Z.reset();
  // W_i = pinv(X_i) T = (Gx_ii)^+ rhs_i
  int64_t wofs = 0;
  for (int i = 0; i < m; ++i) {
    const int64_t di = dims[i];
    DBuf Gii(c, di * di), keep(c, di * di);
    cov_block(c, G, D, s, n, ctr, 1.0, off[i], di, off[i], di, Gii, di);
    d2d(c, keep, Gii, size_t(di) * di * 8);
    double* rb = rhs.get() + off[i] * kk;
    if (potrf_lower(c, Gii, di, di) == 0) {
      chol_solve_inplace(c, di, kk, Gii, di, rb, kk);
    } else {  // rank-deficient view: eigen pseudo-inverse with a relative floor
      std::vector<double> ev;
      DBuf V(c, di * di);
      syev_full_impl(c, keep, di, true, ev, V, di);
      const double fl = kRankTol * double(di) * std::max(ev[0], 0.0);
      int64_t r = 0;
      while (r < di && ev[r] > fl) ++r;
      if (r == 0) fail(CCZ_ENOTSPD, "view %d has zero variance", i);
      std::vector<int64_t> perm(r);
      std::vector<double> sc(r);
      for (int64_t t = 0; t < r; ++t) { perm[t] = t; sc[t] = 1.0 / std::sqrt(ev[t]); }
      DBuf Vs(c, r * di), t1(c, r * kk), t2(c, di * kk);
      gather_rows(c, r, di, V, di, perm.data(), sc.data(), Vs, di);      // rows v_t / sqrt(lam_t)
      gemm(c, false, false, r, kk, di, 1.0, Vs, di, rb, kk, 0.0, t1, kk);
      gemm(c, true, false, di, kk, r, 1.0, Vs, di, t1, kk, 0.0, t2, kk);
      copy2d(c, di, kk, t2, kk, rb, kk);
    }
    std::vector<double> h(size_t(di) * kk);
    DBuf tmp(c, di * kk);
    copy2d(c, di, kk, rb, kk, tmp, kk);
    d2h(c, h.data(), tmp, h.size() * 8);
    std::copy(h.begin(), h.end(), W_host + wofs);
    wofs += di * kk;
  }
  means_out(c, s, D, n, ctr, means_host);
  if (vals_host) std::copy(lam.begin(), lam.begin() + kk, vals_host);
  if (k_out) *k_out = kk;
}

// ===========================================================================
// top-k generalised eigenpairs  A v = lam B v  (B SPD or null), rows-form output Vt (kk x p), v'Bv = 1
// reference: gevp, cca_zoo/_utils/_linalg.py:44-73
// ===========================================================================
void gevp_topk_rows(ccz_ctx* c, const double* A, const double* B, int64_t p, int kk, std::vector<double>& lam, double* Vt) {
  if (!B) {
    eig_topk_dense(c, A, p, kk, lam, Vt, p);
    return;
  }
  DBuf L(c, p * p), Y(c, p * p), St(c, p * p);
  d2d(c, L, B, size_t(p) * p * 8);
  if (potrf_lower(c, L, p, p) != 0) fail(CCZ_ENOTSPD, "B is not positive definite");
  d2d(c, Y, A, size_t(p) * p * 8);
  trsm_right_lower(c, true, p, p, L, p, Y, p);          // A L^-T
  transpose(c, p, p, Y, p, St, p);
  trsm_right_lower(c, true, p, p, L, p, St, p);         // (L^-1 A L^-T)'
  transpose(c, p, p, St, p, Y, p);
  axpby2d(c, p, p, 0.5, Y, p, 0.5, St, p);             // symmetrise
  eig_topk_dense(c, Y, p, kk, lam, Vt, p);
  trsm_right_lower(c, false, kk, p, L, p, Vt, p);       // rows y' L^-1 = (L^-T y)'
}

// ===========================================================================
// GCCALoss from the batch moments      reference: cca_zoo/deep/objectives.py:155-220
//   loss  = -(n-1) sum_{j<k} lam_j,   C u = lam B u,  B = blockdiag(C_ii) + eps I,  u'Bu = 1
//   Gamma = -2 sum_j (u_j u_j' - lam_j blockdiag(u_ji u_ji'))    with   dL/dZ = (Z - 1 mean') Gamma
// (maths: oracle/losses.py::gcca_loss_closed_form).  Everything stays on the device; only the k eigenvalues
// come to the host.
// ===========================================================================
static void gcca_loss_moments_impl(ccz_ctx* c, const double* mom, int64_t n, const int64_t* dims, int m, double eps,
                                   int k, double* loss_host, double* gamma_dev, double* mean_dev) {
  if (!mom || !dims || !loss_host) fail(CCZ_EINVAL, "gcca_loss_moments: null argument");
  if (m < 1 || n < 2 || k < 1) fail(CCZ_EINVAL, "gcca_loss_moments: bad shape");
  if (gamma_dev && !mean_dev) fail(CCZ_EINVAL, "gcca_loss_moments: mean_dev is required with gamma_dev");
  for (int i = 0; i < m; ++i)
    if (dims[i] < 1) fail(CCZ_EINVAL, "gcca_loss_moments: view %d has no features", i);
  auto off = offsets(dims, m);
  const int64_t D = off[m];
  const double* G = mom;
  const double* s = mom + D * D;
  const double inv = 1.0 / double(n - 1);
  const int kk = int(std::min<int64_t>(k, D));
  DBuf Cm(c, D * D), Bm(c, D * D);
  cov_block(c, G, D, s, n, true, inv, 0, D, 0, D, Cm, D);
  fill2d(c, D, D, Bm, D, 0.0);
  for (int i = 0; i < m; ++i)
    copy2d(c, dims[i], dims[i], Cm.get() + off[i] * D + off[i], D, Bm.get() + off[i] * D + off[i], D);
  add_diag(c, D, Bm, D, eps);
  std::vector<double> lam;
  DBuf Vt(c, int64_t(kk) * D);
  gevp_topk_rows(c, Cm, Bm, D, kk, lam, Vt);
  double total = 0.0;
  for (double v : lam) total += v;
  *loss_host = -double(n - 1) * total;
  if (!gamma_dev) return;
  // Gamma = -2 Vt'Vt  +  2 blockdiag_i( (Lam Vt_i)' Vt_i )
  gemm(c, true, false, D, D, kk, -2.0, Vt, D, Vt, D, 0.0, gamma_dev, D);
  std::vector<int64_t> perm(kk);
  std::iota(perm.begin(), perm.end(), 0);
  DBuf VtL(c, int64_t(kk) * D);
  gather_rows(c, kk, D, Vt, D, perm.data(), lam.data(), VtL, D);
  for (int i = 0; i < m; ++i)
    gemm(c, true, false, dims[i], dims[i], kk, 2.0, VtL.get() + off[i], D, Vt.get() + off[i], D, 1.0,
         gamma_dev + off[i] * D + off[i], D);
  d2d(c, mean_dev, s, size_t(D) * 8);
  axpby2d(c, 1, D, 1.0 / double(n), mean_dev, D, 0.0, nullptr, 0);
}

// ===========================================================================
// factor loadings of ONE view from its second moments      reference: cca_zoo/_base.py:208-234
//   out[j][t] = corr(x_j, z_t),  z = (x - mean) W:   cov(x, z) = C W,  var z = diag(W'CW),  var x = diag(C)
// with the reference's guards std = max(std, 1e-12).
// ===========================================================================
static void factor_loadings_impl(ccz_ctx* c, const double* mom, int64_t n, int64_t d, const double* W, int64_t k,
                                 double* out) {
  if (!mom || !W || !out || n < 2 || d < 1 || k < 1) fail(CCZ_EINVAL, "factor_loadings: bad argument");
  DBuf Cm(c, d * d), CW(c, d * k), ones(c, d), vx(c, d), vt(c, k), Wt(c, k * d), CWt(c, k * d);
  cov_block(c, mom, d, mom + d * d, n, true, 1.0 / double(n - 1), 0, d, 0, d, Cm, d);
  gemm(c, false, false, d, k, d, 1.0, Cm, d, W, k, 0.0, CW, k);
  fill2d(c, 1, d, ones, d, 1.0);
  row_dots(c, d, 1, Cm, d + 1, ones, 1, vx);                 // diag(C)
  transpose(c, d, k, W, k, Wt, d);
  transpose(c, d, k, CW, k, CWt, d);
  row_dots(c, k, d, Wt, d, CWt, d, vt);                      // diag(W' C W)
  std::vector<double> hx(d), ht(k), hg(d);
  d2h(c, hx.data(), vx, size_t(d) * 8);
  d2h(c, ht.data(), vt, size_t(k) * 8);
  row_dots(c, d, 1, mom, d + 1, ones, 1, vx);                // diag(G): raw second moments
  d2h(c, hg.data(), vx, size_t(d) * 8);
  std::vector<int64_t> perm(d);
  std::iota(perm.begin(), perm.end(), 0);
  for (int64_t j = 0; j < d; ++j) {
    // a feature that is constant up to the rounding of G - s s'/n has a centred column of exact zeros in the
    // reference (covariance 0, loading 0 / 1e-12 = 0): give 0 instead of cancellation noise over 1e-12
    const double noise = 256.0 * 2.220446049250313e-16 * std::fabs(hg[j]) / double(n - 1);
    hx[j] = hx[j] <= noise ? 0.0 : 1.0 / std::max(std::sqrt(hx[j]), 1e-12);
  }
  for (auto& v : ht) v = std::max(std::sqrt(std::max(v, 0.0)), 1e-12);
  DBuf td(c, k);
  h2d(c, td, ht.data(), size_t(k) * 8);
  gather_rows(c, d, k, CW, k, perm.data(), hx.data(), out, k);   // rows / std_x
  scale_cols(c, d, k, out, k, td, 1);                             // columns / std_z
}

}  // namespace ccz

// ===========================================================================
// C ABI (solver part)
// ===========================================================================
#define CCZ_GUARD(h, ...)                                              \
  if (!(h)) return CCZ_EINVAL;                                         \
  try {                                                                \
    ::ccz::DeviceScope ccz_scope_(h);                                              \
    __VA_ARGS__;                                                       \
    return CCZ_OK;                                                     \
  } catch (const ccz::Error& e) {                                      \
    (h)->err = e.msg;                                                  \
    return e.code;                                                     \
  } catch (const std::bad_alloc&) {                                    \
    (h)->err = "host allocation failed";                               \
    return CCZ_ENOMEM;                                                 \
  } catch (...) {                                                      \
    (h)->err = "unknown internal error";                               \
    return CCZ_EHIP;                                                   \
  }

namespace {
// The off-diagonal half of a sharded exchange may still be in flight when a solve starts (ccz_solve_defer).  Whatever
// way a solve ends -- including an early fail() in the argument / SPD checks, before its own wait_deferred -- the
// registration is consumed here, so that it can neither leak into the next (possibly unsharded) solve nor leave the
// side stream's unpack un-awaited.
struct DeferGuard {
  ccz_ctx* c;
  explicit DeferGuard(ccz_ctx* c_) : c(c_) {}
  ~DeferGuard() { try { ccz::wait_deferred(c); } catch (...) {} }
};
}  // namespace

extern "C" {

int ccz_rcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t dims[2],
                   const double c[2], int center, int k, double* weights_host, double* means_host,
                   double* vals_host, int* k_out) {
  CCZ_GUARD(h, {
    DeferGuard defer_guard(h);
    if (!c || !weights_host) ccz::fail(CCZ_EINVAL, "null argument");
    ccz::rcca_solve_impl(h, moments_dev, n, dims, c, center, k, weights_host, means_host, vals_host, k_out);
  })
}

int ccz_mcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t* dims,
                   int n_views, const double* c, double eps, int center, int k,
                   double* weights_host, double* means_host, double* vals_host, int* k_out) {
  CCZ_GUARD(h, {
    DeferGuard defer_guard(h);
    if (!c || !weights_host) ccz::fail(CCZ_EINVAL, "null argument");
    ccz::mcca_solve_impl(h, moments_dev, n, dims, n_views, c, eps, center, k, weights_host, means_host, vals_host, k_out);
  })
}

int ccz_gcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t* dims,
                   int n_views, const double* c, const double* view_weights, double eps,
                   int center, int k, double* weights_host, double* means_host,
                   double* vals_host, int* k_out) {
  CCZ_GUARD(h, {
    DeferGuard defer_guard(h);
    if (!c || !weights_host) ccz::fail(CCZ_EINVAL, "null argument");
    ccz::gcca_solve_impl(h, moments_dev, n, dims, n_views, c, view_weights, eps, center, k, weights_host, means_host, vals_host, k_out);
  })
}

int ccz_syevj(ccz_handle h, double* A_dev, int64_t d, double* w_dev, double* V_dev, int* sweeps_out) {
  CCZ_GUARD(h, {
    if (!A_dev || !w_dev || d < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    std::vector<double> w;
    int sw = ccz::syev_full_impl(h, A_dev, d, false, w, V_dev, d);
    ccz::h2d(h, w_dev, w.data(), size_t(d) * 8);
    if (sweeps_out) *sweeps_out = sw;
  })
}

int ccz_gesvj(ccz_handle h, const double* A_dev, int64_t p, int64_t q, double* U_dev,
              double* s_dev, double* Vt_dev, int* sweeps_out) {
  CCZ_GUARD(h, {
    if (!A_dev || !s_dev || p < 1 || q < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    const int64_t r = std::min(p, q);
    ccz::DBuf Ut(h, r * p);
    std::vector<double> s;
    int sw = ccz::gesvj_rows(h, A_dev, p, q, q, Ut, s, Vt_dev);
    if (U_dev) ccz::transpose(h, r, p, Ut, p, U_dev, r);
    ccz::h2d(h, s_dev, s.data(), size_t(r) * 8);
    if (sweeps_out) *sweeps_out = sw;
  })
}

int ccz_gevp_topk(ccz_handle h, const double* A_dev, const double* B_dev, int64_t p, int k,
                  double* w_dev, double* V_dev) {
  CCZ_GUARD(h, {
    if (!A_dev || !w_dev || !V_dev || p < 1 || k < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    const int kk = int(std::min<int64_t>(k, p));
    std::vector<double> lam;
    ccz::DBuf Vt(h, int64_t(kk) * p);
    ccz::gevp_topk_rows(h, A_dev, B_dev, p, kk, lam, Vt);
    ccz::transpose(h, kk, p, Vt, p, V_dev, kk);
    ccz::h2d(h, w_dev, lam.data(), size_t(kk) * 8);
  })
}

int ccz_svd_topk(ccz_handle h, const double* T_dev, int64_t p, int64_t q, int k, double* U_dev,
                 double* s_dev, double* V_dev) {
  CCZ_GUARD(h, {
    if (!T_dev || !s_dev || p < 1 || q < 1 || k < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    const int kk = int(std::min<int64_t>({int64_t(k), p, q}));
    ccz::DBuf Tt(h, q * p), Ut(h, int64_t(kk) * p), Vt(h, int64_t(kk) * q);
    ccz::transpose(h, p, q, T_dev, q, Tt, p);
    std::vector<double> s;
    ccz::svd_topk_dense(h, Tt, p, q, kk, s, Ut, Vt);
    if (U_dev) ccz::transpose(h, kk, p, Ut, p, U_dev, kk);
    if (V_dev) ccz::transpose(h, kk, q, Vt, q, V_dev, kk);
    ccz::h2d(h, s_dev, s.data(), size_t(kk) * 8);
  })
}

int ccz_whitener(ccz_handle h, const double* Gxx_dev, int64_t d, int64_t n, double ridge,
                 double* W_dev, double* lam_dev, int64_t* r_out) {
  CCZ_GUARD(h, {
    if (!Gxx_dev || !W_dev || d < 1 || n < 2) ccz::fail(CCZ_EINVAL, "bad argument");
    ccz::DBuf A(h, d * d), V(h, d * d);
    ccz::d2d(h, A, Gxx_dev, size_t(d) * d * 8);
    ccz::axpby2d(h, d, d, 1.0 / double(n - 1), A, d, 0.0, nullptr, 0);
    std::vector<double> lam;
    ccz::syev_full_impl(h, A, d, true, lam, V, d);
    const int64_t r = std::min(n, d);
    std::vector<int64_t> perm(r);
    std::vector<double> sc(r);
    for (int64_t i = 0; i < r; ++i) {
      perm[i] = i;
      lam[i] = std::max(lam[i], 0.0);
      sc[i] = 1.0 / std::sqrt((1.0 - ridge) * lam[i] + ridge);
    }
    ccz::DBuf Wt(h, r * d);
    ccz::gather_rows(h, r, d, V, d, perm.data(), sc.data(), Wt, d);
    ccz::transpose(h, r, d, Wt, d, W_dev, r);
    if (lam_dev) ccz::h2d(h, lam_dev, lam.data(), size_t(r) * 8);
    if (r_out) *r_out = r;
  })
}

int ccz_inv_sqrtm(ccz_handle h, const double* A_dev, int64_t d, double eps, double* out_dev) {
  CCZ_GUARD(h, {
    if (!A_dev || !out_dev || d < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    ccz::DBuf A(h, d * d), V(h, d * d), Vs(h, d * d);
    ccz::d2d(h, A, A_dev, size_t(d) * d * 8);
    std::vector<double> lam;
    ccz::syev_full_impl(h, A, d, false, lam, V, d);
    std::vector<int64_t> perm(d);
    std::vector<double> sc(d);
    for (int64_t i = 0; i < d; ++i) {
      perm[i] = i;
      sc[i] = std::pow(std::max(lam[i], eps), -0.25);
    }
    ccz::gather_rows(h, d, d, V, d, perm.data(), sc.data(), Vs, d);
    ccz::gemm(h, true, false, d, d, d, 1.0, Vs, d, Vs, d, 0.0, out_dev, d);
  })
}

int ccz_potrf_lower(ccz_handle h, double* A_dev, int64_t d, int64_t lda) {
  CCZ_GUARD(h, {
    if (!A_dev || d < 1 || lda < d) ccz::fail(CCZ_EINVAL, "bad argument");
    int info = ccz::potrf_lower(h, A_dev, d, lda);
    if (info != 0) ccz::fail(CCZ_ENOTSPD, "pivot %d is not positive", info - 1);
  })
}

int ccz_trsm_right_lower(ccz_handle h, int trans, int64_t r, int64_t d, const double* L_dev,
                         int64_t ldl, double* X_dev, int64_t ldx) {
  CCZ_GUARD(h, {
    if (!L_dev || !X_dev || r < 1 || d < 1 || ldl < d || ldx < d) ccz::fail(CCZ_EINVAL, "bad argument");
    ccz::trsm_right_lower(h, trans != 0, r, d, L_dev, ldl, X_dev, ldx);
  })
}

int ccz_moments_axpby(ccz_handle h, int64_t D, double alpha, const double* x_dev, double beta, double* y_dev) {
  CCZ_GUARD(h, {
    if (!x_dev || !y_dev || D < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    const int64_t total = D * D + D;
    ccz::axpby2d(h, 1, total, beta, y_dev, total, alpha, x_dev, total);
  })
}

int ccz_moments_subset(ccz_handle h, const double* moments_dev, int64_t D, int64_t col0, int64_t D_sub,
                       double* subset_dev) {
  CCZ_GUARD(h, {
    if (!moments_dev || !subset_dev || D < 1 || D_sub < 1 || col0 < 0 || col0 + D_sub > D)
      ccz::fail(CCZ_EINVAL, "bad argument");
    ccz::copy2d(h, D_sub, D_sub, moments_dev + col0 * D + col0, D, subset_dev, D_sub);
    ccz::copy2d(h, 1, D_sub, moments_dev + D * D + col0, D, subset_dev + D_sub * D_sub, D_sub);
  })
}

int ccz_gemm_f64(ccz_handle h, int transA, int transB, int64_t M, int64_t N, int64_t K,
                 double alpha, const double* A_dev, int64_t lda, const double* B_dev, int64_t ldb,
                 double beta, double* C_dev, int64_t ldc) {
  CCZ_GUARD(h, {
    if (!A_dev || !B_dev || !C_dev || M < 1 || N < 1 || K < 1) ccz::fail(CCZ_EINVAL, "bad argument");
    ccz::gemm(h, transA != 0, transB != 0, M, N, K, alpha, A_dev, lda, B_dev, ldb, beta, C_dev, ldc);
  })
}

int ccz_gcca_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, const int64_t* dims, int n_views,
                          double eps, int k, double* loss_host, double* gamma_dev, double* mean_dev) {
  CCZ_GUARD(h, ccz::gcca_loss_moments_impl(h, moments_dev, n_rows, dims, n_views, eps, k, loss_host, gamma_dev, mean_dev));
}

int ccz_factor_loadings(ccz_handle h, const double* moments_dev, int64_t n_rows, int64_t d, const double* W_dev,
                        int64_t k, double* out_dev) {
  CCZ_GUARD(h, ccz::factor_loadings_impl(h, moments_dev, n_rows, d, W_dev, k, out_dev));
}

}  // extern "C"
