/*
 * libccz -- MI355X (gfx950) CCA solver core: C ABI.
 *
 * The reference (jameschapman19/cca_zoo) is pure Python and has NO FFI: its
 * hot path calls NumPy/SciPy/scikit-learn/torch directly.  This header is the
 * flat C boundary a maintainer would bind (ctypes) to put
 * the device path behind the reference's own seams.  Each entry cites the
 * reference call site(s) it replaces, relative to the reference repo root.
 *
 * Conventions
 *   - return 0 (CCZ_OK) or a negative CCZ_E* code; text via ccz_last_error()
 *   - the caller owns every input/output buffer; the handle owns only scratch
 *   - "dev" pointers are device (HBM) addresses, "host" pointers host addresses
 *   - matrices are row-major, leading dimension in ELEMENTS, sizes int64_t
 *   - a handle is bound to one device and one HIP stream; not thread-safe
 *   - no C++/torch types cross this boundary
 */
#ifndef CCZ_H
#define CCZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCZ_VERSION 150 /* 0.1.5: ccz_k1_route (split-bf16 route of K1 and of the loss's large backward), ccz_moments_last_route,
                          * ccz_loss_last_route, ccz_pool_trim; the loss state grew by 2 D doubles (opaque: ccz_pair_loss_state_bytes).
                          * 0.1.4: ccz_pair_loss_forward / _backward / _state_bytes (the loss as an autograd node in two phases),
                          * ccz_moments_exchange (the whole exchange step), ccz_solve_defer(h, NULL) = await a pending deferral now */

#if defined(__GNUC__)
#define CCZ_API __attribute__((visibility("default")))
#else
#define CCZ_API
#endif

#define CCZ_OK 0
#define CCZ_EINVAL (-1)   /* bad argument                                   -> ValueError   */
#define CCZ_ENOMEM (-2)   /* device allocation failed                       -> MemoryError  */
#define CCZ_EHIP (-3)     /* HIP runtime error                              -> RuntimeError */
#define CCZ_ENOCONV (-4)  /* iterative solver did not converge              -> LinAlgError  */
#define CCZ_ENOTSPD (-5)  /* Cholesky met a non-positive pivot              -> LinAlgError  */
#define CCZ_EUNSUP (-6)   /* unsupported dtype / configuration              -> ValueError   */
#define CCZ_ERCCL (-7)    /* RCCL unavailable / collective failed           -> RuntimeError */

#define CCZ_F32 0
#define CCZ_F64 1

typedef struct ccz_ctx* ccz_handle;

/* One view (an n x cols block of the sample matrix), row-major. */
typedef struct ccz_view {
  const void* data; /* first element (host or device, see the call) */
  int64_t cols;     /* features d_i                                  */
  int64_t ld;       /* elements between consecutive rows (>= cols)   */
} ccz_view;

typedef struct ccz_devinfo {
  char name[128];
  char arch[32];
  int compute_units;
  int wavefront;
  int64_t hbm_bytes;
  int64_t lds_bytes_per_cu;
} ccz_devinfo;

/* ---- lifecycle ---------------------------------------------------------- */
CCZ_API int ccz_version(void);
CCZ_API int ccz_create(ccz_handle* out, int device);
CCZ_API int ccz_destroy(ccz_handle h);
CCZ_API const char* ccz_last_error(ccz_handle h);
/* stream = hipStream_t as void* (torch: torch.cuda.current_stream().cuda_stream); NULL = the handle's own stream.
 * Waits for the handle's pending work first (a host synchronisation: see ccz_stream_adopt for the non-blocking form) */
CCZ_API int ccz_set_stream(ccz_handle h, void* stream);
CCZ_API int ccz_sync(ccz_handle h);
/* Stream-native use from a framework that owns its streams (the DCCA objective inside a training step,
 * deep/_base.py:78-104: the loss is one autograd node between the encoders' forward and backward): instead of
 * draining the caller's stream before a libccz call and libccz's stream after it,
 *   ccz_stream_acquire(h, s):  the handle's stream waits (on the DEVICE) for everything enqueued on s so far;
 *   ccz_stream_release(h, s):  s waits (on the device) for everything the handle has enqueued so far.
 * Neither blocks the host.  s = hipStream_t as void*; NULL = the legacy default stream, with which the handle's own
 * (blocking) stream is ordered implicitly -- then nothing is enqueued at all. */
CCZ_API int ccz_stream_acquire(ccz_handle h, void* stream);
CCZ_API int ccz_stream_release(ccz_handle h, void* stream);
/* ccz_stream_adopt(h, s): from now on the handle ENQUEUES INTO s itself (ordered after what it had pending on its previous
 * stream) -- the objective's kernels then sit in the same hardware queue as the encoders' and no cross-queue signal is
 * waited for on either side (measured: ~0.1 ms per hand-over at configs[3]).  It stays there until the next
 * ccz_stream_acquire (which first returns to the handle's own stream) or ccz_set_stream; ccz_stream_release(h, s) after
 * an adopt of the same s is a no-op.  Sequences the library replays as hipGraphs are launched directly on the legacy
 * default stream (it cannot be captured). */
CCZ_API int ccz_stream_adopt(ccz_handle h, void* stream);
CCZ_API int ccz_device_info(ccz_handle h, ccz_devinfo* out);

/* ---- raw device memory for callers that do not bring torch tensors ------- */
CCZ_API int ccz_dev_alloc(ccz_handle h, void** out, size_t bytes);
CCZ_API int ccz_dev_free(ccz_handle h, void* p);
CCZ_API int ccz_memcpy_h2d(ccz_handle h, void* dst_dev, const void* src_host, size_t bytes);
CCZ_API int ccz_memcpy_d2h(ccz_handle h, void* dst_host, const void* src_dev, size_t bytes);
CCZ_API int ccz_memset0(ccz_handle h, void* dst_dev, size_t bytes);
/* The handle keeps released scratch (solver blocks, the split route's bf16 planes and partial tiles: tens of GB at
 * n = 1e6) in a pool for the next call.  ccz_pool_trim waits for the handle's stream and returns every unused block to
 * the driver -- for callers about to allocate most of HBM themselves (torch.cuda.empty_cache() is the analogue). */
CCZ_API int ccz_pool_trim(ccz_handle h, size_t* released_bytes);

/* ---- K1: second moments ---------------------------------------------------
 * moments = [ G (D x D, ld = D) | colsum (D) ]  float64, D = sum cols, device.
 *   G      += [X_1..X_m]' [X_1..X_m]   (upper-triangular TILES only; the solves read
 *                                        the upper triangle, ccz_moments_symmetrize
 *                                        completes G for other consumers)
 *   colsum += 1' [X_1..X_m]
 * One call handles one row shard; accumulate=0 zeroes `moments` first.  fp32
 * views are multiplied on the fp32 MFMA pipe in row chunks and accumulated in
 * fp64 across chunks; fp64 views use the fp64 MFMA pipe throughout.
 * Replaces: np.linalg.svd(X) _utils/_linalg.py:28 + X1_w.T @ X2_w
 *   linear/_rcca.py:96; np.cov linear/_mcca.py:151-152,166, linear/_gcca.py:101;
 *   v.mean(axis=0) / v - m  _base.py:97-99.
 */
CCZ_API int ccz_moments(ccz_handle h, int dtype, const ccz_view* views, int n_views, int64_t n_rows,
                int views_on_device, double* moments_dev, int accumulate);
/* ccz_moments with its two policies exposed: pilot_mode 0 = never shift, 1 = automatic (ccz_moments; one small
 * read-back of the column sums decides), 2 = always shift fp32 views (no read-back); timed = 0 skips the HIP-event
 * timing of ccz_moments_last_ms, whose read-out makes the host wait for K1 -- with (2, 0) or fp64 views the call only
 * ENQUEUES work on the handle's stream (the stream-native losses use it that way). */
CCZ_API int ccz_moments_opts(ccz_handle h, int dtype, const ccz_view* views, int n_views, int64_t n_rows,
                int views_on_device, double* moments_dev, int accumulate, int pilot_mode, int timed);
CCZ_API int ccz_moments_symmetrize(ccz_handle h, double* moments_dev, int64_t D);
/* Packed form for the one collective of the sharded path: [ upper triangle of G, row-major,
 * D(D+1)/2 | colsum (D) ] -- half the bytes of the full buffer on the wire.  pack: moments -> packed;
 * unpack: packed -> upper triangle of moments (+ colsum); the lower triangle is left untouched. */
CCZ_API int ccz_moments_pack(ccz_handle h, const double* moments_dev, int64_t D, double* packed_dev);
CCZ_API int ccz_moments_unpack(ccz_handle h, const double* packed_dev, int64_t D, double* moments_dev);
/* The same exchange in TWO parts, ordered so that the solve can start before the exchange has finished (SURVEY.md 8(e):
 * the per-view factorizations need the diagonal blocks only).  Blocks layout, doubles:
 *   head = [ upper triangle of C_11 | .. | upper triangle of C_mm | colsum (D) | 1 spare slot for the caller's row count ]
 *   tail = [ C_12 | C_13 | .. | C_(m-1)m ]   (each d_i x d_j, row-major)
 *   packed = [ head | tail ],  sum d_i (d_i + 1) / 2 + D + 1 + sum_{i<j} d_i d_j  =  D (D + 1) / 2 + D + 1 doubles.
 * which: 1 = head, 2 = tail, 3 = both.  unpack's on_stream (hipStream_t as void*, NULL = the handle's stream) lets the
 * tail be unpacked on the stream the collective completes on; ccz_solve_defer(h, event) then makes the NEXT
 * ccz_{rcca,mcca,gcca}_solve wait for `event` (hipEvent_t as void*, recorded after that unpack) on the device right
 * before its first read of an off-diagonal block -- i.e. after the Cholesky chain of the diagonal blocks.
 * ccz_solve_defer(h, NULL): whatever is pending (this registration, a foreign-stream unpack's, ccz_moments_exchange's tail) is
 * awaited NOW on the handle's stream (device-side; the host does not block) and cleared -- for callers that read off-diagonal
 * blocks of the moments through entry points other than the three solves. */
CCZ_API int ccz_moments_pack_blocks(ccz_handle h, const double* moments_dev, int64_t D, const int64_t* dims, int n_views,
                                    double* packed_dev, int which);
CCZ_API int ccz_moments_unpack_blocks(ccz_handle h, const double* packed_dev, int64_t D, const int64_t* dims, int n_views,
                                      double* moments_dev, int which, void* on_stream);
CCZ_API int ccz_solve_defer(ccz_handle h, void* event);
/* ---- the exchange step itself: RCCL all-reduce(sum) over xGMI (SURVEY.md 8(b) "ccz_allreduce_sum_f64", 8(e)) ----------
 * The reference has no counterpart (single process, NumPy).  A caller WITHOUT torch.distributed shards like this:
 *   one process per GPU:  rank 0 calls ccz_comm_unique_id and ships the 128 bytes to the other ranks (file, socket, MPI);
 *     every rank: ccz_comm_init_rank(h, id, world, rank);  per fit: ccz_moments -> ccz_moments_pack[_blocks] ->
 *     ccz_allreduce_sum_f64(h, packed_dev, count) -> ccz_moments_unpack[_blocks] -> ccz_*_solve.
 *   one process, several GPUs:  ccz_comm_init_all(handles, n) once; per fit the packed buffers of all devices go through
 *     ONE grouped call, ccz_allreduce_sum_f64_multi(handles, bufs_dev, n, count).
 * The collective is enqueued on the handle's stream (in place, float64, sum) and does not block the host.  librccl is
 * dlopen'ed at the first of these calls (CCZ_ERCCL if it cannot be found); ccz_comm_destroy (or ccz_destroy) frees the
 * communicator.  ccz_comm_info: world size (0: none) and this handle's rank. */
CCZ_API int ccz_comm_unique_id(ccz_handle h, void* id_out_128);
CCZ_API int ccz_comm_init_rank(ccz_handle h, const void* id_128, int world, int rank);
CCZ_API int ccz_comm_init_all(ccz_handle* handles, int n);
CCZ_API int ccz_comm_info(ccz_handle h, int* world_out, int* rank_out);
CCZ_API int ccz_comm_destroy(ccz_handle h);
CCZ_API int ccz_allreduce_sum_f64(ccz_handle h, double* buf_dev, int64_t count);
CCZ_API int ccz_allreduce_sum_f64_multi(ccz_handle* handles, double* const* bufs_dev, int n, int64_t count);
/* The whole exchange step of a row-sharded fit in ONE call -- what cca_zoo's fit would run between its second-moment pass and
 * its solve (linear/_rcca.py:69-101, _mcca.py:99-197, _gcca.py:80-110 on a rank's rows): moments_dev ([G | s] of THIS rank's
 * n_local rows) is packed in the blocks layout into a buffer the handle keeps between fits, the row count is written into the
 * head's spare slot on the device, head and tail are all-reduced on a stream of the handle's own, the head is unpacked on the
 * handle's stream, the tail behind its collective -- the next ccz_{rcca,mcca,gcca}_solve waits for that on the device right
 * before its first off-diagonal read, i.e. the tail's transfer overlaps the per-view factorizations (any OTHER reader of the
 * off-diagonal blocks calls ccz_solve_defer(h, NULL) first).  *n_total_out: the global row count (the call's only host read).  Needs ccz_comm_init_rank / _init_all.  CCZ_RCCL_LIB=path pins the RCCL
 * library that is dlopen'ed. */
CCZ_API int ccz_moments_exchange(ccz_handle h, double* moments_dev, int64_t D, const int64_t* dims, int n_views,
                                 int64_t n_local, int64_t* n_total_out);
/* Moments are additive over disjoint row sets: y <- alpha x + beta y over the D*D + D doubles of two
 * moment buffers.  With (alpha, beta) = (-1, 1) it turns the moments of all rows into those of the rows
 * outside a cross-validation fold -- the Gram reuse behind cca_zoo_amd.model_selection.GridSearchCV
 * (the reference refits from the data for every fold and setting: cca_zoo/model_selection/_search.py:211-262). */
CCZ_API int ccz_moments_axpby(ccz_handle h, int64_t D, double alpha, const double* x_dev, double beta, double* y_dev);
/* Moments of a contiguous subset [col0, col0 + D_sub) of the stacked columns: the D_sub x D_sub block of G
 * (upper triangle authoritative, as produced by ccz_moments) and the matching column sums, repacked into a
 * [G | s] buffer of width D_sub.  Lets one K1 pass over [confounds | views] serve PartialCCA
 * (cca_zoo/linear/_partialcca.py:67-103: the views' block is then corrected by a rank-(1 + dz) GEMM). */
CCZ_API int ccz_moments_subset(ccz_handle h, const double* moments_dev, int64_t D, int64_t col0, int64_t D_sub,
                               double* subset_dev);
/* kernel timing of the last ccz_moments call on this handle (HIP events on the
 * handle's stream): milliseconds of the Gram kernel(s) and of the column-sum pass */
CCZ_API int ccz_moments_last_ms(ccz_handle h, double* gram_ms, double* colsum_ms);
/* fp32 views whose column means are large against their spread (max_j |mean_j| / std_j > 2, read off the column
 * sums and sums of squares that K1 forms first) are multiplied as (x - p)(x - p)' with the pilot p = fl32(mean of
 * the launch's rows) subtracted while staging, and the shift is undone on the d x d side in fp64 -- the reference
 * centres before any product (_base.py:97-99) and raw fp32 products would cancel catastrophically.  *used = 1 if
 * the last ccz_moments launch on this handle took that path. */
CCZ_API int ccz_moments_last_pilot(ccz_handle h, int* used);
/* Arithmetic route of fp32 views through K1 (fp64 views always run on the fp64 matrix pipe).
 *   CCZ_K1_FP32    v_mfma_f32_32x32x2_f32 on the fp32 rows as they lie (the reference's own precision:
 *                  np.linalg.svd in float32, _utils/_linalg.py:28; X1_w.T @ X2_w, linear/_rcca.py:96);
 *   CCZ_K1_BF16X2  x - pilot = hi + mid (two bf16 planes, written by one transposing pass over the rows), the
 *                  products hi'hi + hi'mid + mid'hi as three v_mfma_f32_32x32x16_bf16 into one fp32 accumulator,
 *                  diag(sum mid^2) added back exactly -- 3 bf16 MFMAs for each fp32 one at 16x the rate; agreement
 *                  with float64 moments is measured beside the fp32 route's in bench.py (k1_rel_err);
 *   CCZ_K1_AUTO    (default) CCZ_K1_BF16X2 where it pays AND is at least as accurate as the fp32 kernel (n >= 32768 rows,
 *                  n D (D+1) >= 1e11, D >= 256), else CCZ_K1_FP32.
 * The environment variable CCZ_K1_ROUTE = fp32 | bf16x2 overrides AUTO.  route = -1 only queries; *previous (may be
 * NULL) receives the handle's setting before the call. */
#define CCZ_K1_AUTO 0
#define CCZ_K1_FP32 1
#define CCZ_K1_BF16X2 2
#define CCZ_K1_FP64 3
CCZ_API int ccz_k1_route(ccz_handle h, int route, int* previous);
/* route the last ccz_moments launch on this handle took (CCZ_K1_FP32 / CCZ_K1_BF16X2 / CCZ_K1_FP64) and, for a timed
 * CCZ_K1_BF16X2 launch, the HIP-event milliseconds of its three stages: the split pass (HBM-bound), the bf16 MFMA kernel
 * and the fp64 reduce of the partial tiles (0 otherwise).  Any pointer may be NULL. */
CCZ_API int ccz_moments_last_route(ccz_handle h, int* route, double* split_ms, double* mfma_ms, double* reduce_ms);

/* ---- fused solves on reduced moments (replicated after the all-reduce) ----
 * Inputs: moments (device; only the upper triangle of G is read, so no symmetrisation is
 * needed), total rows n, per-view widths.
 * Outputs (HOST, float64, row-major): weights packed view after view, each
 * (d_i x k_out); means (D) ; vals (k_out).
 */
/* linear/_rcca.py:69-101 (rCCA.fit; CCA c=0 linear/_cca.py:52; PLS c=1 linear/_pls.py:53) */
CCZ_API int ccz_rcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t dims[2],
                   const double c[2], int center, int k, double* weights_host,
                   double* means_host, double* vals_host, int* k_out);
/* linear/_mcca.py:99-197 (MCCA.fit, _build_A, _build_B, _build_B_pca, gevp) */
CCZ_API int ccz_mcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t* dims,
                   int n_views, const double* c, double eps, int center, int k,
                   double* weights_host, double* means_host, double* vals_host, int* k_out);
/* linear/_gcca.py:80-110 (GCCA.fit) restated in D x D Gram form */
CCZ_API int ccz_gcca_solve(ccz_handle h, const double* moments_dev, int64_t n, const int64_t* dims,
                   int n_views, const double* c, const double* view_weights, double eps,
                   int center, int k, double* weights_host, double* means_host,
                   double* vals_host, int* k_out);

/* ---- dense function seams (device in/out, float64) ------------------------ */
/* full symmetric EVD by one-sided Jacobi: A (d x d, overwritten) -> w (d, descending),
 * V (d x d, row i = eigenvector i).  torch.linalg.eigh deep/objectives.py:19;
 * np.linalg.eigvalsh linear/_mcca.py:170,194, linear/_gcca.py:102 */
CCZ_API int ccz_syevj(ccz_handle h, double* A_dev, int64_t d, double* w_dev, double* V_dev,
              int* sweeps_out);
/* full SVD by one-sided Jacobi: A (p x q) = U diag(s) Vt, r = min(p,q); U (p x r),
 * s (r, descending), Vt (r x q).  np.linalg.svd linear/_rcca.py:97 */
CCZ_API int ccz_gesvj(ccz_handle h, const double* A_dev, int64_t p, int64_t q, double* U_dev,
              double* s_dev, double* Vt_dev, int* sweeps_out);
/* top-k symmetric (generalised) eigenpairs, descending; B_dev may be NULL.
 * V (p x k), B-normalised (v'Bv = 1).  gevp _utils/_linalg.py:44-73 */
CCZ_API int ccz_gevp_topk(ccz_handle h, const double* A_dev, const double* B_dev, int64_t p, int k,
                  double* w_dev, double* V_dev);
/* top-k singular triplets of T (p x q): U (p x k), s (k), V (q x k) */
CCZ_API int ccz_svd_topk(ccz_handle h, const double* T_dev, int64_t p, int64_t q, int k,
                 double* U_dev, double* s_dev, double* V_dev);
/* whitening matrix from the Gram of a centred view: lam (r) descending eigenvalues of
 * Gxx/(n-1), W (d x r) = V ((1-ridge) lam + ridge)^-1/2, r = min(n, d).
 * svd_whiten _utils/_linalg.py:9-41 */
CCZ_API int ccz_whitener(ccz_handle h, const double* Gxx_dev, int64_t d, int64_t n, double ridge,
                 double* W_dev, double* lam_dev, int64_t* r_out);
/* A^-1/2 with eigenvalues clamped at eps.  _inv_sqrtm deep/objectives.py:9-21 */
CCZ_API int ccz_inv_sqrtm(ccz_handle h, const double* A_dev, int64_t d, double eps, double* out_dev);

/* building blocks (exported for tests and for the seams above) */
CCZ_API int ccz_potrf_lower(ccz_handle h, double* A_dev, int64_t d, int64_t lda);
/* X (r x d) <- X L^-T (trans=1) or X L^-1 (trans=0), L lower (d x d) */
CCZ_API int ccz_trsm_right_lower(ccz_handle h, int trans, int64_t r, int64_t d, const double* L_dev,
                         int64_t ldl, double* X_dev, int64_t ldx);
/* C (M x N) = alpha op(A) op(B) + beta C; op = transpose when the flag is 1 */
CCZ_API int ccz_gemm_f64(ccz_handle h, int transA, int transB, int64_t M, int64_t N, int64_t K,
                 double alpha, const double* A_dev, int64_t lda, const double* B_dev, int64_t ldb,
                 double beta, double* C_dev, int64_t ldc);

/* ---- DCCA correlation loss -------------------------------------------------
 * loss = -tr(S11^-1 S12 S22^-1 S21), S from the centred batch (+ eps I), and the
 * closed-form input gradients.  dtype CCZ_F32 or CCZ_F64 for z / grads / loss.
 * Any of g1/g2 may be NULL (forward only).  loss_dev: one element of `dtype`.
 * deep/objectives.py:61-102 (CCALoss.forward) + its autograd backward.
 */
CCZ_API int ccz_cca_loss(ccz_handle h, int dtype, const void* z1_dev, const void* z2_dev, int64_t n,
                 int64_t d1, int64_t d2, int64_t ld1, int64_t ld2, double eps, void* loss_dev,
                 void* g1_dev, void* g2_dev, int64_t ldg1, int64_t ldg2);
/* The sum over all view pairs a < b of that loss for n_views (2 .. 8) views of one batch, any widths -- MCCALoss,
 * deep/objectives.py:138-153 (n_views = 2 IS ccz_cca_loss) -- and its gradient with respect to every view: ONE K1
 * pass over [z_1 .. z_m], ONE Cholesky + inverse per VIEW (the reference re-centres every view and recomputes its
 * S_aa^-1/2 once per PAIR).  z_dev: HOST array of n_views device views; g_dev: NULL (forward only) or a HOST array of
 * n_views device pointers (entries may be NULL), ldg their leading dimensions; loss_dev: one element of `dtype`. */
CCZ_API int ccz_pair_loss(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n, double eps,
                  void* loss_dev, void* const* g_dev, const int64_t* ldg);
/* The same loss in two phases, for callers inside an autograd graph (the reference's loss IS such a node: CCALoss.forward,
 * deep/objectives.py:61-102, is differentiated by torch; its backward receives the upstream gradient of the scalar loss,
 * cca_zoo/deep/_base.py:78-104).  ccz_pair_loss_forward evaluates the loss and leaves what the backward needs -- Gamma, the
 * centring row, for fp32 views also Gamma in fp32 -- in state_dev: caller-owned device memory of
 * ccz_pair_loss_state_bytes(dtype, dims, n_views) bytes (state_dev = NULL: forward only).  ccz_pair_loss_backward writes
 *   g_a = (*grad_out_dev) * d loss / d z_a     (grad_out_dev: ONE element of `dtype` on the device, NULL = 1)
 * for the SAME views (entries of g_dev may be NULL).  Two aligned fp32 views: one fp32 MFMA product that reads the views
 * where they lie; the upstream gradient is applied inside it.  ccz_pair_loss == forward + backward(NULL). */
CCZ_API int64_t ccz_pair_loss_state_bytes(int dtype, const int64_t* dims, int n_views);
CCZ_API int ccz_pair_loss_forward(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n, double eps,
                          void* loss_dev, void* state_dev);
CCZ_API int ccz_pair_loss_backward(ccz_handle h, int dtype, const ccz_view* z_dev, int n_views, int64_t n,
                           const void* state_dev, const void* grad_out_dev, void* const* g_dev, const int64_t* ldg);
/* ccz_cca_loss / ccz_pair_loss never read the factorization's pivot flags back (no host synchronisation between the encoders' forward
 * and backward).  If S_aa + eps I was not positive definite the loss written to loss_dev is NaN and the handle keeps a
 * sticky record: *view = 1 + index of the failing view (0: none since the last query), *pivot = the failing pivot;
 * querying clears it.  synchronise = 0 reads what has been recorded so far without waiting (a wrapper calls this
 * before its NEXT loss call); synchronise = 1 drains the handle's stream first.  (The reference clamps eigenvalues at
 * eps instead, deep/objectives.py:9-21, and cannot fail; with eps > 0 neither can this, short of NaN inputs.) */
CCZ_API int ccz_loss_status(ccz_handle h, int synchronise, int* view, int* pivot);
/* Arithmetic routes of the last loss on this handle: *forward_route = route of its K1 (as ccz_moments_last_route),
 * *backward_route = route of the gradient product ([dz_1 | dz_2] = ([z_1 | z_2] - 1 mean') Gamma): CCZ_K1_BF16X2 when a
 * two-view fp32 backward is a large product (n >= 32768, 2 n D^2 >= 2e11) and the handle's ccz_k1_route is not CCZ_K1_FP32 --
 * the same split arithmetic as K1's (csrc/gemm_split.hip) -- else CCZ_K1_FP32 / CCZ_K1_FP64; 0 before the first backward. */
CCZ_API int ccz_loss_last_route(ccz_handle h, int* forward_route, int* backward_route);
/* The same loss for a batch that is row-sharded over ranks: `moments_dev` holds the batch moments of
 * [z1 | z2] summed over all shards (ccz_moments per rank + one all-reduce), n_rows the total batch size.
 * Returns the loss (host) and, if gamma_dev != NULL, the (d1+d2) x (d1+d2) matrix Gamma and the batch mean
 * (d1+d2) such that  [dz1 | dz2] = ([z1 | z2] - 1 mean') Gamma  for ANY subset of the rows: each rank
 * applies it to its own shard with ccz_transform.  (Reference: the single-process CCALoss.forward,
 * cca_zoo/deep/objectives.py:61-102; a DDP batch would otherwise need an all-gather of the embeddings.) */
CCZ_API int ccz_cca_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, int64_t d1, int64_t d2,
                                 double eps, double* loss_host, double* gamma_dev, double* mean_dev);

/* Sum over all pairs a < b of the CCA loss of views a and b -- MCCALoss, cca_zoo/deep/objectives.py:138-153, which
 * re-centres every view and recomputes its S_aa^-1/2 once per PAIR -- from ONE set of batch moments of [z_1 .. z_m]
 * with ONE Cholesky + inverse per VIEW.  Outputs as ccz_cca_loss_moments (n_views = 2 is that entry). */
CCZ_API int ccz_pair_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, const int64_t* dims,
                                  int n_views, double eps, double* loss_host, double* gamma_dev, double* mean_dev);

/* MAX-VAR GCCA loss of a batch from its moments (ccz_moments over [z_1 .. z_m], summed over the ranks when the
 * batch is row-sharded): loss = -(n-1) sum of the top-k generalised eigenvalues of  C u = lam B u  with C the
 * centred covariance of the stacked views and B = blockdiag(C_ii) + eps I -- the non-zero spectrum of the
 * reference's n x n matrix sum_i H_i H_i' -- and, if gamma_dev != NULL, the D x D matrix Gamma and the batch mean
 * (D) with  [dz_1 .. dz_m] = ([z_1 .. z_m] - 1 mean') Gamma  (apply with ccz_transform).  All D x D work stays on
 * the device.  cca_zoo/deep/objectives.py:155-220 (GCCALoss.forward) + its autograd backward. */
CCZ_API int ccz_gcca_loss_moments(ccz_handle h, const double* moments_dev, int64_t n_rows, const int64_t* dims,
                                  int n_views, double eps, int k, double* loss_host, double* gamma_dev,
                                  double* mean_dev);

/* ---- transform / score -------------------------------
 * out (n x k, dtype) = (X - mean) W ; X,out device; mean (d), W (d x k) device float64.
 * Enqueue-only on the handle's stream (no host synchronisation): follow with ccz_sync / ccz_stream_release /
 * ccz_memcpy_d2h before another stream or the host reads `out`.
 * _base.py:108-123 */
CCZ_API int ccz_transform(ccz_handle h, int dtype, const void* X_dev, int64_t n, int64_t d, int64_t ld,
                  const double* mean_dev, const double* W_dev, int64_t k, void* out_dev,
                  int64_t ldo);

/* Batched Cholesky factor + triangular inverse (building block of the loss and of the blocked solves; exported
 * for tests): `count` <= 8 SPD matrices A_dev[b] (d[b] x d[b], ld d[b], destroyed), L_dev[b] <- lower factor (the
 * part above the diagonal is left untouched), X_dev[b] <- L^-1 (lower; blocks of 64 columns strictly above the
 * diagonal blocks are left untouched -- zero X first if they are read).  X_dev may be NULL (factor only).  All matrices
 * advance together, d_max / 64 + 1 launches in total.  Pointer arrays are HOST arrays of device pointers.
 * torch.linalg.eigh + clamp + V diag(L^-1/2) V' in _inv_sqrtm, deep/objectives.py:9-21, as used by :94-97 */
CCZ_API int ccz_cholinv(ccz_handle h, int count, double* const* A_dev, const int64_t* d, double* const* L_dev,
                        double* const* X_dev);

/* Counter-based standard normals written straight into HBM (the at-scale JointData inputs):
 *   out[r][c] = (accumulate ? out[r][c] : 0) + scale * N(seed, (row0 + r) * row_stride + c)
 * with N(seed, i) a pure function of its arguments (csrc/rng_hash.h::hash_normal_pair; NumPy restatement in
 * oracle/rng.py), so any row range can be regenerated anywhere.  row_stride even and >= cols; dtype CCZ_F32 / CCZ_F64.
 * cca_zoo/datasets/_simulated.py:116-130 (rng.standard_normal for z and the per-view noise) */
CCZ_API int ccz_randn_fill(ccz_handle h, int dtype, void* out_dev, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                           int64_t row0, int64_t row_stride, double scale, int accumulate);

/* Factor loadings of ONE view from its second moments (ccz_moments on that view alone, d x d + d):
 * out (d x k, device, float64) = corr(feature j, variate t) = (C W)_jt / (std_x_j std_z_t) with C the centred
 * covariance, std_z^2 = diag(W'CW), and the reference's guards max(std, 1e-12).  W (d x k) device float64.
 * cca_zoo/_base.py:208-234 (get_factor_loadings: n x d x k products on the host) */
CCZ_API int ccz_factor_loadings(ccz_handle h, const double* moments_dev, int64_t n_rows, int64_t d,
                                const double* W_dev, int64_t k, double* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* CCZ_H */
