// Shared HIP-side declarations of libccz (product build only).
#pragma once

#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

#include "ops.h"

namespace ccz {

struct PoolBlock {
  void* p;
  size_t bytes;
  bool used;
};

struct GraphEntry {
  uint64_t key;
  hipGraphExec_t exec;
  uint64_t tick;
};

struct Impl {
  hipDeviceProp_t props;
  std::vector<PoolBlock> pool;
  // launch-bound fixed-shape sequences (blocked Cholesky / triangular solves: hundreds of tiny kernels)
  // are captured once per (shape, pointers) into a hipGraph and replayed
  std::vector<GraphEntry> graphs;
  uint64_t tick = 0;
  int graphs_on = 1;
  hipStream_t own_stream = nullptr;   // the handle's default stream (blocking: ordered with the null stream)
  hipEvent_t ev[4];
  int* d_flag = nullptr;      // small device scratch: ints
  double* d_small = nullptr;  // small device scratch: 64K doubles
  size_t small_cap = 65536;
  // host -> device streaming of pageable inputs (gram.hip): two pinned bounce buffers, a copy stream and
  // per-slot events; created on first use, kept for the life of the handle
  void* pin_buf[2] = {nullptr, nullptr};
  size_t pin_cap = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t pipe_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // h2d done [2], compute done [2]
  // ring of pinned slots for small asynchronous host -> device copies (tile tables, descriptors)
  static constexpr int kSmallSlots = 32;
  static constexpr size_t kSmallBytes = size_t(64) << 10;
  void* small_pin[kSmallSlots] = {};
  hipEvent_t small_ev[kSmallSlots] = {};
  int small_next = 0;
  // second stream + events for the look-ahead of the super-blocked Cholesky (ops_hip.hip), created on first use
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_ev[2] = {nullptr, nullptr};
  hipEvent_t bj_ev[8] = {};           // evd_block.hip: pair kernel / tile update hand-overs of the two-stream schedule (CCZ_BJ_FUSED=1)
  // device copies of recently used K1 tile tables (gram.hip): the table is a pure function of the views' pointers,
  // widths and strides, so a training loop (same pooled staging buffer every step) never copies one again
  struct TileTab { uint64_t hash; size_t bytes; void* dev; uint64_t tick; };
  std::vector<TileTab> tile_tabs;
  // sticky failure record of the stream-native loss (loss.hip): pinned + mapped host words written by the device
  int* loss_status = nullptr;       // host view
  int* loss_status_dev = nullptr;   // device view of the same words
  // hand-over events between a caller's stream and the handle's stream (ccz_stream_acquire / ccz_stream_release)
  hipEvent_t xs_ev[2] = {nullptr, nullptr};
  // polled waits (ops_hip.hip: wait_stream_short) and the pinned landing buffer of small device -> host copies
  hipEvent_t wait_ev = nullptr;
  void* d2h_pin = nullptr;
  void* d2h_pin_dev = nullptr;      // device view of d2h_pin (host-mapped): written by a copy kernel
  hipEvent_t d2h_tev[2] = {nullptr, nullptr};   // CCZ_TRACE_D2H: timing events around a read-back (this handle's device)
  static constexpr size_t kD2hPinBytes = size_t(4) << 20;
  void* deferred_event = nullptr;   // awaited by the next solve before it reads off-diagonal blocks (ccz_solve_defer, or the
                                    // handle's own event behind an unpack on a foreign stream)
  hipEvent_t defer_own_ev = nullptr;
  bool adopted = false;             // c->stream is a caller's stream (ccz_stream_adopt) until the next acquire
  // cholinv.hip: per-stream sync blocks of the persistent chain kernel (stream, device block); chain_cap > 0 limits its
  // workgroups (set by callers that run throughput work on another stream next to the chain)
  std::vector<std::pair<void*, void*>> chain_sync;
  int chain_cap = 0;
  void* chain_dbg = nullptr;             // CCZ_CHAIN_DEBUG stamps
  struct K1Plan { uint64_t key; void* dev; int wgs; };
  std::vector<K1Plan> k1_plans;          // gram.hip: per-tile row splits of k_gram_f32_fifo_small, by batch shape
  // gram_split.hip: stage boundaries and hand-overs of a split-route launch (grown on demand), and the side stream the split pass
  // of the NEXT row piece runs on under the MFMA kernel of the current one (CU-masked when the runtime allows; created on first use)
  std::vector<hipEvent_t> sp_ev;
  hipStream_t split_stream = nullptr;
  bool split_stream_tried = false;
  int split_stream_req = 0;              // CCZ_SPLIT_PIPE_CUS the side stream was made for
  int split_stream_cus = 0;              // CUs the side stream is confined to (0: no mask)
  struct SplitTab { uint64_t key; void* panels; void* tiles; void* gtiles; int np, ntiles; };
  std::vector<SplitTab> split_tabs;      // gram_split.hip: panel / tile tables by view widths (pointer-free: uploaded once per shape)
  std::vector<std::pair<void*, void*>> colsum_sync;   // gram.hip: per-stream arrival counters of k_colsum_pilot (64 words each, zero between launches)
  // comm.hip: ccz_moments_exchange -- the packed blocks buffer the handle keeps between fits (grown on demand), the stream its
  // collectives run on and the events that tie it to the handle's stream
  void* xchg_buf = nullptr;
  size_t xchg_cap = 0;
  hipStream_t xchg_stream = nullptr;
  hipEvent_t xchg_ev[3] = {nullptr, nullptr, nullptr};   // packed (main -> exchange), head reduced, tail unpacked
  // comm.hip: the RCCL communicator of this handle's device (ncclComm_t), its size and this handle's rank
  void* comm = nullptr;
  int comm_world = 0, comm_rank = -1;
};

inline Impl* impl(ccz_ctx* c) { return static_cast<Impl*>(c->impl); }
inline hipStream_t stream(ccz_ctx* c) { return static_cast<hipStream_t>(c->stream); }

#define CCZ_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      ::ccz::fail(CCZ_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

#define CCZ_LAUNCH_CHECK() CCZ_HIP(hipGetLastError())

#ifdef __HIPCC__
// ---- words that ANOTHER workgroup of the same launch writes or reads ----
// The per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by other CUs' stores
// (/opt/skills/guides/MI355X_MICROARCH.md, "inter-workgroup visibility").  Instead of release / acquire fences
// (buffer_wbl2 + buffer_inv: 1.7 - 6.5 us EACH, per workgroup and hand-over) every shared word is stored write-through
// and loaded past the L1 with 8-byte agent-scope accesses (global_store / global_load ... sc1) -- the guide's
// "{8-B agent atomics both sides}" form; a hand-over is then: every storing wave drains its stores (s_waitcnt vmcnt(0)),
// barrier, ONE lane updates the progress word; the consumer polls that word and loads.
typedef unsigned long long __attribute__((address_space(1)))* gu64_ptr;
typedef unsigned __attribute__((address_space(1)))* gu32_ptr;
__device__ __forceinline__ double ld_shared(const double* p) {
  const unsigned long long u = __hip_atomic_load((gu64_ptr)(reinterpret_cast<uintptr_t>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void st_shared(double* p, double v) {
  __hip_atomic_store((gu64_ptr)(reinterpret_cast<uintptr_t>(p)), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// two consecutive doubles as ONE 16-byte write-through store (p 16-byte aligned): an 8-byte sc1 store is one fabric write per
// lane and costs a wave ~120 cycles to issue (measured: 3100 of them held the chain up for 5.8k cycles per link); the 16-byte
// form moves twice the bytes per instruction at the plain-store rate (MI355X_MICROARCH.md, "stores of each flavour")
typedef unsigned int ccz_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_shared2(double* p, double a, double b) {
  ccz_v4u32 v;
  v[0] = unsigned(__double2loint(a)); v[1] = unsigned(__double2hiint(a));
  v[2] = unsigned(__double2loint(b)); v[3] = unsigned(__double2hiint(b));
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
#endif

// ops_hip.hip: capture-once / replay-later for launch-bound fixed sequences, keyed on shapes AND pointers
uint64_t graph_key_mix(uint64_t h, uint64_t v);
void graph_run_fn(ccz_ctx* c, uint64_t key, const std::function<void()>& fn);
void sync_short(ccz_ctx* c);   // polled wait for the handle's stream (short waits)

// evd_block.hip: one-sided block Jacobi on the rows of W (p a multiple of 64, even leading dimensions)
int jacobi_rows_block(ccz_ctx* c, int64_t p, int64_t q, double* W, int64_t ldw, double* Q, int64_t qc, int64_t ldq, int max_sweeps);

// api.hip: blocks layout of the sharded exchange (ccz.h), pack or unpack, head / tail / both, optionally on a foreign stream
void moments_blocks(ccz_ctx* c, bool pack, double* mom, int64_t D, const int64_t* dims, int m, double* packed, int which,
                    void* on_stream);

// gram.hip: one entry of a launch's tile table -- the two column panels (of one or two views) whose product is one
// 256 x 256 (fp32) / 128 x 128 (fp64) tile of the stacked Gram matrix
struct GramTile {
  const void* a;
  const void* b;
  int64_t lda, ldb;
  int64_t out_row, out_col;
  int32_t wa, wb;
  int32_t diag;
  int32_t pad_;
};

// gram.hip, loss fast path: per-(row chunk, tile) fp32 partial sums of the pilot-shifted batch Gram (partial[(chunk * ntiles +
// tile) * 65536 + 256 i + j]), the exact fp64 column sums and the fp32 pilot they were shifted by -- all pooled scratch
struct GramPartials {
  const GramTile* tiles = nullptr;
  int ntiles = 0;
  int64_t ksplit = 0;
  float* partial = nullptr;
  float* pilot = nullptr;
  double* colsum = nullptr;
  // null: slot of (chunk, tile) = chunk * ntiles + tile, ksplit chunks per tile, full tiles (k_gram_f32).  Else (k_gram_f32_fifo_small):
  // 3 ints per tile {first slot, slots, rows per slot}, slots of a tile contiguous, diagonal tiles in the FIFO kernel's layout
  int* tile_plan = nullptr;
  // split route (gram_split.hip: gram_partials_split_f32): full tiles in the (chunk, tile) layout above, tile_plan == null, and
  const double* msq = nullptr;     // sum_k mid^2 per stacked column: added on the diagonal by the consumer
  char* planes = nullptr;          // the bf16 planes (pooled scratch, released with the rest)
};
// gram_split.hip: the same sums through two bf16 planes per view on the bf16 matrix pipe (hi'hi + hi'mid + mid'hi in one fp32
// accumulator, the diagonal's mid'mid added back exactly): G (upper tiles) += sum_rows (x - pilot)(x - pilot)'.
bool gram_split_worthwhile(int64_t n, int64_t D);
void gram_split_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, double* G, int64_t D, const float* pilot, double* colsum,
                    bool time_it);
// gemm_split.hip: the two-view loss backward [C1 | C2] = alpha (*alpha_dev) ([A1 | A2] - 1 mean') Gamma on the bf16 pipe (same split arithmetic)
bool gemm_split_pair_eligible(int64_t M, int64_t N, int64_t K, int64_t K1, int64_t nsplit, const void* A1, int64_t lda1, const void* A2,
                              int64_t lda2, const void* C1, int64_t ldc1, const void* C2, int64_t ldc2);
void gemm_split_pair(ccz_ctx* c, int64_t M, int64_t N, int64_t K, int64_t K1, float alpha, const float* alpha_dev, const float* A1, int64_t lda1,
                     const float* A2, int64_t lda2, const float* gamma32, const double* corr, const double* mean, float* C1,
                     int64_t ldc1, float* C2, int64_t ldc2, int64_t nsplit);
// project_split.hip: out (n x k <= 64, fp32) = (X - 1 mean') W with the split arithmetic, X converted in registers (no extra pass)
bool project_split_eligible(ccz_ctx* c, int64_t n, int64_t d, int64_t k, int64_t ld, const void* X, int64_t ldo);
void project_split(ccz_ctx* c, const float* X, int64_t n, int64_t d, int64_t ld, const double* mean, const double* W, int64_t k, float* out,
                   int64_t ldo);
bool gram_partials_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, GramPartials* out);
// gram.hip: exact fp64 column sums of all views + the fp32 pilot (their mean) in ONE launch (k_colsum_pilot); `zero_me` (optional,
// nzero doubles) is cleared by the same launch.  false: shape not supported (more than 64 column blocks / 8 views), nothing enqueued
bool colsum_pilot_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, int64_t D, double* colsum, float* pilot, double* zero_me,
                      int64_t nzero);
// gram_split.hip: the same partial sums through the split route (DCCA batches from 4096 rows on); false: does not apply
bool gram_partials_split_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, GramPartials* out);
void gram_partials_release(ccz_ctx* c, GramPartials* gp);

// gram.hip
// pilot_mode: 0 never / 1 automatic (one small host read-back) / 2 always (no host sync) -- fp32 views only;
// time_it: record HIP events around the Gram and column-sum kernels (costs a host wait at the end)
void moments_impl(ccz_ctx* c, int dtype, const ccz_view* views, int n_views, int64_t n_rows,
                  bool on_device, double* moments, bool accumulate, int pilot_mode = 1, bool time_it = true);
// small host -> device copy through a ring of pinned slots: asynchronous on the handle's stream (the pageable source
// may be reused as soon as this returns); larger than a slot falls back to the synchronous copy
void h2d_small(ccz_ctx* c, void* dst, const void* src, size_t bytes);
// typed GEMM used by the loss backward / transform: C (M x N) = alpha A (M x K) B (K x N) + beta C
// A, C of type T (float or double); B is float64 on the device and converted on load.
void gemm_mixed(ccz_ctx* c, int dtype, int64_t M, int64_t N, int64_t K, double alpha, const void* A,
                int64_t lda, const double* B, int64_t ldb, double beta, void* C, int64_t ldc,
                const double* bias_row /* N, subtracted before alpha; may be null */);

// gemm_big.hip: 256x256-tile fp32 GEMM for sample-side products (loss backward, wide transforms)
bool gemm_f32_big_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* A,
                           const void* C);
void gemm_f32_big(ccz_ctx* c, int64_t M, int64_t N, int64_t K, double alpha, const float* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, float* C, int64_t ldc, const double* bias_row);

// the FIFO kernel on prepared fp32 operands with a column-split destination: C1 <- columns [0, nsplit), C2 <- the rest
bool gemm_f32_fifo_split_eligible(int64_t M, int64_t N, int64_t K, int64_t nsplit, const void* C1, int64_t ldc1, const void* C2,
                                  int64_t ldc2);
void gemm_f32_fifo_split(ccz_ctx* c, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B32,
                         const float* bias32, float* C1, int64_t ldc1, float* C2, int64_t ldc2, int64_t nsplit);

// the same with the A operand as two column ranges [A1 | A2] (two views where they lie), a device-side factor on alpha and
// a float64 centring row: the lazy backward of the two-view DCCA loss
bool gemm_f32_fifo_pair_eligible(int64_t M, int64_t N, int64_t K, int64_t K1, int64_t nsplit, const void* A1, int64_t lda1, const void* A2,
                                 int64_t lda2, const void* C1, int64_t ldc1, const void* C2, int64_t ldc2);
void gemm_f32_fifo_pair(ccz_ctx* c, int64_t M, int64_t N, int64_t K, int64_t K1, float alpha, const float* alpha_dev, const float* A1,
                        int64_t lda1, const float* A2, int64_t lda2, const float* B32, const double* bias64, float* C1, int64_t ldc1,
                        float* C2, int64_t ldc2, int64_t nsplit);

// gemm64_big.hip: 128x128-tile fp64 GEMM (solver stage)
bool gemm_f64_big_eligible(bool tA, bool tB, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                           const double* B, int64_t ldb, const double* C, int64_t ldc);
void gemm_f64_big(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                  int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only);

// gemm64_skinny.hip: 128-row stripes x all N <= 192 columns (subspace-iteration applies)
bool gemm_f64_skinny_eligible(bool tA, bool tB, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                              const double* B, int64_t ldb);
void gemm_f64_skinny(ccz_ctx* c, bool tA, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double beta, double* C, int64_t ldc);

// cholinv.hip: batched (<= 8 matrices) Cholesky factor + triangular inverse in d / 64 + 1 launches, no host sync;
// A destroyed, L lower factor, X = L^-1 lower (blocks strictly above the diagonal are NOT written: zero X first if a
// consumer reads them), T scratch of ceil(d / 64) * 4096 doubles per matrix, info_dev[b] = 0x7fffffff or 1 + bad pivot
void cholinv_batched(ccz_ctx* c, int count, double* const* A, const int64_t* lda, const int64_t* d, double* const* L,
                     const int64_t* ldl, double* const* X, const int64_t* ldx, double* const* T, int* info_dev);
// X[b] = L[b]^-1 for <= 8 lower-triangular blocks with known 64-block diagonal inverses T[b] (rows advance together)
void trinv_batched(ccz_ctx* c, int count, const double* const* L, const int64_t* ldl, const int64_t* d, double* const* X,
                   const int64_t* ldx, const double* const* T);
void copy_lower(ccz_ctx* c, int64_t d, const double* src, int64_t lds_, double* dst, int64_t ldd);
// up to 8 independent fp64 products per launch: C = alpha op(A) op(B) + beta C, optional transposed copy Ct = C'
struct MultiGemmArgs {
  const double* A; const double* B; double* C; double* Ct;
  int64_t lda, ldb, ldc, ldct;
  int64_t M, N, K;
  bool tA, tB, lower_only;
  double alpha, beta;
  bool k_lower = false;    // op(A) = X', op(B) = X, X lower triangular (blocks above the diagonal are never read)
  int ksplit = 1;          // > 1: K cut into slices over extra workgroups, C += alpha * product atomically (beta must be 1)
  // optional rider: *dot_acc += dot_scale * sum_ij (alpha op(A) op(B))_ij dotB_ij (one accumulator per launch)
  const double* dotB = nullptr;
  int64_t lddot = 0;
  double dot_scale = 0.0;
  double* dot_acc = nullptr;
};
void gemm_f64_multi(ccz_ctx* c, int count, const MultiGemmArgs* problems);

}  // namespace ccz
