// Batched Cholesky factor + triangular inverse for the small / medium SPD blocks of the hot path
// (the S_ii + eps I of the DCCA loss, cca_zoo/deep/objectives.py:86-97, and the diagonal super-blocks of the
// blocked factorizations behind rCCA / MCCA / GCCA), plus the batched 64-tile fp64 GEMM that consumes them.
//
// Why a new kernel family: at d ~ 512 the right-looking blocked Cholesky is a chain of d / 64 diagonal-block
// factorizations (each a sequential 64-pivot recurrence on ONE wavefront) with two small GEMM launches between
// consecutive links and a third pass for the triangular inverse -- ~60 dependent launches, 1.2 ms, for 0.1 GFLOP.
// Here ONE launch per block column does everything that can run concurrently:
//
//   launch j:  * every trailing tile (i, k), j < k <= i, recomputes the two panel blocks it needs from the
//                still-unfactored column,  L_ij = A_ij T_j,  L_kj = A_kj T_j  (T_j = L_jj^-T, 64 x 64), and applies
//                A_ik -= L_ij L_kj'  (3 x 64^3 MFMA flops instead of 1: the panel solve needs no launch of its own
//                and no workgroup waits for another);
//              * the workgroup that owns tile (j+1, j+1) goes straight on to factor it (LOOK-AHEAD: one wave,
//                shift-register recurrence, Newton rsqrt instead of sqrt + divide, reciprocal diagonal kept for
//                the inverse) and publishes L_(j+1)(j+1) and T_(j+1) for the next launch;
//              * row j of X = L^-1 is formed from the rows above it:  X_jk = -T_j' sum_{t=k}^{j-1} L_jt X_tk.
//
// so a d x d factor + inverse costs d / 64 + 1 launches whose critical path is the diagonal recurrence alone.
// Several matrices (<= 8) share every launch.  All tile products run on v_mfma_f64_16x16x4_f64 from LDS tiles.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hip_common.h"

namespace ccz {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int CB = 64;          // block edge
constexpr int CLD = CB + 1;     // LDS tile stride (doubles): row and column walks both conflict-free for the wave code
constexpr int CTILE = CB * CLD; // doubles per LDS tile
constexpr int CPLD = 17;        // row stride (doubles) of the 64 x 16 panel image of the 16-column form
constexpr size_t CHOLINV_LDS = (size_t(3) * CTILE + 2 * CB + 8 + CPLD * CB) * sizeof(double);   // 3 tiles, Rd, flags, panel image
constexpr size_t CHAIN_LDS = (size_t(4) * CTILE + 2 * CB + 8 + CPLD * CB) * sizeof(double);     // + a fourth tile (k_cholinv_chain)

// ---------------------------------------------------------------------------
// 64 x 64 tile products on the fp64 matrix pipe, operands in LDS tiles (stride CLD).
// Wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 result; acc[t] is the 16 x 16 tile of columns 16 t ..
// A operand lane layout: (m = lane & 15, k = lane >> 4); B: (k = lane >> 4, n = lane & 15);
// C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.
// TA: the A tile is stored transposed (As[k][m]); TB: the B tile is stored transposed (Bs[n][k]).
// ---------------------------------------------------------------------------
template <bool TA, bool TB>
__device__ __forceinline__ void tile_mm(const double* As, const double* Bs, int w, int lane, v4f64 (&acc)[4]) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll 4
  for (int k0 = 0; k0 < CB; k0 += 4) {
    const int kk = k0 + lk;
    const double a = TA ? As[kk * CLD + 16 * w + lr] : As[(16 * w + lr) * CLD + kk];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = TB ? Bs[(16 * t + lr) * CLD + kk] : Bs[kk * CLD + 16 * t + lr];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
}

// A (64 x 64, row-major) times an UPPER-triangular B (row-major: T_j = L_jj^-T): column tile t only needs k < 16 (t + 1),
// 40 MFMAs per wave instead of 64
__device__ __forceinline__ void tile_mm_upper(const double* As, const double* Bs, int w, int lane, v4f64 (&acc)[4]) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < CB; k0 += 4) {
    const int kk = k0 + lk;
    const double a = As[(16 * w + lr) * CLD + kk];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (k0 < 16 * (t + 1)) {
        const double b = Bs[kk * CLD + 16 * t + lr];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void acc_zero(v4f64 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
}

// accumulator tile -> LDS tile (row-major, stride CLD), scaled
__device__ __forceinline__ void acc_to_lds(const v4f64 (&acc)[4], double* dst, int w, int lane, double scale) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * w + (lane >> 4) + 4 * r) * CLD + 16 * t + (lane & 15)] = scale * acc[t][r];
}

// global (rows x cols valid, zero elsewhere) -> registers -> LDS tile; 256 threads, coalesced rows: a thread owns
// column tid & 63 of rows (tid >> 6) + 4 i.  All 16 loads are issued before the first use (one memory latency per tile,
// not sixteen); fetch / store are separate so that a K loop can fetch chunk i + 1 while chunk i is multiplied.
struct TileRegs { double v[16]; };

__device__ __forceinline__ void fetch_tile(TileRegs& t, const double* __restrict__ src, int64_t ld, int rows, int cols, int tid) {
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + 4 * i;
    t.v[i] = (r < rows && c < cols) ? src[int64_t(r) * ld + c] : 0.0;
  }
}

template <bool TRANS>
__device__ __forceinline__ void store_tile(double* dst, const TileRegs& t, int tid) {
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + 4 * i;
    if (TRANS) dst[c * CLD + r] = t.v[i]; else dst[r * CLD + c] = t.v[i];
  }
}

template <bool TRANS>
__device__ __forceinline__ void load_tile(double* dst, const double* __restrict__ src, int64_t ld, int rows, int cols, int tid) {
  TileRegs t;
  fetch_tile(t, src, ld, rows, cols, tid);
  store_tile<TRANS>(dst, t, tid);
}

// ---- tiles that ANOTHER workgroup of the same launch writes (k_cholinv_chain): ld_shared / st_shared, hip_common.h ----
__device__ __forceinline__ void fetch_tile_shared(TileRegs& t, const double* src, int64_t ld, int rows, int cols, int tid) {
  const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + 4 * i;
    t.v[i] = (r < rows && c < cols) ? ld_shared(src + int64_t(r) * ld + c) : 0.0;
  }
}
template <bool TRANS>
__device__ __forceinline__ void load_tile_shared(double* dst, const double* src, int64_t ld, int rows, int cols, int tid) {
  TileRegs t;
  fetch_tile_shared(t, src, ld, rows, cols, tid);
  store_tile<TRANS>(dst, t, tid);
}

// ---------------------------------------------------------------------------
// One wavefront factors the 64 x 64 SPD tile held in `Xs` (row-major, stride CLD, identity-padded beyond the valid
// size) and forms the inverse of the factor:
//   Ls[t * CLD + i] = L[i][t]            (column-major factor, left in LDS)
//   Xs[c * CLD + t] = (L^-T)[c][t]       (row-major L^-T)
// Lane i owns row i of the current Schur complement in 64 registers used as a SHIFT REGISTER (step j reads column
// j from a[0] and writes the updated row one register down: all register indices are compile-time constants while
// j is a run-time counter); the multipliers come back as wave-uniform LDS broadcasts.  1 / sqrt(pivot) is a
// v_rsq_f64 seed + two Newton steps (no IEEE sqrt / divide chains on the critical path); it is also 1 / L_jj, which
// the forward substitution of the inverse phase multiplies by instead of dividing.  Returns the first non-positive
// pivot (0-based) or 0x7fffffff.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = __builtin_fma(-x * y, y, 1.0);       // 1 - x y^2
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}

// v_rsq_f64 seed + ONE third-order step (y (1 + e/2 + 3 e^2/8), e = 1 - x y^2): five dependent operations instead of
// the seven of two Newton steps -- a dependent fp64 operation costs 32 cycles on this chip (profiles/r02c_clock_probe.md)
// and this is the pivot chain of every 64 x 64 block.  Relative error ~ (5/16) e0^3 with e0 the seed's (~2^-26): rounding.
__device__ __forceinline__ double rsqrt_cubic(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double t = x * y0;
  const double e = __builtin_fma(-t, y0, 1.0);
  const double p = __builtin_fma(e, 0.375, 0.5);
  const double q = y0 * e;
  return __builtin_fma(q, p, y0);
}

// 16-byte reads of a buffer that is written through plain `double` lvalues: may_alias, or type-based alias analysis
// may reorder / forward across the two access types
typedef double v2f64 __attribute__((ext_vector_type(2), may_alias));

// The multipliers of a step are 63 wave-uniform doubles from one LDS column: read as 32 ds_read_b128 (the column
// starts at byte 528 j: 16-byte aligned) -- a single wave issues narrow LDS reads at a fifth of the array rate and the
// 63 ds_read_b64 of the first version were the critical path (1000 cycles per pivot).
// `progress` (LDS): number of finished columns, published with workgroup-scope release after the column and its
// reciprocal diagonal are in LDS; the inverse wave consumes column t as soon as progress > t.
template <int KMAX>
__device__ __forceinline__ void cf_chol_steps(double (&a)[CB], int j_begin, int j_end, int lane, double* Ls, double* Rd, int* progress,
                                              int& first_bad) {
#pragma unroll 1
  for (int j = j_begin; j < j_end; ++j) {
    const double col = a[0];
    double piv = bcast_lane(col, j);
    const bool bad = !(piv > 0.0);                 // wave-uniform; NaN counts as bad
    first_bad = bad ? min(first_bad, j) : first_bad;
    piv = bad ? 1.0 : piv;
    const double rs = rsqrt_newton(piv);
    const double l = col * rs;                     // lanes >= j: L[lane][j]; lanes < j hold junk that nobody reads
    double* lcol = Ls + j * CLD;
    lcol[lane] = lane >= j ? l : 0.0;
    if (lane == 0) Rd[j] = rs;                     // 1 / L_jj
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(progress, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const v2f64* lrow2 = reinterpret_cast<const v2f64*>(Ls) + j * (CLD + 1) / 2;   // &Ls[j * CLD + j], (CLD + 1) j / 2 pairs
    // reads past row 63 land in the slack behind the tile and only feed junk registers
#pragma unroll
    for (int p = 0; p <= KMAX / 2; ++p) {
      const v2f64 m = lrow2[p];
      if (2 * p >= 1 && 2 * p <= KMAX) a[2 * p - 1] = a[2 * p] - l * m[0];
      if (2 * p + 1 <= KMAX) a[2 * p] = a[2 * p + 1] - l * m[1];
      if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int KMAX>
__device__ __forceinline__ void cf_inv_steps(double (&v)[CB], int t_begin, int t_end, int lane, const double* Ls, const double* Rd,
                                             int* progress, double* Xs) {
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
    // wait until the factorization wave has published column t (uniform branch: lane 0 polls, everyone follows)
    while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= t) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const v2f64* lcol2 = reinterpret_cast<const v2f64*>(Ls) + t * (CLD + 1) / 2;    // &Ls[t * CLD + t]: L[t][t], L[t+1][t], ...
    const double x = v[0] * Rd[t];
    Xs[lane * CLD + t] = x;                        // (L^-1)[t][lane] = (L^-T)[lane][t]
#pragma unroll
    for (int p = 0; p <= KMAX / 2; ++p) {
      const v2f64 m = lcol2[p];
      if (2 * p >= 1 && 2 * p <= KMAX) v[2 * p - 1] = v[2 * p] - m[0] * x;
      if (2 * p + 1 <= KMAX) v[2 * p] = v[2 * p + 1] - m[1] * x;
      if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Waves 0 and 1 of the calling workgroup (tid < 128), after a __syncthreads() that made the input tile `In`
// (row-major, stride CLD, identity-padded) visible and with *progress == 0:
//   wave 0: Cholesky recurrence, columns into Ls[t * CLD + i] = L[i][t], reciprocal diagonal into Rd
//   wave 1: forward substitution L x = e_c for all 64 right-hand sides at once, one step behind wave 0,
//           Xs[c * CLD + t] = (L^-T)[c][t]
// Returns (wave 0) the first non-positive pivot (0-based) or 0x7fffffff.
__device__ __forceinline__ int wave_factor_lds(double* Ls, const double* In, double* Xs, double* Rd, int* progress, int tid) {
  const int lane = tid & 63;
  int first_bad = 0x7fffffff;
  if (tid < 64) {
    double a[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) a[k] = In[lane * CLD + k];
    cf_chol_steps<63>(a, 0, 16, lane, Ls, Rd, progress, first_bad);
    cf_chol_steps<47>(a, 16, 32, lane, Ls, Rd, progress, first_bad);
    cf_chol_steps<31>(a, 32, 48, lane, Ls, Rd, progress, first_bad);
    cf_chol_steps<15>(a, 48, 64, lane, Ls, Rd, progress, first_bad);
  } else {
    double v[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) v[k] = (k == lane) ? 1.0 : 0.0;
    cf_inv_steps<63>(v, 0, 16, lane, Ls, Rd, progress, Xs);
    cf_inv_steps<47>(v, 16, 32, lane, Ls, Rd, progress, Xs);
    cf_inv_steps<31>(v, 32, 48, lane, Ls, Rd, progress, Xs);
    cf_inv_steps<15>(v, 48, 64, lane, Ls, Rd, progress, Xs);
  }
  return first_bad;
}

constexpr int CMAXB = 8;
struct CholInvBatch {
  double* A[CMAXB];       // working matrices: lower triangle read and updated in place (destroyed)
  double* L[CMAXB];       // out: lower Cholesky factor
  double* X[CMAXB];       // out: L^-1 (lower); may be null (factor only)
  double* T[CMAXB];       // scratch: ceil(d / 64) blocks of 64 x 64, T_j = L_jj^-T (row-major, identity-padded)
  int64_t lda[CMAXB], ldl[CMAXB], ldx[CMAXB], d[CMAXB];
  int first[CMAXB + 1];   // prefix sums of the work items of this launch
  int count;
  int inv_only;           // 1: L and T are given; only the rows of X = L^-1 are formed (no trailing updates)
};

// ---------------------------------------------------------------------------
// MFMA form of the 64 x 64 factorization (default).  The shift-register recurrence above spends ~1000 cycles per
// pivot on the rank-1 update (63 FMAs + 32 wave-uniform LDS reads issued by ONE wave); here the Schur complement
// lives in v_mfma_f64_16x16x4_f64 accumulators (10 lower 16 x 16 tiles, 40 registers) and is updated once per
// PANEL of four pivots by <= 10 MFMAs; only the 4-column panel itself goes through the scalar recurrence
// (lane = row: four values per lane, multipliers by v_readlane, no LDS on the pivot chain):
//   per panel:  panel columns accumulators -> LDS (64 x 4)  ->  4 pivots in registers  ->  L panel -> LDS
//               ->  four ds_read_b64 give the A (and, by symmetry, B) fragments  ->  acc[ti][tj] -= L_ti L_tj'
// All 16 panels are unrolled (every register / lane index is a constant).
// The inverse is blocked as well: the four 16 x 16 diagonal blocks are inverted by substitution (one wave each, 16
// steps), the six blocks below the diagonal follow from 16 x 16 MFMA products, column j on wave j:
//   X_ij = -X_ii sum_{t=j}^{i-1} L_it X_tj
// ---------------------------------------------------------------------------
// Lanes of ONE wave exchange data through LDS below.  LDS operations of a wave execute in order, but the COMPILER
// reasons per thread: without a fence it forwards a thread's own earlier store to its later load of the same address
// even though another lane has overwritten it in between (first version of chol_panel: rows 8..15 of every panel
// kept the previous panel's values).  A wavefront-scope release/acquire pair is the (free) way to say "shared".
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int Q>
__device__ __forceinline__ void chol_panel(v4f64 (&acc)[4][4], int lane, double* Ls, double* Rd, double* PL, int& first_bad) {
  constexpr int C0 = 4 * Q, TJ0 = Q / 4, CL = C0 % 16;
  // 1. the panel's four columns of the Schur complement: accumulators -> PL[row][0..3]
  const int pc = (lane & 15) - CL;
  if (pc >= 0 && pc < 4) {
#pragma unroll
    for (int ti = TJ0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) PL[(16 * ti + (lane >> 4) + 4 * r) * 4 + pc] = acc[ti][TJ0][r];
  }
  wave_lds_sync();
  // 2. lane = row: its four panel values (rows above the panel's tile row were not written: never used)
  // (plain double accesses on purpose: reading this buffer through a vector type while it is written through
  // `double` lets type-based alias analysis forward the PREVIOUS panel's stores to the lanes that did not write)
  double pv[4] = {PL[lane * 4 + 0], PL[lane * 4 + 1], PL[lane * 4 + 2], PL[lane * 4 + 3]};
  double l[4];
  // 3. four pivots: everything in registers, multipliers by v_readlane with constant lane numbers
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    double piv = bcast_lane(pv[t], C0 + t);
    const bool bad = !(piv > 0.0);
    first_bad = bad ? min(first_bad, C0 + t) : first_bad;
    piv = bad ? 1.0 : piv;
    const double rs = rsqrt_newton(piv);
    l[t] = lane >= C0 + t ? pv[t] * rs : 0.0;       // select, not multiply: rows above hold stale values
    if (lane == 0) Rd[C0 + t] = rs;                 // 1 / L_jj for the substitution of the inverse
#pragma unroll
    for (int u = t + 1; u < 4; ++u) pv[u] -= l[t] * bcast_lane(l[t], C0 + u);
  }
  // 4. L panel: column-major factor Ls[col][row] (final output) and the 64 x 4 operand image PL[row][0..3]
#pragma unroll
  for (int t = 0; t < 4; ++t) Ls[(C0 + t) * CLD + lane] = l[t];
#pragma unroll
  for (int t = 0; t < 4; ++t) PL[lane * 4 + t] = l[t];
  if (Q == 15) return;
  wave_lds_sync();
  // 5. acc[ti][tj] -= L[rows of ti][panel] L[rows of tj][panel]'   (A fragment: (m = lane & 15, k = lane >> 4);
  //    the B fragment of tile column tj is the A fragment of tile row tj)
  double a[4];
#pragma unroll
  for (int t4 = TJ0; t4 < 4; ++t4) a[t4] = PL[(16 * t4 + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
  for (int tj = TJ0; tj < 4; ++tj)
#pragma unroll
    for (int ti = tj; ti < 4; ++ti)
      acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[ti], a[tj], acc[ti][tj], 0, 0, 0);
  wave_lds_sync();          // the next panel overwrites PL
}

template <int Q>
struct CholPanels {
  static __device__ __forceinline__ void run(v4f64 (&acc)[4][4], int lane, double* Ls, double* Rd, double* PL, int& first_bad) {
    CholPanels<Q - 1>::run(acc, lane, Ls, Rd, PL, first_bad);
    chol_panel<Q>(acc, lane, Ls, Rd, PL, first_bad);
  }
};
template <>
struct CholPanels<-1> {
  static __device__ __forceinline__ void run(v4f64 (&)[4][4], int, double*, double*, double*, int&) {}
};

// wave 0: In (row-major, stride CLD) -> Ls[col * CLD + row] = L[row][col] (whole columns, zeros above the diagonal),
// Rd[j] = 1 / L_jj.  Returns the first non-positive pivot or 0x7fffffff.
__device__ __forceinline__ int chol64_mfma(const double* In, double* Ls, double* Rd, double* PL, int lane) {
  v4f64 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[ti][tj][r] = tj <= ti ? In[(16 * ti + (lane >> 4) + 4 * r) * CLD + 16 * tj + (lane & 15)] : 0.0;
  int first_bad = 0x7fffffff;
  CholPanels<15>::run(acc, lane, Ls, Rd, PL, first_bad);
  return first_bad;
}

// ---------------------------------------------------------------------------
// MFMA form with 16-COLUMN panels (round 5, the default).  The 4-column form above makes 15 accumulator -> LDS -> lane
// = row -> LDS -> MFMA-fragment round trips per block (~500 cycles each) around a pivot chain of ~290 cycles per pivot;
// here a panel is a whole tile column: ONE trip through LDS brings it into lane = row registers (16 values per lane),
// all 16 pivots run in registers -- the update of the panel's remaining columns by pivot t is (15 - t) readlane + FMA
// pairs that are independent of each other and fill the latency shadow of the NEXT pivot's rsqrt chain -- and the
// trailing tiles (strictly right of the panel) take the panel as FOUR k-steps of v_mfma_f64_16x16x4_f64 whose operand
// fragments are read straight from the column-major factor image.  3 + 3 LDS trips per block instead of 31, and the
// tile column of panel 0 never visits the accumulators at all.
// ---------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void chol_panel16(v4f64 (&acc)[4][4], int lane, const double* In, double* Ls, double* Rd, double* PL,
                                             int& first_bad) {
  constexpr int C0 = 16 * P;
  double pv[16], l[16];
  if (P == 0) {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) pv[cc] = In[lane * CLD + cc];
  } else {
    // the panel's tile column: accumulators -> PL[row][0..15] (rows >= C0 only: rows above the panel are never used)
#pragma unroll
    for (int ti = P; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) PL[(16 * ti + (lane >> 4) + 4 * r) * CPLD + (lane & 15)] = acc[ti][P][r];
    wave_lds_sync();
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) pv[cc] = PL[lane * CPLD + cc];
  }
  // The pivot chain.  A dependent fp64 operation costs 32 cycles here, so what a pivot costs is the number of DEPENDENT
  // operations between two pivots: the next pivot is formed wave-uniformly from broadcasts taken BEFORE this pivot's
  // reciprocal is known,  piv' = a - b^2 / piv  (a, b = entries (t+1, t+1), (t+1, t) of the current Schur complement), with
  // 1 / piv from the same v_rsq_f64 seed by a third-order step -- rsq, 2 to e, e + e^2, w, fma: six dependent operations,
  // no lane exchange on the chain.  The column itself (x rs, third-order rsqrt from the same seed), its broadcasts and the
  // updates of the panel's other columns fill the shadow.  A non-positive pivot is recorded and NOT repaired: NaN / inf flow
  // through the rest of the block, whose result is reported as failed anyway.  Lanes above the diagonal carry junk that
  // nobody reads (the factor image is only ever read on and below its diagonal).
  double piv = bcast_lane(pv[0], C0);
  double rd_mine = 0.0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const bool bad = !(piv > 0.0);                     // wave-uniform; NaN counts as bad; off the chain
    first_bad = bad ? min(first_bad, C0 + t) : first_bad;
    const double y0 = __builtin_amdgcn_rsq(piv);
    const double tq = piv * y0;
    const double e = __builtin_fma(-tq, y0, 1.0);      // 1 - piv y0^2
    const double y2 = y0 * y0;
    const double ge = __builtin_fma(e, e, e);
    const double w = __builtin_fma(y2, ge, y2);        // 1 / piv  (relative error e^3)
    double piv_next = 0.0;
    if (t < 15) {
      const double a = bcast_lane(pv[t + 1 < 16 ? t + 1 : 15], C0 + t + 1), bb = bcast_lane(pv[t], C0 + t + 1);
      piv_next = __builtin_fma(-(bb * bb), w, a);
    }
    const double rs = __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);   // 1 / sqrt(piv)  (relative error (5/16) e^3)
    l[t] = pv[t] * rs;
    rd_mine = (lane == C0 + t) ? rs : rd_mine;         // 1 / L_jj for the substitution of the inverse
#pragma unroll
    for (int u = t + 1; u < 16; ++u) pv[u] = __builtin_fma(-l[t], bcast_lane(l[t], C0 + u), pv[u]);
    piv = piv_next;
  }
  if (lane >= C0 && lane < C0 + 16) Rd[lane] = rd_mine;
  // column-major factor image Ls[col][row] -- also the MFMA operand image
#pragma unroll
  for (int t = 0; t < 16; ++t) Ls[(C0 + t) * CLD + lane] = l[t];
  if (P == 3) return;
  wave_lds_sync();
  // acc[ti][tj] -= L[rows of ti][panel] L[rows of tj][panel]'  for the tiles right of the panel: A fragment of k-step ks
  // = L[16 ti + (lane & 15)][C0 + 4 ks + (lane >> 4)]; the B fragment of tile column tj is the A fragment of tile row tj
  double a[4][4];
#pragma unroll
  for (int ti = P + 1; ti < 4; ++ti)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[ti][ks] = Ls[(C0 + 4 * ks + (lane >> 4)) * CLD + 16 * ti + (lane & 15)];
#pragma unroll
  for (int tj = P + 1; tj < 4; ++tj)
#pragma unroll
    for (int ti = tj; ti < 4; ++ti)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[ti][ks], a[tj][ks], acc[ti][tj], 0, 0, 0);
}

// wave 0: In (row-major, stride CLD, symmetric, identity-padded) -> Ls[col * CLD + row] = L[row][col], Rd[j] = 1 / L_jj
__device__ __forceinline__ int chol64_p16(const double* In, double* Ls, double* Rd, double* PL, int lane) {
  v4f64 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[ti][tj][r] = (tj >= 1 && tj <= ti) ? In[(16 * ti + (lane >> 4) + 4 * r) * CLD + 16 * tj + (lane & 15)] : 0.0;
  int first_bad = 0x7fffffff;
  chol_panel16<0>(acc, lane, In, Ls, Rd, PL, first_bad);
  chol_panel16<1>(acc, lane, In, Ls, Rd, PL, first_bad);
  chol_panel16<2>(acc, lane, In, Ls, Rd, PL, first_bad);
  chol_panel16<3>(acc, lane, In, Ls, Rd, PL, first_bad);
  return first_bad;
}

// all four waves: the diagonal 16 x 16 blocks of Xs[col * CLD + row] = (L^-1)[row][col] (== row-major L^-T) from Ls / Rd, and
// zeros in the blocks above the block diagonal.  No barrier inside.
__device__ __forceinline__ void inv64_phaseA(const double* __restrict__ Ls, const double* __restrict__ Rd, double* __restrict__ Xs, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  // zero the blocks above the block diagonal (the substitution leaves exact zeros inside the diagonal blocks)
  for (int e = tid; e < CB * CB; e += 256) {
    const int row = e & 63, col = e >> 6;
    if ((row >> 4) < (col >> 4)) Xs[col * CLD + row] = 0.0;
  }
  // wave w inverts the 16 x 16 diagonal block w by forward substitution, lane c < 16 = right-hand side e_c
  const int o = 16 * w;
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (k == lane) ? 1.0 : 0.0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const double x = v[0] * Rd[o + t];
    if (lane < 16) Xs[(o + lane) * CLD + o + t] = x;           // (L^-1)[o + t][o + lane]
    const double* lcol = Ls + (o + t) * CLD + o + t;            // L[o + t + k][o + t], wave-uniform
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k - 1] = (t + k < 16) ? v[k] - lcol[k] * x : 0.0;
  }
}

// ONE wave forms column block j of L^-1 below the diagonal, top to bottom (needs the diagonal blocks of phase A); only
// wave-level synchronisation inside
__device__ __forceinline__ void inv64_phaseB_col(const double* Ls, double* Xs, int lane, int j) {
  const int lr = lane & 15, lk = lane >> 4;
  for (int i = j + 1; i < 4; ++i) {
    v4f64 sacc = {0.0, 0.0, 0.0, 0.0};
    for (int t = j; t < i; ++t) {
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += 4) {
        const int k = k0 + lk;
        const double a = Ls[(16 * t + k) * CLD + 16 * i + lr];      // L[16 i + lr][16 t + k]
        const double b = Xs[(16 * j + lr) * CLD + 16 * t + k];      // (L^-1)[16 t + k][16 j + lr]
        sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, sacc, 0, 0, 0);
      }
    }
    // X_ij = -X_ii S : the B fragment of k-step k0 = 4 r is accumulator register r of S (row = lk + 4 r, col = lr)
    v4f64 xacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double a = Xs[(16 * i + 4 * r + lk) * CLD + 16 * i + lr];   // (L^-1)[16 i + lr][16 i + 4 r + lk]
      xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, sacc[r], xacc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Xs[(16 * j + lr) * CLD + 16 * i + lk + 4 * r] = xacc[r];   // (L^-1)[16 i + lk + 4 r][16 j + lr]
    wave_lds_sync();        // the next block row of this column reads what OTHER lanes of this wave have just written
  }
}

// all four waves: Xs[col * CLD + row] = (L^-1)[row][col]  (== row-major L^-T) from Ls / Rd
__device__ __forceinline__ void inv64_mfma(const double* Ls, const double* Rd, double* Xs, int tid) {
  inv64_phaseA(Ls, Rd, Xs, tid);
  __syncthreads();
  inv64_phaseB_col(Ls, Xs, tid & 63, tid >> 6);       // wave j forms column block j
}

// ONE wave's 16-row strip of  X = A L^-T  (X L' = A) from the factor image Ls[col * CLD + row] and the diagonal 16 x 16
// blocks of Xs (phase A) alone -- the off-diagonal blocks of L^-1 are not needed, so the chain does not wait for them.
// Worked on TRANSPOSED 16 x 16 blocks: with X_c' in the MFMA accumulator layout (row = lk + 4 r, col = lr) an accumulator
// IS the B fragment of the next product, so the whole recurrence
//   R_c' = A_c' - sum_{s<c} L_cs X_s',   X_c' = Dinv_c R_c'
// runs in registers: 40 MFMAs, no LDS round trip.  As: the A tile (row-major, stride CLD); out: X (row-major) for the
// strip's rows.
__device__ __forceinline__ void solve_strip_LT(const double* As, const double* Ls, const double* Xs, double* out, int w, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  v4f64 xt[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v4f64 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = As[(16 * w + lr) * CLD + 16 * c + lk + 4 * q];     // R_c'[lk + 4 q][lr] = A[16 w + lr][16 c + lk + 4 q]
#pragma unroll
    for (int sb = 0; sb < c; ++sb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double a = Ls[(16 * sb + 4 * q + lk) * CLD + 16 * c + lr];                   // L[16 c + lr][16 sb + 4 q + lk]
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, xt[sb][q], r, 0, 0, 0);
      }
    v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double a = Xs[(16 * c + 4 * q + lk) * CLD + 16 * c + lr];                      // (L^-1)[16 c + lr][16 c + 4 q + lk]
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, r[q], x, 0, 0, 0);
    }
    xt[c] = x;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(16 * w + lr) * CLD + 16 * c + lk + 4 * q] = x[q];   // X[16 w + lr][16 c + lk + 4 q]
  }
}

// rows [0, rows) x 64 columns of a row-major LDS tile (stride CLD) -> global (ld ldg), write-through; `nthr` threads (tid <
// nthr) take part.  16-byte stores when the destination rows are 16-byte aligned.
__device__ __forceinline__ void store_rows_shared(double* dst, int64_t ldg, const double* src, int rows, int tid, int nthr) {
  if (((reinterpret_cast<uintptr_t>(dst) | uintptr_t(ldg * 8)) & 15) == 0) {
    for (int e = tid; e < rows * 32; e += nthr) {
      const int r = e >> 5, c2 = (e & 31) * 2;
      st_shared2(dst + int64_t(r) * ldg + c2, src[r * CLD + c2], src[r * CLD + c2 + 1]);
    }
  } else {
    for (int e = tid; e < rows * 64; e += nthr) {
      const int r = e >> 6, c = e & 63;
      st_shared(dst + int64_t(r) * ldg + c, src[r * CLD + c]);
    }
  }
}

// the factor's 64 x 64 block of L (lower triangle) and T = L^-T out of the LDS images, write-through
__device__ __forceinline__ void publish_block_shared(const double* lds, int tid, int nbv, double* Lblk, int64_t ldl, double* Tblk) {
  const double* Ls = lds;
  const double* Xs = lds + 2 * CTILE;
  const int c = tid & 63;
  for (int r = tid >> 6; r < CB; r += 4) {
    if (r < nbv && c <= r) st_shared(Lblk + int64_t(r) * ldl + c, Ls[c * CLD + r]);
    st_shared(Tblk + r * CB + c, Xs[r * CLD + c]);
  }
}

template <int FORM, bool SHARED = false, bool STORE = true>
__device__ __forceinline__ void factor_and_publish_t(double* lds, int tid, int nbv, double* __restrict__ Lblk, int64_t ldl,
                                                     double* __restrict__ Tblk, int* __restrict__ info, int64_t col0,
                                                     unsigned long long* stamps = nullptr) {
  double* Ls = lds;
  const double* In = lds + CTILE;
  double* Xs = lds + 2 * CTILE;
  double* Rd = lds + 3 * CTILE;
  int* progress = reinterpret_cast<int*>(Rd + CB);
  double* PL = Rd + CB + 8;
  if (FORM != 0) {
    __syncthreads();
    if (tid < 64) {
      if (stamps && tid == 0) stamps[0] = __builtin_readcyclecounter();
      const int bad = FORM == 2 ? chol64_p16(In, Ls, Rd, PL, tid) : chol64_mfma(In, Ls, Rd, PL, tid);
      if (tid == 0 && bad != 0x7fffffff) atomicMin(info, int(col0 + bad + 1));
      if (stamps && tid == 0) stamps[1] = __builtin_readcyclecounter();
    }
    __syncthreads();
    inv64_mfma(Ls, Rd, Xs, tid);
    if (stamps && tid == 0) stamps[2] = __builtin_readcyclecounter();
  } else {
    if (tid == 0) *progress = 0;
    __syncthreads();
    if (tid < 128) {
      const int bad = wave_factor_lds(Ls, In, Xs, Rd, progress, tid);
      if (tid == 0 && bad != 0x7fffffff) atomicMin(info, int(col0 + bad + 1));
    }
  }
  __syncthreads();
  if (!STORE) return;
  const int c = tid & 63;
  for (int r = tid >> 6; r < CB; r += 4) {
    if (SHARED) {
      if (r < nbv && c <= r) st_shared(Lblk + int64_t(r) * ldl + c, Ls[c * CLD + r]);
      st_shared(Tblk + r * CB + c, Xs[r * CLD + c]);
    } else {
      if (r < nbv && c <= r) Lblk[int64_t(r) * ldl + c] = Ls[c * CLD + r];
      Tblk[r * CB + c] = Xs[r * CLD + c];
    }
  }
}

// first diagonal block of every matrix
template <int FORM>
__global__ __launch_bounds__(256) void k_cholinv_first(CholInvBatch bt, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char ci_smem[];
  double* lds = reinterpret_cast<double*>(ci_smem);
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int nbv = int(min<int64_t>(CB, bt.d[b]));
  double* Xs = lds + CTILE;
  if (tid == 0) info[b] = 0x7fffffff;                    // before the barrier inside factor_and_publish
  const int c = tid & 63;
  for (int r = tid >> 6; r < CB; r += 4) {
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nbv && c < nbv) v = c <= r ? bt.A[b][int64_t(r) * bt.lda[b] + c] : bt.A[b][int64_t(c) * bt.lda[b] + r];
    Xs[r * CLD + c] = v;
  }
  factor_and_publish_t<FORM>(lds, tid, nbv, bt.L[b], bt.ldl[b], bt.T[b], info + b, 0);
}

// step j: trailing update (+ look-ahead factorization of block j + 1) and row j of the inverse
template <int FORM>
__global__ __launch_bounds__(256) void k_cholinv_step(CholInvBatch bt, int j, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char ci_smem[];
  double* lds = reinterpret_cast<double*>(ci_smem);
  double* P = lds;               // tile 0
  double* Q = lds + CTILE;       // tile 1
  double* TT = lds + 2 * CTILE;  // tile 2
  int b = 0;
  while (b + 1 < bt.count && int(blockIdx.x) >= bt.first[b + 1]) ++b;
  const int item = int(blockIdx.x) - bt.first[b];
  const int64_t d = bt.d[b];
  const int nb = int((d + CB - 1) / CB);
  const int r = nb - 1 - j;                       // trailing block rows (may be <= 0)
  const int nU = (r > 0 && !bt.inv_only) ? r * (r + 1) / 2 : 0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t lda = bt.lda[b], ldl = bt.ldl[b];
  double* A = bt.A[b];
  double* L = bt.L[b];
  const double* Tj = bt.T[b] + int64_t(j) * CB * CB;
  const int64_t cj = int64_t(j) * CB;

  if (item < nU) {
    // ---- trailing tile (i, k), j < k <= i ----
    int ii = int((sqrtf(8.0f * float(item) + 1.0f) - 1.0f) * 0.5f);
    while (ii * (ii + 1) / 2 > item) --ii;
    while ((ii + 1) * (ii + 2) / 2 <= item) ++ii;
    const int kk = item - ii * (ii + 1) / 2;
    const int i = j + 1 + ii, k = j + 1 + kk;
    const int64_t ri = int64_t(i) * CB, rk = int64_t(k) * CB;
    const int rows_i = int(min<int64_t>(CB, d - ri)), rows_k = int(min<int64_t>(CB, d - rk));
    load_tile<false>(P, A + ri * lda + cj, lda, rows_i, CB, tid);
    if (k != i) load_tile<false>(Q, A + rk * lda + cj, lda, rows_k, CB, tid);
    load_tile<false>(TT, Tj, CB, CB, CB, tid);
    __syncthreads();
    v4f64 a1[4], a2[4];
    acc_zero(a1);
    tile_mm<false, false>(P, TT, w, lane, a1);            // L_ij = A_ij T_j
    if (k != i) {
      acc_zero(a2);
      tile_mm<false, false>(Q, TT, w, lane, a2);          // L_kj = A_kj T_j
    }
    __syncthreads();
    acc_to_lds(a1, P, w, lane, 1.0);
    if (k != i) acc_to_lds(a2, Q, w, lane, 1.0);
    __syncthreads();
    if (k == j + 1) {                                     // exactly one tile per block row writes L_ij
      const int c = tid & 63;
      for (int rr = tid >> 6; rr < rows_i; rr += 4) L[(ri + rr) * ldl + cj + c] = P[rr * CLD + c];
    }
    v4f64 u[4];
    acc_zero(u);
    tile_mm<false, true>(P, k != i ? Q : P, w, lane, u);  // L_ij L_kj'
    const bool critical = (i == k) && (k == j + 1);
    if (!critical) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
          if (rr < rows_i && cc < rows_k) {
            double* p = A + (ri + rr) * lda + rk + cc;
            *p -= u[t][rg];
          }
        }
      return;
    }
    // ---- look-ahead: this workgroup owns block (j+1, j+1): update it in LDS and factor it right away ----
    __syncthreads();                                      // all waves are done reading P
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        double v = (rr == cc) ? 1.0 : 0.0;
        if (rr < rows_i && cc < rows_i) {
          // the Schur complement is symmetric; read the authoritative lower triangle
          const double base = cc <= rr ? A[(ri + rr) * lda + ri + cc] : A[(ri + cc) * lda + ri + rr];
          v = base - u[t][rg];
        }
        Q[rr * CLD + cc] = v;                             // tile 1 = Xs of the wave factorization
      }
    factor_and_publish_t<FORM>(lds, tid, rows_i, L + ri * ldl + ri, ldl, bt.T[b] + int64_t(j + 1) * CB * CB, info + b, ri);
    return;
  }

  // ---- row j of X = L^-1 ----
  double* X = bt.X[b];
  if (!X || j >= nb) return;
  const int64_t ldx = bt.ldx[b];
  const int k = item - nU;                                // 0 .. j
  const int rows_j = int(min<int64_t>(CB, d - cj));
  if (k == j) {                                           // X_jj = T_j'
    load_tile<true>(P, Tj, CB, CB, CB, tid);
    __syncthreads();
    const int c = tid & 63;
    for (int rr = tid >> 6; rr < rows_j; rr += 4)
      if (c < rows_j) X[(cj + rr) * ldx + cj + c] = P[rr * CLD + c];
    return;
  }
  v4f64 s[4];
  acc_zero(s);
  for (int t = k; t < j; ++t) {
    load_tile<false>(P, L + cj * ldl + int64_t(t) * CB, ldl, rows_j, CB, tid);                 // L_jt
    load_tile<false>(Q, X + int64_t(t) * CB * ldx + int64_t(k) * CB, ldx, CB, CB, tid);         // X_tk
    __syncthreads();
    tile_mm<false, false>(P, Q, w, lane, s);
    __syncthreads();
  }
  acc_to_lds(s, P, w, lane, -1.0);
  load_tile<false>(TT, Tj, CB, CB, CB, tid);
  __syncthreads();
  v4f64 o[4];
  acc_zero(o);
  tile_mm<true, false>(TT, P, w, lane, o);                // T_j' (-S)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
      if (rr < rows_j) X[(cj + rr) * ldx + int64_t(k) * CB + cc] = o[t][rg];
    }
}

__global__ void k_fill_int(int* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

struct MultiGemm;
__global__ void k_gemm_f64_multi(MultiGemm g);
constexpr size_t MG_LDS_FWD = size_t(4) * CTILE * sizeof(double);

// ---------------------------------------------------------------------------
// ONE persistent launch for the whole factor + inverse (round 5).
//
// The launch-per-link form above pays, per 64-column link, a kernel launch, a reload of T_j and of the two tiles the
// diagonal recurrence needs, and the trailing tiles' latency in FRONT of the next factorization (34 us per link, of which
// the 64 x 64 factorization is ~20).  Here the work items of ALL links are handed out through a ticket counter to a
// fixed set of workgroups, and device-side progress counters order them:
//
//   * chain workgroup b (ticket b < count) stays with matrix b for the whole launch: factor block j, invert it (T_j
//     stays in its LDS), form L_{j+1,j} = A_{j+1,j} T_j and S = A_{j+1,j+1} - L_{j+1,j} L_{j+1,j}' from tiles that the
//     helpers brought up to date while block j was being factored, factor S, ...  -- its link is two tile products and
//     the factorization, nothing else;
//   * helper items (tickets in link-major order): trailing tile (i, k) of link j (all but the chain's own tile);
//     XS(i, k): the partial sums  S_ik = sum_{t=k}^{i-1} L_it X_tk  of row i of X = L^-1, which need rows < i and the
//     panel L_{i,i-1} but NOT T_i -- they run while block i is being factored; XT(j, k): X_jk = -T_j' S_jk (X_jj = T_j'),
//     one tile product once T_j is published.
//
// Deadlock freedom without co-residency: a helper item only waits for items with LOWER tickets (tickets are taken in
// start order, so those have started) and for chain progress that the chain reaches without waiting for that item's
// link; the chain only waits for helpers of link j - 1, which wait for nothing the chain has not published already.
// Spin waits poll one word with agent-scope atomics and sleep in between; a watchdog (~2 s) turns a protocol bug into a
// reported failure (info = 0x7ffffff0) instead of a hung device.  The sync block is zero between launches: the last
// workgroup to finish clears it.
// ---------------------------------------------------------------------------
constexpr int CH_MAXNB = 64;        // block columns per matrix the chain kernel serves (d <= 4096)
constexpr int CH_WATCHDOG = 1 << 22;
constexpr int CH_TIMEOUT_INFO = 0x7ffffff0;

struct ChainSync {
  unsigned ticket;                  // next work item
  unsigned done;                    // workgroups that have left the item loop
  unsigned abort;                   // a wait ran into the watchdog
  unsigned pad_;
  unsigned F[CMAXB];                // diagonal blocks factored and published: T_0 .. T_{F-1}, L_jj
  unsigned G[CMAXB];                // panel blocks L_{g,g-1} written for all g <= G
  unsigned U[CMAXB][CH_MAXNB];      // finished trailing-tile items of link j (the chain's own tile is not counted)
  unsigned XS[CMAXB][CH_MAXNB];     // finished partial-sum items of row i of X
  unsigned XR[CMAXB][CH_MAXNB];     // finished items of row j of X
};

struct ChainPlan {
  int link_first[CH_MAXNB + 1];     // helper items of the links before j, all matrices; [nbmax] = all helper items
  int nbmax;
  int total;                        // count chain items + helper items
  int poll_sleep;                   // s_sleep(4) periods between two polls of a progress word (CCZ_CHAIN_SLEEP)
};

// helper items of matrix (nb block columns, with / without X) at link j: trailing tiles without the chain's, XT row j, XS row j + 1
__host__ __device__ __forceinline__ void chain_counts(int nb, bool with_x, int j, int& hU, int& nXT, int& nXS) {
  const int r = nb - 1 - j;
  const int nU = r > 0 ? r * (r + 1) / 2 : 0;
  hU = nU > 0 ? nU - 1 : 0;
  nXT = (with_x && j < nb) ? j + 1 : 0;
  nXS = (with_x && j + 1 < nb) ? j + 1 : 0;
}

// Thread 0 polls up to three progress words (agent scope, relaxed) until each has reached its target; the workgroup
// follows through a barrier.  No acquire fence: every shared word is read with ld_shared (past the L1).  flag: one LDS
// int.  Returns false after the watchdog / an abort raised elsewhere (the caller goes on with whatever is in memory: the
// launch must terminate, its result is reported as failed).
struct ChainWaitList {
  unsigned* p[3];
  unsigned target[3];
  int n = 0;
  int sleep = 1;
  bool first_is_step = false;     // p[0] counts chain steps (F / G): being two or more behind means a link away
  __device__ __forceinline__ void add(unsigned* q, unsigned t) { p[n] = q; target[n] = t; ++n; }
};

__device__ __forceinline__ bool chain_wait(const ChainWaitList& wl, ChainSync* sy, int* flag, int tid) {
  if (wl.n == 0) return true;
  if (tid == 0) {
    int good = 1, spins = 0;
    for (int q = 0; q < wl.n && good; ++q) {
      const gu32_ptr word = (gu32_ptr)(reinterpret_cast<uintptr_t>(wl.p[q]));
      for (;;) {
        const int behind = int(wl.target[q] - __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (behind <= 0) break;
        // a target several steps away (a link that has not started) is polled rarely: idle pollers cost the memory system
        // of the workgroups that do the work (MI355X_MICROARCH.md: "255 pollers cut chip bandwidth 37-71 %")
        const int naps = wl.sleep * (q == 0 && wl.first_is_step && behind >= 2 ? 16 : 1);
        for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 63) == 0 && __hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { good = 0; break; }
        if (spins > CH_WATCHDOG) {
          __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          good = 0;
          break;
        }
      }
    }
    *flag = good;
  }
  __syncthreads();
  const bool ok = *flag != 0;
  __syncthreads();                     // the next wait may overwrite the flag
  return ok;
}

// hand-over: every wave drains its write-through stores, barrier, one lane moves the progress word
__device__ __forceinline__ void chain_publish_add(unsigned* p, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_publish_set(unsigned* p, unsigned v, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The chain workgroup of matrix b.  lds4: four tiles -- two factor images used alternately (while block j + 1 is being
// factored into one of them on wave 0, waves 1-3 still read block j's image from the other), the input / Schur tile, the
// inverse tile -- then Rd, flags and the panel image.  Per link (block j factored -> block j + 1 factored):
//   all waves:  phase A of block j's inverse (the four 16 x 16 diagonal inverses)
//               tiles (j+1, j), (j+1, j+1) requested (the helpers of link j - 1 must be done)
//               L_{j+1,j} = A_{j+1,j} L_jj^-T  by the register-resident strip solve (needs phase A only), stored write-through
//               S = A_{j+1,j+1} - L_{j+1,j} L_{j+1,j}'  -> input tile
//   wave 0:     factor S                            || waves 1-3: phase B of block j's inverse, T_j and L_jj stored
//                                                      write-through, drained, F[b] = j + 1  (block j is public)
// so the publication of a block -- off-diagonal inverse blocks, 64 KB of stores, their drain -- costs the chain nothing;
// the helpers of link j start while block j + 1 is being factored and have until its end.
// dbg (CCZ_CHAIN_DEBUG=1, matrix 0 only): shader-clock stamps per link -- [0] link start, [1] phase A done, [2] helpers
// awaited + tiles in LDS, [3] L_ij formed, [4] Schur block ready (fork), [5] factorization done, [6] block j published
// (stamped by wave 1), [7] joined.
// (the thread index passes through an empty asm at the head of this path and of the helper items: without it the compiler
// forms every path's index constants at the kernel's entry, keeps them alive across the whole kernel, and spills 65 to 126
// registers that are reloaded from scratch memory inside the chain)
template <int FORM>
__device__ __forceinline__ void chain_matrix(int64_t d, int64_t lda, int64_t ldl, double* A, double* L, double* T, bool with_x, int b,
                                                       ChainSync* sy, int* __restrict__ info, double* lds4, int* flag, int tid,
                                                       unsigned long long* dbg) {
  asm volatile("" : "+v"(tid));        // index arithmetic of this path is formed HERE, not hoisted to the kernel's entry (see above)
  const int nb = int((d + CB - 1) / CB);
  double* IN = lds4 + 2 * CTILE;            // input of the factorization / staged A_ij
  double* XS = lds4 + 3 * CTILE;            // L^-T of the current block
  double* Rd = lds4 + 4 * CTILE;
  int* pubcnt = reinterpret_cast<int*>(Rd + CB) + 4;     // arrivals of waves 1-3 at the end of a publication
  double* PL = Rd + CB + 8;
  const int lane = tid & 63, w = tid >> 6;
  if (b != 0) dbg = nullptr;
  bool ok = true;
  int cur = 0;
  // ---- block 0 ----
  {
    const int nbv = int(min<int64_t>(CB, d));
    if (tid == 0) { info[b] = 0x7fffffff; *pubcnt = 0; }
    if (dbg && tid == 0) dbg[0] = __builtin_readcyclecounter();
    const int c = tid & 63;
    for (int r = tid >> 6; r < CB; r += 4) {
      double v = (r == c) ? 1.0 : 0.0;
      if (r < nbv && c < nbv) v = c <= r ? A[int64_t(r) * lda + c] : A[int64_t(c) * lda + r];   // written before the launch
      IN[r * CLD + c] = v;
    }
    __syncthreads();
    if (tid < 64) {
      const int bad = FORM == 2 ? chol64_p16(IN, lds4 + cur * CTILE, Rd, PL, tid) : chol64_mfma(IN, lds4 + cur * CTILE, Rd, PL, tid);
      if (tid == 0 && bad != 0x7fffffff) atomicMin(info + b, int(bad + 1));
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[5] = __builtin_readcyclecounter();
  }
  for (int j = 0; j < nb; ++j) {
    unsigned long long* dj = dbg ? dbg + 8 * (j + 1) : nullptr;
    const bool more = j + 1 < nb;
    const int64_t cj = int64_t(j) * CB;
    const int rows_j = int(min<int64_t>(CB, d - cj));
    double* Lcur = lds4 + cur * CTILE;            // factor images in tiles 0 / 1, alternating
    double* Lnxt = lds4 + (cur ^ 1) * CTILE;
    if (dj && tid == 0) dj[0] = __builtin_readcyclecounter();
    inv64_phaseA(Lcur, Rd, XS, tid);
    __syncthreads();
    // L_jj and the four 16 x 16 diagonal blocks of T_j = L_jj^-T are all the trailing tiles of link j need (they form their
    // panels by the same strip solve as below): stored NOW, announced with G below -- a whole factorization earlier than T_j
    auto early_stores = [&]() {
      double* Lb = L + cj * ldl + cj;
      double* Tb = T + int64_t(j) * CB * CB;
      const bool al = ((reinterpret_cast<uintptr_t>(Lb) | uintptr_t(ldl * 8)) & 15) == 0;
      for (int e = tid; e < CB * 32; e += 256) {          // pairs of columns (c2, c2 + 1) of row r
        const int r = e >> 5, c2 = (e & 31) * 2;
        if (r < rows_j && c2 <= r) {                      // lower triangle of L_jj: nothing is written above the diagonal
          const double v0 = Lcur[c2 * CLD + r];
          if (c2 + 1 <= r) {
            const double v1 = Lcur[(c2 + 1) * CLD + r];
            if (al) st_shared2(Lb + int64_t(r) * ldl + c2, v0, v1);
            else { st_shared(Lb + int64_t(r) * ldl + c2, v0); st_shared(Lb + int64_t(r) * ldl + c2 + 1, v1); }
          } else {
            st_shared(Lb + int64_t(r) * ldl + c2, v0);
          }
        }
        if ((r >> 4) == (c2 >> 4)) st_shared2(Tb + r * CB + c2, XS[r * CLD + c2], XS[r * CLD + c2 + 1]);
      }
    };
    if (dj && tid == 0) dj[1] = __builtin_readcyclecounter();
    const int i = j + 1;
    const int64_t ri = int64_t(i) * CB;
    const int rows_i = more ? int(min<int64_t>(CB, d - ri)) : 0;
    double aii[4][4];                                     // A_ii in the accumulator layout of the Schur product
    if (more) {
      if (j >= 1) {                                        // tiles (j+1, j) and (j+1, j+1) carry the updates of the links < j
        int hU, nXT, nXS;
        chain_counts(nb, with_x, j - 1, hU, nXT, nXS);
        ChainWaitList wl;
        if (hU > 0) wl.add(&sy->U[b][j - 1], unsigned(hU));
        ok = chain_wait(wl, sy, flag, tid) && ok;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
          double v = (rr == cc) ? 1.0 : 0.0;
          if (rr < rows_i && cc < rows_i) v = ld_shared(cc <= rr ? A + (ri + rr) * lda + ri + cc : A + (ri + cc) * lda + ri + rr);   // lower triangle authoritative
          aii[t][rg] = v;
        }
      TileRegs rpan;
      fetch_tile_shared(rpan, A + ri * lda + cj, lda, rows_i, CB, tid);        // A_ij, row-major
      store_tile<false>(IN, rpan, tid);
      __syncthreads();
      if (dj && tid == 0) dj[2] = __builtin_readcyclecounter();
      solve_strip_LT(IN, Lcur, XS, Lnxt, w, lane);          // rows 16 w .. of L_ij = A_ij L_jj^-T -> the OTHER factor image (scratch until the factorization)
      __syncthreads();
      // vmcnt counts loads and stores in one in-order queue: a store issued BEFORE a load is paid for, round trip and all, by
      // the wait for that load.  So everything this link stores goes out here, behind its last load, and lands under the
      // Schur product; the drain in front of the G publication finds it done.
      early_stores();
      // (an 8- or 16-byte write-through store is one fabric write per lane: ~400 cycles per wave instruction, measured -- so
      // the chain stores as little as it can: L_ij itself goes out with the trailing tile (j+2, j+1), which forms it anyway
      // as its second panel block; only the last link of a matrix has no such tile)
      if (i + 1 >= nb) store_rows_shared(L + ri * ldl + cj, ldl, Lnxt, rows_i, tid, 256);
      if (dj && tid == 0) dj[3] = __builtin_readcyclecounter();
      v4f64 u[4];
      acc_zero(u);
      tile_mm<false, true>(Lnxt, Lnxt, w, lane, u);         // L_ij L_ij'
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) IN[(16 * w + (lane >> 4) + 4 * rg) * CLD + 16 * t + (lane & 15)] = aii[t][rg] - u[t][rg];
      chain_publish_set(&sy->G[b], unsigned(i), tid);       // L_jj, diag(T_j) and L_ij have had the products to land: the trailing tiles
                                                            // of link j and the sums of row i of X may start.  (barrier inside)
    }
    if (!more) early_stores();
    if (dj && tid == 0) dj[4] = __builtin_readcyclecounter();
    // ---- fork: wave 0 factors block j + 1, waves 1-3 finish and publish block j ----
    if (w == 0) {
      if (more) {
        const int bad = FORM == 2 ? chol64_p16(IN, Lnxt, Rd, PL, lane) : chol64_mfma(IN, Lnxt, Rd, PL, lane);
        if (lane == 0 && bad != 0x7fffffff) atomicMin(info + b, int(ri + bad + 1));
        if (dj && lane == 0) dj[5] = __builtin_readcyclecounter();
      }
    } else {
      inv64_phaseB_col(Lcur, XS, lane, w - 1);              // column blocks 0, 1, 2 of L^-1 (column 3 has no block below its diagonal)
      // three-wave rendezvous: the stores below read blocks the OTHER two waves have just written
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) {
        __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(pubcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int t3 = tid - 64;                              // 0 .. 191
      for (int e = t3; e < CB * 32; e += 192) {
        const int r = e >> 5, c2 = (e & 31) * 2;
        double* q = T + int64_t(j) * CB * CB + r * CB + c2;
        if ((r >> 4) < (c2 >> 4)) st_shared2(q, XS[r * CLD + c2], XS[r * CLD + c2 + 1]);   // the blocks phase B has formed
        else if ((r >> 4) > (c2 >> 4)) st_shared2(q, 0.0, 0.0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) {
        const int arrived = __hip_atomic_fetch_add(pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (arrived == 5) {                                 // the last of the three waves: every store of the block has landed
          __hip_atomic_store(&sy->F[b], unsigned(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(pubcnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (dj) dj[6] = __builtin_readcyclecounter();
        }
      }
    }
    __syncthreads();                                        // join
    if (dj && tid == 0) dj[7] = __builtin_readcyclecounter();
    cur ^= 1;
  }
  if (!ok && tid == 0) atomicMin(info + b, CH_TIMEOUT_INFO);
}

// lds4: FOUR tiles (extra tile first), then Rd / flags / panel image
// one helper item (decoded by the caller): q < hU trailing tile, < hU + nXT row j of X from its sums, else the sums of row j + 1
__device__ __forceinline__ void chain_helper(int j, int q, int hU, int nXT, int nb, bool with_x, int64_t d, int64_t lda, int64_t ldl,
                                                       int64_t ldx, double* A, double* L, double* X, const double* Tj, int b, int poll_sleep,
                                                       ChainSync* sy, int* __restrict__ info, double* lds4, int* flag, int tid) {
  asm volatile("" : "+v"(tid));        // as in chain_matrix
  const int64_t cj = int64_t(j) * CB;
  const int lane = tid & 63, w = tid >> 6;
  double* P = lds4 + CTILE;
  double* Q = lds4 + 2 * CTILE;
  double* TT = lds4 + 3 * CTILE;
  bool ok = true;

  if (q < hU) {
    // ---- trailing tile (i, k) of link j, j < k <= i, (i, k) != (j+1, j+1) ----
    // Needs L_jj and the diagonal blocks of T_j only (announced with G, a factorization before the full T_j): both panel
    // blocks L_ij = A_ij L_jj^-T, L_kj = A_kj L_jj^-T come from the register-resident strip solve.
    ChainWaitList wl;
    wl.sleep = poll_sleep;
    wl.first_is_step = true;
    wl.add(&sy->G[b], unsigned(j + 1));
    if (j >= 1) {
      int pU, a_, b_;
      chain_counts(nb, with_x, j - 1, pU, a_, b_);
      if (pU > 0) wl.add(&sy->U[b][j - 1], unsigned(pU));
    }
    ok = chain_wait(wl, sy, flag, tid) && ok;
    const int item = q + 1;
    int ii = int((sqrtf(8.0f * float(item) + 1.0f) - 1.0f) * 0.5f);
    while (ii * (ii + 1) / 2 > item) --ii;
    while ((ii + 1) * (ii + 2) / 2 <= item) ++ii;
    const int kk = item - ii * (ii + 1) / 2;
    const int i = j + 1 + ii, k = j + 1 + kk;
    const int64_t ri = int64_t(i) * CB, rk = int64_t(k) * CB;
    const int rows_i = int(min<int64_t>(CB, d - ri)), rows_k = int(min<int64_t>(CB, d - rk));
    double* LsT = lds4;                                   // L_jj as the factor image Ls[col][row]
    double* XsT = TT;                                     // T_j (its diagonal blocks are what the solve reads)
    // the tile being updated is requested with the operands: its loads fly under the products
    double cur[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        cur[t][rg] = (rr < rows_i && cc < rows_k) ? ld_shared(A + (ri + rr) * lda + rk + cc) : 0.0;
      }
    load_tile_shared<false>(P, A + ri * lda + cj, lda, rows_i, CB, tid);
    if (k != i) load_tile_shared<false>(Q, A + rk * lda + cj, lda, rows_k, CB, tid);
    {
      // L_jj: lower triangle from memory, identity beyond the matrix (a partial last block never is block j of a link)
      TileRegs tl;
      const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        const int r = r0 + 4 * qq;
        tl.v[qq] = c <= r ? ld_shared(L + (cj + r) * ldl + cj + c) : 0.0;
      }
      store_tile<true>(LsT, tl, tid);                     // Ls[c][r] = L[r][c]
    }
    load_tile_shared<false>(XsT, Tj, CB, CB, CB, tid);
    __syncthreads();
    solve_strip_LT(P, LsT, XsT, P, w, lane);              // in place: a strip's rows belong to one wave
    if (k != i) solve_strip_LT(Q, LsT, XsT, Q, w, lane);
    __syncthreads();
    if (k == j + 1) store_rows_shared(L + ri * ldl + cj, ldl, P, rows_i, tid, 256);   // exactly one tile per block row writes L_ij
    if (k == j + 1 && i == j + 2) store_rows_shared(L + rk * ldl + cj, ldl, Q, rows_k, tid, 256);   // ... and this one the chain's own L_{j+1,j}
    v4f64 u[4];
    acc_zero(u);
    tile_mm<false, true>(P, k != i ? Q : P, w, lane, u);  // L_ij L_kj'
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        if (rr < rows_i && cc < rows_k) st_shared(A + (ri + rr) * lda + rk + cc, cur[t][rg] - u[t][rg]);
      }
    chain_publish_add(&sy->U[b][j], tid);
  } else if (q < hU + nXT) {
    // ---- XT(j, k): X_jk = -T_j' S_jk, X_jj = T_j' ----
    const int k = q - hU;
    const int rows_j = int(min<int64_t>(CB, d - cj));
    ChainWaitList wl;
    wl.sleep = poll_sleep;
    wl.first_is_step = true;
    wl.add(&sy->F[b], unsigned(j + 1));
    if (k != j) wl.add(&sy->XS[b][j], unsigned(j));
    ok = chain_wait(wl, sy, flag, tid) && ok;
    if (k == j) {
      load_tile_shared<true>(P, Tj, CB, CB, CB, tid);
      __syncthreads();
      const int c = tid & 63;
      for (int rr = tid >> 6; rr < rows_j; rr += 4)
        if (c < rows_j) st_shared(X + (cj + rr) * ldx + cj + c, P[rr * CLD + c]);
    } else {
      load_tile_shared<false>(P, X + cj * ldx + int64_t(k) * CB, ldx, rows_j, CB, tid);     // S_jk
      load_tile_shared<false>(TT, Tj, CB, CB, CB, tid);
      __syncthreads();
      v4f64 o[4];
      acc_zero(o);
      tile_mm<true, false>(TT, P, w, lane, o);            // T_j' S
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
          if (rr < rows_j) st_shared(X + (cj + rr) * ldx + int64_t(k) * CB + cc, -o[t][rg]);
        }
    }
    chain_publish_add(&sy->XR[b][j], tid);
  } else {
    // ---- XS(i, k), i = j + 1: S_ik = sum_{t=k}^{i-1} L_it X_tk, stored where X_ik will be ----
    const int k = q - hU - nXT;
    const int i = j + 1;
    const int64_t ri = int64_t(i) * CB;
    const int rows_i = int(min<int64_t>(CB, d - ri));
    ChainWaitList wl;
    wl.sleep = poll_sleep;
    wl.first_is_step = true;
    wl.add(&sy->G[b], unsigned(i));           // the chain has passed link j (every L_it, t < j, is final) and stored L_{i,j} if it does so itself
    wl.add(&sy->XR[b][j], unsigned(j + 1));   // rows <= j of X
    if (hU > 0) wl.add(&sy->U[b][j], unsigned(hU));   // L_{i,j} written by the trailing tile (j+2, j+1) of this link
    ok = chain_wait(wl, sy, flag, tid) && ok;
    v4f64 sacc[4];
    acc_zero(sacc);
    TileRegs ra, rb;
    auto fetch = [&](int t) {
      fetch_tile_shared(ra, L + ri * ldl + int64_t(t) * CB, ldl, rows_i, CB, tid);                      // L_it
      fetch_tile_shared(rb, X + int64_t(t) * CB * ldx + int64_t(k) * CB, ldx, CB, CB, tid);             // X_tk
    };
    auto stash = [&](int buf) {
      store_tile<false>(lds4 + buf * 2 * CTILE, ra, tid);
      store_tile<false>(lds4 + buf * 2 * CTILE + CTILE, rb, tid);
    };
    fetch(k);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int t = k; t < i; ++t) {
      const bool more = t + 1 < i;
      if (more) fetch(t + 1);
      tile_mm<false, false>(lds4 + buf * 2 * CTILE, lds4 + buf * 2 * CTILE + CTILE, w, lane, sacc);
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        if (rr < rows_i) st_shared(X + (ri + rr) * ldx + int64_t(k) * CB + cc, sacc[t][rg]);
      }
    chain_publish_add(&sy->XS[b][i], tid);
  }
  if (!ok && tid == 0) atomicMin(info + b, CH_TIMEOUT_INFO);
}

template <int FORM>
__global__ __launch_bounds__(256) void k_cholinv_chain(CholInvBatch bt, ChainPlan plan, ChainSync* __restrict__ sy, int* __restrict__ info,
                                                       unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char ci_smem[];
  double* lds4 = reinterpret_cast<double*>(ci_smem);          // extra tile, then the step kernels' layout
  double* lds = lds4 + CTILE;
  int* flags = reinterpret_cast<int*>(lds + 3 * CTILE + CB);   // [0] progress word of the shift-register form, [2] wait flag, [3] ticket
  const int tid = threadIdx.x;
  for (;;) {
    __syncthreads();                                           // everyone is done with the previous item (LDS, flags)
    if (tid == 0) flags[3] = int(__hip_atomic_fetch_add(&sy->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const int ticket = flags[3];
    if (ticket >= plan.total) break;
    if (ticket < bt.count) {
      const int b = ticket;
      chain_matrix<FORM>(bt.d[b], bt.lda[b], bt.ldl[b], bt.A[b], bt.L[b], bt.T[b], bt.X[b] != nullptr, b, sy, info, lds4, flags + 2, tid, dbg);
    } else {
      int q = ticket - bt.count;
      int j = 0;
      while (j + 1 < plan.nbmax && q >= plan.link_first[j + 1]) ++j;
      q -= plan.link_first[j];
      int b = 0, hU = 0, nXT = 0, nXS = 0, nb = 0;
      for (;; ++b) {
        nb = int((bt.d[b] + CB - 1) / CB);
        chain_counts(nb, bt.X[b] != nullptr, j, hU, nXT, nXS);
        const int tot = hU + nXT + nXS;
        if (q < tot || b + 1 >= bt.count) break;
        q -= tot;
      }
      chain_helper(j, q, hU, nXT, nb, bt.X[b] != nullptr, bt.d[b], bt.lda[b], bt.ldl[b], bt.ldx[b], bt.A[b], bt.L[b], bt.X[b],
                   bt.T[b] + int64_t(j) * CB * CB, b, plan.poll_sleep, sy, info, lds4, flags + 2, tid);
    }
  }
  // the last workgroup to leave clears the sync block for the next launch (nobody reads or writes it any more)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) flags[3] = __hip_atomic_fetch_add(&sy->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (flags[3]) {
    unsigned* wds = reinterpret_cast<unsigned*>(sy);
    for (int e = tid; e < int(sizeof(ChainSync) / sizeof(unsigned)); e += 256)
      __hip_atomic_store(wds + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int FORM>
static void cholinv_attr_form() {
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_first<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_step<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
}

static void cholinv_attr_once() {
  static thread_local int done_for_device = -1;
  int dev = -1;
  CCZ_HIP(hipGetDevice(&dev));
  if (done_for_device == dev) return;
  cholinv_attr_form<0>();
  cholinv_attr_form<1>();
  cholinv_attr_form<2>();
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHAIN_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHAIN_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_multi), hipFuncAttributeMaxDynamicSharedMemorySize, int(MG_LDS_FWD)));
  done_for_device = dev;
}

// CCZ_CHOLINV_MFMA: 0 shift-register (two-wave) form of the 64 x 64 factorization, 1 MFMA form with 4-column panels
// (rounds 2-4: 30.4k shader cycles per block inside the chain kernel), 2 (default) MFMA form with 16-column panels (24.3k).
// Either way ONE wave issues ~3000 instructions per block: a dependent fp64 operation costs 4-8 cycles, not the 32 that
// rounds 2-4 designed around (tools/probes/lat_probe.hip: that probe timed a loop branch) -- the block is issue-bound.
static int cholinv_form() {
  static const int form = [] { const char* e = getenv("CCZ_CHOLINV_MFMA"); const int v = e ? atoi(e) : 2; return v < 0 || v > 2 ? 2 : v; }();
  return form;
}

// The sync block of the chain kernel belongs to the STREAM the launch goes into: launches on one stream are ordered, so
// they can share a block (it is zero again when a launch ends); launches on different streams of one handle (the
// factorization's look-ahead stream, a caller's adopted streams) get blocks of their own.
static ChainSync* chain_sync_for(ccz_ctx* c) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  for (auto& e : im->chain_sync)
    if (e.first == static_cast<void*>(st)) return static_cast<ChainSync*>(e.second);
  if (im->chain_sync.size() >= 32) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (cs != hipStreamCaptureStatusNone) return nullptr;      // no allocation inside a capture: the launch-per-link form runs
  void* p = nullptr;
  if (hipMalloc(&p, sizeof(ChainSync)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemsetAsync(p, 0, sizeof(ChainSync), st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  im->chain_sync.emplace_back(static_cast<void*>(st), p);
  return static_cast<ChainSync*>(p);
}

// CCZ_CHOLINV_CHAIN=0: the launch-per-link form (rounds 2-4).  CCZ_CHAIN_WGS: workgroups of the persistent launch
// (default 128; Impl::chain_cap overrides it for callers that share the chip with throughput work on another stream).
static bool chain_launch(ccz_ctx* c, const CholInvBatch& bt, int nbmax, int* info_dev) {
  static const int on = [] { const char* e = getenv("CCZ_CHOLINV_CHAIN"); return e ? atoi(e) : 1; }();
  static const int wgs_env = [] { const char* e = getenv("CCZ_CHAIN_WGS"); return e ? atoi(e) : 128; }();
  const int form = cholinv_form();
  if (!on || form == 0 || bt.inv_only || nbmax > CH_MAXNB) return false;
  ChainPlan plan{};
  plan.nbmax = nbmax;
  int total = 0;
  for (int j = 0; j < nbmax; ++j) {
    plan.link_first[j] = total;
    for (int b = 0; b < bt.count; ++b) {
      int hU, nXT, nXS;
      chain_counts(int((bt.d[b] + CB - 1) / CB), bt.X[b] != nullptr, j, hU, nXT, nXS);
      total += hU + nXT + nXS;
    }
  }
  plan.link_first[nbmax] = total;
  plan.total = total + bt.count;
  static const int sleep_env = [] { const char* e = getenv("CCZ_CHAIN_SLEEP"); return e ? std::max(1, atoi(e)) : 1; }();
  plan.poll_sleep = sleep_env;
  ChainSync* sy = chain_sync_for(c);
  if (!sy) return false;
  Impl* im = impl(c);
  const int cap = im->chain_cap > 0 ? im->chain_cap : std::max(wgs_env, 2);
  // at least one helper workgroup next to the chain workgroups (they never leave their matrix)
  const int grid = std::min(plan.total, std::max(cap, bt.count + 1));
  // CCZ_CHAIN_DEBUG=1: shader-clock stamps of matrix 0's chain workgroup, printed per launch (synchronises: measurement only)
  static const int debug = [] { const char* e = getenv("CCZ_CHAIN_DEBUG"); return e ? atoi(e) : 0; }();
  unsigned long long* dbg = nullptr;
  if (debug) {
    if (!im->chain_dbg) CCZ_HIP(hipMalloc(&im->chain_dbg, 8 * (CH_MAXNB + 1) * sizeof(unsigned long long)));
    dbg = static_cast<unsigned long long*>(im->chain_dbg);
    CCZ_HIP(hipMemsetAsync(dbg, 0, 8 * (CH_MAXNB + 1) * sizeof(unsigned long long), stream(c)));
  }
  if (form == 2) hipLaunchKernelGGL(k_cholinv_chain<2>, dim3(grid), dim3(256), CHAIN_LDS, stream(c), bt, plan, sy, info_dev, dbg);
  else hipLaunchKernelGGL(k_cholinv_chain<1>, dim3(grid), dim3(256), CHAIN_LDS, stream(c), bt, plan, sy, info_dev, dbg);
  CCZ_LAUNCH_CHECK();
  if (debug) {
    std::vector<unsigned long long> hst(size_t(8) * (CH_MAXNB + 1));
    CCZ_HIP(hipStreamSynchronize(stream(c)));
    CCZ_HIP(hipMemcpy(hst.data(), dbg, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const int nb0 = int((bt.d[0] + CB - 1) / CB);
    fprintf(stderr, "[ccz chain] grid %d, %d items, matrix 0: %d links; cycles since the launch's first stamp\n"
            "  link:   start   phaseA    tiles     L_ij     fork |   factored published   joined\n", grid, plan.total, nb0);
    const unsigned long long t0 = hst[0];
    for (int j = 0; j < nb0; ++j) {
      const unsigned long long* r = hst.data() + 8 * (j + 1);
      fprintf(stderr, "  %4d: %7lld %8lld %8lld %8lld %8lld | %8lld %8lld %8lld\n", j, (long long)(r[0] - t0), (long long)(r[1] - t0),
              (long long)(r[2] - t0), (long long)(r[3] - t0), (long long)(r[4] - t0), (long long)(r[5] - t0), (long long)(r[6] - t0),
              (long long)(r[7] - t0));
    }
  }
  return true;
}

// Factor `count` (<= 8) SPD matrices and (X != null) invert the factors, all in d_max / 64 + 1 launches on the
// handle's stream, no host synchronisation.  A[b] (lower triangle authoritative, ld lda) is destroyed; L[b] receives
// the lower factor (strictly-upper part untouched); X[b] the lower-triangular L^-1 (strictly-upper part untouched).
// T[b]: scratch of ceil(d / 64) * 4096 doubles.  info_dev[b]: 0x7fffffff, or 1 + index of the first bad pivot.
void cholinv_batched(ccz_ctx* c, int count, double* const* A, const int64_t* lda, const int64_t* d, double* const* L,
                     const int64_t* ldl, double* const* X, const int64_t* ldx, double* const* T, int* info_dev) {
  if (count < 1 || count > CMAXB) fail(CCZ_EINVAL, "cholinv_batched: 1..8 matrices per call");
  cholinv_attr_once();
  hipStream_t st = stream(c);
  CholInvBatch bt{};
  bt.count = count;
  int nbmax = 0;
  for (int b = 0; b < count; ++b) {
    if (d[b] < 1) fail(CCZ_EINVAL, "cholinv_batched: empty matrix");
    bt.A[b] = A[b]; bt.L[b] = L[b]; bt.X[b] = X ? X[b] : nullptr; bt.T[b] = T[b];
    bt.lda[b] = lda[b]; bt.ldl[b] = ldl[b]; bt.ldx[b] = X ? ldx[b] : 0; bt.d[b] = d[b];
    nbmax = std::max(nbmax, int((d[b] + CB - 1) / CB));
  }
  const int form = cholinv_form();
  if (chain_launch(c, bt, nbmax, info_dev)) return;        // ONE persistent launch (below); false: not applicable / switched off
  if (form == 2) hipLaunchKernelGGL(k_cholinv_first<2>, dim3(count), dim3(256), CHOLINV_LDS, st, bt, info_dev);
  else if (form == 1) hipLaunchKernelGGL(k_cholinv_first<1>, dim3(count), dim3(256), CHOLINV_LDS, st, bt, info_dev);
  else hipLaunchKernelGGL(k_cholinv_first<0>, dim3(count), dim3(256), CHOLINV_LDS, st, bt, info_dev);
  for (int j = 0; j < nbmax; ++j) {
    int total = 0;
    for (int b = 0; b < count; ++b) {
      bt.first[b] = total;
      const int nb = int((d[b] + CB - 1) / CB);
      const int r = nb - 1 - j;
      const int nU = (r > 0 && !bt.inv_only) ? r * (r + 1) / 2 : 0;
      const int nV = (bt.X[b] && j < nb) ? j + 1 : 0;
      total += nU + nV;
    }
    bt.first[count] = total;
    if (total == 0) continue;
    if (form == 2) hipLaunchKernelGGL(k_cholinv_step<2>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, info_dev);
    else if (form == 1) hipLaunchKernelGGL(k_cholinv_step<1>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, info_dev);
    else hipLaunchKernelGGL(k_cholinv_step<0>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, info_dev);
  }
  CCZ_LAUNCH_CHECK();
}

// X[b] = L[b]^-1 for `count` (<= 8) lower-triangular blocks whose 64 x 64 diagonal inverses T[b] (L_jj^-T, as
// k_wave_chol_inv / cholinv_batched leave them) are known: d_max / 64 launches, rows of all blocks advance together.
// Blocks of X above the diagonal are not written.
void trinv_batched(ccz_ctx* c, int count, const double* const* L, const int64_t* ldl, const int64_t* d, double* const* X,
                   const int64_t* ldx, const double* const* T) {
  if (count < 1 || count > CMAXB) fail(CCZ_EINVAL, "trinv_batched: 1..8 blocks per call");
  cholinv_attr_once();
  hipStream_t st = stream(c);
  CholInvBatch bt{};
  bt.count = count;
  bt.inv_only = 1;
  int nbmax = 0;
  for (int b = 0; b < count; ++b) {
    bt.A[b] = nullptr; bt.L[b] = const_cast<double*>(L[b]); bt.X[b] = X[b]; bt.T[b] = const_cast<double*>(T[b]);
    bt.lda[b] = 0; bt.ldl[b] = ldl[b]; bt.ldx[b] = ldx[b]; bt.d[b] = d[b];
    nbmax = std::max(nbmax, int((d[b] + CB - 1) / CB));
  }
  for (int j = 0; j < nbmax; ++j) {
    int total = 0;
    for (int b = 0; b < count; ++b) {
      bt.first[b] = total;
      total += j < int((d[b] + CB - 1) / CB) ? j + 1 : 0;
    }
    bt.first[count] = total;
    if (total == 0) continue;
    hipLaunchKernelGGL(k_cholinv_step<2>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, static_cast<int*>(nullptr));
  }
  CCZ_LAUNCH_CHECK();
}

// dst (lower triangle incl. diagonal) <- src, both d x d
__global__ void k_copy_lower(int64_t d, const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd) {
  const int64_t i = blockIdx.y;
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j <= i && j < d) dst[i * ldd + j] = src[i * lds_ + j];
}

void copy_lower(ccz_ctx* c, int64_t d, const double* src, int64_t lds_, double* dst, int64_t ldd) {
  if (d <= 0) return;
  if (d > 65535) fail(CCZ_EUNSUP, "copy_lower: d too large");
  hipLaunchKernelGGL(k_copy_lower, dim3((unsigned)((d + 255) / 256), (unsigned)d), dim3(256), 0, stream(c), d, src, lds_, dst, ldd);
  CCZ_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------
// Batched fp64 GEMM on 64 x 64 tiles: up to 8 independent problems per launch,
//   C = alpha op(A) op(B) + beta C    (+ optional transposed second destination Ct = C')
// The d x d products between the loss's Cholesky inverses and the batch covariances are individually too small to
// fill the chip (512^3: 64 tiles) and would otherwise be a dozen dependent launches.
// ---------------------------------------------------------------------------
constexpr int MG_MAX = 8;
struct MultiGemm {
  const double* A[MG_MAX];
  const double* B[MG_MAX];
  double* C[MG_MAX];
  double* Ct[MG_MAX];      // optional: receives C' (ld ldct)
  int64_t lda[MG_MAX], ldb[MG_MAX], ldc[MG_MAX], ldct[MG_MAX];
  int M[MG_MAX], N[MG_MAX], K[MG_MAX];
  int tA[MG_MAX], tB[MG_MAX];
  int lower_only[MG_MAX];  // skip tiles strictly above the diagonal (symmetric results; C square)
  int k_lower[MG_MAX];     // op(A) = X', op(B) = X with X lower triangular: only k >= max(m0, n0) contributes
  int ksplit[MG_MAX];      // > 1: the K range is cut into this many slices, C += alpha * (slice product) atomically
  double alpha[MG_MAX], beta[MG_MAX];
  const double* dotB[MG_MAX];   // optional: *dot_acc += dot_scale * sum_ij (alpha op(A) op(B))_ij dotB_ij (ld lddot) -- linear in the
  int64_t lddot[MG_MAX];        // product, so split-K slices simply add their shares (the DCCA loss value rides on the Gamma_ab stage)
  double dot_scale[MG_MAX];
  double* dot_acc;
  int first[MG_MAX + 1];
  int count;
};

constexpr size_t MG_LDS = size_t(4) * CTILE * sizeof(double);     // two (A, B) tile pairs: one sync per K chunk

__global__ __launch_bounds__(256) void k_gemm_f64_multi(MultiGemm g) {
  extern __shared__ __attribute__((aligned(16))) char mg_smem[];
  double* lds = reinterpret_cast<double*>(mg_smem);
  int p = 0;
  while (p + 1 < g.count && int(blockIdx.x) >= g.first[p + 1]) ++p;
  const int item = int(blockIdx.x) - g.first[p];
  const int M = g.M[p], N = g.N[p], K = g.K[p];
  const int tn = (N + CB - 1) / CB;
  const int ks = g.ksplit[p];
  const int tile = item / ks, slice = item - tile * ks;
  const int bm = tile / tn, bn = tile % tn;
  if (g.lower_only[p] && bn > bm) return;
  const int m0 = bm * CB, n0 = bn * CB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double* A = g.A[p];
  const double* B = g.B[p];
  const int64_t lda = g.lda[p], ldb = g.ldb[p];
  const bool tA = g.tA[p] != 0, tB = g.tB[p] != 0;
  const int rows_m = min(CB, M - m0), cols_n = min(CB, N - n0);
  int k_first = g.k_lower[p] ? max(m0, n0) : 0;
  int K_end = K;
  if (ks > 1) {                                            // this slice's run of 64-wide K chunks
    const int nch = (K - k_first + CB - 1) / CB, per = (nch + ks - 1) / ks;
    k_first += slice * per * CB;
    K_end = min(K, k_first + per * CB);
    if (k_first >= K_end) return;
  }
  v4f64 acc[4];
  acc_zero(acc);
  // As[m][k], Bs[k][n] regardless of the storage order of the operands; chunk i + 1 is fetched into registers while
  // chunk i is multiplied out of LDS, and lands in the other LDS tile pair (one barrier per chunk)
  TileRegs ra, rb;
  auto fetch = [&](int k0) {
    const int kc = min(CB, K_end - k0);
    if (!tA) fetch_tile(ra, A + int64_t(m0) * lda + k0, lda, rows_m, kc, tid);
    else fetch_tile(ra, A + int64_t(k0) * lda + m0, lda, kc, rows_m, tid);
    if (!tB) fetch_tile(rb, B + int64_t(k0) * ldb + n0, ldb, kc, cols_n, tid);
    else fetch_tile(rb, B + int64_t(n0) * ldb + k0, ldb, cols_n, kc, tid);
  };
  auto stash = [&](int buf) {
    double* As = lds + buf * 2 * CTILE;
    double* Bs = As + CTILE;
    if (!tA) store_tile<false>(As, ra, tid); else store_tile<true>(As, ra, tid);
    if (!tB) store_tile<false>(Bs, rb, tid); else store_tile<true>(Bs, rb, tid);
  };
  if (k_first < K_end) {
    fetch(k_first);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = k_first; k0 < K_end; k0 += CB) {
    const bool more = k0 + CB < K_end;
    if (more) fetch(k0 + CB);
    tile_mm<false, false>(lds + buf * 2 * CTILE, lds + buf * 2 * CTILE + CTILE, w, lane, acc);
    if (more) stash(buf ^ 1);          // the other pair: nobody reads it during this chunk
    __syncthreads();
    buf ^= 1;
  }
  const double alpha = g.alpha[p], beta = g.beta[p];
  double* C = g.C[p];
  double* Ct = g.Ct[p];
  const int64_t ldc = g.ldc[p], ldct = g.ldct[p];
  const double* dotB = g.dotB[p];
  if (dotB) {                                                // uniform per workgroup
    const int64_t lddot = g.lddot[p];
    double dsum = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        if (rr < rows_m && cc < cols_n) dsum += alpha * acc[t][rg] * dotB[int64_t(m0 + rr) * lddot + n0 + cc];
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
    __syncthreads();                                         // every wave is done with the operand tiles: reuse their first doubles
    if (lane == 0) lds[w] = dsum;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(g.dot_acc, g.dot_scale[p] * (lds[0] + lds[1] + lds[2] + lds[3]));
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
      if (rr < rows_m && cc < cols_n) {
        double v = alpha * acc[t][rg];
        if (ks > 1) {
          atomicAdd(C + int64_t(m0 + rr) * ldc + n0 + cc, v);
          continue;
        }
        if (C) {
          double* q = C + int64_t(m0 + rr) * ldc + n0 + cc;
          if (beta != 0.0) v += beta * *q;
          *q = v;
        }
        if (Ct) Ct[int64_t(n0 + cc) * ldct + m0 + rr] = v;
      }
    }
}

// problems: arrays of `count` entries; Ct may be null (or hold nulls)
void gemm_f64_multi(ccz_ctx* c, int count, const MultiGemmArgs* pr) {
  if (count < 1 || count > MG_MAX) fail(CCZ_EINVAL, "gemm_f64_multi: 1..8 problems per launch");
  MultiGemm g{};
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    const MultiGemmArgs& a = pr[i];
    if (a.M < 1 || a.N < 1 || a.K < 1 || !a.A || !a.B || (!a.C && !a.Ct)) fail(CCZ_EINVAL, "gemm_f64_multi: bad problem %d", i);
    g.A[i] = a.A; g.B[i] = a.B; g.C[i] = a.C; g.Ct[i] = a.Ct;
    g.lda[i] = a.lda; g.ldb[i] = a.ldb; g.ldc[i] = a.ldc; g.ldct[i] = a.ldct;
    g.M[i] = int(a.M); g.N[i] = int(a.N); g.K[i] = int(a.K);
    g.tA[i] = a.tA ? 1 : 0; g.tB[i] = a.tB ? 1 : 0;
    g.lower_only[i] = a.lower_only ? 1 : 0;
    g.k_lower[i] = a.k_lower ? 1 : 0;
    g.ksplit[i] = std::max(1, a.ksplit);
    if (g.ksplit[i] > 1 && (a.beta != 1.0 || a.Ct || !a.C)) fail(CCZ_EINVAL, "gemm_f64_multi: split-K accumulates into C (beta = 1, no Ct)");
    g.alpha[i] = a.alpha; g.beta[i] = a.beta;
    g.dotB[i] = a.dotB; g.lddot[i] = a.lddot; g.dot_scale[i] = a.dot_scale;
    if (a.dotB) {
      if (!a.dot_acc || (g.dot_acc && g.dot_acc != a.dot_acc)) fail(CCZ_EINVAL, "gemm_f64_multi: one dot accumulator per launch");
      if (g.ksplit[i] <= 1 && a.beta != 0.0) fail(CCZ_EINVAL, "gemm_f64_multi: the dot rider needs beta = 0 or split-K");
      g.dot_acc = a.dot_acc;
    }
    g.first[i] = total;
    total += int((a.M + CB - 1) / CB) * int((a.N + CB - 1) / CB) * g.ksplit[i];
  }
  g.first[count] = total;
  cholinv_attr_once();
  hipLaunchKernelGGL(k_gemm_f64_multi, dim3(total), dim3(256), MG_LDS, stream(c), g);
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
