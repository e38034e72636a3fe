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
#include <cstdlib>

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
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    double piv = bcast_lane(pv[t], C0 + t);
    const bool bad = !(piv > 0.0);                     // wave-uniform; NaN counts as bad
    first_bad = bad ? min(first_bad, C0 + t) : first_bad;
    piv = bad ? 1.0 : piv;
    const double rs = rsqrt_cubic(piv);
    l[t] = lane >= C0 + t ? pv[t] * rs : 0.0;          // select, not multiply: lanes above hold stale / foreign values
    if (lane == 0) Rd[C0 + t] = rs;                    // 1 / L_jj for the substitution of the inverse
#pragma unroll
    for (int u = t + 1; u < 16; ++u) pv[u] = __builtin_fma(-l[t], bcast_lane(l[t], C0 + u), pv[u]);
  }
  // column-major factor image Ls[col][row] (whole columns, zeros above the diagonal) -- also the MFMA operand image
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

// all four waves: Xs[col * CLD + row] = (L^-1)[row][col]  (== row-major L^-T) from Ls / Rd
__device__ __forceinline__ void inv64_mfma(const double* Ls, const double* Rd, double* Xs, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  // zero the blocks above the block diagonal (the substitution leaves exact zeros inside the diagonal blocks)
  for (int e = tid; e < CB * CB; e += 256) {
    const int row = e & 63, col = e >> 6;
    if ((row >> 4) < (col >> 4)) Xs[col * CLD + row] = 0.0;
  }
  // phase A: wave w inverts the 16 x 16 diagonal block w by forward substitution, lane c < 16 = right-hand side e_c
  {
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
  __syncthreads();
  // phase B: wave j forms column block j below the diagonal, top to bottom
  const int j = w;
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

// Whole workgroup (256 threads), input tile in LDS tile 1 (identity-padded): factor (tile 0 <- L, column-major),
// invert (tile 2 <- L^-T, row-major), publish the L block and T = L^-T to global memory.
// FORM: 0 shift-register recurrence on two waves, 1 MFMA form with 4-column panels (rounds 2-4), 2 MFMA form with
// 16-column panels (default)
template <int FORM>
__device__ __forceinline__ void factor_and_publish_t(double* lds, int tid, int nbv, double* __restrict__ Lblk, int64_t ldl,
                                                     double* __restrict__ Tblk, int* __restrict__ info, int64_t col0) {
  double* Ls = lds;
  const double* In = lds + CTILE;
  double* Xs = lds + 2 * CTILE;
  double* Rd = lds + 3 * CTILE;
  int* progress = reinterpret_cast<int*>(Rd + CB);
  double* PL = Rd + CB + 8;
  if (FORM != 0) {
    __syncthreads();
    if (tid < 64) {
      const int bad = FORM == 2 ? chol64_p16(In, Ls, Rd, PL, tid) : chol64_mfma(In, Ls, Rd, PL, tid);
      if (tid == 0 && bad != 0x7fffffff) atomicMin(info, int(col0 + bad + 1));
    }
    __syncthreads();
    inv64_mfma(Ls, Rd, Xs, tid);
  } else {
    if (tid == 0) *progress = 0;
    __syncthreads();
    if (tid < 128) {
      const int bad = wave_factor_lds(Ls, In, Xs, Rd, progress, tid);
      if (tid == 0 && bad != 0x7fffffff) atomicMin(info, int(col0 + bad + 1));
    }
  }
  __syncthreads();
  const int c = tid & 63;
  for (int r = tid >> 6; r < CB; r += 4) {
    if (r < nbv && c <= r) Lblk[int64_t(r) * ldl + c] = Ls[c * CLD + r];
    Tblk[r * CB + c] = Xs[r * CLD + c];
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
};

// helper items of matrix (nb block columns, with / without X) at link j: trailing tiles without the chain's, XT row j, XS row j + 1
__host__ __device__ __forceinline__ void chain_counts(int nb, bool with_x, int j, int& hU, int& nXT, int& nXS) {
  const int r = nb - 1 - j;
  const int nU = r > 0 ? r * (r + 1) / 2 : 0;
  hU = nU > 0 ? nU - 1 : 0;
  nXT = (with_x && j < nb) ? j + 1 : 0;
  nXS = (with_x && j + 1 < nb) ? j + 1 : 0;
}

// Thread 0 polls *p (agent scope) until it reaches `target`; the workgroup follows through a barrier and acquires.
// flag: one LDS int.  Returns false after the watchdog / an abort raised elsewhere (the caller goes on with whatever is
// in memory: the launch must terminate, its result is reported as failed).
__device__ __forceinline__ bool chain_wait(unsigned* p, unsigned target, ChainSync* sy, int* flag, int tid) {
  if (tid == 0) {
    int good = 1, spins = 0;
    while (int(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(8);
      ++spins;
      if ((spins & 63) == 0 && __hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { good = 0; break; }
      if (spins > CH_WATCHDOG) {
        __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    *flag = good;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const bool ok = *flag != 0;
  __syncthreads();                     // the next wait may overwrite the flag
  return ok;
}

// every thread's global stores of this item -> visible device-wide, then one counter update
__device__ __forceinline__ void chain_publish_add(unsigned* p, int tid) {
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_publish_set(unsigned* p, unsigned v, int tid) {
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// lds: tiles 0..2, Rd, flags, panel image (the layout factor_and_publish_t expects)
template <int FORM>
__device__ __forceinline__ void chain_matrix(const CholInvBatch& bt, int b, ChainSync* sy, int* __restrict__ info, double* lds, int* flag,
                                             int tid) {
  const int64_t d = bt.d[b];
  const int nb = int((d + CB - 1) / CB);
  const bool with_x = bt.X[b] != nullptr;
  const int64_t lda = bt.lda[b], ldl = bt.ldl[b];
  double* A = bt.A[b];
  double* L = bt.L[b];
  double* T = bt.T[b];
  double* P = lds;               // tile 0: becomes Ls of the factorization
  double* Q = lds + CTILE;       // tile 1: input of the factorization
  double* TT = lds + 2 * CTILE;  // tile 2: T_j = L_jj^-T, left there by the factorization of block j
  const int lane = tid & 63, w = tid >> 6;
  bool ok = true;
  {
    const int nbv = int(min<int64_t>(CB, d));
    if (tid == 0) info[b] = 0x7fffffff;
    const int c = tid & 63;
    for (int r = tid >> 6; r < CB; r += 4) {
      double v = (r == c) ? 1.0 : 0.0;
      if (r < nbv && c < nbv) v = c <= r ? A[int64_t(r) * lda + c] : A[int64_t(c) * lda + r];
      Q[r * CLD + c] = v;
    }
    factor_and_publish_t<FORM>(lds, tid, nbv, L, ldl, T, info + b, 0);
    chain_publish_set(&sy->F[b], 1u, tid);
  }
  for (int j = 0; j + 1 < nb; ++j) {
    if (j >= 1) {                                        // tiles (j+1, j) and (j+1, j+1) carry the updates of the links < j
      int hU, nXT, nXS;
      chain_counts(nb, with_x, j - 1, hU, nXT, nXS);
      if (hU > 0) ok = chain_wait(&sy->U[b][j - 1], unsigned(hU), sy, flag, tid) && ok;
    }
    const int i = j + 1;
    const int64_t ri = int64_t(i) * CB, cj = int64_t(j) * CB;
    const int rows_i = int(min<int64_t>(CB, d - ri));
    load_tile<false>(P, A + ri * lda + cj, lda, rows_i, CB, tid);
    __syncthreads();
    v4f64 a1[4];
    acc_zero(a1);
    tile_mm<false, false>(P, TT, w, lane, a1);            // L_ij = A_ij T_j
    __syncthreads();
    acc_to_lds(a1, P, w, lane, 1.0);
    __syncthreads();
    {
      const int c = tid & 63;
      for (int rr = tid >> 6; rr < rows_i; rr += 4) L[(ri + rr) * ldl + cj + c] = P[rr * CLD + c];
    }
    chain_publish_set(&sy->G[b], unsigned(i), tid);       // row i of X may start its partial sums
    v4f64 u[4];
    acc_zero(u);
    tile_mm<false, true>(P, P, w, lane, u);               // L_ij L_ij'
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
        double v = (rr == cc) ? 1.0 : 0.0;
        if (rr < rows_i && cc < rows_i) {
          const double base = cc <= rr ? A[(ri + rr) * lda + ri + cc] : A[(ri + cc) * lda + ri + rr];
          v = base - u[t][rg];
        }
        Q[rr * CLD + cc] = v;
      }
    // (the barrier at the head of factor_and_publish_t separates the reads of P above from the factor image written there)
    factor_and_publish_t<FORM>(lds, tid, rows_i, L + ri * ldl + ri, ldl, T + int64_t(i) * CB * CB, info + b, ri);
    chain_publish_set(&sy->F[b], unsigned(i + 1), tid);
  }
  if (!ok && tid == 0) atomicMin(info + b, CH_TIMEOUT_INFO);
}

// lds4: FOUR tiles (extra tile first), then Rd / flags / panel image
__device__ __forceinline__ void chain_helper(const CholInvBatch& bt, const ChainPlan& plan, int q, ChainSync* sy, int* __restrict__ info,
                                             double* lds4, int* flag, int tid) {
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
  const bool with_x = bt.X[b] != nullptr;
  const int64_t d = bt.d[b];
  const int64_t lda = bt.lda[b], ldl = bt.ldl[b], ldx = bt.ldx[b];
  double* A = bt.A[b];
  double* L = bt.L[b];
  double* X = bt.X[b];
  const double* Tj = bt.T[b] + int64_t(j) * CB * CB;
  const int64_t cj = int64_t(j) * CB;
  const int lane = tid & 63, w = tid >> 6;
  double* P = lds4 + CTILE;
  double* Q = lds4 + 2 * CTILE;
  double* TT = lds4 + 3 * CTILE;
  bool ok = true;

  if (q < hU) {
    // ---- trailing tile (i, k) of link j, j < k <= i, (i, k) != (j+1, j+1) ----
    ok = chain_wait(&sy->F[b], unsigned(j + 1), sy, flag, tid) && ok;
    if (j >= 1) {
      int pU, a_, b_;
      chain_counts(nb, with_x, j - 1, pU, a_, b_);
      if (pU > 0) ok = chain_wait(&sy->U[b][j - 1], unsigned(pU), sy, flag, tid) && ok;
    }
    const int item = q + 1;
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
    chain_publish_add(&sy->U[b][j], tid);
  } else if (q < hU + nXT) {
    // ---- XT(j, k): X_jk = -T_j' S_jk, X_jj = T_j' ----
    const int k = q - hU;
    const int rows_j = int(min<int64_t>(CB, d - cj));
    ok = chain_wait(&sy->F[b], unsigned(j + 1), sy, flag, tid) && ok;
    if (k == j) {
      load_tile<true>(P, Tj, CB, CB, CB, tid);
      __syncthreads();
      const int c = tid & 63;
      for (int rr = tid >> 6; rr < rows_j; rr += 4)
        if (c < rows_j) X[(cj + rr) * ldx + cj + c] = P[rr * CLD + c];
    } else {
      ok = chain_wait(&sy->XS[b][j], unsigned(j), sy, flag, tid) && ok;
      load_tile<false>(P, X + cj * ldx + int64_t(k) * CB, ldx, rows_j, CB, tid);     // S_jk
      load_tile<false>(TT, Tj, CB, CB, CB, tid);
      __syncthreads();
      v4f64 o[4];
      acc_zero(o);
      tile_mm<true, false>(TT, P, w, lane, o);            // T_j' S
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int rr = 16 * w + (lane >> 4) + 4 * rg, cc = 16 * t + (lane & 15);
          if (rr < rows_j) X[(cj + rr) * ldx + int64_t(k) * CB + cc] = -o[t][rg];
        }
    }
    chain_publish_add(&sy->XR[b][j], tid);
  } else {
    // ---- XS(i, k), i = j + 1: S_ik = sum_{t=k}^{i-1} L_it X_tk, stored where X_ik will be ----
    const int k = q - hU - nXT;
    const int i = j + 1;
    const int64_t ri = int64_t(i) * CB;
    const int rows_i = int(min<int64_t>(CB, d - ri));
    ok = chain_wait(&sy->G[b], unsigned(i), sy, flag, tid) && ok;           // L_{i,i-1} (and, through the chain, every L_it)
    ok = chain_wait(&sy->XR[b][j], unsigned(j + 1), sy, flag, tid) && ok;   // rows <= j of X
    v4f64 sacc[4];
    acc_zero(sacc);
    TileRegs ra, rb;
    auto fetch = [&](int t) {
      fetch_tile(ra, L + ri * ldl + int64_t(t) * CB, ldl, rows_i, CB, tid);                      // L_it
      fetch_tile(rb, X + int64_t(t) * CB * ldx + int64_t(k) * CB, ldx, CB, CB, tid);             // X_tk
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
        if (rr < rows_i) X[(ri + rr) * ldx + int64_t(k) * CB + cc] = sacc[t][rg];
      }
    chain_publish_add(&sy->XS[b][i], tid);
  }
  if (!ok && tid == 0) atomicMin(info + b, CH_TIMEOUT_INFO);
}

template <int FORM>
__global__ __launch_bounds__(256) void k_cholinv_chain(CholInvBatch bt, ChainPlan plan, ChainSync* __restrict__ sy, int* __restrict__ info) {
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
    if (ticket < bt.count) chain_matrix<FORM>(bt, ticket, sy, info, lds, flags + 2, tid);
    else chain_helper(bt, plan, ticket - bt.count, sy, info, lds4, flags + 2, tid);
  }
  // the last workgroup to leave clears the sync block for the next launch (nobody reads or writes it any more)
  __threadfence();
  __syncthreads();
  if (tid == 0) flags[3] = __hip_atomic_fetch_add(&sy->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (flags[3]) {
    unsigned* wds = reinterpret_cast<unsigned*>(sy);
    for (int e = tid; e < int(sizeof(ChainSync) / sizeof(unsigned)); e += 256) wds[e] = 0u;
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
// (rounds 2-4), 2 (default) MFMA form with 16-column panels
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
  ChainSync* sy = chain_sync_for(c);
  if (!sy) return false;
  Impl* im = impl(c);
  const int cap = im->chain_cap > 0 ? im->chain_cap : std::max(wgs_env, 2);
  // at least one helper workgroup next to the chain workgroups (they never leave their matrix)
  const int grid = std::min(plan.total, std::max(cap, bt.count + 1));
  if (form == 2) hipLaunchKernelGGL(k_cholinv_chain<2>, dim3(grid), dim3(256), CHAIN_LDS, stream(c), bt, plan, sy, info_dev);
  else hipLaunchKernelGGL(k_cholinv_chain<1>, dim3(grid), dim3(256), CHAIN_LDS, stream(c), bt, plan, sy, info_dev);
  CCZ_LAUNCH_CHECK();
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
