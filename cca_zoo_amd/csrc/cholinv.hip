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
constexpr size_t CHOLINV_LDS = (size_t(3) * CTILE + 2 * CB + 8 + 4 * CB) * sizeof(double);   // 3 tiles, Rd, flags, 64 x 4 panel

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
template <bool MFMA_FORM>
__device__ __forceinline__ void factor_and_publish_t(double* lds, int tid, int nbv, double* __restrict__ Lblk, int64_t ldl,
                                                     double* __restrict__ Tblk, int* __restrict__ info, int64_t col0) {
  double* Ls = lds;
  const double* In = lds + CTILE;
  double* Xs = lds + 2 * CTILE;
  double* Rd = lds + 3 * CTILE;
  int* progress = reinterpret_cast<int*>(Rd + CB);
  double* PL = Rd + CB + 8;
  if (MFMA_FORM) {
    __syncthreads();
    if (tid < 64) {
      const int bad = chol64_mfma(In, Ls, Rd, PL, tid);
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
template <bool MFMA_FORM>
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
  factor_and_publish_t<MFMA_FORM>(lds, tid, nbv, bt.L[b], bt.ldl[b], bt.T[b], info + b, 0);
}

// step j: trailing update (+ look-ahead factorization of block j + 1) and row j of the inverse
template <bool MFMA_FORM>
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
    factor_and_publish_t<MFMA_FORM>(lds, tid, rows_i, L + ri * ldl + ri, ldl, bt.T[b] + int64_t(j + 1) * CB * CB, info + b, ri);
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

static void cholinv_attr_once() {
  static thread_local int done_for_device = -1;
  int dev = -1;
  CCZ_HIP(hipGetDevice(&dev));
  if (done_for_device == dev) return;
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_first<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_step<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_first<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cholinv_step<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(CHOLINV_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_multi), hipFuncAttributeMaxDynamicSharedMemorySize, int(MG_LDS_FWD)));
  done_for_device = dev;
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
  // CCZ_CHOLINV_MFMA=0 selects the shift-register (two-wave) form of the 64 x 64 factorization
  static const int mfma_form = [] { const char* e = getenv("CCZ_CHOLINV_MFMA"); return e ? atoi(e) : 1; }();
  if (mfma_form) hipLaunchKernelGGL(k_cholinv_first<true>, dim3(count), dim3(256), CHOLINV_LDS, st, bt, info_dev);
  else hipLaunchKernelGGL(k_cholinv_first<false>, dim3(count), dim3(256), CHOLINV_LDS, st, bt, info_dev);
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
    if (mfma_form) hipLaunchKernelGGL(k_cholinv_step<true>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, info_dev);
    else hipLaunchKernelGGL(k_cholinv_step<false>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, info_dev);
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
    hipLaunchKernelGGL(k_cholinv_step<true>, dim3(total), dim3(256), CHOLINV_LDS, st, bt, j, static_cast<int*>(nullptr));
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
    g.first[i] = total;
    total += int((a.M + CB - 1) / CB) * int((a.N + CB - 1) / CB) * g.ksplit[i];
  }
  g.first[count] = total;
  cholinv_attr_once();
  hipLaunchKernelGGL(k_gemm_f64_multi, dim3(total), dim3(256), MG_LDS, stream(c), g);
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
