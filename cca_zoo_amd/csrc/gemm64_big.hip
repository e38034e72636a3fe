// Large fp64 GEMM for the solver stage:  C (M x N) = alpha op(A) op(B) + beta C   on v_mfma_f64_16x16x4_f64.
//
// 128 x 128 tile per workgroup, 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA tiles (128 accumulator
// registers, two workgroups per CU).  Both operands are staged into LDS as k-major images [k][m] / [k][n]
// so that a lane's fragment is two ds_read_b128 (four consecutive m or n of row k; the four values feed
// four MFMA tiles, tile t owning the columns == t mod 4 -- the same strided ownership as K1).
//   * an operand that is already k-major in memory (A with transA, B without transB) is copied row by row;
//   * an operand that is m-major (A without transA, B with transB) is transposed while staging: a thread
//     owns one row, reads 64 contiguous bytes of it and scatters 8 ds_write_b64 (consecutive lanes ->
//     consecutive addresses).
// Rows / columns past M / N are clamped on load and masked on store; K must be a multiple of 16.
// The recursive Cholesky / triangular solves in ops_hip.hip put ~all solver flops through this kernel
// with K >= 128 (the 64 x 64-tile generic kernel keeps the small and ragged products).
#include <algorithm>
#include <cstdlib>

#include "hip_common.h"

namespace ccz {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int DT = 128;   // tile edge
constexpr int DK = 16;    // k-block

constexpr int DS = DT + 2;   // LDS row stride (doubles): 16-byte aligned rows, transposing writes at most 2-way conflicted

// stage a 16 (k) x 128 (m) block of an operand into `dst` ([16][DS] doubles)
//   KMAJOR: element (k, m) at P[k * ld + m]: a wave copies one full 1-KiB row per load instruction
//   else  : element (k, m) at P[m * ld + k]: 8 lanes cover the 16 k (128 contiguous bytes) of a row, a wave
//           covers 8 rows per load instruction (full cache lines), and the 2 doubles are scattered to [k][m]
template <bool KMAJOR>
struct Stager {
  v2f64 r[4];
  __device__ __forceinline__ void load(const double* __restrict__ P, int64_t ld, int64_t m0, int64_t mlim, int64_t k0,
                                       int tid) {
    if (KMAJOR) {
      const int64_t m = m0 + 2 * (tid & 63);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double* p = P + (k0 + (tid >> 6) + 4 * i) * ld;
        if (m + 1 < mlim) {
          r[i] = *reinterpret_cast<const v2f64*>(p + m);
        } else {
          r[i][0] = m < mlim ? p[m] : 0.0;
          r[i][1] = 0.0;
        }
      }
    } else {
      const int kq = tid & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = std::min<int64_t>(m0 + (tid >> 3) + 32 * i, mlim - 1);
        r[i] = *reinterpret_cast<const v2f64*>(P + m * ld + k0 + 2 * kq);
      }
    }
  }
  __device__ __forceinline__ void store(double* dst, int tid) const {
    if (KMAJOR) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<v2f64*>(dst + ((tid >> 6) + 4 * i) * DS + 2 * (tid & 63)) = r[i];
    } else {
      const int kq = tid & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = (tid >> 3) + 32 * i;
        dst[(2 * kq) * DS + m] = r[i][0];
        dst[(2 * kq + 1) * DS + m] = r[i][1];
      }
    }
  }
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void k_gemm_f64_big(int64_t M, int64_t N, int64_t K, double alpha,
                                                         const double* __restrict__ A, int64_t lda,
                                                         const double* __restrict__ B, int64_t ldb, double beta,
                                                         double* __restrict__ C, int64_t ldc, int lower_only,
                                                         int64_t k_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);  // [2 buffers][A | B][16][DS]
  const int64_t m0 = int64_t(blockIdx.y) * DT, n0 = int64_t(blockIdx.x) * DT;
  if (lower_only && n0 > m0 + (DT - 1)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  // A is k-major in memory iff TA (stored K x M); B is k-major iff !TB (stored K x N)
  Stager<TA> sa;
  Stager<!TB> sb;
  // split-K: slice z accumulates its K range into C with fp64 atomics (the launcher pre-scaled C by beta)
  const bool split = gridDim.z > 1;
  const int64_t kz0 = int64_t(blockIdx.z) * k_per_split;
  const int64_t nkb = (min(K, kz0 + k_per_split) - kz0) / DK;
  if (nkb <= 0) return;
  sa.load(A, lda, m0, M, kz0, tid);
  sb.load(B, ldb, n0, N, kz0, tid);
  sa.store(lds, tid);
  sb.store(lds + DK * DS, tid);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) {
      sa.load(A, lda, m0, M, kz0 + (kb + 1) * DK, tid);
      sb.load(B, ldb, n0, N, kz0 + (kb + 1) * DK, tid);
    }
    const double* as = lds + cur * (2 * DK * DS);
    const double* bs = as + DK * DS;
#pragma unroll
    for (int kk = 0; kk < DK / 4; ++kk) {
      const int krow = 4 * kk + (lane >> 4);
      const double* ap = as + krow * DS + wr * 64 + 4 * (lane & 15);
      const double* bp = bs + krow * DS + wc * 64 + 4 * (lane & 15);
      const v2f64 a01 = *reinterpret_cast<const v2f64*>(ap), a23 = *reinterpret_cast<const v2f64*>(ap + 2);
      const v2f64 b01 = *reinterpret_cast<const v2f64*>(bp), b23 = *reinterpret_cast<const v2f64*>(bp + 2);
      const double a4[4] = {a01[0], a01[1], a23[0], a23[1]};
      const double b4[4] = {b01[0], b01[1], b23[0], b23[1]};
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[ti], b4[tj], acc[ti][tj], 0, 0, 0);
    }
    if (kb + 1 < nkb) {
      double* nx = lds + (cur ^ 1) * (2 * DK * DS);
      sa.store(nx, tid);
      sb.store(nx + DK * DS, tid);
    }
    __syncthreads();
  }

  // f64 16x16 C/D layout: col (B side) = lane & 15, row (A side) = (lane >> 4) + 4 * reg
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wr * 64 + 4 * ((lane >> 4) + 4 * r) + ti;
      if (m >= M) continue;
      const int64_t nb = n0 + wc * 64 + 4 * (lane & 15);
      double* cp = C + m * ldc + nb;
      if (split) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          if (nb + tj < N) unsafeAtomicAdd(cp + tj, alpha * acc[ti][tj][r]);
      } else if (nb + 3 < N) {
        v4f64 v = {acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]};
        v *= alpha;
        if (beta != 0.0) {
          const v2f64 c01 = *reinterpret_cast<const v2f64*>(cp), c23 = *reinterpret_cast<const v2f64*>(cp + 2);
          v[0] += beta * c01[0]; v[1] += beta * c01[1]; v[2] += beta * c23[0]; v[3] += beta * c23[1];
        }
        *reinterpret_cast<v2f64*>(cp) = v2f64{v[0], v[1]};
        *reinterpret_cast<v2f64*>(cp + 2) = v2f64{v[2], v[3]};
      } else {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          if (nb + tj < N) {
            double v = alpha * acc[ti][tj][r];
            if (beta != 0.0) v += beta * cp[tj];
            cp[tj] = v;
          }
      }
    }
}

// ---------------------------------------------------------------------------
// 64 x 128 tile variant for grids that the 128 x 128 tile cannot fill.  The super-block products of the triangular
// solves and Cholesky panels are M x 512 x 512: at M = 4096 that is 128 tiles on 256 CUs, and a tile's 128 x 128 x 512
// block is 55 us of fp64 MFMA time on its CU whatever the rest of the chip does.  Halving the tile height doubles
// the workgroups (one per CU) and halves each one's MFMA time; the four waves sit side by side (64 rows x 32 columns
// each: 4 x 2 MFMA tiles, all waves read the same A fragment rows).
// ---------------------------------------------------------------------------
constexpr int HM = 64;         // tile rows
constexpr int HSA = HM + 2;    // LDS row stride of the A image

template <bool KMAJOR, int W>   // 16 (k) x W (m) block, W = 64 or 128; same two layouts as Stager
struct StagerW {
  static constexpr int NI = W / 32;          // 16-byte loads per thread
  static constexpr int TPR = W / 2;          // k-major: threads per k-row
  static constexpr int RPP = 256 / TPR;      // k-major: k-rows per pass
  v2f64 r[NI];
  __device__ __forceinline__ void load(const double* __restrict__ P, int64_t ld, int64_t m0, int64_t mlim, int64_t k0, int tid) {
    if (KMAJOR) {
      const int64_t m = m0 + 2 * (tid % TPR);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const double* p = P + (k0 + tid / TPR + RPP * i) * ld;
        if (m + 1 < mlim) {
          r[i] = *reinterpret_cast<const v2f64*>(p + m);
        } else {
          r[i][0] = m < mlim ? p[m] : 0.0;
          r[i][1] = 0.0;
        }
      }
    } else {
      const int kq = tid & 7;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t m = std::min<int64_t>(m0 + (tid >> 3) + 32 * i, mlim - 1);
        r[i] = *reinterpret_cast<const v2f64*>(P + m * ld + k0 + 2 * kq);
      }
    }
  }
  __device__ __forceinline__ void store(double* dst, int stride, int tid) const {
    if (KMAJOR) {
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<v2f64*>(dst + (tid / TPR + RPP * i) * stride + 2 * (tid % TPR)) = r[i];
    } else {
      const int kq = tid & 7;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = (tid >> 3) + 32 * i;
        dst[(2 * kq) * stride + m] = r[i][0];
        dst[(2 * kq + 1) * stride + m] = r[i][1];
      }
    }
  }
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void k_gemm_f64_half(int64_t M, int64_t N, int64_t K, double alpha,
                                                          const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb, double beta,
                                                          double* __restrict__ C, int64_t ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem_h[];
  double* lds = reinterpret_cast<double*>(smem_h);   // [2 buffers][A: 16 x HSA | B: 16 x DS]
  constexpr int BUF = DK * (HSA + DS);
  const int64_t m0 = int64_t(blockIdx.y) * HM, n0 = int64_t(blockIdx.x) * DT;
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;     // wave wc: columns [32 wc, 32 wc + 32)
  v4f64 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  StagerW<TA, HM> sa;
  StagerW<!TB, DT> sb;
  const int64_t nkb = K / DK;
  sa.load(A, lda, m0, M, 0, tid);
  sb.load(B, ldb, n0, N, 0, tid);
  sa.store(lds, HSA, tid);
  sb.store(lds + DK * HSA, DS, tid);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) {
      sa.load(A, lda, m0, M, (kb + 1) * DK, tid);
      sb.load(B, ldb, n0, N, (kb + 1) * DK, tid);
    }
    const double* as = lds + cur * BUF;
    const double* bs = as + DK * HSA;
#pragma unroll
    for (int kk = 0; kk < DK / 4; ++kk) {
      const int krow = 4 * kk + (lane >> 4);
      const double* ap = as + krow * HSA + 4 * (lane & 15);
      const double* bp = bs + krow * DS + wc * 32 + 2 * (lane & 15);
      const v2f64 a01 = *reinterpret_cast<const v2f64*>(ap), a23 = *reinterpret_cast<const v2f64*>(ap + 2);
      const v2f64 b01 = *reinterpret_cast<const v2f64*>(bp);
      const double a4[4] = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        acc[ti][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[ti], b01[0], acc[ti][0], 0, 0, 0);
        acc[ti][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[ti], b01[1], acc[ti][1], 0, 0, 0);
      }
    }
    if (kb + 1 < nkb) {
      double* nx = lds + (cur ^ 1) * BUF;
      sa.store(nx, HSA, tid);
      sb.store(nx + DK * HSA, DS, tid);
    }
    __syncthreads();
  }
  // C/D layout of the 16 x 16 tile: col (B side) = lane & 15, row (A side) = (lane >> 4) + 4 * reg; tile ti owns the rows
  // == ti mod 4, tile tj the columns == tj mod 2 of the wave's 32
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + 4 * ((lane >> 4) + 4 * r) + ti;
      if (m >= M) continue;
      const int64_t nb = n0 + wc * 32 + 2 * (lane & 15);
      double* cp = C + m * ldc + nb;
      if (nb + 1 < N) {
        v2f64 v = {alpha * acc[ti][0][r], alpha * acc[ti][1][r]};
        if (beta != 0.0) {
          const v2f64 c01 = *reinterpret_cast<const v2f64*>(cp);
          v[0] += beta * c01[0];
          v[1] += beta * c01[1];
        }
        *reinterpret_cast<v2f64*>(cp) = v;
      } else if (nb < N) {
        double v = alpha * acc[ti][0][r];
        if (beta != 0.0) v += beta * cp[0];
        cp[0] = v;
      }
    }
}

// ---------------------------------------------------------------------------
// Round 6: both tiles on ONE software-pipelined loop (k_gemm_f64_pipe<TA, TB, TM>, TM = 128 or 64).  The kernels above keep a
// k-block's MFMA stream apart from its staging: loads at the top, 64 (32) MFMAs, `s_waitcnt vmcnt(0)`, transposing ds_writes,
// `s_waitcnt lgkmcnt(0)`, barrier, and the next block starts with the latency of its first fragment reads -- 400 - 500 idle
// cycles per 4096 (2048) MFMA cycles with one workgroup per CU, and two workgroups per CU fall into step.  Measured with
// tools/gemm64_probe.py (profiles/r06_gemm64_probe.log): steady state 77 % (one workgroup per CU) / 84 % (two) of the fp64 matrix
// peak against rocBLAS' 92 / 98 %.  Here the staging rides INSIDE the MFMA stream (an fp64 MFMA holds the pipe for 64 cycles: every
// one of them hides a few other instructions):
//   kk = 0:  fragments of kk = 1 from LDS                                              | 16 (8) MFMAs of kk = 0
//   kk = 1:  fragments of kk = 2; registers (block kb + 1) -> the other LDS buffer;     | MFMAs of kk = 1
//            global loads of block kb + 2 into the same registers (a whole block of latency budget)
//   kk = 2:  fragments of kk = 3                                                       | MFMAs of kk = 2
//            s_waitcnt lgkmcnt(0); s_barrier   (everybody's writes have landed; nobody reads this block's buffer any more)
//   kk = 3:  fragments of kk = 0 of block kb + 1 from the other buffer                 | MFMAs of kk = 3
// One barrier per k-block, never followed by a latency the MFMA pipe has to wait out; global loads stay in flight across it.
// ---------------------------------------------------------------------------
template <bool TA, bool TB, int TM>
__global__ __launch_bounds__(256, 2) void k_gemm_f64_pipe(int64_t M, int64_t N, int64_t K, double alpha,
                                                          const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb, double beta,
                                                          double* __restrict__ C, int64_t ldc, int lower_only,
                                                          int64_t k_per_split) {
  constexpr int WN = TM == 128 ? 64 : 32;     // columns of a wave
  constexpr int NB = WN / 16;                 // its MFMA tiles along N (4 or 2); along M always 4
  constexpr int SA = TM + 2;                  // row stride of the A image (doubles)
  constexpr int BUF = DK * (SA + DS);
  constexpr int NSTG = TM / 32 + DT / 32;     // 16-byte global loads (and LDS write instructions) of a thread per k-block: 8 or 6
  extern __shared__ __attribute__((aligned(16))) char smem_p[];
  double* lds = reinterpret_cast<double*>(smem_p);   // [2 buffers][A: 16 x SA | B: 16 x DS]
  const int64_t m0 = int64_t(blockIdx.y) * TM, n0 = int64_t(blockIdx.x) * DT;
  if (lower_only && n0 > m0 + (TM - 1)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a_off = TM == 128 ? (wave >> 1) * 64 : 0;
  const int b_off = TM == 128 ? (wave & 1) * 64 : wave * 32;

  v4f64 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  StagerW<TA, TM> sa;
  StagerW<!TB, DT> sb;
  const bool split = gridDim.z > 1;
  const int64_t kz0 = int64_t(blockIdx.z) * k_per_split;
  const int64_t nkb = (min(K, kz0 + k_per_split) - kz0) / DK;
  if (nkb <= 0) return;

  // fragment of k-group kk: 4 consecutive rows of A (MFMA tile ti owns the rows == ti mod 4 of the wave's 64), NB consecutive
  // columns of B, at k = 4 kk + lane / 16
  auto rd = [&](const double* as, int kk, double (&a4)[4], double (&b4)[NB]) {
    const int krow = 4 * kk + (lane >> 4);
    const double* ap = as + krow * SA + a_off + 4 * (lane & 15);
    const v2f64 a01 = *reinterpret_cast<const v2f64*>(ap), a23 = *reinterpret_cast<const v2f64*>(ap + 2);
    a4[0] = a01[0]; a4[1] = a01[1]; a4[2] = a23[0]; a4[3] = a23[1];
    const double* bp = as + DK * SA + krow * DS + b_off + NB * (lane & 15);
    const v2f64 b01 = *reinterpret_cast<const v2f64*>(bp);
    b4[0] = b01[0]; b4[1] = b01[1];
    if constexpr (NB == 4) {
      const v2f64 b23 = *reinterpret_cast<const v2f64*>(bp + 2);
      b4[2] = b23[0]; b4[3] = b23[1];
    }
  };
  auto mm = [&](const double (&a4)[4], const double (&b4)[NB]) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < NB; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[ti], b4[tj], acc[ti][tj], 0, 0, 0);
  };

  sa.load(A, lda, m0, M, kz0, tid);
  sb.load(B, ldb, n0, N, kz0, tid);
  sa.store(lds, SA, tid);
  sb.store(lds + DK * SA, DS, tid);
  if (nkb > 1) {
    sa.load(A, lda, m0, M, kz0 + DK, tid);
    sb.load(B, ldb, n0, N, kz0 + DK, tid);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  double fa[2][4], fb[2][NB];
  rd(lds, 0, fa[0], fb[0]);
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const double* as = lds + int(kb & 1) * BUF;
    double* nx = lds + int((kb + 1) & 1) * BUF;
    // kk = 0
    rd(as, 1, fa[1], fb[1]);
    mm(fa[0], fb[0]);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // (the next group's fragment reads right behind the first MFMA)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    // kk = 1
    rd(as, 2, fa[0], fb[0]);
    // (branch-free, so that the staging stays inside the MFMA stream: past the end the registers are stored once more into a
    // buffer nobody reads again, and the last block is loaded once more)
    sa.store(nx, SA, tid);
    sb.store(nx + DK * SA, DS, tid);
    {
      const int64_t kl = kz0 + min(kb + 2, nkb - 1) * DK;
      sa.load(A, lda, m0, M, kl, tid);
      sb.load(B, ldb, n0, N, kl, tid);
    }
    mm(fa[1], fb[1]);
    // issue order of this group: the fragment reads, then one ds_write behind each of the first MFMAs, then the global loads behind
    // the following ones (the scheduler would otherwise issue all of the staging in one run in front of the 16 MFMAs)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 4 * NB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < NSTG) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      else if (NB == 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      else __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // kk = 2
    rd(as, 3, fa[1], fb[1]);
    mm(fa[0], fb[0]);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // kk = 3
    rd(nx, 0, fa[0], fb[0]);                 // (after the last block: stale fragments, never used)
    mm(fa[1], fb[1]);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - 1, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // f64 16x16 C/D layout: col (B side) = lane & 15, row (A side) = (lane >> 4) + 4 * reg
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + a_off + 4 * ((lane >> 4) + 4 * r) + ti;
      if (m >= M) continue;
      const int64_t nb = n0 + b_off + NB * (lane & 15);
      double* cp = C + m * ldc + nb;
      if (split) {
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
          if (nb + tj < N) unsafeAtomicAdd(cp + tj, alpha * acc[ti][tj][r]);
      } else if (nb + NB - 1 < N) {
        double v[NB];
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) v[tj] = alpha * acc[ti][tj][r];
        if (beta != 0.0) {
#pragma unroll
          for (int q = 0; q < NB; q += 2) {
            const v2f64 c01 = *reinterpret_cast<const v2f64*>(cp + q);
            v[q] += beta * c01[0];
            v[q + 1] += beta * c01[1];
          }
        }
#pragma unroll
        for (int q = 0; q < NB; q += 2) *reinterpret_cast<v2f64*>(cp + q) = v2f64{v[q], v[q + 1]};
      } else {
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
          if (nb + tj < N) {
            double v = alpha * acc[ti][tj][r];
            if (beta != 0.0) v += beta * cp[tj];
            cp[tj] = v;
          }
      }
    }
}

// Round 3 tried the wave-private LDS-DMA FIFO of K1 on the A B' shape of the Cholesky updates and forward triangular
// solves (lane l of a 16-row MFMA tile loads X[16 t + (l & 15)][k0 + 2 (l >> 4) .. + 1] as one 16-byte
// buffer_load ... lds; no barriers, no transposing scatter, 8 DMA + 8 ds_read_b128 per 32 MFMAs).  Correct, and SLOWER than
// the staged kernels below: rCCA solve 14.4 vs 13.6 ms, GCCA (D = 16384) 201 vs 178 ms.  With the projection kernel's
// result (gemm_big.hip) the lesson is that the FIFO pays when one DMA instruction moves 1 KiB of CONTIGUOUS memory (K1:
// a k-step of X'X is two full row segments); gathering 16 - 64-byte pieces from 16 - 32 different rows per instruction runs
// the texture path at a fraction of that rate, and a cooperative, coalesced stage + LDS transpose wins.  (Measured on the
// GPU and dropped before it was ever committed; DESIGN.md section 4 keeps the numbers.)
static int64_t env_ll(const char* name, int64_t dflt) {
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
}

// Shapes the 128x128-tile kernel takes: products with at least 128 rows and columns and enough tiles.
// Products with few tiles and deep K (subspace-iteration applies S X with 128 <= N < 256) run split-K.
// Narrower outputs (N < 128) stay on the 64x64-tile kernel: measured 23 vs 15 TFLOP/s at N = 80.
bool gemm_f64_big_eligible(bool tA, bool tB, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                           const double* B, int64_t ldb, const double* C, int64_t ldc) {
  (void)tA; (void)tB;
  static const int64_t min_tiles = env_ll("CCZ_GEMM_BIG_MIN_TILES", 32);
  static const int64_t min_k = env_ll("CCZ_GEMM_BIG_MIN_K", 64);
  if (K % DK != 0 || K < min_k) return false;
  if ((lda | ldb | ldc) & 1) return false;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15) return false;
  if ((M + DT - 1) / DT > 65535) return false;
  const int64_t tiles = (M + DT - 1) / DT * ((N + DT - 1) / DT);
  if (M >= 128 && N >= 128 && tiles >= min_tiles) return true;
  return false;
}

__global__ void k_scale2d_f64(int64_t total, int64_t cols, double* __restrict__ A, int64_t lda, double beta) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    A[r * lda + cc] = beta == 0.0 ? 0.0 : beta * A[r * lda + cc];
  }
}

void gemm_f64_big(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                  int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only) {
  const size_t lds_bytes = size_t(2) * 2 * DK * DS * 8;   // 65 KiB
  hipStream_t st = stream(c);
  const int64_t tm = (M + DT - 1) / DT, tn = (N + DT - 1) / DT;
  const int ncu = std::max(1, impl(c)->props.multiProcessorCount);
  int splits = 1;
  // (split-K with an fp64-atomic epilogue only where the tiles leave at least half of the chip idle: at one workgroup per CU the
  // pipelined kernel runs at 0.89 of the peak, and a 256-tile product measured 541 us split four ways against 272 us for rocBLAS)
  static const int64_t pipe_sel = env_ll("CCZ_GEMM64_PIPE", 1);
  if (!lower_only && tm * tn * (pipe_sel ? 2 : 1) < 2 * int64_t(ncu) && K >= 2048) {
    splits = int(std::min<int64_t>({int64_t(16), (4 * ncu) / (tm * tn), K / 512}));
    if (splits < 2) splits = 1;
  }
  static const int64_t half_on = env_ll("CCZ_GEMM_HALF_TILE", 1);
  static const int64_t pipe_on = env_ll("CCZ_GEMM64_PIPE", 1);      // 0: the round-2 kernels (A/B)
  const bool use_half = half_on && splits == 1 && !lower_only && tm * tn < int64_t(ncu) && M > HM;
  if (pipe_on) {
    int64_t kps = K;
    if (splits > 1) {
      kps = ((K + splits - 1) / splits + DK - 1) / DK * DK;
      splits = int((K + kps - 1) / kps);
      const int64_t total = M * N;
      hipLaunchKernelGGL(k_scale2d_f64, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                         total, N, C, ldc, beta);
    }
    const int lo = lower_only ? 1 : 0;
#define CCZ_LAUNCH_PIPE(TA_, TB_, TM_)                                                                                  \
  do {                                                                                                                  \
    const size_t lds_p = size_t(2) * DK * ((TM_) + 2 + DS) * 8;                                                         \
    const dim3 gridp((unsigned)tn, (unsigned)((M + (TM_) - 1) / (TM_)), (unsigned)splits);                              \
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_pipe<TA_, TB_, TM_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_p)));                               \
    hipLaunchKernelGGL((k_gemm_f64_pipe<TA_, TB_, TM_>), gridp, dim3(256), lds_p, st, M, N, K, alpha, A, lda, B, ldb,   \
                       beta, C, ldc, lo, kps);                                                                          \
  } while (0)
#define CCZ_LAUNCH_PIPE_T(TM_)                              \
  do {                                                      \
    if (!tA && !tB) CCZ_LAUNCH_PIPE(false, false, TM_);     \
    else if (tA && !tB) CCZ_LAUNCH_PIPE(true, false, TM_);  \
    else if (!tA && tB) CCZ_LAUNCH_PIPE(false, true, TM_);  \
    else CCZ_LAUNCH_PIPE(true, true, TM_);                  \
  } while (0)
    if (use_half) CCZ_LAUNCH_PIPE_T(64);
    else CCZ_LAUNCH_PIPE_T(128);
#undef CCZ_LAUNCH_PIPE_T
#undef CCZ_LAUNCH_PIPE
    CCZ_LAUNCH_CHECK();
    return;
  }
  if (use_half) {
    const size_t lds_h = size_t(2) * DK * (HSA + DS) * 8;    // 49 KiB
    dim3 gridh((unsigned)tn, (unsigned)((M + HM - 1) / HM), 1);
#define CCZ_LAUNCH_HALF(TA_, TB_)                                                                                     \
  do {                                                                                                                \
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_half<TA_, TB_>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_h)));                             \
    hipLaunchKernelGGL((k_gemm_f64_half<TA_, TB_>), gridh, dim3(256), lds_h, st, M, N, K, alpha, A, lda, B, ldb, beta, \
                       C, ldc);                                                                                       \
  } while (0)
    if (!tA && !tB) CCZ_LAUNCH_HALF(false, false);
    else if (tA && !tB) CCZ_LAUNCH_HALF(true, false);
    else if (!tA && tB) CCZ_LAUNCH_HALF(false, true);
    else CCZ_LAUNCH_HALF(true, true);
#undef CCZ_LAUNCH_HALF
    CCZ_LAUNCH_CHECK();
    return;
  }
  int64_t kps = K;
  if (splits > 1) {
    kps = ((K + splits - 1) / splits + DK - 1) / DK * DK;
    splits = int((K + kps - 1) / kps);
    const int64_t total = M * N;
    hipLaunchKernelGGL(k_scale2d_f64, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                       total, N, C, ldc, beta);
  }
  dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)splits);
  const int lo = lower_only ? 1 : 0;
#define CCZ_LAUNCH_BIG(TA_, TB_)                                                                                  \
  do {                                                                                                            \
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_big<TA_, TB_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));                     \
    hipLaunchKernelGGL((k_gemm_f64_big<TA_, TB_>), grid, dim3(256), lds_bytes, st, M, N, K, alpha, A, lda, B, ldb, \
                       beta, C, ldc, lo, kps);                                                                    \
  } while (0)
  if (!tA && !tB) CCZ_LAUNCH_BIG(false, false);
  else if (tA && !tB) CCZ_LAUNCH_BIG(true, false);
  else if (!tA && tB) CCZ_LAUNCH_BIG(false, true);
  else CCZ_LAUNCH_BIG(true, true);
#undef CCZ_LAUNCH_BIG
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
