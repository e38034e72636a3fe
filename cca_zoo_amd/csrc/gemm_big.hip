// Large fp32 sample-side GEMM of the hot path:  C (M x N) = alpha (A (M x K) B (K x N) - bias) + beta C
// with M = number of samples (up to 1e6+), K, N = feature widths (multiples of 16 / 256).
//
// Used by the DCCA loss backward (dz = (z G11s + z' G12' - 1 bias)/(n-1), deep/objectives.py:61-102 +
// autograd in the reference) where it carries 2x the flops of the Gram pass, and by transform/score
// on wide outputs.  Same MFMA body as K1 (256 x 256 tile, 4 waves x 128 x 128 quadrants,
// v_mfma_f32_32x32x2_f32); the difference is the A operand: samples are ROWS of A, so the k-major
// fragment layout needs a transpose, done while staging through LDS (each thread owns one sample row,
// reads 64 contiguous bytes of it per k-block and scatters them as 16 ds_write_b32 into the [k][m]
// image; consecutive lanes hit consecutive addresses, no bank conflicts).  B (K x N, row-major) is
// already k-major and is staged exactly like a K1 panel.
#include <algorithm>
#include <cstdlib>

#include "hip_common.h"

namespace ccz {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

constexpr int BT = 256;   // tile edge
constexpr int BKK = 16;   // k-block

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(p));
  const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(p >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane(unsigned(bytes));
  void* q = reinterpret_cast<void*>((uint64_t(hi) << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, int(nb), 0x00020000);
}

__global__ __launch_bounds__(256, 1) void k_gemm_f32_nn_big(int64_t M, int64_t N, int64_t K, float alpha,
                                                            const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb, float beta,
                                                            float* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);  // [2 buffers][A^T | B][BKK][256]
  const int64_t ntn = N / BT;
  const int64_t m0 = int64_t(blockIdx.x / ntn) * BT, n0 = int64_t(blockIdx.x % ntn) * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int cg = tid & 63, r4 = tid >> 6;

  // A: rows m0 .. of this tile; rows past M are outside the descriptor and read as 0
  const int64_t rows_valid = min<int64_t>(BT, M - m0);
  const __amdgpu_buffer_rsrc_t srcA = make_rsrc(A + m0 * lda, ((rows_valid - 1) * lda + K) * 4);
  const __amdgpu_buffer_rsrc_t srcB = make_rsrc(B + n0, ((K - 1) * ldb + BT) * 4);
  const int voffA = int(int64_t(tid) * lda * 4);                    // thread t owns sample row m0 + t
  int voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voffB[i] = int(((r4 + 4 * i) * ldb + 4 * cg) * 4);

  v16f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4f32 ra[4], rb[4];
  auto gload = [&](int64_t k0) {
    const int soffA = __builtin_amdgcn_readfirstlane(int(k0 * 4));
    const int soffB = __builtin_amdgcn_readfirstlane(int(k0 * ldb * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcA, voffA + 16 * i, soffA, 0));
      rb[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcB, voffB[i], soffB, 0));
    }
  };
  auto lstore = [&](int buf) {
    float* as = lds + buf * (2 * BKK * BT);
    float* bs = as + BKK * BT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) as[(4 * i + e) * BT + tid] = ra[i][e];          // transpose: [k][m]
      *reinterpret_cast<v4f32*>(bs + (r4 + 4 * i) * BT + 4 * cg) = rb[i];
    }
  };

  const int64_t nkb = K / BKK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) gload((kb + 1) * BKK);
    const float* as = lds + cur * (2 * BKK * BT);
    const float* bs = as + BKK * BT;
    v4f32 af[2], bf[2];
    af[0] = *reinterpret_cast<const v4f32*>(as + (lane >> 5) * BT + wr * 128 + 4 * (lane & 31));
    bf[0] = *reinterpret_cast<const v4f32*>(bs + (lane >> 5) * BT + wc * 128 + 4 * (lane & 31));
#pragma unroll
    for (int kk = 0; kk < BKK / 2; ++kk) {
      if (kk + 1 < BKK / 2) {
        const int krow = 2 * (kk + 1) + (lane >> 5);
        af[(kk + 1) & 1] = *reinterpret_cast<const v4f32*>(as + krow * BT + wr * 128 + 4 * (lane & 31));
        bf[(kk + 1) & 1] = *reinterpret_cast<const v4f32*>(bs + krow * BT + wc * 128 + 4 * (lane & 31));
      }
      __builtin_amdgcn_sched_barrier(0);
      const v4f32 a4 = af[kk & 1], b4 = bf[kk & 1];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[ti], b4[tj], acc[ti][tj], 0, 0, 0);
    }
    if (kb + 1 < nkb) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds, for every (ti, r), four consecutive n (tj = 0..3) of sample row m
  const int64_t nbase = n0 + wc * 128 + 4 * (lane & 31);
  v4f32 b4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) b4 = *reinterpret_cast<const v4f32*>(bias + nbase);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t m = m0 + wr * 128 + 4 * trow + ti;
      if (m >= M) continue;
      float* cp = C + m * ldc + nbase;
      v4f32 v = {acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]};
      v = (v - b4) * alpha;
      if (beta != 0.f) v += beta * *reinterpret_cast<const v4f32*>(cp);
      *reinterpret_cast<v4f32*>(cp) = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// The same product on the wave-private LDS-DMA FIFO of K1 (gram.hip: k_gram_f32_fifo): no barriers, no
// register staging, no LDS transpose.  What makes it possible for a SAMPLE-major A: the k order inside a
// k-block is free, so lane l takes A[m = 32 ti + (l & 31)][k0 + 4 (l >> 5) .. + 3] as ONE 16-byte load and
// uses its four floats as the A operand of k-steps t = 0..3 of tile ti (k-step t then multiplies
// k = k0 + t on lanes 0-31 and k = k0 + 4 + t on lanes 32-63); the B fragment of step t is rows
// k0 + t / k0 + 4 + t of B, four consecutive columns feeding the four column tiles as in K1.
// Per 8-deep k-block a wave issues 4 A + 4 B buffer_load...lds (8 KiB slot, ring of 4), reads back its own
// bytes with ds_read_b128 and orders everything with counted s_waitcnt vmcnt -- the K1 pipeline verbatim.
// Block -> tile mapping keeps the 32 workgroups an XCD runs on two A stripes (shared through its L2).
// ---------------------------------------------------------------------------------------------------
constexpr int GFB = 8;                       // k per FIFO block
constexpr int GFR = 4;                       // ring slots per wave
constexpr int GFSLAB = 4 * 1024;             // bytes of the A (or B) part of a slot
constexpr int GFSLOT = 2 * GFSLAB;

// NTI = row tiles (of 32 samples) per wave: 4 -> 256 x 256 workgroup tile (the big products: every CU has a tile anyway),
// 2 -> 128 x 256 (products with fewer than one 256-tile per CU, e.g. the DCCA gradient at batch 8192: 128 tiles would
// leave half of the SIMDs without a wave; 256 half-height tiles put one wave on every SIMD).
// DMA instructions per 8-deep block: NTI (A) + 4 (B); the counted waits follow from that.
// (inline asm lives in plain __device__ helpers: inside a __global__ TEMPLATE the host pass tries to instantiate the
// body, rejects the AMDGPU asm string and silently drops the kernel's host stub)
__device__ __forceinline__ void wait_vmcnt_16() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt_14() { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt_12() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt_11() { asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt_0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// TWO (the lazy backward of the DCCA loss, loss.hip): the A operand is [A | A2] -- columns [0, K1) from A (ld lda), columns
// [K1, K) from A2 (ld lda2), K1 a multiple of the ring period (32) -- so two views are multiplied where they lie instead of
// through a gathered copy; alpha is multiplied by *alpha_dev (the upstream gradient, a device scalar) and the centring row
// is read as float64 (bias64) and rounded on load.
template <int NTI, bool TWO>
__device__ __forceinline__ void gemm_f32_nn_fifo_body(int64_t M, int64_t N, int64_t K, float alpha,
                                                      const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb, float beta,
                                                      float* __restrict__ C, int64_t ldc,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ C2, int64_t ldc2, int64_t nsplit,
                                                      const float* __restrict__ A2 = nullptr, int64_t lda2 = 0, int64_t K1 = 0,
                                                      const float* __restrict__ alpha_dev = nullptr,
                                                      const double* __restrict__ bias64 = nullptr) {
  // C2 != null: output columns [nsplit, N) go to C2 (columns renumbered from 0), nsplit a multiple of 256
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BTM = 64 * NTI;                         // rows per workgroup tile
  const int64_t tn = N / BT, tm = (M + BTM - 1) / BTM;
  const int64_t q = int64_t(blockIdx.x) >> 3;
  const int64_t mt = (q / tn) * 8 + (blockIdx.x & 7), nt = q % tn;
  if (mt >= tm) return;
  const int64_t m0 = mt * BTM, n0 = nt * BT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* ring = smem + wave * (GFR * GFSLOT);
  const char* rd = ring + lane * 16;

  const int64_t mw = m0 + wr * (32 * NTI);                        // first sample row of this wave's slab
  const int64_t rows_valid = min<int64_t>(32 * NTI, M - mw);      // <= 0: slab entirely past M (nothing stored)
  const int64_t KA = TWO ? K1 : K;                                // columns served by the first source
  const __amdgpu_buffer_rsrc_t srcA1 = make_rsrc(A + (rows_valid > 0 ? mw : 0) * lda,
                                                 rows_valid > 0 ? ((rows_valid - 1) * lda + KA) * 4 : 0);
  const __amdgpu_buffer_rsrc_t srcA2 = TWO ? make_rsrc(A2 + (rows_valid > 0 ? mw : 0) * lda2,
                                                       rows_valid > 0 ? ((rows_valid - 1) * lda2 + (K - K1)) * 4 : 0)
                                           : srcA1;
  const __amdgpu_buffer_rsrc_t srcB = make_rsrc(B + n0 + wc * 128, ((K - 1) * ldb + 128) * 4);
  int voffA[NTI], voffA2[NTI];
#pragma unroll
  for (int ti = 0; ti < NTI; ++ti) {
    voffA[ti] = int(((32 * ti + (lane & 31)) * lda + 4 * (lane >> 5)) * 4);
    voffA2[ti] = TWO ? int(((32 * ti + (lane & 31)) * lda2 + 4 * (lane >> 5)) * 4) : voffA[ti];
  }
  const int swapA = TWO ? __builtin_amdgcn_readfirstlane(int(K1 * 4)) : 0x7fffffff;   // soffA at which the second source takes over
  const int voffB = int(((4 * (lane >> 5)) * ldb + 4 * (lane & 31)) * 4);
  const int rowB = __builtin_amdgcn_readfirstlane(int(ldb * 4));   // bytes per k row of B
  int soffA = 0, soffB = 0;

  v16f32 acc[NTI][4];
#pragma unroll
  for (int i = 0; i < NTI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: blocks 0, 1, 2 -> slots 0, 1, 2
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < NTI) {
        if (TWO && soffA >= swapA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA2, (lds_ptr)(ring + s * GFSLOT + u * 1024), 16, voffA2[u < NTI ? u : 0], soffA - swapA, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA1, (lds_ptr)(ring + s * GFSLOT + u * 1024), 16, voffA[u < NTI ? u : 0], soffA, 0, 0);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + s * GFSLOT + GFSLAB + u * 1024), 16, voffB, soffB, 0, 0);
      soffB += rowB;
    }
    soffA += GFB * 4;
    soffB += 4 * rowB;
  }
  if (NTI == 4) wait_vmcnt_16();    // block 0 landed: two blocks (2 x PER_BLOCK) may stay in flight
  else wait_vmcnt_12();
  v4f32 ab[2][NTI], bf[2];
#pragma unroll
  for (int ti = 0; ti < NTI; ++ti) ab[0][ti] = *reinterpret_cast<const v4f32*>(rd + ti * 1024);
  bf[0] = *reinterpret_cast<const v4f32*>(rd + GFSLAB);

  const int64_t nblk = K / GFB;
  for (int64_t b0 = 0; b0 < nblk; b0 += GFR) {      // one trip = the whole ring period: slots are static
#pragma unroll
    for (int bb = 0; bb < GFR; ++bb) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cur = (bb * 4 + u) & 1, nxt = cur ^ 1;
        const int nslot = (u + 1 < 4) ? bb : (bb + 1) % GFR;
        const int nu = (u + 1 < 4) ? u + 1 : 0;
        const int wsl = (bb + 3) % GFR;               // slot being refilled: block b0 + bb + 3
        const v4f32 b4 = bf[cur];
        // -- gap 0: fragment reads for the next k-step (and, at the end of a block, the next block's A)
        if (u == 3) {
          // block b+1 landed; newer: block b+2 (PER_BLOCK) and three steps of b+3 (min(3, NTI) A + 3 B)
          if (NTI == 4) wait_vmcnt_14();
          else wait_vmcnt_11();
#pragma unroll
          for (int ti = 0; ti < NTI; ++ti)
            ab[(bb + 1) & 1][ti] = *reinterpret_cast<const v4f32*>(rd + nslot * GFSLOT + ti * 1024);
        }
        bf[nxt] = *reinterpret_cast<const v4f32*>(rd + nslot * GFSLOT + GFSLAB + nu * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[0][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[bb & 1][0][u], b4[tj], acc[0][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- gap 1: DMA of A tile u of block b+3
        if (u < NTI) {
          if (TWO && soffA >= swapA)          // wave-uniform: the k-block lies in the second source
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA2, (lds_ptr)(ring + wsl * GFSLOT + u * 1024), 16, voffA2[u < NTI ? u : 0], soffA - swapA, 0, 0);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA1, (lds_ptr)(ring + wsl * GFSLOT + u * 1024), 16, voffA[u < NTI ? u : 0], soffA, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[bb & 1][1][u], b4[tj], acc[1][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- gap 2: DMA of B rows of k-step u of block b+3
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + wsl * GFSLOT + GFSLAB + u * 1024), 16, voffB, soffB, 0, 0);
        soffB += rowB;
        if (u == 3) { soffA += GFB * 4; soffB += 4 * rowB; }
        __builtin_amdgcn_sched_barrier(0);
        if (NTI == 4) {
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            acc[NTI - 2][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[bb & 1][NTI - 2][u], b4[tj], acc[NTI - 2][tj], 0, 0, 0);
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            acc[NTI - 1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[bb & 1][NTI - 1][u], b4[tj], acc[NTI - 1][tj], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  wait_vmcnt_0();

  // epilogue: tile ti owns rows 32 ti + trow (plain), column tiles are strided (n = 4 (lane & 31) + tj)
  const int64_t nbase = n0 + wc * 128 + 4 * (lane & 31);
  v4f32 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) bias4 = *reinterpret_cast<const v4f32*>(bias + nbase);
  if (TWO && bias64) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bias4[e] = float(bias64[nbase + e]);
  }
  if (TWO && alpha_dev) alpha *= *alpha_dev;
  float* Cout = C;
  int64_t ldo = ldc, ncol = nbase;
  if (C2 && n0 >= nsplit) { Cout = C2; ldo = ldc2; ncol = nbase - nsplit; }
#pragma unroll
  for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t m = mw + 32 * ti + trow;
      if (m >= M) continue;
      float* cp = Cout + m * ldo + ncol;
      v4f32 v = {acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]};
      v = (v - bias4) * alpha;
      if (beta != 0.f) v += beta * *reinterpret_cast<const v4f32*>(cp);
      *reinterpret_cast<v4f32*>(cp) = v;
    }
}

// The __global__ template is a bare forwarder: the host pass instantiates a __global__ template's body, and the LDS
// address-space casts / AMDGPU asm of the pipeline above make that instantiation fail silently (no host stub).
template <int NTI>
__global__ __launch_bounds__(256, 1) void k_gemm_f32_nn_fifo(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                                             int64_t lda, const float* __restrict__ B, int64_t ldb, float beta,
                                                             float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                             float* __restrict__ C2, int64_t ldc2, int64_t nsplit) {
  gemm_f32_nn_fifo_body<NTI, false>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, C2, ldc2, nsplit);
}

// [A | A2] form (see the body): the lazy backward of the two-view DCCA loss
template <int NTI>
__global__ __launch_bounds__(256, 1) void k_gemm_f32_nn_fifo2(int64_t M, int64_t N, int64_t K, float alpha, const float* __restrict__ A,
                                                              int64_t lda, const float* __restrict__ A2, int64_t lda2, int64_t K1,
                                                              const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                              float* __restrict__ C2, int64_t ldc2, int64_t nsplit,
                                                              const float* __restrict__ alpha_dev, const double* __restrict__ bias64) {
  gemm_f32_nn_fifo_body<NTI, true>(M, N, K, alpha, A, lda, B, ldb, 0.0f, C, ldc, static_cast<const float*>(nullptr), C2, ldc2, nsplit, A2, lda2,
                                   K1, alpha_dev, bias64);
}

// ---------------------------------------------------------------------------------------------------
// Projection shape: N <= 64 outputs per sample (transform / score: (X - mean) W with k <= 64 directions).
// 2 n d k flop against n d 4 bytes: at k = 64 the MFMA time (3.3 ms per 1e6 x 4096 view) and the HBM time
// (2.7 ms at 6 TB/s) are about equal, so the kernel has to keep both busy.  Measured 5.3 ms (95 TF, 3.0 TB/s) for any
// k <= 64 and any d.  Round 3 tried the wave-private LDS-DMA FIFO pipeline of K1 on this shape (128 rows x 64 columns
// per wave, W re-packed so that a lane's B operands are two 16-byte pieces, 6 DMA + 6 ds_read_b128 per 32 MFMAs): 5.5 ms,
// i.e. no gain, at d = 512 as well as at d = 4096 -- so neither the staging nor the row stride is what holds this shape at
// 60 % of the MFMA rate (two column tiles per A fragment instead of K1's four: every MFMA pair needs a fresh A
// operand).  The attempt is in the history (commit "FIFO projection kernel for k <= 64"), not in the tree.  256 rows x 64 columns per
// workgroup, 4 waves x (64 x 64) = 2 x 2 MFMA tiles each, the same LDS-transposed A staging as above,
// B (converted once to fp32 and zero-padded to 64 columns) staged as [k][64]; 40 KiB of LDS -> three
// workgroups per CU hide each other's staging.  Fragments are 8-byte reads (two consecutive m / n feed the
// two tiles of a dimension: the stride-2 version of the ownership trick).
// ---------------------------------------------------------------------------------------------------
typedef float v2f32 __attribute__((ext_vector_type(2)));
constexpr int TN = 64;

__global__ __launch_bounds__(256, 3) void k_gemm_f32_nn_tall(int64_t M, int64_t N, int64_t K, float alpha,
                                                             const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B /* K x 64, padded */, float beta,
                                                             float* __restrict__ C, int64_t ldc,
                                                             const float* __restrict__ bias /* 64, padded */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);   // [2 buffers][A^T 16 x 256 | B 16 x 64]
  constexpr int STG = BKK * (BT + TN);
  const int64_t m0 = int64_t(blockIdx.x) * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t rows_valid = min<int64_t>(BT, M - m0);
  const __amdgpu_buffer_rsrc_t srcA = make_rsrc(A + m0 * lda, ((rows_valid - 1) * lda + K) * 4);
  const __amdgpu_buffer_rsrc_t srcB = make_rsrc(B, K * TN * 4);
  const int voffA = int(int64_t(tid) * lda * 4);                       // thread t owns sample row m0 + t
  const int voffB = ((tid >> 4) * TN + 4 * (tid & 15)) * 4;            // 16 rows x 16 float4 per k-block

  v16f32 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4f32 ra[4], rb;
  auto gload = [&](int64_t k0) {
    const int soffA = __builtin_amdgcn_readfirstlane(int(k0 * 4));
    const int soffB = __builtin_amdgcn_readfirstlane(int(k0 * TN * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcA, voffA + 16 * i, soffA, 0));
    rb = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcB, voffB, soffB, 0));
  };
  auto lstore = [&](int buf) {
    float* as = lds + buf * STG;
    float* bs = as + BKK * BT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) as[(4 * i + e) * BT + tid] = ra[i][e];          // transpose: [k][m]
    *reinterpret_cast<v4f32*>(bs + (tid >> 4) * TN + 4 * (tid & 15)) = rb;
  };

  const int64_t nkb = K / BKK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) gload((kb + 1) * BKK);
    const float* as = lds + cur * STG;
    const float* bs = as + BKK * BT;
    v2f32 af[2], bf[2];
    af[0] = *reinterpret_cast<const v2f32*>(as + (lane >> 5) * BT + wave * 64 + 2 * (lane & 31));
    bf[0] = *reinterpret_cast<const v2f32*>(bs + (lane >> 5) * TN + 2 * (lane & 31));
#pragma unroll
    for (int kk = 0; kk < BKK / 2; ++kk) {
      if (kk + 1 < BKK / 2) {
        const int krow = 2 * (kk + 1) + (lane >> 5);
        af[(kk + 1) & 1] = *reinterpret_cast<const v2f32*>(as + krow * BT + wave * 64 + 2 * (lane & 31));
        bf[(kk + 1) & 1] = *reinterpret_cast<const v2f32*>(bs + krow * TN + 2 * (lane & 31));
      }
      __builtin_amdgcn_sched_barrier(0);
      const v2f32 a2 = af[kk & 1], b2 = bf[kk & 1];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ti], b2[tj], acc[ti][tj], 0, 0, 0);
    }
    if (kb + 1 < nkb) lstore(cur ^ 1);
    __syncthreads();
  }

  // lane holds, for every (ti, r), two consecutive n (tj = 0, 1) of sample row m
  const int nb = 2 * (lane & 31);
  v2f32 b2 = {0.f, 0.f};
  if (bias) b2 = *reinterpret_cast<const v2f32*>(bias + nb);
  const bool pair_ok = (ldc & 1) == 0 && nb + 1 < N;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t m = m0 + wave * 64 + 2 * trow + ti;
      if (m >= M || nb >= N) continue;
      float* cp = C + m * ldc + nb;
      v2f32 v = {acc[ti][0][r], acc[ti][1][r]};
      v = (v - b2) * alpha;
      if (pair_ok) {
        if (beta != 0.f) v += beta * *reinterpret_cast<const v2f32*>(cp);
        *reinterpret_cast<v2f32*>(cp) = v;
      } else {
        cp[0] = beta != 0.f ? v[0] + beta * cp[0] : v[0];
        if (nb + 1 < N) cp[1] = beta != 0.f ? v[1] + beta * cp[1] : v[1];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 6: the same projection with WHOLE-LINE loads.  k_gemm_f32_nn_tall above gives every lane
// its own sample row and fetches 64 bytes of it per k-block: each wave-level b128 load touches 64 different 128-byte
// lines and uses 16 bytes of each, and the second half of a line is asked for one k-block later, when 12 waves x 64 rows
// x 128 B = 96 KB per CU have gone through a 32 KB L1 in between.  Here eight consecutive lanes fetch the 128 bytes
// (32 k) of ONE row -- a wave-level load is eight whole lines -- and the transpose into the [k][m] image happens in the
// LDS store (row stride ROWS + 2 floats: the 64 lanes of a store land on every bank twice, the minimum for 64 x 4 bytes).
// k-block 32; ONE LDS buffer (41 KB at 256 rows, 25 KB at 128) with the next block held in registers while the current one
// is multiplied.  Same fragment reads, MFMA order and epilogue as above.  Measured (profiles/r06_transform_pmc.md, per
// 1e6 x 4096 view): fabric reads 1.55 x -> 0.99 x the view; k = 64: 5.18 -> 4.70 ms (fp32 pipe 90 % busy at 1.9 GHz);
// k <= 32: 4.9 -> 3.0 ms = 5.4 - 5.6 TB/s.
// ---------------------------------------------------------------------------------------------------
constexpr int BK2 = 32;

// TI = 2: 256 rows per workgroup (a wave owns 64 rows = 2 x 2 MFMA tiles); TI = 1: 128 rows (a wave owns 32 rows = 1 x 2 tiles):
// half the work per workgroup and twice the workgroups per CU -- the last, partly filled round of workgroups (3907 on 768
// slots at n = 1e6) costs half as much, and more workgroups in different phases keep the matrix pipe fed across the barriers.
// NJ = 2: 64 output columns (two 32-wide MFMA tiles, a lane's two fragments are consecutive n); NJ = 1: N <= 32 -- one tile, half the
// MFMA work (k <= 32 directions are the common case of the linear models), only the first 32 columns of B staged.
template <int TI, int NJ>
__global__ __launch_bounds__(256, TI == 2 ? 3 : 5) void k_gemm_f32_nn_tall2(int64_t M, int64_t N, int64_t K, float alpha,
                                                                            const float* __restrict__ A, int64_t lda,
                                                                            const float* __restrict__ B /* K x 64, padded */,
                                                                            float beta, float* __restrict__ C, int64_t ldc,
                                                                            const float* __restrict__ bias /* 64, padded */) {
  constexpr int ROWS = 128 * TI, RW = 32 * TI, RP = ROWS + 2, NL = 4 * TI, BW = 32 * NJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* as = reinterpret_cast<float*>(smem);    // A^T [32][ROWS + 2]
  float* bs = as + BK2 * RP;                     // B   [32][BW]
  const int64_t m0 = int64_t(blockIdx.x) * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t rows_valid = min<int64_t>(ROWS, M - m0);
  const __amdgpu_buffer_rsrc_t srcA = make_rsrc(A + m0 * lda, ((rows_valid - 1) * lda + K) * 4);
  const __amdgpu_buffer_rsrc_t srcB = make_rsrc(B, K * TN * 4);
  const int rl = wave * RW + (lane >> 3), pc = lane & 7;               // load i: row rl + 8 i, 16-byte piece pc of the 128
  const int voffA = int((int64_t(rl) * lda + 4 * pc) * 4);
  const int rstep = int(int64_t(8) * lda * 4);
  // B: 32 x BW floats per k-block = 256 NJ float4, NJ per thread (NJ = 2: the padded rows as they lie; NJ = 1: their first half)
  const int voffB = NJ == 2 ? tid * 16 : (tid >> 3) * (TN * 4) + (tid & 7) * 16;

  v16f32 acc[TI][NJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4f32 ra[NL], rb[NJ];
  auto gload = [&](int64_t k0) {
    const int soffA = __builtin_amdgcn_readfirstlane(int(k0 * 4));
    const int soffB = __builtin_amdgcn_readfirstlane(int(k0 * TN * 4));
#pragma unroll
    for (int i = 0; i < NL; ++i)
      ra[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcA, voffA + i * rstep, soffA, 0));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      rb[j] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcB, voffB + j * 4096, soffB, 0));
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) as[(4 * pc + e) * RP + rl + 8 * i] = ra[i][e];   // transpose: [k][m]
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<v4f32*>(bs + (tid + 256 * j) * 4) = rb[j];
  };

  const int64_t nkb = K / BK2;
  gload(0);
  lstore();
  __syncthreads();
  const float* ap = as + (lane >> 5) * RP + wave * RW + TI * (lane & 31);
  const float* bp = bs + (lane >> 5) * BW + NJ * (lane & 31);
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const bool more = kb + 1 < nkb;
    if (more) gload((kb + 1) * BK2);
    float af[2][TI], bf[2][NJ];
    auto fread = [&](int slot, int kk) {
      if constexpr (TI == 2) {
        const v2f32 t = *reinterpret_cast<const v2f32*>(ap + 2 * kk * RP);
        af[slot][0] = t[0];
        af[slot][1] = t[1];
      } else {
        af[slot][0] = ap[2 * kk * RP];
      }
      if constexpr (NJ == 2) {
        const v2f32 t = *reinterpret_cast<const v2f32*>(bp + 2 * kk * BW);
        bf[slot][0] = t[0];
        bf[slot][1] = t[1];
      } else {
        bf[slot][0] = bp[2 * kk * BW];
      }
    };
    fread(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK2 / 2; ++kk) {
      if (kk + 1 < BK2 / 2) fread((kk + 1) & 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < NJ; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][ti], bf[kk & 1][tj], acc[ti][tj], 0, 0, 0);
    }
    __syncthreads();                       // every wave has read the block
    if (more) {
      lstore();
      __syncthreads();
    }
  }

  // NJ = 2: a lane holds, for every (ti, r), two consecutive n (tj = 0, 1) of sample row m; NJ = 1: one n
  const int nb = NJ * (lane & 31);
  float bv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bv[j] = bias ? bias[nb + j] : 0.f;
  const bool pair_ok = NJ == 2 && (ldc & 1) == 0 && nb + 1 < N;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t m = m0 + wave * RW + TI * trow + ti;
      if (m >= M || nb >= N) continue;
      float* cp = C + m * ldc + nb;
      if constexpr (NJ == 2) {
        v2f32 v = {acc[ti][0][r], acc[ti][1][r]};
        const v2f32 b2 = {bv[0], bv[1]};
        v = (v - b2) * alpha;
        if (pair_ok) {
          if (beta != 0.f) v += beta * *reinterpret_cast<const v2f32*>(cp);
          *reinterpret_cast<v2f32*>(cp) = v;
        } else {
          cp[0] = beta != 0.f ? v[0] + beta * cp[0] : v[0];
          if (nb + 1 < N) cp[1] = beta != 0.f ? v[1] + beta * cp[1] : v[1];
        }
      } else {
        const float v = (acc[ti][0][r] - bv[0]) * alpha;
        cp[0] = beta != 0.f ? v + beta * cp[0] : v;
      }
    }
}

// fp64 (rows x cols, ld ldi) -> fp32 (rows x cols_pad, zero-padded columns)
__global__ void k_f64_to_f32_pad(int64_t rows, int64_t cols, int64_t cols_pad, const double* __restrict__ in, int64_t ldi,
                                 float* __restrict__ out) {
  const int64_t total = rows * cols_pad;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols_pad, cc = i - r * cols_pad;
    out[i] = cc < cols ? float(in[r * ldi + cc]) : 0.f;
  }
}

__global__ void k_f64_to_f32(int64_t total, int64_t cols, const double* __restrict__ in, int64_t ldi,
                             float* __restrict__ out, int64_t ldo) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    out[r * ldo + cc] = float(in[r * ldi + cc]);
  }
}

// true if the big kernel can take this problem (else the caller uses the generic tiled GEMM)
static bool tall_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, const void* A, const void* C) {
  if (N < 1 || N > TN || K % BKK != 0 || K < 256 || M < 8192) return false;
  if (lda % 4 != 0 || reinterpret_cast<uintptr_t>(A) % 16 != 0 || reinterpret_cast<uintptr_t>(C) % 8 != 0) return false;
  if (int64_t(BT) * lda * 4 >= (int64_t(1) << 31) || K * TN * 4 >= (int64_t(1) << 31)) return false;
  if ((M + BT - 1) / BT >= (int64_t(1) << 31)) return false;
  return true;
}

bool gemm_f32_big_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* A,
                           const void* C) {
  if (tall_eligible(M, N, K, lda, A, C)) return true;
  if (N % BT != 0 || K % BKK != 0 || K < BKK) return false;
  if ((M + BT - 1) / BT * (N / BT) < 256) return false;   // fewer tiles than CUs: the 64x64-tile kernel spreads better
  if (lda % 4 != 0 || ldc % 4 != 0) return false;
  if (reinterpret_cast<uintptr_t>(A) % 16 != 0 || reinterpret_cast<uintptr_t>(C) % 16 != 0) return false;
  if (int64_t(BT) * lda * 4 >= (int64_t(1) << 31) || K * N * 4 >= (int64_t(1) << 31)) return false;
  if ((M + BT - 1) / BT * (N / BT) >= (int64_t(1) << 31)) return false;
  return true;
}

// C (M x N float) = alpha (A (M x K float) B (K x N float64, converted once) - bias (N float64)) + beta C
void gemm_f32_big(ccz_ctx* c, int64_t M, int64_t N, int64_t K, double alpha, const float* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, float* C, int64_t ldc, const double* bias_row) {
  hipStream_t st = stream(c);
  if (tall_eligible(M, N, K, lda, A, C)) {
    float* B32 = static_cast<float*>(dev_alloc(c, size_t(K + 1) * TN * 4));
    float* bias32 = bias_row ? B32 + K * TN : nullptr;
    hipLaunchKernelGGL(k_f64_to_f32_pad, dim3((unsigned)std::min<int64_t>((K * TN + 255) / 256, 1 << 20)), dim3(256), 0, st, K,
                       N, int64_t(TN), B, ldb, B32);
    if (bias_row)
      hipLaunchKernelGGL(k_f64_to_f32_pad, dim3(1), dim3(256), 0, st, int64_t(1), N, int64_t(TN), bias_row, N, bias32);
    const char* tall_env = getenv("CCZ_TALL_IMPL");     // 3: whole-line loads, 128 rows per workgroup (default); 2: 256 rows; 1: a row per lane
    const int tall_impl = tall_env ? atoi(tall_env) : 3;
    const char* nj_env = getenv("CCZ_TALL_NJ1");         // 0: always two column tiles (A/B switch of the N <= 32 form)
    const bool nj1 = N <= 32 && !(nj_env && atoi(nj_env) == 0);
    if (tall_impl == 3 && K % BK2 == 0 && (M + 127) / 128 < (int64_t(1) << 31)) {
      const size_t lds_bytes = size_t(BK2) * (128 + 2 + (nj1 ? 32 : TN)) * 4;
      auto kern = nj1 ? &k_gemm_f32_nn_tall2<1, 1> : &k_gemm_f32_nn_tall2<1, 2>;
      hipLaunchKernelGGL(kern, dim3((unsigned)((M + 127) / 128)), dim3(256), lds_bytes, st, M, N, K, float(alpha), A, lda, B32, float(beta),
                         C, ldc, bias32);
    } else if (tall_impl >= 2 && K % BK2 == 0) {
      const size_t lds_bytes = size_t(BK2) * (256 + 2 + TN) * 4;
      hipLaunchKernelGGL((k_gemm_f32_nn_tall2<2, 2>), dim3((unsigned)((M + BT - 1) / BT)), dim3(256), lds_bytes, st, M, N, K, float(alpha),
                         A, lda, B32, float(beta), C, ldc, bias32);
    } else {
      const size_t lds_bytes = size_t(2) * BKK * (BT + TN) * 4;
      hipLaunchKernelGGL(k_gemm_f32_nn_tall, dim3((unsigned)((M + BT - 1) / BT)), dim3(256), lds_bytes, st, M, N, K, float(alpha),
                         A, lda, B32, float(beta), C, ldc, bias32);
    }
    CCZ_LAUNCH_CHECK();
    dev_free(c, B32);                    // pooled scratch is recycled in stream order: no host wait
    return;
  }
  float* B32 = static_cast<float*>(dev_alloc(c, size_t(K) * N * 4 + (bias_row ? size_t(N) * 4 : 0)));
  float* bias32 = bias_row ? B32 + K * N : nullptr;
  {
    const int64_t total = K * N;
    hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                       total, N, B, ldb, B32, N);
    if (bias_row)
      hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, N, bias_row, N, bias32, N);
  }
  static const int nn_impl = [] { const char* e = getenv("CCZ_GEMM_NN_IMPL"); return e ? atoi(e) : 1; }();   // 1: LDS-DMA FIFO, 0: staged tile
  const int64_t tmb = (M + BT - 1) / BT, tnb = N / BT;
  const bool fifo_ok = nn_impl != 0 && K % 32 == 0 && int64_t(128) * lda * 4 < (int64_t(1) << 31) &&
                       (tmb + 7) / 8 * 8 * tnb < (int64_t(1) << 31);
  if (fifo_ok) {
    const size_t fifo_bytes = size_t(4) * GFR * GFSLOT;   // 128 KiB: four wave-private rings
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_fifo<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gemm_f32_nn_fifo<4>, dim3((unsigned)((tmb + 7) / 8 * 8 * tnb)), dim3(256), fifo_bytes, st, M, N, K,
                       float(alpha), A, lda, B32, N, float(beta), C, ldc, bias32, static_cast<float*>(nullptr), int64_t(0), N);
  } else {
    const size_t lds_bytes = size_t(2) * 2 * BKK * BT * 4;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_big), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
    const int64_t nblocks = tmb * tnb;
    hipLaunchKernelGGL(k_gemm_f32_nn_big, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, M, N, K, float(alpha), A, lda, B32,
                       N, float(beta), C, ldc, bias32);
  }
  CCZ_LAUNCH_CHECK();
  dev_free(c, B32);                      // pooled scratch is recycled in stream order: no host wait
}

// ---------------------------------------------------------------------------------------------------
// FIFO product with prepared fp32 operands and a column-split destination (DCCA loss: [dz1 | dz2] = (Z - mean) Gamma
// in ONE launch, the two column ranges landing in the two gradient tensors).  No conversion, no allocation, no
// synchronisation: everything the caller hands in stays alive until its own synchronisation point.
// ---------------------------------------------------------------------------------------------------
bool gemm_f32_fifo_split_eligible(int64_t M, int64_t N, int64_t K, int64_t nsplit, const void* C1, int64_t ldc1, const void* C2,
                                  int64_t ldc2) {
  if (N % BT != 0 || nsplit % BT != 0 || nsplit <= 0 || nsplit >= N || K % 32 != 0 || K < 32 || M < 1) return false;
  if (ldc1 % 4 != 0 || ldc2 % 4 != 0) return false;
  if (reinterpret_cast<uintptr_t>(C1) % 16 != 0 || reinterpret_cast<uintptr_t>(C2) % 16 != 0) return false;
  if (int64_t(128) * K * 4 >= (int64_t(1) << 31) || K * N * 4 >= (int64_t(1) << 31)) return false;
  const int64_t tmb = (M + BT - 1) / BT, tnb = N / BT;
  return (tmb + 7) / 8 * 8 * tnb < (int64_t(1) << 31);
}

void gemm_f32_fifo_split(ccz_ctx* c, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B32,
                         const float* bias32, float* C1, int64_t ldc1, float* C2, int64_t ldc2, int64_t nsplit) {
  const int64_t tnb = N / BT;
  const size_t fifo_bytes = size_t(4) * GFR * GFSLOT;
  const int ncu = std::max(1, impl(c)->props.multiProcessorCount);
  // fewer full-height tiles than CUs: half-height tiles put a wave on every SIMD
  const bool half = (M + BT - 1) / BT * tnb < int64_t(ncu);
  if (half) {
    const int64_t tmb = (M + 127) / 128;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_fifo<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gemm_f32_nn_fifo<2>, dim3((unsigned)((tmb + 7) / 8 * 8 * tnb)), dim3(256), fifo_bytes, stream(c), M, N, K, alpha, A,
                       lda, B32, N, 0.0f, C1, ldc1, bias32, C2, ldc2, nsplit);
  } else {
    const int64_t tmb = (M + BT - 1) / BT;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_fifo<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gemm_f32_nn_fifo<4>, dim3((unsigned)((tmb + 7) / 8 * 8 * tnb)), dim3(256), fifo_bytes, stream(c), M, N, K, alpha, A,
                       lda, B32, N, 0.0f, C1, ldc1, bias32, C2, ldc2, nsplit);
  }
  CCZ_LAUNCH_CHECK();
}

// [C1 | C2] = alpha (*alpha_dev) ([A1 | A2] B32 - bias64): the two-view DCCA gradient with the views where they lie.
// A1: M x K1 (ld lda1), A2: M x (K - K1) (ld lda2), B32: K x N fp32 (ld N), C1: columns [0, nsplit), C2: the rest.
bool gemm_f32_fifo_pair_eligible(int64_t M, int64_t N, int64_t K, int64_t K1, int64_t nsplit, const void* A1, int64_t lda1, const void* A2,
                                 int64_t lda2, const void* C1, int64_t ldc1, const void* C2, int64_t ldc2) {
  if (!gemm_f32_fifo_split_eligible(M, N, K, nsplit, C1, ldc1, C2, ldc2)) return false;
  if (K1 <= 0 || K1 >= K || K1 % 32 != 0 || (K - K1) % 32 != 0) return false;
  if (lda1 % 4 != 0 || lda2 % 4 != 0 || lda1 < K1 || lda2 < K - K1) return false;
  if (reinterpret_cast<uintptr_t>(A1) % 16 != 0 || reinterpret_cast<uintptr_t>(A2) % 16 != 0) return false;
  if (int64_t(128) * lda1 * 4 >= (int64_t(1) << 31) || int64_t(128) * lda2 * 4 >= (int64_t(1) << 31)) return false;
  return true;
}

void gemm_f32_fifo_pair(ccz_ctx* c, int64_t M, int64_t N, int64_t K, int64_t K1, float alpha, const float* alpha_dev, const float* A1,
                        int64_t lda1, const float* A2, int64_t lda2, const float* B32, const double* bias64, float* C1, int64_t ldc1,
                        float* C2, int64_t ldc2, int64_t nsplit) {
  const int64_t tnb = N / BT;
  const size_t fifo_bytes = size_t(4) * GFR * GFSLOT;
  const int ncu = std::max(1, impl(c)->props.multiProcessorCount);
  const bool half = (M + BT - 1) / BT * tnb < int64_t(ncu);
  if (half) {
    const int64_t tmb = (M + 127) / 128;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_fifo2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gemm_f32_nn_fifo2<2>, dim3((unsigned)((tmb + 7) / 8 * 8 * tnb)), dim3(256), fifo_bytes, stream(c), M, N, K, alpha, A1, lda1,
                       A2, lda2, K1, B32, N, C1, ldc1, C2, ldc2, nsplit, alpha_dev, bias64);
  } else {
    const int64_t tmb = (M + BT - 1) / BT;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_fifo2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gemm_f32_nn_fifo2<4>, dim3((unsigned)((tmb + 7) / 8 * 8 * tnb)), dim3(256), fifo_bytes, stream(c), M, N, K, alpha, A1, lda1,
                       A2, lda2, K1, B32, N, C1, ldc1, C2, ldc2, nsplit, alpha_dev, bias64);
  }
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
