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

__global__ void k_f64_to_f32(int64_t total, int64_t cols, const double* __restrict__ in, int64_t ldi,
                             float* __restrict__ out, int64_t ldo) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    out[r * ldo + cc] = float(in[r * ldi + cc]);
  }
}

// true if the big kernel can take this problem (else the caller uses the generic tiled GEMM)
bool gemm_f32_big_eligible(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* A,
                           const void* C) {
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
  float* B32 = static_cast<float*>(dev_alloc(c, size_t(K) * N * 4 + (bias_row ? size_t(N) * 4 : 0)));
  float* bias32 = bias_row ? B32 + K * N : nullptr;
  {
    const int64_t total = K * N;
    hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                       total, N, B, ldb, B32, N);
    if (bias_row)
      hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, N, bias_row, N, bias32, N);
  }
  const size_t lds_bytes = size_t(2) * 2 * BKK * BT * 4;
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_nn_big), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
  const int64_t nblocks = (M + BT - 1) / BT * (N / BT);
  hipLaunchKernelGGL(k_gemm_f32_nn_big, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, M, N, K, float(alpha), A, lda, B32,
                     N, float(beta), C, ldc, bias32);
  CCZ_LAUNCH_CHECK();
  CCZ_HIP(hipStreamSynchronize(st));   // B32 is pooled scratch
  dev_free(c, B32);
}

}  // namespace ccz
