// Tall-times-skinny fp64 GEMM of the subspace iteration:  C (M x N) = alpha op(A) (M x K) B (K x N) + beta C
// with 48 < N <= 192 (the CheFSI block width b = k + max(8, k/4)) and M, K in the thousands: every
// Chebyshev step of the top-k eigensolver (solve.cpp: topk_symmetric) is one such product with the
// whitened d x d operator, so it carries ~70% of the MCCA / GCCA solve time at D >= 8192.
//
// The 128 x 128-tile kernel pads N to 128 / 256 (37% of its MFMAs idle at N = 160) and the 64 x 64-tile
// kernel re-reads A once per column tile.  Here one workgroup (4 waves) owns a 64-row stripe and ALL N
// columns: each wave 16 rows x NT (= ceil(N/16)) MFMA tiles of v_mfma_f64_16x16x4_f64, so A is read exactly
// once from HBM and B (K x N, a few MB) is served from L2.  Two workgroups per CU; split-K slices (cost model
// rounds / splits) fill the chip and accumulate with fp64 atomics.
//   * loads: unconditional buffer loads (masked lanes carry an out-of-range offset), TWO k-blocks of register
//     prefetch in flight so that a block has two compute phases to arrive, exact s_waitcnt vmcnt(N);
//   * LDS images chosen for conflict-free 8-byte fragment reads (k-major: row stride == 16 mod 32 doubles;
//     m-major: [m][18]), fragments of k-step kk+1 fetched under the MFMAs of step kk.
// Measured 52 TFLOP/s at 16384^2 x 160 (a pure-MFMA loop sustains 72-75 on this part; tools/mfma_probe.hip).
// Round 6 built the loop of k_gemm_f64_pipe (gemm64_big.hip) for this shape too -- 128-row stripes, two MFMA row tiles per wave
// sharing every B fragment, register stage -> LDS and the buffer loads interleaved into the MFMA stream, one barrier per k-block --
// and measured it against this kernel (tools/gemm64_probe.py skinny): 4096^2 x 80 67.9 vs 69.9 us, 8192^2 x 80 227 vs 225 us,
// 16384^2 x 160 1556 vs 1558 us (1355 vs 1327 transposed), and SLOWER on ragged or short products (5000 x 80 x 4096: 98 vs 74 us --
// half as many stripes to spread).  The barrier bubble is not what holds this kernel: it was dropped again (not in the tree).
#include <algorithm>
#include <cstdlib>

#include "hip_common.h"

namespace ccz {

typedef double v4f64s __attribute__((ext_vector_type(4)));
typedef double v2f64s __attribute__((ext_vector_type(2)));

constexpr int SK = 16;         // k-block
// LDS images.  A fragment read is 16 consecutive m (128 B) x 4 k per wave; a b64 read is served per half
// wave, so the two k rows of a half wave must sit 32 banks apart:
//   k-major image [k][m] (TA):  row stride == 16 (mod 32) doubles;
//   m-major image [m][k] (!TA, rows of A copied as loaded, no transposing scatter):  row stride 18 doubles
//   (18 m mod 32 is a permutation of the even residues for m = 0..15, k in {2j, 2j+1} fills the odd ones).
constexpr int SAS_M = SK + 2;    // [rows][18]
constexpr int SKW = 4;           // waves per workgroup: 64-row stripes, two independent workgroups per CU

typedef unsigned int v4u32s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const void* base, int64_t bytes) {
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(p));
  const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(p >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane(unsigned(bytes));
  void* q = reinterpret_cast<void*>((uint64_t(hi) << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, int(nb), 0x00020000);
}

template <bool TA, int NT>
__global__ __launch_bounds__(64 * SKW, 2) void k_gemm_f64_skinny(int64_t M, int64_t N, int64_t K, double alpha,
                                                            const double* __restrict__ A, int64_t lda,
                                                            const double* __restrict__ B, int64_t ldb, double beta,
                                                            double* __restrict__ C, int64_t ldc, int64_t k_per_split) {
  constexpr int NP = NT * 16;          // padded width
  constexpr int SBS = (NP % 32 == 16) ? NP : NP + 16;   // B image row stride == 16 (mod 32) doubles
  constexpr int BCH = NT * 8;          // 16-byte chunks per B row
  constexpr int SM = 16 * SKW, NTHR = 64 * SKW;
  constexpr int SAS_K = SM + 16;       // k-major A image [16][SM + 16]
  constexpr int SA_ELEMS = SK * SAS_K > SM * SAS_M ? SK * SAS_K : SM * SAS_M;
  constexpr int AIT = (SM * 8 + NTHR - 1) / NTHR;      // 16-byte chunks of the A block per thread
  constexpr int BIT = (16 * BCH + NTHR - 1) / NTHR;
  constexpr int STAGE = SA_ELEMS + SK * SBS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);  // [2][A image | B image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = int64_t(blockIdx.x) * SM;
  const bool split = gridDim.y > 1;
  const int64_t kz0 = int64_t(blockIdx.y) * k_per_split;
  const int64_t nkb = (min(K, kz0 + k_per_split) - kz0) / SK;
  if (nkb <= 0) return;

  v4f64s acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4f64s{0.0, 0.0, 0.0, 0.0};

  // register stage of one k-block (16 x 128 of A, 16 x N of B): two of them are in flight, so a block has two
  // compute phases (~2 x 2 us) to arrive -- one phase does not cover the loaded HBM/L2 latency
  // All loads are unconditional buffer loads (masked lanes carry an out-of-range voffset and read 0): with
  // no exec-masked branches the compiler can count outstanding loads exactly (s_waitcnt vmcnt(N)).
  struct Stage { v2f64s ra[AIT], rb[BIT]; };
  const int64_t kspan = min(K, kz0 + k_per_split) - kz0;
  const __amdgpu_buffer_rsrc_t rsA =
      TA ? sk_rsrc(A + kz0 * lda + m0, ((kspan - 1) * lda + (M - m0)) * 8)
         : sk_rsrc(A + m0 * lda + kz0, ((min<int64_t>(SM, M - m0) - 1) * lda + kspan) * 8);
  const __amdgpu_buffer_rsrc_t rsB = sk_rsrc(B + kz0 * ldb, ((kspan - 1) * ldb + N) * 8);
  constexpr unsigned OOR = 0xFFFFFFF0u;
  unsigned voA[AIT], voB[BIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int ch = tid + NTHR * i;
    if (TA) {   // A stored K x M: row ch / (SM/2) of the k-block, columns m0 + 2 (ch % (SM/2)) (M is even)
      constexpr int RC = SM / 2;   // chunks per k row
      voA[i] = m0 + 2 * (ch % RC) < M ? unsigned(((ch / RC) * lda + 2 * (ch % RC)) * 8) : OOR;
    } else {    // A stored M x K: row m0 + (ch >> 3) (clamped: rows >= M are never stored), 16 bytes at 2 (ch & 7)
      const int64_t r = std::min<int64_t>(ch >> 3, M - 1 - m0);
      voA[i] = unsigned((r * lda + 2 * (ch & 7)) * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    const int ch = tid + NTHR * i;
    const int kr = ch / BCH, nq = ch - kr * BCH;
    voB[i] = (ch < 16 * BCH && 2 * nq < N) ? unsigned((kr * ldb + 2 * nq) * 8) : OOR;   // N is even
  }
  const int64_t stepA = TA ? SK * lda * 8 : SK * 8, stepB = SK * ldb * 8;
  auto gload = [&](Stage& sg, int64_t kb) {
    const int soA = __builtin_amdgcn_readfirstlane(int(unsigned(kb * stepA)));
    const int soB = __builtin_amdgcn_readfirstlane(int(unsigned(kb * stepB)));
#pragma unroll
    for (int i = 0; i < AIT; ++i)
      sg.ra[i] = __builtin_bit_cast(v2f64s, __builtin_amdgcn_raw_buffer_load_b128(rsA, int(voA[i]), soA, 0));
#pragma unroll
    for (int i = 0; i < BIT; ++i)
      sg.rb[i] = __builtin_bit_cast(v2f64s, __builtin_amdgcn_raw_buffer_load_b128(rsB, int(voB[i]), soB, 0));
  };
  auto lstore = [&](const Stage& sg, int buf) {
    double* as = lds + buf * STAGE;
    double* bs = as + SA_ELEMS;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int ch = tid + NTHR * i;
      if (TA) *reinterpret_cast<v2f64s*>(as + (ch / (SM / 2)) * SAS_K + 2 * (ch % (SM / 2))) = sg.ra[i];
      else *reinterpret_cast<v2f64s*>(as + (ch >> 3) * SAS_M + 2 * (ch & 7)) = sg.ra[i];
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int ch = tid + NTHR * i;
      const int kr = ch / BCH, nq = ch - kr * BCH;
      if (ch < 16 * BCH) *reinterpret_cast<v2f64s*>(bs + kr * SBS + 2 * nq) = sg.rb[i];
    }
  };
  auto compute = [&](int buf) {
    const double* as = lds + buf * STAGE;
    const double* bs = as + SA_ELEMS;
    // fragments of k-step kk + 1 are fetched while the MFMAs of step kk run
    double af[2], bf[2][NT];
    auto frag = [&](int kk, int slot) {
      const int krow = 4 * kk + (lane >> 4);
      af[slot] = TA ? as[krow * SAS_K + 16 * wave + (lane & 15)] : as[(16 * wave + (lane & 15)) * SAS_M + krow];
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[slot][t] = bs[krow * SBS + 16 * t + (lane & 15)];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < SK / 4; ++kk) {
      if (kk + 1 < SK / 4) frag(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1], bf[kk & 1][t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // blocks past the end re-read the last one (loaded, never consumed)
  auto kof = [&](int64_t kb) { return min(kb, nkb - 1); };

  Stage sA, sB;
  gload(sA, kof(0));
  gload(sB, kof(1));
  lstore(sA, 0);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; kb += 2) {
    // block kb is in LDS buffer 0, block kb + 1 in flight in sB
    gload(sA, kof(kb + 2));
    compute(0);
    lstore(sB, 1);
    __syncthreads();
    if (kb + 1 >= nkb) break;
    gload(sB, kof(kb + 3));
    compute(1);
    lstore(sA, 0);
    __syncthreads();
  }

  // C/D layout: row (A side) = (lane >> 4) + 4 r, col (B side) = lane & 15
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t m = m0 + 16 * wave + (lane >> 4) + 4 * r;
    if (m >= M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t n = 16 * t + (lane & 15);
      if (n >= N) continue;
      double* cp = C + m * ldc + n;
      if (split) {
        unsafeAtomicAdd(cp, alpha * acc[t][r]);
      } else {
        double v = alpha * acc[t][r];
        if (beta != 0.0) v += beta * *cp;
        *cp = v;
      }
    }
  }
}

__global__ void k_scale2d_skinny(int64_t total, int64_t cols, double* __restrict__ A, int64_t lda, double beta) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    A[r * lda + cc] = beta == 0.0 ? 0.0 : beta * A[r * lda + cc];
  }
}

bool gemm_f64_skinny_eligible(bool tA, bool tB, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                              const double* B, int64_t ldb) {
  static const char* off = getenv("CCZ_GEMM_SKINNY_OFF");
  if (off && off[0] == '1') return false;
  if (tB) return false;
  if (N <= 48 || N > 192 || (N & 1)) return false;
  if (M < 1024 || K < 1024 || K % SK != 0) return false;
  if (tA && (M & 1)) return false;
  // 32-bit buffer offsets: a 128-row stripe (A as stored) / the whole B panel must span < 4 GiB
  if (!tA && int64_t(16 * SKW) * lda * 8 >= (int64_t(1) << 32)) return false;
  if (K * ldb * 8 >= (int64_t(1) << 32)) return false;
  if (tA && (K / 16 + SK) * lda * 8 >= (int64_t(1) << 32)) return false;
  if ((lda | ldb) & 1) return false;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return false;
  return true;
}

template <bool TA, int NT>
static void launch_skinny(hipStream_t st, dim3 grid, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                          int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int64_t kps) {
  constexpr int SM = 16 * SKW;
  const int sbs = (NT * 16) % 32 == 16 ? NT * 16 : NT * 16 + 16;
  const int sa = std::max(SK * (SM + 16), SM * SAS_M);
  const size_t lds_bytes = size_t(2) * (sa + SK * sbs) * 8;
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f64_skinny<TA, NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
  hipLaunchKernelGGL((k_gemm_f64_skinny<TA, NT>), grid, dim3(64 * SKW), lds_bytes, st, M, N, K, alpha, A, lda, B, ldb, beta, C,
                     ldc, kps);
}

void gemm_f64_skinny(ccz_ctx* c, bool tA, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  hipStream_t st = stream(c);
  constexpr int SM = 16 * SKW;
  const int64_t tm = (M + SM - 1) / SM;
  const int slots = 2 * std::max(1, impl(c)->props.multiProcessorCount);   // two workgroups per CU
  // one workgroup per CU.  Split K so that the grid fills whole rounds of the chip: cost ~ rounds / splits
  // (+2% per slice for the atomic epilogue); with transA a slice must also span < 4 GiB of A (32-bit offsets)
  const int smax = int(std::max<int64_t>(1, std::min<int64_t>(16, K / 512)));
  int smin = 1;
  if (tA) smin = int((K * lda * 8 + (int64_t(1) << 32) - 1) >> 32) + ((K * lda * 8) % (int64_t(1) << 32) == 0 ? 1 : 0);
  if (smin < 1) smin = 1;
  int splits = smin;
  double best = 1e300;
  for (int sp = smin; sp <= std::max(smin, smax); ++sp) {
    const double rounds = double((tm * sp + slots - 1) / slots);
    const double cost = rounds / sp + 0.02 * sp;
    if (cost < best) { best = cost; splits = sp; }
  }
  int64_t kps = K;
  if (splits > 1) {
    kps = ((K + splits - 1) / splits + SK - 1) / SK * SK;
    splits = int((K + kps - 1) / kps);
  }
  if (splits > 1) {
    const int64_t total = M * N;
    hipLaunchKernelGGL(k_scale2d_skinny, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                       total, N, C, ldc, beta);
  }
  dim3 grid((unsigned)tm, (unsigned)splits);
  const int nt = int((N + 15) / 16);
#define CCZ_SK(NT_)                                                                                       \
  do {                                                                                                    \
    if (tA) launch_skinny<true, NT_>(st, grid, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kps);        \
    else launch_skinny<false, NT_>(st, grid, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, kps);          \
  } while (0)
  if (nt <= 4) CCZ_SK(4);
  else if (nt == 5) CCZ_SK(5);
  else if (nt == 6) CCZ_SK(6);
  else if (nt <= 8) CCZ_SK(8);
  else if (nt <= 10) CCZ_SK(10);
  else CCZ_SK(12);
#undef CCZ_SK
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
