// K1 -- second moments of the column-stacked views on the gfx950 matrix pipe.
//
//   G += [X_1..X_m]' [X_1..X_m]      (upper-triangular tiles, fp64 in HBM)
//   s += 1' [X_1..X_m]
//
// Data layout: every view is row-major n x d_i (the reference's layout,
// cca_zoo/_base.py:78-102), so a k-step of the contraction is a ROW of the
// data: the A and B operands of X'X are both "k-major", exactly the layout the
// MFMA A (i, k) / B (k, j) fragments want.  A workgroup stages BK rows of a
// 256-column (fp32) / 128-column (fp64) panel into LDS with coalesced 16-byte
// loads; each wave then reads its fragments with ONE ds_read_b128 per operand
// per k-step: lane l takes 4 consecutive columns 4*(l % 32) .. +3 of row
// (l / 32), and the 4 values feed 4 different 32x32 MFMA tiles (tile t owns the
// columns == t mod 4).  That strided tile ownership is undone in the epilogue.
//
// Work decomposition: grid = (upper-triangular tile) x (row chunk).  fp32 views
// accumulate a chunk (<= 16384 rows) in fp32 MFMA accumulators and flush into
// the fp64 G with hardware fp64 atomics (the flush costs ~45 us per workgroup:
// 4096-row chunks lose 4.5% to it, 16384-row chunks 1.2%), so cross-chunk / cross-GPU
// accumulation is fp64 (hard part 2).  blockIdx is chunk-major:
// the ~256 resident workgroups stream the same rows of different panels, which
// keeps the panel rows hot in L2 / Infinity Cache while HBM sees each input
// byte about once.
//
// Roofline: F = n D (D+1) flop on v_mfma_f32_32x32x2_f32 (157.3 TF peak) or
// v_mfma_f64_16x16x4_f64; B = n D sizeof(T) bytes.  MFMA-bound.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hip_common.h"
#include "gram_map.h"

namespace ccz {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;

// The panel pointers come out of the tile table (memory), so the compiler cannot prove their
// address space and would emit flat_load (which also ties up lgkmcnt and stalls the LDS reads).
// Cast to the global address space explicitly.
typedef const float __attribute__((address_space(1)))* gptr_f32;
typedef const double __attribute__((address_space(1)))* gptr_f64;

// Fast-path staging loads go through a buffer resource: the descriptor (panel base + byte
// extent of this workgroup's rows) sits in SGPRs, the per-lane byte offset is a loop-invariant
// VGPR and the k-block advance is a scalar add -- zero vector address arithmetic inside the
// MFMA loop (PMC: the 64-bit address math of plain global loads cost ~13% of the wave's issue
// time).  Rows past the extent are out of range for the descriptor and read as 0, which is
// exactly the zero padding the row tail needs.
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// fp32: 256 x 256 tile per workgroup, 4 waves (2 x 2), each wave 128 x 128 =
// 4 x 4 MFMA 32x32x2 tiles (256 accumulator registers, one wave per SIMD)
// ---------------------------------------------------------------------------
constexpr int T32 = 256;

template <bool FAST>
__device__ __forceinline__ v4f32 load4_f32(gptr_f32 base, int64_t row, int64_t last_row, int cg, int64_t ld, int width) {
  // rows past the shard end are clamped to a valid row here and zeroed when staged into LDS
  // (a select on the loaded value here would force the vmcnt wait ahead of the MFMAs)
  const bool ok = row <= last_row;
  gptr_f32 p = base + (ok ? row : last_row) * ld + 4 * cg;
  v4f32 v = {0.f, 0.f, 0.f, 0.f};
  if (FAST) {
    v = *reinterpret_cast<const v4f32 __attribute__((address_space(1)))*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * cg + e < width) v[e] = p[e];
  }
  return v;
}

// PILOT (fp32 robustness, reference semantics cca_zoo/_base.py:97-99 "centre before any product"): `pilot` holds one
// float per stacked column (the column mean of this launch's rows, rounded to fp32) and is subtracted while the rows
// are staged into LDS, so the MFMA accumulates (x - p)(x - p)' -- numbers of the size of the covariance instead of
// mean^2 + covariance.  Every thread stages the same four columns of every row, so the pilot costs eight registers
// and 32 VALU subtractions per 128 MFMAs.  The shift is undone on the d x d side in fp64 (k_pilot_fixup).
template <bool FAST>
__global__ __launch_bounds__(256, 1) void k_gram_f32(const GramTile* __restrict__ tiles, int ntiles, int per_xcd,
                                                     int64_t ksplit, int64_t n, int64_t rows_per_wg,
                                                     double* __restrict__ G, int64_t ldg,
                                                     const float* __restrict__ pilot, float* __restrict__ partial, int64_t row0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);  // [2 buffers][A | B][BK][256]
  const WorkItem wi = locate_work(blockIdx.x, ntiles, per_xcd, ksplit);
  if (!wi.valid) return;
  const GramTile t = tiles[wi.tile];
  const int64_t k_begin = row0 + wi.chunk * rows_per_wg;      // row0: first row of this launch (the tail after a FIFO launch)
  const int64_t k_end = min(n, k_begin + rows_per_wg);
  if (k_begin >= k_end) return;
  gptr_f32 A = (gptr_f32)(t.a);
  gptr_f32 B = (gptr_f32)(t.b);
  const int64_t last_row = k_end - 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int cg = tid & 63, r4 = tid >> 6;

  v16f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4f32 ra[4], rb[4];
  // fast path: descriptors over this workgroup's rows of the two panels
  const int64_t nrows = k_end - k_begin;
  __amdgpu_buffer_rsrc_t srcA, srcB;
  int voffA[4], voffB[4];
  if (FAST) {
    srcA = panel_rsrc(static_cast<const float*>(t.a) + k_begin * t.lda, nrows * t.lda * 4);
    srcB = panel_rsrc(static_cast<const float*>(t.b) + k_begin * t.ldb, nrows * t.ldb * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      voffA[i] = int(((r4 + 4 * i) * t.lda + 4 * cg) * 4);
      voffB[i] = int(((r4 + 4 * i) * t.ldb + 4 * cg) * 4);
    }
  }
  auto gload = [&](int64_t k0) {
    if (FAST) {
      const int soffA = __builtin_amdgcn_readfirstlane(int((k0 - k_begin) * t.lda * 4));
      const int soffB = __builtin_amdgcn_readfirstlane(int((k0 - k_begin) * t.ldb * 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcA, voffA[i], soffA, 0));
        rb[i] = __builtin_bit_cast(v4f32, __builtin_amdgcn_raw_buffer_load_b128(srcB, voffB[i], soffB, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = k0 + r4 + 4 * i;
        ra[i] = load4_f32<false>(A, row, last_row, cg, t.lda, t.wa);
        rb[i] = load4_f32<false>(B, row, last_row, cg, t.ldb, t.wb);
      }
    }
  };
  // pilot values of this thread's four A-panel and four B-panel columns (0 past the panel width: those lanes
  // staged zeros and must keep them)
  v4f32 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
  if (pilot) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (4 * cg + e < t.wa) pa[e] = pilot[t.out_row + 4 * cg + e];
      if (4 * cg + e < t.wb) pb[e] = pilot[t.out_col + 4 * cg + e];
    }
  }
  auto lstore = [&](int buf, int64_t k0) {
    float* as = lds + buf * (2 * BK * T32);
    float* bs = as + BK * T32;
    const v4f32 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // fast path without a pilot: the descriptor already zero-fills; with a pilot the rows past the end must
      // stay exactly zero (0 - p would count p p' for every padded row)
      const bool ok = (FAST && !pilot) || k0 + r4 + 4 * i <= last_row;
      *reinterpret_cast<v4f32*>(as + (r4 + 4 * i) * T32 + 4 * cg) = ok ? ra[i] - pa : z;
      *reinterpret_cast<v4f32*>(bs + (r4 + 4 * i) * T32 + 4 * cg) = ok ? rb[i] - pb : z;
    }
  };

  const int64_t nkb = (k_end - k_begin + BK - 1) / BK;
  gload(k_begin);
  lstore(0, k_begin);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) gload(k_begin + (kb + 1) * BK);
    const float* as = lds + cur * (2 * BK * T32);
    const float* bs = as + BK * T32;
    // fragment reads run one k-step ahead of the MFMAs that consume them
    v4f32 af[2], bf[2];
    af[0] = *reinterpret_cast<const v4f32*>(as + (lane >> 5) * T32 + wr * 128 + 4 * (lane & 31));
    bf[0] = *reinterpret_cast<const v4f32*>(bs + (lane >> 5) * T32 + wc * 128 + 4 * (lane & 31));
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) {
        const int krow = 2 * (kk + 1) + (lane >> 5);
        af[(kk + 1) & 1] = *reinterpret_cast<const v4f32*>(as + krow * T32 + wr * 128 + 4 * (lane & 31));
        bf[(kk + 1) & 1] = *reinterpret_cast<const v4f32*>(bs + krow * T32 + wc * 128 + 4 * (lane & 31));
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next k-step's LDS reads in flight under these MFMAs
      const v4f32 a4 = af[kk & 1], b4 = bf[kk & 1];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[ti], b4[tj], acc[ti][tj], 0, 0, 0);
    }
    if (kb + 1 < nkb) lstore(cur ^ 1, k_begin + (kb + 1) * BK);
    __syncthreads();
  }

  // epilogue: fp32 chunk sums -> fp64 G.  32x32 C/D layout: col = lane & 31,
  // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); tile (ti, tj) owns the
  // columns == ti (A side) / tj (B side) mod 4 of the wave's 128-wide slabs.
  if (partial) {
    // small problems (a DCCA batch: 10 tiles x 25 row chunks) would send 25 workgroups' atomics to every address of
    // G (measured: 640 us for 70 us of MFMA work); instead every (chunk, tile) stores its 256 x 256 fp32 sums with plain
    // 16-byte stores and k_gram_reduce adds the chunks up in fp64
    float* pt = partial + (wi.chunk * int64_t(ntiles) + wi.tile) * int64_t(T32 * T32);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = wr * 128 + 4 * trow + ti;
        const v4f32 v = {acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]};
        *reinterpret_cast<v4f32*>(pt + i * T32 + wc * 128 + 4 * (lane & 31)) = v;
      }
    return;
  }
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int i = wr * 128 + 4 * trow + ti;
      if (!FAST && i >= t.wa) continue;
      double* grow = G + (t.out_row + i) * ldg + t.out_col;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int j = wc * 128 + 4 * (lane & 31) + tj;
        if (FAST || j < t.wb) unsafeAtomicAdd(grow + j, double(acc[ti][tj][r]));
      }
    }
}

// ---------------------------------------------------------------------------
// fp32 fast path, WAVE-PRIVATE LDS FIFO: no barriers, counted vmcnt only.
//
// For X'X on row-major X the MFMA A/B fragment of a k-step IS a coalesced 16-byte-per-lane
// global read (lane l: row k0 + l/32, columns 4 (l%32) .. +3 of the wave's 128-column slab), so
// a wave can stream its own operands without sharing anything with the other waves.  Keeping the
// in-flight window in VGPRs does not pipeline (hipcc drains vmcnt(0) at every loop back-edge for
// register-destination loads: measured 126 TF); instead the window lives in LDS:
// `buffer_load_dwordx4 ... lds` has no VGPR destination, so hipcc does not force vmcnt(0) at
// the loop back-edge and the hand-placed `s_waitcnt vmcnt(14)` keeps ~3 blocks (24 rows,
// ~12k cycles) of loads in flight per wave.  Each wave owns a 4-slot ring of 8-row blocks
// (A slab | B slab, 8 KiB per slot, 32 KiB per wave, 128 KiB per workgroup); a lane reads back
// with ds_read_b128 exactly the 16 bytes it DMA'd, so there is no cross-wave hazard and the
// only ordering needed is the issuing wave's own vmcnt.
//
// Issue budget (measured): instructions of the same wave are NOT hidden under its MFMAs -- 8
// extra VALU ops per k-step cost 8% -- so the loop is unrolled over the whole ring period
// (4 blocks x 4 k-steps): every LDS address is base + immediate, and a k-step is 16 MFMAs +
// 2 ds_read_b128 + 2 LDS-DMA (+ m0 / soffset scalar adds), spread over the four MFMA groups.
// For the same reason the column sums stay in their own HBM-bound pass (5 ms at the headline
// shape) instead of riding in this loop.
// ---------------------------------------------------------------------------
constexpr int FB = 8;        // rows per FIFO block
constexpr int FR = 4;        // ring slots per wave
constexpr int FSLAB = FB * 128 * 4;       // bytes of one slab (A or B) in a slot
constexpr int FSLOT = 2 * FSLAB;          // bytes per slot: A slab + B slab

// One wave's whole pipeline for a 128 x 128 quadrant: rows [0, nrows) of the two 128-column slabs behind srcA / srcB
// -> acc.  SYM (quadrant on the diagonal of G: A slab == B slab) skips the MFMA tiles with ti > tj: with the strided
// tile ownership tile (tj, ti) is the transpose of tile (ti, tj), so 10 of the 16 tiles carry all the information.
// a + b on packed pairs: two v_pk_add_f32 instead of four v_add_f32 (instructions of a wave are not hidden under
// its own MFMAs -- every VALU op in the k-step loop costs ~1 % of the kernel)
typedef float v2f32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f32 pk_add4(v4f32 a, v4f32 b) {
  v2f32 lo = {a[0], a[1]}, hi = {a[2], a[3]};
  const v2f32 blo = {b[0], b[1]}, bhi = {b[2], b[3]};
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(bhi));
  return v4f32{lo[0], lo[1], hi[0], hi[1]};
}

// PILOT: the per-lane pilot values pa / pb (the fp32 column means of the launch's rows, for the four columns this lane
// feeds into the A / B operands) are subtracted from the fragments right after the LDS read: 8 VALU subtractions per
// k-step as four v_pk_add_f32, so that data far from zero accumulates (x - p)(x - p)' like the reference's
// centre-then-multiply.  Rows past the extent would read as 0 - p: the caller guarantees nrows % (FB * FR) == 0.
template <bool SYM, bool PILOT>
__device__ __forceinline__ void gram_fifo_quadrant(v16f32 (&acc)[4][4], char* ring, const char* rd,
                                                   __amdgpu_buffer_rsrc_t srcA, __amdgpu_buffer_rsrc_t srcB, int voffA,
                                                   int voffB, int stepA, int stepB, int64_t nrows, v4f32 pa, v4f32 pb) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int soffA = 0, soffB = 0;          // byte offset of the next k-step to DMA (rows past the extent arrive as 0)
  // prologue: blocks 0, 1, 2 -> slots 0, 1, 2 (12 k-steps = 24 DMA instructions)
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int u = 0; u < FB / 2; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + s * FSLOT + u * 1024), 16, voffA, soffA, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + s * FSLOT + FSLAB + u * 1024), 16, voffB, soffB, 0, 0);
      soffA += stepA;
      soffB += stepB;
    }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                  // block 0 landed
  v4f32 af[2], bf[2];
  af[0] = *reinterpret_cast<const v4f32*>(rd);
  bf[0] = *reinterpret_cast<const v4f32*>(rd + FSLAB);

  const int64_t nblk = (nrows + FB - 1) / FB;
  for (int64_t b0 = 0; b0 < nblk; b0 += FR) {      // one trip = the whole ring period: slots are static
#pragma unroll
    for (int bb = 0; bb < FR; ++bb) {
#pragma unroll
      for (int u = 0; u < FB / 2; ++u) {
        constexpr int NU = FB / 2;
        const int cur = (bb * NU + u) & 1, nxt = cur ^ 1;
        // next k-step to read: (bb, u+1) or the first of the next slot
        const int nslot = (u + 1 < NU) ? bb : (bb + 1) % FR;
        const int nu = (u + 1 < NU) ? u + 1 : 0;
        const int wsl = (bb + 3) % FR;               // slot being refilled: block b0 + bb + 3
        v4f32 a4 = af[cur], b4 = bf[cur];
        if (PILOT) { a4 = pk_add4(a4, pa); b4 = pk_add4(b4, pb); }      // pa / pb hold the NEGATED pilot
        // -- gap 0: fragment reads for the next k-step
        if (u == NU - 1) {
          // the next k-step opens block b+1: newer than it are block b+2 (8) and 3 k-steps of b+3 (6)
          asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        }
        af[nxt] = *reinterpret_cast<const v4f32*>(rd + nslot * FSLOT + nu * 1024);
        bf[nxt] = *reinterpret_cast<const v4f32*>(rd + nslot * FSLOT + FSLAB + nu * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[0][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4[tj], acc[0][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- gap 1: DMA of the A slab rows of k-step u of block b+3
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + wsl * FSLOT + u * 1024), 16, voffA, soffA, 0, 0);
        soffA += stepA;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = SYM ? 1 : 0; tj < 4; ++tj) acc[1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b4[tj], acc[1][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- gap 2: DMA of the B slab rows
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + wsl * FSLOT + FSLAB + u * 1024), 16, voffB, soffB, 0, 0);
        soffB += stepB;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = SYM ? 2 : 0; tj < 4; ++tj) acc[2][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b4[tj], acc[2][tj], 0, 0, 0);
#pragma unroll
        for (int tj = SYM ? 3 : 0; tj < 4; ++tj) acc[3][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b4[tj], acc[3][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool PILOT>
__global__ __launch_bounds__(256, 1) void k_gram_f32_fifo(const GramTile* __restrict__ tiles, int ntiles, int per_xcd,
                                                          int64_t ksplit, int64_t n, int64_t rows_per_wg,
                                                          double* __restrict__ G, int64_t ldg, const float* __restrict__ pilot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WorkItem wi = locate_work(blockIdx.x, ntiles, per_xcd, ksplit);
  if (!wi.valid) return;
  const GramTile t = tiles[wi.tile];
  const int64_t k_begin = wi.chunk * rows_per_wg;
  const int64_t k_end = min(n, k_begin + rows_per_wg);
  if (k_begin >= k_end) return;
  const int64_t nrows_wg = k_end - k_begin;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wr = wave >> 1, wc = wave & 1;
  // Diagonal tiles (panel a == panel b) hold three distinct 128 x 128 quadrants: Q00 and Q11 are symmetric
  // (SYM path: 10 of 16 MFMA tiles) and Q10 = Q01'.  The wave that would compute Q10 instead takes the second
  // half of Q01's rows, so the tile finishes in 62.5% of the time of an off-diagonal tile.  (No barriers in this
  // kernel: the four waves are independent pipelines.)
  const bool diag = t.diag != 0;
  const bool sym = diag && wr == wc;
  int64_t row0 = 0, nrows = nrows_wg;
  if (diag && wr != wc) {
    constexpr int HB = PILOT ? FB * FR : FB;           // (pilot: both halves stay whole ring periods -- no zero-filled rows)
    const int64_t half = ((nrows_wg + 1) / 2 + HB - 1) / HB * HB;
    if (wr == 0) nrows = min(half, nrows_wg);          // wave (0,1): rows [0, half)
    else { row0 = min(half, nrows_wg); nrows = nrows_wg - row0; wr = 0; wc = 1; }   // wave (1,0): the rest of Q01
    if (nrows <= 0) return;
  }
  char* ring = smem + wave * (FR * FSLOT);       // wave-uniform base of this wave's ring
  const char* rd = ring + lane * 16;             // per-lane read base: every read is rd + immediate

  const __amdgpu_buffer_rsrc_t srcA =
      panel_rsrc(static_cast<const float*>(t.a) + (k_begin + row0) * t.lda + wr * 128, ((nrows - 1) * t.lda + 128) * 4);
  const __amdgpu_buffer_rsrc_t srcB =
      panel_rsrc(static_cast<const float*>(t.b) + (k_begin + row0) * t.ldb + wc * 128, ((nrows - 1) * t.ldb + 128) * 4);
  const int voffA = int(((lane >> 5) * t.lda + 4 * (lane & 31)) * 4);
  const int voffB = int(((lane >> 5) * t.ldb + 4 * (lane & 31)) * 4);
  const int stepA = __builtin_amdgcn_readfirstlane(int(2 * t.lda * 4));   // bytes per k-step (2 rows)
  const int stepB = __builtin_amdgcn_readfirstlane(int(2 * t.ldb * 4));

  v16f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  v4f32 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
  if (PILOT) {
    pa = -*reinterpret_cast<const v4f32*>(pilot + t.out_row + wr * 128 + 4 * (lane & 31));    // negated: the loop adds
    pb = -*reinterpret_cast<const v4f32*>(pilot + t.out_col + wc * 128 + 4 * (lane & 31));
  }
  if (sym) gram_fifo_quadrant<true, PILOT>(acc, ring, rd, srcA, srcB, voffA, voffB, stepA, stepB, nrows, pa, pb);
  else gram_fifo_quadrant<false, PILOT>(acc, ring, rd, srcA, srcB, voffA, voffB, stepA, stepB, nrows, pa, pb);

  // epilogue: fp32 chunk sums -> fp64 G (atomics: other row chunks / the other half of Q01 add into the same tile).
  // Tile (ti, tj) owns rows 4 trow + ti and columns 4 (lane & 31) + tj of the quadrant.
  if (!sym) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = wr * 128 + 4 * trow + ti;
        double* grow = G + (t.out_row + i) * ldg + t.out_col;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const int j = wc * 128 + 4 * (lane & 31) + tj;
          unsafeAtomicAdd(grow + j, double(acc[ti][tj][r]));
        }
      }
  } else {
    // symmetric quadrant: an element below the diagonal (i > j) of a computed tile ti < tj is the mirror of an
    // element of the skipped tile (tj, ti): it is added at (j, i).  Tiles ti == tj keep their upper half only.
    double* Q = G + (t.out_row + wr * 128) * ldg + t.out_col + wc * 128;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = 4 * trow + ti;
#pragma unroll
        for (int tj = ti; tj < 4; ++tj) {
          const int j = 4 * (lane & 31) + tj;
          const double v = double(acc[ti][tj][r]);
          if (i <= j) unsafeAtomicAdd(Q + int64_t(i) * ldg + j, v);
          else if (tj != ti) unsafeAtomicAdd(Q + int64_t(j) * ldg + i, v);
        }
      }
  }
}

// ---------------------------------------------------------------------------
// The FIFO kernel on a SMALL grid (a DCCA batch: ten tiles x a few hundred rows per workgroup, one round of workgroups):
// per-workgroup fp32 partial tiles instead of atomics (25 workgroups adding to every address of G cost 640 us), and a
// per-TILE row split -- a diagonal tile costs 62.5 % of an off-diagonal one, so it gets 1.6x the rows and the round ends
// together (uniform chunks left 15 % of the chip idle: 352 rows x 0.213 us = 75 us against 61 us).
//   plan[3 t .. 3 t + 2] = {first workgroup, workgroups, rows per workgroup (whole ring periods)} of tile t.
// Partial layout of slot blockIdx.x (65536 floats), read by loss.hip::k_loss_prep_partials(fifo_layout):
//   off-diagonal tile: the 256 x 256 tile, row-major;
//   diagonal tile: Q00 and Q11 hold their UPPER triangles only (the symmetric quadrants compute 10 of 16 MFMA tiles);
//     Q01 is the sum of its own slot (rows [0, half) of the workgroup) and of the Q10 slot (the remaining rows, stored there
//     un-transposed by the wave that would otherwise repeat Q01').
// ---------------------------------------------------------------------------
template <bool PILOT>
__global__ __launch_bounds__(256, 1) void k_gram_f32_fifo_small(const GramTile* __restrict__ tiles, int ntiles, const int* __restrict__ plan,
                                                                int64_t n, float* __restrict__ partial, const float* __restrict__ pilot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tile = 0;
  while (tile + 1 < ntiles && int(blockIdx.x) >= plan[3 * (tile + 1)]) ++tile;
  const int chunk = int(blockIdx.x) - plan[3 * tile];
  if (chunk >= plan[3 * tile + 1]) return;
  const int64_t rows_per_wg = plan[3 * tile + 2];
  const GramTile t = tiles[tile];
  const int64_t k_begin = int64_t(chunk) * rows_per_wg;
  const int64_t k_end = min(n, k_begin + rows_per_wg);
  float* pt = partial + int64_t(blockIdx.x) * 65536;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wr = wave >> 1, wc = wave & 1;
  const bool diag = t.diag != 0;
  const bool sym = diag && wr == wc;
  const int64_t nrows_wg = max<int64_t>(k_end - k_begin, 0);
  int64_t row0 = 0, nrows = nrows_wg;
  bool second_half = false;
  if (diag && wr != wc) {
    constexpr int HB = FB * FR;                              // both halves stay whole ring periods
    const int64_t half = ((nrows_wg + 1) / 2 + HB - 1) / HB * HB;
    if (wr == 0) nrows = min(half, nrows_wg);
    else { row0 = min(half, nrows_wg); nrows = nrows_wg - row0; wr = 0; wc = 1; second_half = true; }
  }
  v16f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nrows > 0) {
    char* ring = smem + wave * (FR * FSLOT);
    const char* rd = ring + lane * 16;
    const __amdgpu_buffer_rsrc_t srcA =
        panel_rsrc(static_cast<const float*>(t.a) + (k_begin + row0) * t.lda + wr * 128, ((nrows - 1) * t.lda + 128) * 4);
    const __amdgpu_buffer_rsrc_t srcB =
        panel_rsrc(static_cast<const float*>(t.b) + (k_begin + row0) * t.ldb + wc * 128, ((nrows - 1) * t.ldb + 128) * 4);
    const int voffA = int(((lane >> 5) * t.lda + 4 * (lane & 31)) * 4);
    const int voffB = int(((lane >> 5) * t.ldb + 4 * (lane & 31)) * 4);
    const int stepA = __builtin_amdgcn_readfirstlane(int(2 * t.lda * 4));
    const int stepB = __builtin_amdgcn_readfirstlane(int(2 * t.ldb * 4));
    v4f32 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
    if (PILOT) {
      pa = -*reinterpret_cast<const v4f32*>(pilot + t.out_row + wr * 128 + 4 * (lane & 31));
      pb = -*reinterpret_cast<const v4f32*>(pilot + t.out_col + wc * 128 + 4 * (lane & 31));
    }
    if (sym) gram_fifo_quadrant<true, PILOT>(acc, ring, rd, srcA, srcB, voffA, voffB, stepA, stepB, nrows, pa, pb);
    else gram_fifo_quadrant<false, PILOT>(acc, ring, rd, srcA, srcB, voffA, voffB, stepA, stepB, nrows, pa, pb);
  }
  // epilogue: plain stores into this workgroup's slot (zeros from a wave that had no rows)
  if (!sym) {
    const int qr = second_half ? 1 : wr, qc = second_half ? 0 : wc;     // the second half of Q01 lands in the Q10 slot
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = qr * 128 + 4 * trow + ti;
        const v4f32 v = {acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]};
        *reinterpret_cast<v4f32*>(pt + i * 256 + qc * 128 + 4 * (lane & 31)) = v;
      }
  } else {
    float* Q = pt + (wr * 128) * 256 + wc * 128;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = 4 * trow + ti;
#pragma unroll
        for (int tj = ti; tj < 4; ++tj) {
          const int j = 4 * (lane & 31) + tj;
          const float v = acc[ti][tj][r];
          if (i <= j) Q[i * 256 + j] = v;
          else if (tj != ti) Q[j * 256 + i] = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------
// fp64: 128 x 128 tile per workgroup, 4 waves (2 x 2), each wave 64 x 64 =
// 4 x 4 MFMA f64 16x16x4 tiles (128 accumulator registers)
// ---------------------------------------------------------------------------
constexpr int T64 = 128;

template <bool FAST>
__device__ __forceinline__ v2f64 load2_f64(gptr_f64 base, int64_t row, int64_t last_row, int cg, int64_t ld, int width) {
  const bool ok = row <= last_row;
  gptr_f64 p = base + (ok ? row : last_row) * ld + 2 * cg;
  v2f64 v = {0.0, 0.0};
  if (FAST) {
    v = *reinterpret_cast<const v2f64 __attribute__((address_space(1)))*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (2 * cg + e < width) v[e] = p[e];
  }
  return v;
}

template <bool FAST>
__global__ __launch_bounds__(256, 1) void k_gram_f64(const GramTile* __restrict__ tiles, int ntiles, int per_xcd,
                                                     int64_t ksplit, int64_t n, int64_t rows_per_wg,
                                                     double* __restrict__ G, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);  // [2 buffers][A | B][BK][128]
  const WorkItem wi = locate_work(blockIdx.x, ntiles, per_xcd, ksplit);
  if (!wi.valid) return;
  const GramTile t = tiles[wi.tile];
  const int64_t k_begin = wi.chunk * rows_per_wg;
  const int64_t k_end = min(n, k_begin + rows_per_wg);
  if (k_begin >= k_end) return;
  gptr_f64 A = (gptr_f64)(t.a);
  gptr_f64 B = (gptr_f64)(t.b);
  const int64_t last_row = k_end - 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int cg = tid & 63, r4 = tid >> 6;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  v2f64 ra[4], rb[4];
  const int64_t nrows = k_end - k_begin;
  __amdgpu_buffer_rsrc_t srcA, srcB;
  int voffA[4], voffB[4];
  if (FAST) {
    srcA = panel_rsrc(static_cast<const double*>(t.a) + k_begin * t.lda, nrows * t.lda * 8);
    srcB = panel_rsrc(static_cast<const double*>(t.b) + k_begin * t.ldb, nrows * t.ldb * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      voffA[i] = int(((r4 + 4 * i) * t.lda + 2 * cg) * 8);
      voffB[i] = int(((r4 + 4 * i) * t.ldb + 2 * cg) * 8);
    }
  }
  auto gload = [&](int64_t k0) {
    if (FAST) {
      const int soffA = __builtin_amdgcn_readfirstlane(int((k0 - k_begin) * t.lda * 8));
      const int soffB = __builtin_amdgcn_readfirstlane(int((k0 - k_begin) * t.ldb * 8));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = __builtin_bit_cast(v2f64, __builtin_amdgcn_raw_buffer_load_b128(srcA, voffA[i], soffA, 0));
        rb[i] = __builtin_bit_cast(v2f64, __builtin_amdgcn_raw_buffer_load_b128(srcB, voffB[i], soffB, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = k0 + r4 + 4 * i;
        ra[i] = load2_f64<false>(A, row, last_row, cg, t.lda, t.wa);
        rb[i] = load2_f64<false>(B, row, last_row, cg, t.ldb, t.wb);
      }
    }
  };
  auto lstore = [&](int buf, int64_t k0) {
    double* as = lds + buf * (2 * BK * T64);
    double* bs = as + BK * T64;
    const v2f64 z = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = FAST || k0 + r4 + 4 * i <= last_row;
      *reinterpret_cast<v2f64*>(as + (r4 + 4 * i) * T64 + 2 * cg) = ok ? ra[i] : z;
      *reinterpret_cast<v2f64*>(bs + (r4 + 4 * i) * T64 + 2 * cg) = ok ? rb[i] : z;
    }
  };

  const int64_t nkb = (k_end - k_begin + BK - 1) / BK;
  gload(k_begin);
  lstore(0, k_begin);
  __syncthreads();
  for (int64_t kb = 0; kb < nkb; ++kb) {
    const int cur = int(kb & 1);
    if (kb + 1 < nkb) gload(k_begin + (kb + 1) * BK);
    const double* as = lds + cur * (2 * BK * T64);
    const double* bs = as + BK * T64;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int krow = 4 * kk + (lane >> 4);
      const double* ap = as + krow * T64 + wr * 64 + 4 * (lane & 15);
      const double* bp = bs + krow * T64 + wc * 64 + 4 * (lane & 15);
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
    if (kb + 1 < nkb) lstore(cur ^ 1, k_begin + (kb + 1) * BK);
    __syncthreads();
  }

  // f64 16x16 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = (lane >> 4) + 4 * r;
      const int i = wr * 64 + 4 * trow + ti;
      if (!FAST && i >= t.wa) continue;
      double* grow = G + (t.out_row + i) * ldg + t.out_col;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int j = wc * 64 + 4 * (lane & 15) + tj;
        if (FAST || j < t.wb) unsafeAtomicAdd(grow + j, acc[ti][tj][r]);
      }
    }
}

// ---------------------------------------------------------------------------
// fp64 Gram on the same wave-private LDS-DMA FIFO as k_gram_f32_fifo: 128 x 128 tile per workgroup, each wave a
// 64 x 64 quadrant = 4 x 4 v_mfma_f64_16x16x4_f64 tiles.  A k-step is 4 rows; lane l takes row k0 + (l >> 4),
// columns 4 (l & 15) .. + 3 of its 64-column slab as two 16-byte DMAs (the four doubles feed the four column
// tiles: strided ownership as in fp32).  A FIFO block is 8 rows = 2 k-steps = 8 DMA instructions, the slot
// layout is [A: k-step][half][lane] | [B: ...], ring of 4 slots (32 KiB per wave), counted vmcnt.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_gram_f64_fifo(const GramTile* __restrict__ tiles, int ntiles, int per_xcd,
                                                          int64_t ksplit, int64_t n, int64_t rows_per_wg,
                                                          double* __restrict__ G, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WorkItem wi = locate_work(blockIdx.x, ntiles, per_xcd, ksplit);
  if (!wi.valid) return;
  const GramTile t = tiles[wi.tile];
  const int64_t k_begin = wi.chunk * rows_per_wg;
  const int64_t k_end = min(n, k_begin + rows_per_wg);
  if (k_begin >= k_end) return;
  const int64_t nrows = k_end - k_begin;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* ring = smem + wave * (FR * FSLOT);
  const char* rd = ring + lane * 16;

  const __amdgpu_buffer_rsrc_t srcA =
      panel_rsrc(static_cast<const double*>(t.a) + k_begin * t.lda + wr * 64, ((nrows - 1) * t.lda + 64) * 8);
  const __amdgpu_buffer_rsrc_t srcB =
      panel_rsrc(static_cast<const double*>(t.b) + k_begin * t.ldb + wc * 64, ((nrows - 1) * t.ldb + 64) * 8);
  const int voffA = int(((lane >> 4) * t.lda + 4 * (lane & 15)) * 8);
  const int voffB = int(((lane >> 4) * t.ldb + 4 * (lane & 15)) * 8);
  const int voffA2 = voffA + 16, voffB2 = voffB + 16;   // second pair of doubles (no instruction offset: it would also move the LDS address)
  const int stepA = __builtin_amdgcn_readfirstlane(int(4 * t.lda * 8));   // bytes per k-step (4 rows)
  const int stepB = __builtin_amdgcn_readfirstlane(int(4 * t.ldb * 8));
  int soffA = 0, soffB = 0;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  // slot: A part [k-step u (2)][half (2)][lane (64)][16 B] = 4 KiB, B part the same at + FSLAB
  // prologue: blocks 0, 1, 2 -> slots 0, 1, 2 (6 k-steps = 24 DMA instructions)
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + s * FSLOT + u * 2048), 16, voffA, soffA, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + s * FSLOT + u * 2048 + 1024), 16, voffA2, soffA, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + s * FSLOT + FSLAB + u * 2048), 16, voffB, soffB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + s * FSLOT + FSLAB + u * 2048 + 1024), 16, voffB2, soffB, 0, 0);
      soffA += stepA;
      soffB += stepB;
    }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                  // block 0 landed
  v2f64 af[2][2], bf[2][2];
  af[0][0] = *reinterpret_cast<const v2f64*>(rd);
  af[0][1] = *reinterpret_cast<const v2f64*>(rd + 1024);
  bf[0][0] = *reinterpret_cast<const v2f64*>(rd + FSLAB);
  bf[0][1] = *reinterpret_cast<const v2f64*>(rd + FSLAB + 1024);

  const int64_t nblk = (nrows + FB - 1) / FB;
  for (int64_t b0 = 0; b0 < nblk; b0 += FR) {
#pragma unroll
    for (int bb = 0; bb < FR; ++bb) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cur = (bb * 2 + u) & 1, nxt = cur ^ 1;
        const int nslot = (u == 0) ? bb : (bb + 1) % FR;
        const int nu = (u == 0) ? 1 : 0;
        const int wsl = (bb + 3) % FR;               // slot being refilled: block b0 + bb + 3
        const double a4[4] = {af[cur][0][0], af[cur][0][1], af[cur][1][0], af[cur][1][1]};
        const double b4[4] = {bf[cur][0][0], bf[cur][0][1], bf[cur][1][0], bf[cur][1][1]};
        // -- gap 0: fragment reads for the next k-step
        if (u == 1) {
          // the next k-step opens block b+1: newer than it are block b+2 (8) and the first k-step of b+3 (4)
          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        af[nxt][0] = *reinterpret_cast<const v2f64*>(rd + nslot * FSLOT + nu * 2048);
        af[nxt][1] = *reinterpret_cast<const v2f64*>(rd + nslot * FSLOT + nu * 2048 + 1024);
        bf[nxt][0] = *reinterpret_cast<const v2f64*>(rd + nslot * FSLOT + FSLAB + nu * 2048);
        bf[nxt][1] = *reinterpret_cast<const v2f64*>(rd + nslot * FSLOT + FSLAB + nu * 2048 + 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[0][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], b4[tj], acc[0][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + wsl * FSLOT + u * 2048), 16, voffA, soffA, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[1][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], b4[tj], acc[1][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA, (lds_ptr)(ring + wsl * FSLOT + u * 2048 + 1024), 16, voffA2, soffA, 0, 0);
        soffA += stepA;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[2][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], b4[tj], acc[2][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + wsl * FSLOT + FSLAB + u * 2048), 16, voffB, soffB, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[3][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], b4[tj], acc[3][tj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB, (lds_ptr)(ring + wsl * FSLOT + FSLAB + u * 2048 + 1024), 16, voffB2, soffB, 0, 0);
        soffB += stepB;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // f64 16x16 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = (lane >> 4) + 4 * r;
      const int i = wr * 64 + 4 * trow + ti;
      double* grow = G + (t.out_row + i) * ldg + t.out_col;
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int j = wc * 64 + 4 * (lane & 15) + tj;
        unsafeAtomicAdd(grow + j, acc[ti][tj][r]);
      }
    }
}

// ---------------------------------------------------------------------------
// column sums: HBM-bound single pass, fp64 accumulation
// ---------------------------------------------------------------------------
template <typename T, bool SQ>
__global__ __launch_bounds__(256) void k_colsum(const T* __restrict__ X, int64_t n, int64_t cols, int64_t ld,
                                                double* __restrict__ out, double* __restrict__ out_sq,
                                                int64_t rows_per_block) {
  const int64_t col = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (col >= cols) return;
  const int64_t r0 = int64_t(blockIdx.y) * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  double q0 = 0.0, q1 = 0.0;
  int64_t r = r0;
  for (; r + 3 < r1; r += 4) {
    const double x0 = double(X[(r + 0) * ld + col]), x1 = double(X[(r + 1) * ld + col]);
    const double x2 = double(X[(r + 2) * ld + col]), x3 = double(X[(r + 3) * ld + col]);
    a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    if (SQ) { q0 += x0 * x0 + x2 * x2; q1 += x1 * x1 + x3 * x3; }
  }
  for (; r < r1; ++r) {
    const double x = double(X[r * ld + col]);
    a0 += x;
    if (SQ) q0 += x * x;
  }
  unsafeAtomicAdd(out + col, (a0 + a1) + (a2 + a3));
  if (SQ) unsafeAtomicAdd(out_sq + col, q0 + q1);
}

// ---------------------------------------------------------------------------
// Loss fast path (loss.hip): exact fp64 column sums of ALL views and the fp32 pilot (column mean) in ONE launch, no
// atomics on data, nothing to clear beforehand.  grid = (column blocks of 256 over the stacked width, row blocks); a
// block writes its partial sums to part[row block][D]; the LAST row block of a column block to arrive (one counter per
// column block, left at zero again) adds the row blocks up in a fixed order -- the sums do not depend on the arrival
// order -- and writes sums[j] and pilot[j] = fl32(sums[j] / n).
// ---------------------------------------------------------------------------
struct ColsumViews {
  const float* data[8];
  int64_t ld[8];
  int off[9];
  int m;
};

__global__ __launch_bounds__(256) void k_colsum_pilot(ColsumViews cv, int64_t n, int64_t D, int64_t rows_per_block, double* __restrict__ part,
                                                      unsigned* __restrict__ counters, double* __restrict__ sums, float* __restrict__ pilot,
                                                      double* __restrict__ zero_me, int64_t nzero) {
  __shared__ int last;
  if (zero_me && blockIdx.y == 0)                      // a rider: clear an accumulation target of the NEXT kernel (the split pass's msq)
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < nzero; e += int64_t(gridDim.x) * 256) zero_me[e] = 0.0;
  const int64_t j = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t r0 = int64_t(blockIdx.y) * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  if (j < D) {
    int a = 0;
    while (a + 1 < cv.m && j >= cv.off[a + 1]) ++a;
    const float* X = cv.data[a] + (j - cv.off[a]);
    const int64_t ld = cv.ld[a];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
      a0 += double(X[(r + 0) * ld]); a1 += double(X[(r + 1) * ld]);
      a2 += double(X[(r + 2) * ld]); a3 += double(X[(r + 3) * ld]);
    }
    for (; r < r1; ++r) a0 += double(X[r * ld]);
    st_shared(part + int64_t(blockIdx.y) * D + j, (a0 + a1) + (a2 + a3));     // write-through: read by another workgroup
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1 ? 1 : 0;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  if (j >= D) return;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  int rb = 0;
  const int nrb = int(gridDim.y);
  for (; rb + 3 < nrb; rb += 4) {
    t0 += ld_shared(part + int64_t(rb) * D + j); t1 += ld_shared(part + int64_t(rb + 1) * D + j);
    t2 += ld_shared(part + int64_t(rb + 2) * D + j); t3 += ld_shared(part + int64_t(rb + 3) * D + j);
  }
  for (; rb < nrb; ++rb) t0 += ld_shared(part + int64_t(rb) * D + j);
  const double sj = (t0 + t1) + (t2 + t3);
  sums[j] = sj;
  pilot[j] = float(sj / double(n));
}

// pilot[j] = fl32(colsum[j] / n): the fp32 value closest to the mean of this launch's rows
__global__ void k_pilot_from_sums(const double* __restrict__ s, int64_t D, double inv_n, float* __restrict__ pilot) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < D) pilot[j] = float(s[j] * inv_n);
}

// Undo the pilot shift on the d x d side, in fp64, and fold the launch's column sums into the running ones:
//   sum x_i x_j = sum (x_i - p_i)(x_j - p_j) + p_i s_j + p_j s_i - n p_i p_j      (s = column sums of these rows)
// Only elements on or above the diagonal are authoritative (every consumer reads G through its upper triangle).
__global__ void k_pilot_fixup(double* __restrict__ G, int64_t D, const double* __restrict__ s_launch, double n,
                              const float* __restrict__ pilot) {
  const int64_t i = blockIdx.y;
  const int64_t j = i + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= D) return;
  const double pi = double(pilot[i]), pj = double(pilot[j]);
  G[i * D + j] += pi * s_launch[j] + pj * s_launch[i] - n * pi * pj;
}

// G (upper tiles) += sum over the row chunks of the per-(chunk, tile) fp32 partial sums, in fp64
__global__ __launch_bounds__(256) void k_gram_reduce(const float* __restrict__ partial, const GramTile* __restrict__ tiles, int ntiles,
                                                     int64_t ksplit, double* __restrict__ G, int64_t ldg) {
  const int tile = blockIdx.y;
  const GramTile t = tiles[tile];
  const int e = blockIdx.x * 256 + threadIdx.x;        // element of the 256 x 256 tile
  const int i = e >> 8, j = e & 255;
  if (i >= t.wa || j >= t.wb) return;
  double acc = 0.0;
  const float* p = partial + int64_t(tile) * (T32 * T32) + e;
  for (int64_t c = 0; c < ksplit; ++c) acc += double(p[c * int64_t(ntiles) * (T32 * T32)]);
  G[(t.out_row + i) * ldg + t.out_col + j] += acc;
}

__global__ void k_vec_add(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) dst[j] += src[j];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

struct Panel {
  int view;
  int64_t col0, width, gcol0;
};

// The tile table of a launch (device copy, cached) and what the kernels need to know about it.
struct TileTable {
  GramTile* dev = nullptr;
  int ntiles = 0;
  bool fast = true;      // every view: width a multiple of the tile, 16-byte aligned rows -> the buffer-descriptor kernels
  std::vector<unsigned char> diag;   // host copy of the tiles' diag flags, in table order
};

template <typename T>
TileTable build_tile_table(ccz_ctx* c, const ccz_view* views, int n_views, void** tile_cache) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  constexpr bool is32 = sizeof(T) == 4;
  const int tile = is32 ? T32 : T64;
  std::vector<Panel> panels;
  int64_t g0 = 0;
  bool fast = true;
  for (int v = 0; v < n_views; ++v) {
    for (int64_t c0 = 0; c0 < views[v].cols; c0 += tile)
      panels.push_back({v, c0, std::min<int64_t>(tile, views[v].cols - c0), g0 + c0});
    if (views[v].cols % tile != 0) fast = false;
    if ((views[v].ld * sizeof(T)) % 16 != 0) fast = false;
    if (reinterpret_cast<uintptr_t>(views[v].data) % 16 != 0) fast = false;
    g0 += views[v].cols;
  }
  std::vector<GramTile> tiles;
  const int np = int(panels.size());
  // supertile order: 4 x 8 blocks of tiles, so that 32 consecutive tiles share 12 panels
  std::vector<std::pair<int, int>> order;
  for (int I = 0; I < np; I += 4)
    for (int J = (I / 8) * 8; J < np; J += 8)
      for (int i = I; i < std::min(np, I + 4); ++i)
        for (int j = std::max(i, J); j < std::min(np, J + 8); ++j) order.emplace_back(i, j);
  for (const auto& ij : order) {
    const int i = ij.first, j = ij.second;
    {
      const Panel& a = panels[i];
      const Panel& b = panels[j];
      GramTile t;
      t.a = static_cast<const T*>(views[a.view].data) + a.col0;
      t.b = static_cast<const T*>(views[b.view].data) + b.col0;
      t.lda = views[a.view].ld;
      t.ldb = views[b.view].ld;
      t.out_row = a.gcol0;
      t.out_col = b.gcol0;
      t.wa = int32_t(a.width);
      t.wb = int32_t(b.width);
      t.diag = (i == j) ? 1 : 0;
      t.pad_ = 0;
      tiles.push_back(t);
    }
  }
  // Per-XCD slices (locate_work): slice x = entries [x * per, (x+1) * per) of the table.  Diagonal tiles cost ~65%
  // of an off-diagonal one (k_gram_f32_fifo skips their redundant quadrant work), so they are dealt round-robin
  // to the slices and the off-diagonal tiles fill each slice up to `per` in supertile order (the last slice takes
  // what is left).  The table stays dense: the plain order of small grids uses the same list.
  // CCZ_GRAM_MAP=1 (default): chunk-per-XCD order (locate_work, per_xcd < 0).  The list is arranged so that rounds of
  // 32 consecutive tiles have uniform cost and shared panels: first the complete 4 x 8 supertiles (no diagonal tiles:
  // 12 panels per 32 tiles), then all diagonal tiles together (they cost ~65% of a full tile -- mixed into other
  // rounds they would stagger the XCD's workgroups for good), then the off-diagonal rest of the ragged supertiles.
  static const int map_mode = [] { const char* e = getenv("CCZ_GRAM_MAP"); return e ? atoi(e) : 1; }();
  if (map_mode == 1) {
    std::vector<GramTile> full, dg, rest;
    size_t pos = 0;
    for (int I = 0; I < np; I += 4)
      for (int J = (I / 8) * 8; J < np; J += 8) {
        size_t cnt = 0;
        bool has_diag = false;
        for (int i = I; i < std::min(np, I + 4); ++i)
          for (int j = std::max(i, J); j < std::min(np, J + 8); ++j) { ++cnt; has_diag = has_diag || i == j; }
        for (size_t q = 0; q < cnt; ++q) {
          const GramTile& tl = tiles[pos + q];
          if (cnt == 32 && !has_diag) full.push_back(tl);
          else (tl.diag ? dg : rest).push_back(tl);
        }
        pos += cnt;
      }
    if (pos != tiles.size()) fail(CCZ_EHIP, "gram: tile ordering lost tiles (internal error)");
    tiles.clear();
    tiles.insert(tiles.end(), full.begin(), full.end());
    tiles.insert(tiles.end(), dg.begin(), dg.end());
    tiles.insert(tiles.end(), rest.begin(), rest.end());
  } else {
    std::vector<GramTile> dg, off, balanced;
    for (const GramTile& tl : tiles) (tl.diag ? dg : off).push_back(tl);
    const size_t per = (tiles.size() + 7) / 8;
    size_t id = 0, io = 0;
    for (int x = 0; x < 8; ++x) {
      size_t cnt = 0;
      const size_t want_d = dg.size() / 8 + (size_t(x) < dg.size() % 8 ? 1 : 0);
      for (size_t q = 0; q < want_d && id < dg.size() && cnt < per; ++q, ++cnt) balanced.push_back(dg[id++]);
      for (; cnt < per && io < off.size(); ++cnt) balanced.push_back(off[io++]);
      for (; cnt < per && id < dg.size(); ++cnt) balanced.push_back(dg[id++]);
    }
    if (balanced.size() != tiles.size()) fail(CCZ_EHIP, "gram: tile slicing lost tiles (internal error)");
    tiles.swap(balanced);
  }
  const int ntiles = int(tiles.size());
  // the tile table depends only on the views (pointers, widths, strides): pipelined callers keep it on
  // the device across their chunks (*tile_cache) so that a launch enqueues no host copy and never blocks
  GramTile* d_tiles;
  bool tiles_from_handle_cache = false;
  if (tile_cache && *tile_cache) {
    d_tiles = static_cast<GramTile*>(*tile_cache);
  } else if (!tile_cache) {
    // handle-level cache keyed on the table's content (FNV-1a over its bytes): repeated launches on the same buffers
    // (a training loop; the bench's fit loop) enqueue no copy at all.  An evicted entry's buffer is rewritten by a
    // copy on the handle's stream, i.e. after every kernel that read it.
    const size_t tb = tiles.size() * sizeof(GramTile);
    uint64_t hsh = 1469598103934665603ull;
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(tiles.data());
    for (size_t q = 0; q < tb; ++q) { hsh ^= pb[q]; hsh *= 1099511628211ull; }
    static uint64_t tick = 0;
    Impl::TileTab* hit = nullptr;
    for (auto& e : im->tile_tabs)
      if (e.hash == hsh && e.bytes == tb) { hit = &e; break; }
    if (!hit) {
      if (im->tile_tabs.size() < 16) {
        void* dp = nullptr;
        CCZ_HIP(hipMalloc(&dp, std::max<size_t>(tb, 4096)));
        im->tile_tabs.push_back({hsh, tb, dp, 0});
        hit = &im->tile_tabs.back();
      } else {
        hit = &im->tile_tabs[0];
        for (auto& e : im->tile_tabs) if (e.tick < hit->tick) hit = &e;
        if (std::max<size_t>(hit->bytes, 4096) < tb) {
          CCZ_HIP(hipStreamSynchronize(st));
          CCZ_HIP(hipFree(hit->dev));
          hit->dev = nullptr;
          CCZ_HIP(hipMalloc(&hit->dev, tb));
        }
        hit->hash = hsh;
        hit->bytes = tb;
      }
      h2d_small(c, hit->dev, tiles.data(), tb);
    }
    hit->tick = ++tick;
    d_tiles = static_cast<GramTile*>(hit->dev);
    tiles_from_handle_cache = true;
  } else {
    d_tiles = static_cast<GramTile*>(dev_alloc(c, tiles.size() * sizeof(GramTile)));
    h2d_small(c, d_tiles, tiles.data(), tiles.size() * sizeof(GramTile));   // through a pinned slot: no stream sync
    *tile_cache = d_tiles;
  }

  TileTable tt;
  tt.dev = d_tiles;
  tt.ntiles = ntiles;
  tt.fast = fast;
  tt.diag.resize(tiles.size());
  for (size_t q = 0; q < tiles.size(); ++q) tt.diag[q] = tiles[q].diag ? 1 : 0;
  (void)tiles_from_handle_cache;          // handle-cached tables stay with the handle
  return tt;
}

// Row-chunk length and grid of a launch over n rows (see the cost model inside).
struct RowPlan {
  int64_t rows_per_wg = 0, ksplit = 0, nblocks = 0;
  int per_xcd = 0;
  bool sliced = false;
  bool fast = true;      // false: a view's rows do not fit the 32-bit buffer descriptor any more
};

template <typename T>
RowPlan plan_rows(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, int ntiles, bool fast) {
  Impl* im = impl(c);
  constexpr bool is32 = sizeof(T) == 4;
  static const int map_mode = [] { const char* e = getenv("CCZ_GRAM_MAP"); return e ? atoi(e) : 1; }();
  const int ncu = std::max(1, im->props.multiProcessorCount);
  static const int64_t rows_env = [] { const char* e = getenv("CCZ_GRAM_ROWS"); return e ? atoll(e) : 0LL; }();
  int64_t max_rows = rows_env > 0 ? rows_env : 16384;
  if (fast) {   // keep (rows + FIFO run-ahead) * ld * sizeof(T) inside the 32-bit buffer descriptor
    int64_t cap = max_rows;
    for (int v = 0; v < n_views; ++v)
      cap = std::min<int64_t>(cap, ((int64_t(1) << 31) - 1) / (views[v].ld * int64_t(sizeof(T))) - 128);
    if (cap < 256) fast = false; else max_rows = cap;
  }
  // Row-chunk length: minimise (rounds of workgroups over the CUs) x (rows per workgroup + fixed cost).
  // Every workgroup pays ~45 us to flush its tile (~210 rows of MFMA time) plus the pipeline prologue,
  // and a grid that overshoots a multiple of the CU count by a few workgroups wastes a whole round.
  // fp32 chunks are whole ring periods of the FIFO kernel (32 rows): its pilot-shifted form must not see zero-filled
  // rows inside a chunk (they would enter as 0 - p), only the launch's last chunk may be ragged
  const int64_t gran = (is32 && fast) ? FB * FR : BK;
  int64_t rows_per_wg;
  {
    const int64_t kmin = std::max<int64_t>(1, (n + max_rows - 1) / max_rows);
    const int64_t kmax = std::max<int64_t>(kmin, std::min<int64_t>(n / 256, 8192));
    double best = 1e300;
    int64_t best_k = kmin;
    for (int64_t ks = kmin; ks <= kmax; ++ks) {
      const int64_t rows = ((n + ks - 1) / ks + gran - 1) / gran * gran;
      const bool big = int64_t(ntiles) * ks >= 16 * int64_t(ncu);
      int64_t rounds;
      double penalty = 1.0;
      if (big && map_mode == 1 && ks % 8 == 0) rounds = ((ks / 8) * int64_t(ntiles) + ncu / 8 - 1) / (ncu / 8);   // chunk-per-XCD
      else if (big) { rounds = (int64_t((ntiles + 7) / 8) * ks + ncu / 8 - 1) / (ncu / 8); if (map_mode == 1) penalty = 1.03; }   // per-XCD slice
      else rounds = (int64_t(ntiles) * ks + ncu - 1) / ncu;
      const double cost = penalty * double(rounds) * double(rows + 320);
      if (cost < best) { best = cost; best_k = ks; }
    }
    rows_per_wg = ((n + best_k - 1) / best_k + gran - 1) / gran * gran;
  }
  const int64_t ksplit = (n + rows_per_wg - 1) / rows_per_wg;
  // XCD-aware slicing needs many workgroups per XCD to stay balanced; small grids keep the plain order
  const bool sliced = int64_t(ntiles) * ksplit >= 16 * int64_t(ncu);
  const bool xchunks = sliced && map_mode == 1 && ksplit % 8 == 0;
  const int per_xcd = xchunks ? -1 : (sliced ? (ntiles + 7) / 8 : 0);
  const int64_t nblocks = xchunks ? int64_t(ntiles) * ksplit : (sliced ? int64_t(8) * per_xcd * ksplit : int64_t(ntiles) * ksplit);
  if (nblocks > 0x7fffffffLL) fail(CCZ_EUNSUP, "gram: grid too large");

  RowPlan rp;
  rp.rows_per_wg = rows_per_wg;
  rp.ksplit = ksplit;
  rp.nblocks = nblocks;
  rp.per_xcd = per_xcd;
  rp.sliced = sliced;
  rp.fast = fast;
  return rp;
}

// pilot_mode (fp32 only): 0 = never, 1 = automatic (column sums and sums of squares are inspected on the host: one small
// read-back), 2 = always (device-side only, no host synchronisation).  Returns whether the pilot path ran.
template <typename T>
bool launch_moments(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, double* G, double* s, int64_t D,
                    bool time_it, void** tile_cache = nullptr, int pilot_mode = 0) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  constexpr bool is32 = sizeof(T) == 4;
  const int tile = is32 ? T32 : T64;
  const TileTable tt = build_tile_table<T>(c, views, n_views, tile_cache);
  GramTile* d_tiles = tt.dev;
  const int ntiles = tt.ntiles;
  const RowPlan rp = plan_rows<T>(c, views, n_views, n, ntiles, tt.fast);
  const bool fast = rp.fast;
  const int64_t rows_per_wg = rp.rows_per_wg, ksplit = rp.ksplit, nblocks = rp.nblocks;
  const int per_xcd = rp.per_xcd;
  const bool sliced = rp.sliced;
  const int ncu = std::max(1, im->props.multiProcessorCount);
  const size_t lds_bytes = size_t(2) * 2 * BK * tile * sizeof(T);  // 64 KiB either way
  // ---- column sums first: they are the means, and for fp32 views they decide (and define) the pilot shift ----
  // (Round 3 tried to hide this 5.4 ms HBM-bound pass under the MFMA-bound K1 on a second stream, with the pilot decided
  // from a strided sample: K1 then ran 7 ms LONGER -- the column-sum workgroups share the CUs' issue slots with K1's one
  // wave per SIMD -- so the pass stays in front.)
  static const int pilot_env = [] { const char* e = getenv("CCZ_GRAM_PILOT"); return e ? atoi(e) : -1; }();   // -1: caller's mode
  if (pilot_env == 0 || pilot_env == 1) pilot_mode = pilot_env == 1 ? 2 : 0;
  if (!is32) pilot_mode = 0;                                  // fp64 views accumulate in fp64: nothing to protect
  // arithmetic route of fp32 views (ccz_k1_route): the split-bf16 route always shifts by the pilot -- the subtraction rides
  // in its split pass for free and needs no read-back
  bool split = false;
  if (is32) {
    static const int route_env = [] {
      const char* e = getenv("CCZ_K1_ROUTE");
      if (!e) return 0;
      if (!strcmp(e, "fp32")) return int(CCZ_K1_FP32);
      if (!strcmp(e, "bf16x2")) return int(CCZ_K1_BF16X2);
      return 0;
    }();
    const int route = c->k1_route != CCZ_K1_AUTO ? c->k1_route : route_env;
    split = (route == CCZ_K1_BF16X2 || (route == CCZ_K1_AUTO && gram_split_worthwhile(n, D))) && n_views <= 16;
    if (split) pilot_mode = 2;
  }
  c->last_route = !is32 ? CCZ_K1_FP64 : (split ? CCZ_K1_BF16X2 : CCZ_K1_FP32);
  static const double pilot_thr = [] { const char* e = getenv("CCZ_GRAM_PILOT_RATIO"); return e ? atof(e) : 2.0; }();
  double* s_launch = s;            // column sums of THIS launch's rows (separate from the running sums in pilot modes)
  double* sq = nullptr;
  if (pilot_mode != 0) {
    s_launch = static_cast<double*>(dev_alloc(c, size_t(D) * 8 * (pilot_mode == 1 ? 2 : 1)));
    zero(c, s_launch, size_t(D) * 8 * (pilot_mode == 1 ? 2 : 1));
    if (pilot_mode == 1) sq = s_launch + D;
  }
  if (time_it) CCZ_HIP(hipEventRecord(im->ev[2], st));
  int64_t off = 0;
  float* pilot = nullptr;
  if (split) {
    // Split route: the exact column sums ride in the split pass (one read of the rows instead of two), so the pilot cannot be
    // their mean -- it is the mean of a strided sample of <= 2048 rows spread over the whole launch (sorted / drifting inputs
    // included).  Any pilot near the mean serves: the fix-up  sum x x' = sum (x-p)(x-p)' + p s' + s p' - n p p'  is exact in p.
    const int64_t nsamp = std::min<int64_t>(n, 2048), stride = n / nsamp;
    double* samp = static_cast<double*>(dev_alloc(c, size_t(D) * 8));
    zero(c, samp, size_t(D) * 8);
    for (int v = 0; v < n_views; ++v) {
      const int64_t colblocks = (views[v].cols + 255) / 256;
      int64_t rpb = 256;
      while (rpb > 16 && colblocks * ((nsamp + rpb - 1) / rpb) < 2 * int64_t(ncu)) rpb /= 2;
      dim3 grid((unsigned)colblocks, (unsigned)((nsamp + rpb - 1) / rpb));
      hipLaunchKernelGGL((k_colsum<T, false>), grid, dim3(256), 0, st, static_cast<const T*>(views[v].data), nsamp, views[v].cols,
                         views[v].ld * stride, samp + off, static_cast<double*>(nullptr), rpb);
      off += views[v].cols;
    }
    pilot = static_cast<float*>(dev_alloc(c, size_t(D) * 4));
    hipLaunchKernelGGL(k_pilot_from_sums, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, samp, D, 1.0 / double(nsamp), pilot);
    dev_free(c, samp);
  }
  for (int v = 0; v < n_views && !split; ++v) {
    // enough row blocks to cover the chip even for narrow / short views
    const int64_t colblocks = (views[v].cols + 255) / 256;
    int64_t rpb = 2048;
    while (rpb > 64 && colblocks * ((n + rpb - 1) / rpb) < 4 * int64_t(ncu)) rpb /= 2;
    dim3 grid((unsigned)colblocks, (unsigned)((n + rpb - 1) / rpb));
    if (sq)
      hipLaunchKernelGGL((k_colsum<T, true>), grid, dim3(256), 0, st, static_cast<const T*>(views[v].data), n, views[v].cols,
                         views[v].ld, s_launch + off, sq + off, rpb);
    else
      hipLaunchKernelGGL((k_colsum<T, false>), grid, dim3(256), 0, st, static_cast<const T*>(views[v].data), n, views[v].cols,
                         views[v].ld, s_launch + off, static_cast<double*>(nullptr), rpb);
    off += views[v].cols;
  }
  CCZ_LAUNCH_CHECK();
  if (time_it) CCZ_HIP(hipEventRecord(im->ev[3], st));
  bool use_pilot = pilot_mode == 2;
  if (pilot_mode == 1) {
    // largest |mean| / std over the columns: fp32 accumulation of raw products loses ~ eps32 sqrt(rows) (mean/std)^2 of
    // a covariance entry (6e-4 at ratio 10, garbage at 1000); the FIFO kernel is kept for ratio <= 2
    std::vector<double> hs(size_t(2) * D);
    d2h(c, hs.data(), s_launch, size_t(2) * D * 8);
    double worst = 0.0;
    for (int64_t j = 0; j < D; ++j) {
      const double mu = hs[j] / double(n), var = hs[D + j] / double(n) - mu * mu;
      if (!std::isfinite(mu) || !std::isfinite(var)) continue;          // NaN / inf inputs are reported by the caller
      const double sd = var > 0.0 ? std::sqrt(var) : 0.0;
      const double ratio = std::fabs(mu) <= pilot_thr * sd ? 0.0 : (sd > 0.0 ? std::fabs(mu) / sd : 1e300);
      if (ratio > worst) worst = ratio;
    }
    use_pilot = worst > pilot_thr;
  }
  if (use_pilot && !split) {
    pilot = static_cast<float*>(dev_alloc(c, size_t(D) * 4));
    hipLaunchKernelGGL(k_pilot_from_sums, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, s_launch, D, 1.0 / double(n), pilot);
  }

  if (time_it) CCZ_HIP(hipEventRecord(im->ev[0], st));
  static const int impl_sel = [] { const char* e = getenv("CCZ_GRAM_IMPL"); return e ? atoi(e) : 1; }();   // 1: wave-private FIFO (default), 0: register-staged shared tile
  // pilot-shifted data on a chip-filling grid: the FIFO kernel with the subtraction at the fragment read, on the rows
  // that form whole ring periods; the (< 32) rows left over go through the staged kernel, same pilot
  static const int fifo_pilot_env = [] { const char* e = getenv("CCZ_GRAM_FIFO_PILOT"); return e ? atoi(e) : 1; }();
  const int64_t n_main = n / (FB * FR) * (FB * FR);
  const bool fifo_pilot = is32 && fast && impl_sel != 0 && use_pilot && fifo_pilot_env != 0 && sliced && n_main >= rows_per_wg;
  // staged fp32 kernel on a small grid: per-(chunk, tile) partial sums + one reduce instead of contended atomics
  float* partial = nullptr;
  const bool staged32 = is32 && !(fast && impl_sel != 0 && !use_pilot) && !fifo_pilot;
  if (staged32 && ksplit >= 2 && !sliced && !split) {
    static const int64_t partial_cap = [] { const char* e = getenv("CCZ_GRAM_PARTIAL_MB"); return (e ? atoll(e) : 192LL) << 20; }();
    const int64_t bytes = ksplit * int64_t(ntiles) * T32 * T32 * 4;
    if (bytes <= partial_cap) partial = static_cast<float*>(dev_alloc(c, size_t(bytes)));
  }
  if (split) {
    gram_split_f32(c, views, n_views, n, G, D, pilot, s_launch, time_it);
  } else if (is32) {
    if (fifo_pilot) {
      const size_t fifo_bytes = size_t(4) * FR * FSLOT;
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32_fifo<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
      hipLaunchKernelGGL(k_gram_f32_fifo<true>, dim3((unsigned)nblocks), dim3(256), fifo_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n_main,
                         rows_per_wg, G, D, pilot);
      if (n_main < n) {
        CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
        hipLaunchKernelGGL(k_gram_f32<true>, dim3((unsigned)ntiles), dim3(256), lds_bytes, st, d_tiles, ntiles, 0, int64_t(1), n, int64_t(FB * FR), G, D,
                           pilot, static_cast<float*>(nullptr), n_main);
      }
    } else if (fast && impl_sel != 0 && !use_pilot) {
      const size_t fifo_bytes = size_t(4) * FR * FSLOT;   // 128 KiB: four wave-private rings
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32_fifo<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
      hipLaunchKernelGGL(k_gram_f32_fifo<false>, dim3((unsigned)nblocks), dim3(256), fifo_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D,
                         static_cast<const float*>(nullptr));
    } else if (fast) {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
      hipLaunchKernelGGL(k_gram_f32<true>, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D, pilot, partial, int64_t(0));
    } else {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
      hipLaunchKernelGGL(k_gram_f32<false>, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D, pilot, partial, int64_t(0));
    }
  } else {
    static const int impl64 = [] { const char* e = getenv("CCZ_GRAM64_IMPL"); return e ? atoi(e) : 1; }();   // 1: FIFO, 0: staged
    if (fast && impl64 != 0) {
      const size_t fifo_bytes = size_t(4) * FR * FSLOT;
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f64_fifo), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
      hipLaunchKernelGGL(k_gram_f64_fifo, dim3((unsigned)nblocks), dim3(256), fifo_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D);
    } else if (fast) {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f64<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
      hipLaunchKernelGGL(k_gram_f64<true>, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D);
    } else {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f64<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
      hipLaunchKernelGGL(k_gram_f64<false>, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, d_tiles, ntiles, per_xcd, ksplit, n, rows_per_wg, G, D);
    }
  }
  CCZ_LAUNCH_CHECK();
  if (partial) {
    hipLaunchKernelGGL(k_gram_reduce, dim3(256, (unsigned)ntiles), dim3(256), 0, st, partial, d_tiles, ntiles, ksplit, G, D);
    CCZ_LAUNCH_CHECK();
    dev_free(c, partial);
  }
  if (time_it) CCZ_HIP(hipEventRecord(im->ev[1], st));
  if (use_pilot) {
    if (D > 65535) fail(CCZ_EUNSUP, "gram: pilot fix-up supports D <= 65535");
    hipLaunchKernelGGL(k_pilot_fixup, dim3((unsigned)((D + 255) / 256), (unsigned)D), dim3(256), 0, st, G, D, s_launch, double(n), pilot);
  }
  if (s_launch != s) hipLaunchKernelGGL(k_vec_add, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, s, s_launch, D);
  CCZ_LAUNCH_CHECK();
  if (time_it) {
    CCZ_HIP(hipEventSynchronize(im->ev[1]));
    float g = 0.f, cs = 0.f;
    CCZ_HIP(hipEventElapsedTime(&g, im->ev[0], im->ev[1]));
    CCZ_HIP(hipEventElapsedTime(&cs, im->ev[2], im->ev[3]));
    c->last_gram_ms += g;
    c->last_colsum_ms += cs;
  }
  c->last_pilot = use_pilot ? 1 : 0;
  // scratch goes back to the handle's pool right away: the pool is stream-ordered (one stream per handle), so a later
  // allocation that reuses a block can only touch it after the kernels enqueued above
  if (pilot) dev_free(c, pilot);
  if (s_launch != s) dev_free(c, s_launch);
  return use_pilot;
}

// pinned bounce buffers + copy stream of the host-input pipeline; false if pinned memory is unavailable
bool ensure_pipe(ccz_ctx* c, size_t bytes) {
  Impl* im = impl(c);
  if (!im->copy_stream) {
    if (hipStreamCreateWithFlags(&im->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); im->copy_stream = nullptr; return false; }
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&im->pipe_ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  }
  for (int i = 0; i < 4; ++i) if (!im->pipe_ev[i]) return false;
  if (im->pin_cap >= bytes) return true;
  for (int i = 0; i < 2; ++i) {
    if (im->pin_buf[i]) (void)hipHostFree(im->pin_buf[i]);
    im->pin_buf[i] = nullptr;
  }
  im->pin_cap = 0;
  for (int i = 0; i < 2; ++i)
    if (hipHostMalloc(&im->pin_buf[i], bytes, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      for (int j = 0; j < 2; ++j) { if (im->pin_buf[j]) (void)hipHostFree(im->pin_buf[j]); im->pin_buf[j] = nullptr; }
      return false;
    }
  im->pin_cap = bytes;
  return true;
}

// rows [r0, r0 + rows) of every view, packed densely view after view into `dst`, by `nthreads` host threads
void pack_rows(char* dst, const ccz_view* views, int n_views, size_t es, int64_t r0, int64_t rows, int nthreads) {
  auto work = [&](int64_t a, int64_t b) {
    size_t off = 0;
    for (int v = 0; v < n_views; ++v) {
      const size_t rb = size_t(views[v].cols) * es, ldb = size_t(views[v].ld) * es;
      const char* src = static_cast<const char*>(views[v].data) + size_t(r0 + a) * ldb;
      char* d = dst + off + size_t(a) * rb;
      if (rb == ldb) memcpy(d, src, size_t(b - a) * rb);
      else for (int64_t r = a; r < b; ++r, src += ldb, d += rb) memcpy(d, src, rb);
      off += size_t(rows) * rb;
    }
  };
  nthreads = int(std::max<int64_t>(1, std::min<int64_t>(nthreads, rows / 64)));
  if (nthreads == 1) { work(0, rows); return; }
  std::vector<std::thread> pool;
  const int64_t per = (rows + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    const int64_t a = t * per, b = std::min<int64_t>(rows, a + per);
    if (a < b) pool.emplace_back(work, a, b);
  }
  for (auto& th : pool) th.join();
}

}  // namespace

bool colsum_pilot_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, int64_t D, double* colsum, float* pilot, double* zero_me,
                      int64_t nzero) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  if (n_views < 1 || n_views > 8) return false;
  const int ncu = std::max(1, im->props.multiProcessorCount);
  const int64_t colblocks = (D + 255) / 256;
  if (colblocks > 64) return false;
  int64_t rpb = 2048;
  while (rpb > 32 && colblocks * ((n + rpb - 1) / rpb) < int64_t(ncu)) rpb /= 2;
  const int64_t rowblocks = (n + rpb - 1) / rpb;
  // arrival counters of k_colsum_pilot: one block of 64 words per STREAM (like the chain kernel's sync block): two losses on
  // one handle enqueued on different streams may overlap, and shared counters would elect the wrong "last row block"
  unsigned* counters = nullptr;
  for (auto& e : im->colsum_sync)
    if (e.first == static_cast<void*>(st)) { counters = static_cast<unsigned*>(e.second); break; }
  if (!counters) {
    if (im->colsum_sync.size() >= 32) return false;         // (a caller cycling through streams: the general route)
    CCZ_HIP(hipMalloc(reinterpret_cast<void**>(&counters), 64 * sizeof(unsigned)));
    CCZ_HIP(hipMemsetAsync(counters, 0, 64 * sizeof(unsigned), st));
    im->colsum_sync.emplace_back(static_cast<void*>(st), static_cast<void*>(counters));
  }
  double* part = static_cast<double*>(dev_alloc(c, size_t(rowblocks) * D * 8));
  ColsumViews cv{};
  cv.m = n_views;
  cv.off[0] = 0;
  for (int v = 0; v < n_views; ++v) {
    cv.data[v] = static_cast<const float*>(views[v].data);
    cv.ld[v] = views[v].ld;
    cv.off[v + 1] = cv.off[v] + int(views[v].cols);
  }
  hipLaunchKernelGGL(k_colsum_pilot, dim3((unsigned)colblocks, (unsigned)rowblocks), dim3(256), 0, st, cv, n, D, rpb, part, counters, colsum, pilot,
                     zero_me, nzero);
  dev_free(c, part);                    // stream-ordered pool: reused only behind the kernel above
  return true;
}

// Loss fast path: the pilot-shifted batch Gram of a DCCA batch as per-(row chunk, tile) fp32 partial sums + the exact column
// sums, in TWO launches (k_colsum_pilot, k_gram_f32 on the views where they lie: no gathered copy, no atomics, no fills,
// no read-back).  The consumer (loss.hip: k_loss_prep_partials) adds the chunks up in fp64 and undoes the shift.
// Returns false -- nothing enqueued -- when the shape does not suit (a grid that fills the chip several times, or partial
// sums beyond the cap): the caller takes the general route through ccz_moments.
bool gram_partials_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, GramPartials* out) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  if (n_views < 1 || n_views > 8 || n < 2) return false;
  int64_t D = 0;
  for (int v = 0; v < n_views; ++v) D += views[v].cols;
  if (gram_partials_split_f32(c, views, n_views, n, out)) return true;      // DCCA batches from 4096 rows on: the split route
  const TileTable tt = build_tile_table<float>(c, views, n_views, nullptr);
  const RowPlan rp = plan_rows<float>(c, views, n_views, n, tt.ntiles, tt.fast);
  static const int64_t partial_cap = [] { const char* e = getenv("CCZ_GRAM_PARTIAL_MB"); return (e ? atoll(e) : 192LL) << 20; }();
  int64_t bytes = rp.ksplit * int64_t(tt.ntiles) * T32 * T32 * 4;
  if (rp.sliced || bytes > partial_cap) return false;
  const int ncu = std::max(1, im->props.multiProcessorCount);
  // ---- the FIFO kernel with a per-tile row split (k_gram_f32_fifo_small), when every chunk can be whole ring periods ----
  static const int fifo_env = [] { const char* e = getenv("CCZ_LOSS_K1_FIFO"); return e ? atoi(e) : 1; }();
  bool fifo_plan_ok = false;
  int fifo_wgs = 0;
  int* fifo_plan_dev = nullptr;
  if (fifo_env && rp.fast && n % (FB * FR) == 0 && tt.ntiles <= 64) {
    // the plan depends on (n, which tiles are diagonal, CUs) only: formed once, kept on the device with the handle (a training
    // loop meets the same batch shape every step -- the search below and a host -> device copy per call cost more than they saved)
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key ^= v; key *= 1099511628211ull; };
    mix(uint64_t(n)); mix(uint64_t(tt.ntiles)); mix(uint64_t(ncu));
    for (unsigned char dflag : tt.diag) mix(dflag);
    for (auto& e : im->k1_plans)
      if (e.key == key) { fifo_plan_dev = static_cast<int*>(e.dev); fifo_wgs = e.wgs; break; }
    if (!fifo_plan_dev) {
      int n_diag = 0;
      for (unsigned char dflag : tt.diag) n_diag += dflag ? 1 : 0;
      const int n_off = tt.ntiles - n_diag;
      const int64_t P = n / (FB * FR);                        // ring periods of the batch
      double best = 1e300;
      int64_t bo = 0, bd = 0;
      for (int64_t ko = (n_off ? 1 : 0); ko <= (n_off ? std::min<int64_t>(P, ncu) : 0); ++ko)
        for (int64_t kd = (n_diag ? 1 : 0); kd <= (n_diag ? std::min<int64_t>(P, ncu) : 0); ++kd) {
          if (n_off * ko + n_diag * kd > ncu) break;
          const double po = n_off ? double((P + ko - 1) / ko) : 0.0, pd = n_diag ? 0.625 * double((P + kd - 1) / kd) : 0.0;
          const double cost = std::max(po, pd) + 1.2;         // + the fixed cost of a workgroup (prologue, 256 KB of stores), in periods
          if (cost < best) { best = cost; bo = ko; bd = kd; }
        }
      if (best < 1e299 && im->k1_plans.size() < 64) {
        std::vector<int> plan_h(size_t(3) * tt.ntiles);
        int first = 0;
        for (int t = 0; t < tt.ntiles; ++t) {
          const int64_t kk = tt.diag[size_t(t)] ? bd : bo;
          const int64_t per = (P + kk - 1) / kk;
          const int wgs = int((P + per - 1) / per);
          plan_h[size_t(3 * t)] = first;
          plan_h[size_t(3 * t + 1)] = wgs;
          plan_h[size_t(3 * t + 2)] = int(per * FB * FR);
          first += wgs;
        }
        void* dp = nullptr;
        CCZ_HIP(hipMalloc(&dp, plan_h.size() * sizeof(int)));
        h2d_small(c, dp, plan_h.data(), plan_h.size() * sizeof(int));
        im->k1_plans.push_back({key, dp, first});
        fifo_plan_dev = static_cast<int*>(dp);
        fifo_wgs = first;
      }
    }
    if (fifo_plan_dev) {
      bytes = int64_t(fifo_wgs) * T32 * T32 * 4;
      fifo_plan_ok = bytes <= partial_cap;
    }
  }
  // column sums + pilot
  if ((D + 255) / 256 > 64) return false;
  out->colsum = static_cast<double*>(dev_alloc(c, size_t(D) * 8));
  out->pilot = static_cast<float*>(dev_alloc(c, size_t(D) * 4));
  out->partial = static_cast<float*>(dev_alloc(c, size_t(bytes)));
  out->tile_plan = fifo_plan_ok ? fifo_plan_dev : nullptr;   // owned by the handle
  if (!colsum_pilot_f32(c, views, n_views, n, D, out->colsum, out->pilot, nullptr, 0)) {
    gram_partials_release(c, out);
    return false;
  }
  const size_t lds_bytes = size_t(2) * 2 * BK * T32 * sizeof(float);
  if (fifo_plan_ok) {
    const size_t fifo_bytes = size_t(4) * FR * FSLOT;
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32_fifo_small<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(fifo_bytes)));
    hipLaunchKernelGGL(k_gram_f32_fifo_small<true>, dim3((unsigned)fifo_wgs), dim3(256), fifo_bytes, st, tt.dev, tt.ntiles, out->tile_plan, n,
                       out->partial, out->pilot);
  } else if (rp.fast) {
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
    hipLaunchKernelGGL(k_gram_f32<true>, dim3((unsigned)rp.nblocks), dim3(256), lds_bytes, st, tt.dev, tt.ntiles, rp.per_xcd, rp.ksplit, n, rp.rows_per_wg,
                       static_cast<double*>(nullptr), D, out->pilot, out->partial, int64_t(0));
  } else {
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_f32<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
    hipLaunchKernelGGL(k_gram_f32<false>, dim3((unsigned)rp.nblocks), dim3(256), lds_bytes, st, tt.dev, tt.ntiles, rp.per_xcd, rp.ksplit, n, rp.rows_per_wg,
                       static_cast<double*>(nullptr), D, out->pilot, out->partial, int64_t(0));
  }
  CCZ_LAUNCH_CHECK();
  out->tiles = tt.dev;
  out->ntiles = tt.ntiles;
  out->ksplit = rp.ksplit;
  c->last_pilot = 1;
  c->last_route = CCZ_K1_FP32;
  return true;
}

void gram_partials_release(ccz_ctx* c, GramPartials* gp) {
  if (gp->planes) dev_free(c, gp->planes);
  if (gp->msq) dev_free(c, const_cast<double*>(gp->msq));
  gp->planes = nullptr; gp->msq = nullptr;
  if (gp->partial) dev_free(c, gp->partial);
  if (gp->pilot) dev_free(c, gp->pilot);
  if (gp->colsum) dev_free(c, gp->colsum);
  gp->partial = nullptr; gp->pilot = nullptr; gp->colsum = nullptr; gp->tile_plan = nullptr;
}

void moments_impl(ccz_ctx* c, int dtype, const ccz_view* views, int n_views, int64_t n_rows, bool on_device,
                  double* moments, bool accumulate, int pilot_mode, bool time_it) {
  if (!views || n_views < 1 || !moments) fail(CCZ_EINVAL, "moments: null argument");
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "moments: dtype must be CCZ_F32 or CCZ_F64");
  if (n_rows < 0) fail(CCZ_EINVAL, "moments: negative row count");
  int64_t D = 0;
  for (int v = 0; v < n_views; ++v) {
    if (views[v].cols < 1 || views[v].ld < views[v].cols || (!views[v].data && n_rows > 0))
      fail(CCZ_EINVAL, "moments: view %d is malformed (cols=%lld ld=%lld)", v, (long long)views[v].cols, (long long)views[v].ld);
    D += views[v].cols;
  }
  double* G = moments;
  double* s = moments + D * D;
  if (!accumulate) zero(c, moments, size_t(D * D + D) * 8);
  c->last_gram_ms = 0.0;
  c->last_colsum_ms = 0.0;
  if (n_rows == 0) return;
  const size_t es = dtype == CCZ_F32 ? 4 : 8;
  if (on_device) {
    if (dtype == CCZ_F32) launch_moments<float>(c, views, n_views, n_rows, G, s, D, time_it, nullptr, pilot_mode);
    else launch_moments<double>(c, views, n_views, n_rows, G, s, D, time_it, nullptr, 0);
    return;
  }
  // Host-resident (pageable) views: a three-stage pipeline over ~1 GiB row chunks (tapering at the end),
  //   host threads pack chunk i+1 into a pinned bounce buffer  ||  DMA of chunk i (copy stream)  ||  K1 on chunk i-1,
  // two slots of pinned + device staging, events between the stages.  Small inputs, or a host that
  // refuses pinned memory, take the plain copy-then-compute loop.
  int64_t row_bytes = 0;
  for (int v = 0; v < n_views; ++v) row_bytes += views[v].cols * int64_t(es);
  const int64_t chunk_mb = [] { const char* e = getenv("CCZ_H2D_CHUNK_MB"); return e ? std::max<int64_t>(1, atoll(e)) : 1024LL; }();
  const int n_threads = [] {
    const char* e = getenv("CCZ_H2D_THREADS");
    if (e) return std::max(1, atoi(e));
    return int(std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
  }();
  int64_t chunk = std::max<int64_t>(64, (chunk_mb << 20) / row_bytes / 64 * 64);
  chunk = std::min(chunk, n_rows);
  const bool piped = n_rows * row_bytes >= (int64_t(64) << 20) && n_rows > chunk / 2 && ensure_pipe(c, size_t(chunk) * row_bytes);
  const int nslots = piped ? 2 : 1;
  Impl* im = impl(c);
  // PINNED sources (hipHostMalloc / hipHostRegister, torch's pin_memory()): the DMA engine reads the caller's pages
  // directly -- no packing pass through the bounce buffers, the host only enqueues
  bool src_pinned = piped;
  for (int v = 0; v < n_views && src_pinned; ++v) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, views[v].data) != hipSuccess) { (void)hipGetLastError(); src_pinned = false; break; }
    src_pinned = at.type == hipMemoryTypeHost;
    if (src_pinned) {   // ... for the whole extent that will be read
      const char* last = static_cast<const char*>(views[v].data) + (size_t(n_rows - 1) * views[v].ld + views[v].cols) * es - 1;
      hipPointerAttribute_t a2{};
      if (hipPointerGetAttributes(&a2, last) != hipSuccess) { (void)hipGetLastError(); src_pinned = false; }
      else src_pinned = a2.type == hipMemoryTypeHost;
    }
  }
  static const bool pinned_direct = [] { const char* e = getenv("CCZ_H2D_PINNED_DIRECT"); return !e || atoi(e) != 0; }();
  src_pinned = src_pinned && pinned_direct;
  std::vector<void*> stage(size_t(nslots) * n_views, nullptr);
  std::vector<ccz_view> dv(size_t(nslots) * n_views);
  void* tile_tab[2] = {nullptr, nullptr};
  auto cleanup = [&] {
    (void)hipStreamSynchronize(stream(c));
    if (piped) (void)hipStreamSynchronize(im->copy_stream);
    for (void* p : stage) if (p) dev_free(c, p);
    for (void* p : tile_tab) if (p) dev_free(c, p);
  };
  try {
    for (int sl = 0; sl < nslots; ++sl)
      for (int v = 0; v < n_views; ++v) {
        stage[sl * n_views + v] = dev_alloc(c, size_t(chunk) * views[v].cols * es);
        dv[sl * n_views + v] = ccz_view{stage[sl * n_views + v], views[v].cols, views[v].cols};
      }
    if (piped) {
      // pooled scratch is recycled in stream order on the handle's stream: make the copy stream a part of that order
      // before it writes into freshly pooled staging blocks
      CCZ_HIP(hipEventRecord(im->pipe_ev[2], stream(c)));
      CCZ_HIP(hipStreamWaitEvent(im->copy_stream, im->pipe_ev[2], 0));
    }
    int64_t ci = 0;
    // Streamed chunks ALWAYS take the pilot-shifted kernel (each chunk with the pilot of its own rows): it hides behind
    // the PCIe copy (47 GB/s against > 100 TF), needs no read-back, and a decision taken on the first chunk alone
    // would be wrong for data whose later rows drift away from zero (sorted / padded / time-ordered inputs).
    const int chunk_mode = pilot_mode == 1 ? 2 : pilot_mode;
    // The pipeline is bound by the copy; what it cannot hide is the K1 of the LAST chunk.  So the chunks taper towards
    // the end (each at most half of what is left, down to ~128 MiB): the drain shrinks from one full chunk's K1 to a
    // few milliseconds, while the bulk still moves in large chunks on which K1 runs at its full rate.
    const int64_t min_chunk = std::max<int64_t>(64, ((int64_t(128) << 20) / row_bytes) / 64 * 64);
    int64_t rows = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += rows, ++ci) {
      const int64_t left = n_rows - r0;
      rows = std::min(chunk, left);
      if (piped && left < 2 * chunk && left > min_chunk) rows = std::min(rows, std::max(min_chunk, (left / 2 + 63) / 64 * 64));
      const int sl = piped ? int(ci & 1) : 0;
      if (piped) {
        if (ci >= 2 && !src_pinned) CCZ_HIP(hipEventSynchronize(im->pipe_ev[sl]));   // DMA of chunk ci-2 has drained this bounce buffer
        if (!src_pinned) pack_rows(static_cast<char*>(im->pin_buf[sl]), views, n_views, es, r0, rows, n_threads);
        if (ci >= 2) CCZ_HIP(hipStreamWaitEvent(im->copy_stream, im->pipe_ev[2 + sl], 0));   // K1 of chunk ci-2 done with this staging slot
        size_t off = 0;
        for (int v = 0; v < n_views; ++v) {
          const size_t bytes = size_t(rows) * views[v].cols * es;
          if (src_pinned) {
            const char* src = static_cast<const char*>(views[v].data) + size_t(r0) * views[v].ld * es;
            if (views[v].ld == views[v].cols) CCZ_HIP(hipMemcpyAsync(stage[sl * n_views + v], src, bytes, hipMemcpyHostToDevice, im->copy_stream));
            else CCZ_HIP(hipMemcpy2DAsync(stage[sl * n_views + v], size_t(views[v].cols) * es, src, size_t(views[v].ld) * es,
                                          size_t(views[v].cols) * es, size_t(rows), hipMemcpyHostToDevice, im->copy_stream));
          } else {
            CCZ_HIP(hipMemcpyAsync(stage[sl * n_views + v], static_cast<char*>(im->pin_buf[sl]) + off, bytes, hipMemcpyHostToDevice, im->copy_stream));
          }
          off += bytes;
        }
        CCZ_HIP(hipEventRecord(im->pipe_ev[sl], im->copy_stream));
        CCZ_HIP(hipStreamWaitEvent(stream(c), im->pipe_ev[sl], 0));
      } else {
        for (int v = 0; v < n_views; ++v) {
          const char* src = static_cast<const char*>(views[v].data) + size_t(r0) * views[v].ld * es;
          CCZ_HIP(hipMemcpy2DAsync(stage[v], size_t(views[v].cols) * es, src, size_t(views[v].ld) * es,
                                   size_t(views[v].cols) * es, size_t(rows), hipMemcpyHostToDevice, stream(c)));
        }
        CCZ_HIP(hipStreamSynchronize(stream(c)));
      }
      if (dtype == CCZ_F32) {
        launch_moments<float>(c, &dv[sl * n_views], n_views, rows, G, s, D, false, &tile_tab[sl], chunk_mode);
      } else {
        launch_moments<double>(c, &dv[sl * n_views], n_views, rows, G, s, D, false, &tile_tab[sl], 0);
      }
      if (piped) CCZ_HIP(hipEventRecord(im->pipe_ev[2 + sl], stream(c)));
    }
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
}

}  // namespace ccz
