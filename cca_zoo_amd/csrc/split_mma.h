// Split-bf16 MFMA core shared by the Gram kernel (gram_split.hip) and the sample-side product (gemm_split.hip): an fp32
// operand d is carried as two bf16 planes d = hi + mid (+ lo, dropped), and a product of two such operands is
//   hi_A hi_B + hi_A mid_B + mid_A hi_B      (three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator).
// Both operands arrive as streams of 16 KiB blocks, one per k-step of 16:  [plane H | M][tile t: 8][k half h: 2][index 32][k: 8]
// bf16 -- 256 indices of the non-contracted dimension x 16 of the contracted one, every 1 KiB run one MFMA operand of one wave.
#pragma once

#include <mutex>

#include "hip_common.h"
#include "gram_map.h"

namespace ccz {

typedef float sp_v16f32 __attribute__((ext_vector_type(16)));
typedef float sp_v4f32 __attribute__((ext_vector_type(4)));
typedef __bf16 sp_v8bf16 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_v2bf16 __attribute__((ext_vector_type(2)));
typedef float sp_v2f32 __attribute__((ext_vector_type(2)));
typedef unsigned int sp_v4u32 __attribute__((ext_vector_type(4)));

constexpr int SP_T = 256;            // columns per panel / tile edge
constexpr int SP_K = 16;             // rows per k-step
constexpr int SP_PSTEP = 16384;      // bytes of one panel x one k-step: [plane 2][tile 8][half 2][col 32][k 8] bf16
constexpr int SP_PLANE = 8192;       // bytes of one plane of it
constexpr int SP_STAGE = 2 * SP_PSTEP;   // LDS slot: A panel | B panel
constexpr int SP_NST = 4;            // ring slots
constexpr int SP_RB = 512;           // rows per workgroup of the split pass

__device__ __forceinline__ unsigned sp_pack2(float a, float b) {
  const sp_v2f32 f = {a, b};
  const sp_v2bf16 h = __builtin_convertvector(f, sp_v2bf16);      // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, h);
}

typedef __attribute__((address_space(3))) void* sp_lds_ptr;

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per function and device: set it once per (kernel, device) instead of in
// front of every launch (a driver call per launch shows in a training step whose host thread is the bottleneck)
inline void sp_allow_lds(const void* fn, int device, int bytes) {
  struct Seen { const void* fn; int device; int bytes; };
  static Seen seen[64];
  static int nseen = 0;
  static std::mutex mu;                       // (handles of different threads share the table)
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nseen; ++i)
    if (seen[i].fn == fn && seen[i].device == device && seen[i].bytes >= bytes) return;
  CCZ_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (nseen < 64) seen[nseen++] = Seen{fn, device, bytes};
}

// one of the eight 1 KiB DMAs of this wave's quarter of a k-step
#define SP_DMA1(slot, soff, i_)                                                                                     \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (sp_lds_ptr)(wr_base + (slot) * SP_STAGE + (i_) * 1024), 16, voff,  \
                                           (soff) + (i_) * 1024, 0, 0)

// DMA of this wave's quarter (8 KiB: one plane of one panel) of k-step `soff / SP_PSTEP` into slot `slot`
#define SP_DMA(slot, soff)                                                                                          \
  do {                                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (sp_lds_ptr)(wr_base + (slot) * SP_STAGE + i_ * 1024), 16,   \
                                                 voff, (soff) + i_ * 1024, 0, 0);                                   \
  } while (0)

// The pipeline of one workgroup: acc (a 128 x 128 quadrant of the 256 x 256 tile, 4 x 4 MFMA tiles) += sum over `nsteps` k-steps of
//   hi_A' hi_B + hi_A' mid_B + mid_A' hi_B
// for an "A side" and a "B side" block stream (one 16 KiB block [H | M] per k-step each, consecutive in memory).  baseA /
// baseB point at the first block of this workgroup's range; k-steps past nsteps arrive as zeros (buffer range check).
// MFMA operands are swapped (A operand <- B side, B operand <- A side) so that a lane ends up with 4 CONSECUTIVE B-side
// indices j of ONE A-side index i: lane l, register r of tile (ti, tj):  i = 32 ti + (l & 31),  j = 32 tj + (r & 3) + 8 (r >> 2)
// + 4 (l >> 5) -- 16-byte stores along j in the epilogues.
// (qr, qc): the quadrant this wave computes (its DMA share is fixed by `wave`).  MODE restricts the MFMA tiles of the quadrant --
// the diagonal tiles of a Gram matrix (A side == B side) hold only three distinct quadrants, two of them symmetric:
//   0 all 16;  1 ti <= tj (a symmetric quadrant: 10 tiles);  2 ti in {0, 1};  3 ti in {2, 3} (two waves share quadrant (0, 1)).
// Every wave runs every barrier; a diagonal workgroup's k-step costs 30 MFMAs instead of 48.
template <int MODE>
__device__ __forceinline__ void split_mma_core(sp_v16f32 (&acc)[4][4], char* smem, const char* baseA, const char* baseB, int nsteps,
                                               int wave, int lane, int qr, int qc) {
  constexpr int TI0 = MODE == 3 ? 2 : 0, TI1 = MODE == 2 ? 2 : 4;
  // this wave's DMA share: plane (wave & 1) of the A side (waves 0, 1) or the B side (waves 2, 3)
  const __amdgpu_buffer_rsrc_t src =
      panel_rsrc((wave < 2 ? baseA : baseB) + (wave & 1) * SP_PLANE, int64_t(nsteps - 1) * SP_PSTEP + SP_PLANE);
  const int voff = lane * 16;
  char* wr_base = smem + wave * SP_PLANE;
  // fragment read bases: A side tiles 4 qr .. 4 qr + 3, B side tiles 4 qc .. 4 qc + 3
  const char* rdA = smem + lane * 16 + qr * 4096;
  const char* rdB = smem + SP_PSTEP + lane * 16 + qc * 4096;

  sp_v8bf16 ah[2][4], am[2][4], bh[2][4], bm[2][4];
  int soff = 0;
#pragma unroll
  for (int s = 0; s < SP_NST; ++s) {
    SP_DMA(s, soff);
    soff += SP_PSTEP;
  }
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ti >= TI0 && ti < TI1) {
      ah[0][ti] = *reinterpret_cast<const sp_v8bf16*>(rdA + ti * 1024);
      am[0][ti] = *reinterpret_cast<const sp_v8bf16*>(rdA + SP_PLANE + ti * 1024);
    }
    bh[0][ti] = *reinterpret_cast<const sp_v8bf16*>(rdB + ti * 1024);
    bm[0][ti] = *reinterpret_cast<const sp_v8bf16*>(rdB + SP_PLANE + ti * 1024);
  }

  const int nloop = (nsteps + SP_NST - 1) / SP_NST;
  for (int it = 0; it < nloop; ++it) {
#pragma unroll
    for (int u = 0; u < SP_NST; ++u) {
      const int cur = u & 1, nxt = cur ^ 1;
      const int nslot = (u + 1) % SP_NST;
      // step s = 4 it + u: its fragments are in set `cur`; steps s+1 .. s+3 are in flight / landed in the other slots
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* nA = rdA + nslot * SP_STAGE;
      const char* nB = rdB + nslot * SP_STAGE;
      // The step's other instructions -- NR = 2 NA + 8 fragment reads of step s + 1 and the 8 DMAs of step s + 4 (into the slot
      // step s was read from) -- ride ONE PER MFMA behind the first NR + 8 MFMAs: an in-order wave hides about five issue
      // slots per 32-cycle MFMA (MI355X_MICROARCH.md), and blocks of ten between MFMA groups kept the matrix pipe waiting.
      constexpr int NA = TI1 - TI0, NR = 2 * NA + 8;
      int fi = 0;                                   // filler index: a compile-time constant after unrolling
      auto filler = [&](int k) {
        // k < 16: reads and DMAs alternate (read k / 2, DMA k / 2); from 16 on: reads 8 .. NR - 1
        const bool is_dma = k < 16 && (k & 1);
        const int r = k < 16 ? (k >> 1) : k - 8;
        if (is_dma) {
          SP_DMA1(u, soff, k >> 1);
        } else if (r < NR) {
          // read order: A hi (NA), B hi (4), B mid (4), A mid (NA)
          if (r < NA) ah[nxt][TI0 + r] = *reinterpret_cast<const sp_v8bf16*>(nA + (TI0 + r) * 1024);
          else if (r < NA + 4) bh[nxt][r - NA] = *reinterpret_cast<const sp_v8bf16*>(nB + (r - NA) * 1024);
          else if (r < NA + 8) bm[nxt][r - NA - 4] = *reinterpret_cast<const sp_v8bf16*>(nB + SP_PLANE + (r - NA - 4) * 1024);
          else am[nxt][TI0 + r - NA - 8] = *reinterpret_cast<const sp_v8bf16*>(nA + SP_PLANE + (TI0 + r - NA - 8) * 1024);
        }
      };
#pragma unroll
      for (int g = 0; g < 3; ++g) {                 // 0: hi' hi, 1: hi' mid, 2: mid' hi
#pragma unroll
        for (int ti = TI0; ti < TI1; ++ti) {
#pragma unroll
          for (int tj = (MODE == 1 ? ti : 0); tj < 4; ++tj) {
            const sp_v8bf16 opB = g == 1 ? bm[cur][tj] : bh[cur][tj];
            const sp_v8bf16 opA = g == 2 ? am[cur][ti] : ah[cur][ti];
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opB, opA, acc[ti][tj], 0, 0, 0);
            if (fi < NR + 8) filler(fi);
            ++fi;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      soff += SP_PSTEP;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the run-ahead DMAs (zeros past the extent) must land before the LDS is handed on
}

}  // namespace ccz
