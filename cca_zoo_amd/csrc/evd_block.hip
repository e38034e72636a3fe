// Blocked Jacobi for the dense symmetric EVD / SVD seams above the one-workgroup kernels (d > 160).
//
// Reference call sites: numpy.linalg.svd behind svd_whiten (cca_zoo/_utils/_linalg.py:28-40), torch.linalg.eigh
// behind _inv_sqrtm (cca_zoo/deep/objectives.py:9-21, deep/_base.py:176-188) and _BatchWhiten
// (cca_zoo/deep/_dcca_noi.py:62-67), numpy.linalg.svd(cross_cov) (cca_zoo/linear/_rcca.py:97).
//
// Scalar cyclic Jacobi is a chain of ~ sweeps * d dependent rotation rounds whatever the blocking; what the
// blocking decides is where the O(d^3) of work per sweep runs.  Here a round of the OUTER tournament pairs the
// d / 32 column blocks (32 wide) into d / 64 disjoint pairs and splits into two launches:
//
//   k_bj_inner2  one workgroup per block pair: the pair's 64 x 64 symmetric matrix S (two-sided form: gathered from A;
//                one-sided form: the Gram matrix of the pair's 64 rows of W) lives in LDS as its packed lower
//                triangle; 32 rounds of 32 simultaneous rotations annihilate the CROSS block only (pairs (i, 32 +
//                (i + r) mod 32): every element pair of the two blocks exactly once) -- in the first outer round of a
//                sweep a full 63-round tournament instead, which also covers the pairs inside each block.  The
//                accumulated rotation R (64 x 64) lives in registers (cross-block rounds) or transposed in LDS (full
//                tournament) and is written out; a pair with nothing above the threshold rests (R = I, flag); no O(d) work here.
//   k_bj_apply   every other tile of the matrix takes  A_KL <- R_K' A_KL R_L  (two 64^3 products on
//                v_mfma_f64_16x16x4_f64, the second one with the first one's accumulators as its B fragments) and the
//                eigenvector rows  V'[K] <- R_K' V'[K]: all of the O(d^3) work is MFMA work on 64-wide tiles, one
//                launch per round, in place (block pairs are disjoint); tiles whose pairs both rest are skipped; on large
//                grids a workgroup walks a run of four tiles with the next operands prefetched into registers.
//
// A is kept in "block-upper" canonical form: 32 x 32 sub-block (x, y), x < y, is stored at its natural place, the
// mirrored one is never read or written (tiles transpose on load / store), diagonal sub-blocks are stored whole.
// One sweep = d / 32 - 1 rounds = one hipGraph, replayed per sweep; the host reads one rotation counter per sweep.
// The one-sided form (thin SVD: rows of W orthogonalised, ccz_gesvj) shares k_bj_inner2 and the row-tile half of
// k_bj_apply; its Gram blocks come from k_bj_gram (split over the row length, partials summed in a fixed order).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hip_common.h"
#include "jacobi_dev.h"

namespace ccz {

namespace {

constexpr int BJB = 32;                        // block edge
constexpr int BJP = 64;                        // pair edge
constexpr int BJ_NH = BJP * (BJP + 1) / 2;     // packed lower triangle of the pair matrix
constexpr int AP_SX = 66;                      // LDS row stride (doubles) for A-fragment reads  (= 2 mod 32)
constexpr int AP_SB = 80;                      // LDS row stride for B-fragment reads            (= 16 mod 32)
constexpr double BJ_EPS = 2.220446049250313e-16;

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

struct BjStatus {       // device-resident
  int rotations;        // of the sweep in flight
  int bad;              // non-finite input
  double hmax;          // max |A| (two-sided) / largest squared row norm (one-sided), as set by the prep kernels
  double maxoff;        // largest |s_pq| that was rotated in the sweep in flight (two-sided form)
  long long g0;         // fused form: global round index of round 0 of the sweep in flight (buffers rotate with it)
  long long direct_g;   // fused form: the global round whose pair kernels read their cross blocks directly (first round
                        // of a solve, first round after a refresh: no previous round to re-apply)
};

__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {   // v >= 0 or non-finite (NaN -> inf)
  if (!(v <= 1.79769313486231570e308)) v = __builtin_inf();
  atomicMax(reinterpret_cast<unsigned long long*>(addr), static_cast<unsigned long long>(__double_as_longlong(v)));
}

template <int FULL>
__device__ __forceinline__ void bj_pair_c(int r, int k, int& p, int& q) {
  if (FULL) { pair_of(r, k, BJP - 1, p, q); return; }
  p = k;
  q = BJB + ((k + r) & (BJB - 1));
}

// MODE 0: two-sided; S gathered from the canonical symmetric A, written back; rotate when |s_pq| > eps * max|A|
// MODE 1: one-sided; S = sum of the `nsplit` Gram partials of this pair, not written back; rotate when
//         |g_pq| > tol * sqrt(g_pp g_qq) and both squared norms exceed the floor (numerically zero rows rest)
// MODE 2: MODE 0 on three rotating planes of A (the two-stream schedule of syev_block, off by default)
// ---- the pair kernel ---------------------------------------------------------------------------------------------
// The first form of this kernel (round 4, in the history) spent a round on  [all of S rotated | barrier | 3 LDS reads + the
// (c, s) chain | barrier]: 25 us for a pass at rest, 35-41 us busy.  A dependent fp64 operation costs 32 cycles on gfx950
// (profiles/r02c_clock_probe.md), so the chain IS the round.  Here the 32 parameter lanes never wait for the bulk update:
//   * lane k carries (h_pp, h_qq, h_pq) of ITS pair in registers.  The next round's pair k takes its two indices from
//     fixed (pair, member) sources -- the tournament only shifts positions -- so its new diagonal entries are the
//     sources' rotated diagonals (h_pp - t h_pq, h_qq + t h_pq: shuffled from the source lanes) and its new off-diagonal
//     entry is one element of the rotated 2 x 2 block (source pair A, source pair B) of the CURRENT S: four LDS reads
//     that depend on nothing computed this round (issued first), the sources' (c, s) by shuffle;
//   * S is double-buffered in LDS: waves 1..9 rotate its 528 blocks (the rotated pairs' own blocks included: with the exact
//     (c, s), so the transformation is an exact similarity whatever the angle's precision) from S[cur] into S[nxt] and
//     waves 10..13 (cross-block rounds: R in registers) or 10..15 (full tournament: R in LDS) rotate R while lane k
//     computes the next (c, s);
//   * ONE barrier per round; the loop-carried chain is  t (fp32: the angle needs no more -- a float32-accurate rotation
//     leaves a 1e-7 remainder that the next sweep removes) -> c = (1 + t^2)^-1/2 (fp32 seed, one third-order step in
//     fp64: c^2 + s^2 = 1 to rounding) -> s = t c -> four multiply-adds for the next h_pq: ~13 dependent fp64 operations
//     instead of ~26 plus two barriers and an LDS round trip.
constexpr int BJ2_THREADS = 1024;              // wave 0: parameters; waves 1..9: the 528 blocks of S; waves 10..15: R

struct Rot { double c, s, t; };
template <int MODE>
__device__ __forceinline__ Rot bj_rotation(double hpp, double hqq, double hpq, double half_ih, double ih) {
  double al = (hqq - hpp) * half_ih, hq = hpq * ih;
  if (MODE == 1) {                                                  // Gram entries span the whole exponent range: rescale
    const int e = max(__builtin_amdgcn_frexp_exp(al), __builtin_amdgcn_frexp_exp(hq));
    al = __builtin_amdgcn_ldexp(al, -e);
    hq = __builtin_amdgcn_ldexp(hq, -e);
  }
  const float a = fabsf(float(al)), h = float(hq);
  const float r = __builtin_amdgcn_sqrtf(fmaf(a, a, h * h));
  float tf = h * __builtin_amdgcn_rcpf(a + r);
  tf = al >= 0.0 ? tf : -tf;
  const float y0f = __builtin_amdgcn_rsqf(fmaf(tf, tf, 1.0f));
  const double t = double(tf), x = fma(t, t, 1.0), y0 = double(y0f);
  const double hh = fma(-x, y0 * y0, 1.0);                          // 1 - x y0^2 ~ 1e-7
  const double q = fma(0.375, hh, 0.5) * hh;
  Rot o;
  o.c = fma(y0, q, y0);                                             // y0 (1 + hh / 2 + 3 hh^2 / 8): error ~ hh^3
  o.s = t * o.c;
  o.t = t;
  return o;
}

// FULL (compile time: the tournament's index arithmetic is a large share of every wave's instruction stream, and the CU
// is VALU-issue bound -- 16 waves on 4 SIMDs, measured with the stamps below): 1 = 63-round full tournament, 0 = cross block.
template <int MODE, int FULL>
__global__ __launch_bounds__(BJ2_THREADS) void k_bj_inner2(double* __restrict__ A, int64_t lda, const double* __restrict__ G,
                                                           int nsplit, double* __restrict__ Rt_out, BjStatus* __restrict__ st,
                                                           int nb, int round, double tol, long long* __restrict__ dbg,
                                                           int* __restrict__ ident) {
  __shared__ __attribute__((aligned(16))) double Hs[2][BJ_NH];
  __shared__ __attribute__((aligned(16))) double Rt[BJP * BJP];     // Rt[j][i] = R[i][j]
  __shared__ __attribute__((aligned(16))) jac_cs csn[2][BJB];
  __shared__ __attribute__((aligned(16))) double Xs[MODE == 2 ? BJP * AP_SX + 2 * BJB * AP_SX : 2];   // fused form: old tile + two R parts
  __shared__ int src_pair[4];                                       // fused form: (pair, member) of ba and of bb in the previous round
  __shared__ __attribute__((aligned(16))) double Xs_mail[2 * 4 * 64];  // cross-block rounds: the R column each wave hands to its neighbour
  const int tid = threadIdx.x;
  int ba, bb;
  pair_of(round, blockIdx.x, nb - 1, ba, bb);
  // fused form (MODE 2): three rotating copies of A (state g in plane g % 3) and two of the rotations (plane g & 1)
  const int64_t plane = lda * lda;
  const long long g = MODE == 2 ? st->g0 + round : 0;
  double* A_nxt = MODE == 2 ? A + ((g + 1) % 3) * plane : A;
  if (MODE == 2) A += (g % 3) * plane;
  if (MODE == 2) Rt_out += (g & 1) * int64_t(nb >> 1) * (BJP * BJP);
  // CCZ_BJ_DEBUG: shader-clock stamps of workgroup 0 (thread 0: the parameter wave; thread 64: an S wave; thread 640: an R wave)
  const bool stamp = dbg && blockIdx.x == 0 && (tid == 0 || tid == 64 || tid == 640);
  long long* my_dbg = dbg ? dbg + (tid == 0 ? 0 : (tid == 64 ? 64 : 128)) : nullptr;
  int nstamp = 0;
  auto mark = [&]() { if (stamp && nstamp < 64) my_dbg[nstamp++] = __builtin_readcyclecounter(); };
  mark();
  const double hmax = st->hmax;
  // ---- load S (coalesced: the two diagonal sub-blocks and the canonical cross block) ----
  const bool refuse = MODE == 2 && g != st->direct_g;               // the cross block is re-derived from the previous round
  if (MODE != 1) {
    const int lo = min(ba, bb), hi = max(ba, bb), offlo = lo == ba ? 0 : BJB, offhi = hi == ba ? 0 : BJB;
    for (int e = tid; e < 3 * BJB * BJB; e += BJ2_THREADS) {
      const int blk = e >> 10, r = (e >> 5) & (BJB - 1), cc = e & (BJB - 1);
      if (blk == 2) {
        if (!refuse) Hs[0][tri_off(offlo + r, offhi + cc)] = A[(int64_t(lo) * BJB + r) * lda + hi * BJB + cc];
      } else if (cc <= r) {
        const int b = blk == 0 ? ba : bb, off = blk * BJB;
        Hs[0][tri_off(off + r, off + cc)] = A[(int64_t(b) * BJB + r) * lda + b * BJB + cc];
      }
    }
  } else {
    for (int e = tid; e < BJ_NH; e += BJ2_THREADS) {
      int i, j;
      tri_decode(e, i, j);
      double h = 0.0;
      const double* gp = G + int64_t(blockIdx.x) * nsplit * (BJP * BJP) + i * BJP + j;
      for (int s = 0; s < nsplit; ++s) h += gp[int64_t(s) * (BJP * BJP)];
      Hs[0][e] = h;
    }
  }
  if (MODE == 2 && refuse) {
    // The pair's cross block after the PREVIOUS round is one 32 x 32 piece of that round's tile (pair of ba, pair of bb):
    //   S_ab = Rt_Ka[rows of ba] . X . Rt_Kb[rows of bb]'   with X the old tile (state g - 1) -- two small products on
    // the matrix pipe here, so that this kernel depends on the previous PAIR kernel only and the previous round's
    // O(d^2) tile update (k_bj_apply, on a second stream) runs underneath it instead of in front of it.
    const int np_ = nb >> 1, rp = round == 0 ? nb - 2 : round - 1;
    const long long go = g - 1;
    const double* Ao = (A - (g % 3) * plane) + (go % 3) * plane;
    const double* Rp = (Rt_out - (g & 1) * int64_t(np_) * (BJP * BJP)) + (go & 1) * int64_t(np_) * (BJP * BJP);
    if (tid < np_) {
      int x, y;
      pair_of(rp, tid, nb - 1, x, y);
      if (x == ba) { src_pair[0] = tid; src_pair[1] = 0; }
      if (y == ba) { src_pair[0] = tid; src_pair[1] = 1; }
      if (x == bb) { src_pair[2] = tid; src_pair[3] = 0; }
      if (y == bb) { src_pair[2] = tid; src_pair[3] = 1; }
    }
    __syncthreads();
    const int Ka = src_pair[0], ia = src_pair[1], Kb = src_pair[2], ib = src_pair[3];
    int xb[2], yb[2];
    pair_of(rp, Ka, nb - 1, xb[0], xb[1]);
    pair_of(rp, Kb, nb - 1, yb[0], yb[1]);
    double* PA = Xs + BJP * AP_SX;
    double* PB = PA + BJB * AP_SX;
    for (int e = tid; e < BJP * BJP / 2; e += BJ2_THREADS) {        // the old tile, 16-byte pieces, mirrored blocks transposed
      const int sblk = e >> 9, piece = e & 511, row = piece >> 4, c2 = (piece & 15) * 2;
      const int si = sblk >> 1, sj = sblk & 1, x = xb[si], y = yb[sj];
      const bool direct = x < y;
      const double* base = direct ? Ao + int64_t(x) * BJB * lda + y * BJB : Ao + int64_t(y) * BJB * lda + x * BJB;
      const v2f64 v = *reinterpret_cast<const v2f64*>(base + int64_t(row) * lda + c2);
      if (direct) {
        *reinterpret_cast<v2f64*>(Xs + (BJB * si + row) * AP_SX + BJB * sj + c2) = v;
      } else {
        Xs[(BJB * si + c2) * AP_SX + BJB * sj + row] = v.x;
        Xs[(BJB * si + c2 + 1) * AP_SX + BJB * sj + row] = v.y;
      }
    }
    for (int e = tid; e < 2 * BJB * BJP / 2; e += BJ2_THREADS) {    // 32 rows of each of the two rotations
      const int which = e >> 10, piece = e & 1023, row = piece >> 5, c2 = (piece & 31) * 2;
      const double* src = Rp + int64_t(which ? Kb : Ka) * (BJP * BJP) + (BJB * (which ? ib : ia) + row) * BJP + c2;
      *reinterpret_cast<v2f64*>((which ? PB : PA) + row * AP_SX + c2) = *reinterpret_cast<const v2f64*>(src);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double* Ys = Rt;                                                // 64 x 32, stride 48 (B-fragment reads); Rt is set up later
    if (wave < 8) {                                                 // Y = X Rt_Kb[rows of bb]'  (64 x 32)
      const int mt = wave >> 1, nt = wave & 1;
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const double a = Xs[(16 * mt + l15) * AP_SX + 4 * ks + l4];
        const double b = PB[(16 * nt + l15) * AP_SX + 4 * ks + l4];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Ys[(16 * mt + l4 + 4 * r) * 48 + 16 * nt + l15] = acc[r];
    }
    __syncthreads();
    if (wave < 4) {                                                 // S_ab = Rt_Ka[rows of ba] Y  (32 x 32)
      const int it = wave >> 1, jt = wave & 1;
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const double a = PA[(16 * it + l15) * AP_SX + 4 * ks + l4];
        const double b = Ys[(4 * ks + l4) * 48 + 16 * jt + l15];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Hs[0][tri_off(BJB + 16 * jt + l15, 16 * it + l4 + 4 * r)] = acc[r];
    }
    __syncthreads();
  }
  if (!(hmax > 0.0) || !(hmax < __builtin_inf())) {                 // zero or non-finite input: identity, nothing rotates
    for (int e = tid; e < BJP * BJP; e += BJ2_THREADS) Rt_out[int64_t(blockIdx.x) * (BJP * BJP) + e] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
    if (tid == 0 && !(hmax < __builtin_inf())) st->bad = 1;
    if (MODE == 2) {                                                // the next state's plane still gets this pair's blocks
      __syncthreads();
      const int lo = min(ba, bb), hi = max(ba, bb), offlo = lo == ba ? 0 : BJB, offhi = hi == ba ? 0 : BJB;
      for (int e = tid; e < 3 * BJB * BJB; e += BJ2_THREADS) {
        const int blk = e >> 10, r = (e >> 5) & (BJB - 1), cc = e & (BJB - 1);
        if (blk == 2) A_nxt[(int64_t(lo) * BJB + r) * lda + hi * BJB + cc] = Hs[0][tri_off(offlo + r, offhi + cc)];
        else { const int b = blk == 0 ? ba : bb, off = blk * BJB; A_nxt[(int64_t(b) * BJB + r) * lda + b * BJB + cc] = Hs[0][tri_off(off + r, off + cc)]; }
      }
    }
    return;
  }
  for (int e = tid; e < BJP * BJP; e += BJ2_THREADS) Rt[e] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
  const double thr = MODE != 1 ? BJ_EPS * hmax : 0.0, ih = 1.0 / hmax, half_ih = 0.5 * ih, floor2 = hmax * 1e-28;
  constexpr int nr = FULL ? BJP - 1 : BJB;
  if (MODE != 2) {
    // Nothing to rotate in this pair (the rule in the last sweeps, and all of the final, verifying one)?  Then R = I, S
    // stays as it is, and the tile update skips every tile whose two pairs are both at rest (ident[]).
    __syncthreads();
    int live = 0;
    for (int e = tid; e < (FULL ? BJP * (BJP - 1) / 2 : BJB * BJB); e += BJ2_THREADS) {
      int i, j;
      if (FULL) { tri_decode(e, i, j); ++i; }                       // i > j: every pair of the 64
      else { i = BJB + (e >> 5); j = e & (BJB - 1); }               // the cross block
      const double hpq = Hs[0][tri_off(i, j)];
      if (MODE == 0) live |= fabs(hpq) > thr;
      else { const double hpp = Hs[0][tri_off(j, j)], hqq = Hs[0][tri_off(i, i)]; live |= hpp > floor2 && hqq > floor2 && hpq * hpq > tol * tol * hpp * hqq; }
    }
    live = __syncthreads_or(live);
    if (tid == 0) ident[blockIdx.x] = live ? 0 : 1;
    if (!live) {
      v2f64* ro = reinterpret_cast<v2f64*>(Rt_out + int64_t(blockIdx.x) * (BJP * BJP));
      for (int e = tid; e < BJP * BJP / 2; e += BJ2_THREADS) ro[e] = reinterpret_cast<const v2f64*>(Rt)[e];
      return;
    }
  }
  // roles.  fp64 FMAs pipeline inside a wave (tools/probes/dp_issue_probe.hip: 2.4 cycles per instruction per SIMD with 4 waves x 8
  // independent chains; 32-44 cycles for a dependent one), so only the parameter lanes' chain is latency-bound; the other
  // waves are bound by what the CU ISSUES per round -- VALU index arithmetic and, above all, LDS traffic (a ds_write_b64
  // costs ~6 LDS cycles): shader-clock stamps (CCZ_BJ_DEBUG) put the R waves of the LDS-resident form at 1850-2400 cycles per
  // round against 1150 for the parameter chain.  Hence the tournament type as a template parameter (index arithmetic at
  // compile time) and, for the cross-block rounds (all but the first round of a sweep), R in REGISTERS (below).
  const int sb = tid - 64;                                          // S block index on waves 1..9 (diagonal blocks included)
  int ka = 0, kb = 0;
  const bool son = sb >= 0 && sb < BJB * (BJB + 1) / 2;
  if (son) tri_decode(sb, ka, kb);                                  // ka >= kb
  const int o0_cross = tri_off(ka, kb);
  const int rt0 = tid - 640;                                        // R rotations rt0 + 384 j (pair = id >> 6, row = id & 63)
  int rk6[6], rrow6[6];
  bool rok6[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int id = min(max(rt0, 0) + 384 * j, BJP * BJB - 1);
    rok6[j] = rt0 >= 0 && rt0 + 384 * j < BJP * BJB;
    rk6[j] = id >> 6;
    rrow6[j] = id & 63;
  }
  // parameter lanes: where next round's pair k takes its two indices from (pair, member), fixed for the whole pass
  int srcA = 0, srcB = 0;
  bool memA = false, memB = false;
  if (tid < BJB) {
    const int k = tid;
    if (!FULL) { srcA = k; memA = false; srcB = (k + 1) & (BJB - 1); memB = true; }
    else if (k == 0) { srcA = 0; memA = false; srcB = 1; memB = false; }
    else {
      if (k <= BJB - 2) { srcA = k + 1; memA = false; } else { srcA = BJB - 1; memA = true; }
      if (k >= 2) { srcB = k - 1; memB = true; } else { srcB = 0; memB = true; }
    }
  }
  // Cross-block rounds: R lives in the registers of waves 10..13 -- lane = row, wave w holds the I-half columns 8 w .. 8 w + 7
  // (rx) and the eight J-half columns currently paired with them (ry).  Round r pairs column k with 32 + (k + r) mod 32:
  // from one round to the next the J columns move by ONE position, i.e. ry[j] <- ry[j + 1] inside the wave and one column
  // per wave crosses to the neighbour through a 2 KB LDS mailbox.  No R traffic in LDS at all: one uniform (c, s) read per
  // rotation, one 8-byte hand-off per lane and round.
  const int rw = (tid >> 6) - 10, rrow = tid & 63;
  const bool r_regs = !FULL && rw >= 0 && rw < 4;
  double rx[8], ry[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { rx[j] = (rrow == 8 * rw + j) ? 1.0 : 0.0; ry[j] = (rrow == BJB + 8 * rw + j) ? 1.0 : 0.0; }
  double* mail = Xs_mail;                                           // [2][4][64]
  double hpp = 0.0, hqq = 0.0, hpq = 0.0, my_max = 0.0;
  Rot cur_rot{1.0, 0.0, 0.0};
  int my_rot = 0;
  auto decide = [&]() {                                             // lanes 0..31: (hpp, hqq, hpq) -> cur_rot
    bool go;
    if (MODE != 1) go = fabs(hpq) > thr;
    else go = hpp > floor2 && hqq > floor2 && hpq * hpq > tol * tol * hpp * hqq;
    cur_rot = Rot{1.0, 0.0, 0.0};
    if (go) {
      cur_rot = bj_rotation<(MODE == 1 ? 1 : 0)>(hpp, hqq, hpq, half_ih, ih);
      ++my_rot;
      my_max = fmax(my_max, fabs(hpq));
    }
  };
  __syncthreads();
  if (tid < BJB) {
    int a, b;
    bj_pair_c<FULL>(0, tid, a, b);
    hpq = Hs[0][tri_off(a, b)]; hqq = Hs[0][tri_off(b, b)]; hpp = Hs[0][tri_off(a, a)];
    decide();
    csn[0][tid] = jac_cs{cur_rot.c, cur_rot.s};
  }
  __syncthreads();
  mark();
  for (int r = 0; r < nr; ++r) {
    const int cb_ = r & 1;
    const double* Sc = Hs[cb_];
    double* Sn = Hs[cb_ ^ 1];
    if (tid < 64) {
      if (tid < BJB) {
        // the four cells of (source pair A, source pair B) in the current S: independent of this round's arithmetic
        int pA, qA, pB, qB, p, q;
        bj_pair_c<FULL>(r, srcA, pA, qA);
        bj_pair_c<FULL>(r, srcB, pB, qB);
        bj_pair_c<FULL>(r, tid, p, q);
        const double m0 = Sc[tri_off(pA, pB)], m1 = Sc[tri_off(pA, qB)], m2 = Sc[tri_off(qA, pB)], m3 = Sc[tri_off(qA, qB)];
        // this round's own pair as STORED: the next angle's diagonal estimates need t only (the stored cells themselves
        // are rotated with the exact (c, s) by the S waves, like every other block)
        const double app = Sc[tri_off(p, p)], aqq = Sc[tri_off(q, q)], apq = Sc[tri_off(p, q)];
        const double c = cur_rot.c, s = cur_rot.s, t = cur_rot.t;
        const double dpp = fma(-t, apq, app), dqq = fma(t, apq, aqq);
        if (r + 1 < nr) {
          const double dppA = __shfl(dpp, srcA, 64), dqqA = __shfl(dqq, srcA, 64);
          const double dppB = __shfl(dpp, srcB, 64), dqqB = __shfl(dqq, srcB, 64);
          const double cA = __shfl(c, srcA, 64), sA = __shfl(s, srcA, 64);
          const double cB = __shfl(c, srcB, 64), sB = __shfl(s, srcB, 64);
          const double b0 = memB ? sB : cB, b1 = memB ? cB : -sB;
          const double a0 = memA ? sA : cA, a1 = memA ? cA : -sA;
          const double u0 = fma(b0, m0, b1 * m1), u1 = fma(b0, m2, b1 * m3);
          hpq = fma(a0, u0, a1 * u1);
          hpp = memA ? dqqA : dppA;
          hqq = memB ? dqqB : dppB;
          decide();
          csn[cb_ ^ 1][tid] = jac_cs{cur_rot.c, cur_rot.s};
        }
      }
    } else if (tid < 640) {
      if (son) {
        int p1, q1, p2, q2;
        bj_pair_c<FULL>(r, ka, p1, q1);
        bj_pair_c<FULL>(r, kb, p2, q2);
        int o0, o1, o2, o3;
        if (FULL) {
          o0 = tri_off(p1, p2); o1 = tri_off(p1, q2); o2 = tri_off(q1, p2); o3 = tri_off(q1, q2);
        } else {                                                    // cross block: p < 32 <= q, ka >= kb -- no max / min needed
          o0 = o0_cross;
          o1 = (__mul24(q2, q2 + 1) >> 1) + p1;
          o2 = (__mul24(q1, q1 + 1) >> 1) + p2;
          o3 = tri_off(q1, q2);
        }
        const jac_cs ra = csn[cb_][ka], rb = csn[cb_][kb];
        const double m0 = Sc[o0], m1 = Sc[o1], m2 = Sc[o2], m3 = Sc[o3];
        const double ca = ra.x, sa = ra.y, cb = rb.x, sb2 = rb.y;
        const double n00 = cb * m0 - sb2 * m1, n01 = sb2 * m0 + cb * m1;
        const double n10 = cb * m2 - sb2 * m3, n11 = sb2 * m2 + cb * m3;
        // (ka == kb: o1 == o2 is one cell, written twice with the same value up to rounding -- the rotated pair's own
        // off-diagonal element, which a float32-accurate angle leaves at ~1e-7 of its size, NOT at zero)
        Sn[o0] = ca * n00 - sa * n10;
        Sn[o2] = sa * n00 + ca * n10;
        Sn[o1] = ca * n01 - sa * n11;
        Sn[o3] = sa * n01 + ca * n11;
      }
    } else if (!FULL) {
      if (r_regs) {
        if (r > 0) ry[7] = mail[(((r - 1) & 1) * 4 + ((rw + 1) & 3)) * 64 + rrow];    // the neighbour's hand-off of the last round
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const jac_cs cs = csn[cb_][8 * rw + j];
          const double x = rx[j], y = ry[j];
          rx[j] = cs.x * x - cs.y * y;
          ry[j] = cs.y * x + cs.x * y;
        }
        mail[((r & 1) * 4 + rw) * 64 + rrow] = ry[0];
#pragma unroll
        for (int j = 0; j < 7; ++j) ry[j] = ry[j + 1];
      }
    } else {
      // all loads first (two LDS latencies for the six rotations, not twelve), then the arithmetic, then the stores; the
      // rotation's pair index and row are fixed per thread (rk6 / rrow6 below), only the pair's two columns move with r
      int op[6], oq[6];
      jac_cs cs6[6];
      double x6[6], y6[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        int p, q;
        bj_pair_c<FULL>(r, rk6[j], p, q);
        op[j] = (p << 6) + rrow6[j];
        oq[j] = (q << 6) + rrow6[j];
        cs6[j] = csn[cb_][rk6[j]];
        x6[j] = Rt[op[j]];
        y6[j] = Rt[oq[j]];
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (rok6[j] && cs6[j].y != 0.0) {
          Rt[op[j]] = cs6[j].x * x6[j] - cs6[j].y * y6[j];
          Rt[oq[j]] = cs6[j].y * x6[j] + cs6[j].x * y6[j];
        }
    }
    if (r < 12) mark();                                             // own work of the round done
    lds_barrier();
    if (r < 12) mark();                                             // barrier passed
  }
  __syncthreads();
  mark();
  const double* Sf = Hs[nr & 1];
  if (MODE != 1) {                                                  // (fused form: into the NEXT state's plane)
    const int lo = min(ba, bb), hi = max(ba, bb), offlo = lo == ba ? 0 : BJB, offhi = hi == ba ? 0 : BJB;
    for (int e = tid; e < 3 * BJB * BJB; e += BJ2_THREADS) {
      const int blk = e >> 10, r = (e >> 5) & (BJB - 1), cc = e & (BJB - 1);
      if (blk == 2) {
        A_nxt[(int64_t(lo) * BJB + r) * lda + hi * BJB + cc] = Sf[tri_off(offlo + r, offhi + cc)];
      } else {
        const int b = blk == 0 ? ba : bb, off = blk * BJB;
        A_nxt[(int64_t(b) * BJB + r) * lda + b * BJB + cc] = Sf[tri_off(off + r, off + cc)];
      }
    }
  }
  if (!FULL) {
    // after 32 rounds the J columns are back home, except for the last hand-off that is still in the mailbox
    if (r_regs) {
      ry[7] = mail[(((nr - 1) & 1) * 4 + ((rw + 1) & 3)) * 64 + rrow];
      double* ro = Rt_out + int64_t(blockIdx.x) * (BJP * BJP);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ro[(8 * rw + j) * BJP + rrow] = rx[j];                      // Rt[col][row]
        ro[(BJB + 8 * rw + j) * BJP + rrow] = ry[j];
      }
    }
  } else {
    v2f64* ro = reinterpret_cast<v2f64*>(Rt_out + int64_t(blockIdx.x) * (BJP * BJP));
    for (int e = tid; e < BJP * BJP / 2; e += BJ2_THREADS) ro[e] = reinterpret_cast<const v2f64*>(Rt)[e];
  }
  if (tid < 64) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      my_max = fmax(my_max, __shfl_xor(my_max, o, 64));
      my_rot += __shfl_xor(my_rot, o, 64);
    }
    if (tid == 0 && my_rot > 0) {
      atomicAdd(&st->rotations, my_rot);
      if (MODE != 1) atomic_max_nonneg(&st->maxoff, my_max);
    }
  }
  mark();
  if (stamp) my_dbg[63] = nstamp;
}

// ---- the O(d^3) half: tiles on the fp64 matrix pipe -------------------------------------------------------------
// blocks [0, nA): symmetric tiles (K < L) of A;  blocks [nA, nA + nV): row tiles of up to two row-major matrices
// (V' for the two-sided form; W and Q for the one-sided form), 64 rows (the pair's two 32-row groups) x 64 columns.
struct BjRows {
  double* M[2];
  int64_t ld[2];
  int chunks[2];          // ceil(cols / 64)
  int groups[2];          // ceil(chunks / rg): workgroups per pair
  int64_t cols[2];
  int rg;                 // 64-column chunks per row-tile workgroup: 1 while the grid is small (a tile's latency is the
                          // launch's duration), 4 once the chip is full several times over (throughput: prefetch pays)
};
constexpr int BJ_AG = 4;  // tiles (K, L .. L + 3) per A-tile workgroup when the grid is large (rows.rg > 1)
inline int bj_row_group(int np, int chunks_total) {
  static const int64_t min_tiles = [] { const char* e = getenv("CCZ_BJ_GROUP_MIN_TILES"); return e ? atoll(e) : 2048LL; }();
  return int64_t(np) * chunks_total >= min_tiles ? 4 : 1;
}

// part: 0 = both kinds of tile in one launch, 1 = the A tiles only, 2 = the row tiles only.  fused: A is three rotating
// planes (read state g, write state g + 1) and Rt_all two (g & 1), g = st->g0 + round.
__global__ __launch_bounds__(256, 2) void k_bj_apply(double* __restrict__ A, int64_t lda, BjRows rows, const double* __restrict__ Rt_all,
                                                     int nb, int round, int nA, const BjStatus* __restrict__ st, int fused,
                                                     const int* __restrict__ ident) {
  extern __shared__ __attribute__((aligned(16))) char bj_smem[];
  double* X = reinterpret_cast<double*>(bj_smem);                   // 64 x 80
  double* Rs = X + BJP * AP_SB;                                     // 64 x 66
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4, cw = wave * 16;
  const int m1 = nb - 1;
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
  double* A_out = A;
  if (fused) {
    const long long g = st->g0 + round;
    const int64_t plane = lda * lda;
    A_out = A + ((g + 1) % 3) * plane;
    A += (g % 3) * plane;
    Rt_all += (g & 1) * int64_t(nb >> 1) * (BJP * BJP);
  }

  if (int(blockIdx.x) < nA) {
    // one tile per workgroup while the grid is small; a run of up to four tiles (K, L .. L + 3) once the chip is full
    // several times over: R_K stays in registers, the next tile's X and R_L travel global -> registers under the products
    const int np_ = nb >> 1;
    int K, L0, L1;
    if (rows.rg == 1) {
      int ti, tj;
      tri_decode(blockIdx.x, ti, tj);
      K = tj; L0 = ti + 1; L1 = L0 + 1;                             // K < L
    } else {
      int b = blockIdx.x;
      K = 0;
      for (;;) { const int ng = (np_ - 1 - K + BJ_AG - 1) / BJ_AG; if (b < ng) break; b -= ng; ++K; }
      L0 = K + 1 + b * BJ_AG; L1 = min(np_, L0 + BJ_AG);
    }
    int Ls[BJ_AG], nl = 0;
    for (int L = L0; L < L1; ++L)
      if (!(ident && ident[K] && ident[L])) Ls[nl++] = L;           // both pairs at rest: the tile does not change
    if (nl == 0) return;
    int bk[2];
    pair_of(round, K, m1, bk[0], bk[1]);
    v2f64 preX[8], preR[8], rk[8];
    auto fetchA = [&](int L) {
      int bl[2];
      pair_of(round, L, m1, bl[0], bl[1]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int si = s4 >> 1, sj = s4 & 1, x = bk[si], y = bl[sj];
        const double* base = x < y ? A + int64_t(x) * BJB * lda + y * BJB : A + int64_t(y) * BJB * lda + x * BJB;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int piece = tid + 256 * it, row = piece >> 4, c2 = (piece & 15) * 2;
          preX[2 * s4 + it] = *reinterpret_cast<const v2f64*>(base + int64_t(row) * lda + c2);
        }
      }
      const double* rl = Rt_all + int64_t(L) * (BJP * BJP);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
        preR[it] = *reinterpret_cast<const v2f64*>(rl + row * BJP + c2);
      }
    };
    // X (64 x 64, stride 66) <- the four sub-blocks, transposing the ones whose canonical copy is the mirrored one; Rs <- R_L
    auto stashA = [&](int L) {
      int bl[2];
      pair_of(round, L, m1, bl[0], bl[1]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int si = s4 >> 1, sj = s4 & 1;
        const bool direct = bk[si] < bl[sj];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int piece = tid + 256 * it, row = piece >> 4, c2 = (piece & 15) * 2;
          const v2f64 v = preX[2 * s4 + it];
          if (direct) {
            *reinterpret_cast<v2f64*>(X + (BJB * si + row) * AP_SX + BJB * sj + c2) = v;
          } else {
            X[(BJB * si + c2) * AP_SX + BJB * sj + row] = v.x;
            X[(BJB * si + c2 + 1) * AP_SX + BJB * sj + row] = v.y;
          }
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
        *reinterpret_cast<v2f64*>(Rs + row * AP_SX + c2) = preR[it];
      }
    };
    fetchA(Ls[0]);
    {
      const double* rkp = Rt_all + int64_t(K) * (BJP * BJP);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
        rk[it] = *reinterpret_cast<const v2f64*>(rkp + row * BJP + c2);
      }
    }
    stashA(Ls[0]);
    __syncthreads();
    for (int i = 0; i < nl; ++i) {
      const int L = Ls[i];
      if (i + 1 < nl) fetchA(Ls[i + 1]);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[tt] = v4f64{0.0, 0.0, 0.0, 0.0};
      // Y[:, cw .. cw+15] = X R_L :  A fragment X[m][k], B fragment R_L[k][n] = Rt_L[n][k]
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const double b = Rs[(cw + l15) * AP_SX + 4 * ks + l4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const double a = X[(16 * mt + l15) * AP_SX + 4 * ks + l4];
          acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[mt], 0, 0, 0);
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
        *reinterpret_cast<v2f64*>(Rs + row * AP_SX + c2) = rk[it];
      }
      __syncthreads();
      // Z[:, cw ..] = R_K' Y :  A fragment Rt_K[m][k]; B fragment of k-step ks = Y rows 4 ks .. 4 ks + 3 = accumulator
      // register (ks & 3) of row tile (ks >> 2)  (C layout of v_mfma_f64_16x16x4: row = (lane >> 4) + 4 reg)
      v4f64 z[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) z[tt] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const double b = acc[ks >> 2][ks & 3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const double a = Rs[(16 * mt + l15) * AP_SX + 4 * ks + l4];
          z[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z[mt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(16 * mt + l4 + 4 * r) * AP_SX + cw + l15] = z[mt][r];
      __syncthreads();
      int bl[2];
      pair_of(round, L, m1, bl[0], bl[1]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int si = s4 >> 1, sj = s4 & 1, x = bk[si], y = bl[sj];
        const bool direct = x < y;
        double* base = direct ? A_out + int64_t(x) * BJB * lda + y * BJB : A_out + int64_t(y) * BJB * lda + x * BJB;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int piece = tid + 256 * it, row = piece >> 4, c2 = (piece & 15) * 2;
          v2f64 v;
          if (direct) {
            v = *reinterpret_cast<const v2f64*>(X + (BJB * si + row) * AP_SX + BJB * sj + c2);
          } else {
            v.x = X[(BJB * si + c2) * AP_SX + BJB * sj + row];
            v.y = X[(BJB * si + c2 + 1) * AP_SX + BJB * sj + row];
          }
          *reinterpret_cast<v2f64*>(base + int64_t(row) * lda + c2) = v;
        }
      }
      if (i + 1 < nl) {
        __syncthreads();                                            // X has been written out, Rs (R_K) has been read
        stashA(Ls[i + 1]);
        __syncthreads();
      }
    }
    return;
  }

  // ---- row tiles: M[rows of pair K, a group of up to rows.rg 64-column chunks] <- R_K' M[...] ----
  // One workgroup walks its group with R_K resident in LDS; the next chunk's 32 KB travel global -> registers while the
  // current chunk is on the matrix pipe (with one chunk per workgroup and two workgroups per CU the load / product / store
  // phases of a tile did not hide each other: 39 % MFMA busy at d = 4096, profiles/r04_syev_pmc.md).
  int t = int(blockIdx.x) - nA;
  const int np = nb >> 1;
  int which = 0;
  if (t >= np * rows.groups[0]) { t -= np * rows.groups[0]; which = 1; }
  const int ngr = rows.groups[which], nch = rows.chunks[which];
  const int K = t / ngr, ch0 = (t - K * ngr) * rows.rg, ch1 = min(nch, ch0 + rows.rg);
  if (ident && ident[K]) return;
  double* M = rows.M[which];
  const int64_t ld = rows.ld[which], ncol = rows.cols[which];
  int bk[2];
  pair_of(round, K, m1, bk[0], bk[1]);
  const double* rkp = Rt_all + int64_t(K) * (BJP * BJP);
  v2f64 pre[8];
  auto fetch = [&](int ch) {
    const int64_t c0 = int64_t(ch) * 64;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
      const int64_t grow = int64_t(bk[row >> 5]) * BJB + (row & (BJB - 1));
      v2f64 v = {0.0, 0.0};
      if (c0 + c2 + 1 < ncol) v = *reinterpret_cast<const v2f64*>(M + grow * ld + c0 + c2);
      else if (c0 + c2 < ncol) v.x = M[grow * ld + c0 + c2];
      pre[it] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
      *reinterpret_cast<v2f64*>(X + row * AP_SB + c2) = pre[it];
    }
  };
  fetch(ch0);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
    *reinterpret_cast<v2f64*>(Rs + row * AP_SX + c2) = *reinterpret_cast<const v2f64*>(rkp + row * BJP + c2);
  }
  stash();
  __syncthreads();
  for (int ch = ch0; ch < ch1; ++ch) {
    if (ch + 1 < ch1) fetch(ch + 1);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const double b = X[(4 * ks + l4) * AP_SB + cw + l15];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const double a = Rs[(16 * mt + l15) * AP_SX + 4 * ks + l4];
        acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[mt], 0, 0, 0);
      }
    }
    const int64_t gc = int64_t(ch) * 64 + cw + l15;
    if (gc < ncol) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + l4 + 4 * r;
          const int64_t grow = int64_t(bk[row >> 5]) * BJB + (row & (BJB - 1));
          M[grow * ld + gc] = acc[mt][r];
        }
    }
    if (ch + 1 < ch1) {
      __syncthreads();                                              // every wave is done with this chunk's X
      stash();
      __syncthreads();
    }
  }
}

// ---- one-sided form: Gram blocks of the pairs' rows, split over the row length --------------------------------------
// grid (pairs, nsplit): G[pair][split] (64 x 64, full) = W_K[:, slice] W_K[:, slice]'
__global__ __launch_bounds__(256, 2) void k_bj_gram(const double* __restrict__ W, int64_t ldw, int64_t q, int nb, int round, int nsplit,
                                                    double* __restrict__ G) {
  __shared__ __attribute__((aligned(16))) double Xs[BJP * AP_SX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4, cw = wave * 16;
  int bk[2];
  pair_of(round, blockIdx.x, nb - 1, bk[0], bk[1]);
  const int64_t nchunks = (q + 63) / 64;
  const int64_t per = (nchunks + nsplit - 1) / nsplit;
  const int64_t ch0 = int64_t(blockIdx.y) * per, ch1 = min(nchunks, ch0 + per);
  v4f64 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
  for (int64_t ch = ch0; ch < ch1; ++ch) {
    const int64_t c0 = ch * 64;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = tid + 256 * it, row = piece >> 5, c2 = (piece & 31) * 2;
      const int64_t grow = int64_t(bk[row >> 5]) * BJB + (row & (BJB - 1));
      v2f64 v = {0.0, 0.0};
      if (c0 + c2 + 1 < q) v = *reinterpret_cast<const v2f64*>(W + grow * ldw + c0 + c2);
      else if (c0 + c2 < q) v.x = W[grow * ldw + c0 + c2];
      *reinterpret_cast<v2f64*>(Xs + row * AP_SX + c2) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const double b = Xs[(cw + l15) * AP_SX + 4 * ks + l4];        // B[k][n] = W_K[n][k]
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const double a = Xs[(16 * mt + l15) * AP_SX + 4 * ks + l4];
        acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[mt], 0, 0, 0);
      }
    }
  }
  double* g = G + (int64_t(blockIdx.x) * nsplit + blockIdx.y) * (BJP * BJP);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) g[(16 * mt + l4 + 4 * r) * BJP + cw + l15] = acc[mt][r];
}

// ---- preparation / extraction -----------------------------------------------------------------------------------

// Aw (dp x dp, canonical + whole diagonal sub-blocks; everything is written) <- (A + A') / 2, zero padded; Vt <- I
__global__ __launch_bounds__(256) void k_bj_prep_sym(const double* __restrict__ A, int64_t lda, int64_t d, int64_t dp, double* __restrict__ Aw,
                                                     double* __restrict__ A0, double* __restrict__ Vt, BjStatus* __restrict__ st) {
  __shared__ double red[4];
  double mx = 0.0;
  const int64_t total = dp * dp;
  for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
    const int64_t i = e / dp, j = e - i * dp;
    double h = 0.0;
    if (i < d && j < d) {
      h = 0.5 * (A[i * lda + j] + A[j * lda + i]);
      const double a = fabs(h);
      mx = (a <= 1.79769313486231570e308) ? fmax(mx, a) : __builtin_inf();
    }
    Aw[e] = h;
    if (A0) A0[e] = h;
    if (Vt) Vt[e] = (i == j) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0 && st) atomic_max_nonneg(&st->hmax, fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
}

// one-sided: hmax <- largest squared row norm of W (p x q); one wave per row
__global__ __launch_bounds__(64) void k_bj_prep_rows(const double* __restrict__ W, int64_t ldw, int64_t q, BjStatus* __restrict__ st) {
  const double* w = W + int64_t(blockIdx.x) * ldw;
  double s = 0.0;
  for (int64_t t = threadIdx.x; t < q; t += 64) { const double x = w[t]; s += x * x; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) atomic_max_nonneg(&st->hmax, s);
}

__global__ void k_bj_diag(const double* __restrict__ Aw, int64_t dp, int64_t d, double* __restrict__ w) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < d) w[i] = Aw[i * dp + i];
}

bool bj_fused() {   // CCZ_BJ_FUSED=1: pair kernels and tile updates on two streams (A/B; measured: no gain, see syev_block)
  static const bool v = [] { const char* e = getenv("CCZ_BJ_FUSED"); return e && atoi(e) == 1; }();
  return v;
}

int bj_max_gram_split() {
  static const int v = [] { const char* e = getenv("CCZ_BJ_GRAM_SPLIT"); return e ? std::max(1, atoi(e)) : 0; }();
  return v;
}

struct StatusBuf {     // BjStatus followed by one "pair at rest" flag per pair
  ccz_ctx* c;
  BjStatus* dev;
  int* ident;
  StatusBuf(ccz_ctx* c_, int np) : c(c_), dev(static_cast<BjStatus*>(dev_alloc(c_, sizeof(BjStatus) + size_t(np) * sizeof(int)))) {
    ident = reinterpret_cast<int*>(dev + 1);
  }
  ~StatusBuf() { dev_free(c, dev); }
};

constexpr size_t kApplyLds = size_t(BJP) * (AP_SB + AP_SX) * 8;     // 74752 B: two workgroups per CU

void apply_attr_once() {
  static const bool done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_apply), hipFuncAttributeMaxDynamicSharedMemorySize, int(kApplyLds));
    return true;
  }();
  (void)done;
}

}  // namespace

int syev_block_min(ccz_ctx*) { return 2; }

// Two-sided block Jacobi: A (d x d, only read, symmetrised on load) -> w_dev (d, unsorted), rows of Vrows = eigenvectors
//
// Refresh.  Every round multiplies all of A and V' by 64 x 64 orthogonal blocks, and the rounding of those ~ sweeps * d / 32
// products accumulates: measured ||A V - V L|| / ||A|| = 0.6e-15 * d at convergence (2.5e-12 at d = 4096, the same in a
// NumPy float64 emulation of the algorithm).  So once the off-diagonal mass is small (largest rotated element below
// 1e-6 max|A|: two sweeps from the end) the iteration is RESTARTED from clean data: V' is re-orthogonalised by one
// Newton-Schulz step, B = V' A V is formed from the ORIGINAL matrix (four d^3 GEMMs at the fp64 matrix rate, ~4 % of a
// solve) and the remaining sweeps run on B -- their rounding is all that is left in the result.
int syev_block(ccz_ctx* c, const double* A, int64_t d, int64_t lda, double* w_dev, double* Vrows, int64_t ldv, int max_sweeps) {
  if (d < 1) fail(CCZ_EINVAL, "syev_block: d >= 1 required");
  static const int refresh_min = [] { const char* e = getenv("CCZ_EVD_REFRESH_MIN"); return e ? atoi(e) : 1536; }();
  const bool want_refresh = d >= refresh_min;
  const int64_t dp = (d + BJP - 1) / BJP * BJP, plane = dp * dp;
  const int nb = int(dp / BJB), np = nb / 2;
  // Fused schedule (default from 4 blocks on): the pair kernel of round g + 1 re-derives its cross block from the OLD tile
  // and round g's rotations, so it depends on the pair kernel of round g only; round g's tile update (k_bj_apply) runs on a
  // second stream underneath it.  A then lives in three rotating planes (state g in plane g % 3: read by the tile update of
  // round g and by the pair kernels of round g + 1, written by those of round g - 1), the rotations in two.
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  // MEASURED (round 4): no gain.  The pair kernel's workgroups (1024 threads, 134 KB of LDS: a whole CU each) do not get
  // CUs while the tile update floods the chip with thousands of small workgroups -- 190 us per round at d = 4096 against
  // 174 serial, 19.8 against 19.3 ms at d = 1024, with or without hipGraphs; a CU-masked stream for the tile updates
  // (hipExtStreamCreateWithCUMask, every 8th / 4th CU left free) made d = 1024 slower still (29 ms).  The schedule is
  // kept behind CCZ_BJ_FUSED=1 for A/B runs; the default is the serial alternation on one stream.
  hipStream_t side = nullptr;
  const bool fused = bj_fused() && nb >= 4 && st != nullptr;
  if (fused) {
    if (!im->aux_stream) CCZ_HIP(hipStreamCreateWithFlags(&im->aux_stream, hipStreamNonBlocking));
    side = im->aux_stream;
  }
  const int nplanes = fused ? 3 : 1;
  DBuf Aw(c, nplanes * plane), Vt(c, plane), Rt(c, int64_t(fused ? 2 : 1) * np * BJP * BJP), A0(c, want_refresh ? plane : 0);
  StatusBuf sb(c, np);
  apply_attr_once();
  if (fused)
    for (auto& e : im->bj_ev) if (!e) CCZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CCZ_HIP(hipMemsetAsync(sb.dev, 0, sizeof(BjStatus) + size_t(np) * sizeof(int), st));
  const dim3 pgrid((unsigned)std::min<int64_t>((plane + 255) / 256, 4096));
  hipLaunchKernelGGL(k_bj_prep_sym, pgrid, dim3(256), 0, st, A, lda, d, dp, Aw.get(), want_refresh ? A0.get() : (double*)nullptr, Vt.get(),
                     sb.dev);
  CCZ_LAUNCH_CHECK();
  const int vch = int(dp / 64);
  int nA = np * (np - 1) / 2;
  BjRows rows{};
  rows.rg = bj_row_group(np, vch);
  if (rows.rg > 1) { nA = 0; for (int K = 0; K < np; ++K) nA += (np - 1 - K + BJ_AG - 1) / BJ_AG; }
  rows.M[0] = Vt.get(); rows.ld[0] = dp; rows.chunks[0] = vch; rows.cols[0] = dp; rows.groups[0] = (vch + rows.rg - 1) / rows.rg;
  rows.M[1] = nullptr; rows.ld[1] = 0; rows.chunks[1] = 0; rows.cols[1] = 0; rows.groups[1] = 0;
  const int nV = np * rows.groups[0];
  uint64_t key = graph_key_mix(graph_key_mix(0x424a5359ull + (fused ? 1 : 0), uint64_t(dp)), reinterpret_cast<uint64_t>(Aw.get()));
  key = graph_key_mix(graph_key_mix(key, reinterpret_cast<uint64_t>(Vt.get())), reinterpret_cast<uint64_t>(Rt.get()));
  key = graph_key_mix(graph_key_mix(key, reinterpret_cast<uint64_t>(sb.dev)), reinterpret_cast<uint64_t>(side));
  int sweeps = -1;
  bool refreshed = false;
  long long g_total = 0, direct_g = 0;                    // (fused) global round counter; the round that reads its cross blocks directly
  static const bool dbg_on = getenv("CCZ_BJ_DEBUG") != nullptr;
  DBuf dbgb(c, dbg_on ? 192 : 0);
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    CCZ_HIP(hipMemsetAsync(&sb.dev->rotations, 0, sizeof(int), st));
    CCZ_HIP(hipMemsetAsync(&sb.dev->maxoff, 0, sizeof(double), st));
    if (fused) {
      const long long ctl[2] = {g_total, direct_g};
      h2d_small(c, &sb.dev->g0, ctl, sizeof ctl);
    }
    graph_run_fn(c, key, [&] {
      if (!fused) {
        for (int round = 0; round < nb - 1; ++round) {
          hipLaunchKernelGGL((round == 0 ? k_bj_inner2<0, 1> : k_bj_inner2<0, 0>), dim3(np), dim3(BJ2_THREADS), 0, st, Aw.get(), dp,
                             (const double*)nullptr, 0, Rt.get(), sb.dev, nb, round, 0.0,
                             dbg_on && round == 1 ? (long long*)dbgb.get() : (long long*)nullptr, sb.ident);
          hipLaunchKernelGGL(k_bj_apply, dim3(nA + nV), dim3(256), kApplyLds, st, Aw.get(), dp, rows, (const double*)Rt.get(), nb, round, nA,
                             (const BjStatus*)sb.dev, 0, (const int*)sb.ident);
        }
      } else {
        // main stream: the chain of pair kernels; side stream: the tile updates.  pair(r) waits for update(r - 2) (it
        // overwrites the plane and the rotation buffer that update read); update(r) waits for pair(r).
        hipEvent_t* evI = im->bj_ev;        // [4]: pair kernel r done
        hipEvent_t* evU = im->bj_ev + 4;    // [4]: tile update r done
        for (int round = 0; round < nb - 1; ++round) {
          if (round >= 2) CCZ_HIP(hipStreamWaitEvent(st, evU[(round - 2) & 3], 0));
          hipLaunchKernelGGL((round == 0 ? k_bj_inner2<2, 1> : k_bj_inner2<2, 0>), dim3(np), dim3(BJ2_THREADS), 0, st, Aw.get(), dp,
                             (const double*)nullptr, 0, Rt.get(), sb.dev, nb, round, 0.0,
                             dbg_on && round == 1 ? (long long*)dbgb.get() : (long long*)nullptr, sb.ident);
          CCZ_HIP(hipEventRecord(evI[round & 3], st));
          CCZ_HIP(hipStreamWaitEvent(side, evI[round & 3], 0));
          hipLaunchKernelGGL(k_bj_apply, dim3(nA + nV), dim3(256), kApplyLds, side, Aw.get(), dp, rows, (const double*)Rt.get(), nb, round, nA,
                             (const BjStatus*)sb.dev, 1, (const int*)nullptr);
          CCZ_HIP(hipEventRecord(evU[round & 3], side));
        }
        for (int round = std::max(0, nb - 3); round < nb - 1; ++round) CCZ_HIP(hipStreamWaitEvent(st, evU[round & 3], 0));   // join
      }
      CCZ_LAUNCH_CHECK();
    });
    g_total += nb - 1;
    BjStatus h{};
    d2h(c, &h, sb.dev, sizeof(BjStatus));
    if (dbg_on && sweep == 2) {
      long long t[192];
      d2h(c, t, dbgb.get(), sizeof t);
      for (int w = 0; w < 3; ++w) {
        fprintf(stderr, "[bj dbg] %s:", w == 0 ? "param wave" : (w == 1 ? "S wave" : "R wave"));
        for (int i = 1; i < int(t[w * 64 + 63]) && i < 40; ++i) fprintf(stderr, " %lld", t[w * 64 + i] - t[w * 64 + i - 1]);
        fprintf(stderr, "\n");
      }
    }
    if (h.bad || !(h.hmax < INFINITY)) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
    if (h.rotations == 0) { sweeps = sweep; break; }
    if (want_refresh && !refreshed && h.maxoff <= 1e-6 * h.hmax) {
      refreshed = true;
      double* cur = Aw.get() + (fused ? (g_total % 3) * plane : 0);
      DBuf G(c, plane), V2(c, plane);
      gemm(c, false, true, dp, dp, dp, 1.0, Vt, dp, Vt, dp, 0.0, G, dp);             // G = V' V
      gemm(c, false, false, dp, dp, dp, -0.5, G, dp, Vt, dp, 0.0, V2, dp);           // V2 = (1.5 I - 0.5 G) V'
      axpby2d(c, dp, dp, 1.0, V2, dp, 1.5, Vt, dp);
      gemm(c, false, false, dp, dp, dp, 1.0, V2, dp, A0, dp, 0.0, G, dp);            // G = V2 A0
      gemm(c, false, true, dp, dp, dp, 1.0, G, dp, V2, dp, 0.0, cur, dp);            // B = V2 A0 V2'
      d2d(c, Vt, V2, size_t(plane) * 8);
      direct_g = g_total;                                                            // no previous round to re-apply
    }
  }
  if (sweeps < 0) fail(CCZ_ENOCONV, "block Jacobi did not converge in %d sweeps (d=%lld)", max_sweeps, (long long)d);
  const double* fin = Aw.get() + (fused ? (g_total % 3) * plane : 0);
  hipLaunchKernelGGL(k_bj_diag, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, fin, dp, d, w_dev);
  CCZ_LAUNCH_CHECK();
  if (Vrows) copy2d(c, d, d, Vt.get(), dp, Vrows, ldv);
  return sweeps;
}

// One-sided block Jacobi on the ROWS of W (p x q, p a multiple of 64 -- the caller pads with zero rows): the rows
// end up mutually orthogonal; the same rotations go to the rows of Q (p x qc) when Q != null.
int jacobi_rows_block(ccz_ctx* c, int64_t p, int64_t q, double* W, int64_t ldw, double* Q, int64_t qc, int64_t ldq, int max_sweeps) {
  if (p < BJP || p % BJP != 0) fail(CCZ_EINVAL, "jacobi_rows_block: p must be a positive multiple of 64");
  const int nb = int(p / BJB), np = nb / 2;
  int nsplit = bj_max_gram_split();
  if (nsplit <= 0) {                                   // fill the chip: pairs x splits ~ 512 workgroups, >= 4 chunks of 64 per split
    const int64_t nchunks = (q + 63) / 64;
    nsplit = int(std::max<int64_t>(1, std::min<int64_t>(nchunks / 4, (512 + np - 1) / np)));
  }
  DBuf Rt(c, int64_t(np) * BJP * BJP), G(c, int64_t(np) * nsplit * BJP * BJP);
  StatusBuf sb(c, np);
  hipStream_t st = stream(c);
  apply_attr_once();
  CCZ_HIP(hipMemsetAsync(sb.dev, 0, sizeof(BjStatus) + size_t(np) * sizeof(int), st));
  hipLaunchKernelGGL(k_bj_prep_rows, dim3((unsigned)p), dim3(64), 0, st, (const double*)W, ldw, q, sb.dev);
  CCZ_LAUNCH_CHECK();
  BjRows rows{};
  rows.M[0] = W; rows.ld[0] = ldw; rows.chunks[0] = int((q + 63) / 64); rows.cols[0] = q;
  rows.M[1] = Q; rows.ld[1] = ldq; rows.chunks[1] = Q ? int((qc + 63) / 64) : 0; rows.cols[1] = Q ? qc : 0;
  rows.rg = bj_row_group(np, rows.chunks[0] + rows.chunks[1]);
  rows.groups[0] = (rows.chunks[0] + rows.rg - 1) / rows.rg; rows.groups[1] = (rows.chunks[1] + rows.rg - 1) / rows.rg;
  const int nV = np * (rows.groups[0] + rows.groups[1]);
  const double tol = BJ_EPS * std::sqrt(double(q)) * 4.0;
  uint64_t key = graph_key_mix(graph_key_mix(0x424a4f53ull, uint64_t(p)), uint64_t(q));
  key = graph_key_mix(graph_key_mix(key, reinterpret_cast<uint64_t>(W)), reinterpret_cast<uint64_t>(Q));
  key = graph_key_mix(graph_key_mix(key, uint64_t(ldw)), uint64_t(ldq));
  key = graph_key_mix(graph_key_mix(key, uint64_t(qc)), reinterpret_cast<uint64_t>(Rt.get()));
  key = graph_key_mix(graph_key_mix(key, reinterpret_cast<uint64_t>(G.get())), reinterpret_cast<uint64_t>(sb.dev));
  key = graph_key_mix(key, uint64_t(nsplit));
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    CCZ_HIP(hipMemsetAsync(&sb.dev->rotations, 0, sizeof(int), st));
    graph_run_fn(c, key, [&] {
      for (int round = 0; round < nb - 1; ++round) {
        hipLaunchKernelGGL(k_bj_gram, dim3(np, nsplit), dim3(256), 0, st, (const double*)W, ldw, q, nb, round, nsplit, G.get());
        hipLaunchKernelGGL((round == 0 ? k_bj_inner2<1, 1> : k_bj_inner2<1, 0>), dim3(np), dim3(BJ2_THREADS), 0, st, (double*)nullptr,
                           int64_t(0), (const double*)G.get(), nsplit, Rt.get(), sb.dev, nb, round, tol, (long long*)nullptr, sb.ident);
        hipLaunchKernelGGL(k_bj_apply, dim3(nV), dim3(256), kApplyLds, st, (double*)nullptr, int64_t(0), rows, (const double*)Rt.get(), nb,
                           round, 0, (const BjStatus*)sb.dev, 0, (const int*)sb.ident);
      }
      CCZ_LAUNCH_CHECK();
    });
    BjStatus h{};
    d2h(c, &h, sb.dev, sizeof(BjStatus));
    if (h.bad || !(h.hmax < INFINITY)) fail(CCZ_EINVAL, "gesvj: matrix has non-finite entries");
    if (h.rotations == 0) return sweep;
  }
  fail(CCZ_ENOCONV, "block Jacobi did not converge in %d sweeps (p=%lld, q=%lld)", max_sweeps, (long long)p, (long long)q);
}

}  // namespace ccz
