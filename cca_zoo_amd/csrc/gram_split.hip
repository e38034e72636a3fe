// K1 on the bf16 matrix pipe: fp32 second moments from TWO bf16 planes ("bf16 split-accumulate ... only if
// it holds the 1e-3 bar").
//
//   d = fl32(x - p)            p: the launch's pilot (fp32 column mean), subtracted once, undone in fp64 (k_pilot_fixup)
//   d = hi + mid + lo          hi = bf16(d), mid = bf16(d - hi)  (d - hi is exact in fp32), |mid| <= 2^-8 |d|, |lo| <= 2^-16 |d|
//   d_i d_j = hi_i hi_j + hi_i mid_j + mid_i hi_j  [+ mid_i mid_j + hi_i lo_j + lo_i hi_j + ...]
//
// The three kept products run as THREE MFMAs into ONE fp32 accumulator per k-step (operand sequences (H, H, M) x (H, M, H)),
// so an output tile needs no symmetrisation pass and costs 3 bf16 MFMAs per fp32 one it replaces -- at 16x the fp32 MFMA
// rate.  Of the dropped terms only mid_i mid_i on the DIAGONAL is systematic (a sum of squares); it is added back exactly
// from sum_k mid_ik^2, which the split pass accumulates on the side.  The rest are zero-mean at 2^-16 per product and
// average out over the rows like the fp32 kernel's own accumulation error (measured in bench.py: k1_rel_err of both routes).
// Accumulation is unchanged from the fp32 kernel: fp32 inside a row chunk (<= 16384 rows), fp64 across chunks.
//
// Data layout.  The MFMA operand of v_mfma_f32_32x32x16_bf16 is 8 consecutive k (= ROWS of a view) of one column per lane,
// which row-major views cannot deliver coalesced -- so the split pass also transposes, into the exact LDS image the Gram
// kernel wants:  planes[panel p][k-step s][plane H|M][column tile t: 8][k half h: 2][column c: 32][k: 8]  (bf16), i.e. one
// panel (256 columns) x one k-step (16 rows) = 16 KiB, and every 1 KiB run of it is one MFMA operand of one wave
// (lane l = (h, c) holds its 16 bytes).  The Gram kernel streams those runs with buffer_load ... lds (LDS-DMA: lane-linear
// image, no VGPRs, no ds_write) and reads them back with conflict-free ds_read_b128 -- no address arithmetic anywhere.
// Views are padded per view to whole panels and the rows to whole k-steps with zeros (d = 0, not x - p), so every tile is
// a full tile: ragged widths, unaligned rows and odd leading dimensions all take this one path.
//
// Gram kernel: 256 x 256 output tile per workgroup, 4 waves (2 x 2) of 128 x 128 = 4 x 4 MFMA tiles, one wave per SIMD
// (256 accumulator registers).  LDS: a 4-slot ring of k-steps (32 KiB each: A panel H|M, B panel H|M); each wave DMAs a
// quarter of every slot.  One barrier per k-step: wave-side `s_waitcnt vmcnt(16)` (its own DMAs of step s + 1 have landed)
// -> s_barrier (everybody's have; everybody has finished reading slot s) -> DMA of step s + 4 into slot s -> fragments of
// step s + 1 from LDS while the 48 MFMAs of step s run.  Three k-steps of loads are always in flight.
//
// Output: per (row chunk, tile) fp32 partial tiles with plain 16-byte stores (no atomics); k_split_reduce adds the chunks
// up in fp64 into G and applies the diagonal correction.  The loss fast path (gram_partials_split_f32) hands the partial
// tiles to loss.hip's preparation kernel instead.  The split pass also gathers the exact fp64 column sums of x (the means; s of
// the pilot fix-up), so the route reads the rows once.  Panel / tile tables hold no pointers and are cached with the handle.
//
// Roofline: executed flops = 3 x n Dp (Dp + 256) on the bf16 pipe (2.5 PF dense peak); algorithmic F = n D (D + 1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip_common.h"
#include "gram_map.h"
#include "split_mma.h"

namespace ccz {

constexpr int SP_MAXV = 16;          // views per launch (their base pointers travel as a kernel argument, the tables hold no pointers)

struct SplitPanel {
  int32_t view;          // which view
  int32_t col0;          // first column of the panel in its view
  int32_t width;         // valid columns (<= 256); the rest of the panel is zero
  int32_t gcol0;         // column of G (and of pilot / msq) of the panel's first column
};

struct SplitViews {
  const float* data[SP_MAXV];
  int64_t ld[SP_MAXV];
};

struct SplitTile {
  int32_t pa, pb;        // panels (pa <= pb)
  int32_t out_row, out_col;
  int32_t wa, wb;
  int32_t diag;
  int32_t pad_;
};

// ---------------------------------------------------------------------------
// split pass: fp32 rows [r0, r0 + nrows) of every view -> the two bf16 planes of this launch (k-steps [0, ksteps)),
// and msq[j] += sum_k mid_kj^2.  grid = (row blocks of rb <= SP_RB rows, panels).  A thread owns 4 consecutive columns and, per
// pass, the 8 rows of one k half: its 8 float4 loads are one 1 KiB-per-wave row segment each, its stores 64 contiguous
// bytes per plane.  HBM-bound: 4 bytes in, 4 bytes out per element.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sp_wave_lds_sync() {    // lanes of ONE wave exchange data through LDS (see cholinv.hip::wave_lds_sync)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCH (default; CCZ_SPLIT_XCH=0 restores the direct stores): the 64 bytes a thread produces per plane are FOUR 16-byte pieces of a
// 512-byte run, so a wave-level store of one piece per lane is 64 separate 16-byte writes at a stride of 64 bytes -- the PMC pass
// (profiles/r06_split_pass_pmc_raw.md) counts 6.05e8 L2 requests per 262144 rows, 5.4e8 of them these partial writes, i.e. 80
// requests per clock on 128 L2 channels.  With XCH the eight lanes of a column tile swap pieces through a wave-private 4 KB of LDS
// (no workgroup barrier) so that store e of lane j is piece 8 e + j: every wave-level store is eight whole 128-byte lines.
template <bool ALIGNED, bool XCH>
__global__ __launch_bounds__(256) void k_split_bf16x2(const SplitPanel* __restrict__ panels, SplitViews vws, int64_t r0, int64_t nrows, int64_t ksteps,
                                                      const float* __restrict__ pilot, char* __restrict__ planes,
                                                      double* __restrict__ msq, double* __restrict__ csum, int rb, int panel_fast) {
  __shared__ float red[4][256];
  __shared__ double redc[4][256];
  __shared__ sp_v4u32 xch[XCH ? 4 : 1][XCH ? 256 : 1];      // per wave: 8 column tiles x 32 pieces of 16 bytes (one plane, one k half)
  // panel_fast: consecutive workgroups take the panels of ONE row block (together they read whole rows) instead of the row blocks
  // of one panel (a 1 KiB stripe of every 16 KiB row: the same few HBM channels for everybody) -- CCZ_SPLIT_ORDER
  const unsigned pidx = panel_fast ? blockIdx.x : blockIdx.y, ridx = panel_fast ? blockIdx.y : blockIdx.x;
  const SplitPanel pn = panels[pidx];
  const int tid = threadIdx.x, cg = tid & 63, rg = tid >> 6;
  const int64_t rows_pad = ksteps * SP_K;
  const int64_t rb0 = int64_t(ridx) * rb;                   // rb rows per workgroup (a multiple of 32, <= SP_RB)
  const int c0 = 4 * cg;
  sp_v4f32 p = {0.f, 0.f, 0.f, 0.f};
  bool cok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cok[e] = c0 + e < pn.width;
    if (pilot && cok[e]) p[e] = pilot[pn.gcol0 + c0 + e];
  }
  const float* __restrict__ X = vws.data[pn.view] + pn.col0 + c0;
  const int64_t pld = vws.ld[pn.view];
  char* out = planes + int64_t(pidx) * ksteps * SP_PSTEP + (cg >> 3) * 1024 + (cg & 7) * 64;
  sp_v4f32 q = {0.f, 0.f, 0.f, 0.f};
  double cs[4] = {0.0, 0.0, 0.0, 0.0};               // exact column sums of x (not of x - p): the means, and the pilot fix-up's s
  for (int pass = 0; pass < rb / 32; ++pass) {
    const int64_t row = rb0 + pass * 32 + rg * 8;          // first of this thread's 8 rows (within the launch)
    if (row >= rows_pad) break;
    sp_v4f32 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool rok = row + k < nrows;
      const float* src = X + (r0 + (rok ? row + k : 0)) * pld;
      if (ALIGNED) {
        v[k] = p;                                          // (whole 4-column groups are inside or outside the panel's width)
        if (cok[0]) v[k] = __builtin_nontemporal_load(reinterpret_cast<const sp_v4f32*>(src));   // read once: keep L2 / MALL for the planes
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = cok[e] ? src[e] : 0.f;
      }
      if (!rok) v[k] = p;                                  // d = 0 exactly for the rows that pad the last k-step
      else if (csum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[e] += cok[e] ? double(v[k][e]) : 0.0;
      }
    }
    sp_v4u32 hw[4], mw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d[8], m[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = cok[e] ? v[k][e] - p[e] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const unsigned hb = sp_pack2(d[k], d[k + 1]);
        hw[e][k >> 1] = hb;
        m[k] = d[k] - __builtin_bit_cast(float, hb << 16);            // exact: hi shares d's leading bits
        m[k + 1] = d[k + 1] - __builtin_bit_cast(float, hb & 0xffff0000u);
        const unsigned mb = sp_pack2(m[k], m[k + 1]);
        mw[e][k >> 1] = mb;
        const float m0 = __builtin_bit_cast(float, mb << 16), m1 = __builtin_bit_cast(float, mb & 0xffff0000u);
        q[e] += m0 * m0 + m1 * m1;
      }
    }
    char* dst = out + (row >> 4) * SP_PSTEP + ((row >> 3) & 1) * 512;
    if constexpr (XCH) {
      // piece (tile t, column c) sits at xw[32 t + (c ^ ((c >> 3) & 3) ^ ((t & 1) << 3))]: the 8 lanes of a ds_write_b128 group
      // (c = 4 j + e) then cover all eight 16-byte bank quads, and the 16 lanes of a ds_read_b128 group (c = 8 e + j over four
      // tiles) all sixteen of a 256-byte row (MI355X_MICROARCH.md, LDS service groups) -- plain [32 t + c] is 4-way / 2-way
      sp_v4u32* xw = xch[rg];
      char* dst2 = dst - (cg & 7) * 48;                                // tile base + (cg & 7) * 16
      const int t8 = ((cg >> 3) & 1) << 3, j = cg & 7;
      const int wr = (cg >> 3) * 32 + ((4 * j) ^ t8), wx = j >> 1;     // + (e ^ wx)
      const int rd = (cg >> 3) * 32;                                   // + ((8 e) ^ t8) + (j ^ e)
      sp_wave_lds_sync();                                              // the previous pass's reads are done
#pragma unroll
      for (int e = 0; e < 4; ++e) xw[wr + (e ^ wx)] = hw[e];
      sp_wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 4; ++e) hw[e] = xw[rd + ((8 * e) ^ t8) + (j ^ e)];
      sp_wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 4; ++e) xw[wr + (e ^ wx)] = mw[e];
      sp_wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 4; ++e) mw[e] = xw[rd + ((8 * e) ^ t8) + (j ^ e)];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<sp_v4u32*>(dst2 + e * 128) = hw[e];
        *reinterpret_cast<sp_v4u32*>(dst2 + SP_PLANE + e * 128) = mw[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<sp_v4u32*>(dst + e * 16) = hw[e];              // (plain stores: non-temporal ones made this pass 3.5x slower)
        *reinterpret_cast<sp_v4u32*>(dst + SP_PLANE + e * 16) = mw[e];
      }
    }
  }
  // msq: the four row groups of the workgroup -> one fp64 atomic per column
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rg][c0 + e] = q[e];
  if (csum) {
#pragma unroll
    for (int e = 0; e < 4; ++e) redc[rg][c0 + e] = cs[e];
  }
  __syncthreads();
  if (msq && tid < pn.width) {
    const double s = double(red[0][tid]) + double(red[1][tid]) + double(red[2][tid]) + double(red[3][tid]);
    unsafeAtomicAdd(msq + pn.gcol0 + tid, s);
  }
  if (csum && tid < pn.width) unsafeAtomicAdd(csum + pn.gcol0 + tid, (redc[0][tid] + redc[1][tid]) + (redc[2][tid] + redc[3][tid]));
}

// ---------------------------------------------------------------------------
// Gram kernel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_gram_bf16x2(const SplitTile* __restrict__ tiles, int ntiles, int per_xcd, int64_t ksplit,
                                                        const char* __restrict__ planes, int64_t ksteps, int64_t steps_per_wg,
                                                        float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WorkItem wi = locate_work(blockIdx.x, ntiles, per_xcd, ksplit);
  if (!wi.valid) return;
  const SplitTile t = tiles[wi.tile];
  const int64_t s0 = wi.chunk * steps_per_wg;
  const int64_t s1 = min(ksteps, s0 + steps_per_wg);
  if (s0 >= s1) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  sp_v16f32 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* srcA = planes + (int64_t(t.pa) * ksteps + s0) * SP_PSTEP;
  const char* srcB = planes + (int64_t(t.pb) * ksteps + s0) * SP_PSTEP;
  const int nsteps = int(s1 - s0);
  // a diagonal tile (A side == B side): the two symmetric quadrants compute their upper MFMA tiles, and the wave of the
  // redundant quadrant (1, 0) takes the lower half of quadrant (0, 1) -- 30 MFMAs per k-step for the slowest wave instead of 48.
  // What is not computed lies below the diagonal of G and is never read (k_split_reduce skips it).
  int qr = wr, qc = wc, ti0 = 0, ti1 = 4;
  bool sym = false;
  if (t.diag && wr == wc) {
    sym = true;
    split_mma_core<1>(acc, smem, srcA, srcB, nsteps, wave, lane, qr, qc);
  } else if (t.diag && wr == 0) {
    ti1 = 2;
    split_mma_core<2>(acc, smem, srcA, srcB, nsteps, wave, lane, qr, qc);
  } else if (t.diag) {
    qr = 0; qc = 1; ti0 = 2;
    split_mma_core<3>(acc, smem, srcA, srcB, nsteps, wave, lane, qr, qc);
  } else {
    split_mma_core<0>(acc, smem, srcA, srcB, nsteps, wave, lane, qr, qc);
  }

  // epilogue: the chunk's fp32 sums -> this (chunk, tile)'s slot, row-major 256 x 256
  float* pt = partial + (wi.chunk * int64_t(ntiles) + wi.tile) * int64_t(SP_T * SP_T);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ti < ti0 || ti >= ti1) continue;
    float* prow = pt + (qr * 128 + ti * 32 + (lane & 31)) * SP_T + qc * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      if (sym && tj < ti) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const sp_v4f32 v = {acc[ti][tj][4 * g], acc[ti][tj][4 * g + 1], acc[ti][tj][4 * g + 2], acc[ti][tj][4 * g + 3]};
        *reinterpret_cast<sp_v4f32*>(prow + tj * 32 + 8 * g) = v;
      }
    }
  }
}

// G (upper tiles) += sum over the row chunks of the fp32 partial tiles, in fp64; diagonal: += sum_k mid^2 (the one dropped
// product that does not average out).  grid = (64, tiles), a thread owns 4 consecutive elements of a tile row.
__global__ __launch_bounds__(256) void k_split_reduce(const float* __restrict__ partial, const SplitTile* __restrict__ tiles, int ntiles,
                                                      int64_t ksplit, double* __restrict__ G, int64_t ldg, const double* __restrict__ msq) {
  const int tile = blockIdx.y;
  const SplitTile t = tiles[tile];
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int i = e >> 8, j = e & 255;
  if (i >= t.wa || j >= t.wb) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const float* p = partial + int64_t(tile) * (SP_T * SP_T) + e;
  const int64_t stride = int64_t(ntiles) * (SP_T * SP_T);
  for (int64_t c = 0; c < ksplit; ++c) {
    const sp_v4f32 v = __builtin_nontemporal_load(reinterpret_cast<const sp_v4f32*>(p + c * stride));   // read once
    a0 += double(v[0]); a1 += double(v[1]); a2 += double(v[2]); a3 += double(v[3]);
  }
  const double a[4] = {a0, a1, a2, a3};
  double* g = G + (int64_t(t.out_row) + i) * ldg + t.out_col + j;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (j + q >= t.wb) break;
    if (t.diag && j + q < i) continue;          // below the diagonal of G: not authoritative anywhere (the pilot fix-up skips it too)
    double v = a[q];
    if (t.diag && i == j + q) v += msq[t.out_row + i];
    g[q] += v;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

// (i, j) panel pairs, i <= j, in launch order -- the order of gram.hip's tile table: whole 4 x 8 supertiles without a
// diagonal tile first (32 consecutive tiles share 12 panels through an XCD's L2), then the diagonal tiles, then the rest
std::vector<std::pair<int, int>> split_tile_order(int np) {
  std::vector<std::pair<int, int>> full, dg, rest;
  for (int I = 0; I < np; I += 4)
    for (int J = (I / 8) * 8; J < np; J += 8) {
      std::vector<std::pair<int, int>> blk;
      bool has_diag = false;
      for (int i = I; i < std::min(np, I + 4); ++i)
        for (int j = std::max(i, J); j < std::min(np, J + 8); ++j) { blk.emplace_back(i, j); has_diag = has_diag || i == j; }
      for (const auto& ij : blk) {
        if (blk.size() == 32 && !has_diag) full.push_back(ij);
        else (ij.first == ij.second ? dg : rest).push_back(ij);
      }
    }
  full.insert(full.end(), dg.begin(), dg.end());
  full.insert(full.end(), rest.begin(), rest.end());
  return full;
}

}  // namespace

size_t split_scratch_budget(ccz_ctx* c) {
  const char* e = getenv("CCZ_SPLIT_SCRATCH_GB");         // read per call: tests shrink it to force several row super-chunks
  const double env_gb = e ? atof(e) : 48.0;
  const double cap = 0.4 * double(impl(c)->props.totalGlobalMem);
  return size_t(std::max(8.0 * 1048576.0, std::min(env_gb * 1073741824.0, cap)));
}

// rows per workgroup of the split pass: SP_RB on large inputs (fewer atomics), down to 32 so that short inputs (a DCCA batch,
// the Gamma matrix of a backward) still spread over the chip
static int split_rows_per_block(int64_t rows_pad, int64_t npanels, int ncu) {
  int rb = SP_RB;
  while (rb > 32 && ((rows_pad + rb - 1) / rb) * npanels < 2 * int64_t(ncu)) rb /= 2;
  return rb;
}

// Panel and tile tables of a set of view widths, on the device, kept with the handle: they depend on the widths only (the
// views' pointers and strides travel as a kernel argument), so a training loop or a fit loop uploads them once.
struct SplitTables {
  const SplitPanel* panels;
  const SplitTile* tiles;
  const GramTile* gtiles;      // the same tiles for loss.hip's preparation kernel (out_row / out_col / wa / wb / diag; a, b unused)
  int np, ntiles;
};

static SplitTables split_tables(ccz_ctx* c, const int64_t* cols, int n_views) {
  Impl* im = impl(c);
  uint64_t key = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { key ^= v; key *= 1099511628211ull; };
  mix(uint64_t(n_views));
  int np = 0;
  for (int v = 0; v < n_views; ++v) { mix(uint64_t(cols[v])); np += int((cols[v] + SP_T - 1) / SP_T); }
  SplitTables tb;
  tb.np = np;
  for (auto& e : im->split_tabs)
    if (e.key == key && e.np == np) {                   // the hot path: a lookup, no allocation
      tb.panels = static_cast<const SplitPanel*>(e.panels);
      tb.tiles = static_cast<const SplitTile*>(e.tiles);
      tb.gtiles = static_cast<const GramTile*>(e.gtiles);
      tb.ntiles = e.ntiles;
      return tb;
    }
  std::vector<SplitPanel> panels;
  int64_t g0 = 0;
  for (int v = 0; v < n_views; ++v) {
    for (int64_t c0 = 0; c0 < cols[v]; c0 += SP_T)
      panels.push_back(SplitPanel{int32_t(v), int32_t(c0), int32_t(std::min<int64_t>(SP_T, cols[v] - c0)), int32_t(g0 + c0)});
    g0 += cols[v];
  }
  std::vector<SplitTile> tiles;
  std::vector<GramTile> gt;
  for (const auto& ij : split_tile_order(tb.np)) {
    SplitTile t;
    t.pa = ij.first; t.pb = ij.second;
    t.out_row = panels[ij.first].gcol0; t.out_col = panels[ij.second].gcol0;
    t.wa = panels[ij.first].width; t.wb = panels[ij.second].width;
    t.diag = ij.first == ij.second ? 1 : 0;
    t.pad_ = 0;
    tiles.push_back(t);
    GramTile g{};
    g.a = nullptr; g.b = nullptr; g.lda = 0; g.ldb = 0;
    g.out_row = t.out_row; g.out_col = t.out_col; g.wa = t.wa; g.wb = t.wb; g.diag = t.diag; g.pad_ = 0;
    gt.push_back(g);
  }
  tb.ntiles = int(tiles.size());
  if (im->split_tabs.size() >= 32) {                      // a caller cycling through many shapes: start over
    CCZ_HIP(hipStreamSynchronize(stream(c)));
    for (auto& e : im->split_tabs) { (void)hipFree(e.panels); (void)hipFree(e.tiles); (void)hipFree(e.gtiles); }
    im->split_tabs.clear();
  }
  void *dp = nullptr, *dt = nullptr, *dg = nullptr;
  CCZ_HIP(hipMalloc(&dp, std::max<size_t>(panels.size() * sizeof(SplitPanel), 256)));
  CCZ_HIP(hipMalloc(&dt, std::max<size_t>(tiles.size() * sizeof(SplitTile), 256)));
  CCZ_HIP(hipMalloc(&dg, std::max<size_t>(gt.size() * sizeof(GramTile), 256)));
  h2d_small(c, dp, panels.data(), panels.size() * sizeof(SplitPanel));
  h2d_small(c, dt, tiles.data(), tiles.size() * sizeof(SplitTile));
  h2d_small(c, dg, gt.data(), gt.size() * sizeof(GramTile));
  im->split_tabs.push_back({key, dp, dt, dg, tb.np, tb.ntiles});
  tb.panels = static_cast<const SplitPanel*>(dp);
  tb.tiles = static_cast<const SplitTile*>(dt);
  tb.gtiles = static_cast<const GramTile*>(dg);
  return tb;
}

static bool split_views_arg(const ccz_view* views, int n_views, SplitViews* vws) {
  bool aligned = true;
  for (int v = 0; v < n_views; ++v) {
    vws->data[v] = static_cast<const float*>(views[v].data);
    vws->ld[v] = views[v].ld;
    if (views[v].cols % 4 != 0 || views[v].ld % 4 != 0 || reinterpret_cast<uintptr_t>(views[v].data) % 16 != 0) aligned = false;
  }
  return aligned;
}

static void launch_split_pass(ccz_ctx* c, const SplitTables& tb, const SplitViews& vws, bool aligned, int64_t r0, int64_t rows, int64_t ksteps,
                              const float* pilot, char* planes, double* msq, double* colsum, hipStream_t st = nullptr) {
  if (!st) st = stream(c);
  const int rb = split_rows_per_block(ksteps * SP_K, tb.np, std::max(1, impl(c)->props.multiProcessorCount));
  const unsigned nrb = (unsigned)((ksteps * SP_K + rb - 1) / rb);
  // panels fastest in the grid (default; CCZ_SPLIT_ORDER=0: row blocks fastest, the first form; read per call).  Measured at the
  // metric shape, alternating runs (tools/r6_split_order.sh): 12.1 / 10.7 / 10.6 ms against 12.6 / 12.3 / 11.9 -- never slower.
  const char* oe = getenv("CCZ_SPLIT_ORDER");
  const int panel_fast = (!(oe && atoi(oe) == 0) && nrb <= 65535u) ? 1 : 0;
  const dim3 grid = panel_fast ? dim3((unsigned)tb.np, nrb) : dim3(nrb, (unsigned)tb.np);
  const char* xe = getenv("CCZ_SPLIT_XCH");                  // 0: direct 16-byte stores at a stride of 64 bytes (the first form); read per call
  const bool xch = !(xe && atoi(xe) == 0);
  auto kern = aligned ? (xch ? &k_split_bf16x2<true, true> : &k_split_bf16x2<true, false>)
                      : (xch ? &k_split_bf16x2<false, true> : &k_split_bf16x2<false, false>);
  hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, tb.panels, vws, r0, rows, ksteps, pilot, planes, msq, colsum, rb, panel_fast);
}

// The K1-layout planes of ONE fp32 matrix X (rows x cols, ld): panels of 256 columns, `ksteps` k-steps of 16 rows each (rows past
// `rows` are zero) -- the B side of gemm_split.hip's product (X = Gamma: its rows are the contraction index).  No pilot, no msq.
void split_k1_layout(ccz_ctx* c, const float* X, int64_t rows, int64_t cols, int64_t ld, int64_t ksteps, char* planes) {
  const SplitTables tb = split_tables(c, &cols, 1);
  const ccz_view v{X, cols, ld};
  SplitViews vws{};
  const bool aligned = split_views_arg(&v, 1, &vws);
  launch_split_pass(c, tb, vws, aligned, 0, rows, ksteps, nullptr, planes, nullptr, nullptr);
  CCZ_LAUNCH_CHECK();
}

// row chunk per workgroup of the Gram kernel over `ksteps` k-steps: minimise rounds x (steps + fixed cost) like gram.hip::plan_rows;
// whole multiples of 8 chunks on a chip-filling grid select the chunk-per-XCD walk
struct SplitRowPlan { int64_t steps_per_wg, ksplit, nblocks; int per_xcd; };
static SplitRowPlan split_row_plan(int64_t ksteps, int ntiles, int64_t max_steps, int ncu) {
  int64_t best_k = 1;
  const int64_t kmin = std::max<int64_t>(1, (ksteps + max_steps - 1) / max_steps);
  const int64_t kmax = std::max<int64_t>(kmin, std::min<int64_t>(ksteps / 16, 8192));
  double best = 1e300;
  for (int64_t ks = kmin; ks <= kmax; ++ks) {
    const int64_t steps = (ksteps + ks - 1) / ks;
    const bool big = int64_t(ntiles) * ks >= 16 * int64_t(ncu);
    int64_t rounds;
    if (big && ks % 8 == 0) rounds = ((ks / 8) * int64_t(ntiles) + ncu / 8 - 1) / (ncu / 8);
    else if (big) rounds = (int64_t((ntiles + 7) / 8) * ks + ncu / 8 - 1) / (ncu / 8);
    else rounds = (int64_t(ntiles) * ks + ncu - 1) / ncu;
    const double cost = (big && ks % 8 != 0 ? 1.03 : 1.0) * double(rounds) * double(steps + 24);
    if (cost < best) { best = cost; best_k = ks; }
  }
  SplitRowPlan rp;
  rp.steps_per_wg = (ksteps + best_k - 1) / best_k;
  rp.ksplit = (ksteps + rp.steps_per_wg - 1) / rp.steps_per_wg;
  const bool sliced = int64_t(ntiles) * rp.ksplit >= 16 * int64_t(ncu);
  const char* e_walk = getenv("CCZ_SPLIT_WALK");          // A/B: 1 = every XCD on the SAME row chunk (slices of the tile list) also when ksplit % 8 == 0
  const bool xchunks = sliced && rp.ksplit % 8 == 0 && !(e_walk && atoi(e_walk) == 1);
  rp.per_xcd = xchunks ? -1 : (sliced ? (ntiles + 7) / 8 : 0);
  rp.nblocks = xchunks ? int64_t(ntiles) * rp.ksplit : (sliced ? int64_t(8) * rp.per_xcd * rp.ksplit : int64_t(ntiles) * rp.ksplit);
  if (rp.nblocks > 0x7fffffffLL) fail(CCZ_EUNSUP, "gram (split route): grid too large");
  if ((rp.steps_per_wg + SP_NST) * SP_PSTEP > 0x7fffffffLL) fail(CCZ_EUNSUP, "gram (split route): row chunk too long");
  return rp;
}

// Loss fast path on the split route (loss.hip: pair_core -> k_loss_prep_partials): the pilot-shifted Gram of a DCCA batch as per-(row
// chunk, tile) fp32 partial tiles -- the layout of gram.hip's staged kernel (slot = chunk * ntiles + tile, full tiles, no plan) --
// plus the exact column sums, the pilot and sum mid^2 for the diagonal.  Three launches: k_colsum_pilot (which also clears msq),
// the split pass, the MFMA kernel; nothing is reduced here (the consumer adds the chunks up in fp64 while it forms Ce).
// Default: from 32768 rows on (K1's own rule: there the route is at least as accurate as the fp32 kernel).  CCZ_LOSS_K1_SPLIT=2 takes
// it from 4096 rows on (8192 rows: 3.5e-7 against the fp32 kernel's 2.2e-7 -- the reference forms these moments by a float32 matmul
// and inverts them through a float32 eigh, cca_zoo/deep/objectives.py:9-21, :86-97, both looser): measured at BASELINE configs[3]
// (batch 8192, 2 x 512) the loss alone gets faster (0.63 -> 0.56 ms with the split backward) but a TRAINING STEP gets slower
// (2.50 -> 2.60 ms): the bf16 MFMA bursts pull the chip's clock down and the encoders' GEMMs next to them run ~10 % longer
// (profiles/r06_loss_c4.md).  So small batches keep the fp32 kernels unless asked.  CCZ_LOSS_K1_SPLIT=0: never.
bool gram_partials_split_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, GramPartials* out) {
  const char* e_on = getenv("CCZ_LOSS_K1_SPLIT");
  const int mode = e_on ? atoi(e_on) : 1;
  if (mode == 0 || c->k1_route == CCZ_K1_FP32) return false;
  if (n_views < 1 || n_views > 8 || n < (mode >= 2 ? 4096 : 32768)) return false;
  int64_t D = 0;
  int64_t cols[8];
  for (int v = 0; v < n_views; ++v) { cols[v] = views[v].cols; D += views[v].cols; }
  if (D < 512 || (D + 255) / 256 > 64 || double(n) * double(D) * double(D + 1) < 5e9) return false;
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  const int ncu = std::max(1, im->props.multiProcessorCount);
  const SplitTables tb = split_tables(c, cols, n_views);
  const int64_t ksteps = (n + SP_K - 1) / SP_K;
  const SplitRowPlan rp = split_row_plan(ksteps, tb.ntiles, 16384 / SP_K, ncu);
  static const int64_t partial_cap = [] { const char* e = getenv("CCZ_GRAM_PARTIAL_MB"); return (e ? atoll(e) : 192LL) << 20; }();
  const int64_t partial_bytes = rp.ksplit * int64_t(tb.ntiles) * (SP_T * SP_T * 4);
  if (rp.per_xcd != 0 || partial_bytes > partial_cap) return false;           // a chip-filling grid: the general route through ccz_moments
  SplitViews vws{};
  const bool aligned = split_views_arg(views, n_views, &vws);
  double* msq = static_cast<double*>(dev_alloc(c, size_t(D) * 8));
  out->msq = msq;
  out->colsum = static_cast<double*>(dev_alloc(c, size_t(D) * 8));
  out->pilot = static_cast<float*>(dev_alloc(c, size_t(D) * 4));
  out->partial = static_cast<float*>(dev_alloc(c, size_t(partial_bytes)));
  out->planes = static_cast<char*>(dev_alloc(c, size_t(tb.np) * size_t(ksteps) * SP_PSTEP));
  out->tile_plan = nullptr;
  if (!colsum_pilot_f32(c, views, n_views, n, D, out->colsum, out->pilot, msq, D)) {
    gram_partials_release(c, out);
    return false;
  }
  launch_split_pass(c, tb, vws, aligned, 0, n, ksteps, out->pilot, out->planes, msq, nullptr);
  const size_t fifo_bytes = size_t(SP_NST) * SP_STAGE;
  sp_allow_lds(reinterpret_cast<const void*>(&k_gram_bf16x2), c->device, int(fifo_bytes));
  hipLaunchKernelGGL(k_gram_bf16x2, dim3((unsigned)rp.nblocks), dim3(256), fifo_bytes, st, tb.tiles, tb.ntiles, rp.per_xcd, rp.ksplit, out->planes, ksteps,
                     rp.steps_per_wg, out->partial);
  CCZ_LAUNCH_CHECK();
  out->tiles = tb.gtiles;
  out->ntiles = tb.ntiles;
  out->ksplit = rp.ksplit;
  c->last_pilot = 1;
  c->last_route = CCZ_K1_BF16X2;
  return true;
}

// Does the split route pay for this launch?  (auto mode)  It carries an HBM pass over the rows, a reduce over the partial
// tiles and two small table uploads; below ~1e11 algorithmic flops the fp32 kernel's single launch wins.
bool gram_split_worthwhile(int64_t n, int64_t D) {
  const char* e_fl = getenv("CCZ_SPLIT_MIN_FLOP");
  const double min_flop = e_fl ? atof(e_fl) : 1e11;
  // n >= 32768: from there on the route's error against float64 moments is at or below the fp32 kernel's on the same rows
  // (the dropped 2^-16 terms average out over the rows, the fp32 kernel's accumulation error grows with them; measured
  // crossover 20k - 30k rows, tools/k1_route_check.py) -- below it the route is only taken when asked for
  return n >= 32768 && D >= 256 && double(n) * double(D) * double(D + 1) >= min_flop;
}

// Row pieces of one launch (in rows; whole units of a workgroup's row chunk except the last) -- an A/B switch, OFF by default.
// The split pass is an HBM pass (13 ms of a 160 ms K1 at the metric shape) in front of a power-bound MFMA kernel, so round 6 tried to
// hide it: cut the launch into a short leading piece and the rest, and run the split pass of piece p + 1 on a side stream under the
// MFMA kernel of piece p.  Measured at n = 1e6, 2 x 4096 (profiles/r06_split_pipe_sweep.log): one piece 156.6 - 157.6 ms; piped with
// the side stream on 32 / 64 / 128 CUs 199.5 / 180.0 / 168.4 ms, unmasked 159.1 ms (the pass's small workgroups fill every CU that
// frees up and the two kernels serialise).  The MFMA kernel loses MORE than the pass's duration whenever the two really overlap: it
// scales with the CUs it holds even at the power limit, and the pass's 4.7 TB/s stream evicts the panels it shares through L2.
// CCZ_SPLIT_PIPE: unset / "0" = one piece; "f0,f1,.." = the leading pieces as fractions of the launch's rows (launches below 8 units =
// 131072 rows stay whole).
static std::vector<int64_t> split_pieces(int64_t rows, int64_t unit) {
  std::vector<double> fr;
  if (const char* e = getenv("CCZ_SPLIT_PIPE")) {
    fr.clear();
    for (const char* q = e; *q;) {
      char* end = nullptr;
      const double f = strtod(q, &end);
      if (end == q) break;
      if (f > 0.0 && f < 1.0) fr.push_back(f);
      q = *end == ',' ? end + 1 : end;
      if (end && *end != ',' ) break;
    }
  }
  std::vector<int64_t> out;
  const int64_t T = (rows + unit - 1) / unit;
  int64_t used = 0;
  if (T >= 8 && fr.size() <= 6) {
    for (double f : fr) {
      const int64_t u = std::max<int64_t>(1, int64_t(f * double(T) + 0.5));
      if (used + u >= T) break;
      out.push_back(u * unit);
      used += u;
    }
  }
  out.push_back(rows - used * unit);
  return out;
}

// The side stream of the piped launch (CCZ_SPLIT_PIPE), confined to a few CUs of every XCD when the runtime takes a CU mask: the
// MFMA kernel's workgroups take whole CUs (512 VGPRs per lane, 128 KiB of LDS), so without the mask the small workgroups of the pass
// fill every CU that frees up and the two kernels run one after the other.
// CCZ_SPLIT_PIPE_CUS: CUs of the side stream (default 64 = 8 per XCD; 0: no mask).  nullptr: no second stream (one piece then).
static hipStream_t split_side_stream(ccz_ctx* c) {
  Impl* im = impl(c);
  const int ncu = std::max(1, im->props.multiProcessorCount);
  const char* e = getenv("CCZ_SPLIT_PIPE_CUS");                  // read per call (an A/B switch): a new width makes a new stream
  const int cus = e ? atoi(e) : 64;
  if (im->split_stream_tried && cus == im->split_stream_req) return im->split_stream;
  if (im->split_stream) {
    (void)hipStreamSynchronize(im->split_stream);
    (void)hipStreamDestroy(im->split_stream);
    im->split_stream = nullptr;
  }
  im->split_stream_tried = true;
  im->split_stream_req = cus;
  im->split_stream_cus = 0;
  hipStream_t st = nullptr;
  if (cus > 0 && cus < ncu && ncu % 8 == 0) {
    // bit i set iff (i / 8) % stride == 0: 8-bit groups, every stride-th one -- uniform over the XCDs whether the runtime deals the
    // mask's bits round-robin over the XCDs (bit i -> XCD i % 8) or XCD by XCD
    const int stride = std::max(1, ncu / cus);
    std::vector<uint32_t> mask(size_t((ncu + 31) / 32), 0u);
    int set = 0;
    for (int i = 0; i < ncu; ++i)
      if ((i / 8) % stride == 0) { mask[size_t(i >> 5)] |= 1u << (i & 31); ++set; }
    if (hipExtStreamCreateWithCUMask(&st, uint32_t(mask.size()), mask.data()) == hipSuccess) im->split_stream_cus = set;
    else { (void)hipGetLastError(); st = nullptr; }
  }
  if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
  im->split_stream = st;
  return st;
}

// G (upper tiles) += sum over rows of d d' for d = x - pilot (pilot may be null: d = x), through the split-bf16 route;
// colsum (may be null) += the exact fp64 column sums of x, gathered by the split pass on its way over the rows.
// Everything is enqueued on the handle's stream; with time_it the three stages are timed with HIP events (one host wait per
// row super-chunk) into c->last_split_ms / last_mfma_ms / last_reduce_ms.
void gram_split_f32(ccz_ctx* c, const ccz_view* views, int n_views, int64_t n, double* G, int64_t D, const float* pilot, double* colsum,
                    bool time_it) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  // ---- panels and tiles (device tables cached with the handle by the views' widths) ----
  if (n_views > SP_MAXV) fail(CCZ_EUNSUP, "gram (split route): more than %d views", SP_MAXV);
  if (D > 0x7fffffffLL - 256) fail(CCZ_EUNSUP, "gram (split route): stacked width too large");
  int64_t cols[SP_MAXV];
  for (int v = 0; v < n_views; ++v) cols[v] = views[v].cols;
  const SplitTables tb = split_tables(c, cols, n_views);
  SplitViews vws{};
  const bool aligned = split_views_arg(views, n_views, &vws);
  const int np = tb.np, ntiles = tb.ntiles;
  const SplitTile* d_tiles = tb.tiles;
  double* msq = static_cast<double*>(dev_alloc(c, size_t(D) * 8));
  char* planes = nullptr;
  float* partial = nullptr;
  bool side_busy = false;                  // split passes enqueued on the side stream that the main stream has not waited for yet
  auto release = [&] {
    // (unwinding between the side stream's launches and the main stream's waits: the pool recycles in the order of the MAIN stream)
    if (side_busy && im->split_stream) (void)hipStreamSynchronize(im->split_stream);
    if (partial) dev_free(c, partial);
    if (planes) dev_free(c, planes);
    dev_free(c, msq);
  };
  try {
    // ---- rows per launch (scratch budget) and per workgroup ----
    const int ncu = std::max(1, im->props.multiProcessorCount);
    const char* rows_env = getenv("CCZ_SPLIT_ROWS");       // fp32 accumulation length (rows per workgroup), default 16384
    const int64_t max_steps = std::max<int64_t>(1, (rows_env ? atoll(rows_env) : 16384LL) / SP_K);
    const double per_row = double(np) * SP_PSTEP / SP_K + double(ntiles) * (SP_T * SP_T * 4) / double(max_steps * SP_K);
    const size_t budget = split_scratch_budget(c);
    int64_t launch_rows = int64_t(double(budget) / per_row) / (max_steps * SP_K) * (max_steps * SP_K);
    launch_rows = std::max<int64_t>(launch_rows, max_steps * SP_K);
    const int64_t n_launch = (n + launch_rows - 1) / launch_rows;
    launch_rows = ((n + n_launch - 1) / n_launch + SP_K - 1) / SP_K * SP_K;       // equal super-chunks
    const size_t fifo_bytes = size_t(SP_NST) * SP_STAGE;
    sp_allow_lds(reinterpret_cast<const void*>(&k_gram_bf16x2), c->device, int(fifo_bytes));
    c->last_split_ms = c->last_mfma_ms = c->last_reduce_ms = 0.0;
    size_t planes_cap = 0, partial_cap = 0;
    bool launched = false;
   retry_smaller:
    try {
    for (int64_t r0 = 0; r0 < n; r0 += launch_rows) {
      const int64_t rows = std::min(launch_rows, n - r0);
      // row pieces of this super-chunk: the split pass of piece p + 1 runs on the side stream under the MFMA kernel of piece p
      std::vector<int64_t> piece_rows = split_pieces(rows, max_steps * SP_K);
      hipStream_t side = piece_rows.size() > 1 ? split_side_stream(c) : nullptr;
      if (!side) piece_rows.assign(1, rows);
      const int npc = int(piece_rows.size());
      struct Piece { int64_t r0, rows, ksteps, slice0; size_t planes_off; SplitRowPlan rp; };
      std::vector<Piece> pcs;
      pcs.resize(size_t(npc));
      size_t planes_bytes = 0;
      int64_t slices = 0;
      {
        int64_t off = 0;
        for (int p = 0; p < npc; ++p) {
          Piece& pc = pcs[size_t(p)];
          pc.r0 = r0 + off;
          pc.rows = piece_rows[size_t(p)];
          pc.ksteps = (pc.rows + SP_K - 1) / SP_K;
          pc.rp = split_row_plan(pc.ksteps, ntiles, max_steps, ncu);
          pc.planes_off = planes_bytes;
          pc.slice0 = slices;
          planes_bytes += size_t(np) * size_t(pc.ksteps) * SP_PSTEP;
          slices += pc.rp.ksplit;
          off += pc.rows;
        }
      }
      const size_t partial_bytes = size_t(slices) * size_t(ntiles) * (SP_T * SP_T * 4);
      if (planes_bytes > planes_cap) { if (planes) dev_free(c, planes); planes = static_cast<char*>(dev_alloc(c, planes_bytes)); planes_cap = planes_bytes; }
      if (partial_bytes > partial_cap) { if (partial) dev_free(c, partial); partial = static_cast<float*>(dev_alloc(c, partial_bytes)); partial_cap = partial_bytes; }
      // events: [0] before split 0, [1] after it (the side stream's go-ahead), [4p + 2] after the MFMA kernel of piece p; per piece
      // p >= 1: [4p - 1] / [4p] before / after its split (side stream), [4p + 1] after the main stream's wait; [4 npc - 1] after the reduce
      const size_t nev = size_t(4 * npc);
      while (im->sp_ev.size() < nev) {
        hipEvent_t e = nullptr;
        CCZ_HIP(hipEventCreate(&e));
        im->sp_ev.push_back(e);
      }
      hipEvent_t* ev = im->sp_ev.data();
      launched = true;                     // (from here on the super-chunks are no larger than this one: no further allocation)
      side_busy = side != nullptr;
      zero(c, msq, size_t(D) * 8);
      if (time_it) CCZ_HIP(hipEventRecord(ev[0], st));
      launch_split_pass(c, tb, vws, aligned, pcs[0].r0, pcs[0].rows, pcs[0].ksteps, pilot, planes, msq, colsum, st);
      if (time_it || side) CCZ_HIP(hipEventRecord(ev[1], st));
      if (side) {
        // (the go-ahead also orders the side stream behind the zeroing of msq, the pilot and the previous super-chunk's readers of `planes`)
        CCZ_HIP(hipStreamWaitEvent(side, ev[1], 0));
        for (int p = 1; p < npc; ++p) {
          const Piece& pc = pcs[size_t(p)];
          if (time_it) CCZ_HIP(hipEventRecord(ev[4 * p - 1], side));
          launch_split_pass(c, tb, vws, aligned, pc.r0, pc.rows, pc.ksteps, pilot, planes + pc.planes_off, msq, colsum, side);
          CCZ_HIP(hipEventRecord(ev[4 * p], side));
        }
      }
      for (int p = 0; p < npc; ++p) {
        const Piece& pc = pcs[size_t(p)];
        if (p > 0) {
          CCZ_HIP(hipStreamWaitEvent(st, ev[4 * p], 0));
          if (time_it) CCZ_HIP(hipEventRecord(ev[4 * p + 1], st));
        }
        hipLaunchKernelGGL(k_gram_bf16x2, dim3((unsigned)pc.rp.nblocks), dim3(256), fifo_bytes, st, d_tiles, ntiles, pc.rp.per_xcd, pc.rp.ksplit,
                           planes + pc.planes_off, pc.ksteps, pc.rp.steps_per_wg, partial + pc.slice0 * int64_t(ntiles) * (SP_T * SP_T));
        if (time_it) CCZ_HIP(hipEventRecord(ev[4 * p + 2], st));
      }
      side_busy = false;                   // (the main stream has waited for every split pass of the side stream)
      hipLaunchKernelGGL(k_split_reduce, dim3(64, (unsigned)ntiles), dim3(256), 0, st, partial, d_tiles, ntiles, slices, G, D, msq);
      CCZ_LAUNCH_CHECK();
      if (time_it) {
        hipEvent_t last = ev[4 * npc - 1];
        CCZ_HIP(hipEventRecord(last, st));
        CCZ_HIP(hipEventSynchronize(last));
        float a = 0.f;
        CCZ_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));
        c->last_split_ms += a;
        CCZ_HIP(hipEventElapsedTime(&a, ev[1], ev[2]));
        c->last_mfma_ms += a;
        for (int p = 1; p < npc; ++p) {
          CCZ_HIP(hipEventElapsedTime(&a, ev[4 * p - 1], ev[4 * p]));
          c->last_split_ms += a;             // (under the previous piece's MFMA kernel: the stage times of a piped launch overlap)
          CCZ_HIP(hipEventElapsedTime(&a, ev[4 * p + 1], ev[4 * p + 2]));
          c->last_mfma_ms += a;
        }
        CCZ_HIP(hipEventElapsedTime(&a, ev[4 * (npc - 1) + 2], last));
        c->last_reduce_ms += a;
      }
    }
    } catch (const Error& e) {
      // the scratch did not fit (another library holds most of HBM): halve the row super-chunk and start over -- nothing has
      // been enqueued yet when the FIRST super-chunk's buffers cannot be allocated
      if (e.code != CCZ_ENOMEM || launched || launch_rows <= max_steps * SP_K) throw;
      if (planes) { dev_free(c, planes); planes = nullptr; }
      if (partial) { dev_free(c, partial); partial = nullptr; }
      planes_cap = partial_cap = 0;
      launch_rows = std::max<int64_t>(max_steps * SP_K, launch_rows / 2 / (max_steps * SP_K) * (max_steps * SP_K));
      goto retry_smaller;
    }
  } catch (...) {
    release();
    throw;
  }
  release();
}

}  // namespace ccz
