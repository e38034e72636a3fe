// transform / score on the bf16 matrix pipe:  out (n x k, fp32) = (X - 1 mean') W  for k <= 64 directions
// (cca_zoo/_base.py:108-123: the projection of every view onto its weights).
//
// The shape is HBM-bound by its algorithmic bytes (n d 4) -- but only if the arithmetic keeps up: at k = 64 the fp32 matrix
// pipe alone needs 2 n d k / 157 TF = 3.3 ms per 1e6 x 4096 view against 2.7 ms of HBM time, and the fp32 kernel
// (gemm_big.hip: k_gemm_f32_nn_tall) measures 5.4 - 5.8 ms.  Here the product runs with the split arithmetic of K1
// (split_mma.h: x - p = hi + mid in bf16, hi hi + hi mid + mid hi as three MFMAs into one fp32 accumulator), which needs
// 0.7 ms of matrix-pipe time -- and, unlike K1, needs NO transposing pass: the contraction runs over the columns of X, so
// the 8 consecutive k of a lane are 32 contiguous bytes of a row.  Each wave streams its own 64 rows with LDS-DMA
// (16-byte pieces, lane l <- row l % 32, columns 8 (l / 32) + 4 piece ..) into a wave-private 4-slot ring, converts in
// registers (8 subtractions of the pilot p = fl32(mean), two v_cvt_pk_bf16_f32 rounds: 32 VALU operations per 6 MFMAs) and
// multiplies with the W planes (prepared once per call: d x 64 x 2 planes, 1 MB at d = 4096, streamed through the same ring
// from L2).  No barriers: the four waves of a workgroup are independent pipelines (the fp32 FIFO kernel's scheme).
// The exact centring is finished in the epilogue: corr_j = sum_c (mean_c - p_c) W_cj (fp64, formed with the planes).
#include <algorithm>
#include <cstdlib>

#include "hip_common.h"
#include "split_mma.h"

namespace ccz {

constexpr int PJ_MT = 2;                          // row tiles (32 samples) per wave
constexpr int PJ_XB = PJ_MT * 2048;               // bytes of X per slot
constexpr int PJ_MAXD = 8128;                     // the pilot row lives in LDS behind the rings (<= 128 KiB of rings + 4 d bytes <= 160 KiB)
// PLANES = 2: x = hi + mid, three products (error of an output ~ 2^-17 of its scale: the dropped hi lo / mid mid terms do not
//             average out in a projection as they do in K1's coherent sums: measured 4.5e-6 against the fp32 kernel's 1.3e-6);
// PLANES = 3: x = hi + mid + lo, five products (+ hi lo + lo hi; mid mid ~ 2^-18 stays out): at the fp32 kernel's accuracy.
template <int PLANES> struct PJ {
  static constexpr int R = PLANES == 2 ? 4 : 3;              // ring slots per wave
  static constexpr int WB = PLANES * 2048;                   // bytes of W planes per k-step: [plane][j tile 2][1 KiB]
  static constexpr int SLOT = PJ_XB + WB;
  static constexpr int PER = 2 * PJ_MT + PLANES * 2;         // DMAs per k-step
};

// W (d x k, fp64, ld ldw) -> planes[k-step s][plane H|M][j tile 2][k half 2][j 32][k 8] bf16 (columns >= k are zero),
// pilot[c] = fl32(mean[c]) and corr[j] += sum_c (mean_c - pilot_c) W_cj.  grid = d / 16 blocks of 128 threads.
__global__ __launch_bounds__(128) void k_project_prep(const double* __restrict__ W, int64_t d, int k, int64_t ldw, const double* __restrict__ mean,
                                                      char* __restrict__ planes, float* __restrict__ pilot, double* __restrict__ corr, int nplanes) {
  const int s = blockIdx.x, t = threadIdx.x;
  const int jt = t >> 6, l = t & 63, h = l >> 5, j = jt * 32 + (l & 31);
  float v[8];
  double cpart = 0.0;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int64_t c = int64_t(s) * 16 + 8 * h + kk;
    const double w = (j < k && c < d) ? W[c * ldw + j] : 0.0;
    v[kk] = float(w);
    if (mean && c < d) {
      const double m = mean[c];
      cpart += (m - double(float(m))) * w;
    }
  }
  sp_v4u32 hw, mw, lw;
#pragma unroll
  for (int kk = 0; kk < 8; kk += 2) {
    const unsigned hb = sp_pack2(v[kk], v[kk + 1]);
    hw[kk >> 1] = hb;
    const float r0 = v[kk] - __builtin_bit_cast(float, hb << 16), r1 = v[kk + 1] - __builtin_bit_cast(float, hb & 0xffff0000u);
    const unsigned mb = sp_pack2(r0, r1);
    mw[kk >> 1] = mb;
    lw[kk >> 1] = sp_pack2(r0 - __builtin_bit_cast(float, mb << 16), r1 - __builtin_bit_cast(float, mb & 0xffff0000u));
  }
  char* dst = planes + int64_t(s) * (nplanes * 2048) + jt * 1024 + l * 16;
  *reinterpret_cast<sp_v4u32*>(dst) = hw;
  *reinterpret_cast<sp_v4u32*>(dst + 2048) = mw;
  if (nplanes == 3) *reinterpret_cast<sp_v4u32*>(dst + 4096) = lw;
  if (t < 16) {
    const int64_t c = int64_t(s) * 16 + t;
    if (c < d) pilot[c] = mean ? float(mean[c]) : 0.f;
  }
  if (mean && j < k) unsafeAtomicAdd(corr + j, cpart);
}

template <int PLANES>
__global__ __launch_bounds__(256, 1) void k_project_split(const float* __restrict__ X, int64_t n, int64_t d, int64_t ld, const float* __restrict__ pilot,
                                                           const char* __restrict__ planes, const double* __restrict__ corr, float* __restrict__ out,
                                                           int64_t ldo, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the pilot row behind the rings (read by every wave: one barrier, the only one)
  constexpr int PJ_R = PJ<PLANES>::R, PJ_SLOT = PJ<PLANES>::SLOT, PJ_WB = PJ<PLANES>::WB;
  float* pl = reinterpret_cast<float*>(smem + 4 * PJ_R * PJ_SLOT);
  for (int64_t c = tid; c < d; c += 256) pl[c] = pilot[c];
  __syncthreads();
  const int64_t m0 = (int64_t(blockIdx.x) * 4 + wave) * (32 * PJ_MT);
  if (m0 >= n) return;
  const int64_t rows = min<int64_t>(32 * PJ_MT, n - m0);
  const int nsteps = int(d / 16);
  char* ring = smem + wave * (PJ_R * PJ_SLOT);
  const char* rd = ring + lane * 16;
  const __amdgpu_buffer_rsrc_t srcX = panel_rsrc(X + m0 * ld, ((rows - 1) * ld + d) * 4);
  const __amdgpu_buffer_rsrc_t srcW = panel_rsrc(planes, int64_t(nsteps) * PJ_WB);
  int voffX[PJ_MT][2];
#pragma unroll
  for (int mt = 0; mt < PJ_MT; ++mt)
#pragma unroll
    for (int p = 0; p < 2; ++p) voffX[mt][p] = int(((mt * 32 + (lane & 31)) * ld + 8 * (lane >> 5) + 4 * p) * 4);
  const int voffW = lane * 16;

  sp_v16f32 acc[PJ_MT][2];
#pragma unroll
  for (int i = 0; i < PJ_MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one k-step's DMAs: 2 PJ_MT pieces of X (rows past the shard, steps past the end: zeros) + 4 KiB of W planes
  auto dma = [&](int slot, int step) {
    const int soX = step * 64, soW = step * PJ_WB;
#pragma unroll
    for (int mt = 0; mt < PJ_MT; ++mt)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srcX, (sp_lds_ptr)(ring + slot * PJ_SLOT + (mt * 2 + p) * 1024), 16, voffX[mt][p], soX, 0, 0);
#pragma unroll
    for (int i = 0; i < 2 * PLANES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srcW, (sp_lds_ptr)(ring + slot * PJ_SLOT + PJ_XB + i * 1024), 16, voffW, soW + i * 1024, 0, 0);
  };
  static_assert(PJ<PLANES>::PER * (PJ<PLANES>::R - 2) == (PLANES == 2 ? 16 : 10), "the counted waits below assume this many DMAs in flight");
#pragma unroll
  for (int s = 0; s < PJ_R - 1; ++s) dma(s, s);
  const int nloop = (nsteps + PJ_R - 1) / PJ_R;
  for (int it = 0; it < nloop; ++it) {
#pragma unroll
    for (int u = 0; u < PJ_R; ++u) {
      const int step = it * PJ_R + u;
      // step's data: everything but the two newest k-steps has landed
      if (PLANES == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // 2 x 8
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                  // 1 x 10
      if (step >= nsteps) {                       // (ring periods past the last k-step: keep the DMA count, skip the arithmetic)
        dma((u + PJ_R - 1) % PJ_R, step + PJ_R - 1);
        continue;
      }
      const char* sl = rd + u * PJ_SLOT;
      const sp_v4f32 p0 = *reinterpret_cast<const sp_v4f32*>(pl + step * 16 + 8 * (lane >> 5));
      const sp_v4f32 p1 = *reinterpret_cast<const sp_v4f32*>(pl + step * 16 + 8 * (lane >> 5) + 4);
      sp_v8bf16 wh[2], wm[2], wl[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        wh[jt] = *reinterpret_cast<const sp_v8bf16*>(sl + PJ_XB + jt * 1024);
        wm[jt] = *reinterpret_cast<const sp_v8bf16*>(sl + PJ_XB + 2048 + jt * 1024);
        if (PLANES == 3) wl[jt] = *reinterpret_cast<const sp_v8bf16*>(sl + PJ_XB + 4096 + jt * 1024);
      }
      sp_v4f32 xa[PJ_MT], xb[PJ_MT];
#pragma unroll
      for (int mt = 0; mt < PJ_MT; ++mt) {
        xa[mt] = *reinterpret_cast<const sp_v4f32*>(sl + (mt * 2) * 1024);
        xb[mt] = *reinterpret_cast<const sp_v4f32*>(sl + (mt * 2 + 1) * 1024);
      }
      // refill the slot of step - 1 with step + R - 1 (its fragments were consumed a whole k-step ago)
      dma((u + PJ_R - 1) % PJ_R, step + PJ_R - 1);
#pragma unroll
      for (int mt = 0; mt < PJ_MT; ++mt) {
        const sp_v4f32 da = xa[mt] - p0, db = xb[mt] - p1;
        const float dv[8] = {da[0], da[1], da[2], da[3], db[0], db[1], db[2], db[3]};
        sp_v4u32 hw, mw, lw;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned hb = sp_pack2(dv[2 * q], dv[2 * q + 1]);
          hw[q] = hb;
          const float r0 = dv[2 * q] - __builtin_bit_cast(float, hb << 16), r1 = dv[2 * q + 1] - __builtin_bit_cast(float, hb & 0xffff0000u);
          const unsigned mb = sp_pack2(r0, r1);
          mw[q] = mb;
          if (PLANES == 3) lw[q] = sp_pack2(r0 - __builtin_bit_cast(float, mb << 16), r1 - __builtin_bit_cast(float, mb & 0xffff0000u));
        }
        const sp_v8bf16 xh = __builtin_bit_cast(sp_v8bf16, hw), xm = __builtin_bit_cast(sp_v8bf16, mw);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[jt], xh, acc[mt][jt], 0, 0, 0);
          acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[jt], xh, acc[mt][jt], 0, 0, 0);
          acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[jt], xm, acc[mt][jt], 0, 0, 0);
          if (PLANES == 3) {
            const sp_v8bf16 xl = __builtin_bit_cast(sp_v8bf16, lw);
            acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[jt], xh, acc[mt][jt], 0, 0, 0);
            acc[mt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[jt], xl, acc[mt][jt], 0, 0, 0);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // epilogue: lane l of tile (mt, jt): sample m0 + 32 mt + (l & 31), directions 32 jt + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
#pragma unroll
  for (int mt = 0; mt < PJ_MT; ++mt) {
    const int64_t m = m0 + mt * 32 + (lane & 31);
    if (m >= n) continue;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = jt * 32 + 8 * g + 4 * (lane >> 5);
        if (j >= k) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mt][jt][4 * g + e] - (corr && j + e < k ? float(corr[j + e]) : 0.f);
        float* o = out + m * ldo + j;
        if (j + 3 < k && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
          *reinterpret_cast<sp_v4f32*>(o) = sp_v4f32{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < k) o[e] = v[e];
        }
      }
  }
}

bool project_split_eligible(ccz_ctx* c, int64_t n, int64_t d, int64_t k, int64_t ld, const void* X, int64_t ldo) {
  const char* e_on = getenv("CCZ_PROJECT_SPLIT");
  if (e_on && atoi(e_on) == 0) return false;
  // OPT-IN: only when the handle's route was set to CCZ_K1_BF16X2 explicitly.  A projection is a random-walk sum, so the dropped
  // 2^-17 terms do not average out as in K1's coherent sums: outputs agree with a float64 product to 4.5e-6 (two planes) / 2.9e-6
  // (three planes, five products) of their scale against the fp32 kernel's 1.3e-6 -- inside the path's 1e-3 bar, but narrower
  // than the reference's float32 product, so `auto` keeps the fp32 kernel (4.07 / 4.63 ms against 5.2 ms per 1e6 x 4096 view).
  if (c->k1_route != CCZ_K1_BF16X2) return false;
  if (k < 1 || k > 64 || d % 16 != 0 || d > PJ_MAXD || ld % 4 != 0 || reinterpret_cast<uintptr_t>(X) % 16 != 0 || ldo < k) return false;
  if (double(n) * double(d) < double(int64_t(1) << 26) || n < 32768) return false;        // small projections keep the fp32 kernel
  if ((int64_t(32 * PJ_MT) * ld + d) * 4 > 0x7fffffffLL) return false;
  return true;
}

// out (n x k, fp32, ld ldo) = (X - 1 mean') W;  X fp32 (n x d, ld) on the device, mean (d, may be null) and W (d x k, ld k) float64 on the device
void project_split(ccz_ctx* c, const float* X, int64_t n, int64_t d, int64_t ld, const double* mean, const double* W, int64_t k, float* out,
                   int64_t ldo) {
  hipStream_t st = stream(c);
  const int64_t nsteps = d / 16;
  // two planes / three products by default (4.07 ms, 4.5e-6); CCZ_PROJECT_PLANES=3: three planes / five products (4.63 ms, 2.9e-6)
  const char* e_pl = getenv("CCZ_PROJECT_PLANES");
  const int nplanes = (e_pl && atoi(e_pl) == 3) ? 3 : 2;
  char* planes = static_cast<char*>(dev_alloc(c, size_t(nsteps) * size_t(nplanes) * 2048));
  float* pilot = static_cast<float*>(dev_alloc(c, size_t(d) * 4));
  double* corr = static_cast<double*>(dev_alloc(c, 64 * 8));
  try {
    zero(c, corr, 64 * 8);
    hipLaunchKernelGGL(k_project_prep, dim3((unsigned)nsteps), dim3(128), 0, st, W, d, int(k), k, mean, planes, pilot, corr, nplanes);
    const int64_t wgs = (n + 4 * 32 * PJ_MT - 1) / (4 * 32 * PJ_MT);
    if (nplanes == 2) {
      const size_t lds = size_t(4) * PJ<2>::R * PJ<2>::SLOT + size_t(d) * 4;
      sp_allow_lds(reinterpret_cast<const void*>(&k_project_split<2>), c->device, int(lds));
      hipLaunchKernelGGL(k_project_split<2>, dim3((unsigned)wgs), dim3(256), lds, st, X, n, d, ld, pilot, planes, mean ? corr : nullptr, out, ldo, int(k));
    } else {
      const size_t lds = size_t(4) * PJ<3>::R * PJ<3>::SLOT + size_t(d) * 4;
      sp_allow_lds(reinterpret_cast<const void*>(&k_project_split<3>), c->device, int(lds));
      hipLaunchKernelGGL(k_project_split<3>, dim3((unsigned)wgs), dim3(256), lds, st, X, n, d, ld, pilot, planes, mean ? corr : nullptr, out, ldo, int(k));
    }
    CCZ_LAUNCH_CHECK();
  } catch (...) {
    dev_free(c, corr); dev_free(c, pilot); dev_free(c, planes);
    throw;
  }
  dev_free(c, corr); dev_free(c, pilot); dev_free(c, planes);
}

}  // namespace ccz
