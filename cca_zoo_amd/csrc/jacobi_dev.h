// Device-side helpers shared by the Jacobi kernels (ops_hip.hip: the one-workgroup eigen-solvers of the Rayleigh-Ritz
// steps; evd_block.hip: the blocked two-sided / one-sided Jacobi for d > 160).
#pragma once

#include <hip/hip_runtime.h>

namespace ccz {

__device__ __forceinline__ double rsq_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double rsq_nr3(double x) {      // the cosine: c^2 + s^2 = 1 must hold to rounding
  double y = rsq_nr(x);
  return y * fma(-0.5 * x * y, y, 1.5);
}
typedef double jac_cs __attribute__((ext_vector_type(2)));   // (cosine, sine) of one rotation
__device__ __forceinline__ void pair_of(int round, int k, int m1, int& a, int& b) {
  if (k == 0) { a = m1; b = round; return; }
  a = round + k; if (a >= m1) a -= m1;
  b = round - k; if (b < 0) b += m1;
}

// (c, s) of the rotation that annihilates h_pq, from the pair's three entries scaled by 1 / max|H|.
// A dependent fp64 VALU operation costs 32 cycles on gfx950 (measured: tools/probes/clock_probe.hip), and this chain is
// the serial part of every round, so it is written for depth, not for operation count:
//   u = |al| / r,  r = sqrt(al^2 + h^2),  al = (h_qq - h_pp) / 2:   c = sqrt((1 + u) / 2),  s = sgn(al) h / (2 r c)
// with two reciprocal square roots (1 / r and 1 / c) -- 18 dependent operations instead of the 31 of
// t = h / (|al| + r), c = 1 / sqrt(1 + t^2), s = t c.  c^2 + s^2 = (1 + u)/2 + (1 - u)/2 holds to rounding.
__device__ __forceinline__ jac_cs jac_rotation(double hpp, double hqq, double hpq, double ih) {
  const double al = 0.5 * (hqq - hpp) * ih, hq = hpq * ih;
  const double x = fma(al, al, hq * hq);
  const double y = rsq_nr(x);                              // 1 / r
  const double c2 = fma(0.5 * fabs(al), y, 0.5);           // c^2 = (1 + |al| / r) / 2  in [1/2, 1]
  const double z = rsq_nr3(c2);                            // 1 / c
  jac_cs r;
  r.x = c2 * z;
  r.y = (al >= 0.0 ? 0.5 : -0.5) * hq * y * z;
  return r;
}


// Workgroup barrier that orders LDS traffic only (see the note in ops_hip.hip above k_syev_packed).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int tri_off(int i, int j) {
  const int hi = max(i, j), lo = min(i, j);
  return (__mul24(hi, hi + 1) >> 1) + lo;                   // (24-bit multiply: full rate; the indices are < 2^12)
}
__device__ __forceinline__ void tri_decode(int e, int& i, int& j) {   // e = i (i + 1) / 2 + j, j <= i
  i = int((sqrtf(8.0f * float(e) + 1.0f) - 1.0f) * 0.5f);
  while (i * (i + 1) / 2 > e) --i;
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  j = e - i * (i + 1) / 2;
}


}  // namespace ccz
