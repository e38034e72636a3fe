// Dependent-chain latencies of ONE wave (shader cycles per step) for the operations the 64 x 64 Cholesky pivot chain is made
// of.  hipcc --offload-arch=gfx950 -O2 lat_probe.hip -o lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k_lat(long long* out, double* sink, int iters) {
  __shared__ double lds[256];
  double x = 1.5 + threadIdx.x * 1e-3, y = 0.75;
  float xf = 1.5f + threadIdx.x * 1e-3f;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  lds[threadIdx.x] = x;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) x = __builtin_fma(x, 1.0000001, 1e-9);                                  // fp64 fma
    if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                                      // rsq_f64 + add
    if (MODE == 2) x = __builtin_amdgcn_rcp(x) + 1.0;                                      // rcp_f64 + add
    if (MODE == 3) xf = __builtin_fmaf(xf, 1.0000001f, 1e-9f);                             // fp32 fma
    if (MODE == 4) xf = __builtin_amdgcn_rsqf(xf) + 1.0f;                                  // rsq_f32 + add
    if (MODE == 5) x = double(__builtin_amdgcn_rsqf(float(x))) + 1.0;                      // cvt, rsq_f32, cvt, add
    if (MODE == 6) x = bcast_lane(x, 5) + 1e-9;                                            // readlane x2 -> fp64 add
    if (MODE == 7) { lds[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = lds[threadIdx.x ^ 1] + 1e-9; }   // LDS round trip
    if (MODE == 8) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); x = acc[0] * 1e-30 + 1.5; }    // mfma -> VALU -> mfma
    if (MODE == 9) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);         // back-to-back dependent mfma (same acc)
    if (MODE == 10) x = __builtin_fma(x, 1.0000001, 1e-9) * 1.0000001;                     // fma + mul (two dependent)
    if (MODE == 11) { const double t = __builtin_amdgcn_rsq(x); const double q = x * t; const double e = __builtin_fma(-q, t, 1.0);
                      x = __builtin_fma(t * e, __builtin_fma(e, 0.375, 0.5), t) + 1.0; }   // full cubic rsqrt + add
    if (MODE == 12) x = __builtin_amdgcn_frexp_mant(x) + 1.0;                              // frexp_mant + add
    if (MODE == 13) x = __builtin_sqrt(x) + 1.0;                                           // IEEE sqrt + add
    if (MODE == 14) { const bool bad = !(x > 0.0); x = (bad ? 1.0 : x) * 1.0000001; }      // compare + select + mul
  }
  const long long c1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = x + xf + acc[0] + acc[1];
  if (threadIdx.x == 0) out[0] = c1 - c0;
}

template <int MODE>
void run(const char* what, long long* d, double* sink) {
  const int iters = 20000;
  long long h = 0;
  hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64), 0, 0, d, sink, iters);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k_lat<MODE>, dim3(1), dim3(64), 0, 0, d, sink, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.1f cycles per step\n", what, double(h) / iters);
}

int main() {
  long long* d; double* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 8 * 4096);
  run<0>("fp64 fma", d, sink);
  run<10>("fp64 fma + mul", d, sink);
  run<1>("v_rsq_f64 + add", d, sink);
  run<2>("v_rcp_f64 + add", d, sink);
  run<11>("cubic rsqrt (rsq + 4 dependent) + add", d, sink);
  run<13>("IEEE sqrt + add", d, sink);
  run<3>("fp32 fma", d, sink);
  run<4>("v_rsq_f32 + add", d, sink);
  run<5>("cvt f64->f32, rsq_f32, cvt back, add", d, sink);
  run<12>("frexp_mant f64 + add", d, sink);
  run<14>("compare + select + mul", d, sink);
  run<6>("readlane pair -> fp64 add", d, sink);
  run<7>("LDS write / wave sync / read + add", d, sink);
  run<9>("mfma f64 16x16x4, dependent accumulate", d, sink);
  run<8>("mfma f64 -> VALU -> mfma", d, sink);
  return 0;
}
