// How fast does ONE CU issue fp64 VALU work?  ILP independent v_fma_f64 chains per lane, W waves per SIMD (one workgroup
// of 4 W waves): shader cycles per fp64 instruction per wave and per SIMD.   hipcc --offload-arch=gfx950 -O2 dp_issue_probe.hip -o dp_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k_dp(long long* out, double* sink, int iters) {
  double x[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) x[j] = 1.0 + threadIdx.x * 1e-9 + j;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = __builtin_fma(x[j], 1.0000001, 1e-9);
  }
  const long long c1 = __builtin_readcyclecounter();
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < ILP; ++j) s += x[j];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}
template <int ILP>
void run(int waves_per_simd, long long* d, double* sink) {
  const int iters = 200000;
  long long h = 0;
  hipLaunchKernelGGL(k_dp<ILP>, dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, d, sink, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double per_wave = double(h) / (double(iters) * ILP);
  printf("ILP %d, %d wave(s)/SIMD: %.2f cycles per fp64 FMA per wave, %.2f per SIMD  (%.1f fp64 FLOP/clk/CU)\n", ILP, waves_per_simd, per_wave,
         per_wave / waves_per_simd, 4.0 * 128.0 * waves_per_simd / per_wave);
}
int main() {
  long long* d; double* sink;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 8 * 4096);
  for (int w : {1, 2, 4}) { run<1>(w, d, sink); run<2>(w, d, sink); run<4>(w, d, sink); run<8>(w, d, sink); }
  return 0;
}
