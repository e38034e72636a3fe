// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_32x32x2_f32 with NACC independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v16f32 __attribute__((ext_vector_type(16)));

template <int NACC, bool SHARE_A>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double x, long long* clk) {
  const long long c0 = clock64(), w0 = wall_clock64();
  v4f64 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = v4f64{0, 0, 0, 0};
  double a[NACC], b[NACC];
  for (int t = 0; t < NACC; ++t) { a[t] = x + t + threadIdx.x; b[t] = x - t; }
  for (int i = 0; i < iters / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int t = 0; t < NACC; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(SHARE_A ? a[0] : a[t], b[t], acc[t], 0, 0, 0);
  }
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float x) {
  v16f32 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
  float a[NACC], b[NACC];
  for (int t = 0; t < NACC; ++t) { a[t] = x + t + threadIdx.x; b[t] = x - t; }
  for (int i = 0; i < iters / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static long long* clk;
static void timeit_unused();
template <typename F>
static void timeit(const char* name, F launch, double flops_per_mfma, int nacc, int iters, int blocks, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = double(blocks) * threads / 64;
  const double mfmas = waves * double(iters) * nacc;
  long long h[2] = {0, 0};
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-24s it=%-6d blocks=%d: %.3f ms  %.1f TFLOP/s  clock64/MFMA=%.1f  wall100MHz ticks=%lld -> shader MHz~%.0f\n", name, iters, blocks, ms,
         mfmas * flops_per_mfma / ms / 1e9, double(h[0]) / (double(iters) * nacc), h[1], h[1] ? double(h[0]) / double(h[1]) * 100.0 : 0.0);
}

int main() {
  double* out;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&clk, 64);
  for (int iters : {8000})
  for (int wgs_per_cu : {1, 2, 4}) {
    const int blocks = 256 * wgs_per_cu;
    timeit("f64 nacc=4  distinctA", [&] { hipLaunchKernelGGL((k_f64<4, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 4, iters, blocks, 256);
    timeit("f64 nacc=10 shareA", [&] { hipLaunchKernelGGL((k_f64<10, true>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 10, iters, blocks, 256);
    timeit("f64 nacc=10 distinctA", [&] { hipLaunchKernelGGL((k_f64<10, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 10, iters, blocks, 256);
    timeit("f64 nacc=16 distinctA", [&] { hipLaunchKernelGGL((k_f64<16, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 16, iters, blocks, 256);
    timeit("f64 nacc=1", [&] { hipLaunchKernelGGL((k_f64<1, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 1, iters, blocks, 256);
    timeit("f64 nacc=2", [&] { hipLaunchKernelGGL((k_f64<2, false>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, clk); }, 2048, 2, iters, blocks, 256);
    timeit("f32 32x32x2 nacc=4", [&] { hipLaunchKernelGGL((k_f32<4>), dim3(blocks), dim3(256), 0, 0, (float*)out, iters, 1.0f); }, 4096, 4, iters, blocks, 256);
  }
  return 0;
}
