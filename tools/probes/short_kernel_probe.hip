// Do 25-us one-workgroup kernels (the Cholesky chain's shape) always run at full clock?  Each launch runs 2000 dependent
// v_fma_f64 (64 000 cycles) and records s_memtime / wall_clock64 deltas: the histogram of per-launch MHz and duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <unistd.h>
__global__ void k_probe(long long* out, int iters, int slot) {
  double x = 1.0 + threadIdx.x * 1e-9;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[3 * slot] = c1 - c0; out[3 * slot + 1] = w1 - w0; out[3 * slot + 2] = (long long)x; }
}
__global__ void k_burn(float* p, int iters) {
  float a = p[threadIdx.x], b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + a * 1e-9f; }
  p[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
static void report(const char* tag, const std::vector<long long>& h, int n) {
  std::vector<double> us(n), mhz(n);
  for (int i = 0; i < n; ++i) { us[i] = h[3 * i + 1] / 100.0; mhz[i] = h[3 * i] / us[i]; }
  std::sort(us.begin(), us.end()); std::sort(mhz.begin(), mhz.end());
  printf("%-28s us: min %.1f p50 %.1f p90 %.1f max %.1f | MHz: min %.0f p10 %.0f p50 %.0f\n", tag, us[0], us[n / 2], us[n * 9 / 10], us[n - 1], mhz[0], mhz[n / 10], mhz[n / 2]);
}
int main() {
  const int N = 1000;
  long long* d; float* burn;
  hipMalloc(&d, 24 * N); hipMalloc(&burn, 4 * 1024 * 256 * 8);
  std::vector<long long> h(3 * N);
  hipStream_t s; hipStreamCreate(&s);
  auto run = [&](const char* tag, int gap_us, int burn_every) {
    for (int i = 0; i < N; ++i) {
      if (burn_every && i % burn_every == 0) hipLaunchKernelGGL(k_burn, dim3(2048), dim3(256), 0, s, burn, 200000);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s, d, 2000, i);
      if (gap_us) { hipStreamSynchronize(s); usleep(gap_us); }
    }
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, 24 * N, hipMemcpyDeviceToHost);
    report(tag, h, N);
  };
  run("back to back", 0, 0);
  run("100 us idle between", 100, 0);
  run("2 ms idle between", 2000, 0);
  run("after a burn kernel each 10", 0, 10);
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_burn, dim3(2048), dim3(256), 0, s, burn, 2000000);   // ~10+ s of load
  hipStreamSynchronize(s);
  run("hot: back to back", 0, 0);
  run("hot: 100 us idle between", 100, 0);
  run("hot: after a burn each 10", 0, 10);
  return 0;
}
