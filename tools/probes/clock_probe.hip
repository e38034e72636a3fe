// What shader clock does a one-workgroup (latency-bound) kernel run at, alone and next to a chip-filling kernel?
// s_memtime counts shader-clock cycles, wall_clock64() a constant 100 MHz.   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(long long* out, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);     // dependent chain
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
__global__ void k_burn(float* p, int iters) {
  float a = p[threadIdx.x], b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + a * 1e-9f; }
  p[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
int main() {
  long long *d, h[3];
  float* burn;
  hipMalloc(&d, 24); hipMalloc(&burn, 4 * 1024 * 256 * 8);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s1, d, 2000000);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("alone:      %lld shader cycles in %.3f ms -> %.0f MHz, %.2f cycles per dependent f64 FMA\n", h[0], h[1] / 1e5, h[0] / (h[1] / 100.0), h[0] / 2e6);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_burn, dim3(2048), dim3(256), 0, s2, burn, 4000000);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s1, d, 2000000);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("with burn:  %lld shader cycles in %.3f ms -> %.0f MHz\n", h[0], h[1] / 1e5, h[0] / (h[1] / 100.0));
    hipDeviceSynchronize();
  }
  return 0;
}
