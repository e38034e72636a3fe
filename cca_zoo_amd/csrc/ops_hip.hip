// HIP (gfx950 / CDNA4) implementation of the device-op interface in ops.h.
//
// Everything here is float64 dense linear algebra on d x d (d <= ~16k) blocks
// that lives AFTER the Gram reduction: it is latency / launch bound, not on the
// MFMA or HBM roofline (DESIGN.md "solver stage").  The GEMM uses the fp64
// matrix pipe (v_mfma_f64_16x16x4_f64); Cholesky / triangular solves are
// blocked around it; the Jacobi rotations run one workgroup per row pair.
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hip_common.h"
#include "jacobi_dev.h"
#include "rng_hash.h"

namespace ccz {

// ===========================================================================
// memory pool + copies
// ===========================================================================
void* dev_alloc(ccz_ctx* c, size_t bytes) {
  Impl* im = impl(c);
  if (bytes == 0) bytes = 8;
  bytes = (bytes + 255) & ~size_t(255);
  int best = -1;
  for (size_t i = 0; i < im->pool.size(); ++i) {
    PoolBlock& b = im->pool[i];
    if (!b.used && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096) {
      if (best < 0 || b.bytes < im->pool[best].bytes) best = int(i);
    }
  }
  if (best >= 0) {
    im->pool[best].used = true;
    return im->pool[best].p;
  }
  void* p = nullptr;
  static const bool trace = getenv("CCZ_TRACE_POOL") != nullptr;
  if (trace) fprintf(stderr, "[ccz] pool miss: hipMalloc(%zu) (%zu blocks cached)\n", bytes, im->pool.size());
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    // release cached blocks and retry once
    (void)hipGetLastError();
    CCZ_HIP(hipStreamSynchronize(stream(c)));
    for (auto it = im->pool.begin(); it != im->pool.end();) {
      if (!it->used) { (void)hipFree(it->p); it = im->pool.erase(it); } else ++it;
    }
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); fail(CCZ_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
  }
  im->pool.push_back({p, bytes, true});
  return p;
}

void dev_free(ccz_ctx* c, void* p) {
  if (!p) return;
  Impl* im = impl(c);
  for (auto& b : im->pool)
    if (b.p == p) { b.used = false; return; }
  // not ours: stream-ordered work may still use it, so synchronise before freeing
  (void)hipStreamSynchronize(stream(c));
  (void)hipFree(p);
}

static void h2d_blocking(ccz_ctx* c, void* dst, const void* src, size_t bytes) {
  CCZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream(c)));
  CCZ_HIP(hipStreamSynchronize(stream(c)));  // src is pageable host memory owned by the caller
}
void h2d(ccz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  // small copies (shifts, Ritz values, permutations: the solver's steering data) ride through the pinned ring and do not
  // make the host wait; the source may be reused as soon as this returns either way
  if (bytes <= Impl::kSmallBytes) { h2d_small(c, dst, src, bytes); return; }
  h2d_blocking(c, dst, src, bytes);
}
void h2d_small(ccz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  Impl* im = impl(c);
  if (bytes > Impl::kSmallBytes) { h2d_blocking(c, dst, src, bytes); return; }
  // next slot of the ring whose previous copy has drained (a deep queue -- many enqueue-only calls behind a long
  // kernel -- must not make the host wait for the device: look for a free slot before blocking on the oldest one)
  int i = im->small_next;
  for (int t = 0; t < Impl::kSmallSlots; ++t) {
    const int j = (im->small_next + t) % Impl::kSmallSlots;
    if (!im->small_pin[j]) { i = j; break; }
    const hipError_t q = hipEventQuery(im->small_ev[j]);
    if (q == hipSuccess) { i = j; break; }
    if (q != hipErrorNotReady) { (void)hipGetLastError(); }
  }
  if (!im->small_pin[i]) {
    if (hipHostMalloc(&im->small_pin[i], Impl::kSmallBytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&im->small_ev[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (im->small_pin[i]) { (void)hipHostFree(im->small_pin[i]); im->small_pin[i] = nullptr; }
      h2d_blocking(c, dst, src, bytes);
      return;
    }
  } else {
    CCZ_HIP(hipEventSynchronize(im->small_ev[i]));      // (returns at once for a drained slot)
  }
  im->small_next = (i + 1) % Impl::kSmallSlots;
  std::memcpy(im->small_pin[i], src, bytes);
  CCZ_HIP(hipMemcpyAsync(dst, im->small_pin[i], bytes, hipMemcpyHostToDevice, stream(c)));
  CCZ_HIP(hipEventRecord(im->small_ev[i], stream(c)));
}
// ---- short host waits ---------------------------------------------------------------------------------------------
// The solve is a chain of small launches with a handful of read-backs (pivot flags, residual norms, Ritz values).  A
// BLOCKING wait (hipStreamSynchronize: the thread sleeps on the completion interrupt) was measured to return 10 - 30 ms
// after the device had finished, in every OTHER fit of a back-to-back loop (DESIGN.md "the period-2 solve"): the device
// time of the phase was unchanged, the host sat in the wait.  Waits that are expected to be short therefore POLL the
// completion event (hipEventQuery) for up to `spin_ms` before they fall back to the blocking call.
// CCZ_SPIN_WAIT_MS=0 restores the blocking waits.
static double spin_budget_ms() {
  static const double v = [] { const char* e = getenv("CCZ_SPIN_WAIT_MS"); return e ? atof(e) : 50.0; }();
  return v;
}
struct WaitStats { double total_ms = 0.0, max_ms = 0.0; long count = 0, fell_back = 0; };
static WaitStats& wait_stats() { static WaitStats w; return w; }

__global__ void k_copy_words(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) dst[i] = src[i];
  __threadfence_system();
}
// the same in 16-byte pieces (both pointers 16-byte aligned, n16 pieces): a store to host-mapped memory leaves the chip as one
// request per lane-group, and 8-byte lanes were seen to take 8 ms per 2 MB in some fits (bench extras: backproject 16 / 25 ms)
typedef unsigned int v4u32_copy __attribute__((ext_vector_type(4)));
__global__ void k_copy_words16(const v4u32_copy* __restrict__ src, v4u32_copy* __restrict__ dst, int64_t n16) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += int64_t(gridDim.x) * blockDim.x) dst[i] = src[i];
  __threadfence_system();
}

static void wait_stream_short(ccz_ctx* c) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  const auto t0 = std::chrono::steady_clock::now();
  const double budget = spin_budget_ms();
  bool done = false;
  if (budget > 0.0) {
    if (!im->wait_ev) CCZ_HIP(hipEventCreateWithFlags(&im->wait_ev, hipEventDisableTiming));
    CCZ_HIP(hipEventRecord(im->wait_ev, st));
    for (;;) {
      const hipError_t q = hipEventQuery(im->wait_ev);
      if (q == hipSuccess) { done = true; break; }
      if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > budget) break;
      __builtin_ia32_pause();
    }
  }
  if (!done) { CCZ_HIP(hipStreamSynchronize(st)); ++wait_stats().fell_back; }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  WaitStats& w = wait_stats();
  w.total_ms += ms;
  w.max_ms = std::max(w.max_ms, ms);
  ++w.count;
}

void d2h(ccz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  Impl* im = impl(c);
  // small read-backs go through a pinned word buffer: the copy is then truly asynchronous (a pageable destination is
  // staged by the runtime, which may block inside the call) and the wait below polls
  if (bytes <= Impl::kD2hPinBytes && spin_budget_ms() > 0.0) {
    if (!im->d2h_pin) {
      if (hipHostMalloc(&im->d2h_pin, Impl::kD2hPinBytes, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        im->d2h_pin = nullptr;
      } else if (hipHostGetDevicePointer(&im->d2h_pin_dev, im->d2h_pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        im->d2h_pin_dev = nullptr;
      }
    }
    // CCZ_D2H_MODE (a measurement switch, tools/d2h_probe.py): 0 copy kernel, 8-byte lanes, up to 2048 workgroups (default) |
    // 1 copy kernel, 16-byte lanes | 2 hipMemcpyAsync into the pinned buffer + polled event | 3 the same + hipStreamSynchronize |
    // 4 one blocking hipMemcpy into the destination
    const char* me = std::getenv("CCZ_D2H_MODE");
    const int mode = me ? std::atoi(me) : 0;
    const bool trace = std::getenv("CCZ_TRACE_D2H") != nullptr;
    hipEvent_t* tev = im->d2h_tev;
    const auto th0 = std::chrono::steady_clock::now();
    if (trace) {
      if (!tev[0]) { CCZ_HIP(hipEventCreate(&tev[0])); CCZ_HIP(hipEventCreate(&tev[1])); }
      CCZ_HIP(hipEventRecord(tev[0], stream(c)));
    }
    auto report = [&](const char* what) {
      if (!trace) return;
      float dev_ms = -1.f;
      (void)hipEventSynchronize(tev[1]);
      (void)hipEventElapsedTime(&dev_ms, tev[0], tev[1]);
      std::fprintf(stderr, "[ccz] d2h %s %zu bytes: device %.3f ms, host %.3f ms\n", what, bytes, dev_ms,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count());
    };
    if (mode == 4) {
      CCZ_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
      if (trace) { CCZ_HIP(hipEventRecord(tev[1], stream(c))); report("blocking"); }
      return;
    }
    if (im->d2h_pin) {
      // a KERNEL writes the pinned (host-mapped) buffer: no SDMA engine, hence no cross-engine dependency for the
      // runtime's helper thread to resolve -- with hipMemcpyAsync the event behind the copy was observed to complete
      // 15 - 19 ms late on the DEVICE time line in some fits (three read-backs of the back-projection phase)
      if (mode <= 1 && im->d2h_pin_dev && bytes % 8 == 0 && reinterpret_cast<uintptr_t>(src) % 8 == 0) {
        const int64_t words = int64_t(bytes / 8);
        if (mode == 1 && bytes % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0) {
          hipLaunchKernelGGL(k_copy_words16, dim3((unsigned)std::min<int64_t>((words / 2 + 255) / 256, 2048)), dim3(256), 0, stream(c),
                             static_cast<const v4u32_copy*>(src), static_cast<v4u32_copy*>(im->d2h_pin_dev), words / 2);
        } else {
          hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<int64_t>((words + 255) / 256, 2048)), dim3(256), 0, stream(c),
                             static_cast<const unsigned long long*>(src), static_cast<unsigned long long*>(im->d2h_pin_dev), words);
        }
        CCZ_LAUNCH_CHECK();
      } else {
        CCZ_HIP(hipMemcpyAsync(im->d2h_pin, src, bytes, hipMemcpyDeviceToHost, stream(c)));
      }
      if (trace) CCZ_HIP(hipEventRecord(tev[1], stream(c)));
      if (mode == 3) CCZ_HIP(hipStreamSynchronize(stream(c)));
      else wait_stream_short(c);
      std::memcpy(dst, im->d2h_pin, bytes);
      report(mode <= 1 ? "kernel" : "sdma");
      return;
    }
  }
  CCZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream(c)));
  CCZ_HIP(hipStreamSynchronize(stream(c)));
}
void d2d(ccz_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes || dst == src) return;
  CCZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream(c)));
}
void zero(ccz_ctx* c, void* dst, size_t bytes) {
  if (!bytes) return;
  CCZ_HIP(hipMemsetAsync(dst, 0, bytes, stream(c)));
}
void sync(ccz_ctx* c) { CCZ_HIP(hipStreamSynchronize(stream(c))); }
void sync_short(ccz_ctx* c) { wait_stream_short(c); }
void wait_deferred(ccz_ctx* c) {
  Impl* im = impl(c);
  if (!im->deferred_event) return;
  hipEvent_t ev = static_cast<hipEvent_t>(im->deferred_event);
  im->deferred_event = nullptr;
  if (hipStreamWaitEvent(stream(c), ev, 0) != hipSuccess) {          // (an event destroyed meanwhile has nothing pending)
    (void)hipGetLastError();
    CCZ_HIP(hipDeviceSynchronize());
  }
}

// ---- phase tracing without synchronisation (ops.h: trace_mark / trace_flush) ----
__global__ void k_clock_probe(long long* out) {
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  while (wall_clock64() - w0 < 2000) {}                     // 20 us of the constant 100 MHz counter
  if (threadIdx.x == 0) { out[0] = __builtin_readcyclecounter() - c0; out[1] = wall_clock64() - w0; }
}
namespace {
struct TraceMark { std::string name; hipEvent_t ev; double host_ms; };
struct TraceState {
  std::vector<TraceMark> marks;
  std::vector<hipEvent_t> spare;
  long long* probes = nullptr;      // 2 x 64 counters, device
  std::chrono::steady_clock::time_point t0;
};
TraceState& trace_state() { static TraceState s; return s; }
}  // namespace
void trace_mark(ccz_ctx* c, const char* name) {
  TraceState& ts = trace_state();
  if (ts.marks.size() >= 64) return;
  if (!ts.probes) CCZ_HIP(hipMalloc(reinterpret_cast<void**>(&ts.probes), 64 * 2 * sizeof(long long)));
  if (ts.marks.empty()) ts.t0 = std::chrono::steady_clock::now();
  hipEvent_t ev;
  if (!ts.spare.empty()) { ev = ts.spare.back(); ts.spare.pop_back(); }
  else CCZ_HIP(hipEventCreate(&ev));
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, stream(c), ts.probes + 2 * ts.marks.size());
  CCZ_HIP(hipEventRecord(ev, stream(c)));
  const double hm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts.t0).count();
  ts.marks.push_back({name, ev, hm});
}
void trace_flush(ccz_ctx* c, const char* what) {
  TraceState& ts = trace_state();
  if (ts.marks.empty()) return;
  CCZ_HIP(hipStreamSynchronize(stream(c)));
  std::vector<long long> pr(ts.marks.size() * 2);
  CCZ_HIP(hipMemcpy(pr.data(), ts.probes, pr.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::string line;
  char buf[160];
  for (size_t i = 0; i < ts.marks.size(); ++i) {
    float dev_ms = 0.f;
    if (i > 0) (void)hipEventElapsedTime(&dev_ms, ts.marks[i - 1].ev, ts.marks[i].ev);
    const double mhz = pr[2 * i + 1] > 0 ? double(pr[2 * i]) / (double(pr[2 * i + 1]) / 100.0) : 0.0;
    snprintf(buf, sizeof(buf), " | %s dev %.2f host %.2f clk %.0f", ts.marks[i].name.c_str(), dev_ms,
             ts.marks[i].host_ms - (i > 0 ? ts.marks[i - 1].host_ms : 0.0), mhz);
    line += buf;
  }
  WaitStats& w = wait_stats();
  snprintf(buf, sizeof(buf), " || host waits: %ld, total %.2f ms, longest %.2f ms, %ld blocking", w.count, w.total_ms, w.max_ms, w.fell_back);
  line += buf;
  w = WaitStats{};
  fprintf(stderr, "[ccz] %s phases (ms; each phase includes a 20 us probe)%s\n", what, line.c_str());
  for (auto& m : ts.marks) ts.spare.push_back(m.ev);
  ts.marks.clear();
}
void wave_kernels_init();
void activate(ccz_ctx* c) {
  CCZ_HIP(hipSetDevice(c->device));
  wave_kernels_init();
}
int device_current() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return dev;
}
void device_set(int dev) { (void)hipSetDevice(dev); }

// ===========================================================================
// GEMM on the matrix pipe: 64x64 block tile, 4 waves (2x2), each wave 2x2 MFMA 16x16x4
// ===========================================================================
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
  typedef v4f64 acc_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma16<float> {
  typedef v4f32 acc_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int GB = 64;   // block tile edge
constexpr int GK = 16;   // k-step
constexpr int GP = 4;    // LDS row padding (elements)

// C = alpha * (op(A) op(B) - bias) + beta * C ;  A, C are T ; B is TB (converted on load)
// Extras: C2 = optional second destination; lower_only skips tiles strictly above the diagonal;
// gridDim.z > 1 = split-K (each z-slice accumulates its K range into C with fp64/fp32 atomics; the
// launcher has already applied beta to C).
template <typename T, typename TBs, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(int64_t M, int64_t N, int64_t K, T alpha,
                                                   const T* __restrict__ A, int64_t lda,
                                                   const TBs* __restrict__ B, int64_t ldb, T beta,
                                                   T* __restrict__ C, int64_t ldc,
                                                   const double* __restrict__ bias, T* __restrict__ C2,
                                                   int64_t ldc2, int lower_only, int64_t k_per_split) {
  __shared__ T As[2][GK][GB + GP];
  __shared__ T Bs[2][GK][GB + GP];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = int64_t(blockIdx.y) * GB, n0 = int64_t(blockIdx.x) * GB;
  if (lower_only && n0 > m0 + (GB - 1)) return;
  const int64_t kz0 = int64_t(blockIdx.z) * k_per_split;
  const int64_t kz1 = min(K, kz0 + k_per_split);
  const bool split = gridDim.z > 1;

  typename Mfma16<T>::acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = T(0);

  T ra[4], rb[4];
  auto load_regs = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (TA) { m = tid & 63; k = (tid >> 6) + 4 * i; } else { k = tid & 15; m = (tid >> 4) + 16 * i; }
      const int64_t gm = m0 + m, gk = k0 + k;
      T v = T(0);
      if (gm < M && gk < kz1) v = TA ? A[gk * lda + gm] : A[gm * lda + gk];
      ra[i] = v;
      int n, kb;
      if (TB) { kb = tid & 15; n = (tid >> 4) + 16 * i; } else { n = tid & 63; kb = (tid >> 6) + 4 * i; }
      const int64_t gn = n0 + n, gkb = k0 + kb;
      T w = T(0);
      if (gn < N && gkb < kz1) w = T(TB ? B[gn * ldb + gkb] : B[gkb * ldb + gn]);
      rb[i] = w;
    }
  };
  auto store_regs = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (TA) { m = tid & 63; k = (tid >> 6) + 4 * i; } else { k = tid & 15; m = (tid >> 4) + 16 * i; }
      As[buf][k][m] = ra[i];
      int n, kb;
      if (TB) { kb = tid & 15; n = (tid >> 4) + 16 * i; } else { n = tid & 63; kb = (tid >> 6) + 4 * i; }
      Bs[buf][kb][n] = rb[i];
    }
  };

  const int64_t nk = (kz1 - kz0 + GK - 1) / GK;
  if (nk <= 0) return;
  load_regs(kz0);
  store_regs(0);
  __syncthreads();
  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = int(kt & 1);
    if (kt + 1 < nk) load_regs(kz0 + (kt + 1) * GK);
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kr = 4 * kk + (lane >> 4);
      T a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[cur][kr][wr * 32 + t * 16 + (lane & 15)];
        b[t] = Bs[cur][kr][wc * 32 + t * 16 + (lane & 15)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mfma16<T>::run(a[i], b[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_regs(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gm = m0 + wr * 32 + i * 16 + Mfma16<T>::row(lane, r);
        const int64_t gn = n0 + wc * 32 + j * 16 + (lane & 15);
        if (gm < M && gn < N) {
          T v = acc[i][j][r];
          if (bias && blockIdx.z == 0) v -= T(bias[gn]);
          v *= alpha;
          if (split) {
            unsafeAtomicAdd(&C[gm * ldc + gn], v);
          } else {
            if (beta != T(0)) v += beta * C[gm * ldc + gn];
            C[gm * ldc + gn] = v;
            if (C2) C2[gm * ldc2 + gn] = v;
          }
        }
      }
}

template <typename T>
__global__ void k_scale2d(int64_t total, int64_t cols, T* __restrict__ A, int64_t lda, T beta) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    A[r * lda + cc] = beta == T(0) ? T(0) : beta * A[r * lda + cc];
  }
}

template <typename T, typename TBs>
static void gemm_launch(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, T alpha, const T* A,
                        int64_t lda, const TBs* B, int64_t ldb, T beta, T* C, int64_t ldc, const double* bias,
                        T* C2 = nullptr, int64_t ldc2 = 0, bool lower_only = false) {
  if (M <= 0 || N <= 0) return;
  const int64_t tm = (M + GB - 1) / GB, tn = (N + GB - 1) / GB;
  if (tm > 65535) fail(CCZ_EUNSUP, "gemm: M=%lld too large for one launch", (long long)M);
  hipStream_t st = stream(c);
  // split-K when the output has too few tiles to fill the chip and K is deep (skinny products of the
  // subspace iteration: 4096 x 80 outputs over K = 4096)
  int splits = 1;
  const int ncu = std::max(1, impl(c)->props.multiProcessorCount);
  if (!C2 && !lower_only && tm * tn * 2 <= ncu && K >= 1024) {
    splits = int(std::min<int64_t>({int64_t(32), (2 * ncu) / (tm * tn), K / 256}));
    if (splits < 2) splits = 1;
  }
  int64_t kps = K;
  if (splits > 1) {
    kps = ((K + splits - 1) / splits + GK - 1) / GK * GK;
    splits = int((K + kps - 1) / kps);
    const int64_t total = M * N;
    hipLaunchKernelGGL(k_scale2d<T>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), dim3(256), 0, st,
                       total, N, C, ldc, beta);
  }
  dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)splits);
  const int lo = lower_only ? 1 : 0;
  if (!tA && !tB) hipLaunchKernelGGL((gemm_kernel<T, TBs, false, false>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, C2, ldc2, lo, kps);
  else if (tA && !tB) hipLaunchKernelGGL((gemm_kernel<T, TBs, true, false>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, C2, ldc2, lo, kps);
  else if (!tA && tB) hipLaunchKernelGGL((gemm_kernel<T, TBs, false, true>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, C2, ldc2, lo, kps);
  else hipLaunchKernelGGL((gemm_kernel<T, TBs, true, true>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, C2, ldc2, lo, kps);
  CCZ_LAUNCH_CHECK();
}

void gemm(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
          int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  if (gemm_f64_skinny_eligible(tA, tB, M, N, K, A, lda, B, ldb)) {
    gemm_f64_skinny(c, tA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    return;
  }
  if (gemm_f64_big_eligible(tA, tB, M, N, K, A, lda, B, ldb, C, ldc)) {
    gemm_f64_big(c, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, false);
    return;
  }
  // split very tall outputs so that grid.y stays legal
  const int64_t maxM = int64_t(65535) * GB;
  for (int64_t m = 0; m < M; m += maxM) {
    const int64_t mm = std::min(maxM, M - m);
    const double* Ap = tA ? A + m : A + m * lda;
    gemm_launch<double, double>(c, tA, tB, mm, N, K, alpha, Ap, lda, B, ldb, beta, C + m * ldc, ldc, nullptr);
  }
}

void gemm_ex(ccz_ctx* c, bool tA, bool tB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* C2,
             int64_t ldc2, bool lower_only) {
  if (!C2 && !lower_only && gemm_f64_skinny_eligible(tA, tB, M, N, K, A, lda, B, ldb)) {
    gemm_f64_skinny(c, tA, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
    return;
  }
  if (!C2 && gemm_f64_big_eligible(tA, tB, M, N, K, A, lda, B, ldb, C, ldc)) {
    gemm_f64_big(c, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower_only);
    return;
  }
  const int64_t maxM = int64_t(65535) * GB;
  if (M > maxM) fail(CCZ_EUNSUP, "gemm_ex: M too large");
  gemm_launch<double, double>(c, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, C2, ldc2, lower_only);
}

void gemm_mixed(ccz_ctx* c, int dtype, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda,
                const double* B, int64_t ldb, double beta, void* C, int64_t ldc, const double* bias_row) {
  if (dtype == CCZ_F32 && gemm_f32_big_eligible(M, N, K, lda, ldb, ldc, A, C)) {
    gemm_f32_big(c, M, N, K, alpha, static_cast<const float*>(A), lda, B, ldb, beta, static_cast<float*>(C), ldc, bias_row);
    return;
  }
  const int64_t maxM = int64_t(65535) * GB;
  for (int64_t m = 0; m < M; m += maxM) {
    const int64_t mm = std::min(maxM, M - m);
    if (dtype == CCZ_F32)
      gemm_launch<float, double>(c, false, false, mm, N, K, float(alpha), static_cast<const float*>(A) + m * lda, lda,
                                 B, ldb, float(beta), static_cast<float*>(C) + m * ldc, ldc, bias_row);
    else
      gemm_launch<double, double>(c, false, false, mm, N, K, alpha, static_cast<const double*>(A) + m * lda, lda, B,
                                  ldb, beta, static_cast<double*>(C) + m * ldc, ldc, bias_row);
  }
}

// ===========================================================================
// elementwise / layout kernels
// ===========================================================================
__global__ void k_transpose(int64_t rows, int64_t cols, const double* __restrict__ in, int64_t ldi,
                            double* __restrict__ out, int64_t ldo) {
  __shared__ double t[32][33];
  const int64_t r0 = int64_t(blockIdx.y) * 32, c0 = int64_t(blockIdx.x) * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t r = r0 + i, cc = c0 + threadIdx.x;
    if (r < rows && cc < cols) t[i][threadIdx.x] = in[r * ldi + cc];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t orow = c0 + i, ocol = r0 + threadIdx.x;   // out is cols x rows
    if (orow < cols && ocol < rows) out[orow * ldo + ocol] = t[threadIdx.x][i];
  }
}

void transpose(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0) return;
  const int64_t by = (rows + 31) / 32;
  for (int64_t y0 = 0; y0 < by; y0 += 65535) {   // grid.y limit
    const int64_t ny = std::min<int64_t>(65535, by - y0);
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)ny);
    hipLaunchKernelGGL(k_transpose, grid, dim3(32, 8), 0, stream(c), rows - y0 * 32, cols, in + y0 * 32 * ldi, ldi,
                       out + y0 * 32, ldo);
  }
  CCZ_LAUNCH_CHECK();
}

#define CCZ_GRID2D(rows, cols) \
  dim3 grid((unsigned)std::min<int64_t>(((rows) * (cols) + 255) / 256, 1 << 20)); \
  const int64_t total = (rows) * (cols)

__global__ void k_copy2d(int64_t total, int64_t cols, const double* __restrict__ in, int64_t ldi,
                         double* __restrict__ out, int64_t ldo) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    out[r * ldo + cc] = in[r * ldi + cc];
  }
}
void copy2d(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi, double* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0 || (in == out && ldi == ldo)) return;
  if (in == out) fail(CCZ_EINVAL, "copy2d: in-place with different strides");
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_copy2d, grid, dim3(256), 0, stream(c), total, cols, in, ldi, out, ldo);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_axpby2d(int64_t total, int64_t cols, double alpha, double* __restrict__ A, int64_t lda, double beta,
                          const double* __restrict__ B, int64_t ldb) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    double v = alpha * A[r * lda + cc];
    if (B) v += beta * B[r * ldb + cc];
    A[r * lda + cc] = v;
  }
}
void axpby2d(ccz_ctx* c, int64_t rows, int64_t cols, double alpha, double* A, int64_t lda, double beta, const double* B,
             int64_t ldb) {
  if (rows <= 0 || cols <= 0) return;
  if (beta == 0.0) B = nullptr;
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_axpby2d, grid, dim3(256), 0, stream(c), total, cols, alpha, A, lda, beta, B, ldb);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_fill2d(int64_t total, int64_t cols, double* __restrict__ A, int64_t lda, double v) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    A[r * lda + cc] = v;
  }
}
void fill2d(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, double v) {
  if (rows <= 0 || cols <= 0) return;
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_fill2d, grid, dim3(256), 0, stream(c), total, cols, A, lda, v);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_add_diag(int64_t d, double* __restrict__ A, int64_t lda, double v) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < d) A[i * lda + i] += v;
}
void add_diag(ccz_ctx* c, int64_t d, double* A, int64_t lda, double v) {
  if (d <= 0) return;
  hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream(c), d, A, lda, v);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_scale_cols(int64_t total, int64_t cols, double* __restrict__ A, int64_t lda,
                             const double* __restrict__ v, int mode) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    const double f = mode == 0 ? v[cc] : (mode == 1 ? 1.0 / v[cc] : 1.0 / sqrt(v[cc]));
    A[r * lda + cc] *= f;
  }
}
void scale_cols(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, const double* v, int mode) {
  if (rows <= 0 || cols <= 0) return;
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_scale_cols, grid, dim3(256), 0, stream(c), total, cols, A, lda, v, mode);
  CCZ_LAUNCH_CHECK();
}

// lower tile (bi > bj) <- transpose of upper tile (bj, bi) through LDS; diagonal tiles in place
__global__ void k_mirror_upper(int64_t d, double* __restrict__ A, int64_t lda) {
  __shared__ double t[32][33];
  const int64_t bi = blockIdx.y, bj = blockIdx.x;
  if (bi < bj) return;
  const int64_t r0 = bj * 32, c0 = bi * 32;   // source tile: rows of block bj, cols of block bi (upper)
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t r = r0 + i, cc = c0 + threadIdx.x;
    if (r < d && cc < d) t[i][threadIdx.x] = A[r * lda + cc];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t r = c0 + i, cc = r0 + threadIdx.x;   // destination (lower): A[r][cc] = A[cc][r]
    if (r < d && cc < d && r > cc) A[r * lda + cc] = t[threadIdx.x][i];
  }
}
void mirror_upper(ccz_ctx* c, int64_t d, double* A, int64_t lda) {
  if (d <= 0) return;
  const unsigned nb = (unsigned)((d + 31) / 32);
  hipLaunchKernelGGL(k_mirror_upper, dim3(nb, nb), dim3(32, 8), 0, stream(c), d, A, lda);
  CCZ_LAUNCH_CHECK();
}

template <bool PACK>
__global__ void k_pack_upper(int64_t d, double* __restrict__ A, int64_t lda, double* __restrict__ packed) {
  const int64_t i = blockIdx.y;
  const int64_t j = i + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= d) return;
  const int64_t off = i * d - (i * (i - 1)) / 2 + (j - i);
  if (PACK) packed[off] = A[i * lda + j]; else A[i * lda + j] = packed[off];
}
void pack_upper(ccz_ctx* c, int64_t d, const double* A, int64_t lda, double* packed) {
  if (d <= 0) return;
  if (d > 65535) fail(CCZ_EUNSUP, "pack_upper: d too large");
  hipLaunchKernelGGL(k_pack_upper<true>, dim3((unsigned)((d + 255) / 256), (unsigned)d), dim3(256), 0, stream(c), d,
                     const_cast<double*>(A), lda, packed);
  CCZ_LAUNCH_CHECK();
}
void unpack_upper(ccz_ctx* c, int64_t d, const double* packed, double* A, int64_t lda) {
  if (d <= 0) return;
  if (d > 65535) fail(CCZ_EUNSUP, "unpack_upper: d too large");
  hipLaunchKernelGGL(k_pack_upper<false>, dim3((unsigned)((d + 255) / 256), (unsigned)d), dim3(256), 0, stream(c), d, A,
                     lda, const_cast<double*>(packed));
  CCZ_LAUNCH_CHECK();
}

__global__ void k_cov_block(int64_t total, int64_t cols, const double* __restrict__ G, int64_t D,
                            const double* __restrict__ s, double inv_n, int centre, double alpha, int64_t r0,
                            int64_t c0, double* __restrict__ out, int64_t ldo) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    // K1 fills upper-triangular tiles only: read every element from the upper triangle
    const int64_t gr = r0 + r, gc = c0 + cc;
    double v = gr <= gc ? G[gr * D + gc] : G[gc * D + gr];
    if (centre) v -= s[gr] * s[gc] * inv_n;
    out[r * ldo + cc] = alpha * v;
  }
}
void cov_block(ccz_ctx* c, const double* G, int64_t D, const double* s, int64_t n, bool centre, double alpha,
               int64_t r0, int64_t rows, int64_t c0, int64_t cols, double* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0) return;
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_cov_block, grid, dim3(256), 0, stream(c), total, cols, G, D, s, 1.0 / double(n), centre ? 1 : 0,
                     alpha, r0, c0, out, ldo);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_randn(int64_t total, int64_t cols, double* __restrict__ A, int64_t lda, uint64_t seed) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    A[r * lda + cc] = hash_normal(seed, uint64_t(i));
  }
}
void randn_fill(ccz_ctx* c, int64_t rows, int64_t cols, double* A, int64_t lda, uint64_t seed) {
  if (rows <= 0 || cols <= 0) return;
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_randn, grid, dim3(256), 0, stream(c), total, cols, A, lda, seed);
  CCZ_LAUNCH_CHECK();
}

// ===========================================================================
// reductions
// ===========================================================================
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum; result valid in every thread.  red: >= (blockDim/64) doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__global__ void k_col_sqnorms(int64_t rows, int64_t cols, const double* __restrict__ A, int64_t lda,
                              double* __restrict__ out, int64_t rows_per_block) {
  const int64_t r0 = int64_t(blockIdx.x) * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) {
    double acc = 0.0;
    for (int64_t r = r0; r < r1; ++r) { const double v = A[r * lda + j]; acc += v * v; }
    unsafeAtomicAdd(&out[j], acc);
  }
}
void col_sqnorms(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda, double* out) {
  if (cols <= 0) return;
  zero(c, out, size_t(cols) * 8);
  if (rows <= 0) return;
  const int64_t rpb = 64;
  hipLaunchKernelGGL(k_col_sqnorms, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, stream(c), rows, cols, A,
                     lda, out, rpb);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_row_abs_sums(int64_t cols, const double* __restrict__ A, int64_t lda, double* __restrict__ out) {
  __shared__ double red[8];
  const double* a = A + int64_t(blockIdx.x) * lda;
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) acc += fabs(a[j]);
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
double norm_inf(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda) {
  if (rows <= 0 || cols <= 0) return 0.0;
  DBuf tmp(c, rows);
  hipLaunchKernelGGL(k_row_abs_sums, dim3((unsigned)rows), dim3(256), 0, stream(c), cols, A, lda, tmp.get());
  CCZ_LAUNCH_CHECK();
  std::vector<double> h(rows);
  d2h(c, h.data(), tmp, size_t(rows) * 8);
  double best = 0.0;
  for (double v : h)
    if (!(v <= best)) best = v;  // NaN propagates
  return best;
}

__global__ void k_row_dots(int64_t cols, const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                           int64_t ldb, double* __restrict__ out) {
  __shared__ double red[8];
  const double* a = A + int64_t(blockIdx.x) * lda;
  const double* b = B + int64_t(blockIdx.x) * ldb;
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) acc += a[j] * b[j];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
void row_dots(ccz_ctx* c, int64_t rows, int64_t cols, const double* A, int64_t lda, const double* B, int64_t ldb,
              double* out) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_row_dots, dim3((unsigned)rows), dim3(256), 0, stream(c), cols, A, lda, B, ldb, out);
  CCZ_LAUNCH_CHECK();
}

__global__ void k_gather_rows(int64_t total, int64_t cols, const double* __restrict__ in, int64_t ldi,
                              const int64_t* __restrict__ perm, const double* __restrict__ scale,
                              double* __restrict__ out, int64_t ldo) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = i / cols, cc = i - r * cols;
    out[r * ldo + cc] = in[perm[r] * ldi + cc] * (scale ? scale[r] : 1.0);
  }
}
void gather_rows(ccz_ctx* c, int64_t rows, int64_t cols, const double* in, int64_t ldi, const int64_t* perm_host,
                 const double* scale_host, double* out, int64_t ldo) {
  if (rows <= 0 || cols <= 0) return;
  DBuf pd(c, rows), sd(c, scale_host ? rows : 0);
  h2d(c, pd.get(), perm_host, size_t(rows) * 8);
  if (scale_host) h2d(c, sd.get(), scale_host, size_t(rows) * 8);
  CCZ_GRID2D(rows, cols);
  hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, stream(c), total, cols, in, ldi,
                     reinterpret_cast<const int64_t*>(pd.get()), scale_host ? sd.get() : nullptr, out, ldo);
  CCZ_LAUNCH_CHECK();
}

// ===========================================================================
// Cholesky (lower, blocked, right-looking) and right triangular solves
// ===========================================================================
constexpr int NB = 64;

// ---------------------------------------------------------------------------
// 64 x 64 diagonal block on ONE wavefront, no barriers.  Lane i owns row i of the current Schur
// complement in 64 registers and the block is consumed as a SHIFT REGISTER: at step j register a[0]
// is column j, the rank-1 update writes its result one register down (a[k-1] = a[k] - l_i l_{j+k}),
// so every register index is a compile-time constant while j is a run-time loop counter
// (v_readlane with an SGPR lane index broadcasts l_{j+k}).  The loop body is ~200 instructions;
// four bodies of decreasing width (63, 47, 31, 15 live columns) keep ~2/3 of the triangular saving.
// The first version unrolled all 2016 (j, k) pairs with static register indices: 134 KB of
// straight-line code for one wavefront -- larger than the instruction cache, so its run time was
// set by where the code happened to be cached (46 us ... 210 us for the same block; rocprofv3).
// The finished column goes to LDS (Ls[j][i], stride 65); the second phase forms invT = L^-T by
// forward substitution in the same shift-register form (lane c solves L x = e_c, the multipliers
// L[t+k][t] are wave-uniform LDS broadcasts).  Global loads / stores are whole 512-byte rows,
// transposed through LDS.  Blocks narrower than 64 are padded with the identity.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

constexpr int WLD = NB + 1;   // LDS tile stride (doubles): column and row accesses both conflict-free

template <int KMAX>
__device__ __forceinline__ void chol_steps(double (&a)[NB], int j_begin, int j_end, int lane, double* Ls, int& first_bad) {
#pragma unroll 1
  for (int j = j_begin; j < j_end; ++j) {
    const double col = a[0];
    double piv = lane_bcast(col, j);
    const bool bad = !(piv > 0.0);                 // wave-uniform
    first_bad = bad ? min(first_bad, j) : first_bad;
    piv = bad ? 1.0 : piv;
    const double l = col * (1.0 / sqrt(piv));      // lanes >= j: L[lane][j]; lanes < j hold junk that nobody reads
    double* lcol = Ls + j * WLD;
    lcol[lane] = lane >= j ? l : 0.0;
    // l_{j+k} comes back as a wave-uniform LDS broadcast (one ds_read_b64 with an immediate offset instead of
    // two v_readlane + an SALU add); reads past row 63 land in junk that only feeds junk registers
    const double* lrow = lcol + j;
#pragma unroll
    for (int k = 1; k <= KMAX; ++k) {
      a[k - 1] = a[k] - l * lrow[k];
      if ((k & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int KMAX>
__device__ __forceinline__ void inv_steps(double (&v)[NB], int t_begin, int t_end, int lane, const double* Ls, double* Xs) {
#pragma unroll 1
  for (int t = t_begin; t < t_end; ++t) {
    const double* lcol = Ls + t * WLD + t;         // L[t][t], L[t+1][t], ... (wave-uniform addresses)
    const double x = v[0] / lcol[0];
    Xs[lane * WLD + t] = x;                        // (L^-1)[t][lane] = (L^-T)[lane][t]
#pragma unroll
    for (int k = 1; k <= KMAX; ++k) {
      v[k - 1] = v[k] - lcol[k] * x;               // rows past 63: junk into registers that are never read
      if ((k & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Ajj: the diagonal block (row stride lda), nb valid rows/columns.  do_chol: factor it in place (lower
// triangle written back) else it already holds L.  invT (may be null): 64 x 64 row-major L^-T.
// Returns the first non-positive pivot (0-based) or 0x7fffffff.
__device__ __forceinline__ int wave_block(double* __restrict__ Ajj, int64_t lda, int nb, bool do_chol,
                                          double* __restrict__ invT, double* lds) {
  const int lane = threadIdx.x;
  double* Ls = lds;                 // Ls[t * WLD + i] = L[i][t]
  double* Xs = lds + NB * WLD;      // input tile first, then Xs[c * WLD + t] = (L^-T)[c][t]
  int first_bad = 0x7fffffff;
  if (do_chol) {
    // rows arrive as full 512-byte lines (lane = column), each lane then picks up its own row
    for (int r = 0; r < NB; ++r) {
      double v = (r == lane) ? 1.0 : 0.0;
      if (r < nb && lane < nb) v = Ajj[int64_t(r) * lda + lane];
      Xs[r * WLD + lane] = v;
    }
    double a[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) a[k] = Xs[lane * WLD + k];
    chol_steps<63>(a, 0, 16, lane, Ls, first_bad);
    chol_steps<47>(a, 16, 32, lane, Ls, first_bad);
    chol_steps<31>(a, 32, 48, lane, Ls, first_bad);
    chol_steps<15>(a, 48, 64, lane, Ls, first_bad);
    for (int r = 0; r < nb; ++r)
      if (lane <= r && lane < nb) Ajj[int64_t(r) * lda + lane] = Ls[lane * WLD + r];
  } else {
    for (int r = 0; r < NB; ++r) {
      double v = (r == lane) ? 1.0 : 0.0;
      if (r < nb && lane < nb) v = lane <= r ? Ajj[int64_t(r) * lda + lane] : 0.0;
      Ls[lane * WLD + r] = v;
    }
  }
  if (invT) {
    double v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = (k == lane) ? 1.0 : 0.0;
    inv_steps<63>(v, 0, 16, lane, Ls, Xs);
    inv_steps<47>(v, 16, 32, lane, Ls, Xs);
    inv_steps<31>(v, 32, 48, lane, Ls, Xs);
    inv_steps<15>(v, 48, 64, lane, Ls, Xs);
    for (int r = 0; r < NB; ++r) invT[r * NB + lane] = Xs[r * WLD + lane];
  }
  return first_bad;
}

constexpr size_t WAVE_BLOCK_LDS = (size_t(2) * NB * WLD + 2 * NB) * sizeof(double);   // two tiles + slack for the over-reads

template <bool DO_CHOL>
__global__ __launch_bounds__(64) void k_wave_chol_inv(double* __restrict__ A, int64_t lda, int64_t d, int64_t j_first,
                                                      int* __restrict__ info, double* __restrict__ invT) {
  extern __shared__ __attribute__((aligned(16))) char wave_smem[];
  const int64_t j0 = j_first + int64_t(blockIdx.x) * NB;
  const int nb = int(min(int64_t(NB), d - j0));
  const int bad = wave_block(A + j0 * lda + j0, lda, nb, DO_CHOL, invT ? invT + int64_t(blockIdx.x) * NB * NB : nullptr,
                             reinterpret_cast<double*>(wave_smem));
  if (DO_CHOL && bad != 0x7fffffff && threadIdx.x == 0) atomicMin(info, int(j0 + bad + 1));
}

constexpr int MAXB = 8;
struct CholBatch {
  double* A[MAXB];
  double* invT[MAXB];
  int64_t lda[MAXB];
  int64_t d[MAXB];
};

// one panel step of up to MAXB independent factorisations: block b factors the diagonal block at
// column j0 of matrix b (if it has one) and forms its L^-T
__global__ __launch_bounds__(64) void k_wave_chol_inv_batched(CholBatch bt, int64_t j0, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char wave_smem[];
  const int b = blockIdx.x;
  if (j0 >= bt.d[b]) return;
  const int nb = int(min(int64_t(NB), bt.d[b] - j0));
  const int64_t lda = bt.lda[b];
  const bool last = j0 + nb >= bt.d[b];          // last panel: no solve against it follows
  const int bad = wave_block(bt.A[b] + j0 * lda + j0, lda, nb, true, last ? nullptr : bt.invT[b],
                             reinterpret_cast<double*>(wave_smem));
  if (bad != 0x7fffffff && threadIdx.x == 0) atomicMin(info + b, int(j0 + bad + 1));
}

// the wave kernels use 65 KiB of dynamic LDS: opt in once per device (never inside a graph capture)
void wave_kernels_init() {
  static thread_local int done_for_device = -1;
  int dev = -1;
  CCZ_HIP(hipGetDevice(&dev));
  if (done_for_device == dev) return;
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_chol_inv<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(WAVE_BLOCK_LDS)));
  CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_chol_inv_batched), hipFuncAttributeMaxDynamicSharedMemorySize, int(WAVE_BLOCK_LDS)));
  done_for_device = dev;
}

// invT[b] = L_bb^-T for every diagonal block b of L (one wave each, all blocks in one launch)
static void diag_inverses(ccz_ctx* c, const double* L, int64_t ldl, int64_t d, double* invT) {
  const unsigned nblk = (unsigned)((d + NB - 1) / NB);
  hipLaunchKernelGGL(k_wave_chol_inv<false>, dim3(nblk), dim3(64), WAVE_BLOCK_LDS, stream(c), const_cast<double*>(L), ldl, d,
                     int64_t(0), static_cast<int*>(nullptr), invT);
  CCZ_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------
// Recursive blocked Cholesky / triangular solves.  The recursion halves the block until it reaches
// the 64-column base case (the wave kernel above, which also leaves L_jj^-T in `invT`); every update
// above the base case is a GEMM whose inner dimension is half the current block, so nearly all flops
// run at K >= 128 on the 128x128-tile kernel instead of K = 64 rank updates (memory bound).
// `invT` holds one 64 x 64 block per 64 columns of the matrix.
// ---------------------------------------------------------------------------
static int64_t split_point(int64_t n) {   // multiple of NB closest to n / 2 (at least NB)
  const int64_t blocks = (n + NB - 1) / NB;
  return std::max<int64_t>(1, blocks / 2) * NB;
}

// X (r x n) <- X L^-T (trans) or X L^-1 (!trans); L = lower n x n block whose diagonal 64-blocks have their
// L^-T in invT[0 .. ceil(n/64))
static void trsm_rec(ccz_ctx* c, bool trans, int64_t r, int64_t n, const double* L, int64_t ldl, const double* invT,
                     double* X, int64_t ldx, double* tmp) {
  if (n <= NB) {
    gemm_ex(c, false, !trans, r, n, n, 1.0, X, ldx, invT, NB, 0.0, tmp, NB, X, ldx, false);
    return;
  }
  const int64_t h = split_point(n), t = n - h;
  const double* L21 = L + h * ldl;
  const double* L22 = L + h * ldl + h;
  const double* inv2 = invT + (h / NB) * NB * NB;
  if (trans) {   // [X1 X2] [L11' L21'; 0 L22'] = [B1 B2]
    trsm_rec(c, true, r, h, L, ldl, invT, X, ldx, tmp);
    gemm(c, false, true, r, t, h, -1.0, X, ldx, L21, ldl, 1.0, X + h, ldx);          // B2 -= X1 L21'
    trsm_rec(c, true, r, t, L22, ldl, inv2, X + h, ldx, tmp);
  } else {       // [X1 X2] [L11 0; L21 L22] = [B1 B2]
    trsm_rec(c, false, r, t, L22, ldl, inv2, X + h, ldx, tmp);
    gemm(c, false, false, r, h, t, -1.0, X + h, ldx, L21, ldl, 1.0, X, ldx);          // B1 -= X2 L21
    trsm_rec(c, false, r, h, L, ldl, invT, X, ldx, tmp);
  }
}

struct PotrfJob {
  double* A;
  int64_t lda, d;
  double* invT;   // ceil(d / 64) blocks
  double* tmp;    // d x 64 scratch for the base-case dual-destination products
};

// factor the diagonal range [j0, j0 + n) of every job (all jobs share the recursion shape of the largest)
static void potrf_rec(ccz_ctx* c, std::vector<PotrfJob>& jobs, int64_t j0, int64_t n, int* d_info) {
  if (n <= NB) {
    CholBatch bt;
    const int nb = int(jobs.size());
    for (int i = 0; i < MAXB; ++i) {
      const PotrfJob& jb = jobs[std::min(i, nb - 1)];
      bt.A[i] = jb.A;
      bt.lda[i] = jb.lda;
      bt.d[i] = i < nb ? jb.d : 0;
      bt.invT[i] = jb.invT + (j0 / NB) * NB * NB;
    }
    hipLaunchKernelGGL(k_wave_chol_inv_batched, dim3(nb), dim3(64), WAVE_BLOCK_LDS, stream(c), bt, j0, d_info);
    CCZ_LAUNCH_CHECK();
    return;
  }
  const int64_t h = split_point(n), t = n - h;
  potrf_rec(c, jobs, j0, h, d_info);
  for (PotrfJob& jb : jobs) {
    const int64_t hi = std::min(j0 + n, jb.d);
    if (hi <= j0 + h) continue;
    const int64_t tt = hi - (j0 + h);
    double* A11 = jb.A + j0 * jb.lda + j0;
    double* A21 = jb.A + (j0 + h) * jb.lda + j0;
    double* A22 = jb.A + (j0 + h) * jb.lda + (j0 + h);
    trsm_rec(c, true, tt, h, A11, jb.lda, jb.invT + (j0 / NB) * NB * NB, A21, jb.lda, jb.tmp);      // L21 = A21 L11^-T
    gemm_ex(c, false, true, tt, tt, h, -1.0, A21, jb.lda, A21, jb.lda, 1.0, A22, jb.lda, nullptr, 0, true);  // A22 -= L21 L21'
  }
  (void)t;
  potrf_rec(c, jobs, j0 + h, n - h, d_info);
}

static void potrf_lower_batched_rec(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info) {
  Impl* im = impl(c);
  for (int b0 = 0; b0 < count; b0 += MAXB) {
    const int nbt = std::min(MAXB, count - b0);
    int big[MAXB];
    for (int i = 0; i < MAXB; ++i) big[i] = 0x7fffffff;
    CCZ_HIP(hipMemcpyAsync(im->d_flag + 8, big, sizeof(big), hipMemcpyHostToDevice, stream(c)));
    CCZ_HIP(hipStreamSynchronize(stream(c)));
    std::vector<DBuf> bufs;
    std::vector<PotrfJob> jobs;
    int64_t dmax = 0;
    for (int i = 0; i < nbt; ++i) {
      const int64_t di = d[b0 + i];
      const int64_t nblk = (di + NB - 1) / NB;
      bufs.emplace_back(c, nblk * NB * NB);
      double* inv = bufs.back().get();
      bufs.emplace_back(c, std::max<int64_t>(di, 1) * NB);
      jobs.push_back({A[b0 + i], lda[b0 + i], di, inv, bufs.back().get()});
      dmax = std::max(dmax, di);
    }
    // jobs narrower than the widest simply run out of columns (the kernel / loops skip them)
    potrf_rec(c, jobs, 0, (dmax + NB - 1) / NB * NB, im->d_flag + 8);
    int got[MAXB];
    d2h(c, got, im->d_flag + 8, sizeof(got));
    for (int i = 0; i < nbt; ++i) info[b0 + i] = got[i] == 0x7fffffff ? 0 : got[i];
  }
}

int potrf_lower(ccz_ctx* c, double* A, int64_t d, int64_t lda) {
  int info = 0;
  double* Ap[1] = {A};
  potrf_lower_batched(c, 1, Ap, &d, &lda, &info);
  return info;
}

static void trsm_right_lower_rec(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                                 int64_t ldx) {
  const int64_t nblk = (d + NB - 1) / NB;
  DBuf invT(c, nblk * NB * NB), tmp(c, r * NB);
  diag_inverses(c, L, ldl, d, invT);
  trsm_rec(c, trans, r, d, L, ldl, invT, X, ldx, tmp);
}


// ---------------------------------------------------------------------------
// hipGraph replay of fixed-shape launch sequences.  `fn` must only enqueue work on the handle's stream
// (no allocation, no synchronisation, no host reads); the key must cover every value baked into the
// kernel arguments (shapes AND pointers).
// ---------------------------------------------------------------------------
static inline uint64_t key_mix(uint64_t h, uint64_t v) { return h ^ (v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2)); }
static inline uint64_t key_ptr(uint64_t h, const void* p) { return key_mix(h, reinterpret_cast<uint64_t>(p)); }

template <typename F>
static void graph_run(ccz_ctx* c, uint64_t key, F&& fn) {
  Impl* im = impl(c);
  hipStream_t st = stream(c);
  if (!im->graphs_on || st == nullptr) { fn(); return; }
  for (auto& g : im->graphs)
    if (g.key == key) {
      g.tick = ++im->tick;
      CCZ_HIP(hipGraphLaunch(g.exec, st));
      return;
    }
  static const bool trace = getenv("CCZ_TRACE_POOL") != nullptr;
  if (trace) fprintf(stderr, "[ccz] graph miss: capturing key %016llx (%zu cached)\n", (unsigned long long)key, im->graphs.size());
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    im->graphs_on = 0;
    fn();
    return;
  }
  hipGraph_t graph = nullptr;
  try {
    fn();
  } catch (...) {
    (void)hipStreamEndCapture(st, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    throw;
  }
  CCZ_HIP(hipStreamEndCapture(st, &graph));
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    im->graphs_on = 0;
    fn();       // nothing ran during capture: run it for real
    return;
  }
  (void)hipGraphDestroy(graph);
  if (im->graphs.size() >= 96) {   // evict the least recently used
    size_t lru = 0;
    for (size_t i = 1; i < im->graphs.size(); ++i)
      if (im->graphs[i].tick < im->graphs[lru].tick) lru = i;
    (void)hipGraphExecDestroy(im->graphs[lru].exec);
    im->graphs.erase(im->graphs.begin() + lru);
  }
  im->graphs.push_back({key, exec, ++im->tick});
  CCZ_HIP(hipGraphLaunch(exec, st));
}

// the same for other translation units (evd_block.hip): key helpers + a type-erased entry
uint64_t graph_key_mix(uint64_t h, uint64_t v) { return key_mix(h, v); }
void graph_run_fn(ccz_ctx* c, uint64_t key, const std::function<void()>& fn) { graph_run(c, key, fn); }

// ---- iterative (one 64-column panel at a time) variants: fewer, smaller launches ----
static void potrf_lower_batched_iter(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info) {
  Impl* im = impl(c);
  for (int b0 = 0; b0 < count; b0 += MAXB) {
    const int nbt = std::min(MAXB, count - b0);
    int big[MAXB];
    for (int i = 0; i < MAXB; ++i) big[i] = 0x7fffffff;
    CCZ_HIP(hipMemcpyAsync(im->d_flag + 8, big, sizeof(big), hipMemcpyHostToDevice, stream(c)));
    CCZ_HIP(hipStreamSynchronize(stream(c)));
    CholBatch bt;
    std::vector<DBuf> invT, tmp;
    int64_t dmax = 0;
    for (int i = 0; i < MAXB; ++i) {
      const int b = b0 + std::min(i, nbt - 1);     // pad the descriptor with the last matrix (never launched)
      bt.A[i] = A[b];
      bt.lda[i] = lda[b];
      bt.d[i] = i < nbt ? d[b] : 0;
      bt.invT[i] = nullptr;
    }
    for (int i = 0; i < nbt; ++i) {
      invT.emplace_back(c, NB * NB);
      tmp.emplace_back(c, std::max<int64_t>(d[b0 + i] - NB, 1) * NB);
      bt.invT[i] = invT.back().get();
      dmax = std::max(dmax, d[b0 + i]);
    }
    uint64_t key = key_mix(0x504f545246ull, uint64_t(nbt));
    for (int i = 0; i < nbt; ++i) {
      key = key_ptr(key, A[b0 + i]);
      key = key_mix(key_mix(key, uint64_t(d[b0 + i])), uint64_t(lda[b0 + i]));
      key = key_ptr(key_ptr(key, invT[i].get()), tmp[i].get());
    }
    graph_run(c, key, [&] {
      for (int64_t j = 0; j < dmax; j += NB) {
        hipLaunchKernelGGL(k_wave_chol_inv_batched, dim3(nbt), dim3(64), WAVE_BLOCK_LDS, stream(c), bt, j, im->d_flag + 8);
        CCZ_LAUNCH_CHECK();
        for (int i = 0; i < nbt; ++i) {
          const int64_t di = d[b0 + i], ld = lda[b0 + i];
          if (j >= di) continue;
          const int nb = int(std::min<int64_t>(NB, di - j));
          const int64_t rem = di - j - nb;
          if (rem <= 0) continue;
          double* Ab = A[b0 + i];
          double* A21 = Ab + (j + nb) * ld + j;
          // L21 = A21 L11^-T to the scratch panel and (second destination) back in place: a workgroup reads
          // only its own 64 rows of A21 (all of K) before it writes them
          gemm_ex(c, false, false, rem, nb, nb, 1.0, A21, ld, invT[i], NB, 0.0, tmp[i], NB, A21, ld, false);
          gemm_ex(c, false, true, rem, rem, nb, -1.0, tmp[i], NB, tmp[i], NB, 1.0, Ab + (j + nb) * ld + (j + nb), ld, nullptr, 0, true);
        }
      }
    });
    int got[MAXB];
    d2h(c, got, im->d_flag + 8, sizeof(got));
    for (int i = 0; i < nbt; ++i) info[b0 + i] = got[i] == 0x7fffffff ? 0 : got[i];
  }
}


static void trsm_right_lower_iter(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                      int64_t ldx) {
  if (r <= 0 || d <= 0) return;
  const int64_t nblk = (d + NB - 1) / NB;
  DBuf invT(c, nblk * NB * NB), tmp(c, r * NB);
  uint64_t key = key_mix(0x5452534dull, uint64_t(trans));
  key = key_mix(key_mix(key, uint64_t(r)), uint64_t(d));
  key = key_mix(key_mix(key_ptr(key_ptr(key, L), X), uint64_t(ldl)), uint64_t(ldx));
  key = key_ptr(key_ptr(key, invT.get()), tmp.get());
  // A graph is keyed on every pointer baked into it; capture + instantiation cost ~10 ms.  The few-block solves of the
  // subspace iteration (d = 80: two blocks, five launches) meet new pointer combinations whenever the pool hands out
  // other blocks -- the second fit of a new shape paid 20 ms for graphs that save nothing.  Replay only pays for long chains.
  auto body = [&] {
  diag_inverses(c, L, ldl, d, invT);
  if (trans) {
    // X L' = B, forward over column blocks:  X_j = (B_j - sum_{t<j} X_t L_jt') L_jj^-T
    for (int64_t bj = 0; bj < nblk; ++bj) {
      const int64_t j = bj * NB;
      const int nb = int(std::min<int64_t>(NB, d - j));
      gemm_ex(c, false, false, r, nb, nb, 1.0, X + j, ldx, invT.get() + bj * NB * NB, NB, 0.0, tmp, NB, X + j, ldx, false);
      const int64_t rem = d - j - nb;
      if (rem > 0)  // X[:, j+nb:] -= X_j  L[j+nb:, j]'
        gemm(c, false, true, r, rem, nb, -1.0, tmp, NB, L + (j + nb) * ldl + j, ldl, 1.0, X + j + nb, ldx);
    }
  } else {
    // X L = B, backward:  X_j = (B_j - sum_{t>j} X_t L_tj) L_jj^-1 ;  L_jj^-1 = (invT_j)'
    for (int64_t bj = nblk - 1; bj >= 0; --bj) {
      const int64_t j = bj * NB;
      const int nb = int(std::min<int64_t>(NB, d - j));
      gemm_ex(c, false, true, r, nb, nb, 1.0, X + j, ldx, invT.get() + bj * NB * NB, NB, 0.0, tmp, NB, X + j, ldx, false);
      if (j > 0)  // X[:, :j] -= X_j L[j:j+nb, :j]
        gemm(c, false, false, r, j, nb, -1.0, tmp, NB, L + j * ldl, ldl, 1.0, X, ldx);
    }
  }
  };
  if (nblk >= 8) graph_run(c, key, body);
  else body();
}


// ---------------------------------------------------------------------------
// Round-2 paths (cholinv.hip):
//  * potrf: the step kernels with look-ahead -- ONE launch per 64-column block for all matrices of the batch (the
//    trailing tiles recompute their panel blocks, the owner of the next diagonal block factors it in the same
//    launch) instead of three launches and a 60 us single-wave factorization per block; above 4096 columns the
//    3x recompute of the tiles costs more than it saves and the factorization runs on 512-column super-blocks
//    (batched step kernels on the diagonal super-blocks, 128-tile GEMMs for panel and trailing update).
//  * trsm: 512-column super-blocks with explicit inverses of the diagonal super-blocks (one batched pass of the
//    inverse rows) -- 8 x (r x 512 x 512 product + rank-512 trailing update) at the big-GEMM rate instead of 64
//    rank-64 updates that were bound by HBM.
// ---------------------------------------------------------------------------
// (CCZ_POTRF_SB: a measurement switch, read once -- 256 / 512 / 1024; every user of the kept inverses sees the same value)
static const int64_t SB = [] {
  const char* e = getenv("CCZ_POTRF_SB");
  const int64_t v = e ? atoll(e) : 512;
  return (v == 256 || v == 1024) ? v : int64_t(512);
}();

static void potrf_lower_batched_steps(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info) {
  for (int b0 = 0; b0 < count; b0 += 8) {
    const int nbt = std::min(8, count - b0);
    std::vector<DBuf> Lb(nbt), Tb(nbt);
    std::vector<double*> Lp(nbt), Tp(nbt);
    std::vector<int64_t> ldl(nbt);
    for (int i = 0; i < nbt; ++i) {
      const int64_t di = d[b0 + i];
      Lb[i] = DBuf(c, di * di);
      Tb[i] = DBuf(c, (di + NB - 1) / NB * NB * NB);
      Lp[i] = Lb[i].get();
      Tp[i] = Tb[i].get();
      ldl[i] = di;
    }
    int* info_dev = impl(c)->d_flag + 8;
    cholinv_batched(c, nbt, A + b0, lda + b0, d + b0, Lp.data(), ldl.data(), nullptr, nullptr, Tp.data(), info_dev);
    for (int i = 0; i < nbt; ++i) copy_lower(c, d[b0 + i], Lp[i], ldl[i], A[b0 + i], lda[b0 + i]);
    int got[8];
    d2h(c, got, info_dev, size_t(nbt) * sizeof(int));
    for (int i = 0; i < nbt; ++i) info[b0 + i] = got[i] == 0x7fffffff ? 0 : got[i];
  }
}

// d > 1024: right-looking over 512-column super-blocks, all matrices of the batch together -- the diagonal
// super-blocks of every matrix are factored (+ inverted) in ONE batched pass of the step kernels (the chain of nine
// dependent launches is shared), panel and trailing update are 128-tile GEMMs with K = 512.  (The step kernels
// alone on a 4096-column matrix are rank-64 updates of the whole trailing matrix -- HBM-bound, measured 8 ms for two
// matrices; the super-blocked form moves 8x less.)  Pivot failures are collected at the end: one host read.
//
// Look-ahead: super-block J + 1 can be factored as soon as ITS diagonal block has received update J, i.e. after the
// top 512 rows of panel J -- not after the whole trailing update.  The factorization is a latency chain on a handful
// of workgroups, the rest of the update is throughput work, so the chain moves to a second stream and the two overlap:
//   main:  copy L_JJ | panel top | diag (J+1,J+1) update | ev_a | panel rest | column J+1 | trailing rest | wait ev_b
//   aux :                                      wait ev_a | factor + invert (J+1,J+1) ......................| ev_b
// CCZ_POTRF_LOOKAHEAD=0 keeps everything on the handle's stream (same order of operations).
//
// keep[i] (optional, nsb * 512 * 512 doubles): receives the inverses of matrix i's diagonal super-blocks for later
// triangular solves with the factor (trsm_right_lower_aux).
static int potrf_lookahead() {
  static const int m = [] { const char* e = getenv("CCZ_POTRF_LOOKAHEAD"); return e ? atoi(e) : 1; }();
  return m;
}

struct StreamSwap {
  ccz_ctx* c;
  void* prev;
  StreamSwap(ccz_ctx* c_, hipStream_t s) : c(c_), prev(c_->stream) { c->stream = s; }
  ~StreamSwap() { c->stream = prev; }
};

static void potrf_lower_batched_sb(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                                   double* const* keep, const TrsmRider* rider = nullptr, int rider_idx = -1) {
  if (count > 8) fail(CCZ_EUNSUP, "potrf_lower_batched_sb: at most 8 matrices per call");
  // the rider's r x 512 product buffer (X_J inv(L_JJ)') -- see the end of the J loop
  DBuf rtmp;
  if (rider) rtmp = DBuf(c, rider->r * SB);
  Impl* im = impl(c);
  int64_t dmax = 0;
  for (int i = 0; i < count; ++i) dmax = std::max(dmax, d[i]);
  const int64_t nsb = (dmax + SB - 1) / SB;
  std::vector<DBuf> Ld(count), Xv(count), Tv(count);
  for (int i = 0; i < count; ++i) {
    Ld[i] = DBuf(c, SB * SB);
    Tv[i] = DBuf(c, (SB / NB) * NB * NB);
    // the panel products read whole blocks; the inverse rows overwrite only the lower blocks
    if (keep && keep[i]) {
      zero(c, keep[i], size_t((d[i] + SB - 1) / SB) * SB * SB * 8);
    } else {
      Xv[i] = DBuf(c, 2 * SB * SB);       // two halves: factor(J + 1) writes one while update J still reads the other
      zero(c, Xv[i], size_t(2) * SB * SB * 8);
    }
  }
  // info of super-step J of matrix i: slot J * 8 + (position among the matrices alive at J); all start at "no failure"
  const int nslots = int(nsb) * 8;
  int* info_dev = static_cast<int*>(dev_alloc(c, size_t(nslots) * sizeof(int)));
  // Unwinding (an allocation or launch failure between the look-ahead launch and the main stream's wait): the pooled
  // buffers below go back to the pool as the DBufs unwind, and the pool recycles them in the order of the MAIN stream --
  // so the look-ahead stream must be drained first, and the pivot-flag block released.
  struct Unwind {
    ccz_ctx* c; Impl* im; int* info; bool armed = true;
    ~Unwind() {
      if (!armed) return;
      if (im->aux_stream) (void)hipStreamSynchronize(im->aux_stream);
      dev_free(c, info);
    }
  } unwind{c, im, info_dev};

  hipStream_t s_main = stream(c), s_aux = s_main;
  bool la = potrf_lookahead() != 0 && nsb > 1;
  if (la) {
    if (!im->aux_stream) {
      hipStream_t st = nullptr;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
          hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) {
        im->aux_stream = st; im->aux_ev[0] = e0; im->aux_ev[1] = e1;
      } else {
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        if (st) (void)hipStreamDestroy(st);
        la = false;
      }
    }
    if (la) s_aux = im->aux_stream;
  }

  // diagonal super-block J of every matrix still alive: factor, invert (on the CURRENT c->stream)
  auto factor = [&](int64_t J) {
    const int64_t j0 = J * SB;
    double* Ap[8]; double* Lp[8]; double* Xp[8]; double* Tp[8];
    int64_t la_[8], dd[8], ll[8];
    int cnt = 0;
    bool need_x = false;
    for (int i = 0; i < count; ++i) {
      if (j0 >= d[i]) continue;
      Ap[cnt] = A[i] + j0 * lda[i] + j0;
      la_[cnt] = lda[i];
      dd[cnt] = std::min(SB, d[i] - j0);
      Lp[cnt] = Ld[i].get();
      Xp[cnt] = (keep && keep[i]) ? keep[i] + J * SB * SB : Xv[i].get() + (J & 1) * SB * SB;
      Tp[cnt] = Tv[i].get();
      ll[cnt] = SB;
      need_x = need_x || (keep && keep[i]) || d[i] - j0 - dd[cnt] > 0;
      ++cnt;
    }
    if (cnt > 0) cholinv_batched(c, cnt, Ap, la_, dd, Lp, ll, need_x ? Xp : nullptr, ll, Tp, info_dev + J * 8);
  };

  factor(0);
  for (int64_t J = 0; J < nsb; ++J) {
    const int64_t j0 = J * SB;
    std::vector<DBuf> tmp(count);
    // ---- the part super-block J + 1 waits for: top block of panel J and the update of diagonal block J + 1.
    // Two 512^3 products per matrix -- far too small for a launch each (40 us on 64 workgroups): one batched,
    // split-K launch per stage for all matrices (512 workgroups, ~12 us)
    {
      MultiGemmArgs top[8], dg[8];
      int nt = 0;
      for (int i = 0; i < count; ++i) {
        if (j0 >= d[i]) continue;
        const int64_t w = std::min(SB, d[i] - j0), rem = d[i] - j0 - w;
        copy_lower(c, w, Ld[i], SB, A[i] + j0 * lda[i] + j0, lda[i]);
        if (rem <= 0) continue;
        const int64_t w1 = std::min(SB, rem);
        const double* Xj = (keep && keep[i]) ? keep[i] + J * SB * SB : Xv[i].get() + (J & 1) * SB * SB;
        tmp[i] = DBuf(c, rem * w);
        zero(c, tmp[i], size_t(w1) * w * 8);
        MultiGemmArgs& a = top[nt];                                                     // top of L21 = A21 L11^-T
        a = MultiGemmArgs{};
        a.A = A[i] + (j0 + w) * lda[i] + j0; a.lda = lda[i]; a.tA = false;
        a.B = Xj; a.ldb = SB; a.tB = true;
        a.C = tmp[i]; a.ldc = w; a.Ct = nullptr; a.ldct = 0;
        a.M = w1; a.N = w; a.K = w; a.alpha = 1.0; a.beta = 1.0; a.lower_only = false; a.ksplit = 4;
        MultiGemmArgs& b = dg[nt];                                                      // A(J+1,J+1) -= top top'
        b = MultiGemmArgs{};
        b.A = tmp[i]; b.lda = w; b.tA = false;
        b.B = tmp[i]; b.ldb = w; b.tB = true;
        b.C = A[i] + (j0 + w) * lda[i] + (j0 + w); b.ldc = lda[i]; b.Ct = nullptr; b.ldct = 0;
        b.M = w1; b.N = w1; b.K = w; b.alpha = -1.0; b.beta = 1.0; b.lower_only = true; b.ksplit = 4;
        ++nt;
      }
      if (nt > 0) {
        gemm_f64_multi(c, nt, top);
        gemm_f64_multi(c, nt, dg);
      }
    }
    if (J + 1 < nsb) {
      if (la) {
        CCZ_HIP(hipEventRecord(im->aux_ev[0], s_main));
        CCZ_HIP(hipStreamWaitEvent(s_aux, im->aux_ev[0], 0));
        {
          StreamSwap sw(c, s_aux);
          // the chain kernel's helper workgroups spin while they wait: next to the update GEMMs of the main stream they get a
          // bounded share of the chip (CCZ_CHAIN_WGS_LA, default 64 workgroups)
          static const int la_cap = [] { const char* e = getenv("CCZ_CHAIN_WGS_LA"); return e ? atoi(e) : 64; }();
          const int cap0 = im->chain_cap;
          im->chain_cap = la_cap;
          try { factor(J + 1); } catch (...) { im->chain_cap = cap0; throw; }
          im->chain_cap = cap0;
        }
        CCZ_HIP(hipEventRecord(im->aux_ev[1], s_aux));
      } else {
        factor(J + 1);
      }
    }
    // ---- the rest of update J (overlaps the factorization of super-block J + 1) ----
    for (int i = 0; i < count; ++i) {
      if (j0 >= d[i]) continue;
      const int64_t w = std::min(SB, d[i] - j0), rem = d[i] - j0 - w;
      if (rem <= 0) continue;
      const int64_t w1 = std::min(SB, rem), rem2 = rem - w1;
      const int64_t j1 = j0 + w;
      copy2d(c, w1, w, tmp[i], w, A[i] + j1 * lda[i] + j0, lda[i]);                                  // top of L21 -> the factor
      if (rem2 <= 0) continue;
      double* A31 = A[i] + (j1 + w1) * lda[i] + j0;
      double* t2 = tmp[i].get() + w1 * w;
      const double* Xj = (keep && keep[i]) ? keep[i] + J * SB * SB : Xv[i].get() + (J & 1) * SB * SB;
      gemm(c, false, true, rem2, w, w, 1.0, A31, lda[i], Xj, SB, 0.0, t2, w);                       // rest of L21
      copy2d(c, rem2, w, t2, w, A31, lda[i]);
      gemm(c, false, true, rem2, w1, w, -1.0, t2, w, tmp[i], w, 1.0, A[i] + (j1 + w1) * lda[i] + j1, lda[i]);   // column J + 1
      gemm_ex(c, false, true, rem2, rem2, w, -1.0, t2, w, t2, w, 1.0, A[i] + (j1 + w1) * lda[i] + (j1 + w1), lda[i], nullptr, 0, true);
    }
    // ---- the rider: column block J of its factor is final now (diagonal block, panel top, panel rest), so step J of
    // X <- X L^-T can go: X_J <- X_J inv(L_JJ)', X[:, below] -= X_J L[below, J]'  (trsm_right_lower_sb, trans branch).
    // It queues behind update J on the main stream -- throughput work under the factorization chain of block J + 1.
    if (rider && j0 < d[rider_idx]) {
      const int i = rider_idx;
      if (J == 0 && rider->prepare) rider->prepare();
      const int64_t w = std::min(SB, d[i] - j0), rem = d[i] - j0 - w;
      const double* Xinv = keep[i] + J * SB * SB;
      double* X = rider->X;
      gemm(c, false, true, rider->r, w, w, 1.0, X + j0, rider->ldx, Xinv, SB, 0.0, rtmp, SB);
      copy2d(c, rider->r, w, rtmp, SB, X + j0, rider->ldx);
      if (rem > 0)
        gemm(c, false, true, rider->r, rem, w, -1.0, rtmp, SB, A[i] + (j0 + w) * lda[i] + j0, lda[i], 1.0, X + j0 + w, rider->ldx);
    }
    if (la && J + 1 < nsb) CCZ_HIP(hipStreamWaitEvent(s_main, im->aux_ev[1], 0));
  }
  std::vector<int> got(nslots, 0x7fffffff);
  // slots of (J, t) with t >= cnt of that super-step were never written: only read what cholinv_batched initialised
  d2h(c, got.data(), info_dev, size_t(nslots) * sizeof(int));
  unwind.armed = false;
  dev_free(c, info_dev);
  for (int i = 0; i < count; ++i) info[i] = 0;
  for (int64_t J = 0; J < nsb; ++J) {
    int cnt = 0;
    for (int i = 0; i < count; ++i) {
      if (J * SB >= d[i]) continue;
      const int v = got[J * 8 + cnt];
      if (v != 0x7fffffff && info[i] == 0) info[i] = int(J * SB) + v;
      ++cnt;
    }
  }
}

static bool potrf_lower_batched_new(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                                    double* const* aux = nullptr, const TrsmRider* rider = nullptr) {
  // matrices up to 1024 columns go through the step kernels directly; wider ones super-blocked; both batched by 8
  std::vector<double*> As, Ab, Kb;
  std::vector<int64_t> ds, lds_, db, ldb_;
  std::vector<int> idx, idb;
  for (int i = 0; i < count; ++i) {
    if (d[i] <= 1024) { As.push_back(A[i]); ds.push_back(d[i]); lds_.push_back(lda[i]); idx.push_back(i); }
    else { Ab.push_back(A[i]); db.push_back(d[i]); ldb_.push_back(lda[i]); idb.push_back(i); Kb.push_back(aux ? aux[i] : nullptr); }
  }
  if (!As.empty()) {
    std::vector<int> inf(As.size(), 0);
    potrf_lower_batched_steps(c, int(As.size()), As.data(), ds.data(), lds_.data(), inf.data());
    for (size_t t = 0; t < idx.size(); ++t) info[idx[t]] = inf[t];
  }
  bool rode = false;
  for (size_t b0 = 0; b0 < Ab.size(); b0 += 8) {
    const int nbt = int(std::min<size_t>(8, Ab.size() - b0));
    int inf[8];
    // the rider needs the kept inverses of its factor's diagonal super-blocks
    int ridx = -1;
    if (rider && rider->X && rider->r > 0)
      for (int t = 0; t < nbt; ++t)
        if (idb[b0 + t] == rider->matrix && Kb[b0 + t]) ridx = t;
    potrf_lower_batched_sb(c, nbt, Ab.data() + b0, db.data() + b0, ldb_.data() + b0, inf, Kb.data() + b0, ridx >= 0 ? rider : nullptr, ridx);
    rode = rode || ridx >= 0;
    for (int t = 0; t < nbt; ++t) info[idb[b0 + t]] = inf[t];
  }
  return rode;
}

static void trsm_right_lower_sb(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X, int64_t ldx,
                                const double* aux = nullptr) {
  const int64_t nblk = (d + NB - 1) / NB, nsb = (d + SB - 1) / SB;
  DBuf tmp(c, r * SB), Xown;
  const double* Xinv = aux;                                    // inverses of the diagonal super-blocks, kept by potrf
  if (!Xinv) {
    DBuf invT(c, nblk * NB * NB);
    Xown = DBuf(c, nsb * SB * SB);
    diag_inverses(c, L, ldl, d, invT);                         // L_jj^-T of every 64-column block, one launch
    zero(c, Xown, size_t(nsb) * SB * SB * 8);                  // the products read whole super-blocks
    for (int64_t J0 = 0; J0 < nsb; J0 += 8) {
      const int cnt = int(std::min<int64_t>(8, nsb - J0));
      const double* Lp[8];
      const double* Tp[8];
      double* Xp[8];
      int64_t dd[8], l1[8], l2[8];
      for (int t = 0; t < cnt; ++t) {
        const int64_t j0 = (J0 + t) * SB;
        Lp[t] = L + j0 * ldl + j0;
        Tp[t] = invT.get() + (j0 / NB) * NB * NB;
        Xp[t] = Xown.get() + (J0 + t) * SB * SB;
        dd[t] = std::min(SB, d - j0);
        l1[t] = ldl;
        l2[t] = SB;
      }
      trinv_batched(c, cnt, Lp, l1, dd, Xp, l2, Tp);
    }
    Xinv = Xown.get();
  }
  if (trans) {
    // X L' = B, forward over the super-blocks:  X_J = (B_J - sum_{t<J} X_t L_Jt') L_JJ^-T
    for (int64_t J = 0; J < nsb; ++J) {
      const int64_t j0 = J * SB, w = std::min(SB, d - j0), rem = d - j0 - w;
      gemm(c, false, true, r, w, w, 1.0, X + j0, ldx, Xinv + J * SB * SB, SB, 0.0, tmp, SB);
      copy2d(c, r, w, tmp, SB, X + j0, ldx);
      if (rem > 0) gemm(c, false, true, r, rem, w, -1.0, tmp, SB, L + (j0 + w) * ldl + j0, ldl, 1.0, X + j0 + w, ldx);
    }
  } else {
    // X L = B, backward:  X_J = (B_J - sum_{t>J} X_t L_tJ) L_JJ^-1
    for (int64_t J = nsb - 1; J >= 0; --J) {
      const int64_t j0 = J * SB, w = std::min(SB, d - j0);
      gemm(c, false, false, r, w, w, 1.0, X + j0, ldx, Xinv + J * SB * SB, SB, 0.0, tmp, SB);
      copy2d(c, r, w, tmp, SB, X + j0, ldx);
      if (j0 > 0) gemm(c, false, false, r, j0, w, -1.0, tmp, SB, L + j0 * ldl, ldl, 1.0, X, ldx);
    }
  }
}

// Dispatch: the recursive forms put the flops into large-K GEMMs but issue ~40% more (tiny) launches;
// they pay off once the matrices are big enough for the GEMMs to dominate the launch latency.
static int solver_mode() {
  static const int m = [] { const char* e = getenv("CCZ_SOLVER_RECURSIVE"); return e ? atoi(e) : -1; }();
  return m;   // -1: automatic, 0: iterative, 1: recursive
}

static int solver_legacy() {
  static const int m = [] { const char* e = getenv("CCZ_SOLVER_LEGACY"); return e ? atoi(e) : 0; }();
  return m;   // 1: the round-1 blocked Cholesky / rank-64 triangular solves
}

void potrf_lower_batched(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info) {
  int64_t dmax = 0;
  for (int i = 0; i < count; ++i) dmax = std::max(dmax, d[i]);
  if (!solver_legacy()) { potrf_lower_batched_new(c, count, A, d, lda, info); return; }
  const int mode = solver_mode();
  if (mode == 1 || (mode < 0 && dmax >= 6144)) potrf_lower_batched_rec(c, count, A, d, lda, info);
  else potrf_lower_batched_iter(c, count, A, d, lda, info);
}

// Factor and inverse of ONE small SPD matrix (ops.h): the chain kernel produces L^-1 next to L (cholinv_batched with X), so the
// Cholesky-QR passes of the subspace iteration need no triangular solve -- X <- X L^-T is one product with the explicit inverse.
int potrf_lower_inv(ccz_ctx* c, double* A, int64_t d, int64_t lda, double* Linv, int64_t ldi) {
  if (d < 1) fail(CCZ_EINVAL, "potrf_lower_inv: d >= 1 required");
  if (solver_legacy() || d > 4096) {
    // (no chain kernel: factor, then L^-1 = I L^-1 by the triangular solve)
    const int info = potrf_lower(c, A, d, lda);
    if (info != 0) return info;
    fill2d(c, d, d, Linv, ldi, 0.0);
    add_diag(c, d, Linv, ldi, 1.0);
    trsm_right_lower(c, false, d, d, A, lda, Linv, ldi);
    return 0;
  }
  DBuf Lb(c, d * d), Tb(c, (d + NB - 1) / NB * NB * NB);
  double* Ap[1] = {A};
  double* Lp[1] = {Lb.get()};
  double* Xp[1] = {Linv};
  double* Tp[1] = {Tb.get()};
  const int64_t dd[1] = {d}, la[1] = {lda}, ll[1] = {d}, lx[1] = {ldi};
  fill2d(c, d, d, Linv, ldi, 0.0);                     // (the chain writes the lower triangle only)
  int* info_dev = impl(c)->d_flag + 8;
  cholinv_batched(c, 1, Ap, la, dd, Lp, ll, Xp, lx, Tp, info_dev);
  int got = 0;
  d2h(c, &got, info_dev, sizeof(int));
  return got == 0x7fffffff ? 0 : got;
}

int64_t trsm_aux_size(ccz_ctx*, int64_t d) {
  if (solver_legacy() || d <= 1024) return 0;
  return (d + SB - 1) / SB * SB * SB;
}

void potrf_lower_batched_aux(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                             double* const* aux) {
  if (solver_legacy() || !aux) { potrf_lower_batched(c, count, A, d, lda, info); return; }
  potrf_lower_batched_new(c, count, A, d, lda, info, aux);
}

bool potrf_lower_batched_aux_rider(ccz_ctx* c, int count, double* const* A, const int64_t* d, const int64_t* lda, int* info,
                                   double* const* aux, const TrsmRider* rider) {
  static const int rider_on = [] { const char* e = getenv("CCZ_POTRF_RIDER"); return e ? atoi(e) : 1; }();
  if (solver_legacy() || !aux || !rider_on || !rider) { potrf_lower_batched_aux(c, count, A, d, lda, info, aux); return false; }
  return potrf_lower_batched_new(c, count, A, d, lda, info, aux, rider);
}

void trsm_right_lower_aux(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                          int64_t ldx, const double* aux) {
  if (r <= 0 || d <= 0) return;
  if (aux && !solver_legacy() && d > 1024) { trsm_right_lower_sb(c, trans, r, d, L, ldl, X, ldx, aux); return; }
  trsm_right_lower(c, trans, r, d, L, ldl, X, ldx);
}

void trsm_right_lower(ccz_ctx* c, bool trans, int64_t r, int64_t d, const double* L, int64_t ldl, double* X,
                      int64_t ldx) {
  if (r <= 0 || d <= 0) return;
  if (!solver_legacy() && d >= 1024) { trsm_right_lower_sb(c, trans, r, d, L, ldl, X, ldx); return; }
  const int mode = solver_mode();
  if (mode == 1 || (mode < 0 && d >= 2048 && r >= 6144)) trsm_right_lower_rec(c, trans, r, d, L, ldl, X, ldx);
  else trsm_right_lower_iter(c, trans, r, d, L, ldl, X, ldx);
}

// Few-row solves of several problems together (back-projection of the k wanted directions through every view's
// factor).  One problem alone is 2 dependent launches per super-block, each a 64-row product that cannot fill the chip
// (31 us apiece, 16 per 4096-column factor); here every step is two batched split-K launches for ALL problems:
//   stage 1:  Y_b[:, J] += X_b[:, J] inv(L_b,JJ)^(T)           (Y zeroed once; result accumulates out of place)
//   stage 2:  X_b[:, rest] -= Y_b[:, J] L_b[., .]^(T)            (the not yet solved columns)
void trsm_right_lower_aux_multi(ccz_ctx* c, int count, bool trans, const int64_t* r, const int64_t* d, const double* const* L,
                                const int64_t* ldl, double* const* X, const int64_t* ldx, const double* const* aux) {
  bool batched = !solver_legacy() && count >= 1 && count <= 8;
  for (int b = 0; b < count && batched; ++b) batched = aux && aux[b] && d[b] > 1024 && r[b] >= 1 && r[b] <= 256;
  if (!batched) {
    for (int b = 0; b < count; ++b) trsm_right_lower_aux(c, trans, r[b], d[b], L[b], ldl[b], X[b], ldx[b], aux ? aux[b] : nullptr);
    return;
  }
  static const int bp_split = [] { const char* e = getenv("CCZ_BACKPROJ_SPLIT"); return e ? atoi(e) : 4; }();
  int64_t total = 0, offs[8], nsb[8], max_nsb = 0;
  for (int b = 0; b < count; ++b) {
    offs[b] = total;
    total += r[b] * d[b];
    nsb[b] = (d[b] + SB - 1) / SB;
    max_nsb = std::max(max_nsb, nsb[b]);
  }
  DBuf Y(c, total);
  zero(c, Y, size_t(total) * 8);
  for (int64_t st = 0; st < max_nsb; ++st) {
    MultiGemmArgs s1[8], s2[8];
    int n1 = 0, n2 = 0;
    for (int b = 0; b < count; ++b) {
      if (st >= nsb[b]) continue;
      const int64_t J = trans ? st : nsb[b] - 1 - st;
      const int64_t j0 = J * SB, w = std::min(SB, d[b] - j0);
      double* Yb = Y.get() + offs[b];
      MultiGemmArgs& a = s1[n1++];
      a = MultiGemmArgs{};
      a.A = X[b] + j0; a.lda = ldx[b]; a.tA = false;
      a.B = aux[b] + J * SB * SB; a.ldb = SB; a.tB = trans;
      a.C = Yb + j0; a.ldc = d[b]; a.Ct = nullptr; a.ldct = 0;
      a.M = r[b]; a.N = w; a.K = w; a.alpha = 1.0; a.beta = 1.0; a.lower_only = false; a.ksplit = bp_split;
      const int64_t rest = trans ? d[b] - j0 - w : j0;
      if (rest <= 0) continue;
      MultiGemmArgs& u = s2[n2++];
      u = MultiGemmArgs{};
      u.A = Yb + j0; u.lda = d[b]; u.tA = false;
      if (trans) { u.B = L[b] + (j0 + w) * ldl[b] + j0; u.tB = true; u.C = X[b] + j0 + w; }      // X[:, below] -= Y_J L[below, J]'
      else { u.B = L[b] + j0 * ldl[b]; u.tB = false; u.C = X[b]; }                                  // X[:, :j0]  -= Y_J L[J, :j0]
      u.ldb = ldl[b]; u.ldc = ldx[b]; u.Ct = nullptr; u.ldct = 0;
      u.M = r[b]; u.N = rest; u.K = w; u.alpha = -1.0; u.beta = 1.0; u.lower_only = false; u.ksplit = bp_split;
    }
    if (n1 > 0) gemm_f64_multi(c, n1, s1);
    if (n2 > 0) gemm_f64_multi(c, n2, s2);
  }
  for (int b = 0; b < count; ++b) copy2d(c, r[b], d[b], Y.get() + offs[b], d[b], X[b], ldx[b]);
}

// ===========================================================================
// one-sided Jacobi on rows: one workgroup per row pair, one launch per tournament round
// ===========================================================================
template <int BS>
__global__ __launch_bounds__(BS) void k_jacobi_round(int64_t p, int64_t pe, int64_t q, double* __restrict__ W,
                                                     int64_t ldw, double* __restrict__ Q, int64_t qc, int64_t ldq,
                                                     int64_t round, double tol, double floor2,
                                                     int* __restrict__ counter) {
  __shared__ double red[3][BS / 64 > 0 ? BS / 64 : 1];
  const int64_t k = blockIdx.x, m1 = pe - 1;
  int64_t a, b;
  if (k == 0) { a = m1; b = round; } else { a = (round + k) % m1; b = (round - k + m1) % m1; }
  if (a >= p || b >= p) return;
  double* wa = W + a * ldw;
  double* wb = W + b * ldw;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (int64_t t = threadIdx.x; t < q; t += BS) {
    const double x = wa[t], y = wb[t];
    al += x * x; be += y * y; ga += x * y;
  }
  al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
  if (BS > 64) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = al; red[1][wave] = be; red[2][wave] = ga; }
    __syncthreads();
    al = be = ga = 0.0;
    for (int i = 0; i < BS / 64; ++i) { al += red[0][i]; be += red[1][i]; ga += red[2][i]; }
  }
  const double prod = al * be;
  // rows whose norm has sunk below 1e-14 of the largest row are numerically zero: rotating rounding
  // noise against real rows never meets the relative criterion (rank-deficient inputs)
  if (!(prod > 0.0) || !(fmin(al, be) > floor2) || !(fabs(ga) > tol * sqrt(prod))) return;
  if (threadIdx.x == 0) atomicAdd(counter, 1);
  const double zeta = (be - al) / (2.0 * ga);
  const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
  for (int64_t t = threadIdx.x; t < q; t += BS) {
    const double x = wa[t], y = wb[t];
    wa[t] = cs * x - sn * y;
    wb[t] = sn * x + cs * y;
  }
  if (Q) {
    double* qa = Q + a * ldq;
    double* qb = Q + b * ldq;
    for (int64_t t = threadIdx.x; t < qc; t += BS) {
      const double x = qa[t], y = qb[t];
      qa[t] = cs * x - sn * y;
      qb[t] = sn * x + cs * y;
    }
  }
}

// Small problems (the Rayleigh-Ritz blocks of the subspace iteration, p ~ 80): the whole W and Q
// live in LDS and ONE workgroup runs every round of every sweep.  A row pair is handled by a
// 16-lane DPP row (64 pairs per pass of the 1024-thread workgroup); the three inner products
// are reduced with DPP lane permutes (quad_perm, row_half_mirror, row_mirror) -- no LDS round
// trips, no launches, no global traffic inside the sweeps.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);   // row_half_mirror
  v += dpp_mov_f64<0x140>(v);   // row_mirror
  return v;
}

__global__ __launch_bounds__(1024) void k_jacobi_lds(int p, int q, double* __restrict__ W, int64_t ldw,
                                                     double* __restrict__ Q, int qc, int64_t ldq, double tol,
                                                     double floor2, int max_sweeps, int* __restrict__ sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int sq = q | 1, sqc = qc | 1;                 // odd row strides: conflict-free column walks
  double* Ws = reinterpret_cast<double*>(smem);
  double* Qs = Ws + size_t(p) * sq;
  // the counter lives at the end of the dynamic region (a static __shared__ object in front of it
  // would shift the region's base off its 16-byte alignment)
  int& rot_count = *reinterpret_cast<int*>(Qs + (Q ? size_t(p) * sqc : 0));
  const int tid = threadIdx.x, gl = tid & 15, grp = tid >> 4, ngrp = blockDim.x >> 4;
  for (int i = tid; i < p * q; i += blockDim.x) Ws[(i / q) * sq + i % q] = W[int64_t(i / q) * ldw + i % q];
  if (Q) for (int i = tid; i < p * qc; i += blockDim.x) Qs[(i / qc) * sqc + i % qc] = Q[int64_t(i / qc) * ldq + i % qc];
  const int pe = (p + 1) & ~1, m1 = pe - 1;
  int sweep = 0, done = 0;
  __syncthreads();
  while (sweep < max_sweeps && !done) {
    ++sweep;
    if (tid == 0) rot_count = 0;
    __syncthreads();
    for (int round = 0; round < m1; ++round) {
      for (int k = grp; k < pe / 2; k += ngrp) {
        int a, b;
        if (k == 0) { a = m1; b = round; } else { a = (round + k) % m1; b = (round - k + m1) % m1; }
        const bool live = a < p && b < p;              // uniform over the 16-lane row
        double* wa = Ws + (live ? a : 0) * sq;
        double* wb = Ws + (live ? b : 0) * sq;
        double al = 0.0, be = 0.0, ga = 0.0;
        if (live)
          for (int t = gl; t < q; t += 16) { const double x = wa[t], y = wb[t]; al += x * x; be += y * y; ga += x * y; }
        al = row16_sum(al); be = row16_sum(be); ga = row16_sum(ga);
        const double prod = al * be;
        if (!live || !(prod > 0.0) || !(fmin(al, be) > floor2) || !(fabs(ga) > tol * sqrt(prod))) continue;
        if (gl == 0) atomicAdd(&rot_count, 1);
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
        for (int t = gl; t < q; t += 16) { const double x = wa[t], y = wb[t]; wa[t] = cs * x - sn * y; wb[t] = sn * x + cs * y; }
        if (Q) {
          double* qa = Qs + a * sqc;
          double* qb = Qs + b * sqc;
          for (int t = gl; t < qc; t += 16) { const double x = qa[t], y = qb[t]; qa[t] = cs * x - sn * y; qb[t] = sn * x + cs * y; }
        }
      }
      __syncthreads();
    }
    done = (rot_count == 0);
    __syncthreads();
  }
  for (int i = tid; i < p * q; i += blockDim.x) W[int64_t(i / q) * ldw + i % q] = Ws[(i / q) * sq + i % q];
  if (Q) for (int i = tid; i < p * qc; i += blockDim.x) Q[int64_t(i / qc) * ldq + i % qc] = Qs[(i / qc) * sqc + i % qc];
  if (tid == 0) *sweeps_out = done ? sweep : -1;
}

int jacobi_rows(ccz_ctx* c, int64_t p, int64_t q, double* W, int64_t ldw, double* Q, int64_t qc, int64_t ldq,
                int max_sweeps) {
  if (p < 2) return 1;
  Impl* im = impl(c);
  const int64_t pe = (p + 1) & ~int64_t(1);
  const double tol = 2.220446049250313e-16 * std::sqrt(double(q)) * 4.0;
  // rows whose squared norm is below 1e-28 of the largest never rotate (one row_dots + one blocking read-back): only the
  // one-workgroup kernel and the legacy loop take it as an argument -- the block form finds its own on the device
  auto rest_floor = [&]() {
    DBuf nn(c, p);
    row_dots(c, p, q, W, ldw, W, ldw, nn);
    std::vector<double> nh(p);
    d2h(c, nh.data(), nn, size_t(p) * 8);
    double mx = 0.0;
    for (double v : nh) mx = std::max(mx, v);
    return mx * 1e-28;
  };
  double floor2 = 0.0;
  const size_t lds_need = (size_t(p) * (q | 1) + (Q ? size_t(p) * (qc | 1) : 0)) * 8;
  if (lds_need <= size_t(144) * 1024) {
    floor2 = rest_floor();
    CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_jacobi_lds), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need + 16)));
    hipLaunchKernelGGL(k_jacobi_lds, dim3(1), dim3(1024), lds_need + 16, stream(c), int(p), int(q), W, ldw, Q, int(Q ? qc : 0), ldq,
                       tol, floor2, max_sweeps, im->d_flag + 1);
    CCZ_LAUNCH_CHECK();
    int sw = 0;
    d2h(c, &sw, im->d_flag + 1, sizeof(int));
    if (sw < 0) fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps (p=%lld, q=%lld)", max_sweeps, (long long)p, (long long)q);
    return sw;
  }
  static const bool legacy = [] { const char* e = getenv("CCZ_EVD_LEGACY"); return e && atoi(e) != 0; }();
  if (!legacy) {
    // evd_block.hip: 32-row blocks, Gram blocks + rotations as MFMA tiles.  It wants a multiple of 64 rows and even
    // leading dimensions (16-byte row pieces): pad with zero rows (they never rotate) when the caller's shape differs.
    const int64_t pp = (p + 63) / 64 * 64;
    if (pp == p && (ldw & 1) == 0 && (!Q || (ldq & 1) == 0)) return jacobi_rows_block(c, p, q, W, ldw, Q, qc, ldq, max_sweeps);
    const int64_t lw = (q + 1) & ~int64_t(1), lq = (qc + 1) & ~int64_t(1);
    DBuf Wp(c, pp * lw), Qp(c, Q ? pp * lq : 0);
    fill2d(c, pp, lw, Wp, lw, 0.0);
    copy2d(c, p, q, W, ldw, Wp, lw);
    if (Q) { fill2d(c, pp, lq, Qp, lq, 0.0); copy2d(c, p, qc, Q, ldq, Qp, lq); }
    const int sw = jacobi_rows_block(c, pp, q, Wp, lw, Q ? Qp.get() : nullptr, qc, lq, max_sweeps);
    copy2d(c, p, q, Wp, lw, W, ldw);
    if (Q) copy2d(c, p, qc, Qp, lq, Q, ldq);
    return sw;
  }
  const bool small = std::max(q, Q ? qc : 0) <= 256;
  dim3 grid((unsigned)(pe / 2));
  floor2 = rest_floor();
  for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
    CCZ_HIP(hipMemsetAsync(im->d_flag + 1, 0, sizeof(int), stream(c)));
    // one sweep = pe - 1 dependent tiny launches: replayed as a hipGraph after the first sweep
    uint64_t key = key_mix(key_mix(0x4a41434full, uint64_t(p)), uint64_t(q));
    key = key_mix(key_mix(key_ptr(key_ptr(key, W), Q), uint64_t(ldw)), uint64_t(ldq));
    key = key_mix(key_mix(key, uint64_t(qc)), uint64_t(floor2 * 1e300));
    graph_run(c, key, [&] {
      for (int64_t round = 0; round < pe - 1; ++round) {
        if (small) hipLaunchKernelGGL(k_jacobi_round<64>, grid, dim3(64), 0, stream(c), p, pe, q, W, ldw, Q, qc, ldq, round, tol, floor2, im->d_flag + 1);
        else hipLaunchKernelGGL(k_jacobi_round<256>, grid, dim3(256), 0, stream(c), p, pe, q, W, ldw, Q, qc, ldq, round, tol, floor2, im->d_flag + 1);
      }
      CCZ_LAUNCH_CHECK();
    });
    int rot = 0;
    d2h(c, &rot, im->d_flag + 1, sizeof(int));
    if (rot == 0) return sweep;
  }
  fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps (p=%lld, q=%lld)", max_sweeps, (long long)p, (long long)q);
}

// ===========================================================================
// Two-sided (classical) Jacobi for the small symmetric eigenproblems of the Rayleigh-Ritz steps (d <= 96)
// ===========================================================================
// The one-sided kernel above spends a round on three length-d inner products per row pair before it can rotate;
// on a symmetric H the rotation of pair (p, q) follows from h_pp, h_qq, h_pq alone, and one round of the tournament
// (d/2 disjoint pairs) is   H <- J' H J,  V' <- J' V'   with J = product of the round's rotations: every 2 x 2 block
// (row pair a, column pair b) becomes R_a' M R_b independently of all others.  One workgroup, H and V' in LDS, a
// two-stage pipeline per round (parameters double-buffered):
//   A: all 16 waves apply round g to H                                                  | barrier
//   B: wave 0 derives the parameters of round g + 1 from the new H  ||  waves 1..15 apply round g to V'   | barrier
// Every thread owns the same blocks / V' items in every round and gets the round's row indices by two integer
// adds (the tournament shifts positions by one per round): no index tables, no divisions inside the loop, one level
// of LDS latency per stage.  The long-latency part of a round is wave 0's (sqrt, divide, reciprocal sqrt) chain: done
// with v_rsq / v_rcp seeds and two Newton steps each (~200 cycles instead of ~550 for the IEEE expansions).
// Stopping: a sweep without a rotation; a pair is rotated when |h_pq| > tol * max|H| (absolute: the Rayleigh-Ritz
// matrices are indefinite, zero diagonals happen -- MCCA with two views has exact +lam / -lam pairs).
// Rows/columns are padded to an even count; the pad row/column is zero and stays zero (its pair never rotates).
template <int NB_, int NV_>   // H blocks / V' items per thread (compile-time: the slot arrays must stay in registers)
__device__ __forceinline__ void syev_small_body(char* smem, int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                double* __restrict__ Vt, int64_t ldv, double tol, int max_sweeps,
                                                int* __restrict__ status) {
  const int pe = (d + 1) & ~1, m1 = pe - 1, np = pe >> 1, sd = pe | 1;
  double* Hs = reinterpret_cast<double*>(smem);
  double* Vs = Hs + pe * sd;
  double* red = Vs + pe * sd;                              // 16 wave maxima
  int* rot = reinterpret_cast<int*>(red + 16);             // [2]: rotations of the even / odd sweeps (+ 2 ints of padding)
  jac_cs* csn = reinterpret_cast<jac_cs*>(red + 18);       // [2][np] (cosine, sine), 16-byte aligned: (2 pe sd + 18) is even
  const int tid = threadIdx.x, nt = blockDim.x;            // nt == 1024
  double mx = 0.0;
  for (int i = tid; i < pe * pe; i += nt) {
    const int r = i / pe, c = i - r * pe;
    double h = 0.0;
    if (r < d && c < d) {
      h = 0.5 * (A[int64_t(r) * lda + c] + A[int64_t(c) * lda + r]);
      const double a = fabs(h);
      mx = (a <= 1.79769313486231570e308) ? fmax(mx, a) : __builtin_inf();   // NaN / inf poison the maximum
    }
    Hs[r * sd + c] = h;
    Vs[r * sd + c] = (r == c && r < d) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  if (tid < 2) rot[tid] = 0;
  __syncthreads();
  double hmax = 0.0;
  for (int i = 0; i < (nt + 63) >> 6; ++i) hmax = fmax(hmax, red[i]);
  if (!(hmax < __builtin_inf())) {
    if (tid == 0) *status = -2;
    return;
  }
  if (hmax == 0.0) {                                       // the zero matrix: eigenvalues 0, V = I
    for (int i = tid; i < d; i += nt) w[i] = 0.0;
    for (int i = tid; i < d * d; i += nt) Vt[int64_t(i / d) * ldv + i % d] = (i / d == i % d) ? 1.0 : 0.0;
    if (tid == 0) *status = 1;
    return;
  }
  const double thr = tol * hmax, ih = 1.0 / hmax;

  // This thread's blocks of H and items of V': the same in every round.  np <= 48: at most 2304 blocks (3 per
  // thread) and 48 * 96 = 4608 V' items on the 960 threads of waves 1..15 (5 per thread; d <= 80: 2 and 4).  Slots past the end are
  // pointed at block / item 0 and masked at the store, so that every load below is unconditional.
  const int nblk = np * np, nitem = np * d;
  int bka[NB_], bkb[NB_];
  bool bon[NB_];
#pragma unroll
  for (int j = 0; j < NB_; ++j) {
    const int b = tid + j * nt;
    bon[j] = b < nblk;
    bka[j] = bon[j] ? b / np : 0;
    bkb[j] = bon[j] ? b - bka[j] * np : 0;
  }
  const int vt = tid - 64, nvt = nt - 64;                  // waves 1..15 rotate V'
  int vk[NV_], vr[NV_];
  bool von[NV_];
#pragma unroll
  for (int j = 0; j < NV_; ++j) {
    const int it = vt + j * nvt;
    von[j] = vt >= 0 && it < nitem;
    vk[j] = von[j] ? it / d : 0;
    vr[j] = von[j] ? it - vk[j] * d : 0;
  }

  auto params = [&](int round, int buf, int sweep_parity) {              // lanes < np of wave 0
    const int k = tid;
    int a, b;
    pair_of(round, k, m1, a, b);
    const double hpq = Hs[a * sd + b], hqq = Hs[b * sd + b], hpp = Hs[a * sd + a];
    jac_cs r = {1.0, 0.0};
    if (fabs(hpq) > thr) {
      r = jac_rotation(hpp, hqq, hpq, ih);
      atomicAdd(rot + sweep_parity, 1);
    }
    csn[buf * np + k] = r;
  };

  if (tid < np) params(0, 0, 1);                           // sweep 1 is odd
  __syncthreads();
  int sweep = 0, g = 0;
  bool done = false;
  while (sweep < max_sweeps && !done) {
    ++sweep;
    const int par = sweep & 1;
    if (tid == 0) rot[par ^ 1] = 0;                        // next sweep's counter: first incremented in this sweep's last stage B
    for (int round = 0; round < m1; ++round, ++g) {
      const jac_cs* cur = csn + (g & 1) * np;
      // ---- stage A: H <- J' H J, this thread's 2 x 2 blocks; all loads first (one level of LDS latency) ----
      {
        int o[NB_][4];
        double m[NB_][4];
        jac_cs ra[NB_], rb[NB_];
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            int p1, q1, p2, q2;
            pair_of(round, bka[j], m1, p1, q1);
            pair_of(round, bkb[j], m1, p2, q2);
            o[j][0] = p1 * sd + p2;
            o[j][1] = p1 * sd + q2;
            o[j][2] = q1 * sd + p2;
            o[j][3] = q1 * sd + q2;
            ra[j] = cur[bka[j]];
            rb[j] = cur[bkb[j]];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[j][e] = Hs[o[j][e]];
          }
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            const double ca = ra[j].x, sa = ra[j].y, cb = rb[j].x, sb = rb[j].y;
            const double n00 = cb * m[j][0] - sb * m[j][1], n01 = sb * m[j][0] + cb * m[j][1];
            const double n10 = cb * m[j][2] - sb * m[j][3], n11 = sb * m[j][2] + cb * m[j][3];
            double o00 = ca * n00 - sa * n10, o10 = sa * n00 + ca * n10;
            double o01 = ca * n01 - sa * n11, o11 = sa * n01 + ca * n11;
            if (bka[j] == bkb[j] && sa != 0.0) { o01 = 0.0; o10 = 0.0; }   // the rotated pair itself: annihilated by construction
            if (bon[j]) { Hs[o[j][0]] = o00; Hs[o[j][1]] = o01; Hs[o[j][2]] = o10; Hs[o[j][3]] = o11; }
          }
      }
      __syncthreads();
      // ---- stage B: wave 0 prepares round g + 1 from the new H; the other waves rotate the rows of V' ----
      if (tid < 64) {
        if (tid < np) {
          const bool last = round + 1 == m1;
          params(last ? 0 : round + 1, (g & 1) ^ 1, last ? (par ^ 1) : par);
        }
      } else {
        int op[NV_], oq[NV_];
        double x[NV_], y[NV_];
        jac_cs r[NV_];
#pragma unroll
        for (int j = 0; j < NV_; ++j) {
            int p, q;
            pair_of(round, vk[j], m1, p, q);
            op[j] = p * sd + vr[j];
            oq[j] = q * sd + vr[j];
            r[j] = cur[vk[j]];
            x[j] = Vs[op[j]];
            y[j] = Vs[oq[j]];
          }
#pragma unroll
        for (int j = 0; j < NV_; ++j)
          if (von[j]) { Vs[op[j]] = r[j].x * x[j] - r[j].y * y[j]; Vs[oq[j]] = r[j].y * x[j] + r[j].x * y[j]; }
      }
      __syncthreads();
    }
    done = (rot[par] == 0);
    __syncthreads();                                       // everyone has read the counter before thread 0 recycles it
  }
  for (int i = tid; i < d; i += nt) w[i] = Hs[i * sd + i];
  for (int i = tid; i < d * d; i += nt) {
    const int r = i / d, c = i - r * d;
    Vt[int64_t(r) * ldv + c] = Vs[r * sd + c];
  }
  if (tid == 0) *status = done ? sweep : -1;
}

__global__ __launch_bounds__(1024) void k_syev_small(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                     double* __restrict__ Vt, int64_t ldv, double tol, int max_sweeps,
                                                     int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  syev_small_body<2, 4>(jac_smem, d, A, lda, w, Vt, ldv, tol, max_sweeps, status);      // d <= 80
}
__global__ __launch_bounds__(1024) void k_syev_small_wide(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                          double* __restrict__ Vt, int64_t ldv, double tol, int max_sweeps,
                                                          int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  syev_small_body<3, 5>(jac_smem, d, A, lda, w, Vt, ldv, tol, max_sweeps, status);      // 80 < d <= 96
}

// ---------------------------------------------------------------------------
// The same eigen-solve split in two (default; d <= 160): the fused kernel above moves 306 KB through the LDS of ONE
// CU per round (H and V', each read + written, plus the rotation parameters) and is bound by exactly that.
//   k_syev_packed   one workgroup: H only, stored as its packed lower triangle (the mirrored 2 x 2 blocks are the same
//                   numbers: half the blocks, a quarter of the LDS bytes per round; 103 KB hold d = 160, which the
//                   square layout could not), every round's (c, s) pairs are logged to global memory;
//   k_jacobi_replay d / 16 workgroups: each replays the whole log on its own 16 columns of V' = I (columns of V' are
//                   independent), one barrier per round, parameters of the next round prefetched.
// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL address spaces:
// with a global store in flight (the rotation log) every wave would wait for its acknowledgement (~1 us) at every
// barrier of the round loop.  The log is consumed by a later kernel, so only this wave's LDS operations must have
// completed before the barrier.
// sync (may be null): two agent-scope words for the replaying workgroups of the SAME launch (k_syev_chase below) --
//   sync[0] = rounds whose parameters are complete in the log (moved every JR_CHUNK rounds), sync[1] = 1 + rounds to replay once the
//   solve has ended (any way it ends).  The log is stored write-through (st_shared2) so that another XCD's workgroup can read it.
constexpr int JR_CHUNK = 16;                                  // rounds of parameters staged through LDS at a time
template <int NB_>
__device__ __forceinline__ void syev_packed_body(char* smem, int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                 jac_cs* __restrict__ log, double tol, int max_sweeps, int* __restrict__ status,
                                                 unsigned* __restrict__ sync = nullptr) {
  const int pe = (d + 1) & ~1, m1 = pe - 1, np = pe >> 1;
  const int nH = pe * (pe + 1) / 2, nHp = (nH + 1) & ~1;
  double* Hs = reinterpret_cast<double*>(smem);
  double* red = Hs + nHp;
  int* rot = reinterpret_cast<int*>(red + 16);
  jac_cs* csn = reinterpret_cast<jac_cs*>(red + 18);       // [2][np]
  const int tid = threadIdx.x, nt = blockDim.x;
  double mx = 0.0;
  for (int e = tid; e < nH; e += nt) {
    int i, j;
    tri_decode(e, i, j);
    double h = 0.0;
    if (i < d) {                                           // j <= i
      h = 0.5 * (A[int64_t(i) * lda + j] + A[int64_t(j) * lda + i]);
      const double a = fabs(h);
      mx = (a <= 1.79769313486231570e308) ? fmax(mx, a) : __builtin_inf();
    }
    Hs[e] = h;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  if (tid < 2) rot[tid] = 0;
  __syncthreads();
  double hmax = 0.0;
  for (int i = 0; i < (nt + 63) >> 6; ++i) hmax = fmax(hmax, red[i]);
  if (!(hmax < __builtin_inf())) {
    if (tid == 0) {
      status[0] = -2; status[1] = 0;
      if (sync) __hip_atomic_store((gu32_ptr)(reinterpret_cast<uintptr_t>(sync + 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (hmax == 0.0) {
    for (int i = tid; i < d; i += nt) w[i] = 0.0;
    if (tid == 0) {
      status[0] = 1; status[1] = 0;
      if (sync) __hip_atomic_store((gu32_ptr)(reinterpret_cast<uintptr_t>(sync + 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const double thr = tol * hmax, ih = 1.0 / hmax;
  const int nblk = np * (np + 1) / 2;
  int bka[NB_], bkb[NB_];
  bool bon[NB_];
#pragma unroll
  for (int j = 0; j < NB_; ++j) {
    const int b = tid + j * nt;
    bon[j] = b < nblk;
    bka[j] = 0; bkb[j] = 0;
    if (bon[j]) tri_decode(b, bka[j], bkb[j]);            // ka >= kb
  }
  auto params = [&](int round, int buf, int sweep_parity, int glog) {   // lanes < np of wave 0
    const int k = tid;
    int a, b;
    pair_of(round, k, m1, a, b);
    const double hpq = Hs[tri_off(a, b)], hqq = Hs[tri_off(b, b)], hpp = Hs[tri_off(a, a)];
    jac_cs r = {1.0, 0.0};
    if (fabs(hpq) > thr) {
      r = jac_rotation(hpp, hqq, hpq, ih);
      atomicAdd(rot + sweep_parity, 1);
    }
    csn[buf * np + k] = r;
    st_shared2(reinterpret_cast<double*>(log + int64_t(glog) * np + k), r.x, r.y);
  };
  if (tid < np) params(0, 0, 1, 0);
  __syncthreads();
  int sweep = 0, g = 0;
  bool done = false;
  while (sweep < max_sweeps && !done) {
    ++sweep;
    const int par = sweep & 1;
    if (tid == 0) rot[par ^ 1] = 0;
    for (int round = 0; round < m1; ++round, ++g) {
      const jac_cs* cur = csn + (g & 1) * np;
      {
        int o[NB_][4];
        double m[NB_][4];
        jac_cs ra[NB_], rb[NB_];
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
          int p1, q1, p2, q2;
          pair_of(round, bka[j], m1, p1, q1);
          pair_of(round, bkb[j], m1, p2, q2);
          o[j][0] = tri_off(p1, p2);
          o[j][1] = tri_off(p1, q2);
          o[j][2] = tri_off(q1, p2);
          o[j][3] = tri_off(q1, q2);
          ra[j] = cur[bka[j]];
          rb[j] = cur[bkb[j]];
#pragma unroll
          for (int e = 0; e < 4; ++e) m[j][e] = Hs[o[j][e]];
        }
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
          const double ca = ra[j].x, sa = ra[j].y, cb = rb[j].x, sb = rb[j].y;
          const double n00 = cb * m[j][0] - sb * m[j][1], n01 = sb * m[j][0] + cb * m[j][1];
          const double n10 = cb * m[j][2] - sb * m[j][3], n11 = sb * m[j][2] + cb * m[j][3];
          double o00 = ca * n00 - sa * n10, o10 = sa * n00 + ca * n10;
          double o01 = ca * n01 - sa * n11, o11 = sa * n01 + ca * n11;
          if (bka[j] == bkb[j] && sa != 0.0) { o01 = 0.0; o10 = 0.0; }   // the rotated pair itself (one storage cell)
          if (bon[j]) { Hs[o[j][0]] = o00; Hs[o[j][1]] = o01; Hs[o[j][2]] = o10; Hs[o[j][3]] = o11; }
        }
      }
      lds_barrier();
      const bool publish = sync && ((g + 2) & (JR_CHUNK - 1)) == 0;      // rounds 0 .. g + 1 are in the log after this step
      if (tid < np) {
        const bool last = round + 1 == m1;
        params(last ? 0 : round + 1, (g & 1) ^ 1, last ? (par ^ 1) : par, g + 1);
        if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the writing lanes drain their write-through stores)
      }
      lds_barrier();
      if (publish && tid == 0)
        __hip_atomic_store((gu32_ptr)(reinterpret_cast<uintptr_t>(sync)), unsigned(g + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    done = (rot[par] == 0);
    lds_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < d; i += nt) w[i] = Hs[tri_off(i, i)];
  if (tid == 0) {
    status[0] = done ? sweep : -1; status[1] = g;
    // the last sweep of a converged run rotated nothing: the log's first (sweeps - 1) * m1 rounds are all of V
    if (sync) __hip_atomic_store((gu32_ptr)(reinterpret_cast<uintptr_t>(sync + 1)), 1u + unsigned(done ? (sweep - 1) * m1 : 0), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(1024) void k_syev_packed1(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                       jac_cs* __restrict__ log, double tol, int max_sweeps, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  syev_packed_body<1>(jac_smem, d, A, lda, w, log, tol, max_sweeps, status);            // d <= 88 (990 blocks)
}
__global__ __launch_bounds__(1024) void k_syev_packed2(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                       jac_cs* __restrict__ log, double tol, int max_sweeps, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  syev_packed_body<2>(jac_smem, d, A, lda, w, log, tol, max_sweeps, status);            // d <= 126
}
__global__ __launch_bounds__(1024) void k_syev_packed4(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                       jac_cs* __restrict__ log, double tol, int max_sweeps, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  syev_packed_body<4>(jac_smem, d, A, lda, w, log, tol, max_sweeps, status);            // d <= 160 (3240 blocks)
}

constexpr int JR_COLS = 16, JR_SLOTS = 16, JR_MAXK = 5;      // 256 threads: 16 columns x 16 pair slots, np <= 80
__global__ __launch_bounds__(256) void k_jacobi_replay(int d, const jac_cs* __restrict__ log, int rounds, double* __restrict__ Vt,
                                                       int64_t ldv) {
  __shared__ double Vs[160 * (JR_COLS + 1)];
  __shared__ jac_cs Ps[2][JR_CHUNK * 80];                     // two chunks of (c, s): one in use, one being filled
  const int pe = (d + 1) & ~1, m1 = pe - 1, np = pe >> 1;
  const int tid = threadIdx.x, col = tid & (JR_COLS - 1), slot = tid >> 4;
  const int col0 = blockIdx.x * JR_COLS;
  for (int e = tid; e < pe * JR_COLS; e += 256) {
    const int r = e >> 4, cc = e & 15;
    Vs[r * (JR_COLS + 1) + cc] = (r == col0 + cc && r < d) ? 1.0 : 0.0;
  }
  // the log is one contiguous array of rounds * np entries: a chunk is JR_CHUNK * np consecutive entries (<= 1280: 5 per thread)
  const int per_chunk = JR_CHUNK * np;
  const int64_t total = int64_t(rounds) * np;
  jac_cs stage[JR_MAXK];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < JR_MAXK; ++j) {
      const int e = tid + 256 * j;
      const int64_t gidx = int64_t(chunk) * per_chunk + e;
      stage[j] = (e < per_chunk && gidx < total) ? log[gidx] : jac_cs{1.0, 0.0};
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < JR_MAXK; ++j) {
      const int e = tid + 256 * j;
      if (e < per_chunk) Ps[buf][e] = stage[j];
    }
  };
  const int nchunks = (rounds + JR_CHUNK - 1) / JR_CHUNK;
  if (nchunks > 0) { fetch(0); stash(0); }
  __syncthreads();
  int round = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 1 < nchunks) fetch(ch + 1);                       // global latency hides behind this chunk's rounds
    const jac_cs* P = Ps[ch & 1];
    const int r_end = min(JR_CHUNK, rounds - ch * JR_CHUNK);
    for (int rr = 0; rr < r_end; ++rr) {
      int op[JR_MAXK], oq[JR_MAXK];
      double x[JR_MAXK], y[JR_MAXK];
      jac_cs cs[JR_MAXK];
#pragma unroll
      for (int j = 0; j < JR_MAXK; ++j) {
        const int k = min(slot + JR_SLOTS * j, np - 1);
        int p, q;
        pair_of(round, k, m1, p, q);
        op[j] = p * (JR_COLS + 1) + col;
        oq[j] = q * (JR_COLS + 1) + col;
        cs[j] = P[rr * np + k];
        x[j] = Vs[op[j]];
        y[j] = Vs[oq[j]];
      }
#pragma unroll
      for (int j = 0; j < JR_MAXK; ++j)
        if (slot + JR_SLOTS * j < np) {
          Vs[op[j]] = cs[j].x * x[j] - cs[j].y * y[j];
          Vs[oq[j]] = cs[j].y * x[j] + cs[j].x * y[j];
        }
      lds_barrier();
      if (++round == m1) round = 0;
    }
    if (ch + 1 < nchunks) stash((ch + 1) & 1);                 // nobody reads that buffer during this chunk
    __syncthreads();
  }
  if (col0 + col < d)
    for (int r = slot; r < d; r += JR_SLOTS) Vt[int64_t(r) * ldv + col0 + col] = Vs[r * (JR_COLS + 1) + col];
}

// ---------------------------------------------------------------------------
// Round 6: eigen-solve and replay as ONE launch (k_syev_chase).  The replay above starts when the solve has ended (and after a host
// round trip for the sweep count): 367 + 294 us behind the 800 + 661 us of the two Rayleigh-Ritz solves of an rCCA fit.  A round of
// the replay takes half the time of a round of the solve, so here the replaying workgroups run NEXT to the solving one and chase
// its log chunk by chunk: workgroup 0 = syev_packed_body (log stored write-through, a progress word every 16 rounds), workgroups
// 1 .. d / 16 = the replay on 16 columns of V' each (their first 256 threads; the rest retire at once), polling the progress word
// and reading the log past their XCD's L2 (ld_shared).  V' is complete one chunk (~15 us) after the solve.  Every exit of the
// solving workgroup publishes the end word; a poller that sees nothing for ~2 s gives up and reports -3.
// ---------------------------------------------------------------------------
constexpr int SC_WATCHDOG = 1 << 24;
__device__ __forceinline__ void replay_chase_body(char* smem, int d, const jac_cs* __restrict__ log, unsigned* __restrict__ sync,
                                                  double* __restrict__ Vt, int64_t ldv, int blk, int* __restrict__ status) {
  double* Vs = reinterpret_cast<double*>(smem);                                  // [pe][JR_COLS + 1]
  jac_cs* Ps = reinterpret_cast<jac_cs*>(Vs + 160 * (JR_COLS + 1));             // one chunk of (c, s)
  int* ctl = reinterpret_cast<int*>(Ps + JR_CHUNK * 80);                         // [0] rounds of this chunk, [1] 1 = last chunk, 2 = gave up
  const int pe = (d + 1) & ~1, m1 = pe - 1, np = pe >> 1;
  const int tid = threadIdx.x, col = tid & (JR_COLS - 1), slot = tid >> 4;
  const int col0 = blk * JR_COLS;
  for (int e = tid; e < pe * JR_COLS; e += 256) {
    const int r = e >> 4, cc = e & 15;
    Vs[r * (JR_COLS + 1) + cc] = (r == col0 + cc && r < d) ? 1.0 : 0.0;
  }
  const gu32_ptr wP = (gu32_ptr)(reinterpret_cast<uintptr_t>(sync)), wF = (gu32_ptr)(reinterpret_cast<uintptr_t>(sync + 1));
  int round = 0;
  bool gave_up = false;
  for (int ch = 0;; ++ch) {
    const int base = ch * JR_CHUNK;
    if (tid == 0) {
      int n = 0, last = 0, spins = 0;
      for (;;) {
        const unsigned F = __hip_atomic_load(wF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (F != 0u) {
          const int fin = int(F - 1u);
          n = max(0, min(JR_CHUNK, fin - base));
          last = base + JR_CHUNK >= fin ? 1 : 0;
          break;
        }
        if (int(__hip_atomic_load(wP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= base + JR_CHUNK) { n = JR_CHUNK; break; }
        __builtin_amdgcn_s_sleep(8);
        if (++spins > SC_WATCHDOG) { n = 0; last = 2; break; }
      }
      ctl[0] = n;
      ctl[1] = last;
    }
    __syncthreads();
    const int n = ctl[0], last = ctl[1];
    for (int e = tid; e < n * np; e += 256) {
      const double* src = reinterpret_cast<const double*>(log + int64_t(base) * np + e);
      jac_cs r;
      r.x = ld_shared(src);
      r.y = ld_shared(src + 1);
      Ps[e] = r;
    }
    __syncthreads();
    for (int rr = 0; rr < n; ++rr) {
      int op[JR_MAXK], oq[JR_MAXK];
      double x[JR_MAXK], y[JR_MAXK];
      jac_cs cs[JR_MAXK];
#pragma unroll
      for (int j = 0; j < JR_MAXK; ++j) {
        const int k = min(slot + JR_SLOTS * j, np - 1);
        int p, q;
        pair_of(round, k, m1, p, q);
        op[j] = p * (JR_COLS + 1) + col;
        oq[j] = q * (JR_COLS + 1) + col;
        cs[j] = Ps[rr * np + k];
        x[j] = Vs[op[j]];
        y[j] = Vs[oq[j]];
      }
#pragma unroll
      for (int j = 0; j < JR_MAXK; ++j)
        if (slot + JR_SLOTS * j < np) {
          Vs[op[j]] = cs[j].x * x[j] - cs[j].y * y[j];
          Vs[oq[j]] = cs[j].y * x[j] + cs[j].x * y[j];
        }
      lds_barrier();
      if (++round == m1) round = 0;
    }
    __syncthreads();                                           // (ctl and Ps are rewritten by the next chunk)
    if (last) { gave_up = last == 2; break; }
  }
  if (gave_up) { if (tid == 0) status[0] = -3; return; }
  if (col0 + col < d)
    for (int r = slot; r < d; r += JR_SLOTS) Vt[int64_t(r) * ldv + col0 + col] = Vs[r * (JR_COLS + 1) + col];
}

template <int NB_>
__global__ __launch_bounds__(1024) void k_syev_chase(int d, const double* __restrict__ A, int64_t lda, double* __restrict__ w,
                                                     jac_cs* __restrict__ log, double tol, int max_sweeps, int* __restrict__ status,
                                                     unsigned* __restrict__ sync, double* __restrict__ Vt, int64_t ldv) {
  extern __shared__ __attribute__((aligned(16))) char jac_smem[];
  if (blockIdx.x == 0) {
    syev_packed_body<NB_>(jac_smem, d, A, lda, w, log, tol, max_sweeps, status, sync);
    return;
  }
  if (threadIdx.x >= 256) return;                              // (retired waves do not take part in the barriers of the rest)
  replay_chase_body(jac_smem, d, log, sync, Vt, ldv, int(blockIdx.x) - 1, status);
}

static size_t syev_small_lds(int64_t d) {
  const int64_t pe = (d + 1) & ~int64_t(1), np = pe / 2, sd = pe | 1;
  return size_t(2 * pe * sd + 18 + 4 * np) * 8;          // H, V', 16 maxima, 2 counters (+ pad), 2 x np (c, s)
}

static int syev_mode() {   // 0: one-sided rows (round 1), 1: fused two-sided LDS kernel (d <= 96), 2: packed H + replayed V' (d <= 160)
  static const int m = [] { const char* e = getenv("CCZ_SYEV_TWOSIDED"); return e ? atoi(e) : 2; }();
  return m;
}

int syev_small_max(ccz_ctx*) { return syev_mode() == 2 ? 160 : (syev_mode() == 1 ? 96 : 0); }

int syev_small(ccz_ctx* c, const double* A, int64_t d, int64_t lda, double* w_dev, double* Vrows, int64_t ldv, int max_sweeps, double tol) {
  const int dmax = syev_small_max(c);
  if (d < 1 || d > dmax) fail(CCZ_EINVAL, "syev_small: 1 <= d <= %d required, got %lld", dmax, (long long)d);
  Impl* im = impl(c);
  int sw = 0;
  if (syev_mode() == 2) {
    const int64_t pe = (d + 1) & ~int64_t(1), m1 = pe - 1, np = pe / 2;
    const int64_t nH = pe * (pe + 1) / 2, nblk = np * (np + 1) / 2;
    const size_t lds_need = size_t(((nH + 1) & ~int64_t(1)) + 18 + 4 * np) * 8;
    DBuf logb(c, (int64_t(max_sweeps) * m1 + 1) * np * 2);
    jac_cs* log = reinterpret_cast<jac_cs*>(logb.get());
    auto launch = [&](auto kern) {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need)));
      hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds_need, stream(c), int(d), A, lda, w_dev, log, tol,
                         max_sweeps, im->d_flag + 2);
    };
    static const int chase_on = [] { const char* e = getenv("CCZ_SYEV_CHASE"); return e ? atoi(e) : 1; }();
    if (chase_on && Vrows) {
      // one launch: the solve and, next to it, the replay chasing its log (k_syev_chase)
      unsigned* sync = reinterpret_cast<unsigned*>(im->d_flag + 32);      // (words 8 .. 15 are the pivot flags of the factorizations)
      CCZ_HIP(hipMemsetAsync(sync, 0, 2 * sizeof(unsigned), stream(c)));
      const size_t lds_chase = std::max(lds_need, size_t(160 * (JR_COLS + 1) * 8 + JR_CHUNK * 80 * 16 + 16));
      const dim3 grid(1u + unsigned((d + JR_COLS - 1) / JR_COLS));
      auto launch2 = [&](auto kern) {
        CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_chase)));
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds_chase, stream(c), int(d), A, lda, w_dev, log, tol, max_sweeps,
                           im->d_flag + 2, sync, Vrows, ldv);
      };
      if (nblk <= 1024) launch2(&k_syev_chase<1>);
      else if (nblk <= 2048) launch2(&k_syev_chase<2>);
      else launch2(&k_syev_chase<4>);
      CCZ_LAUNCH_CHECK();
      int st[2] = {0, 0};
      d2h(c, st, im->d_flag + 2, sizeof(st));
      sw = st[0];
      if (sw == -3) fail(CCZ_EHIP, "syev: the replaying workgroups lost the solving one");
      if (sw == -2) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
      if (sw < 0) fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps (d=%lld)", max_sweeps, (long long)d);
      return sw;
    }
    if (nblk <= 1024) launch(&k_syev_packed1);
    else if (nblk <= 2048) launch(&k_syev_packed2);
    else launch(&k_syev_packed4);
    CCZ_LAUNCH_CHECK();
    int st[2] = {0, 0};
    d2h(c, st, im->d_flag + 2, sizeof(st));
    sw = st[0];
    if (sw >= 1 && Vrows) {
      // the last sweep of a converged run rotated nothing: the log's first (sweeps - 1) * m1 rounds are all of V
      const int rounds = int((sw - 1) * m1);
      hipLaunchKernelGGL(k_jacobi_replay, dim3((unsigned)((d + JR_COLS - 1) / JR_COLS)), dim3(256), 0, stream(c), int(d), log, rounds,
                         Vrows, ldv);
      CCZ_LAUNCH_CHECK();     // (the log goes back to the handle's pool: reuse is ordered behind this launch on the same stream)
    }
  } else {
    const size_t lds_need = syev_small_lds(d);
    if (d <= 80) {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_syev_small), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need)));
      hipLaunchKernelGGL(k_syev_small, dim3(1), dim3(1024), lds_need, stream(c), int(d), A, lda, w_dev, Vrows, ldv,
                         tol, max_sweeps, im->d_flag + 1);
    } else {
      CCZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_syev_small_wide), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need)));
      hipLaunchKernelGGL(k_syev_small_wide, dim3(1), dim3(1024), lds_need, stream(c), int(d), A, lda, w_dev, Vrows, ldv,
                         tol, max_sweeps, im->d_flag + 1);
    }
    CCZ_LAUNCH_CHECK();
    d2h(c, &sw, im->d_flag + 1, sizeof(int));
  }
  if (sw == -2) fail(CCZ_EINVAL, "syev: matrix has non-finite entries");
  if (sw < 0) fail(CCZ_ENOCONV, "Jacobi did not converge in %d sweeps (d=%lld)", max_sweeps, (long long)d);
  return sw;
}

}  // namespace ccz
