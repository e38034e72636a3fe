// Counter-based normal generator for the at-scale measurement inputs (JointData.sample_device).
//
// reference: cca_zoo/datasets/_simulated.py:116-130 draws z and one noise block per view from a host NumPy stream;
// 32.8 GB (north-star shape) cannot be generated or held on the host, so the views are drawn straight into HBM.
// Every element is a pure function of (seed, global row, column) -- rng_hash.h::hash_normal_pair, restated in NumPy
// in oracle/rng.py -- so ANY row range of ANY shard can be regenerated on the host for a parity spot check, and the
// data set does not depend on how many GPUs it is sharded over.
#include "hip_common.h"
#include "rng_hash.h"

namespace ccz {

// out[r][c] = (accumulate ? out[r][c] : 0) + scale * N(seed, (row0 + r) * row_stride + c);  row_stride even, >= cols.
// A thread produces the two normals of one Box-Muller pair (columns 2q, 2q + 1).
template <typename T>
__global__ void k_randn_fill(T* __restrict__ out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, int64_t row0,
                             int64_t row_stride, double scale, int accumulate) {
  const int64_t half = (cols + 1) / 2;
  const int64_t total = rows * half;
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
    const int64_t r = e / half, q = e - r * half;
    const uint64_t pair = (uint64_t(row0 + r) * uint64_t(row_stride)) / 2 + uint64_t(q);
    double n0, n1;
    hash_normal_pair(seed, pair, n0, n1);
    T* p = out + r * ld + 2 * q;
    p[0] = T((accumulate ? double(p[0]) : 0.0) + scale * n0);
    if (2 * q + 1 < cols) p[1] = T((accumulate ? double(p[1]) : 0.0) + scale * n1);
  }
}

void randn_fill_impl(ccz_ctx* c, int dtype, void* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed, int64_t row0,
                     int64_t row_stride, double scale, bool accumulate) {
  if (dtype != CCZ_F32 && dtype != CCZ_F64) fail(CCZ_EUNSUP, "randn_fill: dtype must be CCZ_F32 or CCZ_F64");
  if (!out || rows < 0 || cols < 1 || ld < cols || row0 < 0) fail(CCZ_EINVAL, "randn_fill: bad argument");
  if (row_stride < cols || (row_stride & 1)) fail(CCZ_EINVAL, "randn_fill: row_stride must be even and >= cols");
  if (rows == 0) return;
  const int64_t total = rows * ((cols + 1) / 2);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, int64_t(1) << 20);
  if (dtype == CCZ_F32)
    hipLaunchKernelGGL(k_randn_fill<float>, dim3(grid), dim3(256), 0, stream(c), static_cast<float*>(out), rows, cols, ld, seed, row0,
                       row_stride, scale, accumulate ? 1 : 0);
  else
    hipLaunchKernelGGL(k_randn_fill<double>, dim3(grid), dim3(256), 0, stream(c), static_cast<double*>(out), rows, cols, ld, seed, row0,
                       row_stride, scale, accumulate ? 1 : 0);
  CCZ_LAUNCH_CHECK();
}

}  // namespace ccz
