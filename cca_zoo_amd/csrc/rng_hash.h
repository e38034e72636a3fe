// Counter-based pseudo-normal generator shared by every backend of ops.h
// (device kernels and the host test double produce identical streams).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CCZ_HD __host__ __device__
#else
#define CCZ_HD
#endif

namespace ccz {

CCZ_HD inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// standard normal for (seed, index): Box-Muller on two hashed uniforms
CCZ_HD inline double hash_normal(uint64_t seed, uint64_t index) {
  const uint64_t a = splitmix64(seed ^ splitmix64(2 * index));
  const uint64_t b = splitmix64(seed ^ splitmix64(2 * index + 1));
  const double u1 = (double((a >> 11) + 1)) * (1.0 / 9007199254740993.0);  // (0, 1)
  const double u2 = double(b >> 11) * (1.0 / 9007199254740992.0);          // [0, 1)
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// the two normals of Box-Muller pair `pair` of stream `seed` (element 2 pair -> n0, element 2 pair + 1 -> n1);
// restated in NumPy in oracle/rng.py (the measurement inputs of bench.py can be regenerated on the host)
CCZ_HD inline void hash_normal_pair(uint64_t seed, uint64_t pair, double& n0, double& n1) {
  const uint64_t a = splitmix64(seed ^ splitmix64(2 * pair));
  const uint64_t b = splitmix64(seed ^ splitmix64(2 * pair + 1));
  const double u1 = (double((a >> 11) + 1)) * (1.0 / 9007199254740992.0);  // (0, 1]
  const double u2 = double(b >> 11) * (1.0 / 9007199254740992.0);          // [0, 1)
  const double r = sqrt(-2.0 * log(u1));
  const double t = 6.283185307179586 * u2;
  n0 = r * cos(t);
  n1 = r * sin(t);
}

}  // namespace ccz
