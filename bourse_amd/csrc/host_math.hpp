// host_math.hpp - the host-side integer restatements of the reference's RNG-dependent comparisons (plain C++17, no
// HIP): parameters of the device kernels are derived with these, tests/test_host_math.py checks them on the CPU against
// the floating-point definitions they replace.
#pragma once
#include <cmath>
#include <cstdint>

namespace bkd {

// rand_xoshiro 0.6.0 `seed_from_u64` for Xoroshiro128StarStar: two SplitMix64 outputs
inline void seed_from_u64(uint64_t seed, uint64_t& s0, uint64_t& s1) {
  uint64_t x = seed;
  auto next = [&x]() {
    x += 0x9e3779b97f4a7c15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  s0 = next();
  s1 = next();
}

// `gen::<f32>() < rate` with gen = (u32 >> 8) * 2^-24 (App. B.5) as an integer threshold on k = (u32 >> 8):
// k * 2^-24 < rate  <=>  k < rate * 2^24 (exact in double)  <=>  k < ceil(rate * 2^24).
inline uint32_t activity_threshold(float rate) {
  if (!(rate > 0.0f)) return 0;  // also NaN
  const double x = static_cast<double>(rate) * 16777216.0;
  if (x >= 16777216.0) return 16777216u;
  return static_cast<uint32_t>(std::ceil(x));
}

// cancel_live_orders keeps an order iff `gen::<f32>() > p_cancel` (common.rs:68): k * 2^-24 > p  <=>  k > floor(p * 2^24).
// Returned as a signed bound: keep iff (int32)k > keep_threshold(p).
inline int32_t keep_threshold(float p_cancel) {
  if (p_cancel != p_cancel) return 1 << 24;  // NaN: never kept
  const double y = std::floor(static_cast<double>(p_cancel) * 16777216.0);
  return y < 0.0 ? -1 : (y > 16777216.0 ? (1 << 24) : static_cast<int32_t>(y));
}

// UniformInt<u32>::sample_single's acceptance zone for `range` (App. B.3): accept iff lo32(x * range) <= zone
inline uint32_t sample_zone(uint32_t range) { return (range << __builtin_clz(range)) - 1u; }

}  // namespace bkd
