// host_math.hpp - the host-side integer restatements of the reference's RNG-dependent comparisons (plain C++17, no
// HIP): parameters of the device kernels are derived with these, tests/test_host_math.py checks them on the CPU against
// the floating-point definitions they replace.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------------------
// Jump-ahead of Xoroshiro128StarStar (rand_xoshiro 0.6.0).  The state transition T is linear over
// GF(2) on the 128 state bits, so T^n is a 128 x 128 bit matrix; k_agents_wave applies it to a lane's state as the XOR
// of 32 table entries, one per state nibble: entry (k, v) = T^n applied to the state whose only non-zero nibble is
// nibble k = v.  Layout: 32 tables x 16 entries x 128 bit (little-endian words s0_lo, s0_hi, s1_lo, s1_hi) = 8 KB.
inline void xoroshiro_step(uint64_t& s0, uint64_t& s1) {
  const uint64_t t = s1 ^ s0;
  s0 = ((s0 << 24) | (s0 >> 40)) ^ t ^ (t << 16);
  s1 = (t << 37) | (t >> 27);
}
constexpr int JUMP_TABLE_WORDS = 32 * 16 * 4;  // u32 words of one table set
inline std::vector<uint32_t> xoroshiro_jump_table(uint64_t n_steps) {
  uint64_t col0[128], col1[128];  // T^n of every basis bit
  for (int b = 0; b < 128; ++b) {
    uint64_t s0 = b < 64 ? (1ull << b) : 0ull, s1 = b < 64 ? 0ull : (1ull << (b - 64));
    for (uint64_t i = 0; i < n_steps; ++i) xoroshiro_step(s0, s1);
    col0[b] = s0;
    col1[b] = s1;
  }
  std::vector<uint32_t> tab(JUMP_TABLE_WORDS);
  for (int k = 0; k < 32; ++k)
    for (int v = 0; v < 16; ++v) {
      uint64_t a0 = 0, a1 = 0;
      for (int bit = 0; bit < 4; ++bit)
        if ((v >> bit) & 1) {
          a0 ^= col0[4 * k + bit];
          a1 ^= col1[4 * k + bit];
        }
      uint32_t* e = tab.data() + (k * 16 + v) * 4;
      e[0] = static_cast<uint32_t>(a0);
      e[1] = static_cast<uint32_t>(a0 >> 32);
      e[2] = static_cast<uint32_t>(a1);
      e[3] = static_cast<uint32_t>(a1 >> 32);
    }
  return tab;
}
// Division of any u32 by the invariant d >= 1 as multiply-high + shifts (Granlund & Montgomery, "Division by invariant
// integers using multiplication", round-up method): L = ceil(log2 d), m = floor(2^32 (2^L - d) / d) + 1,
// q = (t + ((n - t) >> min(L, 1))) >> max(L - 1, 0) with t = umulhi(m, n).  Plain integers: the device mirror is
// book_device.hpp udiv().
struct HostUDiv {
  uint32_t m, sh1, sh2, d;
};
inline HostUDiv make_udiv(uint32_t d) {
  uint32_t L = 0;
  while (L < 32 && (1ull << L) < d) ++L;
  const uint64_t m = ((1ull << 32) * ((1ull << L) - d)) / d + 1ull;
  return HostUDiv{static_cast<uint32_t>(m), L < 1u ? L : 1u, L > 0u ? L - 1u : 0u, d};
}
inline uint32_t udiv_apply(uint32_t n, const HostUDiv& dv) {
  const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(dv.m) * n) >> 32);
  return (t + ((n - t) >> dv.sh1)) >> dv.sh2;
}

// the same lookup the device performs (host mirror, used by the CPU test)
inline void xoroshiro_jump_apply(const uint32_t* tab, uint64_t& s0, uint64_t& s1) {
  const uint32_t w[4] = {static_cast<uint32_t>(s0), static_cast<uint32_t>(s0 >> 32), static_cast<uint32_t>(s1),
                         static_cast<uint32_t>(s1 >> 32)};
  uint32_t acc[4] = {0, 0, 0, 0};
  for (int d = 0; d < 4; ++d)
    for (int k = 0; k < 8; ++k) {
      const uint32_t* e = tab + ((d * 8 + k) * 16 + ((w[d] >> (4 * k)) & 15u)) * 4;
      for (int i = 0; i < 4; ++i) acc[i] ^= e[i];
    }
  s0 = (static_cast<uint64_t>(acc[1]) << 32) | acc[0];
  s1 = (static_cast<uint64_t>(acc[3]) << 32) | acc[2];
}

}  // namespace bkd
