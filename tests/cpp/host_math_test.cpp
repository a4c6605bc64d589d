// CPU test of bourse_amd/csrc/host_math.hpp: the integer thresholds must decide exactly like the reference's f32
// comparisons for EVERY k near the boundary (and a random sample elsewhere); SplitMix64 seeding against the oracle.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../bourse_amd/csrc/host_math.hpp"

static float gen_f32(uint32_t k) { return static_cast<float>(k) * (1.0f / 16777216.0f); }  // rand 0.8.5 Standard f32, k = u32 >> 8

int main(int argc, char** argv) {
  std::mt19937_64 rng(7);
  const float special[] = {0.0f, -0.0f, 1.0f, 0.5f, 0.8f, 0.2f, 1e-9f, 0.99999994f, 1.0000001f, 2.0f, -1.0f, 5.9604645e-08f,
                           std::nanf(""), INFINITY, -INFINITY, 0.1f, 0.3f, 0.7f};
  long checked = 0;
  auto check_rate = [&](float rate) -> int {
    const uint32_t thr = bkd::activity_threshold(rate);
    const int32_t keep = bkd::keep_threshold(rate);
    auto one = [&](uint32_t k) -> int {
      const bool want_hit = gen_f32(k) < rate;            // random_agent.rs:91-93
      const bool want_keep = gen_f32(k) > rate;           // common.rs:68
      if ((k < thr) != want_hit) { std::printf("activity: rate %.9g k %u\n", rate, k); return 1; }
      if ((static_cast<int32_t>(k) > keep) != want_keep) { std::printf("keep: p %.9g k %u\n", rate, k); return 1; }
      ++checked;
      return 0;
    };
    for (uint32_t k : {0u, 1u, 2u, 16777214u, 16777215u})
      if (one(k)) return 1;
    for (int d = -3; d <= 3; ++d) {
      const int64_t a = static_cast<int64_t>(thr) + d, b = static_cast<int64_t>(keep) + d;
      if (a >= 0 && a < 16777216 && one(static_cast<uint32_t>(a))) return 1;
      if (b >= 0 && b < 16777216 && one(static_cast<uint32_t>(b))) return 1;
    }
    for (int i = 0; i < 8; ++i)
      if (one(static_cast<uint32_t>(rng() >> 40))) return 1;
    return 0;
  };
  for (float r : special)
    if (check_rate(r)) return 1;
  for (int i = 0; i < 200000; ++i) {
    uint32_t bits = static_cast<uint32_t>(rng());
    if (i % 2) bits = (bits & 0x007FFFFFu) | ((100u + (bits >> 23) % 28u) << 23);  // concentrate on (2^-27, 2)
    float r;
    std::memcpy(&r, &bits, 4);
    if (check_rate(r)) return 1;
  }
  // sample_zone: accept iff lo32(x * range) <= zone must leave exactly floor(2^32 / range) * range accepted values...
  // checked through its defining property: zone + 1 is the largest multiple pattern (range << lz) - 1
  for (uint32_t range : {1u, 2u, 3u, 10u, 16u, 20u, 32u, 64u, 90u, 1000u, 0x80000000u, 0xFFFFFFFFu}) {
    const uint32_t z = bkd::sample_zone(range);
    const int lz = __builtin_clz(range);
    if (z != ((range << lz) - 1u)) return 2;
    if (z < range - 1u) return 2;  // at least one full period of residues is accepted
  }
  // jump tables of the wave-parallel RNG decode: T^n by table == n single steps, for the block jump (256), the lane
  // offsets (4 << b) and a few odd counts, on random states; and the published xoroshiro128** vector for the step itself
  {
    uint64_t a0 = 1, a1 = 2;  // outputs 5760, 97769243520, ... from state (1, 2)
    const uint64_t r0 = ((a0 * 5) << 7 | (a0 * 5) >> 57) * 9;
    if (r0 != 5760ull) return 4;
    bkd::xoroshiro_step(a0, a1);
    const uint64_t x = a0 * 5, r1 = ((x << 7) | (x >> 57)) * 9;
    if (r1 != 97769243520ull) return 4;
    for (uint64_t n : {0ull, 1ull, 4ull, 8ull, 16ull, 32ull, 64ull, 128ull, 256ull, 1000ull}) {
      const std::vector<uint32_t> tab = bkd::xoroshiro_jump_table(n);
      for (int rep = 0; rep < 50; ++rep) {
        uint64_t s0 = rng(), s1 = rng();
        if (rep == 0) { s0 = 0; s1 = 0; }
        if (rep == 1) { s0 = ~0ull; s1 = ~0ull; }
        uint64_t j0 = s0, j1 = s1, t0 = s0, t1 = s1;
        bkd::xoroshiro_jump_apply(tab.data(), j0, j1);
        for (uint64_t i = 0; i < n; ++i) bkd::xoroshiro_step(t0, t1);
        if (j0 != t0 || j1 != t1) { std::printf("jump table n=%llu mismatch\n", (unsigned long long)n); return 4; }
      }
    }
  }
  // make_udiv: multiply-high division == `/` for every divisor class (1, powers of two, odd, near 2^31 / 2^32) against
  // boundary and random numerators
  {
    std::vector<uint32_t> ds = {1u, 2u, 3u, 4u, 5u, 6u, 7u, 10u, 12u, 100u, 641u, 65535u, 65536u, 65537u, 0x7FFFFFFFu, 0x80000000u,
                                0x80000001u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    for (int i = 0; i < 2000; ++i) ds.push_back(static_cast<uint32_t>(rng() >> (rng() % 32)) | 1u);
    for (uint32_t d : ds) {
      const bkd::HostUDiv dv = bkd::make_udiv(d);
      std::vector<uint32_t> ns = {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu};
      for (uint64_t k : {1ull, 2ull, 3ull, 1000ull})
        for (int o = -1; o <= 1; ++o) ns.push_back(static_cast<uint32_t>(k * d + o));
      for (int i = 0; i < 200; ++i) ns.push_back(static_cast<uint32_t>(rng()));
      for (uint32_t n : ns)
        if (bkd::udiv_apply(n, dv) != n / d) { std::printf("udiv: %u / %u\n", n, d); return 5; }
    }
  }
  uint64_t s0, s1;
  bkd::seed_from_u64(101, s0, s1);
  if (argc > 1) std::printf("%llu %llu\n", (unsigned long long)s0, (unsigned long long)s1);
  std::printf("host_math ok (%ld comparisons)\n", checked);
  return 0;
}
