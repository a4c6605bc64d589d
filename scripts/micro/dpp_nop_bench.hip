// Microbenchmark: what do the s_nop wait states of a DPP reduction chain cost when 8 waves share a SIMD?
// Variants: 0 = chain with "s_nop 1" between dependent DPP ops (the shipped form), 1 = no nops (results wrong, timing
// only), 2 = two independent chains interleaved with "s_nop 0" (same wait states, half the nops per reduction).
#include <hip/hip_runtime.h>
#include <cstdio>
#define STEP(OP, CTRL) OP " %0, %0, %0 " CTRL "\n\t"
#define NOP1 "s_nop 1\n\t"
template <int V>
__global__ __launch_bounds__(64) void k(unsigned* out, int iters) {
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u, acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (V == 0) {
      asm volatile(NOP1 STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   NOP1 STEP("v_min_u32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
                   NOP1 STEP("v_min_u32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1" : "+v"(x));
      asm volatile(NOP1 STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   NOP1 STEP("v_min_u32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
                   NOP1 STEP("v_min_u32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") NOP1 STEP("v_min_u32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1" : "+v"(y));
    } else if (V == 1) {
      asm volatile(STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   STEP("v_min_u32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf") STEP("v_min_u32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
                   STEP("v_min_u32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("v_min_u32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") : "+v"(x));
      asm volatile(STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   STEP("v_min_u32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf") STEP("v_min_u32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
                   STEP("v_min_u32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("v_min_u32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") : "+v"(y));
    } else {
#define S2(CTRL) "v_min_u32_dpp %0, %0, %0 " CTRL "\n\tv_min_u32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
      asm volatile("s_nop 1\n\t" S2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") S2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   S2("row_half_mirror row_mask:0xf bank_mask:0xf") S2("row_mirror row_mask:0xf bank_mask:0xf")
                   S2("row_bcast:15 row_mask:0xa bank_mask:0xf") S2("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0" : "+v"(x), "+v"(y));
    }
    acc += __builtin_amdgcn_readlane(x, 63) + __builtin_amdgcn_readlane(y, 63);
    x += acc; y ^= acc;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
template <int V> float run(unsigned* d, int blocks, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  const int blocks = 256 * 32, iters = 20000;  // 32 one-wave workgroups per CU = 8 waves per SIMD
  unsigned* d; hipMalloc(&d, blocks * 4);
  for (int rep = 0; rep < 2; ++rep)
    printf("2 reductions x %d iters x %d waves: nop1 %.2f ms | no nops (wrong results) %.2f ms | 2 chains interleaved + s_nop 0 %.2f ms\n", iters, blocks,
           run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters));
  return 0;
}
