// Does s_nop occupy the scalar issue port?  8 waves per SIMD, each loop iteration issues 64 instructions of one kind:
// (0) s_nop 0, (1) s_nop 1, (2) independent s_add_u32 on 4 registers, (3) v_add_u32 (VALU reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
template <int V>
__global__ __launch_bounds__(64) void k(unsigned* out, int iters) {
  unsigned a = blockIdx.x, b = 1, c = 2, d = 3, v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(R64("s_nop 0\n\t"));
    if (V == 1) asm volatile(R64("s_nop 1\n\t"));
    if (V == 2) asm volatile(R16("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t") : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    if (V == 3) asm volatile(R64("v_add_u32 %0, 1, %0\n\t") : "+v"(v));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = a + b + c + d + v;
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
  const int blocks = 256 * 32, iters = 4000;
  unsigned* d; hipMalloc(&d, blocks * 4);
  for (int rep = 0; rep < 2; ++rep)
    printf("64 x %d instr per wave, 32 waves/CU: s_nop 0 %.2f ms | s_nop 1 %.2f ms | s_add_u32 %.2f ms | v_add_u32 %.2f ms\n", iters,
           run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters));
  // per CU: 32 waves x 64 x iters instructions; at 1 instr/clk/CU and 2.4 GHz: 
  printf("1 instr/clk/CU would be %.2f ms\n", 32.0 * 64 * iters / 2.4e9 * 1e3);
  return 0;
}
