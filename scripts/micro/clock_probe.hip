// What does s_memtime tick at, and what is the shader clock under load?  One probe wave spins for ~2 ms of
// s_memrealtime (a constant 100 MHz counter) and reports (a) s_memtime ticks and (b) how many dependent s_add it issued
// (one per 4 clocks... measured below relative to the idle case) - once on an idle GPU, once while a full-chip VALU+SALU
// load runs on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out) {
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), m0 = __builtin_amdgcn_s_memtime();
  unsigned n = 0, a = 1;
  while (__builtin_amdgcn_s_memrealtime() - r0 < 200000ull) {  // 2 ms at 100 MHz
    asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
                 "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t" : "+s"(a) : : "scc");
    ++n;
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), m1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = m1 - m0; out[2] = n; out[3] = a; }
}
__global__ void load(unsigned* sink, int iters) {
  unsigned a = blockIdx.x, v = threadIdx.x, w = 3;
  for (int i = 0; i < iters; ++i)
    asm volatile("s_add_u32 %0, %0, 1\n\tv_add_u32 %1, 1, %1\n\tv_mul_lo_u32 %2, %2, %1\n\ts_add_u32 %0, %0, 1\n\tv_add_u32 %1, 1, %1\n\tv_mul_lo_u32 %2, %2, %1\n\t"
                 : "+s"(a), "+v"(v), "+v"(w) : : "scc");
  if (a + v + w == 0x1234567u) sink[0] = a;
}
int main() {
  unsigned long long* d; unsigned* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 64);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  unsigned long long h[4];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s1, d); hipStreamSynchronize(s1);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("idle  : %llu realtime ticks (100 MHz) = %.3f ms, s_memtime ticks %llu -> %.1f MHz, loop iterations %llu (%.1f s_memtime ticks each)\n", h[0], h[0] / 1e5,
           h[1], h[1] / (h[0] / 100.0), h[2], (double)h[1] / h[2]);
    hipLaunchKernelGGL(load, dim3(256 * 28), dim3(64), 0, s2, sink, 400000);  // ~7 waves per SIMD of mixed issue for a few ms
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s1, d); hipStreamSynchronize(s1);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("loaded: %llu realtime ticks = %.3f ms, s_memtime ticks %llu -> %.1f MHz, loop iterations %llu (%.1f s_memtime ticks each)\n", h[0], h[0] / 1e5, h[1],
           h[1] / (h[0] / 100.0), h[2], (double)h[1] / h[2]);
    hipDeviceSynchronize();
  }
  return 0;
}
