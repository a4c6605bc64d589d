// What does a CU issue per clock when scalar and vector instructions are MIXED in every wave's stream?
// (salu_nop_bench.hip measured the pure streams: s_add_u32 0.95, v_add_u32 1.7 per CU and clock.  The step kernels are
// ~1:1 mixes of both plus branches and v_readlane / v_writelane hand-overs: is their ceiling the two ports side by
// side - 2.65 - or something lower?)  W waves per SIMD, 64-instruction bodies of INDEPENDENT instructions, shader
// clocks from s_memtime around the loop of every wave; rate = waves per CU x instructions / mean wave clocks.
//   0 s_add only            1 v_add only             2 s,v alternating (1:1)      3 s,s,v (2:1)        4 s,v,v (1:2)
//   5 1:1 + an untaken s_cbranch every 8       6 v_readlane -> s_add on its result -> v_add (hand-over chain, per 3)
//   7 1:1 with a TAKEN s_branch every 16       8 v_add, v_writelane (m0 select), s_add (per 3)
//   9 s_add, s_nop 0 alternating (is a wait state an issue slot of the mix?)
#include <hip/hip_runtime.h>
#include <cstdio>
#define R2(x) x x
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R32(x) R16(x) R16(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define SA "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\t"
#define S1 "s_add_u32 %0, %0, 1\n\t"
#define S2 "s_add_u32 %1, %1, 1\n\t"
#define V1 "v_add_u32 %2, 1, %2\n\t"
#define V2 "v_add_u32 %3, 1, %3\n\t"
template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long* out, unsigned* sink, int iters) {
  unsigned a = blockIdx.x, b = 1, v = threadIdx.x, w = 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(R32(S1 S2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");
    if (V == 1) asm volatile(R32(V1 V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");
    if (V == 2) asm volatile(R16(S1 V1 S2 V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");
    if (V == 3) asm volatile(R16(S1 S2 V1) R16(S2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");      // 48 s + 16 v
    if (V == 4) asm volatile(R16(S1 V1 V2) R16(V1) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");      // 16 s + 48 v
    if (V == 5) asm volatile(R8(S1 V1 S2 V2 S1 V1 V2 "s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t") "1:\n\t"
                             : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");                             // 8 x 9 = 72
    if (V == 6) asm volatile(R16("v_readlane_b32 s40, %2, 3\n\ts_add_u32 %0, %0, s40\n\tv_add_u32 %2, %0, %2\n\t")
                             : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40");                      // 48
    if (V == 7) asm volatile(R4(S1 V1 S2 V2 S1 V1 S2 V2 S1 V1 S2 V2 S1 V1 V2 "s_branch 2f\n\tv_add_u32 %3, 1, %3\n\t2:\n\t")
                             : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");                             // 4 x 16 = 64
    if (V == 8) asm volatile("s_mov_b32 m0, 5\n\t" R16(V1 "v_writelane_b32 %3, %0, m0\n\t" S1)
                             : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");                             // 48
    if (V == 9) asm volatile(R32(S1 "s_nop 0\n\t") : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");
    if (V == 10) asm volatile(R16("v_readlane_b32 s40, %2, 3\n\t" S1 V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40");  // no dependence
    if (V == 11) asm volatile(R16("v_readlane_b32 s40, %2, 3\n\t" V1 V2 V1 V2 "s_add_u32 %0, %0, s40\n\t") : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40");  // 96
    if (V == 12) asm volatile(R16("v_cmp_gt_u32_e64 s[40:41], %2, %3\n\ts_and_b64 s[42:43], s[40:41], exec\n\tv_cndmask_b32_e64 %2, %2, %3, s[42:43]\n\t") : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40", "s41", "s42", "s43");
    if (V == 13) asm volatile(R16("v_readlane_b32 s40, %2, 3\n\ts_add_u32 %0, %0, s40\n\t" V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40");  // s depends on readlane, v independent
    if (V == 14) asm volatile(R16("v_readlane_b32 s40, %2, 3\n\ts_add_u32 %0, %0, s40\n\t" S2 S2 V2 V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc", "s40");  // 96: one hand-over per 6
    if (V == 16) asm volatile(R32("s_cbranch_scc1 3f\n\t" "s_cbranch_scc1 3f\n\t") "3:\n\t" : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // untaken branches only (scc = 0)
    if (V == 17) asm volatile(R32(S1 "s_cbranch_scc1 4f\n\t") "4:\n\t" : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // s_add : untaken branch 1:1
    if (V == 18) asm volatile(R16(S1 V1 "s_cbranch_scc1 5f\n\t" V2) "5:\n\t" : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // 1 s : 2 v : 1 untaken branch
    if (V == 19) asm volatile(R16(S1 V1 S2 "s_cbranch_scc1 6f\n\t") "6:\n\t" : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // 2 s : 1 v : 1 branch (the kernels' mix)
    if (V == 20) asm volatile(R8("s_branch 7f\n\t7:\n\t" S1 S2 V1 V2 S1 V1 V2) : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // a TAKEN branch (to the next instruction) per 8
    if (V == 21) asm volatile(R16("s_cmp_eq_u32 %0, 0\n\t" "s_cbranch_scc1 8f\n\t" V1 V2) "8:\n\t" : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // compare -> dependent branch, 2 v
    if (V == 15) asm volatile(R16(S1 "v_add_u32 %2, %0, %2\n\t" S2 "v_add_u32 %3, %1, %3\n\t") : "+s"(a), "+s"(b), "+v"(v), "+v"(w) : : "scc");  // SALU -> VALU operand only
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a + b + v + w == 0x12345u) sink[0] = a;
}
static const int N_INSTR[22] = {64, 64, 64, 64, 64, 72, 48, 64, 48, 64, 48, 96, 48, 48, 96, 64, 64, 64, 64, 64, 64, 64};
static const char* NAME[22] = {"s_add only", "v_add only", "s:v 1:1", "s:v 3:1", "s:v 1:3", "1:1 + untaken branch /9", "readlane->s_add->v_add",
                               "1:1 + taken branch /16", "v_add,writelane,s_add", "s_add,s_nop 1:1", "readlane,s_add,v_add indep", "readlane,4 v_add,dep s_add", "v_cmp->s_and->cndmask", "readlane->s_add, indep v", "hand-over per 6 instr", "s_add->v_add operand", "untaken branches only", "s_add : untaken branch 1:1", "s : v : branch 1:2:1", "s : v : branch 2:1:1", "taken branch per 8 (2s:... mix)", "s_cmp->branch, 2 v (per 4)"};
template <int V> void run(unsigned long long* d, unsigned* sink, int waves_per_simd, int iters) {
  const int blocks = 256 * 4 * waves_per_simd;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, sink, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[256 * 32];
  hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const double clk = s / blocks;  // mean shader clocks of a wave's loop (s_memtime)
  const double per_cu_clk = 4.0 * waves_per_simd * N_INSTR[V] * iters / clk;
  printf("  %-28s %d waves/SIMD: %.2f instr per CU per clock (s_memtime), %.3f ms -> %.2f per CU per clock at 2.4 GHz wall\n", NAME[V], waves_per_simd,
         per_cu_clk, ms, 4.0 * waves_per_simd * N_INSTR[V] * iters / (ms * 1e-3 * 2.4e9));
}
int main() {
  unsigned long long* d; unsigned* sink;
  hipMalloc(&d, 256 * 32 * 8); hipMalloc(&sink, 64);
  const int it = 3000;
  for (int w : {8, 1}) {
    run<0>(d, sink, w, it); run<1>(d, sink, w, it); run<2>(d, sink, w, it); run<3>(d, sink, w, it); run<4>(d, sink, w, it);
    run<5>(d, sink, w, it); run<6>(d, sink, w, it); run<7>(d, sink, w, it); run<8>(d, sink, w, it); run<9>(d, sink, w, it);
    run<16>(d, sink, w, it); run<17>(d, sink, w, it); run<18>(d, sink, w, it); run<19>(d, sink, w, it); run<20>(d, sink, w, it); run<21>(d, sink, w, it);
    run<10>(d, sink, w, it); run<11>(d, sink, w, it); run<12>(d, sink, w, it); run<13>(d, sink, w, it); run<14>(d, sink, w, it); run<15>(d, sink, w, it);
    printf("\n");
  }
  return 0;
}
