// What does ONE wave per SIMD pay per instruction?  (k_agents_fsm is a ~50 000-instruction dependent chain on a lone
// wave per SIMD.)  One wave per CU, 64-instruction bodies of one pattern, s_memtime around 2000 iterations.
//   0 dependent v_add_u32                     1 four independent v_add_u32 chains
//   2 v_cmp_e64 -> s_and_b64 -> v_cndmask_e64 (mask through an SGPR pair and the SALU)
//   3 v_cmp_e32 (vcc) -> v_cndmask_e32        4 dependent v_mul_lo_u32
//   5 dependent v_mad_u64_u32                 6 dependent s_add_u32
//   7 v_add + taken s_branch every 8 instr.   8 v_alignbit/v_xor chain (RNG-like)
//   9 ds_write_b16 + v_add                    10 v_cmp_e64 -> v_cndmask_e64 (no SALU hop)
//   11 v_bfi-style select from VGPR mask      12 v_cmp_e64 -> s_and_b64 -> s_and_saveexec / s_or exec pair
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
template <int V>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters) {
  __shared__ unsigned short lds[4096];
  unsigned v = threadIdx.x, w = 3, x = 5, y = 7, s = blockIdx.x;
  unsigned long long m = 0;
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(R64("v_add_u32 %0, 1, %0\n\t") : "+v"(v));
    if (V == 1) asm volatile(R16("v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1\n\tv_add_u32 %2, 1, %2\n\tv_add_u32 %3, 1, %3\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y));
    if (V == 2) asm volatile(R16("v_cmp_gt_u32_e64 s[40:41], %0, %1\n\ts_and_b64 s[40:41], s[40:41], exec\n\tv_cndmask_b32_e64 %0, %0, %1, s[40:41]\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v), "+v"(w) : : "s40", "s41", "scc");
    if (V == 3) asm volatile(R16("v_cmp_gt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v), "+v"(w) : : "vcc");
    if (V == 4) asm volatile(R64("v_mul_lo_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(w));
    if (V == 5) asm volatile(R64("v_mad_u64_u32 %0, s[40:41], %1, %2, 0\n\t") : "+v"(m) : "v"(v), "v"(w) : "s40", "s41");
    if (V == 6) asm volatile(R64("s_add_u32 %0, %0, 1\n\t") : "+s"(s) : : "scc");
    if (V == 7) asm volatile(R8("v_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\ts_branch 1f\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t1:\n\t") : "+v"(v));
    if (V == 8) asm volatile(R16("v_alignbit_b32 %0, %0, %1, 8\n\tv_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %1, %1, %0, 27\n\tv_xor_b32 %1, %1, %0\n\t") : "+v"(v), "+v"(w));
    if (V == 9) asm volatile(R16("ds_write_b16 %1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v) : "v"((unsigned)(threadIdx.x * 2)) : "memory");
    if (V == 10) asm volatile(R16("v_cmp_gt_u32_e64 s[40:41], %0, %1\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %0, %1, s[40:41]\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v), "+v"(w) : : "s40", "s41");
    if (V == 11) asm volatile(R16("v_sub_u32 %2, %1, %0\n\tv_ashrrev_i32 %2, 31, %2\n\tv_bfi_b32 %0, %2, %1, %0\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v), "+v"(w), "+v"(x));
    if (V == 12) asm volatile(R8("v_cmp_gt_u32_e64 s[40:41], %0, %1\n\ts_and_saveexec_b64 s[42:43], s[40:41]\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\ts_or_b64 exec, exec, s[42:43]\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %0, 1, %0\n\t") : "+v"(v), "+v"(w) : : "s40", "s41", "s42", "s43", "scc");
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (v + w + x + y + s + (unsigned)m == 0x12345u) out[blockIdx.x + 1024] = lds[v & 4095];
}
template <int V> double run(unsigned long long* d, int blocks, int iters, int instr_per_iter) {
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, iters);
  hipDeviceSynchronize();
  unsigned long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 64; ++i) s += h[i];
  return s / 64 / iters / instr_per_iter;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  const int it = 2000;
  for (int blocks : {256, 256 * 4}) {
    printf("%d waves (1 per %s): s_memtime ticks per instruction (64-instr bodies)\n", blocks, blocks == 256 ? "CU" : "SIMD");
    printf("  dep v_add %.2f | 4 indep v_add chains %.2f | cmp_e64->s_and->cndmask (+add; per 4) %.2f | cmp_e32->cndmask_e32 (+2 add; per 4) %.2f\n",
           run<0>(d, blocks, it, 64), run<1>(d, blocks, it, 64), run<2>(d, blocks, it, 16), run<3>(d, blocks, it, 16));
    printf("  v_mul_lo_u32 %.2f | v_mad_u64_u32 %.2f | dep s_add %.2f | 7 v_add + taken branch (per block) %.2f\n",
           run<4>(d, blocks, it, 64), run<5>(d, blocks, it, 64), run<6>(d, blocks, it, 64), run<7>(d, blocks, it, 8));
    printf("  alignbit/xor chain %.2f | ds_write_b16 + 3 add (per 4) %.2f | cmp_e64->nop->cndmask_e64 (+add; per 4) %.2f | sub/ashr/bfi/add (per 4) %.2f | cmp/saveexec/2 add/or exec/3 add (per 8) %.2f\n",
           run<8>(d, blocks, it, 64), run<9>(d, blocks, it, 16), run<10>(d, blocks, it, 16), run<11>(d, blocks, it, 16), run<12>(d, blocks, it, 8));
  }
  return 0;
}
