// issuebench.hip — diagnostic (not product): what does ONE wave pay per instruction on gfx950?  Cycles per iteration of small
// loops (s_memtime), one wave per workgroup, `blocks` workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP8(x) x x x x x x x x
template <int mode>
__global__ void __launch_bounds__(64) k(long long *out, int iters, int *sink) {
  __shared__ volatile unsigned lds[256];
  lds[threadIdx.x] = threadIdx.x; __syncthreads();
  unsigned s = (unsigned)iters, v = threadIdx.x, acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    switch (mode) {
      case 9: asm volatile("s_nop 0\n" : "+s"(s) :: "scc", "vcc"); break;                                            // loop overhead only
      case 10: asm volatile(REP8(REP8("s_add_u32 %0, %0, 1\n")) : "+s"(s) :: "scc", "vcc"); break;                 // 64 dependent SALU
      case 11: asm volatile(REP8(REP8("v_add_u32 %0, %0, 1\n")) : "+v"(v) :: "scc", "vcc"); break;                 // 64 dependent VALU
      case 12: asm volatile(REP8(REP8("s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n")) : "+s"(s), "+v"(v) :: "scc", "vcc"); break;   // 64 x (independent SALU + VALU)
      case 13: asm volatile(REP8("v_readlane_b32 %0, %1, 1\n s_nop 3\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0\n") : "+s"(s), "+v"(v) :: "scc", "vcc"); break;
      case 14: asm volatile(REP8("v_cmp_ne_u32 vcc, 0, %1\n s_and_b64 vcc, vcc, exec\n s_cbranch_vccz 1f\n v_add_u32 %1, %1, 1\n1:\n") : "+s"(s), "+v"(v) :: "scc", "vcc"); break;  // vector compare -> scalar branch (not taken)

      case 0: asm volatile(REP8("s_add_u32 %0, %0, 1\n") : "+s"(s) :: "scc", "vcc"); break;                       // 8 dependent SALU
      case 1: asm volatile(REP8("v_add_u32 %0, %0, 1\n") : "+v"(v) :: "scc", "vcc"); break;                       // 8 dependent VALU
      case 2: asm volatile(REP8("v_readfirstlane_b32 %0, %1\n s_nop 3\n v_add_u32 %1, %0, %1\n") : "+s"(s), "+v"(v) :: "scc", "vcc"); break;   // VALU->SGPR->VALU x8
      case 3: asm volatile(REP8("s_add_u32 %0, %0, 1\n v_add_u32 %1, %0, %1\n") : "+s"(s), "+v"(v) :: "scc", "vcc"); break;   // SALU -> VALU reads it, x8
      case 4: asm volatile(REP8("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n") : "+s"(s) :: "scc", "vcc"); break; // 8 taken short forward branches  (labels reused: numeric local labels ok)
      case 5: asm volatile(REP8("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0x3fc, %0\n") : "+v"(acc), "+v"(v) :: "scc", "vcc"); break; // 8 dependent LDS reads (pointer chase)
      case 6: asm volatile(REP8("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %0, vcc\n") : "+v"(acc), "+v"(v) :: "scc", "vcc"); break; // cmp->cndmask x8
      case 7: asm volatile(REP8("s_lshl_b32 %0, %0, 1\n s_and_b32 %0, %0, 0xffff\n") : "+s"(s) :: "scc", "vcc"); break;   // 16 dependent SALU
      case 8: asm volatile(REP8("v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n") : "+s"(s), "+v"(v) :: "scc", "vcc"); break; // s->v->s round trips (compiler-style, hazards unhandled? readfirstlane result used by v_mov: SGPR read by VALU needs no wait)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = (long long)(t1 - t0);
  if (s + v + acc == 0x12345u) sink[0] = 1;
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 150, iters = 20000;
  long long *out; int *sink; hipMalloc(&out, blocks * 8); hipMalloc(&sink, 4);
  const char *names[] = {"8 dependent SALU", "8 dependent VALU", "8 x (readfirstlane, s_nop 3, v_add using it)", "8 x (s_add, v_add reading it)", "8 taken forward branches (cmp + branch + skipped nop)",
                         "8 dependent LDS reads (+wait +and)", "8 x (v_cmp -> v_cndmask)", "16 dependent SALU", "8 x (v_mov s->v, readfirstlane v->s)", "empty loop (s_nop)", "64 dependent SALU", "64 dependent VALU", "64 x (SALU + independent VALU)", "8 x (v_readlane, s_nop 3, s_add, v_mov)", "8 x (v_cmp, s_and vcc, s_cbranch not taken, v_add)"};
  for (int mode = 0; mode <= 14; mode++) {
    for (int rep = 0; rep < 2; rep++) {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(64), 0, 0, out, iters, sink); break;
      switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) }
      hipDeviceSynchronize(); }
    std::vector<long long> c(blocks); hipMemcpy(c.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : c) avg += (double)x; avg /= blocks;
    printf("mode %d (%s): %.1f cycles per iteration\n", mode, names[mode], avg / iters);
  }
  return 0;
}
