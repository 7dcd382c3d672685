// Cost of a taken scalar branch (s_branch over one s_nop) against a not-taken one, for 1, 4 and 16 waves of a workgroup
// running the same code (clock64 ticks per branch).   hipcc --offload-arch=gfx950 -O3 branch_lat.hip -o branch_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define B4 "s_branch 1f\n\ts_nop 0\n1:\n\ts_branch 2f\n\ts_nop 0\n2:\n\ts_branch 3f\n\ts_nop 0\n3:\n\ts_branch 4f\n\ts_nop 0\n4:\n\t"
#define N4 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
// a conditional branch that is not taken (scc = 0) in front of one instruction
#define C4 "s_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 9f\n\ts_nop 0\n\ts_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 9f\n\ts_nop 0\n\ts_cmp_eq_u32 %0, 3\n\ts_cbranch_scc1 9f\n\ts_nop 0\n\ts_cmp_eq_u32 %0, 4\n\ts_cbranch_scc1 9f\n\ts_nop 0\n\t"
__global__ void k(long long* t, int mode, int zero) {
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {
    if (mode == 0) asm volatile(B4 B4 B4 B4 ::: "scc");
    else if (mode == 1) asm volatile(N4 N4 N4 N4 ::: "scc");
    else asm volatile(C4 C4 C4 C4 "9:\n\t" :: "s"(zero) : "scc");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main() {
  long long* t; hipMalloc(&t, 64);
  const char* n[3] = {"taken s_branch (+ skipped s_nop)", "two s_nop (no branch)", "s_cmp + not-taken s_cbranch + s_nop"};
  for (int threads : {64, 256, 1024})
    for (int m = 0; m < 3; ++m) {
      k<<<1, threads>>>(t, m, 0); hipDeviceSynchronize();
      long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("%4d threads  %-40s %.1f ticks per unit\n", threads, n[m], h[m] / (256.0 * 16));
    }
  return 0;
}
