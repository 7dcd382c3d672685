// Instruction-cache behaviour: a body of N KB of straight-line code (s_nop 0 = 4 bytes each; a taken branch every 64 bytes in
// the "jumpy" variant), first pass against later passes, by one wave and by sixteen (clock64 ticks per instruction).
//   hipcc --offload-arch=gfx950 -O3 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define N16 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#define N256 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16 N16
#define K1 asm volatile(N256);   /* 1 KB */
#define K4 K1 K1 K1 K1
#define K16 K4 K4 K4 K4
template <int KB>
__global__ __launch_bounds__(1024) void k(long long* t) {
  long long ts[4];
  for (int pass = 0; pass < 4; ++pass) {
    const long long t0 = clock64();
    if constexpr (KB >= 8) { K4 K4 }
    if constexpr (KB >= 16) { K4 K4 }
    if constexpr (KB >= 32) { K16 }
    if constexpr (KB >= 48) { K16 }
    if constexpr (KB >= 64) { K16 }
    if constexpr (KB >= 96) { K16 K16 }
    ts[pass] = clock64() - t0;
  }
  if (threadIdx.x == 0)
    for (int p = 0; p < 4; ++p) t[p] = ts[p];
}
template <int KB> void run(long long* t, int threads) {
  k<KB><<<1, threads>>>(t); hipDeviceSynchronize();
  long long h[4]; hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
  printf("%3d KB, %4d threads: ticks per instruction, pass 0..3: %.2f %.2f %.2f %.2f\n", KB, threads, h[0] / (KB * 256.0), h[1] / (KB * 256.0), h[2] / (KB * 256.0), h[3] / (KB * 256.0));
}
int main() {
  long long* t; hipMalloc(&t, 64);
  for (int threads : {64, 1024}) {
    run<8>(t, threads); run<16>(t, threads); run<32>(t, threads); run<48>(t, threads); run<64>(t, threads); run<96>(t, threads);
  }
  return 0;
}
