// which SIMD does each wave of a 1024-thread workgroup run on?  (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 16 * 4);
  k<<<4, 1024>>>(d);
  unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("wg %d:", b); for (int w = 0; w < 16; ++w) printf(" w%d[wave_id %u simd %u cu %u sh %u se %u]", w, h[b*16+w] & 15, (h[b*16+w] >> 4) & 3, (h[b*16+w] >> 8) & 15, (h[b*16+w] >> 12) & 1, (h[b*16+w] >> 13) & 7); printf("\n"); }
  return 0;
}
