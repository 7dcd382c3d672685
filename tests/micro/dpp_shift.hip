// micro check: which lane does a DPP row_shl:1 / row_shr:1 read, and what does a lane without a source get
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x101, 0xF, 0xF, false);        // row_shl:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x111, 0xF, 0xF, false);   // row_shr:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x101, 0xF, 0xF, true);    // bound_ctrl
  const bool brk = (lane % 5) == 0;
  const unsigned long long m = __ballot(brk);
  out[192 + lane] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"row_shl:1 (0x101)", "row_shr:1 (0x111)", "row_shl:1 bound_ctrl", "mbcnt of lane%5==0"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 34; ++i) printf(" %d", h[64 * r + i]); printf("\n"); }
  return 0;
}
