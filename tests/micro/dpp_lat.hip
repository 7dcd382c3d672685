// Throughput / latency of the 64-bit DPP forms used by ba_ldl16.hpp (one wave):  hipcc --offload-arch=gfx950 -O3 dpp_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define FD(acc, src, mul, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul))
#define FP(acc, src, mul) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(src), "v"(mul))
__global__ void k(double* out, long long* t) {
  const int tid = threadIdx.x;
  double a[16], u = 1e-9 * tid, s = 1.0 + tid;
  for (int i = 0; i < 16; ++i) a[i] = i + tid;
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FD(a[i], s, u, 3);   // 16 independent accumulators, DPP source fixed
  }
  long long t1 = clock64();
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FP(a[i], s, u);      // the same without DPP
  }
  long long t2 = clock64();
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FD(a[0], s, u, 3);   // dependent chain through the accumulator
  }
  long long t3 = clock64();
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FD(a[i], a[i], u, 3);   // accumulator is also the DPP source (as in the elimination)
  }
  long long t4 = clock64();
#pragma unroll 1
  for (int it = 0; it < 100; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FP(a[0], s, u);      // dependent chain, no DPP
  }
  long long t5 = clock64();
  double r = 0;
  for (int i = 0; i < 16; ++i) r += a[i];
  out[tid] = r;
  if (tid == 0) {
    t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4;
  }
}
int main() {
  double* o; long long* t;
  hipMalloc(&o, 64 * 8); hipMalloc(&t, 64);
  k<<<1, 64>>>(o, t); hipDeviceSynchronize();
  long long h[5]; hipMemcpy(h, t, 40, hipMemcpyDeviceToHost);
  const char* n[5] = {"v_fmac_f64_dpp independent", "v_fmac_f64 independent", "v_fmac_f64_dpp dependent", "v_fmac_f64_dpp acc==src independent", "v_fmac_f64 dependent"};
  for (int i = 0; i < 5; ++i) printf("%-40s %.2f cycles/instr\n", n[i], h[i] / 1600.0);
  return 0;
}
