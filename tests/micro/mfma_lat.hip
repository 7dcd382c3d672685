// Latency / issue rate of v_mfma_f64_16x16x4_f64 on one wave and on four waves of one SIMD... (clock64 ticks)
//   hipcc --offload-arch=gfx950 -O3 mfma_lat.hip -o mfma_lat && ./mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* t, int mode) {
  const int tid = threadIdx.x;
  double a = 1.0 + 1e-3 * tid, b = 1.0 - 1e-3 * tid;
  v4 c0{0, 0, 0, 0}, c1{0, 0, 0, 0}, c2{0, 0, 0, 0}, c3{0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
  if (mode == 0) {   // dependent chain
#pragma unroll 1
    for (int it = 0; it < 100; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
  } else if (mode == 1) {   // four independent accumulators
#pragma unroll 1
    for (int it = 0; it < 100; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
    }
  } else {   // result feeds the next operand (as R -> P in the chain): mfma -> v_mul -> mfma
#pragma unroll 1
    for (int it = 0; it < 100; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        a = c0[0] * 1e-30 + 1.0;
      }
    }
  }
  long long t1 = clock64();
  out[tid] = c0[0] + c1[1] + c2[2] + c3[3] + a;
  if (tid == 0) t[mode] = t1 - t0;
}
int main() {
  double* o; long long* t;
  hipMalloc(&o, 1024 * 8); hipMalloc(&t, 64);
  const char* n[3] = {"dependent accumulator chain", "4 independent accumulators", "mfma -> valu -> mfma operand chain"};
  for (int threads : {64, 256, 1024})
    for (int m = 0; m < 3; ++m) {
      k<<<1, threads>>>(o, t, m); hipDeviceSynchronize();
      long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("%4d threads  %-36s %.1f ticks per MFMA (per wave)\n", threads, n[m], h[m] / 800.0);
    }
  return 0;
}
