// Latency micro-benchmarks on one wave (diagnostics; not part of the product): dependent fp64 FMA chain,
// v_readlane round trip, LDS read->use, rsq+Newton chain.  hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* t, int waves_busy) {
  __shared__ double lds[1024];
  const int tid = threadIdx.x;
  lds[tid & 1023] = tid * 1e-3;
  __syncthreads();
  if (tid >= 64) {                      // optional competing waves on the same CU doing FMAs
    if (tid < 64 * (1 + waves_busy)) {
      double a = tid, b = 1.0000001;
      for (int i = 0; i < 20000; ++i) a = fma(a, b, 1e-9);
      out[tid] = a;
    }
    return;
  }
  double a = 1.0 + tid * 1e-9, b = 1.0000001;
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < 100; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) a = fma(a, b, 1e-9);
  }
  long long t1 = clock64();
  // readlane chain: value -> sgpr -> valu -> readlane ...
  double c = a;
#pragma unroll 1
  for (int i = 0; i < 100; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      int lo = __builtin_amdgcn_readlane(__double2loint(c), 3), hi = __builtin_amdgcn_readlane(__double2hiint(c), 3);
      c = c + __hiloint2double(hi, lo);
    }
  }
  long long t2 = clock64();
  // LDS read -> use -> address chain
  int idx = tid;
  double d = 0;
#pragma unroll 1
  for (int i = 0; i < 100; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double v = lds[idx & 1023];
      d += v;
      idx = (idx + (int)v + 1) & 1023;
    }
  }
  long long t3 = clock64();
  // rsq + 2 newton chain
  double e = 2.0 + tid;
#pragma unroll 1
  for (int i = 0; i < 100; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double y = __builtin_amdgcn_rsq(e);
      const double h = 0.5 * e;
      y = y * (1.5 - h * y * y);
      y = y * (1.5 - h * y * y);
      e = e + y;
    }
  }
  long long t4 = clock64();
  // fp32 fma chain
  float f = 1.0f + tid;
#pragma unroll 1
  for (int i = 0; i < 100; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) f = fmaf(f, 1.0000001f, 1e-9f);
  }
  long long t5 = clock64();
  out[tid] = a + c + d + e + f;
  if (tid == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; }
}
int main() {
  double* out; long long* t;
  hipMalloc(&out, 8 * 2048); hipMalloc(&t, 64);
  for (int busy : {0, 3, 15}) {
    for (int rep = 0; rep < 2; ++rep) k<<<1, 1024>>>(out, t, busy);
    long long h[5];
    hipMemcpy(h, t, 40, hipMemcpyDeviceToHost);
    printf("busy waves %2d: per op cycles: fma64 %.1f  readlane+add %.1f  lds-chain %.1f  rsq+2newton+add %.1f  fma32 %.1f\n", busy,
           h[0] / 1000.0, h[1] / 1000.0, h[2] / 1000.0, h[3] / 1000.0, h[4] / 1000.0);
  }
  return 0;
}
