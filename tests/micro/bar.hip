// barrier cost in a 1024-thread workgroup; LDS round trip with all waves active (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* t) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x;
  lds[tid] = tid; lds[tid + 1024] = 1; lds[tid + 2048] = 2; lds[tid + 3072] = 3;
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < 200; ++i) __syncthreads();
  long long t1 = clock64();
  double a = 0;
#pragma unroll 1
  for (int i = 0; i < 200; ++i) {          // read what another wave wrote, barrier, write, barrier
    a += lds[(tid + 64 * i + 64) & 4095];
    __syncthreads();
    lds[(tid + 64 * i) & 4095] = a;
    __syncthreads();
  }
  long long t2 = clock64();
  // only wave 0 works between barriers (others wait at the barrier): chain of 20 dependent FMAs
  double b = tid;
#pragma unroll 1
  for (int i = 0; i < 200; ++i) {
    if (tid < 64) {
#pragma unroll
      for (int j = 0; j < 20; ++j) b = fma(b, 1.0000001, 1e-9);
    }
    __syncthreads();
  }
  long long t3 = clock64();
  out[tid] = a + b;
  if (tid == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; }
}
int main() {
  double* out; long long* t;
  hipMalloc(&out, 8 * 1024); hipMalloc(&t, 64);
  for (int threads : {1024, 256}) {
    for (int rep = 0; rep < 2; ++rep) k<<<1, threads>>>(out, t);
    long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
    printf("%4d threads: barrier %.1f cyc; read+barrier+write+barrier %.1f cyc; 20-FMA chain by wave 0 + barrier %.1f cyc\n", threads, h[0] / 200.0, h[1] / 200.0, h[2] / 200.0);
  }
  return 0;
}
