// LDS latencies seen by one wave of a 1024-thread workgroup (clock64 ticks), the other 15 waves gone or polling an LDS flag:
// dependent ds_read_b64 chain, a batch of 8 independent ds_read_b64 + wait, 4 ds_write_b64 + wait, write -> read back.
//   hipcc --offload-arch=gfx950 -O3 lds_lat.hip -o lds_lat
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(long long* t, double* out, int mode, int nwaves_poll) {
  __shared__ double buf[4096];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 1024) buf[i] = (double)((i * 7 + 64) & 4095 & ~63);   // next index (multiple of 64) as a double
  if (tid == 0) flag = 0;
  __syncthreads();
  if (tid >= 64) {
    if ((tid >> 6) <= nwaves_poll)
      while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    return;
  }
  double acc = 0;
  int idx = 0;
  const long long t0 = clock64();
  if (mode == 0) {   // dependent chain
    for (int it = 0; it < 256; ++it) { const double v = buf[idx + lane]; idx = (int)v; acc += v; idx = __builtin_amdgcn_readfirstlane(idx); }
  } else if (mode == 1) {   // 8 independent reads, one wait
    for (int it = 0; it < 256; ++it) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = buf[((it * 8 + u) * 64 & 4095) + lane];
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      acc += v[0] + v[7];
    }
  } else if (mode == 2) {   // 4 writes, wait
    for (int it = 0; it < 256; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) buf[((it * 4 + u) * 64 & 4095) + lane] = acc + u;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else {   // write -> read back (no wait in between), wait
    for (int it = 0; it < 256; ++it) {
      buf[(it * 64 & 4095) + lane] = acc + it;
      asm volatile("" ::: "memory");
      double v = buf[(it * 64 & 4095) + (lane ^ 1)];
      asm volatile("" : "+v"(v));
      acc += v;
    }
  }
  const long long t1 = clock64();
  if (tid == 0) { t[mode] = t1 - t0; __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  out[tid] = acc + idx;
}
int main() {
  long long* t; double* o;
  hipMalloc(&t, 64); hipMalloc(&o, 1024 * 8);
  const char* n[4] = {"dependent ds_read_b64", "8 ds_read_b64 + wait", "4 ds_write_b64 + wait", "ds_write_b64, ds_read_b64 back, wait"};
  for (int np : {0, 3, 15})
    for (int m = 0; m < 4; ++m) {
      k<<<1, 1024>>>(t, o, m, np); hipDeviceSynchronize();
      long long h[4]; hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
      printf("%2d waves polling  %-40s %.1f ticks\n", np, n[m], h[m] / 256.0);
    }
  return 0;
}
