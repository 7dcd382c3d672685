// Micro-benchmark + correctness check of the MFMA LDL^T solver (okvis_amd/csrc/ba_ldl16.hpp) on random SPD systems.
//   hipcc --offload-arch=gfx950 -O3 -I../../okvis_amd/csrc ldl16.hip -o ldl16 && ./ldl16
// Prints, per dimension, the relative error against a host Cholesky solve and the clock64 cycles of the phases.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "../../include/okvis_amd_ba.h"   // (ba_types.hpp sizes a record by one of its constants)
// #define LDL_TRACE   // (per-wave event log: perturbs the timing by a few hundred cycles per event)
#include "ba_ldl16.hpp"

constexpr int NW = 16;

__global__ __launch_bounds__(NW * 64) void k_solve(const double* __restrict__ Sg, int D, double* xg, long long* stamps, int* fail,
                                                     int reps, int Ds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int area = ba::ldl16_area_doubles(D);
  const size_t off = (size_t)blockIdx.x * area;
  double* S = smem;
  double* x = smem + area;
  __shared__ int s_fail;
  for (int rep = 0; rep < reps; ++rep) {
    for (int i = threadIdx.x; i < area; i += blockDim.x) S[i] = Sg[off + i];
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    ba::ldl16_solve<NW>(S, D, threadIdx.x, x, &s_fail, (blockIdx.x == 0 && rep == reps - 1) ? stamps : nullptr, Ds);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < D; i += blockDim.x) xg[(size_t)blockIdx.x * 192 + i] = x[i];
  if (threadIdx.x == 0) fail[blockIdx.x] = s_fail;
}


// the diagonal chain alone: one wave eliminates a 16x16 block `n` times (the result of one is perturbed into the next so
// that nothing is hoisted); the other waves of the workgroup either wait at the barrier (mode 0) or poll an LDS flag (1)
__global__ __launch_bounds__(NW * 64) void k_elim(const double* __restrict__ A, double* out, long long* cyc, int mode, int n) {
  __shared__ double sA[256];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15;
  if (tid < 256) sA[tid] = A[tid];
  if (tid == 0) flag = 0;
  __syncthreads();
  if (tid >= 64) {
    if (mode == 1) ba::ldl_wait_ge(&flag, 1);
    return;
  }
  double c[16], b[16], acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
    int jj = j;
    asm volatile("" : "+v"(jj));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      c[i] = sA[i * 16 + jj] + 1e-9 * acc;

    }

    double mine = 0.0;
#if LDL_ELIM16
    if (lane < 16)
#endif
    ba::ldl16_eliminate<true>(c, 16, mine, j);
    acc += mine + c[15];
  }
  const long long t1 = clock64();
  if (tid == 0) {
    cyc[0] = (t1 - t0) / n;
    __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  out[tid] = acc;
}

static void run_elim() {
  std::vector<double> A(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + abs(i - j));
  double *dA, *dout;
  long long* dc;
  hipMalloc(&dA, 256 * 8);
  hipMalloc(&dout, 64 * 8);
  hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    long long c = 0;
    k_elim<<<1, NW * 64>>>(dA, dout, dc, mode, 50);
    hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("elimination of one 16x16 block by one wave, %s: %lld cycles\n", mode ? "15 waves polling an LDS flag" : "other waves gone", c);
  }
}

static int at_host(int nb, int i, int j) {   // i >= j
  const int I = j >> 4, J = i >> 4, r = j & 15, c = i & 15;
  return (I * nb - (I * (I - 1)) / 2 + (J - I)) * 256 + (r >> 2) * 64 + (r & 3) * 16 + c;
}

int main(int argc, char** argv) {
  // D > 0: dense random SPD system.  D < 0: a system with the structure of a window's reduced system, K = -D / 15 frames: pose
  // blocks (6) first, dense among themselves, then speed/bias blocks (9), block-tridiagonal, pose k coupled to speed/bias
  // k - 1 .. k + 1 — solved in the solver's ordering (speed/bias part first, L16::perm) with its zero blocks skipped.
  const int dims[] = {150, -150, -120, -165, 162, 174, 175, 160, 144, 90, 31, 16, 15, 6};
  const int nwg = argc > 1 ? atoi(argv[1]) : 1;
  run_elim();
  for (int Dsigned : dims) {
    const int D = abs(Dsigned);
    const bool structured = Dsigned < 0;
    const int K = D / 15, Dp = structured ? 6 * K : D, Ds = D - Dp;
    const int nb = ba::ldl16_nb(D), area = ba::ldl16_area_doubles(D);
    const ba::L16 LYh{nb, Ds, D};
    std::vector<double> A((size_t)D * D), b(D), G((size_t)D * D);
    srand(D);
    if (!structured) {
      for (auto& v : G) v = rand() / (double)RAND_MAX - 0.5;
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
          double s = 0;
          for (int k = 0; k < D; ++k) s += G[(size_t)i * D + k] * G[(size_t)j * D + k];
          A[(size_t)i * D + j] = s + (i == j ? 0.05 * D : 0.0);
        }
    } else {
      auto frame_of = [&](int i) { return i < Dp ? i / 6 : (i - Dp) / 9; };
      for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
          const bool pi = i < Dp, pj = j < Dp;
          const int d = abs(frame_of(i) - frame_of(j));
          const bool on = (pi && pj) || d <= 1;
          const double v = on ? rand() / (double)RAND_MAX - 0.5 : 0.0;
          A[(size_t)i * D + j] = A[(size_t)j * D + i] = v;
        }
      for (int i = 0; i < D; ++i) {   // diagonally dominant: positive definite
        double s = 0;
        for (int j = 0; j < D; ++j) s += j == i ? 0.0 : fabs(A[(size_t)i * D + j]);
        A[(size_t)i * D + i] = s + 1.0;
      }
    }
    for (auto& v : b) v = rand() / (double)RAND_MAX - 0.5;
    std::vector<double> S((size_t)area * nwg, 0.0);
    for (int w = 0; w < nwg; ++w) {
      for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) S[(size_t)w * area + LYh.at(i, j)] = A[(size_t)i * D + j];
      for (int i = 0; i < D; ++i) S[(size_t)w * area + LYh.at(D, i)] = b[i];
    }
    // host Cholesky
    std::vector<double> L(A), y(b), x(D);
    for (int k = 0; k < D; ++k) {
      L[(size_t)k * D + k] = sqrt(L[(size_t)k * D + k]);
      for (int i = k + 1; i < D; ++i) L[(size_t)i * D + k] /= L[(size_t)k * D + k];
      for (int j = k + 1; j < D; ++j)
        for (int i = j; i < D; ++i) L[(size_t)i * D + j] -= L[(size_t)i * D + k] * L[(size_t)j * D + k];
    }
    for (int i = 0; i < D; ++i) {
      double s = y[i];
      for (int k = 0; k < i; ++k) s -= L[(size_t)i * D + k] * y[k];
      y[i] = s / L[(size_t)i * D + i];
    }
    for (int i = D - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < D; ++k) s -= L[(size_t)k * D + i] * x[k];
      x[i] = s / L[(size_t)i * D + i];
    }
    double *dS, *dx;
    long long* dst;
    int* dfail;
    hipMalloc(&dS, S.size() * 8);
    hipMalloc(&dx, (size_t)nwg * 192 * 8);
    hipMalloc(&dst, 2048 * 8);
    hipMalloc(&dfail, nwg * 4);
    hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dst, 0, 2048 * 8);
    const size_t shmem = (size_t)(area + 192) * 8;
    hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    k_solve<<<nwg, NW * 64, shmem>>>(dS, D, dx, dst, dfail, 3, Ds);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      printf("D=%d: %s\n", D, hipGetErrorString(e));
      return 1;
    }
    // wall time of 20 back-to-back solves inside one launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k_solve<<<nwg, NW * 64, shmem>>>(dS, D, dx, dst, dfail, 23, Ds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventRecord(e0);
    k_solve<<<nwg, NW * 64, shmem>>>(dS, D, dx, dst, dfail, 3, Ds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms3 = 0;
    hipEventElapsedTime(&ms3, e0, e1);
    std::vector<double> xg((size_t)nwg * 192);
    std::vector<long long> st(2048);
    std::vector<int> fl(nwg);
    hipMemcpy(xg.data(), dx, xg.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st.data(), dst, 2048 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(fl.data(), dfail, nwg * 4, hipMemcpyDeviceToHost);
    double err = 0, nrm = 0;
    for (int w = 0; w < nwg; ++w)
      for (int i = 0; i < D; ++i) {
        err = fmax(err, fabs(xg[(size_t)w * 192 + i] - x[i]));
        nrm = fmax(nrm, fabs(x[i]));
      }
    printf("D=%4d nb=%2d  rel err %.2e  fail %d  | cycles: load %lld  factor %lld  backsub %lld  total %lld | %.2f us per solve (incl. LDS fill)\n",
           Dsigned, nb, err / nrm, fl[0], st[1] - st[0], st[2] - st[1], st[3] - st[2], st[3] - st[0], (ms - ms3) * 1000.0 / 20.0);
#ifdef LDL_TS_ALL
    if (D == 150) {
      printf("   steps of wave 0 (cycles; waiting for the hand-overs in brackets):");
      for (int kb = 0; kb + 1 < nb; ++kb) printf(" %lld (%lld)", st[32 + kb + 1] - st[32 + kb], st[32 + 32 + kb] - st[32 + 16 + kb]);
      printf("\n");
    }
#endif
#ifdef LDL_OTS_WAVE
    if (D == 150) {
      printf("   wave %d in step %d (cycles after it saw X): operand %lld  dispatch %lld | panel: in %lld  product %lld  buffer free %lld  published %lld | Q: in %lld  flags %lld  product %lld  handed %lld | P: in %lld  flags %lld  product %lld  handed %lld\n",
             LDL_OTS_WAVE, LDL_TS_STEP, st[81] - st[80], st[82] - st[80], st[83] - st[80], st[84] - st[80], st[85] - st[80], st[86] - st[80], st[87] - st[80], st[88] - st[80], st[89] - st[80],
             st[90] - st[80], st[91] - st[80], st[92] - st[80], st[93] - st[80], st[94] - st[80]);
      printf("      X published by wave 0 at %lld, seen at %lld (absolute)\n", st[16 + 2], st[80]);
    }
#endif
#ifdef LDL_TS_STEP
    if (LDL_TS_STEP + 1 < nb)
      printf("   step %d of wave 0: request %lld  eliminate %lld  publish %lld  operands + hand-over %lld  R %lld  P %lld  transpose %lld | %lld\n", LDL_TS_STEP,
             st[16 + 7] - st[16 + 0], st[16 + 1] - st[16 + 7], st[16 + 2] - st[16 + 1], st[16 + 3] - st[16 + 2], st[16 + 4] - st[16 + 3], st[16 + 5] - st[16 + 4],
             st[16 + 6] - st[16 + 5], st[16 + 6] - st[16 + 0]);
#endif
    if (D == 150 && argc > 2) {   // timeline of all waves, merged and sorted
      std::vector<std::pair<long long, int>> ev;
      for (int w = 0; w < 16; ++w)
        for (int i = 0; i < st[128 + w * 65]; ++i) {
          const long long e = st[128 + w * 65 + 1 + i];
          ev.push_back({(e >> 12), (int)((w << 12) | (e & 0xfff))});
        }
      std::sort(ev.begin(), ev.end());
      for (auto& e : ev) printf("   t=%6lld  wave %2d  %03x\n", e.first - (st[1] & 0xfffffffffffffLL), e.second >> 12, e.second & 0xfff);
    }
    hipFree(dS);
    hipFree(dx);
    hipFree(dst);
    hipFree(dfail);
  }
  return 0;
}
