// diagnostics (not a test): do parallel branches of ONE captured hipGraph overlap on this runtime?  A fork/join graph of
// delay kernels (one workgroup each) against the same kernels in a line; and how many streams run concurrently
// (GPU_MAX_HW_QUEUES).   hipcc --offload-arch=gfx950 -O2 graph_branches.hip -o graph_branches
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void delay(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const long long us50 = 5000;   // wall_clock64: 100 MHz
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  for (int branched = 0; branched < 2; ++branched) {
    hipGraph_t g;
    hipGraphExec_t ex;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed));
    for (int it = 0; it < 20; ++it) {
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, a, us50 / 5, nullptr);           // "solve" 10 us
      if (branched) {
        CK(hipEventRecord(fork, a));
        CK(hipStreamWaitEvent(b, fork, 0));
        hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, b, us50, nullptr);             // side branch 50 us
        CK(hipEventRecord(join, b));
      } else {
        hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, a, us50, nullptr);
      }
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, a, us50 / 2, nullptr);           // main branch 25 + 25 us
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, a, us50 / 2, nullptr);
      if (branched) CK(hipStreamWaitEvent(a, join, 0));
    }
    CK(hipStreamEndCapture(a, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipStreamSynchronize(a));
      const double t0 = now();
      CK(hipGraphLaunch(ex, a));
      CK(hipStreamSynchronize(a));
      std::printf("%s graph, 20 x (10 | 50 || 25 + 25) us: %.1f us per iteration (serial 110, overlapped 60)\n", branched ? "fork/join" : "linear   ",
                  (now() - t0) / 20);
    }
    CK(hipGraphExecDestroy(ex));
    CK(hipGraphDestroy(g));
  }
  // two graphs (three kernels a line each) on two streams + cross-stream dependency per iteration is not expressible; what IS:
  // how many independent streams really run side by side
  for (int ns : {2, 3, 4, 6, 8}) {
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipGraphExec_t> ex(ns);
    for (int k = 0; k < ns; ++k) {
      hipGraph_t g;
      CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeRelaxed));
      for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, st[k], us50 / 2, nullptr);
      CK(hipStreamEndCapture(st[k], &g));
      CK(hipGraphInstantiate(&ex[k], g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
    }
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      for (int k = 0; k < ns; ++k) CK(hipGraphLaunch(ex[k], st[k]));
      CK(hipDeviceSynchronize());
      std::printf("%d streams x 40 kernels of 25 us: %.0f us (side by side: 1000 + gaps)\n", ns, now() - t0);
    }
    for (int k = 0; k < ns; ++k) { CK(hipGraphExecDestroy(ex[k])); CK(hipStreamDestroy(st[k])); }
  }
  return 0;
}
