// diagnostics (not a test): how many streams of one process run their kernel chains side by side on this runtime — graphs and
// eager launches, streams of one priority and of mixed priorities (the runtime keeps a pool of hardware queues per priority).
// hipcc --offload-arch=gfx950 -O2 stream_concurrency.hip -o stream_concurrency ;  ./stream_concurrency [extra idle streams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void delay(long long ticks, long long* stamp) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int idle = argc > 1 ? std::atoi(argv[1]) : 0;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::printf("stream priority range: least %d .. greatest %d; %d idle streams created first\n", lo, hi, idle);
  std::vector<hipStream_t> idlers(idle);
  for (auto& s : idlers) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, s, 100, nullptr);
  }
  CK(hipDeviceSynchronize());
  long long* stamps;
  CK(hipMalloc(&stamps, 64 * sizeof(long long)));
  for (int mixed = 0; mixed < 2; ++mixed)
    for (int graph = 1; graph >= 0; --graph)
      for (int ns : {2, 3, 4, 5, 6, 8, 12}) {
        std::vector<hipStream_t> st(ns);
        for (int k = 0; k < ns; ++k) {
          if (mixed) CK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, k % 3 == 0 ? 0 : (k % 3 == 1 ? hi : lo)));
          else CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        }
        std::vector<hipGraphExec_t> ex(ns);
        if (graph)
          for (int k = 0; k < ns; ++k) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeRelaxed));
            for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, st[k], 2500, it == 39 ? stamps + k : nullptr);
            CK(hipStreamEndCapture(st[k], &g));
            CK(hipGraphInstantiate(&ex[k], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
          }
        double best = 1e30;
        std::vector<long long> h(ns);
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipDeviceSynchronize());
          const double t0 = now();
          for (int k = 0; k < ns; ++k) {
            if (graph) CK(hipGraphLaunch(ex[k], st[k]));
            else
              for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, st[k], 2500, it == 39 ? stamps + k : nullptr);
          }
          CK(hipDeviceSynchronize());
          best = std::min(best, now() - t0);
          CK(hipMemcpy(h.data(), stamps, ns * sizeof(long long), hipMemcpyDeviceToHost));
        }
        long long first = h[0];
        for (long long v : h) first = std::min(first, v);
        std::printf("%s priorities, %s, %2d streams x 40 kernels of 25 us: %5.0f us (side by side: 1000 + gaps); streams finish at (us after the first):", mixed ? "mixed" : "one  ",
                    graph ? "graphs" : "eager ", ns, best);
        for (long long v : h) std::printf(" %.0f", (v - first) / 100.0);
        std::printf("\n");
        for (int k = 0; k < ns; ++k) {
          if (graph) CK(hipGraphExecDestroy(ex[k]));
          CK(hipStreamDestroy(st[k]));
        }
      }
  return 0;
}
