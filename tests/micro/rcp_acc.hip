// Accuracy of v_rcp_f64 and of the Newton refinements built on it (max / mean error in ulp against 1/x in long double on the host).
//   hipcc --offload-arch=gfx950 -O3 rcp_acc.hip -o rcp_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double y = __builtin_amdgcn_rcp(d);
  y0[i] = y;
  y = fma(y, fma(-d, y, 1.0), y);
  y1[i] = y;
  y = fma(y, fma(-d, y, 1.0), y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), c(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    const double m = 1.0 + rand() / (double)RAND_MAX + rand() / (double)RAND_MAX / RAND_MAX;
    x[i] = ldexp(m, (rand() % 80) - 40) * ((i & 7) == 0 ? 1.0 : 1.0);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  const char* nm[3] = {"v_rcp_f64", "+ 1 Newton step", "+ 2 Newton steps"};
  std::vector<double>* v[3] = {&a, &b, &c};
  for (int s = 0; s < 3; ++s) {
    long double mx = 0, sum = 0;
    for (int i = 0; i < n; ++i) {
      const long double ex = 1.0L / (long double)x[i];
      const double exd = (double)ex;
      const long double ulp = (long double)(nextafter(fabs(exd), INFINITY) - fabs(exd));
      const long double e = fabsl((long double)(*v[s])[i] - ex) / ulp;
      mx = e > mx ? e : mx; sum += e;
    }
    printf("%-18s max error %.3Lg ulp, mean %.3Lg ulp (relative: max %.3Lg)\n", nm[s], mx, sum / n, mx * 1.11e-16L);
  }
  return 0;
}
