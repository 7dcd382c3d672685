// C-ABI of the batched frontend pieces (include/okvis_amd_frontend.h).  Host buffers in and out: the inputs of one call are
// packed into one pinned staging block and go to the device with one copy, one kernel runs, the outputs come back with one
// copy.  No CPU path: every entry fails with OKVIS_BA_ERR_NO_DEVICE / a HIP status when there is no GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "fe_kernels.hpp"

struct okvis_fe_context {
  int device = 0;
  hipStream_t stream = nullptr;
  char* h_stage = nullptr;  // pinned
  char* d_stage = nullptr;
  size_t cap = 0;
};

namespace {

#define FE_TRY(expr)                                                  \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) return OKVIS_BA_HIP_ERROR_BASE + (int)_e;   \
  } while (0)

int reserve(okvis_fe_context* c, size_t bytes) {
  if (bytes <= c->cap) return OKVIS_BA_OK;
  size_t cap = c->cap ? c->cap : (size_t)1 << 16;
  while (cap < bytes) cap *= 2;
  if (c->h_stage) FE_TRY(hipHostFree(c->h_stage));
  if (c->d_stage) FE_TRY(hipFree(c->d_stage));
  c->h_stage = c->d_stage = nullptr;
  c->cap = 0;
  FE_TRY(hipHostMalloc((void**)&c->h_stage, cap, hipHostMallocDefault));
  FE_TRY(hipMalloc((void**)&c->d_stage, cap));
  c->cap = cap;
  return OKVIS_BA_OK;
}

// sequential layout of the staging block, every array aligned to 16 bytes
struct Layout {
  size_t size = 0;
  size_t add(size_t bytes) {
    const size_t o = size;
    size += (bytes + 15) & ~(size_t)15;
    return o;
  }
};

bool camera_ok(const okvis_fe_camera* c) {
  return c && c->model >= OKVIS_BA_DIST_NONE && c->model <= OKVIS_BA_DIST_RADTAN8 && c->intr[0] > 0 && c->intr[1] > 0 &&
         c->width > 0 && c->height > 0;
}
fe::Camera to_device(const okvis_fe_camera* c) {
  fe::Camera d;
  std::memcpy(d.intr, c->intr, sizeof(d.intr));
  d.model = c->model, d.width = c->width, d.height = c->height;
  return d;
}

// inverse of a symmetric positive definite 6x6 (row-major) through its Cholesky factor; false when not positive definite
bool spd_inverse6(const double* A, double* inv) {
  double L[36] = {0};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return false;
        L[6 * i + i] = std::sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  double Li[36] = {0};  // L^-1
  for (int c = 0; c < 6; ++c)
    for (int i = c; i < 6; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[6 * i + k] * Li[6 * k + c];
      Li[6 * i + c] = s / L[6 * i + i];
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = (i > j ? i : j); k < 6; ++k) s += Li[6 * k + i] * Li[6 * k + j];
      inv[6 * i + j] = s;
    }
  return true;
}

}  // namespace

extern "C" {

int okvis_fe_create(okvis_fe_context** out, int device) {
  if (!out) return OKVIS_BA_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OKVIS_BA_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return OKVIS_BA_ERR_ARG;
  FE_TRY(hipSetDevice(device));
  okvis_fe_context* c = new (std::nothrow) okvis_fe_context();
  if (!c) return OKVIS_BA_ERR_ARG;
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return OKVIS_BA_HIP_ERROR_BASE + (int)e;
  }
  *out = c;
  return OKVIS_BA_OK;
}

void okvis_fe_destroy(okvis_fe_context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream), (void)hipStreamDestroy(c->stream);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->d_stage) (void)hipFree(c->d_stage);
  delete c;
}

int okvis_fe_stereo_triangulate(okvis_fe_context* c, const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b,
                                const double* T_AB, const double* UOplus, int32_t n_a, const float* kp_a, int32_t n_b,
                                const float* kp_b, int32_t n_pairs, const int32_t* pairs, const double* sigma_ray,
                                int32_t want_uncertainty, double* hp_a, double* cov, uint8_t* flags) {
  return okvis_fe_stereo_triangulate_gn(c, cam_a, cam_b, T_AB, UOplus, n_a, kp_a, n_b, kp_b, n_pairs, pairs, sigma_ray, want_uncertainty, hp_a,
                                        cov, flags, nullptr);
}

int okvis_fe_stereo_triangulate_gn(okvis_fe_context* c, const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b,
                                   const double* T_AB, const double* UOplus, int32_t n_a, const float* kp_a, int32_t n_b,
                                   const float* kp_b, int32_t n_pairs, const int32_t* pairs, const double* sigma_ray,
                                   int32_t want_uncertainty, double* hp_a, double* cov, uint8_t* flags, double* gn) {
  if (!c || !camera_ok(cam_a) || !camera_ok(cam_b) || !T_AB || !UOplus || n_a < 0 || n_b < 0 || n_pairs < 0) return OKVIS_BA_ERR_ARG;
  if (n_pairs == 0) return OKVIS_BA_OK;
  if (!kp_a || !kp_b || !pairs || n_a == 0 || n_b == 0) return OKVIS_BA_ERR_ARG;
  for (int i = 0; i < n_pairs; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= n_a || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= n_b) return OKVIS_BA_ERR_ARG;
  fe::TriParams P;
  P.cam_a = to_device(cam_a), P.cam_b = to_device(cam_b);
  std::memcpy(P.T_AB, T_AB, sizeof(P.T_AB));
  if (!spd_inverse6(UOplus, P.info6)) return OKVIS_BA_ERR_NUMERIC;
  P.sigma_ray_own = 0.5 / std::fmin(cam_a->intr[0], cam_b->intr[0]);
  P.n_a = n_a, P.n_b = n_b, P.n_pairs = n_pairs, P.want_uncertainty = want_uncertainty;
  FE_TRY(hipSetDevice(c->device));
  Layout in, all;
  const size_t o_ka = in.add(sizeof(float) * 3 * n_a), o_kb = in.add(sizeof(float) * 3 * n_b);
  const size_t o_pairs = in.add(sizeof(int32_t) * 2 * n_pairs), o_sig = sigma_ray ? in.add(sizeof(double) * n_pairs) : 0;
  all = in;
  const size_t o_hp = all.add(sizeof(double) * 4 * n_pairs), o_cov = all.add(sizeof(double) * 9 * n_pairs);
  const size_t o_fl = all.add(n_pairs);
  const size_t o_gn = gn ? all.add(sizeof(double) * 81 * n_pairs) : 0;
  if (int rc = reserve(c, all.size)) return rc;
  std::memcpy(c->h_stage + o_ka, kp_a, sizeof(float) * 3 * n_a);
  std::memcpy(c->h_stage + o_kb, kp_b, sizeof(float) * 3 * n_b);
  std::memcpy(c->h_stage + o_pairs, pairs, sizeof(int32_t) * 2 * n_pairs);
  if (sigma_ray) std::memcpy(c->h_stage + o_sig, sigma_ray, sizeof(double) * n_pairs);
  FE_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, in.size, hipMemcpyHostToDevice, c->stream));
  if (cov) FE_TRY(hipMemsetAsync(c->d_stage + o_cov, 0, sizeof(double) * 9 * n_pairs, c->stream));
  if (gn) FE_TRY(hipMemsetAsync(c->d_stage + o_gn, 0, sizeof(double) * 81 * n_pairs, c->stream));
  P.gn = gn ? (double*)(c->d_stage + o_gn) : nullptr;
  P.kp_a = (const float*)(c->d_stage + o_ka), P.kp_b = (const float*)(c->d_stage + o_kb);
  P.pairs = (const int32_t*)(c->d_stage + o_pairs);
  P.sigma_ray = sigma_ray ? (const double*)(c->d_stage + o_sig) : nullptr;
  P.hp = (double*)(c->d_stage + o_hp), P.cov = (double*)(c->d_stage + o_cov), P.flags = (uint8_t*)(c->d_stage + o_fl);
  hipLaunchKernelGGL(fe::stereo_triangulate_kernel, dim3((n_pairs + fe::TRI_THREADS - 1) / fe::TRI_THREADS), dim3(fe::TRI_THREADS),
                     0, c->stream, P);
  FE_TRY(hipGetLastError());
  FE_TRY(hipMemcpyAsync(c->h_stage + o_hp, c->d_stage + o_hp, all.size - o_hp, hipMemcpyDeviceToHost, c->stream));
  FE_TRY(hipStreamSynchronize(c->stream));
  if (hp_a) std::memcpy(hp_a, c->h_stage + o_hp, sizeof(double) * 4 * n_pairs);
  if (cov) std::memcpy(cov, c->h_stage + o_cov, sizeof(double) * 9 * n_pairs);
  if (flags) std::memcpy(flags, c->h_stage + o_fl, n_pairs);
  if (gn) std::memcpy(gn, c->h_stage + o_gn, sizeof(double) * 81 * n_pairs);
  return OKVIS_BA_OK;
}

int okvis_fe_project_landmarks(okvis_fe_context* c, const okvis_fe_camera* cam_b, const double* T_CbW, const double* P3,
                               int32_t n, const double* hp_W, double* uv, double* U, uint8_t* status) {
  if (!c || !camera_ok(cam_b) || !T_CbW || !P3 || n < 0) return OKVIS_BA_ERR_ARG;
  if (n == 0) return OKVIS_BA_OK;
  if (!hp_W) return OKVIS_BA_ERR_ARG;
  fe::ProjParams P;
  P.cam = to_device(cam_b);
  std::memcpy(P.T_CbW, T_CbW, sizeof(P.T_CbW));
  std::memcpy(P.P3, P3, sizeof(P.P3));
  P.n = n;
  FE_TRY(hipSetDevice(c->device));
  Layout in, all;
  const size_t o_hp = in.add(sizeof(double) * 4 * n);
  all = in;
  const size_t o_uv = all.add(sizeof(double) * 2 * n), o_U = all.add(sizeof(double) * 4 * n), o_st = all.add(n);
  if (int rc = reserve(c, all.size)) return rc;
  std::memcpy(c->h_stage + o_hp, hp_W, sizeof(double) * 4 * n);
  FE_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, in.size, hipMemcpyHostToDevice, c->stream));
  P.hp_W = (const double*)(c->d_stage + o_hp);
  P.uv = (double*)(c->d_stage + o_uv), P.U = (double*)(c->d_stage + o_U), P.status = (uint8_t*)(c->d_stage + o_st);
  hipLaunchKernelGGL(fe::project_landmarks_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, P);
  FE_TRY(hipGetLastError());
  FE_TRY(hipMemcpyAsync(c->h_stage + o_uv, c->d_stage + o_uv, all.size - o_uv, hipMemcpyDeviceToHost, c->stream));
  FE_TRY(hipStreamSynchronize(c->stream));
  if (uv) std::memcpy(uv, c->h_stage + o_uv, sizeof(double) * 2 * n);
  if (U) std::memcpy(U, c->h_stage + o_U, sizeof(double) * 4 * n);
  if (status) std::memcpy(status, c->h_stage + o_st, n);
  return OKVIS_BA_OK;
}

int okvis_fe_gate_3d2d(okvis_fe_context* c, int32_t n_proj, const double* uv, const double* U, int32_t n_b, const float* kp_b,
                       int32_t n_pairs, const int32_t* pairs, double* chi2, uint8_t* flags) {
  if (!c || n_proj < 0 || n_b < 0 || n_pairs < 0) return OKVIS_BA_ERR_ARG;
  if (n_pairs == 0) return OKVIS_BA_OK;
  if (!uv || !U || !kp_b || !pairs || n_proj == 0 || n_b == 0) return OKVIS_BA_ERR_ARG;
  for (int i = 0; i < n_pairs; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= n_proj || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= n_b) return OKVIS_BA_ERR_ARG;
  fe::GateParams P;
  P.n_proj = n_proj, P.n_b = n_b, P.n_pairs = n_pairs;
  FE_TRY(hipSetDevice(c->device));
  Layout in, all;
  const size_t o_uv = in.add(sizeof(double) * 2 * n_proj), o_U = in.add(sizeof(double) * 4 * n_proj);
  const size_t o_kb = in.add(sizeof(float) * 3 * n_b), o_pairs = in.add(sizeof(int32_t) * 2 * n_pairs);
  all = in;
  const size_t o_chi = all.add(sizeof(double) * n_pairs), o_fl = all.add(n_pairs);
  if (int rc = reserve(c, all.size)) return rc;
  std::memcpy(c->h_stage + o_uv, uv, sizeof(double) * 2 * n_proj);
  std::memcpy(c->h_stage + o_U, U, sizeof(double) * 4 * n_proj);
  std::memcpy(c->h_stage + o_kb, kp_b, sizeof(float) * 3 * n_b);
  std::memcpy(c->h_stage + o_pairs, pairs, sizeof(int32_t) * 2 * n_pairs);
  FE_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, in.size, hipMemcpyHostToDevice, c->stream));
  P.uv = (const double*)(c->d_stage + o_uv), P.U = (const double*)(c->d_stage + o_U);
  P.kp_b = (const float*)(c->d_stage + o_kb), P.pairs = (const int32_t*)(c->d_stage + o_pairs);
  P.chi2 = (double*)(c->d_stage + o_chi), P.flags = (uint8_t*)(c->d_stage + o_fl);
  hipLaunchKernelGGL(fe::gate_3d2d_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream, P);
  FE_TRY(hipGetLastError());
  FE_TRY(hipMemcpyAsync(c->h_stage + o_chi, c->d_stage + o_chi, all.size - o_chi, hipMemcpyDeviceToHost, c->stream));
  FE_TRY(hipStreamSynchronize(c->stream));
  if (chi2) std::memcpy(chi2, c->h_stage + o_chi, sizeof(double) * n_pairs);
  if (flags) std::memcpy(flags, c->h_stage + o_fl, n_pairs);
  return OKVIS_BA_OK;
}

}  // extern "C"
