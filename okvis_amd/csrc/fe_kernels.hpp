// Batched reprojection pieces of the OKVIS frontend (include/okvis_amd_frontend.h): one work-item per candidate match.
//
// Restated from the reference, one candidate at a time there:
//   okvis_frontend/src/ProbabilisticStereoTriangulator.cpp:178-236 (stereoTriangulate), :253-355 (getUncertainty),
//   :358-385 (computeReprojectionError4); okvis_frontend/src/stereo_triangulation.cpp:50-137 (triangulateFast);
//   okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:109-146, 345-378, 426-446 (project, projectHomogeneous,
//   backProject) and the undistort() Gauss-Newton loops of the three distortion classes;
//   okvis_frontend/src/VioKeyframeWindowMatchingAlgorithm.cpp:165-213, 320-337, 494-512 (3D-2D projection and gating).
// The reprojection error and its Jacobians are the backend's own reproj_linearize (ba_math.hpp), the same function the
// linearisation kernel of the optimiser uses.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/okvis_amd_frontend.h"
#include "ba_math.hpp"

namespace fe {

using namespace ba;

struct Camera {
  double intr[12];
  int model, width, height;
};

struct TriParams {
  Camera cam_a, cam_b;
  double T_AB[7];
  double info6[36];       // H_(0:6,0:6) = J^T J of the PoseError at T_AB = inverse of UOplus
  double sigma_ray_own;   // 0.5 / min(fu_A, fu_B)
  int n_a, n_b, n_pairs, want_uncertainty;
  const float* kp_a;      // [n_a][3]
  const float* kp_b;
  const int32_t* pairs;   // [n_pairs][2]
  const double* sigma_ray;  // [n_pairs] or null
  double* hp;             // [n_pairs][4]
  double* cov;            // [n_pairs][9]
  uint8_t* flags;
  double* gn;             // [n_pairs][81] the 9x9 Gauss-Newton matrix of getUncertainty (row-major), or null
};

// D::undistort(pointDistorted, &pointUndistorted): Gauss-Newton on distort(), at most 5 iterations
// (RadialTangentialDistortion.hpp / RadialTangentialDistortion8.hpp: success below 1e-4; EquidistantDistortion.hpp: 1e-2)
__device__ inline bool undistort(int model, const double* k, const double* pd, double* pu) {
  if (model == DIST_NONE) {
    pu[0] = pd[0], pu[1] = pd[1];
    return true;
  }
  const double ok_below = model == DIST_EQUI ? 1e-2 : 1e-4;
  double x0 = pd[0], x1 = pd[1];
  bool success = false;
  for (int it = 0; it < 5; ++it) {
    double d[2] = {x0, x1}, E[4] = {1, 0, 0, 1};
    distort(model, k, x0, x1, d, E);
    const double e0 = pd[0] - d[0], e1 = pd[1] - d[1];
    // du = (E^T E)^-1 E^T e
    const double a = E[0] * E[0] + E[2] * E[2], b = E[0] * E[1] + E[2] * E[3], c = E[1] * E[1] + E[3] * E[3];
    const double idet = 1.0 / (a * c - b * b);
    const double g0 = E[0] * e0 + E[2] * e1, g1 = E[1] * e0 + E[3] * e1;
    x0 += (c * g0 - b * g1) * idet;
    x1 += (a * g1 - b * g0) * idet;
    const double chi2 = e0 * e0 + e1 * e1;
    if (chi2 < ok_below) success = true;
    if (chi2 < 1e-15) {
      success = true;
      break;
    }
  }
  pu[0] = x0, pu[1] = x1;
  return success;
}

// PinholeCamera<D>::backProject (implementation/PinholeCamera.hpp:426-446)
__device__ inline bool back_project(const Camera& cam, double u, double v, double* dir) {
  const double p[2] = {(u - cam.intr[2]) * (1.0 / cam.intr[0]), (v - cam.intr[3]) * (1.0 / cam.intr[1])};
  double und[2];
  const bool ok = undistort(cam.model, cam.intr + 4, p, und);
  dir[0] = und[0], dir[1] = und[1], dir[2] = 1.0;
  return ok;
}

// PinholeCamera<D>::projectHomogeneous -> project (:345-356, :109-146), optionally with the 2x3 point Jacobian (:148-226).
// Returns the ProjectionStatus; uv is written unless the status is INVALID by |z| < 1e-12.
__device__ inline int project_homogeneous(const Camera& cam, const double* hp, double* uv, double* J23) {
  double x = hp[0], y = hp[1], z = hp[2];
  if (hp[3] < 0) x = -x, y = -y, z = -z;
  if (fabs(z) < 1.0e-12) return OKVIS_FE_PROJ_INVALID;
  const double rz = 1.0 / z;
  double d[2] = {0, 0}, Jd[4] = {1, 0, 0, 1};
  const bool ok = distort(cam.model, cam.intr + 4, x * rz, y * rz, d, Jd);
  const double fu = cam.intr[0], fv = cam.intr[1];
  if (J23) {
    const double rz2 = rz * rz;
    J23[0] = fu * Jd[0] * rz;
    J23[1] = fu * Jd[1] * rz;
    J23[2] = -fu * (x * Jd[0] + y * Jd[1]) * rz2;
    J23[3] = fv * Jd[2] * rz;
    J23[4] = fv * Jd[3] * rz;
    J23[5] = -fv * (x * Jd[2] + y * Jd[3]) * rz2;
  }
  if (!ok) return OKVIS_FE_PROJ_INVALID;
  uv[0] = fu * d[0] + cam.intr[2];
  uv[1] = fv * d[1] + cam.intr[3];
  if (uv[0] < 0.0 || uv[1] < 0.0 || uv[0] >= cam.width || uv[1] >= cam.height) return OKVIS_FE_PROJ_OUTSIDE_IMAGE;
  return z > 0.0 ? OKVIS_FE_PROJ_SUCCESSFUL : OKVIS_FE_PROJ_BEHIND;
}

// computeReprojectionError4 (ProbabilisticStereoTriangulator.cpp:358-385)
__device__ inline bool reprojection_error4(const Camera& cam, const float* kp, const double* hp, double* err) {
  double y[2];
  if (project_homogeneous(cam, hp, y, nullptr) != OKVIS_FE_PROJ_SUCCESSFUL) return false;
  const double sd = 0.8 * (double)kp[2] / 12.0;
  const double inv = 1.0 / (sd * sd);
  const double d0 = y[0] - (double)kp[0], d1 = y[1] - (double)kp[1];
  *err = d0 * (inv * d0) + d1 * (inv * d1);
  return true;
}

__device__ inline void normalize3(double* v) {
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n, v[1] /= n, v[2] /= n;
}

// triangulateFast with p1 = 0 (stereo_triangulation.cpp:50-137)
__device__ inline void triangulate_fast(const double* e1, const double* p2, const double* e2, double sigma, double* hp,
                                        bool* is_valid, bool* is_parallel) {
  *is_parallel = false;
  *is_valid = false;
  const double* t12 = p2;
  const double b0 = t12[0] * e1[0] + t12[1] * e1[1] + t12[2] * e1[2];
  const double b1 = t12[0] * e2[0] + t12[1] * e2[1] + t12[2] * e2[2];
  const double a00 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
  double a10 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
  double a01 = -a10;
  const double a11 = -(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
  if (a10 < 0.0) a10 = -a10, a01 = -a01;  // wrong viewing direction
  const double det = a00 * a11 - a01 * a10;
  if (!(fabs(det) > 1.0e-6)) {  // computeInverseWithCheck(..., 1e-6)
    *is_parallel = true;
    const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
    if (sqrt(cx * cx + cy * cy + cz * cz) < 6 * sigma) *is_valid = true;
    double h[4] = {(e1[0] + e2[0]) / 2.0, (e1[1] + e2[1]) / 2.0, (e1[2] + e2[2]) / 2.0, 1e-3};
    const double n = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2] + h[3] * h[3]);
    for (int i = 0; i < 4; ++i) hp[i] = h[i] / n;
    return;
  }
  const double idet = 1.0 / det;
  const double l0 = (a11 * idet) * b0 + (-a01 * idet) * b1;
  const double l1 = (-a10 * idet) * b0 + (a00 * idet) * b1;
  double mid[3], err2 = 0, diff[3], diff2 = 0, dot = 0;
  for (int i = 0; i < 3; ++i) {
    const double xm = l0 * e1[i], xn = l1 * e2[i] + p2[i];
    mid[i] = (xm + xn) / 2.0;
    const double er = mid[i] - xm;
    err2 += er * er;
    diff[i] = mid[i] - 0.5 * t12[i];
    diff2 += diff[i] * diff[i];
    dot += diff[i] * e1[i];
  }
  const double chi2 = err2 * (1.0 / (diff2 * sigma * sigma));
  *is_valid = !(chi2 > 9);
  if (dot < 0)
    for (int i = 0; i < 3; ++i) mid[i] = 0.5 * t12[i] - diff[i];
  const double n = sqrt(mid[0] * mid[0] + mid[1] * mid[1] + mid[2] * mid[2] + 1.0);
  hp[0] = mid[0] / n, hp[1] = mid[1] / n, hp[2] = mid[2] / n, hp[3] = 1.0 / n;
}

constexpr int TRI_THREADS = 64;

// Eigen's ColPivHouseholderQR<Matrix<double,9,9>>::rank() on a column-major 9x9 held in LDS as M[(r + 9 c) * TRI_THREADS]
// (one matrix per work-item, consecutive work-items in consecutive banks): pivot = largest remaining column norm, the
// norms down-dated after each reflection and the chosen one recomputed; rank = number of |R_ii| above 9 eps max |R_jj|
// among the pivots before the first column whose norm falls under the absolute threshold.
__device__ inline int qr_rank9(double* M) {
  auto at = [&](int r, int c) -> double& { return M[(r + 9 * c) * TRI_THREADS]; };
  const double eps = 2.220446049250313e-16;
  double cn[9], maxn = 0;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    double s = 0;
    for (int r = 0; r < 9; ++r) s += at(r, c) * at(r, c);
    cn[c] = s;
    maxn = fmax(maxn, s);
  }
  const double threshold_helper = maxn * eps * eps / 9.0;
  int nonzero = 9;
  double maxpivot = 0, diag[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    int big = k;
    double bn = -1;
#pragma unroll
    for (int c = k; c < 9; ++c)
      if (cn[c] > bn) bn = cn[c], big = c;
    bn = 0;
    for (int r = k; r < 9; ++r) bn += at(r, big) * at(r, big);  // the chosen norm is recomputed
    if (nonzero == 9 && bn < threshold_helper * (9 - k)) nonzero = k;
    if (big != k)
      for (int r = 0; r < 9; ++r) {
        const double t = at(r, k);
        at(r, k) = at(r, big);
        at(r, big) = t;
      }
    // the norms swap with their columns; entry k is not read again
#pragma unroll
    for (int c = k + 1; c < 9; ++c)
      if (c == big) cn[c] = cn[k];
    // makeHouseholderInPlace on the tail of column k
    const double c0 = at(k, k);
    double tail = 0;
    for (int r = k + 1; r < 9; ++r) tail += at(r, k) * at(r, k);
    double beta, tau;
    if (tail <= 2.2250738585072014e-308) {
      tau = 0, beta = c0;
      for (int r = k + 1; r < 9; ++r) at(r, k) = 0;
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      const double sc = 1.0 / (c0 - beta);
      for (int r = k + 1; r < 9; ++r) at(r, k) *= sc;
      tau = (beta - c0) / beta;
    }
    at(k, k) = beta;
    diag[k] = beta;
    maxpivot = fmax(maxpivot, fabs(beta));
    for (int c = k + 1; c < 9; ++c) {
      double t = at(k, c);
      for (int r = k + 1; r < 9; ++r) t += at(r, k) * at(r, c);
      t *= tau;
      at(k, c) -= t;
      for (int r = k + 1; r < 9; ++r) at(r, c) -= t * at(r, k);
    }
#pragma unroll
    for (int c = k + 1; c < 9; ++c) cn[c] -= at(k, c) * at(k, c);
  }
  const double thr = maxpivot * (eps * 9.0);
  int rank = 0;
  for (int i = 0; i < 9; ++i) rank += (i < nonzero && fabs(diag[i]) > thr) ? 1 : 0;
  return rank;
}

// bottom-right 3x3 of the inverse of the symmetric positive definite 9x9 H = [A B; B^T C] (A 6x6): (C - B^T A^-1 B)^-1
__device__ inline void point_covariance(const double* H /*row-major 9x9*/, double* cov) {
  double L[21];  // lower Cholesky factor of A, packed by rows
  auto Lx = [&](int i, int j) -> double& { return L[i * (i + 1) / 2 + j]; };
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = H[9 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lx(i, k) * Lx(j, k);
      Lx(i, j) = (i == j) ? sqrt(s) : s / Lx(j, j);
    }
  double Y[18];  // Y = L^-1 B (6x3)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double s = H[9 * i + 6 + c];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= Lx(i, k) * Y[3 * k + c];
      Y[3 * i + c] = s / Lx(i, i);
    }
  double S[6];  // xx xy xz yy yz zz of C - Y^T Y
  int o = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) {
      double s = H[9 * (6 + a) + 6 + b];
#pragma unroll
      for (int k = 0; k < 6; ++k) s -= Y[3 * k + a] * Y[3 * k + b];
      S[o++] = s;
    }
  const double c00 = S[3] * S[5] - S[4] * S[4], c01 = S[2] * S[4] - S[1] * S[5], c02 = S[1] * S[4] - S[2] * S[3];
  const double idet = 1.0 / (S[0] * c00 + S[1] * c01 + S[2] * c02);
  cov[0] = c00 * idet;
  cov[1] = cov[3] = c01 * idet;
  cov[2] = cov[6] = c02 * idet;
  cov[4] = (S[0] * S[5] - S[2] * S[2]) * idet;
  cov[5] = cov[7] = (S[1] * S[2] - S[0] * S[4]) * idet;
  cov[8] = (S[0] * S[3] - S[1] * S[1]) * idet;
}

__global__ void __launch_bounds__(TRI_THREADS) stereo_triangulate_kernel(const TriParams P) {
  __shared__ double sH[81 * TRI_THREADS];
  const int i = blockIdx.x * TRI_THREADS + threadIdx.x;
  if (i >= P.n_pairs) return;
  const int ia = P.pairs[2 * i], ib = P.pairs[2 * i + 1];
  const float* ka = P.kp_a + 3 * ia;
  const float* kb = P.kp_b + 3 * ib;
  double sigma = P.sigma_ray ? P.sigma_ray[i] : -1.0;
  if (sigma == -1.0) sigma = P.sigma_ray_own;
  unsigned flags = 0;
  double hp[4] = {0, 0, 0, 0};
  // ---- stereoTriangulate (:178-236) ----
  double dA[3], dB[3], C_AB[9], dBA[3];
  back_project(P.cam_a, (double)ka[0], (double)ka[1], dA);
  back_project(P.cam_b, (double)kb[0], (double)kb[1], dB);
  qrot(P.T_AB + 3, C_AB);
  mat3_vec(C_AB, dB, dBA);
  normalize3(dA);
  normalize3(dBA);
  bool valid, parallel;
  triangulate_fast(dA, P.T_AB, dBA, sigma, hp, &valid, &parallel);
  if (!parallel) flags |= OKVIS_FE_TRI_NOT_PARALLEL;
  bool hp_assigned = false;
  if (valid) {
    double errA, errB;
    valid = reprojection_error4(P.cam_a, ka, hp, &errA);
    if (valid) {
      // hp_B = T_BA hp_A = (C_AB^T (p - r w), w)
      const double d[3] = {hp[0] - P.T_AB[0] * hp[3], hp[1] - P.T_AB[1] * hp[3], hp[2] - P.T_AB[2] * hp[3]};
      double hb[4];
      mat3_Tvec(C_AB, d, hb);
      hb[3] = hp[3];
      valid = reprojection_error4(P.cam_b, kb, hb, &errB);
      if (valid) {
        hp_assigned = true;
        if (errA > 4.0 || errB > 4.0) valid = false;
      }
    }
  }
  if (valid) flags |= OKVIS_FE_TRI_VALID;
  if (P.hp)
    for (int k = 0; k < 4; ++k) P.hp[4 * i + k] = hp_assigned ? hp[k] : 0.0;
  // ---- getUncertainty (:253-355), called by the 6-argument overload only after a successful triangulation ----
  if (valid && P.want_uncertainty) {
    const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
    ReprojLin la, lb;
    const double swA = 1.0 / (0.8 * (double)ka[2] / 12.0), swB = 1.0 / (0.8 * (double)kb[2] / 12.0);
    reproj_linearize(ident, ident, hp, P.cam_a.intr, P.cam_a.model, (double)ka[0], (double)ka[1], swA, false, &la);
    // "evaluate again closer": the Jacobians of B that enter H are those at the point pulled 20 % towards the baseline centre
    double hc[4];
    for (int k = 0; k < 3; ++k) hc[k] = 0.8 * (hp[k] - P.T_AB[k] / 2.0 * hp[3]) + P.T_AB[k] / 2.0 * hp[3];
    hc[3] = hp[3];
    reproj_linearize(P.T_AB, ident, hc, P.cam_b.intr, P.cam_b.model, (double)kb[0], (double)kb[1], swB, false, &lb);
    bool can_init = !(lb.r[0] * lb.r[0] + lb.r[1] * lb.r[1] < 4.0);
    double H[81];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        double v = (r < 6 && c < 6) ? P.info6[6 * r + c] : 0.0;
        const double ar0 = r < 6 ? lb.Jp[r] : lb.Jl[r - 6], ar1 = r < 6 ? lb.Jp[6 + r] : lb.Jl[3 + r - 6];
        const double ac0 = c < 6 ? lb.Jp[c] : lb.Jl[c - 6], ac1 = c < 6 ? lb.Jp[6 + c] : lb.Jl[3 + c - 6];
        v += ar0 * ac0 + ar1 * ac1;
        if (r >= 6 && c >= 6) v += la.Jl[r - 6] * la.Jl[c - 6] + la.Jl[3 + r - 6] * la.Jl[3 + c - 6];
        H[9 * r + c] = v;
      }
    if (P.gn)
      for (int k = 0; k < 81; ++k) P.gn[81 * (size_t)i + k] = H[k];
    double* M = sH + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = 0; c < 9; ++c) M[(r + 9 * c) * TRI_THREADS] = H[9 * r + c];
    if (qr_rank9(M) < 9) {
      can_init = false;
      flags |= OKVIS_FE_TRI_RANK_DEFICIENT;
    } else if (P.cov) {
      double cov[9];
      point_covariance(H, cov);
      for (int k = 0; k < 9; ++k) P.cov[9 * i + k] = cov[k];
    }
    if (can_init && !parallel) flags |= OKVIS_FE_TRI_CAN_INIT;
  }
  if (P.flags) P.flags[i] = (uint8_t)flags;
}

struct ProjParams {
  Camera cam;
  double T_CbW[7], P3[9];
  int n;
  const double* hp_W;
  double* uv;
  double* U;
  uint8_t* status;
};

// doSetup, Match3D2D (VioKeyframeWindowMatchingAlgorithm.cpp:177-205)
__global__ void project_landmarks_kernel(const ProjParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const double* h = P.hp_W + 4 * i;
  double C[9], p[3], hc[4];
  qrot(P.T_CbW + 3, C);
  mat3_vec(C, h, p);
  for (int k = 0; k < 3; ++k) hc[k] = p[k] + P.T_CbW[k] * h[3];
  hc[3] = h[3];
  double uv[2] = {0, 0}, J[6] = {0, 0, 0, 0, 0, 0};
  const int st = project_homogeneous(P.cam, hc, uv, J);
  if (P.status) P.status[i] = (uint8_t)st;
  if (P.uv) P.uv[2 * i] = uv[0], P.uv[2 * i + 1] = uv[1];
  if (P.U) {
    double JP[6];
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c) JP[3 * r + c] = J[3 * r] * P.P3[c] + J[3 * r + 1] * P.P3[3 + c] + J[3 * r + 2] * P.P3[6 + c];
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) P.U[4 * i + 2 * r + c] = JP[3 * r] * J[3 * c] + JP[3 * r + 1] * J[3 * c + 1] + JP[3 * r + 2] * J[3 * c + 2];
  }
}

struct GateParams {
  int n_proj, n_b, n_pairs;
  const double* uv;
  const double* U;
  const float* kp_b;
  const int32_t* pairs;
  double* chi2;
  uint8_t* flags;
};

// verifyMatch (:320-337) / setBestMatch (:494-512), Match3D2D
__global__ void gate_3d2d_kernel(const GateParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_pairs) return;
  const int a = P.pairs[2 * i], b = P.pairs[2 * i + 1];
  const float* kb = P.kp_b + 3 * b;
  const double sd = 0.8 * (double)kb[2] / 12.0, s2 = sd * sd;
  const double u00 = s2 + P.U[4 * a], u01 = P.U[4 * a + 1], u10 = P.U[4 * a + 2], u11 = s2 + P.U[4 * a + 3];
  const double e0 = P.uv[2 * a] - (double)kb[0], e1 = P.uv[2 * a + 1] - (double)kb[1];
  const double idet = 1.0 / (u00 * u11 - u01 * u10);
  // err^T U^-1 err with U^-1 = [u11 -u01; -u10 u00] / det
  const double chi2 = e0 * ((u11 * idet) * e0 + (-u01 * idet) * e1) + e1 * ((-u10 * idet) * e0 + (u00 * idet) * e1);
  unsigned f = 0;
  if (chi2 < 4.0 && chi2 > -1.0) f |= OKVIS_FE_GATE_VERIFIED;  // `const int chi2 = ...; chi2 < 4.0`: truncation towards zero
  if (!(chi2 > 4.0)) f |= OKVIS_FE_GATE_ACCEPTED;
  if (sqrt(u00 * u00 + u01 * u01 + u10 * u10 + u11 * u11) > 25.0 / (s2 * sqrt(2.0))) f |= OKVIS_FE_GATE_UNCERTAIN;
  if (P.chi2) P.chi2[i] = chi2;
  if (P.flags) P.flags[i] = (uint8_t)f;
}

}  // namespace fe
