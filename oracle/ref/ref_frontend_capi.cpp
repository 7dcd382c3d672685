// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Flat C wrapper around the okvis reference's OWN ProbabilisticStereoTriangulator (okvis_frontend/src/
// ProbabilisticStereoTriangulator.cpp and stereo_triangulation.cpp, compiled unmodified by oracle/ref/Makefile against the
// stand-in Eigen / glog / OpenCV headers of oracle/shim) and around the reference's camera classes for the 3D-2D projection
// and gating lines of VioKeyframeWindowMatchingAlgorithm.cpp (that file itself needs brisk / the dense matcher and is not
// compilable here: its lines :177-205, :320-337 and :494-512 are restated below on the reference's PinholeCamera<D>, each with
// its citation).  Same argument meaning as include/okvis_amd_frontend.h, one candidate at a time like the reference.
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <okvis/MultiFrame.hpp>
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/NCameraSystem.hpp>
#include <okvis/cameras/NoDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/cameras/RadialTangentialDistortion8.hpp>
#include <okvis/triangulation/ProbabilisticStereoTriangulator.hpp>

#include "okvis_amd_frontend.h"

namespace cam = okvis::cameras;
using okvis::kinematics::Transformation;

namespace {

Transformation to_T(const double* p) {
  Transformation T;
  Eigen::Matrix<double, 7, 1> c;
  for (int i = 0; i < 7; ++i) c[i] = p[i];
  T.setCoeffs(c);
  return T;
}

template <class D>
D make_distortion(const double* k);
template <>
cam::RadialTangentialDistortion make_distortion(const double* k) { return cam::RadialTangentialDistortion(k[0], k[1], k[2], k[3]); }
template <>
cam::EquidistantDistortion make_distortion(const double* k) { return cam::EquidistantDistortion(k[0], k[1], k[2], k[3]); }
template <>
cam::RadialTangentialDistortion8 make_distortion(const double* k) {
  return cam::RadialTangentialDistortion8(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]);
}
template <>
cam::NoDistortion make_distortion(const double*) { return cam::NoDistortion(); }

template <class D>
std::shared_ptr<const cam::PinholeCamera<D> > make_camera(const okvis_fe_camera* c) {
  return std::shared_ptr<const cam::PinholeCamera<D> >(new cam::PinholeCamera<D>(
      c->width, c->height, c->intr[0], c->intr[1], c->intr[2], c->intr[3], make_distortion<D>(c->intr + 4)));
}

template <class D>
std::shared_ptr<okvis::MultiFrame> make_frame(const okvis_fe_camera* c, cam::NCameraSystem::DistortionType dt, int n, const float* kp,
                                              uint64_t id) {
  cam::NCameraSystem ncs;
  ncs.addCamera(std::shared_ptr<const Transformation>(new Transformation()), make_camera<D>(c), dt, false);
  std::shared_ptr<okvis::MultiFrame> mf(new okvis::MultiFrame(ncs, okvis::Time(1, 0), id));
  std::vector<cv::KeyPoint> kps;
  for (int i = 0; i < n; ++i) kps.push_back(cv::KeyPoint(kp[3 * i], kp[3 * i + 1], kp[3 * i + 2]));
  mf->resetKeypoints(0, kps);
  return mf;
}

template <class D>
int triangulate(cam::NCameraSystem::DistortionType dt, const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b, const double* T_AB,
                const double* UOplus, int n_a, const float* kp_a, int n_b, const float* kp_b, int n_pairs, const int32_t* pairs,
                const double* sigma_ray, int want_uncertainty, double* hp_a, double* cov, uint8_t* flags) {
  typedef cam::PinholeCamera<D> G;
  std::shared_ptr<okvis::MultiFrame> fa = make_frame<D>(cam_a, dt, n_a, kp_a, 1), fb = make_frame<D>(cam_b, dt, n_b, kp_b, 2);
  Eigen::Matrix<double, 6, 6> U;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) U(r, c) = UOplus[6 * r + c];
  // what VioKeyframeWindowMatchingAlgorithm does: a default-constructed triangulator, then resetFrames (:127-170)
  okvis::triangulation::ProbabilisticStereoTriangulator<G> tri;
  tri.resetFrames(fa, fb, 0, 0, to_T(T_AB), U);
  for (int i = 0; i < n_pairs; ++i) {
    const size_t ia = (size_t)pairs[2 * i], ib = (size_t)pairs[2 * i + 1];
    const double sigma = sigma_ray ? sigma_ray[i] : -1.0;
    unsigned f = 0;
    Eigen::Vector4d hp(0, 0, 0, 0);
    bool notParallel = false;
    const bool valid = tri.stereoTriangulate(ia, ib, hp, notParallel, sigma);  // the 5-argument overload (:178-236)
    if (valid) f |= OKVIS_FE_TRI_VALID;
    if (notParallel) f |= OKVIS_FE_TRI_NOT_PARALLEL;
    for (int k = 0; k < 4; ++k) hp_a[4 * i + k] = hp[k];
    for (int k = 0; k < 9; ++k) cov[9 * i + k] = 0.0;
    if (valid && want_uncertainty) {
      // the 6-argument overload (:239-250): triangulate again, getUncertainty, AND the two decisions
      Eigen::Vector4d hp2(0, 0, 0, 0);
      Eigen::Matrix3d P = Eigen::Matrix3d::Zero();
      bool canInit = false;
      if (tri.stereoTriangulate(ia, ib, hp2, P, canInit, sigma)) {
        if (canInit) f |= OKVIS_FE_TRI_CAN_INIT;
        // getUncertainty leaves P untouched when the rank test fails (:349-352): a zero P after a valid call says so
        bool written = false;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            cov[9 * i + 3 * r + c] = P(r, c);
            written |= P(r, c) != 0.0;
          }
        if (!written) f |= OKVIS_FE_TRI_RANK_DEFICIENT;
      }
    }
    flags[i] = (uint8_t)f;
  }
  return 0;
}

template <class D>
int project(const okvis_fe_camera* c, const double* T_CbW, const double* P3, int n, const double* hp_W, double* uv, double* U,
            uint8_t* status) {
  std::shared_ptr<const cam::PinholeCamera<D> > g = make_camera<D>(c);
  const Transformation T = to_T(T_CbW);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector4d hp_W_i(hp_W[4 * i], hp_W[4 * i + 1], hp_W[4 * i + 2], hp_W[4 * i + 3]);
    const Eigen::Vector4d hp_Cb = T * hp_W_i;  // VioKeyframeWindowMatchingAlgorithm.cpp:179
    Eigen::Vector2d kptB(0, 0);
    const cam::CameraBase::ProjectionStatus st = g->projectHomogeneous(hp_Cb, &kptB);  // :180-185
    status[i] = (uint8_t)(int)st;
    // project and get uncertainty (:194-203)
    Eigen::Matrix<double, 2, 4> jacobian;
    jacobian.setZero();
    Eigen::Matrix4d P_C = Eigen::Matrix4d::Zero();
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) P_C(r, cc) = P3[3 * r + cc];
    Eigen::Vector2d kpt2(0, 0);
    g->projectHomogeneous(hp_Cb, &kpt2, &jacobian);
    const Eigen::Matrix2d Um = jacobian * P_C * jacobian.transpose();
    uv[2 * i] = kpt2[0], uv[2 * i + 1] = kpt2[1];
    for (int r = 0; r < 2; ++r)
      for (int cc = 0; cc < 2; ++cc) U[4 * i + 2 * r + cc] = Um(r, cc);
  }
  return 0;
}

}  // namespace

extern "C" {

int ref_fe_stereo_triangulate(const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b, const double* T_AB, const double* UOplus,
                              int n_a, const float* kp_a, int n_b, const float* kp_b, int n_pairs, const int32_t* pairs,
                              const double* sigma_ray, int want_uncertainty, double* hp_a, double* cov, uint8_t* flags) {
  if (cam_a->model != cam_b->model) return -1;  // the triangulator is a template of ONE camera geometry
  try {
    switch (cam_a->model) {
      case OKVIS_BA_DIST_RADTAN:
        return triangulate<cam::RadialTangentialDistortion>(cam::NCameraSystem::RadialTangential, cam_a, cam_b, T_AB, UOplus, n_a, kp_a,
                                                            n_b, kp_b, n_pairs, pairs, sigma_ray, want_uncertainty, hp_a, cov, flags);
      case OKVIS_BA_DIST_EQUIDISTANT:
        return triangulate<cam::EquidistantDistortion>(cam::NCameraSystem::Equidistant, cam_a, cam_b, T_AB, UOplus, n_a, kp_a, n_b,
                                                       kp_b, n_pairs, pairs, sigma_ray, want_uncertainty, hp_a, cov, flags);
      case OKVIS_BA_DIST_RADTAN8:
        return triangulate<cam::RadialTangentialDistortion8>(cam::NCameraSystem::RadialTangential8, cam_a, cam_b, T_AB, UOplus, n_a,
                                                             kp_a, n_b, kp_b, n_pairs, pairs, sigma_ray, want_uncertainty, hp_a, cov,
                                                             flags);
      default:
        return -2;  // the reference instantiates the triangulator for these three only (ProbabilisticStereoTriangulator.cpp:387-392)
    }
  } catch (const std::exception&) {
    return -3;
  }
}

int ref_fe_project_landmarks(const okvis_fe_camera* c, const double* T_CbW, const double* P3, int n, const double* hp_W, double* uv,
                             double* U, uint8_t* status) {
  switch (c->model) {
    case OKVIS_BA_DIST_NONE: return project<cam::NoDistortion>(c, T_CbW, P3, n, hp_W, uv, U, status);
    case OKVIS_BA_DIST_RADTAN: return project<cam::RadialTangentialDistortion>(c, T_CbW, P3, n, hp_W, uv, U, status);
    case OKVIS_BA_DIST_EQUIDISTANT: return project<cam::EquidistantDistortion>(c, T_CbW, P3, n, hp_W, uv, U, status);
    default: return project<cam::RadialTangentialDistortion8>(c, T_CbW, P3, n, hp_W, uv, U, status);
  }
}

// verifyMatch (:320-337) and the gate of setBestMatch (:494-512) in the reference's own expressions
int ref_fe_gate_3d2d(int /*n_proj*/, const double* uv, const double* U, int /*n_b*/, const float* kp_b, int n_pairs, const int32_t* pairs,
                     double* chi2_out, uint8_t* flags) {
  for (int i = 0; i < n_pairs; ++i) {
    const int indexA = pairs[2 * i], indexB = pairs[2 * i + 1];
    Eigen::Vector2d kptB(uv[2 * indexA], uv[2 * indexA + 1]);
    double keypointBStdDev = kp_b[3 * indexB + 2];
    keypointBStdDev = 0.8 * keypointBStdDev / 12.0;
    Eigen::Matrix2d Up;
    Up(0, 0) = U[4 * indexA], Up(0, 1) = U[4 * indexA + 1], Up(1, 0) = U[4 * indexA + 2], Up(1, 1) = U[4 * indexA + 3];
    Eigen::Matrix2d U_tot = Eigen::Matrix2d::Identity() * keypointBStdDev * keypointBStdDev + Up;
    Eigen::Vector2d keypointBMeasurement(kp_b[3 * indexB], kp_b[3 * indexB + 1]);
    Eigen::Vector2d err = kptB - keypointBMeasurement;
    const double chi2 = err.transpose() * U_tot.inverse() * err;
    const int chi2_int = chi2;  // verifyMatch declares `const int chi2`
    unsigned f = 0;
    if (chi2_int < 4.0) f |= OKVIS_FE_GATE_VERIFIED;
    if (!(chi2 > 4.0)) f |= OKVIS_FE_GATE_ACCEPTED;
    if (U_tot.norm() > 25.0 / (keypointBStdDev * keypointBStdDev * sqrt(2))) f |= OKVIS_FE_GATE_UNCERTAIN;
    chi2_out[i] = chi2;
    flags[i] = (uint8_t)f;
  }
  return 0;
}

}  // extern "C"
