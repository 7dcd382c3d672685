// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Flat C wrapper around the okvis reference's OWN okvis::Estimator (okvis_ceres/src/Estimator.cpp, compiled unmodified
// by oracle/ref/Makefile together with Map.cpp, MarginalizationError.cpp, the error terms, MultiFrame / Frame /
// NCameraSystem against the stand-in Eigen / Ceres / glog / OpenCV headers of oracle/shim).  Same entry points, argument
// meaning and return values as okvis_amd/csrc/host/estimator_capi.cpp (prefix ref_est_ instead of okvis_est_), so that
// tests/test_gpu_estimator_vs_reference.py can drive the reference class and the MI355X backend with identical call
// sequences (addStates / addLandmark / addObservation / optimize / applyMarginalizationStrategy ...) and compare the states
// after every frame.  ::ceres::Solve behind Estimator::optimize is oracle/ref/ceres_shim_solve.cpp (this repository's
// statement of the DOGLEG policy: Ceres itself is not available); everything else that runs is reference code.
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <okvis/Estimator.hpp>
#include <okvis/MultiFrame.hpp>
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/NCameraSystem.hpp>
#include <okvis/cameras/NoDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/cameras/RadialTangentialDistortion8.hpp>
#include <okvis/ceres/ImuError.hpp>

#include "okvis_amd_ba.h"

namespace cam = okvis::cameras;
using okvis::kinematics::Transformation;

namespace {
thread_local std::string g_err;
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
okvis::Time to_time(int64_t ns) { return okvis::Time((uint32_t)(ns / 1000000000LL), (uint32_t)(ns % 1000000000LL)); }
Transformation to_T(const double* p) {
  Transformation T;
  Eigen::Matrix<double, 7, 1> c;
  for (int i = 0; i < 7; ++i) c[i] = p[i];
  T.setCoeffs(c);
  return T;
}
void from_T(const Transformation& T, double* out) {
  for (int i = 0; i < 7; ++i) out[i] = T.coeffs()[i];
}
okvis::ImuParameters imu_params(const double prm[13]) {
  okvis::ImuParameters p;
  p.a_max = prm[0], p.g_max = prm[1], p.sigma_g_c = prm[2], p.sigma_a_c = prm[3], p.sigma_bg = prm[4];
  p.sigma_ba = prm[5], p.sigma_gw_c = prm[6], p.sigma_aw_c = prm[7], p.tau = prm[8], p.g = prm[9];
  p.a0 = Eigen::Vector3d(prm[10], prm[11], prm[12]);
  p.rate = 200;
  return p;
}
okvis::ImuMeasurementDeque imu_deque(int n, const int64_t* t, const double* gyr, const double* acc) {
  okvis::ImuMeasurementDeque d;
  for (int i = 0; i < n; ++i)
    d.push_back(okvis::ImuMeasurement(
        to_time(t[i]), okvis::ImuSensorReadings(Eigen::Vector3d(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]),
                                                Eigen::Vector3d(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]))));
  return d;
}

struct RefFrame {
  okvis::MultiFramePtr mf;
  std::vector<int> models;
  std::vector<std::vector<cv::KeyPoint> > kps;
};
struct RefEst {
  std::shared_ptr<okvis::ceres::Map> map;
  std::unique_ptr<okvis::Estimator> est;
  std::map<uint64_t, std::vector<int> > models;  // frame id -> distortion model of each camera
  // the marginalisation prior as it sits in the Map (Estimator keeps its pointer private)
  std::shared_ptr<okvis::ceres::ErrorInterface> prior() const {
    for (const auto& kv : map->residualBlockId2ResidualBlockSpecMap())
      if (kv.second.errorInterfacePtr->typeInfo() == "MarginalizationError") return kv.second.errorInterfacePtr;
    return std::shared_ptr<okvis::ceres::ErrorInterface>();
  }
};
}  // namespace

extern "C" {

const char* ref_est_last_error() { return g_err.c_str(); }

void* ref_est_create(int /*device*/) {
  try {
    RefEst* r = new RefEst();
    r->map.reset(new okvis::ceres::Map());
    r->est.reset(new okvis::Estimator(r->map));
    return r;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void ref_est_destroy(void* h) { delete static_cast<RefEst*>(h); }

int ref_est_add_camera(void* h, const double s[4]) {
  return guarded([&] {
    return static_cast<RefEst*>(h)->est->addCamera(okvis::ExtrinsicsEstimationParameters(s[0], s[1], s[2], s[3]));
  });
}
int ref_est_add_imu(void* h, const double prm[13]) {
  return guarded([&] { return static_cast<RefEst*>(h)->est->addImu(imu_params(prm)); });
}

void* ref_est_frame_create(uint64_t id, int64_t t_ns, int ncam, const double* T_SC, const double* intr, const int* models) {
  RefFrame* f = new RefFrame();
  cam::NCameraSystem ncs;
  const int W = 752, H = 480;
  for (int c = 0; c < ncam; ++c) {
    std::shared_ptr<const Transformation> T(new Transformation(to_T(T_SC + 7 * c)));
    const double* k = intr + 12 * c;
    std::shared_ptr<const cam::CameraBase> g;
    cam::NCameraSystem::DistortionType dt = cam::NCameraSystem::RadialTangential;
    switch (models[c]) {
      case OKVIS_BA_DIST_NONE:
        g.reset(new cam::PinholeCamera<cam::NoDistortion>(W, H, k[0], k[1], k[2], k[3], cam::NoDistortion()));
        dt = cam::NCameraSystem::NoDistortion;
        break;
      case OKVIS_BA_DIST_RADTAN:
        g.reset(new cam::PinholeCamera<cam::RadialTangentialDistortion>(
            W, H, k[0], k[1], k[2], k[3], cam::RadialTangentialDistortion(k[4], k[5], k[6], k[7])));
        break;
      case OKVIS_BA_DIST_EQUIDISTANT:
        g.reset(new cam::PinholeCamera<cam::EquidistantDistortion>(W, H, k[0], k[1], k[2], k[3],
                                                                   cam::EquidistantDistortion(k[4], k[5], k[6], k[7])));
        dt = cam::NCameraSystem::Equidistant;
        break;
      default:
        g.reset(new cam::PinholeCamera<cam::RadialTangentialDistortion8>(
            W, H, k[0], k[1], k[2], k[3],
            cam::RadialTangentialDistortion8(k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11])));
        dt = cam::NCameraSystem::RadialTangential8;
    }
    ncs.addCamera(T, g, dt, false);
    f->models.push_back(models[c]);
    f->kps.emplace_back();
  }
  f->mf.reset(new okvis::MultiFrame(ncs, to_time(t_ns), id));
  return f;
}
void ref_est_frame_destroy(void* f) { delete static_cast<RefFrame*>(f); }
int ref_est_frame_add_keypoint(void* fp, int cam_idx, float x, float y, float size) {
  RefFrame* f = static_cast<RefFrame*>(fp);
  f->kps[(size_t)cam_idx].push_back(cv::KeyPoint(x, y, size));
  f->mf->resetKeypoints((size_t)cam_idx, f->kps[(size_t)cam_idx]);
  return (int)f->kps[(size_t)cam_idx].size() - 1;
}

int ref_est_add_states(void* h, void* frame, int n_imu, const int64_t* t, const double* gyr, const double* acc,
                       int asKeyframe) {
  return guarded([&] {
    RefEst* r = static_cast<RefEst*>(h);
    RefFrame* f = static_cast<RefFrame*>(frame);
    r->models[f->mf->id()] = f->models;
    return r->est->addStates(f->mf, imu_deque(n_imu, t, gyr, acc), asKeyframe != 0) ? 1 : 0;
  });
}
int ref_est_add_landmark(void* h, uint64_t id, const double hp[4]) {
  return guarded(
      [&] { return static_cast<RefEst*>(h)->est->addLandmark(id, Eigen::Vector4d(hp[0], hp[1], hp[2], hp[3])) ? 1 : 0; });
}
int ref_est_add_observation(void* h, uint64_t lm, uint64_t pose, int c, int kp, uint64_t* handle) {
  return guarded([&] {
    RefEst* r = static_cast<RefEst*>(h);
    ::ceres::ResidualBlockId id = 0;
    switch (r->models.at(pose).at((size_t)c)) {
      case OKVIS_BA_DIST_NONE:
        id = r->est->addObservation<cam::PinholeCamera<cam::NoDistortion> >(lm, pose, (size_t)c, (size_t)kp);
        break;
      case OKVIS_BA_DIST_RADTAN:
        id = r->est->addObservation<cam::PinholeCamera<cam::RadialTangentialDistortion> >(lm, pose, (size_t)c, (size_t)kp);
        break;
      case OKVIS_BA_DIST_EQUIDISTANT:
        id = r->est->addObservation<cam::PinholeCamera<cam::EquidistantDistortion> >(lm, pose, (size_t)c, (size_t)kp);
        break;
      default:
        id = r->est->addObservation<cam::PinholeCamera<cam::RadialTangentialDistortion8> >(lm, pose, (size_t)c, (size_t)kp);
    }
    if (handle) *handle = reinterpret_cast<uint64_t>(id);
    return id ? 1 : 0;
  });
}
int ref_est_remove_observation(void* h, uint64_t lm, uint64_t pose, int c, int kp) {
  return guarded([&] { return static_cast<RefEst*>(h)->est->removeObservation(lm, pose, (size_t)c, (size_t)kp) ? 1 : 0; });
}
int ref_est_optimize(void* h, int numIter, int numThreads, int verbose, okvis_ba_summary* out) {
  return guarded([&] {
    RefEst* r = static_cast<RefEst*>(h);
    r->est->optimize((size_t)numIter, (size_t)numThreads, verbose != 0);
    if (out) {
      const ::ceres::Solver::Summary& s = r->map->summary;
      std::memset(out, 0, sizeof(*out));
      out->initial_cost = s.initial_cost;
      out->final_cost = s.final_cost;
      out->iterations = (int32_t)s.iterations.size() - 1;
      out->successful_steps = s.num_successful_steps;
      if (!s.iterations.empty()) out->final_radius = s.iterations.back().trust_region_radius;
    }
    return 0;
  });
}
int ref_est_set_time_limit(void* h, double limit, int minIter) {
  return guarded([&] { return static_cast<RefEst*>(h)->est->setOptimizationTimeLimit(limit, minIter) ? 1 : 0; });
}
int ref_est_apply_marginalization2(void* h, int numKeyframes, int numImuFrames, int* n_removed, uint64_t* removed_ids,
                                   int capacity) {
  return guarded([&] {
    okvis::MapPointVector removed;
    const bool ok = static_cast<RefEst*>(h)->est->applyMarginalizationStrategy((size_t)numKeyframes, (size_t)numImuFrames, removed);
    if (n_removed) *n_removed = (int)removed.size();
    for (int i = 0; i < capacity && i < (int)removed.size(); ++i) removed_ids[i] = removed[(size_t)i].id;
    return ok ? 1 : 0;
  });
}
int ref_est_apply_marginalization(void* h, int numKeyframes, int numImuFrames) {
  return ref_est_apply_marginalization2(h, numKeyframes, numImuFrames, 0, 0, 0);
}
int ref_est_get_T_WS(void* h, uint64_t id, double out[7]) {
  return guarded([&] {
    Transformation T;
    if (!static_cast<RefEst*>(h)->est->get_T_WS(id, T)) return 0;
    from_T(T, out);
    return 1;
  });
}
int ref_est_get_speed_and_bias(void* h, uint64_t id, double out[9]) {
  return guarded([&] {
    okvis::SpeedAndBias sb;
    if (!static_cast<RefEst*>(h)->est->getSpeedAndBias(id, 0, sb)) return 0;
    for (int i = 0; i < 9; ++i) out[i] = sb[i];
    return 1;
  });
}
int ref_est_get_extrinsics(void* h, uint64_t id, int c, double out[7]) {
  return guarded([&] {
    Transformation T;
    if (!static_cast<RefEst*>(h)->est->getCameraSensorStates(id, (size_t)c, T)) return 0;
    from_T(T, out);
    return 1;
  });
}
int ref_est_get_landmark(void* h, uint64_t id, double point[4], double* quality, int* n_obs) {
  return guarded([&] {
    okvis::MapPoint mp;
    static_cast<RefEst*>(h)->est->getLandmark(id, mp);
    for (int i = 0; i < 4; ++i) point[i] = mp.point[i];
    if (quality) *quality = mp.quality;
    if (n_obs) *n_obs = (int)mp.observations.size();
    return 1;
  });
}
int ref_est_prior_info(void* h, int* dim, int* nblocks) {
  return guarded([&] {
    std::shared_ptr<okvis::ceres::ErrorInterface> p = static_cast<RefEst*>(h)->prior();
    *dim = p ? (int)p->residualDim() : 0;
    *nblocks = p ? (int)p->parameterBlocks() : 0;
    return 1;
  });
}
int ref_est_frame_id_by_age(void* h, int age, uint64_t* id) {
  return guarded([&] {
    *id = static_cast<RefEst*>(h)->est->frameIdByAge((size_t)age);
    return 1;
  });
}
int ref_est_is_keyframe(void* h, uint64_t id) {
  return guarded([&] { return static_cast<RefEst*>(h)->est->isKeyframe(id) ? 1 : 0; });
}
int ref_est_is_in_imu_window(void* h, uint64_t id) {
  return guarded([&] { return static_cast<RefEst*>(h)->est->isInImuWindow(id) ? 1 : 0; });
}
int ref_est_last_marg_info(void*, double out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
  return 1;
}
int ref_est_last_timings(void*, double out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = 0;
  return 1;
}
int ref_est_set_use_graph(void*, int) { return 1; }
int ref_est_num_frames(void* h) { return (int)static_cast<RefEst*>(h)->est->numFrames(); }
int ref_est_num_landmarks(void* h) { return (int)static_cast<RefEst*>(h)->est->numLandmarks(); }
int ref_est_current_frame_id(void* h, uint64_t* id) {
  return guarded([&] {
    *id = static_cast<RefEst*>(h)->est->currentFrameId();
    return 1;
  });
}
int ref_est_init_pose_from_imu(int n, const double* acc, double out[7]) {
  okvis::ImuMeasurementDeque d;
  for (int i = 0; i < n; ++i)
    d.push_back(okvis::ImuMeasurement(okvis::Time(0, (uint32_t)i),
                                      okvis::ImuSensorReadings(Eigen::Vector3d(0, 0, 0),
                                                               Eigen::Vector3d(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]))));
  Transformation T;
  const bool ok = okvis::Estimator::initPoseFromImu(d, T);
  from_T(T, out);
  return ok ? 1 : 0;
}
int ref_est_propagation(int n, const int64_t* t, const double* gyr, const double* acc, const double prm[13], double T_WS[7],
                        double sb[9], int64_t t_start, int64_t t_end) {
  Transformation T = to_T(T_WS);
  okvis::SpeedAndBias s;
  for (int i = 0; i < 9; ++i) s[i] = sb[i];
  const int r = okvis::ceres::ImuError::propagation(imu_deque(n, t, gyr, acc), imu_params(prm), T, s, to_time(t_start),
                                                    to_time(t_end), 0, 0);
  from_T(T, T_WS);
  for (int i = 0; i < 9; ++i) sb[i] = s[i];
  return r;
}

}  // extern "C"
