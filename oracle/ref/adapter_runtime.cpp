// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Runtime check of the okvis::Estimator drop-in (okvis_amd/csrc/host/okvis_estimator_adapter.hpp) compiled against the
// reference's REAL okvis headers (VioBackendInterface.hpp, MultiFrame.hpp, Frame.hpp, NCameraSystem.hpp, Parameters.hpp,
// Measurements.hpp, Transformation.hpp from /root/reference) with the stand-in Eigen / glog / OpenCV / ceres headers of
// oracle/shim.  It drives the class exactly the way okvis_multisensor_processing does (ThreadedKFVio.cpp:501-533,736-765):
// addStates(MultiFramePtr, ImuMeasurementDeque, asKeyframe) -> addLandmark / addObservation<GEOMETRY> -> optimize ->
// applyMarginalizationStrategy(numKeyframes, numImuFrames, removed), through the abstract VioBackendInterface where the
// caller does, over a sliding window in which frames and landmarks ARE marginalised.  Built by oracle/ref/Makefile into
// oracle/_ref/adapter_runtime (travels to the GPU box); tests/test_gpu_adapter.py runs it and checks the exit code.
#include <cmath>
#include <cstdio>
#include <memory>
#include <random>
#include <sstream>
#include <vector>

#include "okvis_estimator_adapter.hpp"
#ifndef OKVIS_AMD_HAVE_OKVIS
#error "the okvis headers were not found"
#endif
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/NCameraSystem.hpp>
#include <okvis/cameras/PinholeCamera.hpp>

namespace google {
int eshim_log_warnings = 0;
}

typedef okvis::cameras::PinholeCamera<okvis::cameras::EquidistantDistortion> Camera;

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  std::normal_distribution<double> nrm(0.0, 1.0);
  const double IMU_RATE = 100.0, FRAME_DT = 0.5, DT = 1.0 / IMU_RATE;
  const int N_FRAMES = 14;
  okvis::ImuParameters imu;
  imu.a_max = 1000.0, imu.g_max = 1000.0, imu.sigma_g_c = 6.0e-4, imu.sigma_a_c = 2.0e-3, imu.sigma_bg = 0.03;
  imu.sigma_ba = 0.1, imu.sigma_gw_c = 3.0e-6, imu.sigma_aw_c = 2.0e-5, imu.tau = 3600.0, imu.g = 9.81;
  imu.a0 = Eigen::Vector3d(0, 0, 0);
  imu.rate = 100;

  // two equidistant cameras, 0.1 m baseline (TestEstimator.cpp:77-96)
  okvis::cameras::NCameraSystem ncs;
  std::vector<std::shared_ptr<const okvis::kinematics::Transformation> > T_SC;
  std::shared_ptr<const Camera> geometry(
      new Camera(752, 480, 350, 360, 378, 238, okvis::cameras::EquidistantDistortion(-0.21, 0.14, 0.0006, 0.0003)));
  for (int i = 0; i < 2; ++i) {
    T_SC.push_back(std::shared_ptr<const okvis::kinematics::Transformation>(
        new okvis::kinematics::Transformation(Eigen::Vector3d(0, 0.1 * i, 0), Eigen::Quaterniond(1, 0, 0, 0))));
    ncs.addCamera(T_SC[i], geometry, okvis::cameras::NCameraSystem::Equidistant, false);
  }

  okvis::Estimator estimator;
  okvis::VioBackendInterface& backend = estimator;  // what ThreadedKFVio / Frontend see
  okvis::ExtrinsicsEstimationParameters ext(0, 0, 0, 0);
  backend.addCamera(ext);
  backend.addCamera(ext);
  backend.addImu(imu);

  // IMU stream (constant velocity along y) and a wall of landmarks at x = 3 m
  const Eigen::Vector3d speed(0, 1, 0);
  okvis::ImuMeasurementDeque stream;
  const int n_imu = (int)(N_FRAMES * FRAME_DT * IMU_RATE) + 2;
  for (int i = 0; i < n_imu; ++i) {
    const Eigen::Vector3d gyr(uni(rng) * imu.sigma_g_c * std::sqrt(DT), uni(rng) * imu.sigma_g_c * std::sqrt(DT),
                              uni(rng) * imu.sigma_g_c * std::sqrt(DT));
    const Eigen::Vector3d acc(uni(rng) * imu.sigma_a_c * std::sqrt(DT), uni(rng) * imu.sigma_a_c * std::sqrt(DT),
                              imu.g + uni(rng) * imu.sigma_a_c * std::sqrt(DT));
    stream.push_back(okvis::ImuMeasurement(okvis::Time(1, 0) + okvis::Duration(i * DT), okvis::ImuSensorReadings(gyr, acc)));
  }
  std::vector<Eigen::Vector4d> pts;
  for (double y = -6.0; y < N_FRAMES * FRAME_DT + 6.0; y += 0.75)
    for (double z = -6.0; z <= 6.0 + 1e-9; z += 0.75) pts.push_back(Eigen::Vector4d(3.0, y, z, 1.0));
  std::vector<bool> added(pts.size(), false), gone(pts.size(), false);

  size_t removed_total = 0;
  uint64_t last_id = 0;
  for (int k = 0; k < N_FRAMES; ++k) {
    const okvis::Time t_k = okvis::Time(1, 0) + okvis::Duration(k * FRAME_DT);
    const Eigen::Vector3d r_k = speed * (k * FRAME_DT);
    okvis::MultiFramePtr mf(new okvis::MultiFrame(ncs, t_k, 100 + k));
    okvis::ImuMeasurementDeque d;
    for (const okvis::ImuMeasurement& m : stream)
      if (m.timeStamp >= (k ? t_k - okvis::Duration(FRAME_DT + 0.02) : t_k - okvis::Duration(0.02)) &&
          m.timeStamp <= t_k + okvis::Duration(0.03))
        d.push_back(m);
    if (!backend.addStates(mf, d, k % 3 == 0)) return std::printf("addStates failed at frame %d\n", k), 2;
    last_id = mf->id();
    // keypoints of both images first (the frontend detects, then matches)
    std::vector<std::vector<cv::KeyPoint> > kps(2);
    std::vector<std::vector<size_t> > which(2);
    for (size_t i = 0; i < 2; ++i) {
      for (size_t j = 0; j < pts.size(); ++j) {
        if (gone[j] || std::fabs(pts[j][1] - r_k[1]) >= 5.0) continue;
        const Eigen::Vector3d p_C = pts[j].head<3>() - r_k - T_SC[i]->r();
        Eigen::Vector2d uv;
        if (geometry->project(p_C, &uv) != okvis::cameras::CameraBase::ProjectionStatus::Successful) continue;
        kps[i].push_back(cv::KeyPoint((float)(uv[0] + uni(rng)), (float)(uv[1] + uni(rng)), 8.0f));
        which[i].push_back(j);
      }
      mf->resetKeypoints(i, kps[i]);
    }
    size_t n_obs = 0;
    for (size_t i = 0; i < 2; ++i)
      for (size_t q = 0; q < which[i].size(); ++q) {
        const size_t j = which[i][q];
        if (!added[j]) {
          Eigen::Vector4d hp = pts[j];
          hp.head<3>() += Eigen::Vector3d(nrm(rng), nrm(rng), nrm(rng)) * 0.05;
          if (!backend.addLandmark(5000 + j, hp)) return std::printf("addLandmark failed\n"), 3;
          added[j] = true;
        }
        if (estimator.addObservation<Camera>(5000 + j, mf->id(), i, q) == 0) return std::printf("addObservation failed\n"), 4;
        ++n_obs;
      }
    backend.optimize(5, 2, false);
    okvis::MapPointVector removed;
    if (!estimator.applyMarginalizationStrategy(3, 3, removed)) return std::printf("applyMarginalizationStrategy failed\n"), 5;
    for (const okvis::MapPoint& mp : removed) gone[(size_t)(mp.id - 5000)] = true;
    removed_total += removed.size();
    if (backend.numFrames() > 6) return std::printf("window has %zu frames\n", backend.numFrames()), 6;
    std::ostringstream ss;
    estimator.printStates(mf->id(), ss);
    if (ss.str().find("SpeedAndBiasParameterBlock") == std::string::npos) return std::printf("printStates: %s\n", ss.str().c_str()), 7;
    std::printf("frame %2d: %3zu observations, %zu frames, %zu landmarks, %zu removed\n", k, n_obs, backend.numFrames(),
                backend.numLandmarks(), removed.size());
  }
  okvis::kinematics::Transformation T;
  okvis::SpeedAndBias sb;
  if (!backend.get_T_WS(last_id, T) || !backend.getSpeedAndBias(last_id, 0, sb)) return 8;
  const Eigen::Vector3d r_true = speed * ((N_FRAMES - 1) * FRAME_DT);
  const double e_t = (T.r() - r_true).norm(), e_r = 2 * T.q().vec().norm(), e_v = (sb.head<3>() - speed).norm();
  std::printf("final: |dt| %.3e m, |dalpha| %.3e rad, |dv| %.3e m/s, %zu landmarks removed in total\n", e_t, e_r, e_v, removed_total);
  // TestEstimator.cpp:229-236 tolerances; frames and landmarks must have been marginalised on the way
  if (!(e_t < 1e-1 && e_r < 1e-2 && e_v < 0.04)) return 9;
  if (removed_total == 0) return 10;
  std::printf("ADAPTER RUNTIME OK\n");
  return 0;
}
