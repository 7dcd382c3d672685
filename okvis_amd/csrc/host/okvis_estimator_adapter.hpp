// okvis::Estimator adapter — source-compatible drop-in for okvis_ceres/include/okvis/Estimator.hpp.
//
// Converts the Eigen / OKVIS types of the reference's public signatures to the PODs of
// okvis_amd::Estimator (estimator.hpp).  It needs Eigen, OpenCV and the OKVIS headers.  In this repository it is
// (a) compiled against the reference's REAL okvis headers (VioBackendInterface.hpp, MultiFrame.hpp, Frame.hpp,
// NCameraSystem.hpp, Parameters.hpp ...) with the stand-in Eigen / glog / OpenCV / ceres headers of oracle/shim, and
// RUN on the GPU through a sliding window with marginalisation (oracle/ref/adapter_runtime.cpp, tests/test_gpu_adapter.py),
// (b) compile-checked against the minimal stand-in declarations of tests/mock_okvis/ (CPU suite, no reference tree needed).
// See INTEGRATION.md for how a maintainer wires it into okvis_ceres.
#pragma once
#include "estimator.hpp"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<okvis/VioBackendInterface.hpp>)
#define OKVIS_AMD_HAVE_OKVIS 1
#endif
#endif

#ifdef OKVIS_AMD_HAVE_OKVIS
#include <Eigen/Core>
#include <okvis/FrameTypedefs.hpp>
#include <okvis/Measurements.hpp>
#include <okvis/MultiFrame.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/VioBackendInterface.hpp>
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/cameras/RadialTangentialDistortion8.hpp>
#include <okvis/kinematics/Transformation.hpp>

namespace okvis {

/// Same public surface as the reference class (Estimator.hpp:77-581); the body forwards to the GPU backend.
class Estimator : public VioBackendInterface {
 public:
  OKVIS_DEFINE_EXCEPTION(Exception, std::runtime_error)
  Estimator() : impl_(0) {}
  explicit Estimator(std::shared_ptr<okvis::ceres::Map>) : impl_(0) {}  // the Map is not used any more
  virtual ~Estimator() {}

  int addCamera(const ExtrinsicsEstimationParameters& p) override {
    okvis_amd::ExtrinsicsEstimationParameters q;
    q.sigma_absolute_translation = p.sigma_absolute_translation;
    q.sigma_absolute_orientation = p.sigma_absolute_orientation;
    q.sigma_c_relative_translation = p.sigma_c_relative_translation;
    q.sigma_c_relative_orientation = p.sigma_c_relative_orientation;
    return impl_.addCamera(q);
  }
  int addImu(const ImuParameters& p) override {
    okvis_amd::ImuParameters q;
    q.a_max = p.a_max; q.g_max = p.g_max; q.sigma_g_c = p.sigma_g_c; q.sigma_a_c = p.sigma_a_c;
    q.sigma_bg = p.sigma_bg; q.sigma_ba = p.sigma_ba; q.sigma_gw_c = p.sigma_gw_c; q.sigma_aw_c = p.sigma_aw_c;
    q.tau = p.tau; q.g = p.g; q.a0 = {{p.a0[0], p.a0[1], p.a0[2]}}; q.rate = p.rate;
    return impl_.addImu(q);
  }
  void clearCameras() override { impl_.clearCameras(); }
  void clearImus() override { impl_.clearImus(); }

  bool addStates(okvis::MultiFramePtr multiFrame, const okvis::ImuMeasurementDeque& imuMeasurements,
                 bool asKeyframe) override {
    auto mf = std::make_shared<okvis_amd::MultiFrame>();
    mf->id = multiFrame->id();
    mf->t_ns = toNs(multiFrame->timestamp());
    for (size_t i = 0; i < multiFrame->numFrames(); ++i) {
      mf->T_SC.push_back(toPod(*multiFrame->T_SC(i)));
      mf->geometry.push_back(geometryOf(*multiFrame, i));
      mf->keypoints.emplace_back();
    }
    frames_[mf->id] = multiFrame;
    pods_[mf->id] = mf;
    return impl_.addStates(mf, toPod(imuMeasurements), asKeyframe);
  }
  bool addLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) override {
    return impl_.addLandmark(landmarkId, {{landmark[0], landmark[1], landmark[2], landmark[3]}});
  }
  template <class GEOMETRY_TYPE>
  ::ceres::ResidualBlockId addObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) {
    // keypoints are read lazily from the cv::KeyPoint storage of the frame (implementation/Estimator.hpp:57-65)
    auto& pod = pods_.at(poseId);
    auto& mf = frames_.at(poseId);
    while (pod->keypoints[camIdx].size() <= keypointIdx) {
      const size_t k = pod->keypoints[camIdx].size();
      Eigen::Vector2d kp;
      double size = 1.0;
      mf->getKeypoint(camIdx, k, kp);
      mf->getKeypointSize(camIdx, k, size);
      pod->keypoints[camIdx].push_back(okvis_amd::Keypoint{(float)kp[0], (float)kp[1], (float)size});
    }
    return reinterpret_cast<::ceres::ResidualBlockId>(impl_.addObservation(landmarkId, poseId, camIdx, keypointIdx));
  }
  bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) override {
    return impl_.removeObservation(landmarkId, poseId, camIdx, keypointIdx);
  }
  bool applyMarginalizationStrategy(size_t numKeyframes, size_t numImuFrames, okvis::MapPointVector& removed) {  // Estimator.hpp:183
    okvis_amd::MapPointVector r;
    const bool ok = impl_.applyMarginalizationStrategy(numKeyframes, numImuFrames, r);
    for (const okvis_amd::MapPoint& m : r) removed.push_back(toOkvis(m));
    releaseMarginalizedFrames();
    return ok;
  }
  void optimize(size_t numIter, size_t numThreads = 1, bool verbose = false) override {
    impl_.optimize(numIter, numThreads, verbose);
  }
  bool setOptimizationTimeLimit(double timeLimit, int minIterations) override {
    return impl_.setOptimizationTimeLimit(timeLimit, minIterations);
  }
  bool get_T_WS(uint64_t poseId, okvis::kinematics::Transformation& T_WS) const override {
    okvis_amd::Transformation T;
    if (!impl_.get_T_WS(poseId, T)) return false;
    T_WS = fromPod(T);
    return true;
  }
  bool getSpeedAndBias(uint64_t poseId, uint64_t imuIdx, okvis::SpeedAndBias& sb) const override {
    okvis_amd::SpeedAndBias s;
    if (!impl_.getSpeedAndBias(poseId, imuIdx, s)) return false;
    for (int i = 0; i < 9; ++i) sb[i] = s[i];
    return true;
  }
  void printStates(uint64_t poseId, std::ostream& buffer) const { impl_.printStates(poseId, buffer); }   // Estimator.hpp:141
  size_t numFrames() const override { return impl_.numFrames(); }
  size_t numLandmarks() const override { return impl_.numLandmarks(); }
  uint64_t currentKeyframeId() const { return impl_.currentKeyframeId(); }
  uint64_t frameIdByAge(size_t age) const { return impl_.frameIdByAge(age); }
  uint64_t currentFrameId() const override { return impl_.currentFrameId(); }
  bool isKeyframe(uint64_t frameId) const override { return impl_.isKeyframe(frameId); }
  bool isInImuWindow(uint64_t frameId) const { return impl_.isInImuWindow(frameId); }
  okvis::Time timestamp(uint64_t frameId) const override {
    const int64_t t = impl_.timestamp(frameId);
    return okvis::Time((uint32_t)(t / 1000000000LL), (uint32_t)(t % 1000000000LL));
  }
  static bool initPoseFromImu(const okvis::ImuMeasurementDeque& imuMeasurements, okvis::kinematics::Transformation& T_WS) {
    okvis_amd::Transformation T;
    const bool ok = okvis_amd::Estimator::initPoseFromImu(toPod(imuMeasurements), T);
    if (ok) T_WS = fromPod(T);
    return ok;
  }
  bool isLandmarkAdded(uint64_t landmarkId) const override { return impl_.isLandmarkAdded(landmarkId); }
  bool isLandmarkInitialized(uint64_t landmarkId) const override { return impl_.isLandmarkInitialized(landmarkId); }
  bool getLandmark(uint64_t landmarkId, okvis::MapPoint& mapPoint) const override {
    okvis_amd::MapPoint m;
    if (!impl_.getLandmark(landmarkId, m)) return false;
    mapPoint = toOkvis(m);
    return true;
  }
  size_t getLandmarks(okvis::PointMap& landmarks) const override {
    okvis_amd::PointMap pm;
    impl_.getLandmarks(pm);
    landmarks.clear();
    for (const auto& kv : pm) landmarks.insert(std::make_pair(kv.first, toOkvis(kv.second)));
    return landmarks.size();
  }
  size_t getLandmarks(okvis::MapPointVector& landmarks) const {  // Estimator.hpp (not part of VioBackendInterface)
    okvis_amd::MapPointVector v;
    impl_.getLandmarks(v);
    landmarks.clear();
    for (const auto& m : v) landmarks.push_back(toOkvis(m));
    return landmarks.size();
  }
  okvis::MultiFramePtr multiFrame(uint64_t frameId) const override {
    auto it = frames_.find(frameId);
    return it == frames_.end() ? okvis::MultiFramePtr() : it->second;   // dropped when the frame is marginalised
  }
  bool getCameraSensorStates(uint64_t poseId, size_t cameraIdx, okvis::kinematics::Transformation& T_SCi) const override {
    okvis_amd::Transformation T;
    if (!impl_.getCameraSensorStates(poseId, cameraIdx, T)) return false;
    T_SCi = fromPod(T);
    return true;
  }
  bool set_T_WS(uint64_t poseId, const okvis::kinematics::Transformation& T_WS) override {
    return impl_.set_T_WS(poseId, toPod(T_WS));
  }
  bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const okvis::SpeedAndBias& sb) override {
    okvis_amd::SpeedAndBias s;
    for (int i = 0; i < 9; ++i) s[i] = sb[i];
    return impl_.setSpeedAndBias(poseId, imuIdx, s);
  }
  bool setCameraSensorStates(uint64_t poseId, size_t cameraIdx, const okvis::kinematics::Transformation& T_SCi) override {
    return impl_.setCameraSensorStates(poseId, cameraIdx, toPod(T_SCi));
  }
  bool setLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) override {
    return impl_.setLandmark(landmarkId, {{landmark[0], landmark[1], landmark[2], landmark[3]}});
  }
  void setLandmarkInitialized(uint64_t landmarkId, bool initialized) override {
    impl_.setLandmarkInitialized(landmarkId, initialized);
  }
  void setKeyframe(uint64_t frameId, bool isKeyframe) override { impl_.setKeyframe(frameId, isKeyframe); }
  void setMap(std::shared_ptr<okvis::ceres::Map>) override {}   // VioBackendInterface.hpp:329 — no Ceres graph behind this backend
  // frames whose states were marginalised are released like Estimator.cpp:730 does (multiFramePtrMap_.erase)
  void releaseMarginalizedFrames() {
    for (auto it = frames_.begin(); it != frames_.end();) {
      if (impl_.hasFrame(it->first)) {
        ++it;
      } else {
        pods_.erase(it->first);
        it = frames_.erase(it);
      }
    }
  }

 private:
  static int64_t toNs(const okvis::Time& t) { return (int64_t)t.sec * 1000000000LL + (int64_t)t.nsec; }
  static okvis::MapPoint toOkvis(const okvis_amd::MapPoint& m) {  // okvis::MapPoint(id, point, quality, distance), FrameTypedefs.hpp
    okvis::MapPoint mp(m.id, Eigen::Vector4d(m.point[0], m.point[1], m.point[2], m.point[3]), m.quality, m.distance);
    for (const auto& ob : m.observations)
      mp.observations.insert(std::make_pair(
          okvis::KeypointIdentifier(ob.first.frameId, ob.first.cameraIndex, ob.first.keypointIndex), ob.second));
    return mp;
  }
  static okvis_amd::ImuMeasurementDeque toPod(const okvis::ImuMeasurementDeque& imuMeasurements) {
    okvis_amd::ImuMeasurementDeque d;
    d.reserve(imuMeasurements.size());
    for (const auto& m : imuMeasurements) {
      okvis_amd::ImuMeasurement q;
      q.t_ns = toNs(m.timeStamp);
      for (int c = 0; c < 3; ++c) {
        q.gyr[c] = m.measurement.gyroscopes[c];
        q.acc[c] = m.measurement.accelerometers[c];
      }
      d.push_back(q);
    }
    return d;
  }
  static okvis_amd::Transformation toPod(const okvis::kinematics::Transformation& T) {
    okvis_amd::Transformation P;
    const Eigen::Matrix<double, 7, 1> c = T.coeffs();
    for (int i = 0; i < 7; ++i) P.p[i] = c[i];
    return P;
  }
  static okvis::kinematics::Transformation fromPod(const okvis_amd::Transformation& P) {
    return okvis::kinematics::Transformation(Eigen::Vector3d(P.p[0], P.p[1], P.p[2]),
                                             Eigen::Quaterniond(P.p[6], P.p[3], P.p[4], P.p[5]));
  }
  static okvis_amd::CameraGeometry geometryOf(const okvis::MultiFrame& mf, size_t i) {
    okvis_amd::CameraGeometry g;
    Eigen::VectorXd intr;
    mf.geometry(i)->getIntrinsics(intr);  // fu fv cu cv + distortion coefficients
    for (int k = 0; k < intr.size() && k < 12; ++k) g.intr[k] = intr[k];
    const std::string d = mf.geometry(i)->distortionType();
    g.model = d == "RadialTangentialDistortion"    ? OKVIS_BA_DIST_RADTAN
              : d == "EquidistantDistortion"       ? OKVIS_BA_DIST_EQUIDISTANT
              : d == "RadialTangentialDistortion8" ? OKVIS_BA_DIST_RADTAN8
                                                   : OKVIS_BA_DIST_NONE;
    return g;
  }
  okvis_amd::Estimator impl_;
  std::map<uint64_t, okvis::MultiFramePtr> frames_;
  std::map<uint64_t, okvis_amd::MultiFramePtr> pods_;
};

}  // namespace okvis
#endif  // OKVIS_AMD_HAVE_OKVIS
