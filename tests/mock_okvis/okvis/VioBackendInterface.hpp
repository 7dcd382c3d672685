// stand-in for okvis_common/include/okvis/VioBackendInterface.hpp:67-336: every virtual of the interface with the
// reference's signature (pure where the reference's is pure), nothing else.
#pragma once
#include <Eigen/Core>
#include <cstdint>
#include <memory>
#include <okvis/FrameTypedefs.hpp>
#include <okvis/Measurements.hpp>
#include <okvis/MultiFrame.hpp>
#include <okvis/Parameters.hpp>
#include <okvis/assert_macros.hpp>
#include <okvis/kinematics/Transformation.hpp>
namespace ceres {
namespace internal {
class ResidualBlock;
}
typedef internal::ResidualBlock* ResidualBlockId;   // ceres/types.h
}  // namespace ceres
namespace okvis {
namespace ceres {
class Map;
}
class VioBackendInterface {
 public:
  enum class InitializationStatus { NotStarted = 0, Ongoing = 1, Complete = 2 };
  VioBackendInterface() {}
  virtual ~VioBackendInterface() {}
  virtual int addCamera(const ExtrinsicsEstimationParameters& extrinsicsEstimationParameters) = 0;              // :90
  virtual int addImu(const ImuParameters& imuParameters) = 0;                                                    // :99
  virtual void clearCameras() = 0;                                                                               // :104
  virtual void clearImus() = 0;                                                                                  // :109
  virtual bool addStates(okvis::MultiFramePtr multiFrame, const okvis::ImuMeasurementDeque& imuMeasurements,
                         bool asKeyframe) = 0;                                                                   // :120
  virtual bool addLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) = 0;                            // :130
  virtual bool removeObservation(uint64_t landmarkId, uint64_t poseId, size_t camIdx, size_t keypointIdx) = 0;   // :141
  virtual void optimize(size_t numIter, size_t numThreads = 1, bool verbose = false) = 0;                        // :149
  virtual bool setOptimizationTimeLimit(double timeLimit, int minIterations) = 0;                                // :158
  virtual bool isLandmarkAdded(uint64_t landmarkId) const = 0;                                                   // :165
  virtual bool isLandmarkInitialized(uint64_t landmarkId) const = 0;                                             // :172
  virtual bool getLandmark(uint64_t landmarkId, MapPoint& mapPoint) const = 0;                                   // :182
  virtual size_t getLandmarks(PointMap& landmarks) const = 0;                                                    // :189
  virtual okvis::MultiFramePtr multiFrame(uint64_t frameId) const = 0;                                           // :196
  virtual bool get_T_WS(uint64_t poseId, okvis::kinematics::Transformation& T_WS) const = 0;                     // :204
  virtual bool getSpeedAndBias(uint64_t poseId, uint64_t imuIdx, okvis::SpeedAndBias& speedAndBias) const = 0;   // :213
  virtual bool getCameraSensorStates(uint64_t poseId, size_t cameraIdx,
                                     okvis::kinematics::Transformation& T_SCi) const = 0;                       // :222
  virtual size_t numFrames() const = 0;                                                                          // :227
  virtual size_t numLandmarks() const = 0;                                                                       // :231
  virtual uint64_t currentFrameId() const = 0;                                                                   // :235
  virtual InitializationStatus initializationStatus() const { return InitializationStatus::NotStarted; }        // :238
  virtual bool isKeyframe(uint64_t frameId) const = 0;                                                           // :249
  virtual okvis::Time timestamp(uint64_t frameId) const = 0;                                                     // :258
  virtual bool set_T_WS(uint64_t poseId, const okvis::kinematics::Transformation& T_WS) = 0;                     // :289
  virtual bool setSpeedAndBias(uint64_t poseId, size_t imuIdx, const okvis::SpeedAndBias& speedAndBias) = 0;     // :298
  virtual bool setCameraSensorStates(uint64_t poseId, size_t cameraIdx,
                                     const okvis::kinematics::Transformation& T_SCi) = 0;                       // :307
  virtual bool setLandmark(uint64_t landmarkId, const Eigen::Vector4d& landmark) = 0;                            // :314
  virtual void setLandmarkInitialized(uint64_t landmarkId, bool initialized) = 0;                                // :320
  virtual void setKeyframe(uint64_t frameId, bool isKeyframe) = 0;                                               // :325
  virtual void setMap(std::shared_ptr<okvis::ceres::Map> mapPtr) = 0;                                            // :329
  virtual void setComputeUncertainty(bool /*computeUncertainty*/) {}                                             // :333
};
}  // namespace okvis
