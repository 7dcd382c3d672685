// stand-in for okvis_cv/include/okvis/MultiFrame.hpp (:142 geometry, :272 MultiFramePtr) and
// implementation/MultiFrame.hpp:86-105,202-215 (timestamp, id, numFrames, T_SC, getKeypoint, getKeypointSize)
#pragma once
#include <Eigen/Core>
#include <memory>
#include <okvis/Time.hpp>
#include <okvis/cameras/CameraBase.hpp>
#include <okvis/kinematics/Transformation.hpp>
namespace okvis {
class MultiFrame {
 public:
  const okvis::Time& timestamp() const;
  uint64_t id() const;
  size_t numFrames() const;
  std::shared_ptr<const okvis::kinematics::Transformation> T_SC(size_t cameraIdx) const;
  std::shared_ptr<const cameras::CameraBase> geometry(size_t cameraIdx) const;
  bool getKeypoint(size_t cameraIdx, size_t keypointIdx, Eigen::Vector2d& keypoint) const;
  bool getKeypointSize(size_t cameraIdx, size_t keypointIdx, double& keypointSize) const;
};
typedef std::shared_ptr<MultiFrame> MultiFramePtr;
}  // namespace okvis
