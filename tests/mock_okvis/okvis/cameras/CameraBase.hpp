// stand-in for okvis_cv/include/okvis/cameras/CameraBase.hpp:149,337
#pragma once
#include <Eigen/Core>
#include <string>
namespace okvis {
namespace cameras {
class CameraBase {
 public:
  virtual ~CameraBase() {}
  virtual void getIntrinsics(Eigen::VectorXd& intrinsics) const = 0;
  virtual const std::string distortionType() const = 0;
};
}  // namespace cameras
}  // namespace okvis
