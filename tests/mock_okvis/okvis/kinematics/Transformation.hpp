// stand-in for okvis_kinematics/include/okvis/kinematics/Transformation.hpp:95,135
#pragma once
#include <Eigen/Core>
namespace okvis {
namespace kinematics {
class Transformation {
 public:
  Transformation() {}
  Transformation(const Eigen::Vector3d& r_AB, const Eigen::Quaterniond& q_AB) {
    for (int i = 0; i < 3; ++i) parameters_[i] = r_AB[i];
    parameters_[3] = q_AB.x(); parameters_[4] = q_AB.y(); parameters_[5] = q_AB.z(); parameters_[6] = q_AB.w();
  }
  const Eigen::Matrix<double, 7, 1>& coeffs() const { return parameters_; }   // [r_AB, q_AB (x, y, z, w)]
 private:
  Eigen::Matrix<double, 7, 1> parameters_;
};
}  // namespace kinematics
}  // namespace okvis
