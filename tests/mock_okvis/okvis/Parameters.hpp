// stand-in for okvis_common/include/okvis/Parameters.hpp:60-100 (ExtrinsicsEstimationParameters), :107-120 (ImuParameters)
#pragma once
#include <Eigen/Core>
namespace okvis {
struct ExtrinsicsEstimationParameters {
  double sigma_absolute_translation, sigma_absolute_orientation, sigma_c_relative_translation, sigma_c_relative_orientation;
};
struct ImuParameters {
  double a_max, g_max, sigma_g_c, sigma_bg, sigma_a_c, sigma_ba, sigma_gw_c, sigma_aw_c, tau, g;
  Eigen::Vector3d a0;
  int rate;
};
}  // namespace okvis
