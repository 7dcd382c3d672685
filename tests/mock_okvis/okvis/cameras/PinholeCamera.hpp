// stand-in for okvis_cv/include/okvis/cameras/PinholeCamera.hpp: the GEOMETRY_TYPE of Estimator::addObservation<>
#pragma once
#include <okvis/cameras/CameraBase.hpp>
namespace okvis {
namespace cameras {
template <class DISTORTION_T>
class PinholeCamera : public CameraBase {};
}  // namespace cameras
}  // namespace okvis
