// stand-in for okvis_common/include/okvis/Measurements.hpp:60-75 (Measurement<T>), :100-112 (ImuSensorReadings), :163
#pragma once
#include <Eigen/Core>
#include <deque>
#include <okvis/Time.hpp>
namespace okvis {
template <class MEASUREMENT_T>
struct Measurement {
  okvis::Time timeStamp;
  MEASUREMENT_T measurement;
};
struct ImuSensorReadings {
  Eigen::Vector3d gyroscopes, accelerometers;
};
typedef Measurement<ImuSensorReadings> ImuMeasurement;
typedef std::deque<ImuMeasurement, Eigen::aligned_allocator<ImuMeasurement> > ImuMeasurementDeque;
}  // namespace okvis
