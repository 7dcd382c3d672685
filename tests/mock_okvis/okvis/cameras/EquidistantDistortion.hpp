// stand-in: the adapter only names the distortion classes as template arguments of PinholeCamera
#pragma once
namespace okvis {
namespace cameras {
class EquidistantDistortion {};
}  // namespace cameras
}  // namespace okvis
