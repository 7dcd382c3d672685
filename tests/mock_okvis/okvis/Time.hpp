// stand-in for okvis_time/include/okvis/Time.hpp:128,193
#pragma once
#include <cstdint>
namespace okvis {
class Time {
 public:
  uint32_t sec, nsec;
  Time() : sec(0), nsec(0) {}
  Time(uint32_t _sec, uint32_t _nsec) : sec(_sec), nsec(_nsec) {}
};
}  // namespace okvis
