// stand-in (declarations only): what okvis/timing/Timer.hpp uses of Boost.Date_Time
#pragma once
#include <chrono>
namespace boost { namespace posix_time {
struct time_duration {
  long long ns = 0;
  long long total_nanoseconds() const { return ns; }
  long long total_microseconds() const { return ns / 1000; }
};
struct ptime { std::chrono::steady_clock::time_point t; };
inline time_duration operator-(const ptime& a, const ptime& b) {
  time_duration d;
  d.ns = std::chrono::duration_cast<std::chrono::nanoseconds>(a.t - b.t).count();
  return d;
}
struct microsec_clock { static ptime local_time() { return ptime{std::chrono::steady_clock::now()}; } };
}}
