// stand-in (declarations only): what okvis/timing/Timer.hpp uses of Boost.Accumulators
#pragma once
#include <cstddef>
namespace boost { namespace accumulators {
namespace tag {
struct lazy_variance {}; struct sum {}; struct min {}; struct max {}; struct rolling_mean {}; struct mean {};
struct window_size_arg { std::size_t n; };
struct window_size_keyword { window_size_arg operator=(std::size_t n) const { return window_size_arg{n}; } };
struct rolling_window { static constexpr window_size_keyword window_size{}; };
}
template <class... F> struct features {};
template <class T, class F> struct accumulator_set {
  accumulator_set() {}
  explicit accumulator_set(tag::window_size_arg) {}
  void operator()(const T&) {}
};
template <class S> double sum(const S&);
template <class S> double min(const S&);
template <class S> double max(const S&);
template <class S> double mean(const S&);
template <class S> double rolling_mean(const S&);
template <class S> double lazy_variance(const S&);
template <class S> double variance(const S&);
template <class S> std::size_t count(const S&);
namespace extract { using boost::accumulators::sum; using boost::accumulators::min; using boost::accumulators::max;
  using boost::accumulators::mean; using boost::accumulators::rolling_mean; using boost::accumulators::lazy_variance; using boost::accumulators::count; }
}}
