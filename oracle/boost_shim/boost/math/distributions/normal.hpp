// Test-infrastructure shim (NOT Boost): standard normal pdf/cdf for the analytic 1-EI path (gpp_math.cpp:2208, 2243-2244).
#pragma once
#include <cmath>
namespace boost { namespace math {
template <class T = double>
struct normal_distribution {
  T mean_, sd_;
  explicit normal_distribution(T mean = 0, T sd = 1) : mean_(mean), sd_(sd) {}
};
using normal = normal_distribution<double>;
template <class T>
inline T pdf(const normal_distribution<T>& n, T x) {
  const T z = (x - n.mean_) / n.sd_;
  return std::exp(-0.5 * z * z) / (n.sd_ * 2.5066282746310002);
}
template <class T>
inline T cdf(const normal_distribution<T>& n, T x) {
  const T z = (x - n.mean_) / n.sd_;
  return 0.5 * std::erfc(-z * 0.70710678118654752440);
}
}}  // namespace boost::math
