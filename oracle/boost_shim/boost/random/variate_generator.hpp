// Test-infrastructure shim (NOT Boost): engine-reference + distribution functor, as used by gpp_random.hpp:300-302.
#pragma once
#include <type_traits>
namespace boost {
template <class EngineRef, class Distribution>
class variate_generator {
 public:
  using Engine = typename std::remove_reference<EngineRef>::type;
  using result_type = typename Distribution::result_type;
  variate_generator(Engine& e, Distribution d) : eng_(e), dist_(d) {}
  result_type operator()() { return dist_(eng_); }
  Distribution& distribution() { return dist_; }
  const Distribution& distribution() const { return dist_; }
  Engine& engine() { return eng_; }
 private:
  Engine& eng_;
  Distribution dist_;
};
}  // namespace boost
