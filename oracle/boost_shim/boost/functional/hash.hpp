// Test-infrastructure shim (NOT Boost): hash_combine as used by gpp_random.cpp:64-66 (time-based seeding only).
#pragma once
#include <cstddef>
#include <functional>
namespace boost {
template <class T>
inline void hash_combine(std::size_t& seed, const T& v) {
  seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
}  // namespace boost
