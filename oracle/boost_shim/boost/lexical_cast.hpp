// Test-infrastructure shim (NOT Boost): number -> string for exception messages (gpp_exception.cpp:56-89).
#pragma once
#include <sstream>
#include <string>
namespace boost {
template <class To, class From>
inline To lexical_cast(const From& f) {
  std::ostringstream o;
  o.precision(17);
  o << f;
  return o.str();
}
}  // namespace boost
