// Test-infrastructure shim (NOT Boost).
#pragma once
#include <random>
namespace boost { template <class T = int> using uniform_int = std::uniform_int_distribution<T>; }
