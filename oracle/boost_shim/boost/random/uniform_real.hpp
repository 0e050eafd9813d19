// Test-infrastructure shim (NOT Boost).
#pragma once
#include <random>
namespace boost { template <class T = double> using uniform_real = std::uniform_real_distribution<T>; }
