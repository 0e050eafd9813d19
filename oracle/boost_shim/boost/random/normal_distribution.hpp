// Test-infrastructure shim (NOT Boost). NOTE: std::normal_distribution != boost::normal_distribution draw-for-draw;
// parity runs never rely on it -- normals are injected via NormalRNGSimulator (gpp_random.hpp:314-340).
#pragma once
#include <random>
namespace boost { template <class T = double> using normal_distribution = std::normal_distribution<T>; }
