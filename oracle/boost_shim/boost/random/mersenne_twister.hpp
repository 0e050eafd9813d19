// Test-infrastructure shim (NOT Boost): maps the tiny Boost surface the reference core uses onto <random>.
// std::mt19937 is bit-identical to boost::mt19937 (same MT19937 parameters).
#pragma once
#include <random>
namespace boost { using mt19937 = std::mt19937; }
