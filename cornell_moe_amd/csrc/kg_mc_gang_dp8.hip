// cornell_moe_amd/csrc/kg_mc_gang_dp8.hip -- instantiations of the gang MC kernel (kg_mc_gang.hpp) for padded dimension 8.
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_gang_dp8(const KgMcParams& P, int G, int W, int lds_tiles, int blocks, size_t shm, hipStream_t s) {
  mc::launch_gang_dp<8>(P, G, W, lds_tiles, blocks, shm, s);
}

size_t kg_mc_gang_lds_bytes(int dp, int lds_tiles) {
  return sizeof(double) * ((size_t)mc::gang_fixed_doubles(dp) + (size_t)lds_tiles * (dp + 1) * 64);
}

}  // namespace moe
