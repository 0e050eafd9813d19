// cornell_moe_amd/csrc/kg_mc_gang_dp12.hip -- instantiations of the gang MC kernel (kg_mc_gang.hpp) for padded dimension 12.
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_gang_dp12(const KgMcParams& P, int G, int W, int lds_tiles, int blocks, size_t shm, hipStream_t s) {
  mc::launch_gang_dp<12>(P, G, W, lds_tiles, blocks, shm, s);
}

}  // namespace moe
