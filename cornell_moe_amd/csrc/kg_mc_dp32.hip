// cornell_moe_amd/csrc/kg_mc_dp32.hip -- instantiations of the KG Monte-Carlo kernel (kg_mc.hpp) for padded dimension 32
// (d = 25 .. 32): the reduced set of kg_mc.hpp launch_dp_wide / launch_block_dp_wide.
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_dp32(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_dp_wide<32>(P, G, xlds, blocks, waves, shm, s);
}

void launch_kg_mc_block_dp32(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  mc::launch_block_dp_wide<32>(P, G, tr, num_lds_tiles, blocks, waves, s);
}

void launch_kg_mc_stream_dp32(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_stream_dp_wide<32>(P, G, blocks, waves, shm, s);
}

}  // namespace moe
