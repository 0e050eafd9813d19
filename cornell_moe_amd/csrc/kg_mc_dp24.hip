// cornell_moe_amd/csrc/kg_mc_dp24.hip -- instantiations of the KG Monte-Carlo kernel (kg_mc.hpp) for padded dimension 24
// (d = 17 .. 24): the reduced set of kg_mc.hpp launch_dp_wide / launch_block_dp_wide.
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_dp24(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_dp_wide<24>(P, G, xlds, blocks, waves, shm, s);
}

void launch_kg_mc_block_dp24(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  mc::launch_block_dp_wide<24>(P, G, tr, num_lds_tiles, blocks, waves, s);
}

void launch_kg_mc_stream_dp24(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_stream_dp_wide<24>(P, G, blocks, waves, shm, s);
}

}  // namespace moe
