// cornell_moe_amd/csrc/kg_mc_dp16b.hip -- instantiations of the KG Monte-Carlo kernels (kg_mc.hpp) for padded dimension 16: the
// workgroup-per-sample and the streamed-weights kernels (the LDS-table kernels are in kg_mc_dp16.hip).
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_block_dp16(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  mc::launch_block_dp<16>(P, G, tr, num_lds_tiles, blocks, waves, s);
}

void launch_kg_mc_stream_dp16(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_stream_dp<16>(P, G, blocks, waves, shm, s);
}

}  // namespace moe
