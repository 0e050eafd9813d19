// cornell_moe_amd/csrc/kg_mc_dp4.hip -- instantiations of the KG Monte-Carlo kernels (kg_mc.hpp) for padded dimension 4: the LDS-table wave-per-sample kernels
// (frame and lane-parked line search); the workgroup-per-sample and streamed-weights kernels are in kg_mc_dp4b.hip (r6: two
// translation units per dimension -- with the ensemble twins one unit took six minutes to compile).
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_dp4(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_dp<4>(P, G, xlds, blocks, waves, shm, s);
}

size_t kg_mc_block_lds_bytes(int dp, int G, int num_lds_tiles) {
  return sizeof(double) * (mc::kBlockFixed + (size_t)num_lds_tiles * (dp + 1 + G) * 64);
}

size_t kg_mc_lane_fixed_bytes(int dp, int rec_head) {
  return sizeof(double) * ((size_t)kExpTabLen + (size_t)mc::kLaneCstRows * dp + (size_t)((rec_head + 1) & ~1));
}


void launch_kg_mc_lane_dp4(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_lane_dp<4>(P, G, rec_head, blocks, waves, shm, s);
}

}  // namespace moe
