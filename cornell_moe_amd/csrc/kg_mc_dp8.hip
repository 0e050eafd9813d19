// cornell_moe_amd/csrc/kg_mc_dp8.hip -- instantiations of the KG Monte-Carlo kernels (kg_mc.hpp) for padded dimension 8: the LDS-table wave-per-sample kernels
// (frame and lane-parked line search); the workgroup-per-sample and streamed-weights kernels are in kg_mc_dp8b.hip (r6: two
// translation units per dimension -- with the ensemble twins one unit took six minutes to compile).
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_dp8(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_dp<8>(P, G, xlds, blocks, waves, shm, s);
}

void launch_kg_mc_lane_dp8(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_lane_dp<8>(P, G, rec_head, blocks, waves, shm, s);
}

}  // namespace moe
