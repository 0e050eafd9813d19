// cornell_moe_amd/csrc/kg_mc_dp12.hip -- instantiations of the KG Monte-Carlo kernels (kg_mc.hpp) for padded dimension 12: the LDS-table wave-per-sample kernels
// (frame and lane-parked line search); the workgroup-per-sample and streamed-weights kernels are in kg_mc_dp12b.hip (r6: two
// translation units per dimension -- with the ensemble twins one unit took six minutes to compile).
#include "kg_mc.hpp"

namespace moe {

void launch_kg_mc_dp12(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_dp<12>(P, G, xlds, blocks, waves, shm, s);
}

void launch_kg_mc_lane_dp12(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  mc::launch_lane_dp<12>(P, G, rec_head, blocks, waves, shm, s);
}

}  // namespace moe
