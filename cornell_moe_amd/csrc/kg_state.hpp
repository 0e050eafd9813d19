// cornell_moe_amd/csrc/kg_state.hpp -- the m x m algebra of a KnowledgeGradientState on the device (r4; kg_state.hip).
//
// Until round 3 a KG evaluation synchronised after its N-sized state kernels, downloaded the Gram matrices and ran
// PreCompute (gpp_knowledge_gradient_optimization.cpp:292-317: mu, Var + noise, its Cholesky factor, the discretised set's
// mu_n / L^-1 cov_n columns), grad mu and Smith's derivative of the factor with its m triangular solves per coordinate
// (gpp_math.cpp:1389-1474) in scalar host code -- on up to 16 host threads -- before it could launch the MC kernel.  These
// kernels do the same algebra from the device-resident Gram blocks; an evaluation is now one uninterrupted stream of launches
// with a single wait at the end.
#pragma once
#include "gp.hpp"

namespace moe {

struct KgStateParams {
  CovParams cp;
  DerivList derivs;              // the GP's derivative observations (carried by the union points)
  double noise[kMaxDerivs + 1];  // noise variance per observation kind
  double mean;
  double best_so_far;
  int E, u, q, m, g, d, dp, A, ng;  // ng = q (1 + g) d gradient columns (0: value only)
  const double* gkk;    // [E][m x m]        (L^-1 K*)^T (L^-1 K*)                        (slices > 1: [E][slices][...] partial sums over
  const double* gx;     // [E][(ng + A) x m] [dK* | K(X, discretised set)]^T K^-1 K*      K slices, added up in slice order on use: gram_entry)
  int gkk_slices, gx_slices;
  const double* ek;     // [E m | E ng | E A] E^T K^-1 (y - mean), grouped by kind (BatchLayout)
  const double* U;      // [E][u][dp]  union points
  const double* extra;  // [E][A][dp]  discretised set (fidelity coordinates at 1)
  double* blob;         // per-evaluation records (KgRec, kg_mc.hpp): the kernel fills L, mu_disc, C_disc, best_posterior
  int rec_stride, rec_L, rec_mu_disc, rec_C_disc, rec_bp;
  int* flags;    // [E] 0, or failing pivot + 1 of chol(Var + noise)
  int* winner;   // [E] the point of Xu whose posterior mean beats best_so_far by most (-1: none)  (.cpp:146-154)
  double* gmu;   // [E][q][d] grad mu of the function values
  double* dL;    // [E][q d][m (m + 1) / 2] Smith's d chol / d Xq_k,dd, packed: entry (l, j), l >= j, at l (l + 1) / 2 + j
};

// mu, Var + noise, chol, winner, the discretised set's mu_n and c_j = L^-1 cov_n(Xu, x_j), grad mu: one workgroup per evaluation.
void launch_kg_state(const KgStateParams& P, hipStream_t s);
// Smith's derivative of the factor for every (evaluation, point to sample, coordinate): one workgroup each.
void launch_kg_dchol(const KgStateParams& P, hipStream_t s);

struct KgFinishParams {
  int E, q, m, g, d, ng, num_mc, first_sample;
  const double* blob;  // L
  int rec_stride, rec_L;
  const double* out;   // [E][out_stride]: kg_sum | ZC (m x m) | DIR (ng) | GTB (ng)
  int out_stride;
  const int* winner;
  const double* gmu;
  const double* dL;
  double* fin;         // [E][1 + q d + 3]: kg_sum | grad_sum | value passes | gradient passes | singular flag -- everything the host
                       // reads back, in ONE copy
  const unsigned long long* counters;  // [E][2] pass counters of the MC kernel
  const int* flags;                    // [E] kg_state_kernel's singular-matrix flags
  // r5: DIR as kg_dir_kernel leaves it -- [E][ng][dir_slices] partial sums over sample ranges, added up in slice order where they are
  // consumed (a kernel of its own did that before)
  const double* dir_part;
  int dir_slices;
  // r5, m <= 8 only (zc_part != NULL): ZC and kg_sum are formed by the finish kernel from kg_zc_part_kernel's chunk partials and the
  // samples' best values, in kg_zc_sum_kernel's own order (gs lanes per entry striding the chunks, a fixed butterfly; 256-thread block
  // sum) -- that kernel is then not launched.  Otherwise both are read from `out`.
  const double* zc_part;
  int zc_chunks, zc_gs;
  const double* best_value;  // [E][num_local]
  int num_local, rec_bp;
};
// grad KG from the sample sums: < L^-1 dL, ZC > taken as < dL, L^-T tril(ZC) > (one m-column back substitution per evaluation
// instead of q d m forward ones), the DIR - GTB terms and the winner's grad mu.  Y: E m (m + 1) / 2 doubles of workspace.  Two kernels
// (Y, then a wavefront per gradient component); for m <= 8 with zc_part set ONE, which also takes kg_zc_sum_kernel's place (r5).
void launch_kg_finish(const KgFinishParams& P, double* Y, hipStream_t s);

}  // namespace moe
