// cornell_moe_amd/csrc/gemm128.hpp -- FP64 GEMM core on the matrix pipe for the big products of the GP build (the two
// triangular products per level of the inverse factor, the rank-512 update of the factorisation): a 128 x 128 output tile per
// workgroup, four wavefronts in a 2 x 2 arrangement, each owning a 64 x 64 quadrant as 4 x 4 v_mfma_f64_16x16x4_f64 tiles
// (128 accumulator VGPRs).  Per K step of 4 a wavefront reads 8 operand fragments from LDS for 16 MFMAs -- half the LDS and
// a quarter of the L2 traffic per flop of the 64 x 64 kernel (kernels_linalg.hip: mfma_gemm_kernel), which this one replaces
// wherever the output has enough 128-tiles to fill the chip.
//
// LDS (dynamic, 72 KB: two workgroups per CU): two buffers x two operand tiles of 128 x 16 doubles.  An operand whose memory
// runs along the output index (rows of A, columns of B^T) is kept [k][mn] with a leading dimension of 144 doubles, one whose
// memory runs along K is kept [mn][k] with a leading dimension of 18 -- both make the MFMA fragment read (lanes 0-15: sixteen
// consecutive mn at k, lanes 16-31: the same at k + 1, ...; ds_read_b64 is served in two 32-lane groups over 64 banks) and
// the tile stores (ds_write_b64: contiguous 16-lane groups) conflict-free, and neither needs a transpose on the way in.
// One __syncthreads per K tile: tile t + 1 travels global -> registers while tile t is multiplied, and is written to the
// other buffer before the barrier.
#pragma once
#include <hip/hip_runtime.h>

namespace moe {
namespace g128 {

using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int TM = 128, TK = 16, LDM = 144, LDK = 18;
constexpr int kTile = 2304;  // doubles of one operand tile in LDS (16 x 144 = 128 x 18)
constexpr size_t kSmemBytes = sizeof(double) * kTile * 4;
constexpr int kPer = TM * TK / 256;  // elements of each operand tile per thread

// One operand: element (mn, k), mn = output row (A) or column (B).
//   KC (K-contiguous):  p[k + mn * ld]      else (MN-contiguous):  p[mn + k * ld]
// valid for mn < mn_lim and k < k_lim.  Triangular operands (MASK 1: zero where k > mn -- a lower triangle whose rows are mn;
// MASK 2: zero where k < mn * scale -- a transposed lower triangle whose row mn is column mn * scale of the triangle) must HOLD
// their zeros (L^-1 is cleared before it is built and every kernel writes its lower part only): the mask restricts the K range
// tile by tile (gemm128_kernel), the elements are read as they are.
struct Operand {
  const double* p;
  long ld;
  int mn_lim, k_lim, scale;
};

// A thread's source pointers for the tiles of one output tile.  Rows / columns beyond mn_lim are CLAMPED to the last valid one
// (what they produce lands in accumulator rows / columns that are never stored), so the K loop carries no predicate.
template <bool KC>
struct TileSrc {
  const double* p[KC ? kPer : 1];
  long step;  // MN-contiguous: doubles between a thread's consecutive elements (2 k-rows)
};

template <bool KC>
__device__ __forceinline__ void src_init(const Operand& op, int mn0, int k0, TileSrc<KC>& src) {
  const int t = threadIdx.x;
  if (KC) {
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int gm = min(mn0 + (t >> 4) + 16 * it, op.mn_lim - 1);
      src.p[it] = op.p + (long)gm * op.ld + (k0 + (t & 15));
    }
    src.step = 0;
  } else {
    const int gm = min(mn0 + (t & 127), op.mn_lim - 1);
    src.p[0] = op.p + gm + (long)(k0 + (t >> 7)) * op.ld;
    src.step = 2 * op.ld;
  }
}

// the next full tile (all 16 k valid), then advance
template <bool KC>
__device__ __forceinline__ void fetch_full(const Operand& op, TileSrc<KC>& src, double (&r)[kPer]) {
#pragma unroll
  for (int it = 0; it < kPer; ++it) r[it] = KC ? src.p[it][0] : src.p[0][(long)it * src.step];
  if (KC) {
#pragma unroll
    for (int it = 0; it < kPer; ++it) src.p[it] += TK;
  } else {
    src.p[0] += (long)TK * op.ld;
  }
}

// the last, partial tile: k beyond k_lim reads the last valid k and counts as zero
template <bool KC>
__device__ __forceinline__ void fetch_tail(const Operand& op, const TileSrc<KC>& src, int k0, double (&r)[kPer]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < kPer; ++it) {
    const int kk = KC ? (t & 15) : (t >> 7) + 2 * it;
    const int back = max(0, k0 + kk - (op.k_lim - 1));  // steps beyond the last valid k
    const double v = KC ? src.p[it][-back] : src.p[0][(long)it * src.step - (long)back * op.ld];
    r[it] = back > 0 ? 0.0 : v;
  }
}

template <bool KC>
__device__ __forceinline__ void stash_tile(double* __restrict__ S, const double (&r)[kPer]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < kPer; ++it) {
    if (KC)
      S[((t >> 4) + 16 * it) * LDK + (t & 15)] = r[it];
    else
      S[((t >> 7) + 2 * it) * LDM + (t & 127)] = r[it];
  }
}

// acc[a][b] += (A tile)(B tile) for this wavefront's quadrant (rows wi + 16 a + ., columns wj + 16 b + .).  The operands go in
// swapped (MFMA's A <- the B fragment) so that a result register's 16 consecutive lanes hold 16 consecutive ROWS of C:
// lane (lx = lane & 15, lk = lane >> 4) of acc[a][b][r] is C[wi + 16 a + lx][wj + 16 b + lk + 4 r].
template <bool AKC, bool BKC>
__device__ __forceinline__ void mma_tile(const double* __restrict__ As, const double* __restrict__ Bs, f64x4 (&acc)[4][4], int wi,
                                         int wj, int lk, int lx) {
#pragma unroll
  for (int k4 = 0; k4 < TK; k4 += 4) {
    double fa[4], fb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = AKC ? As[(wi + 16 * a + lx) * LDK + k4 + lk] : As[(k4 + lk) * LDM + wi + 16 * a + lx];
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = BKC ? Bs[(wj + 16 * b + lx) * LDK + k4 + lk] : Bs[(k4 + lk) * LDM + wj + 16 * b + lx];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
  }
}

// The product of A's rows [i0, i0 + 128) and B's columns [j0, j0 + 128) over k in [k_lo, k_hi) (k_lo a multiple of 16, k_hi <= both
// operands' k_lim) into acc.  smem: kSmemBytes of dynamic LDS.  Ends with a barrier (the caller may reuse smem).
template <bool AKC, bool BKC>
__device__ __forceinline__ void tile_product(const Operand& A, const Operand& B, int i0, int j0, int k_lo, int k_hi, double* smem,
                                             f64x4 (&acc)[4][4]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
  const int lk = lane >> 4, lx = lane & 15;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  if (k_lo >= k_hi) return;
  const int k_full = k_lo + (k_hi - k_lo) / TK * TK;  // end of the full tiles; [k_full, k_hi) is the partial one, if any
  TileSrc<AKC> sa;
  TileSrc<BKC> sb;
  src_init<AKC>(A, i0, k_lo, sa);
  src_init<BKC>(B, j0, k_lo, sb);
  double ra[kPer], rb[kPer];
  if (k_lo < k_full) {
    fetch_full<AKC>(A, sa, ra);
    fetch_full<BKC>(B, sb, rb);
  } else {
    fetch_tail<AKC>(A, sa, k_lo, ra);
    fetch_tail<BKC>(B, sb, k_lo, rb);
  }
  stash_tile<AKC>(smem, ra);
  stash_tile<BKC>(smem + kTile, rb);
  __syncthreads();
  int cur = 0;
  for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
    const int kn = k0 + TK;
    if (kn < k_full) {
      fetch_full<AKC>(A, sa, ra);
      fetch_full<BKC>(B, sb, rb);
    } else if (kn < k_hi) {
      fetch_tail<AKC>(A, sa, kn, ra);
      fetch_tail<BKC>(B, sb, kn, rb);
    }
    mma_tile<AKC, BKC>(smem + cur * 2 * kTile, smem + cur * 2 * kTile + kTile, acc, wi, wj, lk, lx);
    if (kn < k_hi) {
      stash_tile<AKC>(smem + (1 - cur) * 2 * kTile, ra);
      stash_tile<BKC>(smem + (1 - cur) * 2 * kTile + kTile, rb);
    }
    __syncthreads();
    cur = 1 - cur;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C = [-] A B, batched: problem z (`batch` of them) at element offsets z sA / z sB / z sC, with min(M, m_total - z m_step) rows
// when m_step > 0 (the nodes of one level of the triangular inversion; the last may be cut off by the matrix edge -- for
// AMASK 1 its K shrinks with it).  Triangular operands restrict a tile's K range (AMASK 1 -> k < i0 + 128, AMASK 2 -> k >= i0
// scale, BMASK 2 -> k >= j0 scale), so tiles differ in length by up to the whole K: gridDim.x = batch x row tiles x column tiles,
// numbered so that the longest tiles of ALL problems are dispatched first (the hardware hands a freed slot the next workgroup:
// longest-first list scheduling).  Tried and dropped (r4): persistent workgroups drawing tile numbers from a counter -- the same
// order with its own overhead, 1.29 -> 1.42 ms at the top level of the N = 8000 inverse.
// ---------------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  Operand A, B;
  double* C;
  long ldc;
  int M, N, K;
  long sA, sB, sC;
  int m_total, m_step;
  int batch;
};

template <bool AKC, int AMASK, bool BKC, int BMASK, bool NEG>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int R = (g.M + TM - 1) / TM, Ct = (g.N + TM - 1) / TM;  // (tile counts of the uncut problem)
  const unsigned int id = blockIdx.x;
  int z, it, jt;
  if (BMASK == 2) {  // K shrinks with the column: first columns first
    jt = (int)(id / (unsigned int)(R * g.batch));
    const int rem = (int)(id % (unsigned int)(R * g.batch));
    z = rem / R;
    it = rem % R;
  } else if (AMASK == 1) {  // K grows with the row: last rows first
    it = R - 1 - (int)(id / (unsigned int)(Ct * g.batch));
    const int rem = (int)(id % (unsigned int)(Ct * g.batch));
    z = rem / Ct;
    jt = rem % Ct;
  } else {  // (AMASK 2: K shrinks with the row; none: uniform)
    it = (int)(id / (unsigned int)(Ct * g.batch));
    const int rem = (int)(id % (unsigned int)(Ct * g.batch));
    z = rem / Ct;
    jt = rem % Ct;
  }
  Operand A = g.A, B = g.B;
  A.p += (long)z * g.sA;
  B.p += (long)z * g.sB;
  double* C = g.C + (long)z * g.sC;
  int M = g.M, K = g.K;
  if (g.m_step > 0) {
    const int mz = min(M, g.m_total - z * g.m_step);
    if (AMASK == 1) K = min(K, mz);
    M = mz;
  }
  A.mn_lim = M;
  B.mn_lim = g.N;
  A.k_lim = B.k_lim = K;
  const int i0 = it * TM, j0 = jt * TM;
  if (i0 >= M) return;
  int k_lo = 0, k_hi = K;
  if (AMASK == 1) k_hi = min(K, i0 + TM);
  if (AMASK == 2) k_lo = max(k_lo, (int)(((long)i0 * A.scale) / TK) * TK);
  if (BMASK == 2) k_lo = max(k_lo, (int)(((long)j0 * B.scale) / TK) * TK);
  f64x4 acc[4][4];
  tile_product<AKC, BKC>(A, B, i0, j0, k_lo, k_hi, smem, acc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
  const int lk = lane >> 4, lx = lane & 15;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
        if (gi < M && gj < g.N) C[(long)gi + (long)gj * g.ldc] = NEG ? -acc[a][b][r] : acc[a][b][r];
      }
}

// ---------------------------------------------------------------------------------------------------------------------
// S[i][j] -= sum_{k < kw} P[i][k] P[j][k] on the lower triangle of the trailing block S = A[base.., base..] with the row panel
// P = A[base.., kp0 .. kp0 + kw): the rank-kw update of a right-looking Cholesky factorisation.  gridDim.x = T (T + 1) / 2
// lower-triangle tiles of 128 (T = ceil((N - base) / 128)), walked column by column.
// ---------------------------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256, 2) void syrk128_kernel(double* __restrict__ Amat, long lda, int N, int base, int kp0, int kw,
                                                         const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (*info != 0) return;
  const int n_tr = N - base;
  const int T = (n_tr + TM - 1) / TM;
  // tile id -> (column jt, row it >= jt), columns first: id = jt T - jt (jt - 1) / 2 + (it - jt)
  int id = blockIdx.x, jt = 0;
  {
    // (closed form with a correction step: the square root is only a first guess)
    const double tt = 2.0 * T + 1.0;
    jt = (int)((tt - sqrt(tt * tt - 8.0 * (double)id)) * 0.5);
    jt = max(0, min(T - 1, jt));
    while (jt > 0 && (long)jt * T - (long)jt * (jt - 1) / 2 > id) --jt;
    while ((long)(jt + 1) * T - (long)(jt + 1) * jt / 2 <= id) ++jt;
  }
  const int it = jt + (id - (int)((long)jt * T - (long)jt * (jt - 1) / 2));
  const int i0 = it * TM, j0 = jt * TM;
  Operand P;
  P.p = Amat + (long)base + (long)kp0 * lda;
  P.ld = lda;
  P.mn_lim = n_tr;
  P.k_lim = kw;
  P.scale = 1;
  f64x4 acc[4][4];
  tile_product<false, false>(P, P, i0, j0, 0, kw, smem, acc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
  const int lk = lane >> 4, lx = lane & 15;
  double* S = Amat + (long)base + (long)base * lda;
  // (loads first, then the stores: 64 read-modify-writes in sequence would each pay their own round trip)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double cv[16];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
        cv[b * 4 + r] = S[(long)min(gi, n_tr - 1) + (long)min(gj, n_tr - 1) * lda];
      }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
        if (gi < n_tr && gj < n_tr && gj <= gi) S[(long)gi + (long)gj * lda] = cv[b * 4 + r] - acc[a][b][r];
      }
  }
}

}  // namespace g128
}  // namespace moe
