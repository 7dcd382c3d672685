// Reduced-camera solve with the speed/bias blocks eliminated ALONG THE IMU CHAIN first (round 6).
//
// The reduced system of a window (after the landmark Schur complement; Estimator.cpp:854 SPARSE_SCHUR, in-tree analogue
// MarginalizationError.cpp:617-689) is not a generic dense matrix: a speed/bias block couples only to the speed/bias blocks of
// the neighbouring frames and to poses (ImuError.cpp:608-682: J0 .. J3 of one term touch pose_k, sb_k, pose_k+1, sb_k+1), the
// pose part is dense.  The blocked LDL^T of ba_ldl16.hpp treats the D x D system as dense 16 x 16 blocks: ten dependent
// diagonal-block steps at configs[1] (D = 150), 28.7 of the solve kernel's 44.8 us (profiles/r05_notes.md).  Here:
//
//   1. the speed/bias blocks A_k (9 x 9) are eliminated block by block from BOTH ends of the chain towards the middle (two
//      independent sweeps, depth ceil(K/2) + 1 instead of K): one wave per sweep carries the dependent chain
//        [A_k  M; M^T  A_next]  ->  L_k D_k L_k^T,  R_C = L_k^-1 M,  A_next -= R_C^T D_k^-1 R_C
//      as ONE Gauss elimination on 18 rows held column per lane (pivot column broadcast by DPP row_newbcast, as ba_ldl16.hpp;
//      the update of the next diagonal block falls out of the same instruction stream), and publishes L_k^-1, D_k^-1 and
//      D_k^-1 R_C; one more wave per sweep owns the pose couplings and the right-hand side, a column per lane for the whole
//      sweep:  y = L_k^-1 N_k[:, c],  N_next[:, c] -= (D_k^-1 R_C)^T y.  No workgroup barrier inside the sweeps;
//   2. the pose system takes  S_pp -= sum_k Y_k^T D_k^-1 Y_k  (and its right-hand side) in one pass of all waves, skipping the
//      pose columns a block never reached;
//   3. the Dp x Dp pose system (60 x 60 at configs[1]: four 16-blocks instead of ten) is solved by ldl16_solve unchanged;
//   4. the speed/bias part of the step follows by back-substitution from the middle outwards.
//
// The oracle keeps the landmark-only ordering of Ceres (dense Cholesky of the whole reduced system): the two orderings solve
// the same symmetric positive definite system, parity is on the step and the cost (tests/test_gpu_chain_solve.py).
#pragma once
#include "ba_ldl16.hpp"

namespace ba {

constexpr int CH_MAX_KS = MAX_D_LDS / 9;   // speed/bias blocks of a window solved in LDS
constexpr int CH_NCHUNK = 3;               // pose couplings + right-hand side: columns in chunks of 64 (Dp + 1 <= 192)

// LDS layout of the assembled system in chain mode: the pose system in the blocked layout of ba_ldl16.hpp at offset 0 (its
// right-hand side as column Dp), behind its work area the chain's own arrays.  at(i, j) takes REDUCED coordinates (pose-type
// blocks first, then the speed/bias blocks in chain order; D = the right-hand side), either order, like L16::at.
// Every 9 x 9 block is stored as nine rows of TEN doubles (CB = 90): a row — a column of a symmetric block — is what one lane of
// the sweeps loads or stores as a whole, five aligned 16-byte accesses.
struct LChain {
  static constexpr int CB = 90;
  static constexpr int CH_NCHUNK_ = 3;
  L16 P;
  int Dp, D, Ks, NP;              // pose rows | reduced dimension | speed/bias blocks | pitch of a row of the pose couplings (columns 0 .. Dp-1, column Dp = rhs)
  int oA, oC, oN, oP, oT, oG, oDi, oX, oZ, total;
  __host__ __device__ static LChain make(int D, int Dp) {
    LChain L;
    L.Dp = Dp;
    L.D = D;
    L.Ks = (D - Dp) / 9;
    L.P = L16{ldl16_nb(Dp), 0, Dp};
    L.NP = (Dp + 2) & ~1;
    // What only the sweeps and the pose update need — diagonal blocks, couplings, (D^-1 R_C)^T, the right sweep's share — lies
    // INSIDE the pose system's area, behind its assembled blocks: ldl16_solve's work area, which is scratch until that solver
    // starts (by then these arrays are dead).  What the back-substitution needs lies behind the area.
    const int nC = L.Ks > 1 ? L.Ks - 1 : 1;
    L.oA = L16::blocks(L.P.nb) * 256;                  // diagonal blocks [Ks][CB]: assembled as the lower triangle (i * 10 + j, i >= j), mirrored
                                                       // by chain_solve; the sweeps keep them full (column c at c * 10)
    L.oC = L.oA + L.Ks * CB;                           // couplings [Ks - 1][2][CB]: C_b = H[sb_b+1][sb_b] row-major (assembled), then its transpose
    L.oT = L.oC + nC * 2 * CB;                         // (D_k^-1 R_C)^T [Ks][CB]: column c' of the next block at c' * 10 + m
    L.oX = L.oT + L.Ks * CB;                           // the right sweep's share of the middle block: A [CB], N [9][NP]
    const int dead_end = L.oX + CB + 9 * L.NP, area = ldl16_area_doubles(Dp);
    L.oN = ((dead_end > area ? dead_end : area) + 1) & ~1;   // pose couplings + rhs [Ks][9][NP]; after the sweeps: Y_k = L_k^-1 N_k
    L.oP = L.oN + L.Ks * 9 * L.NP;                     // L_k^-1 by columns [Ks][CB]: entry (i, j), i > j, at j * 10 + i (the rest of a row: dead)
    L.oG = L.oP + L.Ks * CB;                           // L_k^-T D_k^-1 R_C [Ks][CB] (back-substitution), row i at i * 10
    L.oDi = L.oG + L.Ks * CB;                          // d^-1/2  [Ks][10]  (the rows of Y_k are stored scaled by it: Y~ = D^-1/2 Y)
    L.oZ = L.oDi + L.Ks * 10;                          // twelve words that are always zero: entries that do not exist in the chain structure,
    L.total = L.oZ + 12;                               // and what idle lanes load
    return L;
  }
  // the pose system's update runs on the waves between the column waves and the last two, CH_MAXT output tiles each (chain_solve)
  __host__ __device__ static bool tiles_fit(int Dp) { return (Dp + 1 + 63) / 64 <= CH_NCHUNK_; }   // column waves: chunks of 64 columns
  __host__ __device__ __forceinline__ int at(int i, int j) const {
    if (i < j) {
      const int t = i;
      i = j;
      j = t;
    }
    if (i < Dp) return P.at(i, j);
    if (i >= D) {   // the right-hand side
      if (j < Dp) return P.at(Dp, j);
      return oN + (j - Dp) * NP + Dp;
    }
    const int qi = i - Dp, a = qi / 9, ri = qi - 9 * a;
    if (j < Dp) return oN + qi * NP + j;
    const int qj = j - Dp, b = qj / 9, rj = qj - 9 * b;
    if (a == b) return oA + a * CB + ri * 10 + rj;
    if (a == b + 1) return oC + b * 2 * CB + ri * 10 + rj;
    return oZ;
  }
};

// One pivot of the Gauss elimination of a 9 x 9 block held column per lane (lanes j < 9 of a row of 16 lanes; the lanes
// j >= 9 of the row carry further columns, to which the same row operations apply), NR rows per lane: rows 0 .. 8 the block's,
// rows 9 .. NR - 1 whatever rides along (the next block's rows: their Schur complement falls out).  As ldl16_pivot (ba_ldl16.hpp):
// lanes j < K end up with the columns of -L^-1 D below the diagonal, lane K keeps 1 / d_K in `mine`.  No test of the pivot here:
// the caller looks at the reciprocals afterwards (a pivot that is not positive leaves garbage that is never used).
template <int K, int NR>
__device__ __forceinline__ double chain_pivot(double (&r)[NR], double d, double& mine, int j) {
  const double rd = rcp_nr(d);
  const bool me = (j == K);
  double v = -r[K] * rd;
  v = me ? 0.0 : v;
  mine = me ? rd : mine;
  double dn = 1.0;
  if constexpr (K + 1 < NR) {
    fmac_bcast<K>(r[K + 1], r[K + 1], v);
    if constexpr (K < 8) dn = bcast_nop<K + 1>(r[K + 1]);
  }
#pragma unroll
  for (int i = K + 2; i < NR; ++i) fmac_bcast<K>(r[i], r[i], v);
  return dn;
}
template <int NR>
__device__ __forceinline__ void chain_eliminate(double (&r)[NR], double& mine, int j) {
  double d = bcast_nop<0>(r[0]);
  d = chain_pivot<0, NR>(r, d, mine, j);
  d = chain_pivot<1, NR>(r, d, mine, j);
  d = chain_pivot<2, NR>(r, d, mine, j);
  d = chain_pivot<3, NR>(r, d, mine, j);
  d = chain_pivot<4, NR>(r, d, mine, j);
  d = chain_pivot<5, NR>(r, d, mine, j);
  d = chain_pivot<6, NR>(r, d, mine, j);
  d = chain_pivot<7, NR>(r, d, mine, j);
  d = chain_pivot<8, NR>(r, d, mine, j);
  (void)d;
}


// y_i += sum_{jj < i} L^-1[i][jj] v_jj with the packed inverse spread over the lanes of a row of 16 (entry e = i (i - 1) / 2 + jj in
// lane e & 15, register e >> 4), taken through DPP row_newbcast: 36 instructions, no load
// (column by column — jj outer, i inner — so that consecutive instructions add to different registers: nine independent chains
//  instead of one dependent chain per row, whose every link waited for the one before)
template <int I, int JJ>
struct ChApplyP {
  static __device__ __forceinline__ void run(double (&y)[9], const double (&pr)[3], const double (&v)[9]) {
    constexpr int E = I * (I - 1) / 2 + JJ;
    fmac_bcast<E & 15>(y[I], pr[E >> 4], v[JJ]);
    if constexpr (I < 8) ChApplyP<I + 1, JJ>::run(y, pr, v);
    else if constexpr (JJ < 7) ChApplyP<JJ + 2, JJ + 1>::run(y, pr, v);
  }
};
// w_i += sum_m T[i][m] y_m with the 81 entries spread the same way (entry e = 9 i + m; the caller hands in -y)
template <int I, int M>
struct ChApplyT {
  static __device__ __forceinline__ void run(double (&w)[9], const double (&tr)[6], const double (&y)[9]) {
    constexpr int E = 9 * I + M;
    fmac_bcast<E & 15>(w[I], tr[E >> 4], y[M]);
    if constexpr (I < 8) ChApplyT<I + 1, M>::run(w, tr, y);       // (m outer, i inner: nine independent chains)
    else if constexpr (M < 8) ChApplyT<0, M + 1>::run(w, tr, y);
  }
};

typedef double ch_v2 __attribute__((ext_vector_type(2)));
// ten doubles (a row of a block, 16-byte aligned) <-> registers
__device__ __forceinline__ void ch_load10(const double* p, double (&r)[10]) {
  const ch_v2* q = reinterpret_cast<const ch_v2*>(p);
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const ch_v2 a = q[u];
    r[2 * u] = a.x;
    r[2 * u + 1] = a.y;
  }
}
__device__ __forceinline__ void ch_store10(double* p, const double (&r)[10]) {
  ch_v2* q = reinterpret_cast<ch_v2*>(p);
#pragma unroll
  for (int u = 0; u < 5; ++u) q[u] = ch_v2{r[2 * u], r[2 * u + 1]};
}

// ldl_wait_ge with a bound: a hand-over that never comes (a defect, not a state of the data) must not hang the device — the wait
// gives up after ~0.2 s, reports through *s_fail (the step then counts as a failed factorisation) and the kernel runs to its end.
__device__ __forceinline__ void ch_wait_ge(const int* flag, int need, int* s_fail) {
  int polls = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++polls > (1 << 20)) {
      __hip_atomic_store(s_fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// All NW waves must call.  S: the system assembled in the layout LY (damped, right-hand sides in place).  x_out (LDS, >= D
// doubles, outside the matrix area) receives the solution in reduced coordinates.  *s_fail (zeroed by the caller in front of a
// barrier) is set when a pivot is not positive.  Ends with a barrier.  comp_mask: ldl16_solve's, for the POSE system's blocks.
template <int NW>
__device__ __forceinline__ void chain_solve(double* S, const LChain& LY, int tid, double* x_out, int* s_fail, long long* stamps,
                                            unsigned comp_mask) {
  static_assert(NW == 16, "wave roles below assume 16 waves");
  constexpr int CB = LChain::CB;
  const int Ks = LY.Ks, Dp = LY.Dp, NP = LY.NP;
  const int nL = Ks / 2, nR = (Ks - 1) / 2, mid = nL;   // left sweep: blocks 0 .. nL-1, right sweep: Ks-1 .. Ks-nR, then the middle one
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ncol = Dp + 1;                               // pose columns + the right-hand side
  const int nchunk = (ncol + 63) / 64;                   // column waves per sweep
  __shared__ int cf_pub[2];                  // [side] steps whose L^-1, 1/d, (D^-1 R_C)^T are published
  __shared__ int cf_cdone[2];                // [side] 1: the sweep's chain wave is through (right: its share of the middle diagonal block sits in oX)
  __shared__ int cf_ndone[2][CH_NCHUNK];     // [side][chunk] 1: the sweep's column wave is through (right: its share of the middle block's columns sits in oX)
  __shared__ int cf_mid;                     // 1: L^-1, 1/d of the middle block published
  __shared__ int cf_lo[CH_MAX_KS + 1], cf_hi[CH_MAX_KS + 1];   // pose columns [lo, hi) in which Y_k is not exactly zero
  __shared__ int cf_y[CH_MAX_KS + 1];        // [k] column waves that have stored their part of Y_k
  __shared__ int cf_lo0[CH_MAX_KS + 1], cf_hi0[CH_MAX_KS + 1];   // the same ranges as assembled (before the fill)
  if (tid < 2) {
    cf_pub[tid] = 0;
    cf_cdone[tid] = 0;
  }
  if (tid < 2 * CH_NCHUNK) (&cf_ndone[0][0])[tid] = 0;
  if (tid == 0) cf_mid = 0;
  if (tid <= CH_MAX_KS) cf_y[tid] = 0;
  // the assembly wrote the lower triangles of the diagonal blocks and the couplings C_b: the sweeps read whole columns — the
  // upper triangles and the transposes of the couplings, one entry per work-item
  for (int e = tid; e < Ks * 36 + (Ks - 1) * 81; e += NW * 64) {
    if (e < Ks * 36) {
      const int a = e / 36, t = e - 36 * a;
      int ri = (int)((sqrtf(8.0f * (float)t + 1.0f) + 1.0f) * 0.5f);   // t = ri (ri - 1) / 2 + rj, ri > rj
      while (ri * (ri - 1) / 2 > t) --ri;
      while ((ri + 1) * ri / 2 <= t) ++ri;
      const int rj = t - ri * (ri - 1) / 2;
      S[LY.oA + a * CB + rj * 10 + ri] = S[LY.oA + a * CB + ri * 10 + rj];
    } else {
      const int q = e - Ks * 36, b = q / 81, t = q - 81 * b, ri = t / 9, rj = t - 9 * ri;
      S[LY.oC + b * 2 * CB + CB + rj * 10 + ri] = S[LY.oC + b * 2 * CB + ri * 10 + rj];
    }
  }
  // Which pose columns a block's row of couplings reaches is a matter of STRUCTURE (the poses its IMU terms and the prior touch),
  // known before the sweeps: cf_lo0 / cf_hi0 = the columns in which the assembled N_k is not exactly zero (one wave per block,
  // a ballot over its columns); a sweep hands a block's columns on to the next one, so block k of the left sweep reaches the
  // union over the blocks 0 .. k, of the right sweep over k .. Ks-1, the middle block everything (ch_ranges below).
  for (int k = wave; k < Ks; k += NW) {
    int lo = INT_MAX, hi = 0;
    for (int c0 = 0; c0 < Dp; c0 += 64) {
      const int c = c0 + lane, cl = c < Dp ? c : 0;
      bool nz = false;
#pragma unroll
      for (int m = 0; m < 9; ++m) nz = nz || S[LY.oN + (k * 9 + m) * NP + cl] != 0.0;
      const unsigned long long mk = __ballot(nz && c < Dp);
      if (mk != 0) {
        lo = min(lo, c0 + (int)__builtin_ctzll(mk));
        hi = max(hi, c0 + 64 - (int)__builtin_clzll(mk));
      }
    }
    if (lane == 0) cf_lo0[k] = lo, cf_hi0[k] = hi;
  }
  if (stamps && tid == 0) stamps[16] = clock64();
  __syncthreads();
  // lane k of the calling wave: the range of block k after the fill (a prefix minimum / maximum from either end over the lanes)
  auto ch_ranges = [&](int& flo, int& fhi) {
    const int l = lane < Ks ? lane : 0;
    int lo = lane < Ks ? cf_lo0[l] : INT_MAX, hi = lane < Ks ? cf_hi0[l] : 0;
    int plo = lo, phi = hi, slo = lo, shi = hi;   // prefix (blocks <= lane) and suffix (blocks >= lane) extrema
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int a = __shfl_up(plo, d), b = __shfl_up(phi, d), c_ = __shfl_down(slo, d), e_ = __shfl_down(shi, d);
      if (lane >= d) plo = min(plo, a), phi = max(phi, b);
      if (lane + d < 64) slo = min(slo, c_), shi = max(shi, e_);
    }
    flo = lane < mid ? plo : (lane > mid ? slo : min(plo, slo));
    fhi = lane < mid ? phi : (lane > mid ? shi : max(phi, shi));
  };

  if (wave < 2) {
    // ============================================================================ the chain wave of a sweep
    const int side = wave;
    const int nst = side == 0 ? nL : nR;
    const int j = lane & 15, row = lane >> 4;
    const int pc = (j >= 9 && row == 0) ? j - 9 : ((j >= 9 && j < 11 && row == 1) ? j - 2 : -1);   // column of M this lane carries
    const bool is_a = j < 9 && row < 2, is_p = pc >= 0;
    const int jc = is_a ? j : (is_p ? pc : 0);
    bool bad = false;
    long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (diagnostics: start of every step of the left chain wave, stored behind the sweep)
    __builtin_amdgcn_s_setprio(3);
    for (int t = 0; t < nst; ++t) {
      if (stamps && side == 0 && t < 7) ts_[t] = clock64();
      const int k = side == 0 ? t : Ks - 1 - t, n = side == 0 ? k + 1 : k - 1;
      const bool share = (side == 1 && n == mid);   // the right sweep's last step: its share of the middle block, not the block
      // M[m][c'] (row m of block k, column c' of block n) = C_k^T (left sweep) | C_k-1 (right sweep).  A lane of a column of M
      // reads that column; a lane of column j of A reads row j of M (the column of the symmetric 18 x 18 block [A M; M^T A_n]) —
      // each a row of one of the two copies of the coupling
      const int cb = LY.oC + (side == 0 ? k : k - 1) * 2 * CB;
      const int m_col = cb + (side == 0 ? 0 : CB), m_row = cb + (side == 0 ? CB : 0);
      const int b1 = is_a ? LY.oA + k * CB + jc * 10 : (is_p ? m_col + jc * 10 : LY.oZ);
      const int b2 = is_a ? m_row + jc * 10 : ((is_p && !share) ? LY.oA + n * CB + jc * 10 : LY.oZ);
      double r[18];
      {
        double lo[10], hi[10];
        ch_load10(S + b1, lo);
        ch_load10(S + b2, hi);
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = lo[i], r[9 + i] = hi[i];
      }
      double mine = 1.0;
      chain_eliminate<18>(r, mine, j);
      // ---- publish: L^-1 by columns (from the lanes of A's columns), 1 / d, (D^-1 R_C)^T, the next diagonal block
      double tt[10];
      tt[0] = 0.0; fmac_bcast_nop<0>(tt[0], mine, r[0]);
      tt[1] = 0.0; fmac_bcast<1>(tt[1], mine, r[1]);
      tt[2] = 0.0; fmac_bcast<2>(tt[2], mine, r[2]);
      tt[3] = 0.0; fmac_bcast<3>(tt[3], mine, r[3]);
      tt[4] = 0.0; fmac_bcast<4>(tt[4], mine, r[4]);
      tt[5] = 0.0; fmac_bcast<5>(tt[5], mine, r[5]);
      tt[6] = 0.0; fmac_bcast<6>(tt[6], mine, r[6]);
      tt[7] = 0.0; fmac_bcast<7>(tt[7], mine, r[7]);
      tt[8] = 0.0; fmac_bcast<8>(tt[8], mine, r[8]);
      tt[9] = 0.0;
      bad = bad || (is_a && !(mine > 0.0));
      if (is_a) {   // (both copies of A's lanes write the same values: no branch on the row)
        double pcol[10];
#pragma unroll
        for (int i = 0; i < 9; ++i) pcol[i] = -r[i] * mine;
        pcol[9] = 0.0;
        ch_store10(S + LY.oP + k * CB + jc * 10, pcol);
        S[LY.oDi + k * 10 + jc] = mine * rsqrt_nr(mine);   // d^-1/2 = (1/d) (1/d)^-1/2
      } else if (is_p) {
        ch_store10(S + LY.oT + k * CB + jc * 10, tt);
        double an[10];
#pragma unroll
        for (int i = 0; i < 9; ++i) an[i] = r[9 + i];
        an[9] = 0.0;
        ch_store10(S + (share ? LY.oX : LY.oA + n * CB) + jc * 10, an);
      }
      ldl_signal(&cf_pub[side], t + 1, lane);
    }
    ldl_signal(&cf_cdone[side], 1, lane);
    if (side == 0) {
      // ---- the middle block: its diagonal block with both sweeps' updates, eliminated alone (nothing rides along)
      if (nR > 0) ch_wait_ge(&cf_cdone[1], 1, s_fail);
      double r[9];
      {
        double a[10], b[10];
        ch_load10(S + (j < 9 ? LY.oA + mid * CB + j * 10 : LY.oZ), a);
        ch_load10(S + ((j < 9 && nR > 0) ? LY.oX + j * 10 : LY.oZ), b);
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = a[i] + b[i];
      }
      double mine = 1.0;
      chain_eliminate<9>(r, mine, j);
      bad = bad || (j < 9 && !(mine > 0.0));
      if (is_a) {
        double pcol[10];
#pragma unroll
        for (int i = 0; i < 9; ++i) pcol[i] = -r[i] * mine;
        pcol[9] = 0.0;
        ch_store10(S + LY.oP + mid * CB + jc * 10, pcol);
        S[LY.oDi + mid * 10 + jc] = mine * rsqrt_nr(mine);
      }
      ldl_signal(&cf_mid, 1, lane);
    }
    if (__any(bad) && lane == 0) __hip_atomic_store(s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (stamps && lane == 0) {
      stamps[21 + side] = clock64();
      if (side == 0) {
        ts_[nst < 7 ? nst : 7] = stamps[21];
        for (int i = 0; i < 8; ++i) stamps[32 + i] = ts_[i];
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else if (wave < 2 + 2 * nchunk) {
    // ============================================================================ the column waves of a sweep: one column of
    // [pose couplings | right-hand side] per lane, for the whole sweep.  What every lane needs of a step — the 36 entries of
    // L^-1 and the 81 of (D^-1 R_C)^T — is NOT read 64-fold from LDS: each row of 16 lanes holds them once, spread over its lanes
    // (entry e in lane e & 15, register e >> 4: three + six loads per lane and step), and the products take them through DPP
    // row_newbcast like the chain wave's pivots.  (As broadcast loads in front of every row of products the step was 4300
    // cycles — the loads' latency, nine times in a row — against the chain wave's 2450, and the sweeps waited for this wave.)
    const int side = wave & 1, q = (wave >> 1) - 1;
    const int nst = side == 0 ? nL : nR;
    const int c = 64 * q + lane;
    const bool on = c < ncol;
    const int cc = on ? c : NP - 1;   // (a lane without a column works on the padding column of the rows: zeros in, zeros out, no branch)
    const int lam = lane & 15;
    int tofs[6], pofs[3];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int e = min(lam + 16 * t, 80), i = e / 9;
      tofs[t] = i * 10 + (e - 9 * i);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int e = min(lam + 16 * t, 35);
      int i = 1;
      while ((i + 1) * i / 2 <= e) ++i;   // e = i (i - 1) / 2 + jj, jj < i
      pofs[t] = (e - i * (i - 1) / 2) * 10 + i;
    }
    __builtin_amdgcn_s_setprio(2);
    double v[9];
    {
      const int k0 = side == 0 ? 0 : Ks - 1;
#pragma unroll
      for (int m = 0; m < 9; ++m) v[m] = S[LY.oN + (k0 * 9 + m) * NP + cc];
    }
    // y = L^-1 v; stored over the block's row of pose couplings, and the block is announced to the waves that form the pose
    // system's update (cf_y)
    auto apply = [&](int k, double (&y)[9]) {
      double pr[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) pr[t] = S[LY.oP + k * CB + pofs[t]];
      const double sdr = S[LY.oDi + k * 10 + (lam < 9 ? lam : 9)];   // d_m^-1/2 in lane m of every row of 16 lanes
#pragma unroll
      for (int i = 0; i < 9; ++i) y[i] = v[i];
      ChApplyP<1, 0>::run(y, pr, v);   // (starts at (1, 0), walks down column 0, then column 1 from (2, 1), ...)
      // stored scaled, Y~ = D^-1/2 Y: the pose system's update is then the plain product Y~^T Y~ (both operands of the matrix
      // core the same loads) and the back-substitution takes the scale once per entry
      double ys[9];
#pragma unroll
      for (int m = 0; m < 9; ++m) ys[m] = 0.0;
      fmac_bcast<0>(ys[0], sdr, y[0]);
      fmac_bcast<1>(ys[1], sdr, y[1]);
      fmac_bcast<2>(ys[2], sdr, y[2]);
      fmac_bcast<3>(ys[3], sdr, y[3]);
      fmac_bcast<4>(ys[4], sdr, y[4]);
      fmac_bcast<5>(ys[5], sdr, y[5]);
      fmac_bcast<6>(ys[6], sdr, y[6]);
      fmac_bcast<7>(ys[7], sdr, y[7]);
      fmac_bcast<8>(ys[8], sdr, y[8]);
#pragma unroll
      for (int m = 0; m < 9; ++m) S[LY.oN + (k * 9 + m) * NP + cc] = ys[m];
      asm volatile("" ::: "memory");
#ifdef LDL_SIGNAL_FENCE
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
      if (lane == 0) atomicAdd(&cf_y[k], 1);   // (behind this wave's stores of y: the LDS performs a wave's requests in order; ldl_signal's invariant)
    };
    for (int t = 0; t < nst; ++t) {
      const int k = side == 0 ? t : Ks - 1 - t, n = side == 0 ? k + 1 : k - 1;
      const bool share = (side == 1 && n == mid);
      // the column of the next block as assembled (requested in front of the wait)
      const int nbase = share ? LY.oZ : LY.oN + n * 9 * NP + cc;
      const int nstep = share ? 0 : NP;
      double w[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) w[i] = S[nbase + i * nstep];
      ch_wait_ge(&cf_pub[side], t + 1, s_fail);
      double tr[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) tr[u] = S[LY.oT + k * CB + tofs[u]];
      double y[9];
      apply(k, y);
      // N_n[:, c] - (D^-1 R_C)^T y.  (The sign rides on y, the operand that is NOT read through DPP: a register written by a VALU
      // instruction must not be read through DPP by one of the next two instructions, and inline assembly hides that from the
      // compiler's hazard recogniser — what is read through DPP here comes straight from LDS loads.)
      double ny[9];
#pragma unroll
      for (int m = 0; m < 9; ++m) ny[m] = -y[m];
      ChApplyT<0, 0>::run(w, tr, ny);
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = w[i];
    }
    if (side == 1) {
      if (nst > 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) S[LY.oX + CB + i * NP + cc] = v[i];
      }
      ldl_signal(&cf_ndone[1][q], 1, lane);
    } else {
      if (nR > 0) {
        ch_wait_ge(&cf_ndone[1][q], 1, s_fail);
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] += S[LY.oX + CB + i * NP + cc];
      }
      ch_wait_ge(&cf_mid, 1, s_fail);
      double y[9];
      apply(mid, y);
    }
    if (stamps && lane == 0 && q == 0) stamps[23 + side] = clock64();
    __builtin_amdgcn_s_setprio(0);
  } else if (wave >= NW - 2) {
    // ============================================================================ the last two waves, one per sweep: what the
    // back-substitution needs of every block that has a next block,  G_k = L_k^-T D_k^-1 R_C  (G[i][c'] = sum_{m >= i} L^-1[m][i]
    // (D^-1 R_C)[m][c'], L^-1 unit lower), as soon as the chain wave has published the block
    const int side = wave - (NW - 2);
    const int nst = side == 0 ? nL : nR;
    if (side == 0) {   // the filled ranges for the phases behind the sweeps (u of the back-substitution)
      int flo, fhi;
      ch_ranges(flo, fhi);
      if (lane < Ks) cf_lo[lane] = flo, cf_hi[lane] = fhi;
    }
    for (int t = 0; t < nst; ++t) {
      const int k = side == 0 ? t : Ks - 1 - t;
      ch_wait_ge(&cf_pub[side], t + 1, s_fail);
      for (int e = lane; e < 81; e += 64) {
        const int i = e / 9, cq = e - 9 * i;
        const double* pc_ = S + LY.oP + k * CB + i * 10;    // column i of L^-1: entry m at [m], m > i
        const double* tc_ = S + LY.oT + k * CB + cq * 10;   // column c' of D^-1 R_C: entry m at [m]
        double g = tc_[i];
        for (int m = i + 1; m < 9; ++m) g = fma(pc_[m], tc_[m], g);
        S[LY.oG + k * CB + i * 10 + cq] = g;
      }
    }
    if (stamps && lane == 0 && side == 0) stamps[26] = clock64();
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[17] = clock64();
  // ---- 2. the pose system's update
  //   S_pp -= sum_k Y~_k^T Y~_k,  rhs_p -= sum_k Y~_k^T y~_k   (Y~ = D^-1/2 Y as the column waves stored it, y~_k = its column Dp)
  // — a GEMM with K = 9 Ks, on the fp64 matrix core: 16 x 16 output tiles (I >= J; row Dp of the output is the right-hand side),
  // one per wave, the blocks in the order 0 .. Ks-1, a block skipped where its range misses the tile; the operands of the next
  // block are requested while the products of the current one run.
  // A lane = Y~_k[m][16 I + (l & 15)], B lane = Y~_k[m][16 J + (l & 15)], m = 4 s + (l >> 4)  (v_mfma_f64_16x16x4_f64).
  // With all sixteen waves of the workgroup runnable a SIMD issues about one instruction every 4 - 5 cycles IN TOTAL (measured:
  // waves without a tile took 1.3 - 1.9 us to walk through an empty loop next to the working ones): what counts in this phase is
  // the number of instructions, so a wave without a tile leaves at once and a block costs six loads, six address updates, three
  // products.  (Tried: the same WHILE the sweeps run, as the column waves announce the blocks — on four, then ten waves.  The
  // chain waves keep the fp64 pipe of their SIMDs busy back to back, the column waves half of the time: the update got its turn
  // when they were through and trailed the sweeps by 3 - 6 us.)
  {
    const int ntr = (Dp + 1 + 15) / 16, ntc = (Dp + 15) / 16;
    const int ntiles = ntc * (ntc + 1) / 2 + (ntr > ntc ? ntc : 0);
    for (int tix = wave; tix < ntiles; tix += NW) {
      // tile tix of the row-major lower triangle (rows 0 .. ntr-1, row I has min(I, ntc - 1) + 1 tiles)
      int I = 0, rem = tix;
      while (rem > (I < ntc - 1 ? I : ntc - 1)) {
        rem -= (I < ntc - 1 ? I : ntc - 1) + 1;
        ++I;
      }
      const int J = rem;
      const int lc = lane & 15, lk = lane >> 4;
      // which blocks reach this tile: lane k decides for block k
      unsigned um;
      {
        const int l = lane < Ks ? lane : 0;
        const int flo = cf_lo[l], fhi = cf_hi[l];
        const bool rows_in = (16 * I < fhi && 16 * I + 16 > flo) || (Dp >= 16 * I && Dp < 16 * I + 16);
        const bool cols_in = 16 * J < fhi && 16 * J + 16 > flo;
        um = (unsigned)__ballot(lane < Ks && rows_in && cols_in);
      }
      // this lane's operands of block 0: rows lk, 4 + lk and 8 of the block (a lane lk > 0 has no third row: it requests its
      // second one again and its third product is masked)
      const int a_at = LY.oN + lk * NP + min(16 * I + lc, NP - 1), b_at = LY.oN + lk * NP + min(16 * J + lc, NP - 1);
      const int r2 = lk == 0 ? 8 * NP : 4 * NP;
      const bool diag = I == J;
      ldl_v4 acc{0, 0, 0, 0};
      double a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
      auto request = [&](int k) {
        const double* ya = S + a_at + k * 9 * NP;
        a0 = ya[0], a1 = ya[4 * NP], a2 = ya[r2];
        if (!diag) {   // (uniform)
          const double* yb = S + b_at + k * 9 * NP;
          b0 = yb[0], b1 = yb[4 * NP], b2 = yb[r2];
        }
      };
      unsigned rest = um;
      if (rest) request(__builtin_ctz(rest));
      while (rest) {
        rest &= rest - 1;
        const double pa0 = a0, pa1 = a1, pa2 = lk == 0 ? a2 : 0.0;
        const double pb0 = diag ? a0 : b0, pb1 = diag ? a1 : b1, pb2 = lk == 0 ? (diag ? a2 : b2) : 0.0;
        if (rest) request(__builtin_ctz(rest));
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa0, pb0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1, pb1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2, pb2, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int p = 16 * I + 4 * rr + lk, qq = 16 * J + lc;
        if (p <= Dp && qq < Dp && p >= qq) S[LY.P.at(p, qq)] -= acc[rr];
      }
    }
    if (stamps && lane == 0) stamps[40 + wave] = clock64();   // (diagnostics: when each wave is through with its tiles)
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[18] = clock64();
  // ---- 3. the pose system (its work area aliases only its own blocks: the chain's arrays lie behind ldl16_area_doubles(Dp))
  ldl16_solve<NW>(S, Dp, tid, x_out, s_fail, stamps, 0, comp_mask);
  if (stamps && tid == 0) stamps[19] = clock64();
  // ---- 4. back-substitution of the speed/bias blocks.  u_k = y_k - Y_k x_p and a_k = L_k^-T D_k^-1 u_k for every block at once,
  //         then from the middle outwards  x_k = a_k - G_k x_next,  one wave per sweep: nine products per block
  __shared__ double ch_u[CH_MAX_KS * 9 + 8];
  __shared__ double ch_a[CH_MAX_KS * 9 + 8];
  {
    // u~ = y~ - Y~ x_p: a row of 16 lanes per entry (row m of block k): lane lc takes the columns lc, lc + 16, ... of the block's
    // range, all requested together; the row's sum by DPP (quad exchanges + row mirrors: VALU speed, fixed order).
    // (Tried: four waves, a lane per entry walking its columns — one wave per SIMD has the issue slots to itself, but every trip of
    // its loop waits for its own LDS loads: 3.3 us against 1.7.)
    const int e = tid >> 4, lc = tid & 15;
    for (int e0 = 0; e0 < Ks * 9; e0 += (NW * 64) / 16) {
      const int ee = e0 + e, ec = ee < Ks * 9 ? ee : 0;
      const int k = ec / 9;
      const int hi = cf_hi[k], lo = min(cf_lo[k], hi);
      const double* yr = S + LY.oN + ec * NP;
      double a = 0.0;
      for (int c0 = (lo & ~63) + lc; c0 < hi; c0 += 64) {
        double yv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cq = c0 + 16 * u, cl = cq < Dp ? cq : 0;
          yv[u] = yr[cl];
          xv[u] = x_out[cl];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cq = c0 + 16 * u;
          a = fma((cq >= lo && cq < hi) ? yv[u] : 0.0, xv[u], a);
        }
      }
      a += quad_xchg<0xB1>(a);
      a += quad_xchg<0x4E>(a);
      a += quad_xchg<0x141>(a);
      a += quad_xchg<0x140>(a);
      if (ee < Ks * 9 && lc == 0) ch_u[ee] = S[LY.oN + ee * NP + Dp] - a;
    }
  }
  if (stamps && (tid & 63) == 0) stamps[56 + wave] = clock64();   // (diagnostics: when each wave has its entries of u)
  __syncthreads();
  // a_k = L_k^-T D_k^-1/2 u~_k, every entry by a work-item of its own:  a_i = sum_{m >= i} L^-1[m][i] d_m^-1/2 u~_m  (L^-1 unit lower)
  for (int e = tid; e < Ks * 9; e += NW * 64) {
    const int k = e / 9, i = e - 9 * k;
    const double* pc_ = S + LY.oP + k * CB + i * 10;   // column i of L^-1: entry m at [m], m > i
    const double* dk = S + LY.oDi + k * 10;
    double pv[9], uv[9], dv[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) pv[m] = pc_[m], uv[m] = ch_u[k * 9 + m], dv[m] = dk[m];
    double a = 0.0;
#pragma unroll
    for (int m = 0; m < 9; ++m) a = fma(m > i ? pv[m] : (m == i ? 1.0 : 0.0), uv[m] * dv[m], a);
    ch_a[e] = a;
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[27] = clock64();
  if (wave < 2) {
    const int side = wave;
    const int j = lane & 15;
    const bool on = lane < 9;
    const int jz = on ? j : 0;
    const int nst = side == 0 ? nL : nR;
    // (both sweeps start from the middle block's x: each wave takes it itself)
    double xn = on ? ch_a[mid * 9 + jz] : 0.0;
    if (on) x_out[Dp + 9 * mid + jz] = xn;
    // the next block's row of G and its a: requested one block ahead
    double gn[10], an;
    {
      const int k0 = side == 0 ? nst - 1 : Ks - nst;
      ch_load10(S + ((on && nst > 0) ? LY.oG + k0 * CB + jz * 10 : LY.oZ), gn);
      an = (on && nst > 0) ? ch_a[k0 * 9 + jz] : 0.0;
    }
    for (int t = nst - 1; t >= 0; --t) {
      const int k = side == 0 ? t : Ks - 1 - t;
      double g_[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) g_[u] = gn[u];
      const double a0 = an;
      {
        const int k1 = side == 0 ? t - 1 : Ks - t;
        ch_load10(S + ((on && t > 0) ? LY.oG + k1 * CB + jz * 10 : LY.oZ), gn);
        an = (on && t > 0) ? ch_a[k1 * 9 + jz] : 0.0;
      }
      const double xm = -xn;
      double s0 = a0, s1 = 0.0;
      // (xm is a fresh VALU result: the first DPP read keeps its distance inside its own asm statement)
      fmac_bcast_nop<0>(s0, xm, g_[0]);
      fmac_bcast<1>(s1, xm, g_[1]);
      fmac_bcast<2>(s0, xm, g_[2]);
      fmac_bcast<3>(s1, xm, g_[3]);
      fmac_bcast<4>(s0, xm, g_[4]);
      fmac_bcast<5>(s1, xm, g_[5]);
      fmac_bcast<6>(s0, xm, g_[6]);
      fmac_bcast<7>(s1, xm, g_[7]);
      fmac_bcast<8>(s0, xm, g_[8]);
      xn = s0 + s1;
      if (on) x_out[Dp + 9 * k + jz] = xn;
    }
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[20] = clock64();
}

}  // namespace ba
