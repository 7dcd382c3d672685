// Dense solve of the reduced camera system of one window by ONE workgroup on the fp64 matrix core (gfx950):
// blocked LDL^T with 16-wide panels, the matrix resident in MFMA accumulator registers, forward substitution riding
// along as one more column, back-substitution from the register-resident factor.  No workgroup barrier between the
// first and the last instruction of the factorisation: the waves hand data to each other through LDS flags.
//
// This is the algebra of the reduced-camera solve of SPARSE_SCHUR (Estimator.cpp:854; in-tree analogue
// MarginalizationError.cpp:617-689): S x = b with S symmetric positive definite, D <= 175.
//
// Why this shape (profiles/r02_notes.md: the previous 6-wide right-looking Cholesky in LDS spent 25 block columns x
// 2.3 us in barriers and LDS round trips, not in flops):
//   * 16-wide panels: nb = ceil((D+1)/16) <= 11 dependent steps;
//   * upper-triangular 16x16 blocks, each owned for the whole factorisation by one of the waves 1..NW-1 and held in the
//     accumulator layout of v_mfma_f64_16x16x4_f64 (lane l, register r = element (row (l>>4)+4r, column l&15)).  Feeding
//     the accumulator registers of U as the A operand and those of V as the B operand of four MFMAs yields U^T V in
//     accumulator layout again, so the trailing update  A_IJ -= R_I^T D^-1 R_J  needs no transposition and the matrix
//     never returns to LDS; only the current panel row R_J = L_kk^-1 A_kJ goes through LDS (double buffered);
//   * the dependent chain — the 16x16 diagonal blocks — is carried by wave 0 alone and never waits at a barrier:
//     column per lane (16 lanes), pivot column broadcast with DPP row_newbcast, Gauss elimination in LDL^T form
//     (reciprocals, no square roots) on [A | I], so that the unit-lower inverse L^-1 falls out of the same instruction
//     stream; wave 0 then forms the next diagonal block privately (R = L^-1 A_k,k+1 and P_k+1 -= R^T D^-1 R, eight
//     MFMAs on copies handed over by the owners) while the other waves run the panel GEMM with the published inverse
//     and the trailing update one step behind;
//   * the right-hand side is column D of the matrix (a spare column of the last block): the elimination turns it into
//     L^-1 b for free, and with x[D] = -1 the back-substitution is one uniform sweep  t_K = -sum_J R_KJ x_J, whose
//     critical path (the super-diagonal blocks) again stays inside wave 0.
//
// LDS layout of the assembled system (what the caller fills): block (I <= J) at blk(I, J), 256 doubles, element
// (r, c) of the block at (r>>2)*64 + (r&3)*16 + c  (= accumulator register r>>2 of lane (r&3)*16 + c).  Only entries
// with row <= column are referenced (diagonal blocks: upper triangle).  Everything that is not written must be zero.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <type_traits>

#include "ba_device.hpp"

namespace ba {

typedef double ldl_v4 __attribute__((ext_vector_type(4)));

constexpr int LDL_MAX_NB = 11;   // D + 1 <= 176
// A/B switches of the diagonal chain (tests/micro/ldl16.hip builds the variants): LDL_PIVOT 0 = reciprocal first, then the
// multiplier (default); 1 = the multiplier and the reciprocal side by side (measured slower, 2055 against 2006 cycles per block: the
// elimination is bound by instruction issue, not by the chain)
#ifndef LDL_PIVOT
#define LDL_PIVOT 0
#endif
// LDL_W0_REORDER 1 = wave 0 reads its own operands back in front of the flag's release (measured: no gain; default 0)
#ifndef LDL_W0_REORDER
#define LDL_W0_REORDER 0
#endif
// LDL_ELIM16 1 = the elimination of a diagonal block runs with lanes 16 .. 63 masked off
#ifndef LDL_ELIM16
#define LDL_ELIM16 0
#endif
#ifndef LDL_ASSIGN_COLUMNS
#define LDL_ASSIGN_COLUMNS 0
#endif

struct L16 {
  int nb;
  // Solver ordering.  The reduced system numbers the pose-type blocks first (Dp rows), then the speed/bias blocks; the solver
  // eliminates the speed/bias part first: their block-tridiagonal coupling leaves whole 16x16 blocks of the first panel rows
  // zero (the matrix core, 64 cycles per 16x16x4 product on three SIMDs, is what the first block steps wait for when every
  // block is treated as dense), and the solver skips zero blocks.  Reduced index i < Dp sits at solver row i + Ds, i >= Dp at
  // i - Dp (Ds = Dn - Dp); index Dn — the right-hand side column — and beyond keep their places.  Ds = 0: no reordering.
  int Ds = 0;
  int Dn = INT_MAX;
  __host__ __device__ __forceinline__ int perm(int i) const { return i >= Dn ? i : (i < Dn - Ds ? i + Ds : i - (Dn - Ds)); }
  __host__ __device__ __forceinline__ int unperm(int p) const { return p >= Dn ? p : (p < Ds ? p + (Dn - Ds) : p - Ds); }
  __host__ __device__ __forceinline__ int blk(int I, int J) const { return (I * nb - (I * (I - 1)) / 2 + (J - I)) * 256; }   // I <= J
  // scalar index of solver entry (p, q), p >= q: stored at the mirrored position (q, p) of the upper block triangle
  __host__ __device__ __forceinline__ int at_solver(int p, int q) const {
    const int I = q >> 4, J = p >> 4, r = q & 15, c = p & 15;
    return blk(I, J) + (r >> 2) * 64 + (r & 3) * 16 + c;
  }
  // scalar index of entry (i, j) in REDUCED coordinates (the assembly code's; either order)
  __host__ __device__ __forceinline__ int at(int i, int j) const {
    const int p = perm(i), q = perm(j);
    return p >= q ? at_solver(p, q) : at_solver(q, p);
  }
  __host__ __device__ __forceinline__ int sym(int i, int j) const { return at(i, j); }
  __host__ __device__ static int blocks(int nb) { return nb * (nb + 1) / 2; }
};

constexpr int LDL_RS = 18;   // stride of the 16-vectors that one lane reads or writes as a whole (conflict-free b128 accesses)
constexpr int LDL_XB = 16 * LDL_RS;
// doubles of LDS the solver needs from the start of the assembly buffer (it overwrites the assembled matrix)
__host__ __device__ inline int ldl16_work_doubles(int nb) {
  return 2 * nb * 256 /* Rp */ + 3 * nb * LDL_XB /* Xs, XB, Rsup */ + 4 * 256 /* qbuf, dpart */ + 2 * nb * 16 /* dinv, xv */ + LDL_XB /* Pt */;
}
__host__ __device__ inline int ldl16_nb(int D) { return (D + 1 + 15) / 16; }
// size of the matrix area (doubles): the assembled blocks or the work area, whichever is larger
__host__ __device__ inline int ldl16_area_doubles(int D) {
  const int nb = ldl16_nb(D);
  const int a = L16::blocks(nb) * 256, w = ldl16_work_doubles(nb);
  return a > w ? a : w;
}

__device__ __forceinline__ double rcp_nr(double d) {   // 1/d: hardware estimate + two Newton steps
  double y = __builtin_amdgcn_rcp(d);
  y = fma(y, fma(-d, y, 1.0), y);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
}
// (experiment, -DLDL_NR1: one Newton step — <= 11 ulp, tests/micro/rcp_acc.hip — in the blocks that are not compensated; two
//  instructions fewer per pivot.  Measured against the long double referee: profiles/r05_notes.md.  Not the default.)
__device__ __forceinline__ double rcp_nr1(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
}

// DPP row_newbcast (gfx90a+): lane K of the caller's row of 16 lanes, as the first source of a 64-bit VALU operation.
// Written as volatile inline assembly (one instruction per element instead of v_mov_b64_dpp + copy + v_fmac), which the
// compiler's hazard recogniser does not see through: a VGPR written by a VALU instruction must not be read through DPP by
// one of the next two instructions.  The callers keep that distance by construction (volatile statements stay in program
// order) and use the *_nop forms where they cannot.
template <int K>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& src, const double& mul) {   // acc += src[lane K] * mul
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int K>
__device__ __forceinline__ void fmac_bcast_nop(double& acc, const double& src, const double& mul) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int K>
__device__ __forceinline__ double bcast_nop(const double& src) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(K));
  return d;
}

// ---- flags in LDS (workgroup scope): monotone counters written by one wave, polled by others
__device__ __forceinline__ void ldl_wait_ge(const int* flag, int need) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void ldl_wait_eq(const int* flag, int need) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != need) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// bit l of the result: flags[l] >= need (l < n; the other bits are set).  One LDS round trip for all the flags.
__device__ __forceinline__ unsigned long long ldl_ready_mask(const int* flags, int n, int need, int lane) {
  const int v = lane < n ? __hip_atomic_load(&flags[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : INT_MAX;
  return __ballot(v >= need);
}
// the same for flags that carry a property of what they announce in bit 0 (value = count << 1 | bit): *bits receives bit 0 of
// every flag (meaningful where the flag is ready)
__device__ __forceinline__ unsigned long long ldl_ready_mask_bit(const int* flags, int n, int need2, int lane, unsigned long long* bits) {
  const int v = lane < n ? __hip_atomic_load(&flags[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : INT_MAX - 1;
  *bits = __ballot((v & 1) != 0);
  return __ballot(v >= need2);
}
// The calling wave's LDS writes become visible before the flag does.  No wait in between: the LDS executes the requests of one
// wave in the order they were issued (their answers return in that order), so a flag written behind the data is performed
// behind the data; a workgroup-scope release fence would have the wave sit through the round trip of its own writes (~100
// cycles in every hand-over).  The asm statement keeps the compiler from moving the flag's store in front of the data's.
//
// INVARIANT (everything published through ldl_signal / ch_signal must obey it): the payload is in LDS and every one of its
// stores was issued by the wave that signals.  A payload in global memory, or one that a second wave helps to write, needs the
// release fence: build with -DLDL_SIGNAL_FENCE (scripts/build_variant.sh fenced -DLDL_SIGNAL_FENCE) — the fenced library gives
// the same bits (tests/test_gpu_variants.py runs the solver tests against it when the variant has been built).
__device__ __forceinline__ void ldl_signal(int* flag, int v, int lane) {
#ifdef LDL_SIGNAL_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#else
  asm volatile("" ::: "memory");
#endif
  if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// The A operand of the panel products of step k,  L_kk^-1 = Bh D^-1  (unit diagonal, exact zeros above it), and -1/d, for the
// rows (lane >> 4) + 4 q of the product's inner dimension.  Every value is requested before the first is used: written as
// `a[q] = am > ak ? Xs[..] * dinv[..] : ..` the compiler put each load inside its own branch, four LDS round trips in a row
// in every wave that forms a panel block (and in wave 0's chain).
__device__ __forceinline__ void ldl_operand(const double* Xk, const double* dk, int lane, double (&a)[4], double (&dq)[4]) {
  const int am = lane & 15;
  double xh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ak = (lane >> 4) + 4 * q;
    dq[q] = -dk[ak];
    xh[q] = Xk[ak * LDL_RS + am];
  }
  asm volatile("" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ak = (lane >> 4) + 4 * q;
    const double v = xh[q] * dq[q];   // (the published column holds -Bh below the diagonal)
    a[q] = am > ak ? v : (am == ak ? 1.0 : 0.0);
  }
}

// One elimination step of the 16x16 diagonal block held column per lane, r[i] = entry (i, lane & 15), A and the inverse
// of its unit-lower factor IN THE SAME REGISTERS: before step K the lanes j >= K hold the columns of the partially
// eliminated A, the lanes j < K the columns of  G = -L^-1 D  below the diagonal (L^-1 so far, every column scaled by its pivot,
// sign flipped: the consumers fold the sign into the masks they apply anyway; the diagonal entry r[j] of lane j stays d_j).
// With m_i = A[i][K] (lane K) and 1/d:
//   lanes j > K:  A[i][j] -= m_i A[K][j] / d          (Schur complement)
//   lanes j < K:  G[i][j] -= m_i G[K][j] / d          (row operation on the inverse: the same formula, r[K] is G[K][j])
//   lane  j = K:  G[i][K]  = m_i                       (the lane keeps its registers: v = 0)
// i.e. ONE v_fmac_f64_dpp per row for the factorisation and the inverse together.  What is left above the diagonal of
// lane j (rows i < j: entries of D L^T) is dead and masked when the block is published.
// COMP (compensated products, for the blocks the window record marks: ldl_comp).  Row i's multiplier comes from the pivot
// COLUMN (lane K's r[i]) while the row factor v = -A[K][j] / d comes from the pivot ROW (lane j's r[K]): column and row agree in
// exact arithmetic only — entry (i, j) takes fl(a_iK fl(a_Kj / d)), its mirror image fl(a_jK fl(a_Ki / d)), one ulp of the
// PRODUCT apart.  Behind a pivot that cancels many digits — the yaw prior of the first pose, 1e16 n n^T across the three
// rotation rows, leaves pivots 7.5e13, 2.9e6, 4.1e8 — that is 1e-9 of what is left, the elimination turns into the LU
// factorisation of a matrix that is not symmetric any more, and the panel rows (L^-1 from the columns) stop matching the
// trailing updates (which assume D L^T from the rows).  Measured against the oracle built in long double: the Gauss-Newton step
// of the far-start DOGLEG case of test_dogleg_rejected_steps 1e-7 from the extended-precision solution of the same system
// (an unblocked Cholesky: 4e-9), the cost after 20 iterations 3e-7 (tools/gpu_referee_*.py, profiles/r05_notes.md).  With
// v = vh + vl, vl = fma(-A[K][j], 1/d, -vh) the part of the product that vh rounded away, and a second fmac per row,
// entry and mirror image receive a_iK a_Kj / d to twice the working precision — the first fmac cancels exactly, the second
// restores what vh had lost — and agree to an ulp of the RESULT: 1.2e-9 on the same case (the fp64 oracle: 3.5e-9).
// 16 -> 31 instructions per pivot on the rows, +370 cycles per block.
// d = r[K] of lane K (already broadcast).  FULL: all 16 pivots exist; otherwise act = (K < npiv) and an inactive step
// changes nothing.  Lane K keeps 1 / d_K (0 when inactive) in `mine`.  Returns the next pivot, broadcast as soon as its
// entry is final so that its reciprocal (the dependent chain) overlaps the remaining updates of this step.
// GUARD: a pivot that is not positive is replaced by 1 (everything stays finite; the caller reports the failure).
template <int K, bool FULL, bool GUARD = false, bool COMP = false>
__device__ __forceinline__ double ldl16_pivot(double (&r)[16], double d, bool act, double& mine, int j, bool* bad = nullptr) {
  if constexpr (GUARD) {
    const bool neg = !(d > 0.0);
    *bad = *bad || neg;
    d = neg ? 1.0 : d;
  }
#if LDL_PIVOT == 0
#ifdef LDL_NR1
  double rd = (COMP || GUARD) ? rcp_nr(d) : rcp_nr1(d);   // (GUARD: the tiled solver keeps two steps)
#else
  double rd = rcp_nr(d);
#endif
  if (!FULL) rd = act ? rd : 0.0;
  const bool me = (j == K);
  double v = -r[K] * rd;
  double vl = 0.0;
  if constexpr (COMP) vl = fma(-r[K], rd, -v);   // what the product above rounded away (exact)
#else
  static_assert(!COMP, "the compensated elimination is written for LDL_PIVOT 0");
  double vl = 0.0;
  // 1/d = y0 (1 + e)(1 + e^2), y0 the hardware estimate and e = 1 - d y0 (the two Newton steps of rcp_nr written out): the
  // multiplier -r[K] / d takes the same three factors one by one, so that it is ready four dependent instructions behind d
  // (estimate, e | -r[K] y0, two fused multiply-adds) instead of seven (estimate, two Newton steps, product).  The reciprocal
  // itself (kept by lane K) is off the chain.
  const double y0 = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, y0, 1.0);
  const double u0 = -r[K] * y0;
  const double e2 = e * e;
  const double u1 = fma(u0, e, u0);
  double v = fma(u1, e2, u1);
  const double y1 = fma(y0, e, y0);
  double rd = fma(y1, e2, y1);
  if (!FULL) {
    rd = act ? rd : 0.0;
    v = act ? v : 0.0;
  }
  const bool me = (j == K);
#endif
  v = me ? 0.0 : v;
  if constexpr (COMP) vl = me ? 0.0 : vl;
  mine = me ? rd : mine;
  double dn = 1.0;
  if constexpr (K < 15) {
    fmac_bcast<K>(r[K + 1], r[K + 1], v);
    if constexpr (COMP) fmac_bcast_nop<K>(r[K + 1], r[K + 1], vl);
    dn = bcast_nop<K + 1>(r[K + 1]);
  }
#pragma unroll
  for (int i = K + 2; i < 16; ++i) fmac_bcast<K>(r[i], r[i], v);
  if constexpr (COMP) {
    if constexpr (K >= 11) asm volatile("s_nop 1");   // (fewer than three rows left: keep the DPP distance to the writes above)
#pragma unroll
    for (int i = K + 2; i < 16; ++i) fmac_bcast<K>(r[i], r[i], vl);
  }
  return dn;
}
struct LdlNoHook { __device__ __forceinline__ void operator()() const {} };
// (mid: called between the pivots 9 and 10 — wave 0 issues the requests for the hand-overs of its step there, late enough for
// them to have arrived in the common case and early enough for the answers to be there when the elimination ends)
template <bool FULL, bool GUARD = false, class HOOK = LdlNoHook, bool COMP = false>
__device__ __forceinline__ void ldl16_eliminate(double (&r)[16], int npiv, double& mine, int j, bool* bad = nullptr, HOOK mid = HOOK()) {
  double d = bcast_nop<0>(r[0]);
  d = ldl16_pivot<0, FULL, GUARD, COMP>(r, d, 0 < npiv, mine, j, bad);
  d = ldl16_pivot<1, FULL, GUARD, COMP>(r, d, 1 < npiv, mine, j, bad);
  d = ldl16_pivot<2, FULL, GUARD, COMP>(r, d, 2 < npiv, mine, j, bad);
  d = ldl16_pivot<3, FULL, GUARD, COMP>(r, d, 3 < npiv, mine, j, bad);
  d = ldl16_pivot<4, FULL, GUARD, COMP>(r, d, 4 < npiv, mine, j, bad);
  d = ldl16_pivot<5, FULL, GUARD, COMP>(r, d, 5 < npiv, mine, j, bad);
  d = ldl16_pivot<6, FULL, GUARD, COMP>(r, d, 6 < npiv, mine, j, bad);
  d = ldl16_pivot<7, FULL, GUARD, COMP>(r, d, 7 < npiv, mine, j, bad);
  d = ldl16_pivot<8, FULL, GUARD, COMP>(r, d, 8 < npiv, mine, j, bad);
  d = ldl16_pivot<9, FULL, GUARD, COMP>(r, d, 9 < npiv, mine, j, bad);
#ifndef LDL_HOOK_AT
#define LDL_HOOK_AT 9
#endif
  if (LDL_HOOK_AT == 9) mid();
  d = ldl16_pivot<10, FULL, GUARD, COMP>(r, d, 10 < npiv, mine, j, bad);
  d = ldl16_pivot<11, FULL, GUARD, COMP>(r, d, 11 < npiv, mine, j, bad);
  d = ldl16_pivot<12, FULL, GUARD, COMP>(r, d, 12 < npiv, mine, j, bad);
  if (LDL_HOOK_AT == 12) mid();
  d = ldl16_pivot<13, FULL, GUARD, COMP>(r, d, 13 < npiv, mine, j, bad);
  d = ldl16_pivot<14, FULL, GUARD, COMP>(r, d, 14 < npiv, mine, j, bad);
  d = ldl16_pivot<15, FULL, GUARD, COMP>(r, d, 15 < npiv, mine, j, bad);
  if (LDL_HOOK_AT == 15) mid();
}

// NW waves (NW * 64 threads), all of which must call.  S: the assembled system (see above) for an nb = ldl16_nb(D)
// block matrix whose column D holds the right-hand side.  x_out (LDS, >= D doubles, outside the matrix area) receives the
// solution.  *s_fail (LDS int, zeroed by the caller before a barrier) is set when a pivot is not positive.
// Ends with a barrier: x_out and *s_fail are visible to every thread on return.
// comp_mask: bit b set = diagonal block b is eliminated with compensated products (ldl16_pivot<.., COMP>: +370 cycles per block).
template <int NW>
__device__ __forceinline__ void ldl16_solve(double* S, int D, int tid, double* x_out, int* s_fail, long long* stamps = nullptr, int Ds = 0,
                                            unsigned comp_mask = 0) {
  // Wave 0 carries the diagonal chain at raised priority and wants its SIMD for itself: the waves that share it (4, 8, 12:
  // waves w, w+4, w+8, w+12 of a workgroup sit on one SIMD, tests/micro/hwid.hip) would be starved exactly when the chain
  // needs their hand-offs (measured: a flag seen 5500 cycles late), so they own nothing and go straight to the last barrier.
  static_assert(NW == 16, "wave roles below assume 16 waves");
  static_assert(LDL_MAX_NB <= 12, "two block columns per near-diagonal wave");
  constexpr int NREG = 12;                                            // waves that own blocks: (wave & 3) != 0
  constexpr int SLOTS = (LDL_MAX_NB * (LDL_MAX_NB + 1) / 2 + NREG - 1) / NREG;
  const int nb = ldl16_nb(D);
  const L16 LY{nb, Ds, D};   // (S is assembled in the solver's ordering, L16::perm; x_out is written in the caller's)
  const int nblk = L16::blocks(nb);
  const int npl = D - 16 * (nb - 1);                                  // pivots of the last block (its column npl is the rhs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Inside the step loops the lane index is taken through an empty asm statement once per iteration: what is derived from it
  // (LDS addresses, the sixteen `lane == K` masks of the elimination) is then recomputed per step — a few VALU instructions —
  // instead of being hoisted in front of the loops, kept live across them and spilled (every reload of a spilled address
  // inside a step is a scratch-memory round trip on the chain).
#define LDL_LANE() [&]() { int l_ = tid & 63; asm volatile("" : "+v"(l_)); return l_; }()
  const int ridx = (wave >> 2) * 3 + (wave & 3) - 1;                  // 0..11 for the block-owning waves
  // work area (aliases the assembled matrix once every block sits in registers)
  double* Rp = S;                          // [2][nb][256] panel row R_J of step kb in buffer kb & 1, accumulator layout
  double* Xs = Rp + 2 * nb * 256;          // [nb][16][LDL_RS] -Bh = -L_II^-1 D_I below the diagonal, d_j on it, as wave 0 leaves it: column j at j * LDL_RS (rows < j: dead)
  double* XB = Xs + nb * LDL_XB;           // [nb][16][LDL_RS] D^-1 Bh^T ... for the back-substitution: entry (i, j) of D^-1 Bh D^-1 masked, at j * LDL_RS + i
  double* Rsup = XB + nb * LDL_XB;         // [nb][16][LDL_RS] wave 0's copies of R_(K,K+1), row-major
  double* qbuf = Rsup + nb * LDL_XB;       // [2][256] block (I, I+1) with all its updates, accumulator layout
  double* dpart = qbuf + 512;              // [2][256] diagonal block I with the updates k <= I - 2, accumulator layout
  double* dinv = dpart + 512;              // [nb][16]
  double* xv = dinv + nb * 16;             // [nb][16] solution incl. x[D] = -1
  double* Pt = xv + nb * 16;               // [16][LDL_RS] wave 0's own: the next diagonal block on its way from the accumulator layout to column per lane
  double* tcon = Rp;                       // [nb][nb][16] back-substitution: R_KJ x_J, one slot per block (the panel rows are dead by then)
  double* tsv = tcon + nb * nb * 16;       // [nb][16] minus the sum of the slots of a block row, in the fixed order (by wave 4)
  // flags (outside the matrix area: they are zeroed before the assembled matrix is dead)
  __shared__ int f_xready;                 // k + 1: Bh_k and 1/d_k published
  // [m & 1] = m + 1: block (m, m+1) complete in qbuf[m & 1] / diagonal block m with the updates <= m - 2 in dpart[m & 1].
  // One flag per buffer, not one counter: the hand-offs of consecutive steps come from different waves and may complete
  // out of order (the hand-off for step m + 1 does not depend on wave 0 having consumed the one for step m).
  __shared__ int f_hq[2], f_hp[2];
  // [k & 1][J] (k + 1) << 1 | z: R_J of step k published in Rp[k & 1]; z = 1: the block is zero and nothing was written (the
  // consumers skip their products with it).  One set of flags per buffer: a flag keeps its step's z until the buffer is reused.
  __shared__ int f_prdy[2][LDL_MAX_NB];
  __shared__ int f_tdn[16];                // [wave] k + 1: the wave has finished its trailing updates of step k
  __shared__ int f_x;                      // back-substitution: number of solved blocks (from the last one)
  __shared__ int f_tc[LDL_MAX_NB];         // [K] number of slots tcon[K][.] written
  __shared__ int f_xb[LDL_MAX_NB];         // [K] 1: XB[K] prepared
  __shared__ int f_ts[LDL_MAX_NB];         // [K] 1: the slots of block row K summed (tsv[K])
#define LDL_STAMP(k) do { if (stamps && tid == 0) stamps[k] = clock64(); } while (0)
  // diagnostics (tests/micro/ldl16.hip -DLDL_TS_STEP=k): time stamps of wave 0's step k, kept in scalar registers and stored
  // behind the factorisation (a stamp written where it is taken costs ~150 cycles and delays the chain it measures)
#ifdef LDL_TS_ALL   // loop top | hand-overs requested again .. both there, every step, through LDS (costs ~250 cycles per step)
  __shared__ long long ts_all[3][16];
#define LDL_TSA(i) do { if (lane == 0) ts_all[i][kb] = clock64(); } while (0)
#else
#define LDL_TSA(i) do { } while (0)
#endif
#ifdef LDL_OTS_WAVE   // the same for an owner wave (LDL_OTS_WAVE) in step LDL_TS_STEP; stamps 0..15, a stamp index used twice keeps the later time
  long long ots_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define LDL_OTS(i) do { if (kb == LDL_TS_STEP && wave == LDL_OTS_WAVE) ots_[i] = clock64(); } while (0)
#else
#define LDL_OTS(i) do { } while (0)
#endif
#ifdef LDL_TS_STEP
  long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LDL_TS(i) do { if (kb == LDL_TS_STEP) ts_[i] = clock64(); } while (0)
#else
#define LDL_TS(i) do { } while (0)
#endif
#ifdef LDL_TRACE   // diagnostics (tests/micro/ldl16.hip): per-wave event log {clock, code} in LDS, dumped behind the stamps
  __shared__ long long tr_log[16][64];
  int tr_n = 0;
#define LDL_EV(code) do { if (stamps && lane == 0 && tr_n < 64) tr_log[wave][tr_n++] = (clock64() << 12) | (unsigned)(code); } while (0)
#else
#define LDL_EV(code) do { } while (0)
#endif
  LDL_STAMP(0);
  if (tid == 0) {
    f_xready = 0;
    f_hq[0] = f_hq[1] = 0;
    f_hp[0] = f_hp[1] = 0;
    f_x = 0;
  }
  if (tid < LDL_MAX_NB) {
    f_prdy[0][tid] = 0;
    f_prdy[1][tid] = 0;
    f_tc[tid] = 0;
    f_xb[tid] = 0;
    f_ts[tid] = 0;
  }
  if (tid < 16) f_tdn[tid] = 0;

  if (wave > 0 && (wave & 3) == 0) {
    __syncthreads();   // (the barrier after the initial loads)
    if (wave == 4 && nb >= 3) {
      // Back-substitution, off the chain: minus the sum of the slots of block row K (the products R_KJ x_J, J >= K + 2, of the
      // block-owning waves) in the fixed order J = nb-1, nb-2, ..., for wave 0 to start its t_K from.  Wave 0 used to do this
      // itself: a poll of the slot counter, nine loads and nine selected subtractions per step, in front of its chain.
      // (The wave shares wave 0's SIMD: it sleeps until the back-substitution has started.)
      while (__hip_atomic_load(&f_x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 1) __builtin_amdgcn_s_sleep(8);
      const int j = lane & 15;
      for (int K = nb - 3; K >= 0; --K) {
        ldl_wait_ge(&f_tc[K], nb - K - 2);
        double tc[LDL_MAX_NB - 2];
#pragma unroll
        for (int u = 0; u < LDL_MAX_NB - 2; ++u) {
          const int J = nb - 1 - u;
          const int Jc = J >= K + 2 ? J : K + 2;   // (clamped: a load, not a branch; the copy is not added)
          tc[u] = tcon[(K * nb + Jc) * 16 + j];
        }
        double ts = 0.0;
#pragma unroll
        for (int u = 0; u < LDL_MAX_NB - 2; ++u) ts -= (nb - 1 - u >= K + 2) ? tc[u] : 0.0;
        if (lane < 16) tsv[K * 16 + lane] = ts;
        ldl_signal(&f_ts[K], 1, lane);
      }
    }
  } else if (wave > 0) {
    // =============================================================================== the twelve block-owning waves
    ldl_v4 acc[SLOTS];
    // the block coordinates of the slots, four bits each (15 = no block): two scalar registers instead of twelve (which spilled)
    static_assert(LDL_MAX_NB < 15 && SLOTS <= 8, "four bits per coordinate");
    unsigned pkI = 0xFFFFFFFFu, pkJ = 0xFFFFFFFFu;
    int lastStep = -1;   // last step in which this wave has panel or trailing work
    // Which blocks a wave owns: lane s works out slot s, the results travel by v_readlane.
    int uI, uJ;
#if LDL_ASSIGN_COLUMNS
    // (variant, measured slower: the six waves 1-3, 5-7 own the blocks next to the diagonal by block COLUMN — wave ridx the columns
    // c = ridx and ridx + 6, of each the blocks (c-2, c), (c-1, c), (c, c), so that what the chain waits for in step k is the
    // business of one wave — and the other six share the blocks (I, J >= I + 3).  The column that carries the right-hand side is
    // dense in every step: its wave falls behind the chain.)
    if (ridx < NREG / 2) {
      const int c = ridx + (NREG / 2) * (lane / 3), d = 2 - lane % 3;
      uI = (c < nb && c - d >= 0 && lane < SLOTS) ? c - d : nb;
      uJ = c;
    } else {
      int r = lane * (NREG / 2) + (ridx - NREG / 2);
      uI = 0;
      while (uI < nb && r >= max(0, nb - uI - 3)) {
        r -= max(0, nb - uI - 3);
        ++uI;
      }
      uJ = uI + 3 + r;
    }
#else
    // block number b = lane * NREG + ridx of the row-major upper triangle: row I starts at first(I) = I nb - I (I - 1) / 2, so
    // I = floor((2 nb + 1 - sqrt((2 nb + 1)^2 - 8 b)) / 2) — in single precision, put right by at most one either way.  (As a
    // loop over the rows, every lane subtracting row lengths until its number fits: 11 divergent iterations, 180 instructions
    // in front of the first load of every solve.)
    {
      const int b = lane * NREG + ridx;
      const float t = (float)(2 * nb + 1);
      int I = (int)((t - sqrtf(fmaxf(t * t - 8.0f * (float)b, 0.0f))) * 0.5f);
      I = max(0, min(I, nb - 1));
      if (I * nb - (I * (I - 1)) / 2 > b) --I;
      if ((I + 1) * nb - ((I + 1) * I) / 2 <= b) ++I;
      uI = b < nblk ? I : nb;
      uJ = I + (b - (I * nb - (I * (I - 1)) / 2));
    }
#endif
    const int u_lane = lane;
    // the steps in which slot `lane` has trailing work: kb < u_tl (a diagonal block takes its last update, that of step I - 1,
    // inside wave 0)
    const int u_tl = uI < nb ? (uJ > uI ? uI : uI - 1) : -1;
    // The blocks into the accumulator registers.  Every slot is loaded, without a branch: a slot without a block reads block
    // (0, 0) (its registers are never used), and the element a lane reads of a diagonal block — only the upper triangle is
    // assembled, the accumulators hold the full symmetric block — is a select between two offsets that depend on the lane
    // alone.  All 24 loads are in flight together.  (With `if (slot has a block) { if (diagonal) .. else .. }` the compiler
    // loaded every slot into temporaries and waited for them slot by slot.)
    {
      const int ln = LDL_LANE();
      int dofs[4];   // offset of element (row (ln >> 4) + 4 r, column ln & 15) of a diagonal block, read at its mirror image above the diagonal
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (ln >> 4) + 4 * r, col = ln & 15;
        const int lo = row < col ? row : col, hi = row < col ? col : row;
        dofs[r] = (lo >> 2) * 64 + (lo & 3) * 16 + hi;
      }
      int ls = -1;
#pragma unroll
      for (int s = SLOTS - 1; s >= 0; --s) {
        const int I = __builtin_amdgcn_readlane(uI, s), J = __builtin_amdgcn_readlane(uJ, s);
        const bool on = I < nb;
        pkI = (pkI << 4) | (unsigned)(on ? I : 15);
        pkJ = (pkJ << 4) | (unsigned)(on ? J : 15);
        ls = max(ls, on ? (J > I ? I : I - 2) : -1);
        const int Ic = on ? I : 0, Jc = on ? J : 0;
        const double* p = S + LY.blk(Ic, Jc);
        const bool diag = Ic == Jc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = p[diag ? dofs[r] : 64 * r + ln];
      }
      lastStep = ls;
    }
    __syncthreads();   // every block is in registers: the matrix area is free
    LDL_EV(0xB00);
    // the copies wave 0 needs first: block (0, 1) and the diagonal block 1
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int sI_s = (pkI >> (4 * s)) & 15, sJ_s = (pkJ >> (4 * s)) & 15;
      if (sI_s == 0 && sJ_s == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) qbuf[64 * r + lane] = acc[s][r];
        ldl_signal(&f_hq[0], 1, lane);
      }
      if (sI_s == 1 && sJ_s == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dpart[256 + 64 * r + lane] = acc[s][r];
        ldl_signal(&f_hp[1], 2, lane);
      }
    }
    int lastHelp = -1;   // the steps kb = ridx (mod NREG): this wave prepares XB[kb] for the back-substitution
    for (int k = ridx; k < nb; k += NREG) lastHelp = k;
    bool tdn_final = false;
    for (int kb = 0; kb < nb; ++kb) {
      const int lane = LDL_LANE();
      const bool helper = (kb % NREG == ridx);
      if (kb > lastStep && !tdn_final) {   // no panel or trailing work any more: the wave never reads a panel row again
        ldl_signal(&f_tdn[wave], INT_MAX, lane);
        tdn_final = true;
      }
      if (kb > lastStep && !helper) {
        if (kb > lastHelp) break;
        continue;
      }
      LDL_EV(0xA00 | kb);   // at the wait for X_kb
      ldl_wait_ge(&f_xready, kb + 1);
      LDL_EV(0x100 | kb);   // X_kb seen
      LDL_OTS(0);
      double a[4], dq[4];
      ldl_operand(Xs + kb * LDL_XB, dinv + kb * 16, lane, a, dq);
      LDL_OTS(1);
      if (kb <= lastStep) {
        double* Rk = Rp + (kb & 1) * nb * 256;
        // Which of the wave's slots have work in this step: one ballot each over the lanes that worked the slots out (lane s =
        // slot s); a slot without work costs a bit test and a branch.  (As a chain of `if (I == kb && J > kb)` over the slots the
        // bookkeeping was ~25 scalar instructions per slot and loop — coordinates read back from spilled registers, compares,
        // selects — some 300 per step and wave whatever the wave had to do; a wave issues one instruction every 4-8 cycles.
        // A loop over the set bits with a jump to the slot's code was slower still: the compiler moves what the six bodies
        // compute from the step number in front of the loop, for all six.)
        const unsigned pmask = (unsigned)__ballot(u_lane < SLOTS && uI == kb && uJ > kb);
        const unsigned tmask = (unsigned)__ballot(u_lane < SLOTS && kb < u_tl);
        unsigned cI = pkI, cJ = pkJ;
        asm volatile("" : "+s"(cI), "+s"(cJ));   // (what is derived from the slot coordinates is recomputed, not hoisted and spilled)
#define LDL_DISPATCH(mask, F)                                                             \
  {                                                                                       \
    if ((mask) & 1u) F(std::integral_constant<int, 0>{});                                 \
    if ((mask) & 2u) F(std::integral_constant<int, 1>{});                                 \
    if ((mask) & 4u) F(std::integral_constant<int, 2>{});                                 \
    if ((mask) & 8u) F(std::integral_constant<int, 3>{});                                 \
    if ((mask) & 16u) F(std::integral_constant<int, 4>{});                                \
    if ((mask) & 32u) F(std::integral_constant<int, 5>{});                                \
  }
        static_assert(SLOTS == 6, "the cases of LDL_DISPATCH");
        // ---- panel row: R_J = L_kk^-1 A_kJ for the owned blocks (kb, J > kb)
        bool waited = kb < 2;
        LDL_OTS(2);
        auto panel_slot = [&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const int pJ = (cJ >> (4 * s)) & 15;
          LDL_OTS(3);
          // (the SIMD issues oldest-wave-first: without a priority the youngest of its four waves publishes last, whatever the
          // chain is waiting for)
          if (pJ <= kb + 2) __builtin_amdgcn_s_setprio(3);
          else __builtin_amdgcn_s_setprio(1);
          // a block that is still exactly zero (no coupling, no fill so far) stays zero: no product, nothing written
          const bool nz = __any(acc[s][0] != 0.0 || acc[s][1] != 0.0 || acc[s][2] != 0.0 || acc[s][3] != 0.0) != 0;
          if (nz) {
            ldl_v4 R{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], acc[s][q], R, 0, 0, 0);
            acc[s] = R;
          }
          LDL_OTS(4);
          if (!waited) {   // the buffer (and its flags) still belong to the panel row of step kb - 2: every wave must be through with it
            for (;;) {
              const int v = (lane < NW && (lane & 3) != 0) ? __hip_atomic_load(&f_tdn[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : INT_MAX;
              if (__all(v >= kb - 1)) break;
              __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            waited = true;
          }
          LDL_OTS(5);
          if (nz) {
            double* rp = Rk + pJ * 256 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) rp[64 * r] = acc[s][r];
          }
          ldl_signal(&f_prdy[kb & 1][pJ], ((kb + 1) << 1) | (nz ? 0 : 1), lane);
          LDL_EV(0x200 | (kb << 4) | pJ);   // panel block (kb, J) published
          __builtin_amdgcn_s_setprio(0);
          LDL_OTS(6);
        };
        LDL_DISPATCH(pmask, panel_slot)
        // ---- trailing update of the owned blocks (I > kb).  Two of them are handed to wave 0 afterwards: (kb+1, kb+2), complete
        // with this update, and the diagonal block kb + 2.  The diagonal block kb + 1 is completed by wave 0 itself.
        // (a wave's slots are in row-major order, which is the order in which the chain needs them)
        // (Tried: the panel blocks of slot s + 1 requested before the products of slot s are issued, with a look at the flags that
        // does not wait.  Slower — the hand-overs arrive 2000 to 6000 cycles later: the extra looks at flags that are not up yet
        // cost more instructions than the hidden round trips give.)
        unsigned long long ready = 0, zeros = 0;
        auto trail_slot = [&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const int I = (cI >> (4 * s)) & 15, J = (cJ >> (4 * s)) & 15;
          const bool hq = (I == kb + 1 && J == kb + 2), hp = (I == kb + 2 && J == kb + 2);
          if (hq) LDL_OTS(7);
          if (hp) LDL_OTS(11);
          if (hq || hp) __builtin_amdgcn_s_setprio(3);
          const unsigned long long nd = (1ull << I) | (1ull << J);
          while ((ready & nd) != nd) {   // the panel blocks R_I and R_J of this step (one poll fetches the state of the whole panel row)
            ready = ldl_ready_mask_bit(f_prdy[kb & 1], nb, (kb + 1) << 1, lane, &zeros);
            if ((ready & nd) != nd) __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (hq) LDL_OTS(8);
          if (hp) LDL_OTS(12);
          if (!(zeros & nd)) {   // (a zero panel block: R_I^T D^-1 R_J is zero)
            const double* rp = Rk + I * 256 + lane;
            const double* rn = Rk + J * 256 + lane;
            double ra[4], rb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ra[q] = rp[64 * q];
              rb[q] = rn[64 * q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[q], rb[q] * dq[q], acc[s], 0, 0, 0);
          }
          if (hq) {
            LDL_OTS(9);
#pragma unroll
            for (int r = 0; r < 4; ++r) qbuf[(I & 1) * 256 + 64 * r + lane] = acc[s][r];
            ldl_signal(&f_hq[I & 1], I + 1, lane);
            LDL_EV(0x400 | I);   // Q_I handed over
            LDL_OTS(10);
          }
          if (hp) {
            LDL_OTS(13);
#pragma unroll
            for (int r = 0; r < 4; ++r) dpart[(I & 1) * 256 + 64 * r + lane] = acc[s][r];
            ldl_signal(&f_hp[I & 1], I + 1, lane);
            LDL_EV(0x500 | I);   // P_I handed over
            LDL_OTS(14);
          }
          if (hq || hp) __builtin_amdgcn_s_setprio(0);
        };
        LDL_DISPATCH(tmask, trail_slot)
#undef LDL_DISPATCH
        ldl_signal(&f_tdn[wave], kb + 1, lane);
        LDL_EV(0x300 | kb);   // trailing updates of step kb done
      }
      if (helper) {
        // XB[kb] = D^-1 Bh D^-1 with exact zeros above the diagonal: what the back-substitution multiplies t_K with.
        // Entry (i, j): lane = 16 (i >> 2) + j handles i = 4 (lane >> 4) .. + 3 of column j.
        const int j = lane & 15;
        const double dj = dinv[kb * 16 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * (lane >> 4) + e;
          const double v = Xs[kb * LDL_XB + j * LDL_RS + i] * dinv[kb * 16 + i] * dj;
          XB[kb * LDL_XB + j * LDL_RS + i] = i > j ? -v : (i == j ? dj : 0.0);   // (Xs holds -Bh below the diagonal)
        }
        ldl_signal(&f_xb[kb], 1, lane);
      }
    }
#ifdef LDL_OTS_WAVE
    if (stamps && wave == LDL_OTS_WAVE && lane == 0)
      for (int i = 0; i < 16; ++i) stamps[80 + i] = ots_[i];
#endif
    // ---- back-substitution: R_KJ x_J of the owned blocks with K <= J - 2, block column by block column, each into its own
    // slot (wave 0 adds the slots of a block row in a fixed order, so the result does not depend on who finishes first)
    for (int J = nb - 1; J >= 2; --J) {
      const int lane = LDL_LANE();
      bool any = false;
      unsigned cI = pkI, cJ = pkJ;
      asm volatile("" : "+s"(cI), "+s"(cJ));
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) any = any || ((int)((cJ >> (4 * s)) & 15) == J && (int)((cI >> (4 * s)) & 15) + 2 <= J);
      if (!any) continue;
      ldl_wait_ge(&f_x, nb - J);
      const double xj = xv[J * 16 + (lane & 15)];
#pragma unroll
      for (int s = SLOTS - 1; s >= 0; --s) {   // (the block row closest to J first: wave 0 asks for it first)
        const int K = (cI >> (4 * s)) & 15, sj = (cJ >> (4 * s)) & 15;
        if (sj == J && K + 2 <= J) {
          if (K + 2 == J) __builtin_amdgcn_s_setprio(3);   // the slot wave 0 asks for next
          double p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = row16_sum(acc[s][r] * xj);
          if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tcon[(K * nb + J) * 16 + (lane >> 4) + 4 * r] = p[r];
          }
          asm volatile("" ::: "memory");   // (in-order LDS: see ldl_signal)
          if (lane == 0) __hip_atomic_fetch_add(&f_tc[K], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          LDL_EV(0xF00 | (K << 4) | J);   // slot (K, J) written
          __builtin_amdgcn_s_setprio(0);
        }
      }
    }
  } else {
    // =============================================================================== wave 0: the diagonal chain
    // Lanes 0..15 carry the chain (column j = lane); the other three DPP rows execute along on data that is never stored.
    double c[16];
    const int j = LDL_LANE() & 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      c[i] = S[(lo >> 2) * 64 + (lo & 3) * 16 + hi];   // block (0, 0) sits at offset 0
    }
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
    asm volatile("" : "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]));
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    LDL_STAMP(1);
    bool bad = false;
    double mine = 0.0;
    // The hand-overs of step kb — block (kb, kb+1) with all its updates (qbuf) and the diagonal block kb + 1 with the updates
    // <= kb - 1 (dpart) — normally arrive while the block kb is being eliminated.  They are requested BEFORE the elimination,
    // flags first (the LDS serves a wave's requests in order and the producers write data, wait, then flag: data requested behind
    // a flag that reads as set is the data the flag announces), and looked at behind it: the poll, the acquire and the data round
    // trip leave the chain.  Only when a flag was not set yet are they requested again, as before.
    // (inline assembly: the compiler would wait for the loads right where they are issued, in front of the elimination)
    int hq_f = 0, hp_f = 0;
    double q0, q1, q2, q3, p0, p1, p2, p3;
    auto request_handover = [&](int kb, int lane) {
      const unsigned fqa = (unsigned)(size_t)&f_hq[kb & 1], fpa = (unsigned)(size_t)&f_hp[(kb + 1) & 1];
      const unsigned qa = (unsigned)(size_t)(qbuf + (kb & 1) * 256 + lane), pa = (unsigned)(size_t)(dpart + ((kb + 1) & 1) * 256 + lane);
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(hq_f), "=&v"(hp_f) : "v"(fqa), "v"(fpa) : "memory");
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(qa) : "memory");
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(pa) : "memory");
    };
    auto complete_handover = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hq_f), "+v"(hp_f), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : : "memory");
    };
    for (int kb = 0; kb < nb; ++kb) {
      const int lane = LDL_LANE();
      const int j = lane & 15;
      const int npiv = (kb == nb - 1) ? npl : 16;
      mine = 0.0;
      LDL_TSA(0);
      LDL_TS(0);
      LDL_TS(7);
      {
        auto hook = [&]() { request_handover(kb, lane); };
#ifdef LDL_COMP_ALL
        const bool careful = true;
#else
        const bool careful = (comp_mask >> kb) & 1u;   // (uniform: the mask comes from the window record)
#endif
        if (kb < nb - 1) {
          if (careful) ldl16_eliminate<true, false, decltype(hook), true>(c, 16, mine, j, nullptr, hook);
          else ldl16_eliminate<true, false, decltype(hook), false>(c, 16, mine, j, nullptr, hook);
        } else {
          if (careful) ldl16_eliminate<false, false, LdlNoHook, true>(c, npiv, mine, j);
          else ldl16_eliminate<false>(c, npiv, mine, j);
        }
      }
      bad = bad || (j < npiv && !(mine > 0.0 && mine < 1.0e300));   // a pivot was not positive
      LDL_TS(1);
      // publish Bh (column `lane` as it is: the consumers mask the dead entries above the diagonal) and 1/d
      if (lane < 16) {
        ldl_v4* xo = reinterpret_cast<ldl_v4*>(Xs + kb * LDL_XB + lane * LDL_RS);
        typedef double ldl_v2 __attribute__((ext_vector_type(2)));
        ldl_v2* xo2 = reinterpret_cast<ldl_v2*>(xo);
#pragma unroll
        for (int i = 0; i < 8; ++i) xo2[i] = ldl_v2{c[2 * i], c[2 * i + 1]};
        dinv[kb * 16 + lane] = mine;
        if (kb + 1 >= nb && lane == npl) {   // the rhs column after the elimination: L^-1 b of the last block in its rows < npl
#pragma unroll
          for (int i = 0; i < 16; ++i) xv[kb * 16 + i] = c[i];
        }
      }
      // this wave's own operand reads (L_kk^-1 in the A-operand layout) leave right behind the writes they read back — the LDS
      // serves a wave's requests in order — so their round trip runs under the release of the flag instead of behind it
      double a[4], dq[4];
      if (kb + 1 < nb) ldl_operand(Xs + kb * LDL_XB, dinv + kb * 16, lane, a, dq);
      ldl_signal(&f_xready, kb + 1, lane);
      LDL_TS(2);
      if (kb + 1 >= nb) break;
      // ---- R = L_kk^-1 A_(k,k+1) privately, then the last update of the next diagonal block
      LDL_EV(0x600 | kb);   // wave 0: X_kb published, waiting for Q_kb and P_(kb+1)
      complete_handover();
      LDL_TSA(1);
      while (!(__builtin_amdgcn_readfirstlane(hq_f) >= kb + 1 && __builtin_amdgcn_readfirstlane(hp_f) >= kb + 2)) {
        __builtin_amdgcn_s_sleep(1);
        request_handover(kb, lane);
        complete_handover();
      }
      LDL_EV(0x700 | kb);   // both there
      LDL_TSA(2);
      LDL_TS(3);
      ldl_v4 R{0, 0, 0, 0}, P{p0, p1, p2, p3};
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], q0, R, 0, 0, 0);
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], q1, R, 0, 0, 0);
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], q2, R, 0, 0, 0);
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], q3, R, 0, 0, 0);
      LDL_TS(4);
#pragma unroll
      for (int q = 0; q < 4; ++q) P = __builtin_amdgcn_mfma_f64_16x16x4f64(R[q], R[q] * dq[q], P, 0, 0, 0);
      LDL_TS(5);
#pragma unroll
      for (int r = 0; r < 4; ++r) Rsup[kb * LDL_XB + ((lane >> 4) + 4 * r) * LDL_RS + j] = R[r];   // for the back-substitution
      // accumulator layout -> column per lane through LDS: P[r] of DPP row p is entry (p + 4 r, j) = (j, p + 4 r) (the block is
      // symmetric); column j lands contiguous at j * LDL_RS and lane j reads it back as eight 16-byte words (one round trip;
      // the v_permlane16/32_swap network it replaces took 24 dependent cross-lane moves)
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[j * LDL_RS + (lane >> 4) + 4 * r] = P[r];
      if (lane < 16) {
        typedef double ldl_v2 __attribute__((ext_vector_type(2)));
        const ldl_v2* pi = reinterpret_cast<const ldl_v2*>(Pt + lane * LDL_RS);
        ldl_v2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = pi[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          c[2 * i] = t[i][0];
          c[2 * i + 1] = t[i][1];
        }
      }
      LDL_TS(6);
    }
    if (__any(bad && lane < 16) && lane == 0) *s_fail = 1;
    LDL_STAMP(2);
#ifdef LDL_TS_STEP
    if (stamps && tid == 0)
      for (int i = 0; i < 8; ++i) stamps[16 + i] = ts_[i];
#endif
#ifdef LDL_TS_ALL
    if (stamps && tid == 0)
      for (int i = 0; i < 48; ++i) stamps[32 + i] = ts_all[i / 16][i % 16];
#endif
    // ---- back-substitution, block by block from the last one.  Lane i holds entry i of the 16-vectors.
    //   t_K = -(slots of the block columns >= K+2, by the other waves) - R_(K,K+1) x_(K+1) ;  x_K = L_KK^-T D_K^-1 t_K = (D^-1 Bh D^-1)^T t_K
    double x = 0.0, tsum = 0.0;
    double rrow[16], xcol[16];   // row j of R_(K,K+1) and column j of XB[K]: fetched one step ahead (they are final long before)
    ldl_wait_ge(&f_xb[nb - 1], 1);
    {
      const int j = LDL_LANE() & 15;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        xcol[n] = XB[(nb - 1) * LDL_XB + j * LDL_RS + n];
        rrow[n] = 0.0;
      }
    }
    int fxb_next = nb >= 2 ? __hip_atomic_load(&f_xb[nb - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 1;   // (looked at one step ahead)
    for (int K = nb - 1; K >= 0; --K) {
      const int lane = LDL_LANE();
      const int j = lane & 15;
      // the chain: t_K = tsum_K - R_(K,K+1) x_(K+1), x_K = XB_K^T t_K, publish
      double t;
      if (K == nb - 1) {
        t = j < npl ? xv[K * 16 + j] : 0.0;   // (parked there by lane npl)
      } else {
        double s0 = tsum, s1 = 0.0;
        const double xn = -x;
        fmac_bcast_nop<0>(s0, xn, rrow[0]);
        fmac_bcast<1>(s1, xn, rrow[1]);
#define LDL_MV(n) fmac_bcast<n>(s0, xn, rrow[n]); fmac_bcast<n + 1>(s1, xn, rrow[n + 1]);
        LDL_MV(2) LDL_MV(4) LDL_MV(6) LDL_MV(8) LDL_MV(10) LDL_MV(12) LDL_MV(14)
#undef LDL_MV
        t = s0 + s1;
      }
      LDL_EV(0xE00 | K);   // t_K complete
      // (the operands of the next step are requested as soon as their registers are free: R_(K-1,K)'s rows here, under the second
      // product — wave 0 wrote them itself during the factorisation — and XB[K-1] with the sum of the slots behind it)
      if (K > 0) {
#pragma unroll
        for (int n = 0; n < 16; ++n) rrow[n] = Rsup[(K - 1) * LDL_XB + j * LDL_RS + n];
      }
      double s0 = 0.0, s1 = 0.0;
      fmac_bcast_nop<0>(s0, t, xcol[0]);
      fmac_bcast<1>(s1, t, xcol[1]);
#define LDL_MV(n) fmac_bcast<n>(s0, t, xcol[n]); fmac_bcast<n + 1>(s1, t, xcol[n + 1]);
      LDL_MV(2) LDL_MV(4) LDL_MV(6) LDL_MV(8) LDL_MV(10) LDL_MV(12) LDL_MV(14)
#undef LDL_MV
      x = s0 + s1;
      if (K == nb - 1) x = j == npl ? -1.0 : (j < npl ? x : 0.0);
      if (lane < 16) {
        xv[K * 16 + lane] = x;
        if (K * 16 + lane < D) x_out[LY.unperm(K * 16 + lane)] = x;
      }
      ldl_signal(&f_x, nb - K, lane);
      if (K > 0) {
        // what wave 4 has summed of the next step's slots (block columns >= K + 1: none of them needs x_K): the flag of the sum
        // in front of the sum (a sum requested behind a flag that reads as set is the sum the flag announces)
        if (fxb_next < 1) ldl_wait_ge(&f_xb[K - 1], 1);
        int ts_f = 1;
        double ts_v = 0.0;
        const bool summed = K + 1 < nb;
        const unsigned fa = (unsigned)(size_t)&f_ts[K - 1], va = (unsigned)(size_t)(tsv + (K - 1) * 16 + j);
        if (summed) asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3" : "=&v"(ts_f), "=&v"(ts_v) : "v"(fa), "v"(va) : "memory");
#pragma unroll
        for (int n = 0; n < 16; ++n) xcol[n] = XB[(K - 1) * LDL_XB + j * LDL_RS + n];
        fxb_next = K >= 2 ? __hip_atomic_load(&f_xb[K - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 1;
        if (summed) {
          LDL_EV(0xC00 | (K - 1));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ts_f), "+v"(ts_v) : : "memory");
          while (__builtin_amdgcn_readfirstlane(ts_f) < 1) {
            __builtin_amdgcn_s_sleep(1);
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ts_f), "=&v"(ts_v) : "v"(fa), "v"(va) : "memory");
          }
          LDL_EV(0xD00 | (K - 1));
        }
        tsum = ts_v;
      }
    }
    __builtin_amdgcn_s_setprio(0);
    LDL_STAMP(3);
  }
  __syncthreads();
#ifdef LDL_TRACE
  if (stamps && lane == 0) {
    stamps[128 + wave * 65] = tr_n;
    for (int i = 0; i < tr_n; ++i) stamps[128 + wave * 65 + 1 + i] = tr_log[wave][i];
  }
#endif
#undef LDL_STAMP
#undef LDL_LANE
#undef LDL_TS
#undef LDL_TSA
#undef LDL_OTS
#undef LDL_EV
}

}  // namespace ba
