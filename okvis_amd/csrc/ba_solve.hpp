// Kernel 3 — trust-region control + reduced-camera solve, one workgroup per window (fp64, LDS resident).
//
//   1. accept / reject the pending trial (Ceres LevenbergMarquardtStrategy semantics, ba_device.hpp)
//   2. assemble the reduced system of the accepted linearisation in LDS:
//        sum of the Schur partials (fixed chunk order) + per-group U_pp/g_p partials (host-built lists)
//        + precomputed IMU Hessian blocks (colour phases) + prior / marginalisation-prior blocks
//   3. convergence tests of the step just accepted (gradient, function tolerance)
//   4. LM damping  S += lambda * clamp(diag U),  blocked right-looking Cholesky
//   5. trial poses / speed-biases  x (+) delta  (PoseLocalParameterization::plus), model-decrease scalars
//
// LDS layout of S: lower triangle in 6x6 blocks, block (bi >= bj) at ((bi(bi+1)/2 + bj) * 36), row-major
// inside; a work-item's operand block is 288 contiguous bytes.  Cholesky step kb: (P) every panel row
// re-factors the 6x6 diagonal block in registers (no barrier between "factor" and "panel"), solves its
// row and the right-hand side rides along as one more row (forward substitution for free); (T) trailing
// update with one 6x6 register block per work-item.  Two barriers per block column.
//
// This is the only kernel that writes the window's Ctrl record.
#pragma once
#include <climits>
#include "ba_chol_tiles.hpp"
#include "ba_device.hpp"
#include "ba_ldl16.hpp"
#include "ba_chain.hpp"

namespace ba {

// LDS layout of the reduced matrix: lower triangle in 6x6 blocks, stored block-COLUMN by block-column
// (the panel of a Cholesky step is contiguous) with a block stride of 38 doubles: 16 consecutive blocks
// start on 16 distinct 4-bank groups, so a wave's ds_read_b128 of "my block" is conflict-free.
constexpr int SBS = 38;
struct SLayout {
  int nbk;
  __device__ __forceinline__ int blk(int bi, int bj) const {  // bi >= bj
    return (bj * nbk - (bj * (bj - 1)) / 2 + (bi - bj)) * SBS;
  }
  __device__ __forceinline__ int at(int i, int j) const {  // scalar index, i >= j
    const int bi = i / 6, bj = j / 6;
    return blk(bi, bj) + (i - 6 * bi) * 6 + (j - 6 * bj);
  }
};

// the layout of the reduced matrix by instantiation: HBM-resident (LARGE) = SLayout, LDS-resident = the 16x16 accumulator
// blocks of the MFMA LDL^T solver (ba_ldl16.hpp).  Both offer at(i, j) for i >= j.
template <bool LARGE> struct SolveLayout { typedef SLayout type; };
template <> struct SolveLayout<false> { typedef L16 type; };
// where entry d of the host-built IMU destination table lands: x carries the SLayout offset, z the reduced indices (i << 16 | j)
__device__ __forceinline__ int imu_dst_off(const SLayout&, const int4& d) { return d.x & 0xFFFFF; }
__device__ __forceinline__ int imu_dst_off(const L16& LY, const int4& d) { return LY.at(d.z >> 16, d.z & 0xFFFF); }
__device__ __forceinline__ int imu_dst_off(const LChain& LY, const int4& d) { return LY.at(d.z >> 16, d.z & 0xFFFF); }

// accumulate J^T J (lower triangle, reduced coordinates) and J^T r of one small factor.
// J: nres x ncol row-major (ncol = sum of dims), col_off[c] = reduced index of local column c or -1.
template <class LYT>
__device__ __forceinline__ void add_small_factor(double* S, const LYT LY, double* g, double* d2, const double* J,
                                                 const double* r, int nres, int ncol, const int* col_off, int tid,
                                                 int nthreads) {
  for (int wi = tid; wi < ncol * ncol; wi += nthreads) {
    const int a = wi / ncol, b = wi - a * ncol;
    const int ra = col_off[a], rb = col_off[b];
    if (ra < 0 || rb < 0 || ra < rb) continue;
    double s = 0;
    for (int k = 0; k < nres; ++k) s += J[k * ncol + a] * J[k * ncol + b];
    S[LY.at(ra, rb)] += s;
    if (a == b) d2[ra] += s;
  }
  for (int a = tid; a < ncol; a += nthreads) {
    const int ra = col_off[a];
    if (ra < 0) continue;
    double s = 0;
    for (int k = 0; k < nres; ++k) s += J[k * ncol + a] * r[k];
    g[ra] += s;
  }
}

// in-register Cholesky of a 6x6 SPD block given its lower triangle L[i][j] (i >= j); returns false if
// a pivot is not positive.  inv[k] = 1 / L[k][k].
__device__ __forceinline__ bool chol6(double (&L)[6][6], double (&inv)[6]) {
  // right-looking: after each pivot the remaining lower triangle is updated at once (independent FMAs),
  // so the chain per pivot is rsqrt -> scale -> update, not a k-long dependent sum
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double d = L[k][k];
    ok = ok && (d > 0.0);
    const double dd = d > 0.0 ? d : 1.0;
    inv[k] = rsqrt_nr(dd);
    L[k][k] = dd * inv[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) L[i][k] *= inv[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
#pragma unroll
      for (int j = k + 1; j <= i; ++j) L[i][j] -= L[i][k] * L[j][k];
  }
  return ok;
}


// explicit inverse of the lower-triangular factor (row-major, entries above the diagonal untouched):
// six independent columns, so a later  x = L^-1 y  is six independent dot products instead of a 21-step chain
__device__ __forceinline__ void trinv6(const double (&L)[6][6], const double (&inv)[6], double (&X)[6][6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    X[j][j] = inv[j];
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double a = 0;
#pragma unroll
      for (int m = j; m < i; ++m) a += L[i][m] * X[m][j];
      X[i][j] = -a * inv[i];
    }
  }
}

// factor the 6x6 diagonal block (lower triangle at dblk) and publish L^-1 (row-major 6x6, lower triangle): the
// panel and the back-substitution only ever need the inverse
__device__ __forceinline__ bool factor_diag(const double* dblk, double* Xout) {
  double L[6][6], X[6][6], inv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i][j] = dblk[6 * i + j];
  const bool ok = chol6(L, inv);
  trinv6(L, inv, X);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) Xout[6 * i + j] = X[i][j];
  return ok;
}

constexpr int SOLVE_LDS_LIMIT_CHAIN = 144 * 1024;   // the same for the chain instantiations (15 KB of static arrays)
constexpr int SOLVE_LDS_LIMIT = 148 * 1024;   // dynamic LDS of the solve kernel (160 KB per workgroup minus its static arrays, about 11 KB;
                                              // the largest LDS-resident system, D = 174, takes 137.5 KB)
constexpr int PRI_STAGE = 2;  // pose / speed-bias priors whose records the solve kernel stages in LDS ahead of time
constexpr int SOLVE_HELPERS = 4;   // workgroups per window (blockIdx.y < SOLVE_HELPERS) that sum the Schur chunk partials for the solving one
constexpr int SOLVE_HELPED_MAX_WINDOWS = 8;   // launches of more windows sum inside the solving workgroup (a helper takes a whole CU)
constexpr int MARG_BLOCKS_MAX = MAX_MARG_DIM / 6;   // blocks of a marginalisation prior (staged in LDS by assemble_base)
constexpr int PRE_BLOCKS = 32;  // pose / speed-bias blocks whose accepted values the solve kernel stages in LDS ahead of time

// IMU Hessian blocks, priors and the marginalisation prior of linearisation buffer `acc`, accumulated into S
// (block layout LY), g and d2.  (The reprojection part U_pp / U_pe / g_p arrives inside the Schur partials.)
template <class LYT>
__device__ void assemble_base(const WinPtrs& W, int acc, const LYT LY, double* S, double* g, double* d2,
                              int* coloff, int tid, int nthreads, bool skip_imu,
                              const double* pri = nullptr, const int* pricol = nullptr, int n_pri = 0,
                              bool imu_matrix_elsewhere = false, bool skip_priors = false) {
  // pri / pricol: LDS copies of the first n_pri (<= PRI_STAGE) pose priors [f][42], speed/bias priors r [f][9] and
  // sqrtInfo [f][81] of buffer `acc` and their reduced column offsets [f][6] | [f][9], staged by the caller
  // ---- IMU factors: precomputed H (30x30 lower) | g (30); factors of one colour touch disjoint blocks ----
  // Destination (host-built imu_asm) and value of up to NE entries per work-item are requested together, then applied colour
  // by colour (the destinations of one colour are disjoint, so nothing orders them and their read-modify-writes are batched).
  // `imu_matrix_elsewhere` (large windows, matrix in HBM): only the gradient and the diagonal (for the damping) are taken
  // here; the matrix entries are gathered by large_export_kernel through imu_rev, by many workgroups instead of this one.
  static_assert(IMU_LIN_STRIDE == 512, "imu_asm and the factor records share the index");
  {
    const int items = skip_imu ? 0 : W.n_imu * 512;
    constexpr int NE = 6;
    const double* src = W.imu_lin[acc];
    for (int col = 0; col < (skip_imu ? 0 : W.n_imu_color); ++col) {   // colour by colour: ALL of one colour before the next
      if (imu_matrix_elsewhere) {
        // only the 30 diagonal entries (for the damping diagonal) and the 30 gradient entries of every record
        for (int it = tid; it < W.n_imu * 60; it += nthreads) {
          const int f = it / 60, k = it - 60 * f;
          const int e = k < 30 ? k * (k + 3) / 2 : 465 + (k - 30);   // diagonal entry (k, k) of the packed lower triangle | g_k
          const int4 d = W.imu_asm[512 * f + e];
          if (d.x < 0 || (d.x >> 24) != col) continue;
          const double v = src[512 * f + e];
          if (d.x & (1 << 20)) g[d.x & 0xFFFFF] += v;
          else if (d.y >= 0) d2[d.y] += v;
        }
        __syncthreads();
        continue;
      }
      for (int base = tid; base < items; base += NE * nthreads) {
        double v[NE], old[NE];
        int dst[NE], d2i[NE], soff[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
          const int idx = base + u * nthreads;
          dst[u] = -1;
          d2i[u] = -1;
          soff[u] = 0;
          v[u] = 0;
          if (idx < items) {
            const int4 d = W.imu_asm[idx];
            if (d.x >= 0 && (d.x >> 24) == col) {
              dst[u] = d.x;
              d2i[u] = d.y;
              soff[u] = (d.x & (1 << 20)) ? (d.x & 0xFFFFF) : imu_dst_off(LY, d);
            }
            v[u] = (idx & 511) < 495 ? src[idx] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) old[u] = (dst[u] >= 0 && !(dst[u] & (1 << 20))) ? S[soff[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
          const int d = dst[u];
          if (d < 0) continue;
          const int off = soff[u];
          if (d & (1 << 20)) {
            g[off] += v[u];
          } else {
            S[off] = old[u] + v[u];
            if (d2i[u] >= 0) d2[d2i[u]] += v[u];
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- pose priors: J 6x6 | r 6 ----  (skip_priors: the caller has added the pose and speed/bias priors itself)
  for (int f = 0; f < (skip_priors ? 0 : W.n_pprior); ++f) {
    if (f < n_pri) {
      const double* L = pri + 42 * f;
      add_small_factor(S, LY, g, d2, L, L + 36, 6, 6, pricol + 6 * f, tid, nthreads);
      __syncthreads();
      continue;
    }
    if (tid < 6) {
      const int off = W.pose_off[W.pprior_pose[f]];
      coloff[tid] = off < 0 ? -1 : off + tid;
    }
    __syncthreads();
    const double* L = W.pp_lin[acc] + (size_t)f * 42;
    add_small_factor(S, LY, g, d2, L, L + 36, 6, 6, coloff, tid, nthreads);
    __syncthreads();
  }
  // ---- speed/bias priors: J = -sqrtInfo (9x9 const) | r 9 ----
  for (int f = 0; f < (skip_priors ? 0 : W.n_sbprior); ++f) {
    const bool st = f < n_pri;
    if (!st) {
      if (tid < 9) {
        const int off = W.sb_off[W.sbprior_sb[f]];
        coloff[tid] = off < 0 ? -1 : off + tid;
      }
      __syncthreads();
    }
    // J^T J and J^T r are sign-invariant / sign-flipped: use +sqrtInfo with -r
    const double* Jc = st ? pri + PRI_STAGE * 42 + PRI_STAGE * 9 + 81 * f : W.sbprior_sqrtinfo + (size_t)f * 81;
    const double* r = st ? pri + PRI_STAGE * 42 + 9 * f : W.sbp_lin[acc] + (size_t)f * 9;
    const int* co = st ? pricol + PRI_STAGE * 6 + 9 * f : coloff;
    for (int wi = tid; wi < 81; wi += nthreads) {
      const int a = wi / 9, b = wi - 9 * a;
      const int ra = co[a], rb = co[b];
      if (ra < 0 || rb < 0 || ra < rb) continue;
      double s = 0;
      for (int k = 0; k < 9; ++k) s += Jc[k * 9 + a] * Jc[k * 9 + b];
      S[LY.at(ra, rb)] += s;
      if (a == b) d2[ra] += s;
    }
    if (tid < 9 && co[tid] >= 0) {
      double s = 0;
      for (int k = 0; k < 9; ++k) s -= Jc[k * 9 + tid] * r[k];
      g[co[tid]] += s;
    }
    __syncthreads();
  }
  // ---- relative pose factors: [J0 6x6 | J1 6x6 | r 6] stored as J 6x12 row-major | r ----
  for (int f = 0; f < W.n_rel; ++f) {
    if (tid < 12) {
      const int off = W.pose_off[tid < 6 ? W.rel_pose0[f] : W.rel_pose1[f]];
      coloff[tid] = off < 0 ? -1 : off + (tid % 6);
    }
    __syncthreads();
    const double* L = W.rel_lin[acc] + (size_t)f * 78;
    add_small_factor(S, LY, g, d2, L, L + 72, 6, 12, coloff, tid, nthreads);
    __syncthreads();
  }
  // ---- marginalisation prior: H = B^T (J^T J) B, g = B^T J^T e ----
  // (B: identity except the 3x3 rotation blocks M of the pose blocks' orientation parts.)  The block table — offset in the prior,
  // type, reduced offset in the system — is staged in LDS first: looked up per entry in global memory it cost every entry four
  // dependent round trips (offsets -> type / index -> reduced offset -> H0), three passes of them for a 45-row prior.  Now an
  // entry's H0 / M operands are its only global loads, requested together.
  if (W.marg_dim > 0) {
    const int Dm = W.marg_dim, nb = W.marg_nb;
    __shared__ int s_moff[MARG_BLOCKS_MAX], s_mR[MARG_BLOCKS_MAX];
    __shared__ unsigned char s_mpose[MARG_BLOCKS_MAX];
    for (int b = tid; b < nb; b += nthreads) {
      const int type = W.marg_block_type[b], idx = W.marg_block_idx[b];
      s_moff[b] = W.marg_block_off[b];
      s_mpose[b] = type == 0;
      s_mR[b] = type == 0 ? W.pose_off[idx] : W.sb_off[idx];
    }
    __syncthreads();
    const double* M = W.marg_lin_M[acc];
    const double* JTe = W.marg_lin_e[acc] + Dm;  // [e | J^T e]
    const double* H0 = W.marg_H0;
    // (the operands of an entry are named scalars, not arrays: with 128 registers per lane the compiler keeps small private
    //  arrays in scratch memory)
    for (int wi = tid; wi < Dm * Dm; wi += nthreads) {
      const int rr = wi / Dm, cc = wi - rr * Dm;
      int bi = 0, bj = 0;
      for (int b = 0; b < nb; ++b) {
        if (s_moff[b] <= rr) bi = b;
        if (s_moff[b] <= cc) bj = b;
      }
      const int oi = s_moff[bi], oj = s_moff[bj];
      const int li = rr - oi, lj = cc - oj;
      const int Ri = s_mR[bi], Rj = s_mR[bj];
      if (Ri < 0 || Rj < 0 || Ri + li < Rj + lj) continue;
      const bool roti = s_mpose[bi] && li >= 3;
      const bool rotj = s_mpose[bj] && lj >= 3;
      double sacc = 0;
      if (!roti && !rotj) {
        sacc = H0[(size_t)rr * Dm + cc];
      } else {
        const int na = roti ? 3 : 1, nbk = rotj ? 3 : 1;
        // rows r2(a) = oi + 3 + a | rr, columns c2(b) = oj + 3 + b | cc; weights M(bi)[a][li - 3] | 1, M(bj)[b][lj - 3] | 1
#define MP_W(v, rot, blk, k, l) const double v = (rot) ? M[9 * (blk) + 3 * (k) + ((l) - 3)] : 1.0;
        MP_W(wa0, roti, bi, 0, li) MP_W(wa1, roti && 1 < na, bi, 1, li) MP_W(wa2, roti && 2 < na, bi, 2, li)
        MP_W(wb0, rotj, bj, 0, lj) MP_W(wb1, rotj && 1 < nbk, bj, 1, lj) MP_W(wb2, rotj && 2 < nbk, bj, 2, lj)
#undef MP_W
#define MP_H(v, a2, b2) const double v = ((a2) < na && (b2) < nbk) ? H0[(size_t)(roti ? oi + 3 + (a2) : rr) * Dm + (rotj ? oj + 3 + (b2) : cc)] : 0.0;
        MP_H(h00, 0, 0) MP_H(h01, 0, 1) MP_H(h02, 0, 2) MP_H(h10, 1, 0) MP_H(h11, 1, 1) MP_H(h12, 1, 2) MP_H(h20, 2, 0) MP_H(h21, 2, 1) MP_H(h22, 2, 2)
#undef MP_H
        // (the terms in the order of the loops over a and b, every product as wa * H0 * wb)
#define MP_T(a2, b2, w1, h, w2) if ((a2) < na && (b2) < nbk) sacc += (w1) * (h) * (w2);
        MP_T(0, 0, wa0, h00, wb0) MP_T(0, 1, wa0, h01, wb1) MP_T(0, 2, wa0, h02, wb2)
        MP_T(1, 0, wa1, h10, wb0) MP_T(1, 1, wa1, h11, wb1) MP_T(1, 2, wa1, h12, wb2)
        MP_T(2, 0, wa2, h20, wb0) MP_T(2, 1, wa2, h21, wb1) MP_T(2, 2, wa2, h22, wb2)
#undef MP_T
      }
      S[LY.at(Ri + li, Rj + lj)] += sacc;
      if (Ri + li == Rj + lj) d2[Ri + li] += sacc;
    }
    for (int rr = tid; rr < Dm; rr += nthreads) {
      int bi = 0;
      for (int b = 0; b < nb; ++b)
        if (s_moff[b] <= rr) bi = b;
      const int oi = s_moff[bi], li = rr - oi;
      const int Ri = s_mR[bi];
      if (Ri < 0) continue;
      double sacc;
      if (s_mpose[bi] && li >= 3) {
        sacc = 0;
        for (int a2 = 0; a2 < 3; ++a2) sacc += M[9 * bi + 3 * a2 + (li - 3)] * JTe[oi + 3 + a2];
      } else {
        sacc = JTe[rr];
      }
      g[Ri + li] += sacc;
    }
    __syncthreads();
  }

}

// trial poses / speed-biases  x (+) delta  of buffer 1-acc (PoseLocalParameterization::plus, PoseLocalParameterization.cpp:60-87).
// Adds |x|^2 over the free blocks to *x2 and, when `ambient`, |x - x(+)delta|^2 to *s2.
// pre (LDS, may be null): the accepted values of the first PRE_BLOCKS pose blocks [b][7] and speed/bias blocks
// [PRE_BLOCKS * 7 + 9 b], preoff their reduced offsets.  Every value is read before the first store is issued and the LDS /
// HBM sources are separate code paths (a pointer that may be either is a generic pointer: its loads are FLAT instructions
// whose wait also waits for the stores before them — nine serialised store round trips, 1.6 us, in the first version).
__device__ __forceinline__ void trial_states(const WinPtrs& W, int acc, const double* s_x, int tid, bool ambient, double* s2,
                                             double* x2, const double* pre = nullptr, const int* preoff = nullptr) {
  const int trial = 1 - acc;
  double a2 = 0, b2 = 0;
  for (int b = tid; b < W.n_pose; b += SOLVE_THREADS) {
    auto xt = as_global(W.pose[trial] + 7 * (size_t)b);
    double xin[7], xo[7];
    int off;
    if (pre && b < PRE_BLOCKS) {
      off = preoff[b];
#pragma unroll
      for (int k = 0; k < 7; ++k) xin[k] = pre[7 * b + k];
    } else {
      off = W.pose_off[b];
      auto xp = as_global(W.pose[acc] + 7 * (size_t)b);
#pragma unroll
      for (int k = 0; k < 7; ++k) xin[k] = xp[k];
    }
    if (off >= 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) b2 += xin[k] * xin[k];
      pose_oplus_dev(xin, s_x + off, xo);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        xt[k] = xo[k];
        if (ambient) a2 += (xo[k] - xin[k]) * (xo[k] - xin[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 7; ++k) xt[k] = xin[k];
    }
  }
  // (the speed/bias blocks on wave 2's lanes: the pose blocks keep wave 0's busy)
  for (int b = (tid + SOLVE_THREADS - 128) % SOLVE_THREADS; b < W.n_sb; b += SOLVE_THREADS) {
    auto xt = as_global(W.sb[trial] + 9 * (size_t)b);
    double v[9], d[9];
    int off;
    if (pre && b < PRE_BLOCKS) {
      off = preoff[PRE_BLOCKS + b];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = pre[PRE_BLOCKS * 7 + 9 * b + k];
    } else {
      off = W.sb_off[b];
      auto xp = as_global(W.sb[acc] + 9 * (size_t)b);
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = xp[k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = off >= 0 ? s_x[off + k] : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (off >= 0) b2 += v[k] * v[k];
      const double nv = v[k] + d[k];
      xt[k] = nv;
      if (ambient && off >= 0) a2 += (nv - v[k]) * (nv - v[k]);
    }
  }
  *s2 += a2;
  *x2 += b2;
}

// landmark part of the Cauchy-point scalars of the dogleg step (xv = Cauchy direction, pose part in xvp):
//   *Al += b_l . xv_l ,   *Bl += t^T (V_l + mu Dt2_l)^-1 t + 2 t . xv_l + xv_l^T V_l xv_l ,   t = W_l^T xv_p
__device__ __forceinline__ void dl_landmark_sums(const WinPtrs& W, int acc, const double* xvp, double mu, const OptD& opt,
                                                 int tid, double* Al_out, double* Bl_out) {
  double Al = 0, Bl = 0;
  const double* s_x = xvp;
  for (int l = tid; l < W.n_lm; l += SOLVE_THREADS) {
        const double* Vl = W.V[acc] + 6 * (size_t)l;
        const double* bl = W.bl[acc] + 3 * (size_t)l;
        const double* sl = W.lm_scale + 3 * (size_t)l;
        double v[6] = {Vl[0], Vl[1], Vl[2], Vl[3], Vl[4], Vl[5]};
        const double dt[3] = {damp_diag(v[0], sl[0], opt), damp_diag(v[3], sl[1], opt), damp_diag(v[5], sl[2], opt)};
        const double xl[3] = {bl[0] / dt[0], bl[1] / dt[1], bl[2] / dt[2]};
        double t[3] = {0, 0, 0};
        for (int pr = W.lm_pair_begin[l]; pr < W.lm_pair_begin[l + 1]; ++pr) {
          const double* Wp = W.W[acc] + 18 * (size_t)pr;
          const double* xp = s_x + W.pair_off[pr];
          for (int i = 0; i < 6; ++i) {
            t[0] += Wp[3 * i] * xp[i];
            t[1] += Wp[3 * i + 1] * xp[i];
            t[2] += Wp[3 * i + 2] * xp[i];
          }
        }
        const double quad3 = xl[0] * (v[0] * xl[0] + v[1] * xl[1] + v[2] * xl[2]) +
                             xl[1] * (v[1] * xl[0] + v[3] * xl[1] + v[4] * xl[2]) +
                             xl[2] * (v[2] * xl[0] + v[4] * xl[1] + v[5] * xl[2]);
        v[0] += mu * dt[0];
        v[3] += mu * dt[1];
        v[5] += mu * dt[2];
        double vi[6];
        inv3sym(v, vi);
        const double quad1 = t[0] * (vi[0] * t[0] + vi[1] * t[1] + vi[2] * t[2]) +
                             t[1] * (vi[1] * t[0] + vi[3] * t[1] + vi[4] * t[2]) +
                             t[2] * (vi[2] * t[0] + vi[4] * t[1] + vi[5] * t[2]);
        Al += bl[0] * xl[0] + bl[1] * xl[1] + bl[2] * xl[2];
        Bl += quad1 + 2.0 * (t[0] * xl[0] + t[1] * xl[1] + t[2] * xl[2]) + quad3;
      }
  *Al_out += Al;
  *Bl_out += Bl;
}

// DoglegStrategy::ComputeTraditionalDoglegStep in terms of the scalars of the current point:
//   A = |ghat|^2 = g.xv,  B = xv^T H xv,  C = g.dGN = ghat.gnhat,  E = |gnhat|^2  (hat = Jacobi-scaled, D-normalised).
// Step delta = -cA xv + beta dGN; *dln = its hat-norm (dogleg_step_norm_); *model = -g.delta - delta^T H delta / 2 using
// H dGN = -g - mu Dt2 dGN.
__device__ __noinline__ void dogleg_coefficients(double A, double B, double C, double E, double mu, double radius, double* cA,
                                                 double* beta, double* dln, double* model) {
#pragma clang fp contract(off)
  const double alpha = A / B;
  const double gradient_norm = sqrt(A), gn_norm = sqrt(E);
  double a, b, n;
  if (gn_norm <= radius) {  // case 1: the Gauss-Newton point lies inside the trust region
    a = 0.0, b = 1.0, n = gn_norm;
  } else if (gradient_norm * alpha >= radius) {  // case 2: even the Cauchy point lies outside
    a = radius / gradient_norm, b = 0.0, n = radius;
  } else {  // case 3: on the segment Cauchy point -> Gauss-Newton point
    const double b_dot_a = -alpha * C;
    const double a2 = (alpha * gradient_norm) * (alpha * gradient_norm);
    const double bma2 = a2 - 2.0 * b_dot_a + gn_norm * gn_norm;
    const double cc = b_dot_a - a2;
    const double dsc = sqrt(cc * cc + bma2 * (radius * radius - a2));
    b = (cc <= 0.0) ? (dsc - cc) / bma2 : (radius * radius - a2) / (dsc + cc);
    a = alpha * (1.0 - b);
    n = sqrt(a * a * A - 2.0 * a * b * C + b * b * E);
  }
  *cA = a;
  *beta = b;
  *dln = n;
  *model = a * A - b * C - 0.5 * (a * a * B + 2.0 * a * b * (A + mu * C) + b * b * (-C - mu * E));
}

// the Gauss-Newton factorisation failed (or mu has reached max_mu): DoglegStrategy raises mu; once it cannot be
// raised any more the step counts as an invalid iteration (TrustRegionMinimizer: StepIsInvalid, limited number)
__device__ __forceinline__ void solve_failed_dl(Ctrl* c, const OptD& opt) {
  c->explicit_next = 0;
  c->have_tot = 0;
  c->mu *= DL_MU_INCREASE;
  if (c->mu >= DL_MAX_MU) {
    c->iter++;
    c->invalid_steps++;
    if (c->invalid_steps >= opt.max_invalid) c->done = 5 + 1;
  }
}

// LARGE = false: the block matrix lives in LDS (D <= MAX_D_LDS).  LARGE = true: it lives in the window's HBM
// workspace (L2-resident; same algorithm, one workgroup) — the functional path for BASELINE configs[2].
// DBUF: the batch keeps one set of Schur partials per linearisation buffer (fused mode, WinPtrs::spart_buf_stride): the sums
// are taken from the buffer that is accepted if the pending trial is, and taken again when it was not.
// CHAIN (LDS-resident windows of a batch laid out for it, WinPtrs::chain): the system is assembled in the layout of ba_chain.hpp
// and solved by chain_solve — speed/bias blocks eliminated along the IMU chain, then the dense pose system.
template <bool LARGE, bool DBUF, bool CHAIN = false>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(const WinPtrs* __restrict__ wins,
                                                              const OptD* __restrict__ optp, int final_only, CtrlSlot* ctrls) {
  static_assert(!(LARGE && DBUF), "windows solved in HBM are never fused");
  static_assert(!(LARGE && CHAIN), "the chain solver is an LDS solver");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const long long t_start = clock64();   // (diagnostics: stamp 43)
  const WinPtrs& W = wins[blockIdx.x];
  // The control record: its address comes from the kernel arguments (CtrlSlot, ba_types.hpp; == W.ctrl), so its first words —
  // accepted buffer, pending, first, done — are requested together with the window record.  Every speculative load below
  // (Schur partials, IMU / prior records, states, the trial's scalar partials) takes its buffer index from them: they all
  // leave one memory round trip earlier than when the record's address had to be read from the window record first.
  Ctrl* gctrl = &ctrls[blockIdx.x].c;
  typedef int ctrl_v4 __attribute__((ext_vector_type(4)));
  const ctrl_v4 chead = *as_global(reinterpret_cast<const ctrl_v4*>(gctrl));   // acc, pending, first, done
  // The window record is 18 lines of the scalar cache, and the compiler fetches a field where it is first used: line after
  // line, each miss a memory round trip of its own in front of whatever needed the field.  One dword of every line is
  // requested here, next to the control words; the fields then come from the cache.  (Plain loads whose combination feeds a
  // branch that is never taken: any inline assembly in this kernel makes the compiler read the whole record with vector loads.)
  int warm = 0;
  {
    constexpr int lines = (int)((sizeof(WinPtrs) + 63) / 64);   // (never a line beyond the record: the next window's, or whatever follows the last)
    static_assert(sizeof(WinPtrs) % 4 == 0 && (lines - 1) * 64 < (int)sizeof(WinPtrs), "lines touched below");
    const int* wi = reinterpret_cast<const int*>(&W);
#pragma unroll
    for (int k = 0; k < lines; ++k) warm |= wi[16 * k];
  }
  // the buffer that is accepted if the pending trial is (uniform: a scalar register, so are the addresses derived from it)
  const int spec0 = __builtin_amdgcn_readfirstlane(chead.y ? 1 - chead.x : chead.x);
  if (final_only == 0x7ffffff1 && warm == 0x5a5a5a5a) return;   // (never: keeps the loads above)
  if (LARGE != (W.Sg != nullptr)) return;  // each window is handled by the instantiation that fits it
  if (!LARGE && CHAIN != (W.chain != 0)) return;
  const int tid = threadIdx.x;
  if (W.prof && tid == 0 && blockIdx.x == 0 && blockIdx.y + 1 == gridDim.y) {   // diagnostics: first instruction | control words + window record are there
    W.prof[43] = (double)t_start;
    W.prof[40] = (double)clock64();
  }
  if constexpr (!LARGE) {
    // ---- helper workgroups (blockIdx.y < gridDim.y - 1; dispatched before the solving workgroup of their window): the sum
    // of the Schur chunk partials, one item (double of the partials' record) per work-item, every chunk requested at once and
    // added in chunk order.  One CU pulling the partials alone could keep only so many loads in flight (11.5 of the solve
    // kernel's 52 us); four more CUs do it while the solving workgroup takes the trust-region decision.  Every helper arrives
    // exactly once per launch (also for a finished window), the solving workgroup counts the launches: the two stay in step.
    if (blockIdx.y + 1 < gridDim.y) {   // (the last workgroup in y is the solving one; launches of many windows have no helpers)
      const int nh = (int)gridDim.y - 1;
      const int ntot = W.spart_stride, per = (ntot + nh - 1) / nh;
      if (W.n_chunk > 0) {   // (also for a finished window: the sums are not used then, but nothing has to be read to find out)
        const size_t stride = (size_t)__builtin_amdgcn_readfirstlane(W.spart_stride);   // (uniform: the chunk offsets are scalar arithmetic)
        const int nch = __builtin_amdgcn_readfirstlane(W.n_chunk);
        auto sp0 = W.spart + (DBUF ? (size_t)spec0 * W.spart_buf_stride : (size_t)0);   // (the speculated buffer)
        for (int k = tid; k < per; k += SOLVE_THREADS) {
          const int i = blockIdx.y * per + k;
          if (i >= ntot) break;
          auto sp = sp0 + i;
          double a = 0;
          // (every chunk of a fused window - 34 for configs[1] - requested in ONE trip: the helpers' loads come from other
          // CUs' stores, a trip costs a full memory round trip and there is nothing else to do meanwhile)
          constexpr int HB = 36;
          for (int ch = 0; ch < nch; ch += HB) {
            double v[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) v[u] = (ch + u < nch) ? sp[(size_t)(ch + u) * stride] : 0.0;
#pragma unroll
            for (int u = 0; u < HB; ++u) a += v[u];
          }
          // device-coherent store (written through): with the plain store the hand-over needed an agent-scope release,
          // which writes this XCD's whole L2 back (3.5 us, profiles/r03_notes.md) before the counter could be raised
          __hip_atomic_store(&W.spart_sum[i], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been performed (every storing wave drains)
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(W.sum_sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  // ------------------------------------------------------------------ 0. every request that needs no decision, in one go
  // The prologue is straight-line code: every load below is issued before the first of them is waited for (one memory round
  // trip for all of them), indices are clamped instead of guarded (a lane without an element requests element 0 along), and
  // what goes to LDS is stored further down, in the shadow of wave 0's decision.  (As guarded load-store pairs — `if (t < n)
  // s_x[t] = W.x[t]` — every pair waited for its own load: eight memory round trips in a row before the IMU records and the
  // Schur sums were even requested.)
  const bool helped = !LARGE && gridDim.y > 1;   // small launches only: a helper occupies a whole CU (the kernel's LDS footprint)
  __shared__ int s_sum_ready;   // 1 = the helpers' sums are there, 2 = they are late: summed here
  const int t64 = tid - 64;                 // work-item index among the waves 1 .. 15
  const int tl = t64 < 0 ? 0 : t64;
  // helpers delivered so far must reach SOLVE_HELPERS x the number of solve launches of this window (work-item 64 alone keeps
  // this book and polls; the other waves follow its LDS flag)
  const int sum_launches = LARGE ? 0 : as_global(W.sum_sync)[1];
  // The window record (sizes and ~140 pointers, 1.1 KB in HBM) is copied to LDS once: every later W.field is an LDS read
  // instead of a scalar load that misses its cache line by line (measured: 3 us of the 4.6 us tail were such misses)
  __shared__ double s_Wd[(sizeof(WinPtrs) + 7) / 8];
  static_assert(sizeof(WinPtrs) % 8 == 0, "copied as doubles");
  constexpr int WD_N = (int)(sizeof(WinPtrs) / 8);
  const double wd_v = LARGE ? 0.0 : reinterpret_cast<const double*>(&W)[tl < WD_N ? tl : 0];
  if (LARGE && tid == 0) W.ct_flag[W.ct_nT * (W.ct_nT + 1) / 2 + 1] = 0;  // no system exported (yet) this launch
  if (chead.w) {   // done (the launch still counts: the helpers of this window have arrived, or will)
    if (helped && tid == 64) as_global(W.sum_sync)[1] = sum_launches + ((int)gridDim.y - 1);
    return;
  }
  // (a reference, not a copy: the fields are scalar loads from the option record where they are used.  As a local copy the
  //  record lived in scratch memory — its address is handed to the decision functions — and every opt.field was a scratch
  //  load whose wait also waited for every prefetch in flight)
  const OptD& opt = *optp;
  const int D = W.D, Dp = W.Dp;
  const int Dpad = ((D + 5) / 6) * 6, nbk = Dpad / 6;
  // LDS-resident: the matrix area of the LDL^T solver (upper 16x16 blocks incl. the rhs column D, or its work area)
  typedef typename std::conditional<CHAIN, LChain, typename SolveLayout<LARGE>::type>::type LYT;
  // (LDS-resident: the solver eliminates the speed/bias part first, L16::perm; every LY.at() below takes reduced coordinates)
  const LYT LY = [&]() {
    if constexpr (CHAIN) return LChain::make(D, Dp);
    else if constexpr (LARGE) return LYT{nbk};
    else return LYT{ldl16_nb(D), D - Dp, D};
  }();
  const int nS = [&]() {
    if constexpr (CHAIN) return LY.total;
    else return LARGE ? nbk * (nbk + 1) / 2 * SBS : ldl16_area_doubles(D);
  }();

  double* S = LARGE ? W.Sg : smem;            // block-packed lower triangle
  double* s_rhs = LARGE ? smem : smem + nS;   // Dpad: rhs, then y = L^-1 rhs in place
  double* s_g = s_rhs + Dpad; // gradient (pose / speed-bias part)
  double* s_d2 = s_g + Dpad;  // diag(U) -> clamp -> LM damping diagonal
  double* s_x = s_d2 + Dpad;  // solution
  __shared__ Ctrl c;
  __shared__ int s_accepted, s_was_first, s_fail, s_was_pending;
  __shared__ double s_cost_change, s_old_cost, s_lm_gmax;
  __shared__ int s_coloff[64];
  __shared__ double s_red[SOLVE_THREADS / 64];

#define STAMP(k) do { if (W.prof && tid == 0 && blockIdx.x == 0) W.prof[k] = (double)clock64(); } while (0)
  STAMP(0);
  const double dec_pref = LARGE ? 0.0 : W.dec[tid < DEC_COUNT ? tid : 0];   // the Schur kernel's decision record (wave 0 reads it below)
  // the Jacobi scale of this lane's column (used by the damping, section 4): requested now, the value is only wrong in the
  // launch that estimates it
  const double scale_pref = opt.dogleg ? W.scale_p[tid < D ? tid : 0] : 1.0;
  // Pose / speed-bias priors (the stock window has one of each): wave 0's business from the first request to the last sum.  Its
  // lanes request the records of the speculated buffer and the priors' columns in the reduced system (host-built) here, park
  // them in LDS behind the decision (its own writes, read back by itself: no other wave is involved), form the J^T J / J^T r
  // entries while the other waves still wait for the Schur sums, and add them behind the IMU records.
  __shared__ double s_pri[PRI_STAGE * (42 + 9 + 81)];
  __shared__ int s_pricol[PRI_STAGE * (6 + 9)];
  const int n_pri = (!LARGE && W.n_pprior <= PRI_STAGE && W.n_sbprior <= PRI_STAGE) ? PRI_STAGE : 0;
  double pri_a0 = 0, pri_a1 = 0, pri_b = 0, pri_c0 = 0, pri_c1 = 0, pri_c2 = 0;
  int pri_col = -1;
  if (n_pri) {   // (uniform)
    const int npp = W.n_pprior, nsp = W.n_sbprior;
    const int l = tid & 63;
    auto pa = W.pp_lin[spec0];
    auto pc = W.sbprior_sqrtinfo;
    pri_a0 = pa[l < npp * 42 ? l : 0];
    pri_a1 = pa[l + 64 < npp * 42 ? l + 64 : 0];
    pri_b = W.sbp_lin[spec0][l < nsp * 9 ? l : 0];
    pri_c0 = pc[l < nsp * 81 ? l : 0];
    pri_c1 = pc[l + 64 < nsp * 81 ? l + 64 : 0];
    pri_c2 = pc[l + 128 < nsp * 81 ? l + 128 : 0];
    pri_col = W.prior_col[l < npp * 6 + nsp * 9 ? l : 0];
  }
  // accepted pose / speed-bias values of the speculated buffer -> LDS (the convergence test and the trial states read them;
  // one memory round trip here instead of two on the critical path later)
  __shared__ double s_pre[PRE_BLOCKS * 16];
  __shared__ int s_preoff[2 * PRE_BLOCKS];   // reduced offsets of the first PRE_BLOCKS pose blocks | speed/bias blocks
  constexpr bool pre_on = !LARGE;
  // (what is loaded is only selected where it is stored: a select in front of the other requests makes the compiler wait for
  //  the load — and with it for every request issued so far — right here)
  double pre_a0 = 0, pre_b0 = 0, pre_a1 = 0, pre_b1 = 0;
  int pre_po = -1, pre_so = -1;
  if constexpr (pre_on) {
    // two doubles per work-item 64 .. 64 + 8 PRE_BLOCKS - 1: element i = 2 t + u of [pose values (7 per block) | speed/bias values (9)]
    const int np7 = 7 * W.n_pose, ns9 = 9 * W.n_sb;
    const int i0 = 2 * tl, i1 = 2 * tl + 1;
    const int kp0 = i0 < np7 ? i0 : 0, kp1 = i1 < np7 ? i1 : 0;
    const int ks0 = (i0 >= PRE_BLOCKS * 7 && i0 - PRE_BLOCKS * 7 < ns9) ? i0 - PRE_BLOCKS * 7 : 0;
    const int ks1 = (i1 >= PRE_BLOCKS * 7 && i1 - PRE_BLOCKS * 7 < ns9) ? i1 - PRE_BLOCKS * 7 : 0;
    auto pp = W.pose[spec0];
    auto sp = W.sb[spec0];
    pre_a0 = pp[kp0], pre_b0 = sp[ks0], pre_a1 = pp[kp1], pre_b1 = sp[ks1];
    // the reduced offsets: work-items 64 + 8 PRE_BLOCKS .. + 10 PRE_BLOCKS - 1
    const int to = tl - PRE_BLOCKS * 8;
    pre_po = W.pose_off[(to >= 0 && to < PRE_BLOCKS && to < W.n_pose) ? to : 0];
    pre_so = W.sb_off[(to >= PRE_BLOCKS && to < 2 * PRE_BLOCKS && to - PRE_BLOCKS < W.n_sb) ? to - PRE_BLOCKS : 0];
  }
  // IMU factor records (H | g, 495 doubles each): value and destination (host-built imu_fastw: one word per entry) of up to
  // IMU_NPF entries per lane are requested now, from the buffer that is accepted if the pending step is (the common case),
  // and scattered after the decision without any further global round trip.
  constexpr int IMU_NPF = 6, IMU_NL = SOLVE_THREADS - 64;
  const int imu_items = W.n_imu * 512;
  const bool imu_fast = !LARGE && imu_items <= IMU_NPF * IMU_NL && W.n_imu_color < 16;
  const int imu_spec = spec0;
  // (named scalars, not arrays: with 128 registers per lane the compiler parks small private arrays in scratch memory)
  double imu_v0 = 0, imu_v1 = 0, imu_v2 = 0, imu_v3 = 0, imu_v4 = 0, imu_v5 = 0;
  int imu_w0 = -1, imu_w1 = -1, imu_w2 = -1, imu_w3 = -1, imu_w4 = -1, imu_w5 = -1;
  if (imu_fast && imu_items > 0) {   // (uniform)
    auto src = W.imu_lin[imu_spec];
    auto dst = W.imu_fastw;
#define BA_IMU_REQ(j) { const int idx = tl + (j) * IMU_NL; const bool on = t64 >= 0 && idx < imu_items; imu_w##j = dst[on ? idx : 0]; \
                        imu_v##j = src[on ? idx : 0]; }   /* (a lane without an entry: masked behind the barrier, see BA_IMU_MASK) */
    BA_IMU_REQ(0) BA_IMU_REQ(1) BA_IMU_REQ(2) BA_IMU_REQ(3) BA_IMU_REQ(4) BA_IMU_REQ(5)
#undef BA_IMU_REQ
  }
  // ---- the entries of the staged priors (formed by wave 0 behind its decision, added behind the IMU records): J^T J / J^T r
  // with the expressions of add_small_factor (same bits).  Pose prior p and speed/bias prior p never share a block, so they go
  // in one pass: two tasks per lane and pass — 21 + 6 entries of the pose prior, 45 + 9 of the speed/bias prior.
  const int goff = (int)(s_g - smem);   // the gradient as an offset into the dynamic LDS, like the matrix entries
  const bool w0pri = !LARGE && n_pri > 0 && (W.n_pprior > 0 || W.n_sbprior > 0);
  // (named scalars, not arrays: with 128 registers per lane the compiler parks small private arrays in scratch memory)
  static_assert(PRI_STAGE == 2, "the four tasks below");
  double pr_v00 = 0, pr_v01 = 0, pr_v10 = 0, pr_v11 = 0;
  int pr_o00 = -1, pr_o01 = -1, pr_o10 = -1, pr_o11 = -1, pr_d00 = -1, pr_d01 = -1, pr_d10 = -1, pr_d11 = -1;
  auto prior_task = [&](int p, int t, double& val, int& off, int& d2i) {   // task t of pass p (wave 0 only)
    if (t < 27) {
      if (p < W.n_pprior) {
        const double* L = s_pri + 42 * p;
        const int* co = s_pricol + 6 * p;
        if (t < 21) {
          int a = 0;
          while ((a + 1) * (a + 2) / 2 <= t) ++a;
          const int b = t - a * (a + 1) / 2;
          const int ra = co[a], rb = co[b];
          if (ra >= 0 && rb >= 0) {
            double sacc = 0;
            for (int k = 0; k < 6; ++k) sacc += L[k * 6 + a] * L[k * 6 + b];
            val = sacc;
            off = LY.at(ra, rb);
            d2i = a == b ? ra : -1;
          }
        } else {
          const int a = t - 21, ra = co[a];
          if (ra >= 0) {
            double sacc = 0;
            for (int k = 0; k < 6; ++k) sacc += L[k * 6 + a] * L[36 + k];
            val = sacc;
            off = goff + ra;
          }
        }
      }
    } else if (t < 81) {
      if (p < W.n_sbprior) {
        // J^T J and J^T r are sign-invariant / sign-flipped: +sqrtInfo with -r
        const double* Jc = s_pri + PRI_STAGE * 51 + 81 * p;
        const double* r = s_pri + PRI_STAGE * 42 + 9 * p;
        const int* co = s_pricol + PRI_STAGE * 6 + 9 * p;
        if (t < 72) {
          const int e = t - 27;
          int a = 0;
          while ((a + 1) * (a + 2) / 2 <= e) ++a;
          const int b = e - a * (a + 1) / 2;
          const int ra = co[a], rb = co[b];
          if (ra >= 0 && rb >= 0) {
            double sacc = 0;
            for (int k = 0; k < 9; ++k) sacc += Jc[k * 9 + a] * Jc[k * 9 + b];
            val = sacc;
            off = LY.at(ra, rb);
            d2i = a == b ? ra : -1;
          }
        } else {
          const int a = t - 72, ra = co[a];
          if (ra >= 0) {
            double sacc = 0;
            for (int k = 0; k < 9; ++k) sacc -= Jc[k * 9 + a] * r[k];
            val = sacc;
            off = goff + ra;
          }
        }
      }
    }
  };
  auto prior_tasks = [&]() {
    prior_task(0, tid, pr_v00, pr_o00, pr_d00);
    prior_task(0, tid + 64, pr_v01, pr_o01, pr_d01);
    prior_task(1, tid, pr_v10, pr_o10, pr_d10);
    prior_task(1, tid + 64, pr_v11, pr_o11, pr_d11);
  };
  // one pass: every read of the wave leaves before its first write (the entries of a pass are disjoint)
  auto prior_apply = [&](double va, int oa, int da, double vb, int ob, int db) {
    const double a0 = oa >= 0 ? smem[oa] : 0.0, a1 = da >= 0 ? s_d2[da] : 0.0;
    const double b0 = ob >= 0 ? smem[ob] : 0.0, b1 = db >= 0 ? s_d2[db] : 0.0;
    if (oa >= 0) smem[oa] = a0 + va;
    if (da >= 0) s_d2[da] = a1 + va;
    if (ob >= 0) smem[ob] = b0 + vb;
    if (db >= 0) s_d2[db] = b1 + vb;
  };
  // The chunk partials of linearisation buffer `buf` -> S and the pose part of the three vectors (waves 1 .. 15).  use_sums:
  // the helper workgroups have summed them (into W.spart_sum).  One item = one double of the partials' record (lower triangle
  // of the pose part in 6x6 blocks, then  Y b | g | diag U): lanes on consecutive doubles (coalesced; three items per lane
  // and eight chunks per trip are requested together — the loads come from other CUs' stores, what counts is the number of
  // dependent rounds), scattered into the 16x16 accumulator-layout blocks of the LDL^T solver.
  // (readfirstlane: the index is uniform, but it comes from a vector load — as a VGPR it drags every address of the sums into
  // vector registers and the kernel into 100 spills)
  const int sum_spec = DBUF ? spec0 : 0;   // the buffer that is accepted if the pending trial is
  STAMP(41);   // (every request of the prologue has been issued)
  // (pre: the three sums of a lane's items, formed by the caller from loads it requested earlier — launches without helpers ask
  // for the chunk partials BEFORE they clear the matrix area, so that the clearing runs under the loads' round trip)
  auto sum_partials = [&](int buf, bool use_sums, const double* pre = nullptr) {
    if constexpr (!LARGE) {
      const int npose_blk = Dp / 6;
      const int nP = npose_blk * (npose_blk + 1) / 2 * 36, ntot = nP + 3 * Dp;
      constexpr int NL = SOLVE_THREADS - 64;
      auto ssum = W.spart_sum;
      const size_t stride = W.spart_stride;
      const int nch = W.n_chunk;
      auto sp = W.spart + (DBUF ? (size_t)buf * W.spart_buf_stride : (size_t)0);
      for (int base = tid - 64; base < ntot; base += 3 * NL) {
        double a[3] = {0, 0, 0};
        int dst[3];
        if (pre) {
          a[0] = pre[0], a[1] = pre[1], a[2] = pre[2];
        } else if (!use_sums) {
          // large launches: summed here, lanes on consecutive doubles, three items per lane and eight chunks per trip requested
          // together (the loads come from other CUs' stores: what counts is the number of dependent rounds)
          constexpr int CB = 9;   // chunks per trip (a separate Schur launch leaves 9 chunks at configs[1]: ONE trip; a fused window has up to 34)
          for (int ch = 0; ch < nch; ch += CB) {
            double v[3][CB];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int u = 0; u < CB; ++u) v[t][u] = (ch + u < nch && base + t * NL < ntot) ? sp[(size_t)(ch + u) * stride + base + t * NL] : 0.0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int u = 0; u < CB; ++u) a[t] += v[t][u];
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int i = base + t * NL;
          if (use_sums) a[t] = i < ntot ? __hip_atomic_load(&ssum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          dst[t] = -1;
          if (i < nP) {
            const int q = i / 36, e = i - 36 * q, ii = e / 6, jj = e - 6 * ii;
            int bi = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
            while (bi * (bi + 1) / 2 > q) --bi;
            const int bj = q - bi * (bi + 1) / 2;
            if (bi > bj || ii >= jj) dst[t] = LY.at(6 * bi + ii, 6 * bj + jj);   // (the upper halves of the diagonal blocks are not part of the lower triangle)
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int i = base + t * NL;
          if (dst[t] >= 0) {
            S[dst[t]] = a[t];
          } else if (i >= nP && i < ntot) {   // Y b | g | diag U of the pose part
            const int which = (i - nP) / Dp, j = (i - nP) - which * Dp;
            (which == 0 ? s_rhs : (which == 1 ? s_g : s_d2))[j] = a[t];
          }
        }
      }
    }
  };
  // ------------------------------------------------------------------ 1. decision
  if (tid < 64) {
    double sums[6] = {0, 0, 0, 0, 0, 0};
    Decision d;
    d.accept = 0; d.term = 0;
    // the scalar partials of the trial (per group, per IMU factor, priors): requested now, with the decision record of the Schur
    // kernel, and reduced below only when no Schur launch has decided (fused mode)
    // (the control record itself: one coalesced load, requested first, into the LDS copy that decide() reads and lane 0 updates)
    double cword = 0;
    if (tid < (int)(sizeof(Ctrl) / 8)) cword = as_global(reinterpret_cast<const double*>(gctrl))[tid];
    double tpart[6] = {0, 0, 0, 0, 0, 0};
    if (chead.y) wave_trial_partials(W, 1 - chead.x, tid, tpart);
    if (tid < (int)(sizeof(Ctrl) / 8)) reinterpret_cast<double*>(&c)[tid] = cword;
    STAMP(42);   // (control record and scalar partials of the trial have arrived)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
    const int pending = c.pending;
    DecisionDL dl;
    dl.accept = 0; dl.term = 0; dl.explicit_next = 0; dl.judged = 0;
    if (pending) {
      // the Schur kernel of this iteration has taken the decision on the same inputs with the same function and left it in
      // W.dec (requested together with the control record above); without a Schur launch it is computed here
      __shared__ double s_dec[DEC_COUNT];
      if (tid < DEC_COUNT) s_dec[tid] = dec_pref;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      asm volatile("" ::: "memory");
      if (s_dec[DEC_VALID] != 0.0) {
        for (int k = 0; k < 6; ++k) sums[k] = s_dec[DEC_SUMS + k];
        if (opt.dogleg) {
          dl.accept = (int)s_dec[DEC_DL + 0]; dl.term = (int)s_dec[DEC_DL + 1]; dl.explicit_next = (int)s_dec[DEC_DL + 2];
          dl.judged = (int)s_dec[DEC_DL + 3]; dl.invalid_steps = (int)s_dec[DEC_DL + 4]; dl.have_tot = (int)s_dec[DEC_DL + 5];
          dl.radius = s_dec[DEC_DL + 6]; dl.mu = s_dec[DEC_DL + 7]; dl.rho = s_dec[DEC_DL + 8]; dl.model_change = s_dec[DEC_DL + 9];
          dl.tot_C = s_dec[DEC_DL + 10]; dl.tot_E = s_dec[DEC_DL + 11];
        } else {
          d.accept = (int)s_dec[DEC_LM + 0]; d.term = (int)s_dec[DEC_LM + 1]; d.radius = s_dec[DEC_LM + 2];
          d.decrease_factor = s_dec[DEC_LM + 3]; d.rho = s_dec[DEC_LM + 4]; d.model_change = s_dec[DEC_LM + 5];
        }
        if (tid == 0) W.dec[DEC_VALID] = 0.0;   // consumed
      } else {
        wave_trial_reduce(tpart, sums);
        if (opt.dogleg) decide_dl_inl(&c, &opt, sums, final_only != 0, &dl);
        else decide_inl(&c, &opt, sums, &d);
      }
    }
    if (tid == 0) {
      s_accepted = 0;
      s_was_pending = pending;
      s_was_first = c.first;
      s_cost_change = 0;
      s_old_cost = c.cost;
      s_lm_gmax = 0;
      s_fail = 0;
      if (opt.dogleg) {
        if (pending) {
          c.last_rho = dl.rho;
          c.last_model_change = dl.model_change;
          c.have_tot = dl.have_tot;
          c.tot_C = dl.tot_C;
          c.tot_E = dl.tot_E;
          c.invalid_steps = dl.invalid_steps;
          c.mu = dl.mu;
          c.radius = dl.radius;
          if (dl.accept) {
            s_accepted = 1;
            s_cost_change = c.cost - sums[0];
            c.acc = 1 - c.acc;
            c.cost = sums[0];
            if (!c.first) c.successful++;
            s_lm_gmax = sums[5];
          }
          c.explicit_next = dl.explicit_next ? (dl.judged ? 1 : 2) : 0;
          if (dl.term) c.done = dl.term + 1;
          c.pending = 0;
        } else if (!final_only && c.explicit_next != 2 && c.iter >= c.max_iter) {
          c.done = 6 + 1;   // the iteration budget of this call is used up
        }
      } else if (pending) {
        c.last_rho = d.rho;
        c.last_model_change = d.model_change;
        if (d.term) {
          c.done = d.term + 1;
          c.radius = d.radius;
          c.decrease_factor = d.decrease_factor;
        } else if (d.accept) {
          s_accepted = 1;
          s_cost_change = c.cost - sums[0];
          c.acc = 1 - c.acc;
          c.cost = sums[0];
          if (!c.first) c.successful++;
          c.radius = d.radius;
          c.decrease_factor = d.decrease_factor;
          s_lm_gmax = sums[5];
        } else {
          c.radius = d.radius;
          c.decrease_factor = d.decrease_factor;
        }
        c.pending = 0;
      }
      if (W.prof && blockIdx.x == 0) W.prof[3] = (double)clock64();
    }
    if (w0pri) {
      // the prior records into LDS (this wave's own writes, read back by itself) and their entries into registers — with the
      // records of the buffer the decision has just named (a rejected trial: the other buffer's, requested again)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      asm volatile("" ::: "memory");
      const int accn = __builtin_amdgcn_readfirstlane(c.acc);
      const int npp = W.n_pprior, nsp = W.n_sbprior;
      if (accn != spec0) {
        auto pa = W.pp_lin[accn];
        pri_a0 = pa[tid < npp * 42 ? tid : 0];
        pri_a1 = pa[tid + 64 < npp * 42 ? tid + 64 : 0];
        pri_b = W.sbp_lin[accn][tid < nsp * 9 ? tid : 0];
      }
      if (tid < npp * 42) s_pri[tid] = pri_a0;
      if (tid + 64 < npp * 42) s_pri[tid + 64] = pri_a1;
      if (tid < nsp * 9) s_pri[PRI_STAGE * 42 + tid] = pri_b;
      if (tid < nsp * 81) s_pri[PRI_STAGE * 51 + tid] = pri_c0;
      if (tid + 64 < nsp * 81) s_pri[PRI_STAGE * 51 + tid + 64] = pri_c1;
      if (tid + 128 < nsp * 81) s_pri[PRI_STAGE * 51 + tid + 128] = pri_c2;
      if (tid < npp * 6) s_pricol[tid] = pri_col;
      else if (tid < npp * 6 + nsp * 9) s_pricol[PRI_STAGE * 6 + tid - npp * 6] = pri_col;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      asm volatile("" ::: "memory");
      prior_tasks();
      if (W.prof && tid == 0 && blockIdx.x == 0) W.prof[44] = (double)clock64();   // diagnostics: prior entries formed
    }
  } else {
    // meanwhile the other waves prepare what does not depend on the decision
    if constexpr (LARGE) {
      // the HBM matrix is normally left zero by large_export_kernel, which clears every entry it reads (the flag behind the
      // mode word says so; set by solve_large_tail_kernel, dropped below as soon as this launch starts to assemble)
      if (W.ct_flag[W.ct_nT * (W.ct_nT + 1) / 2 + 2] == 0)
        for (int i = tid - 64; i < nS; i += SOLVE_THREADS - 64) S[i] = 0.0;
    } else {
      // The Schur partials do not depend on this kernel's decision (the Schur kernel made the same one and
      // reduced the buffer that is being accepted): sum them into S while wave 0 decides.  One item = one double of the
      // partials' record (lower triangle of the pose part in 6x6 blocks, then  Y b | g | diag U): its sum over the chunks,
      // lanes on consecutive doubles (coalesced; three items per lane and eight chunks per trip are requested together —
      // the loads come from other CUs' stores, what counts is the number of dependent rounds), scattered into the 16x16
      // accumulator-layout blocks of the LDL^T solver.  Everything no item writes starts from zero.
      const int npose_blk = Dp / 6;
      const int nP = npose_blk * (npose_blk + 1) / 2 * 36, ntot = nP + 3 * Dp;
      constexpr int NL = SOLVE_THREADS - 64;
      // launches without helpers whose partials fit one trip (at most 9 chunks, at most three items per lane — configs[1] behind
      // a separate Schur launch): the loads leave here, in front of the clearing of the matrix area
      constexpr int PCB = 9;
      const bool early = !helped && W.n_chunk <= PCB && ntot <= 3 * NL;
      double pv0[PCB], pv1[PCB], pv2[PCB];
      if (early) {
        const size_t stride = W.spart_stride;
        const int nch = W.n_chunk;
        auto sp = W.spart + (DBUF ? (size_t)sum_spec * W.spart_buf_stride : (size_t)0);
        const int i0 = tid - 64;
#pragma unroll
        for (int u = 0; u < PCB; ++u) {
          pv0[u] = (u < nch && i0 < ntot) ? sp[(size_t)u * stride + i0] : 0.0;
          pv1[u] = (u < nch && i0 + NL < ntot) ? sp[(size_t)u * stride + i0 + NL] : 0.0;
          pv2[u] = (u < nch && i0 + 2 * NL < ntot) ? sp[(size_t)u * stride + i0 + 2 * NL] : 0.0;
        }
      }
      {
        // (solver coordinates: the pose part sits behind the speed/bias part, rows / columns Ds .. D - 1; an item lands on every
        // entry (r, c) of the upper triangle with Ds <= r <= c < D)
        // (chain mode: the pose system alone has this layout — no speed/bias part in front of it, Dp rows — and the chain's
        // arrays behind it start from zero as a whole)
        const L16& PL = [&]() -> const L16& {
          if constexpr (CHAIN) return LY.P;
          else return LY;
        }();
        const int nb16 = PL.nb, nblk16 = L16::blocks(nb16);
        const int Ds = CHAIN ? 0 : D - Dp;
        const int Dz = CHAIN ? Dp : D;
        if constexpr (CHAIN)
          for (int i = LY.oA + tid - 64; i < LY.total; i += NL) S[i] = 0.0;   // (oA = the end of the pose system's blocks)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6) - 1, l = tid & 63;
        int I = 0, rem = wv;   // block wv, wv + 15, ... of the row-major upper triangle -> (I, I + rem)
        for (int b = wv; b < nblk16; b += SOLVE_THREADS / 64 - 1) {
          while (rem >= nb16 - I) {
            rem -= nb16 - I;
            ++I;
          }
          const int J = I + rem;
          const int cc = 16 * J + (l & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * I + 4 * r + (l >> 4);
            if (!(rr >= Ds && cc >= rr && cc < Dz)) S[b * 256 + 64 * r + l] = 0.0;   // no item lands here
          }
          rem += SOLVE_THREADS / 64 - 1;
        }
      }
      // the chunk partials, summed by the helper workgroups: wait for all of them.  The wait is bounded: helpers that are late
      // (not co-scheduled: other streams or processes hold the CUs) do not stop the window, this workgroup then sums the chunk
      // partials itself as a launch without helpers does (same chunk order: the same sums), and the time-out is counted
      // (sum_sync[2]; okvis_ba_helper_timeouts) so that it shows in tests and in bench.py.
      // what the prologue requested for LDS: stored here, where the helpers' sums (or this workgroup's own loads of the chunk
      // partials) are the longer wait
      auto stage_stores = [&]() {
        if (t64 < WD_N) s_Wd[t64] = wd_v;
        if (pre_on && t64 < PRE_BLOCKS * 8) {
          const int np7 = 7 * W.n_pose, ns9 = 9 * W.n_sb;
          const int i0 = 2 * t64, i1 = 2 * t64 + 1;
          s_pre[i0] = i0 < PRE_BLOCKS * 7 ? (i0 < np7 ? pre_a0 : 0.0) : (i0 - PRE_BLOCKS * 7 < ns9 ? pre_b0 : 0.0);
          s_pre[i1] = i1 < PRE_BLOCKS * 7 ? (i1 < np7 ? pre_a1 : 0.0) : (i1 - PRE_BLOCKS * 7 < ns9 ? pre_b1 : 0.0);
        } else if (pre_on && t64 < PRE_BLOCKS * 10) {
          const int to = t64 - PRE_BLOCKS * 8;
          s_preoff[to] = to < PRE_BLOCKS ? (to < W.n_pose ? pre_po : -1) : (to - PRE_BLOCKS < W.n_sb ? pre_so : -1);
        }
      };
      // The flag the other waves follow carries this launch's count (a stale word of an earlier workgroup on this CU cannot
      // be mistaken for it): count << 2 | 1 = the helpers' sums are there, | 2 = they are late: summed here.
      const int sum_expected = sum_launches + ((int)gridDim.y - 1);   // helper arrivals this window must have seen after this launch
      if (helped) stage_stores();
      if (helped && tid == 64) {
        as_global(W.sum_sync)[1] = sum_expected;
        int polls = 0;
        while (__hip_atomic_load(W.sum_sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sum_expected && polls < opt.helper_polls) {
          __builtin_amdgcn_s_sleep(2);
          ++polls;
        }
        if (polls >= opt.helper_polls) W.sum_sync[2] += 1;
        __hip_atomic_store(&s_sum_ready, (sum_expected << 2) | (polls >= opt.helper_polls ? 2 : 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      bool use_sums = false;
      if (helped) {
        // (no agent-scope acquire, which would invalidate the caches: the sums are read with device-coherent loads below)
        int v;
        while (((v = __hip_atomic_load(&s_sum_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 2) != sum_expected || (v & 3) == 0)
          __builtin_amdgcn_s_sleep(1);
        use_sums = (v & 3) == 1;
      }
      if (early) {
        double pre[3] = {0, 0, 0};   // (the chunks in their order: the sums of a launch that asks later, bit for bit)
#pragma unroll
        for (int u = 0; u < PCB; ++u) pre[0] += pv0[u];
#pragma unroll
        for (int u = 0; u < PCB; ++u) pre[1] += pv1[u];
#pragma unroll
        for (int u = 0; u < PCB; ++u) pre[2] += pv2[u];
        sum_partials(sum_spec, false, pre);
      } else {
        sum_partials(sum_spec, use_sums);
      }
      if (!helped) stage_stores();
      for (int i = tid - 64; i < 3 * (Dpad - Dp); i += NL) {   // speed/bias part of the vectors starts from zero
        const int which = i / (Dpad - Dp), j = Dp + i - which * (Dpad - Dp);
        (which == 0 ? s_rhs : (which == 1 ? s_g : s_d2))[j] = 0.0;
      }
      for (int i = tid - 64; i < Dpad; i += SOLVE_THREADS - 64) s_x[i] = 0.0;
    }
    if (W.prof && tid == 64 && blockIdx.x == 0) W.prof[4] = (double)clock64();
  }
  if (W.prof && (tid & 63) == 0 && blockIdx.x == 0) W.prof[170 + (tid >> 6)] = (double)clock64();   // diagnostics: when each wave reaches the barrier of the head
  __syncthreads();
  if (c.done) {
    if (tid == 0) *gctrl = c;
    return;
  }
  const int acc = __builtin_amdgcn_readfirstlane(c.acc);   // (uniform: keeps the buffer selection in scalar registers)
  const WinPtrs& Wl = *[&]() -> const WinPtrs* {
    if constexpr (LARGE) return &W;
    else return reinterpret_cast<const WinPtrs*>(s_Wd);
  }();
  if (LARGE && tid == 0) Wl.ct_flag[Wl.ct_nT * (Wl.ct_nT + 1) / 2 + 2] = 0;   // S is being written from here on

  STAMP(1);
  // ------------------------------------------------------------------ 2. assembly
  if constexpr (LARGE) {
    // pose part of the matrix: summed by large_export_kernel; vectors here
    const int npose_blk = Dp / 6;
    const int nP = npose_blk * (npose_blk + 1) / 2 * 36;
    const size_t stride = Wl.spart_stride;
    const int nch = Wl.n_chunk;
    const double* sp = Wl.spart;
    for (int i = tid; i < 3 * Dpad; i += SOLVE_THREADS) {
      const int which = i / Dpad, j = i - which * Dpad;
      double a = 0;
      if (j < Dp) {
        for (int ch = 0; ch < nch; ch += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = (ch + u < nch) ? sp[(size_t)(ch + u) * stride + nP + which * Dp + j] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) a += v[u];
        }
      }
      (which == 0 ? s_rhs : (which == 1 ? s_g : s_d2))[j] = a;
    }
    for (int i = tid; i < Dpad; i += SOLVE_THREADS) s_x[i] = 0.0;
    __syncthreads();
  }
  STAMP(2);
  const bool rejected = s_was_pending && !s_accepted;   // the speculated buffer was the wrong one (rare): records are reloaded below
  if (imu_fast) {
    if (imu_items > 0) {
#define BA_IMU_MASK(j) if (!(t64 >= 0 && t64 + (j) * IMU_NL < imu_items)) imu_w##j = -1;
      BA_IMU_MASK(0) BA_IMU_MASK(1) BA_IMU_MASK(2) BA_IMU_MASK(3) BA_IMU_MASK(4) BA_IMU_MASK(5)
#undef BA_IMU_MASK
    }
    if (t64 >= 0 && acc != imu_spec) {   // the step was rejected: the records of the other buffer are needed
      auto src = Wl.imu_lin[acc];
#define BA_IMU_RELOAD(j) if (imu_w##j >= 0) imu_v##j = src[t64 + (j) * IMU_NL];
      BA_IMU_RELOAD(0) BA_IMU_RELOAD(1) BA_IMU_RELOAD(2) BA_IMU_RELOAD(3) BA_IMU_RELOAD(4) BA_IMU_RELOAD(5)
#undef BA_IMU_RELOAD
    }
    for (int col = 0; col < Wl.n_imu_color; ++col) {   // factors of one colour touch disjoint blocks
      // (the entries of one colour are disjoint, a work-item's own included: all of its reads leave before the first write —
      // two LDS round trips per colour instead of one per entry.  A word of imu_fastw: offset in the dynamic LDS | (index in the
      // diagonal + 1) << 16 | colour << 24.)
#define BA_IMU_RD(j) const bool on##j = imu_w##j >= 0 && (imu_w##j >> 24) == col; const int d##j = ((imu_w##j >> 16) & 0xFF) - 1; \
                     const double o##j = on##j ? smem[imu_w##j & 0xFFFF] : 0.0; const double q##j = (on##j && d##j >= 0) ? s_d2[d##j] : 0.0;
      BA_IMU_RD(0) BA_IMU_RD(1) BA_IMU_RD(2) BA_IMU_RD(3) BA_IMU_RD(4) BA_IMU_RD(5)
#undef BA_IMU_RD
#define BA_IMU_WR(j) if (on##j) { smem[imu_w##j & 0xFFFF] = o##j + imu_v##j; if (d##j >= 0) s_d2[d##j] = q##j + imu_v##j; }
      BA_IMU_WR(0) BA_IMU_WR(1) BA_IMU_WR(2) BA_IMU_WR(3) BA_IMU_WR(4) BA_IMU_WR(5)
#undef BA_IMU_WR
      if (W.prof && tid == 64 && blockIdx.x == 0) W.prof[45 + (col > 0)] = (double)clock64();   // diagnostics: a wave's colour pass is through
      __syncthreads();
    }
  }
  if (W.prof && tid == 0 && blockIdx.x == 0) W.prof[58] = (double)clock64();   // diagnostics: end of the IMU part
  if (pre_on && rejected) {   // rejected step: the other buffer stays accepted
    for (int i = tid; i < PRE_BLOCKS * 16; i += SOLVE_THREADS) {
      double v = 0;
      if (i < PRE_BLOCKS * 7) {
        if (i < 7 * Wl.n_pose) v = Wl.pose[acc][i];
      } else if (i - PRE_BLOCKS * 7 < 9 * Wl.n_sb) {
        v = Wl.sb[acc][i - PRE_BLOCKS * 7];
      }
      s_pre[i] = v;
    }
    __syncthreads();
  }
  if constexpr (DBUF) {
    // one set of Schur partials per linearisation buffer (fused mode): the sums in S and in the three vectors were taken from the
    // buffer of the trial; it was rejected (or replaced by an explicit dogleg step), so they are corrected by the difference of
    // the two sets — here, where the registers of the speculative prefetches are free again, and only in this rare case
    if (rejected && Wl.spart_buf_stride) {
      const int npose_blk = Dp / 6;
      const int nP = npose_blk * (npose_blk + 1) / 2 * 36, ntot = nP + 3 * Dp;
      const size_t stride = Wl.spart_stride;
      const int nch = Wl.n_chunk;
      auto right = Wl.spart + (size_t)acc * Wl.spart_buf_stride;
      auto wrong = Wl.spart + (size_t)(1 - acc) * Wl.spart_buf_stride;
      for (int i = tid; i < ntot; i += SOLVE_THREADS) {
        double a = 0;
        for (int ch = 0; ch < nch; ++ch) a += right[(size_t)ch * stride + i] - wrong[(size_t)ch * stride + i];
        if (i < nP) {
          const int q = i / 36, e = i - 36 * q, ii = e / 6, jj = e - 6 * ii;
          int bi = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
          while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
          while (bi * (bi + 1) / 2 > q) --bi;
          const int bj = q - bi * (bi + 1) / 2;
          if (bi > bj || ii >= jj) S[LY.at(6 * bi + ii, 6 * bj + jj)] += a;
        } else {
          const int which = (i - nP) / Dp, j = (i - nP) - which * Dp;
          (which == 0 ? s_rhs : (which == 1 ? s_g : s_d2))[j] += a;
        }
      }
      __syncthreads();
    }
  }
  if (w0pri) {   // (formed by wave 0 in front of the barrier)
    if (tid < 64) prior_apply(pr_v00, pr_o00, pr_d00, pr_v01, pr_o01, pr_d01);
    __syncthreads();
    if (Wl.n_pprior > 1 || Wl.n_sbprior > 1) {
      if (tid < 64) prior_apply(pr_v10, pr_o10, pr_d10, pr_v11, pr_o11, pr_d11);
      __syncthreads();
    }
  }
  // what is left: factors the fast paths above do not cover (more IMU entries than the prefetch holds, more priors than are
  // staged, relative pose factors, the marginalisation prior); every part ends with its own barrier
  assemble_base(Wl, acc, LY, S, s_g, s_d2, s_coloff, tid, SOLVE_THREADS, imu_fast, s_pri, s_pricol, n_pri, LARGE, w0pri);
  STAMP(5);
  const double lambda = opt.dogleg ? c.mu : 1.0 / c.radius;
  const bool est_scale = s_was_first && s_accepted;   // first linearisation of this call: estimate the Jacobi scale
  // convergence measure of the accepted step: partial maxima per wave
  auto conv_partial = [&]() {
    double m = 0;
    if (opt.dogleg) {
      // Ceres 1.9: gradient_max_norm = || x - Plus(x, -g) ||_inf over the ambient coordinates: the gradient itself for
      // Euclidean blocks, the change of the quaternion coefficients for the rotation part of a pose
      // (the pose blocks on wave 4's lanes: waves 0 .. 2 carry the damping of the same phase)
      for (int b = (tid + SOLVE_THREADS - 256) % SOLVE_THREADS; b < Wl.n_pose; b += SOLVE_THREADS) {
        const int off = (pre_on && b < PRE_BLOCKS) ? s_preoff[b] : Wl.pose_off[b];
        if (off < 0) continue;
        const double* xp = (pre_on && b < PRE_BLOCKS) ? s_pre + 7 * b : Wl.pose[acc] + 7 * (size_t)b;
        double xin[7], dneg[6], xo[7];
        for (int k = 0; k < 7; ++k) xin[k] = xp[k];
        for (int k = 0; k < 6; ++k) dneg[k] = -s_g[off + k];
        pose_oplus_dev(xin, dneg, xo);
        for (int k = 0; k < 7; ++k) m = fmax(m, fabs(xin[k] - xo[k]));
      }
      for (int i = Dp + tid; i < D; i += SOLVE_THREADS) m = fmax(m, fabs(s_g[i]));
      // (extrinsics-role blocks are pose blocks of the same array: covered by the loop over n_pose)
    } else {
      for (int i = tid; i < D; i += SOLVE_THREADS) m = fmax(m, fabs(s_g[i]));
    }
    m = wave_max_full(m);
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
  };
  // the decision every work-item derives from the partial maxima (the same inputs, the same expressions: uniform); work-item 0
  // records it.  Returns the termination code (0 = go on).
  auto conv_decide = [&]() -> int {
    double gm = s_lm_gmax;
    for (int i = 0; i < SOLVE_THREADS / 64; ++i) gm = fmax(gm, s_red[i]);
    int done_now = 0;
    if (s_accepted) {
      const double tol = s_was_first ? (opt.dogleg ? opt.gradient_tolerance : opt.gradient_tolerance * fmax(gm, 2.220446049250313e-16))
                                     : c.abs_grad_tol;
      if (opt.gradient_tolerance > 0 && gm <= tol) {
        done_now = 2 + 1;
      } else if (!opt.dogleg && !s_was_first && opt.function_tolerance > 0 &&
                 fabs(s_cost_change) < opt.function_tolerance * s_old_cost) {
        done_now = 1 + 1;   // (dogleg: tested before the step is taken, in decide_dl)
      }
    }
    return done_now;
  };
  auto damping = [&]() {
    for (int i = tid; i < Dpad; i += SOLVE_THREADS) {
      if (i < D) {
        double sc = 1.0;
        if (opt.dogleg) {
          if (est_scale) {
            sc = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(s_d2[i])) : 1.0;
            Wl.scale_p[i] = sc;
          } else {
            sc = i == tid ? scale_pref : Wl.scale_p[i];
          }
        }
        const double d2 = damp_diag(s_d2[i], sc, opt);
        s_d2[i] = d2;
        S[LY.at(i, i)] += lambda * d2;
        const double rh = s_rhs[i] - s_g[i];
        s_rhs[i] = rh;
        if constexpr (!LARGE) S[LY.at(D, i)] = rh;   // the right-hand side rides along as column D of the matrix (ba_ldl16.hpp)
      } else {
        if constexpr (LARGE) S[LY.at(i, i)] = 1.0;  // identity padding up to a multiple of 6
        s_rhs[i] = 0.0;
      }
    }
  };
  if (final_only) {
    // cost / gradient evaluation and the marginalisation pass: nothing is damped or solved
    conv_partial();
    __syncthreads();
    if (tid == 0) {
      double gm = s_lm_gmax;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) gm = fmax(gm, s_red[i]);
      if (s_accepted) {
        c.grad_max = gm;
        if (s_was_first) {
          c.initial_cost = c.cost;
          c.abs_grad_tol = opt.dogleg ? opt.gradient_tolerance : opt.gradient_tolerance * fmax(gm, 2.220446049250313e-16);
        }
        if (opt.gradient_tolerance > 0 && gm <= c.abs_grad_tol) {
          c.done = 2 + 1;
        } else if (!opt.dogleg && !s_was_first && opt.function_tolerance > 0 &&
                   fabs(s_cost_change) < opt.function_tolerance * s_old_cost) {
          c.done = 1 + 1;   // (dogleg: tested before the step is taken, in decide_dl)
        }
        c.first = 0;
      }
    }
    __syncthreads();
    if (Wl.grad)
      for (int i = tid; i < D; i += SOLVE_THREADS) Wl.grad[i] = s_g[i];
    if (final_only == 2) {
      // marginalisation pass (okvis_ba_marginalize): export the undamped system left after the landmark
      // elimination, H (D x D, full symmetric) and b0 = -(g - W V^+ b_l)  (MarginalizationError.cpp:682-684)
      for (int i = tid; i < D; i += SOLVE_THREADS) Wl.rhs[i] = s_rhs[i] - s_g[i];
      if constexpr (LARGE) {
        // the matrix of a large window is completed by large_export_kernel (Schur partials of the pose part, IMU terms), which
        // also writes the full symmetric copy into W.S: ask for it and stop before any damping is added
        if (tid == 0) {
          *gctrl = c;
          Wl.ct_flag[Wl.ct_nT * (Wl.ct_nT + 1) / 2 + 1] = 1;
        }
        return;
      }
      for (int k = tid; k < D * D; k += SOLVE_THREADS) {
        const int i = k / D, j = k - i * D;
        Wl.S[k] = (i >= j) ? S[LY.at(i, j)] : S[LY.at(j, i)];
      }
    }
    if (tid == 0) *gctrl = c;
    return;
  }
  // ------------------------------------------------------------------ 3 + 4. convergence of the accepted step, damping
  // One phase: the partial maxima of the convergence test, the damping of the diagonal (it does not depend on the test: a
  // window that stops here simply does not use it) and the right-hand side column; behind ONE barrier every work-item derives
  // the same verdict from the partial maxima, work-item 0 records it.
  conv_partial();
  damping();
  if (Wl.grad)
    for (int i = tid; i < D; i += SOLVE_THREADS) Wl.grad[i] = s_g[i];
  if (tid == 0 && opt.dogleg && c.mu >= DL_MAX_MU) s_fail = 1;   // DoglegStrategy: no solve is attempted once mu has reached max_mu
  __syncthreads();
  {
    const int verdict = conv_decide();
    if (tid == 0 && s_accepted) {
      double gm = s_lm_gmax;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) gm = fmax(gm, s_red[i]);
      c.grad_max = gm;
      if (s_was_first) {
        c.initial_cost = c.cost;
        c.abs_grad_tol = opt.dogleg ? opt.gradient_tolerance : opt.gradient_tolerance * fmax(gm, 2.220446049250313e-16);
      }
      if (verdict & 3) c.done = verdict & 3;
      c.first = 0;
    }
    if (verdict & 3) {
      if (tid == 0) *gctrl = c;
      return;
    }
  }
  if (Wl.S) {  // parity/debug copy of the damped system
    for (int k = tid; k < D * D; k += SOLVE_THREADS) {
      const int i = k / D, j = k - i * D;
      Wl.S[k] = (i >= j) ? S[LY.at(i, j)] : S[LY.at(j, i)];
    }
    for (int i = tid; i < D; i += SOLVE_THREADS) {
      Wl.rhs[i] = s_rhs[i];
      Wl.Dp2[i] = s_d2[i];
    }
  }
  STAMP(6);
  if constexpr (!LARGE) {
    if (opt.dogleg && c.explicit_next) {
      // ------------------------------------------------------------ explicit dogleg step (no factorisation)
      // delta = -cA xv + beta dGN from the stored Gauss-Newton point dGN (Wl.step): after a rejected step (radius halved,
      // Ceres' reuse_) or when the speculative Gauss-Newton trial turned out to lie outside the trust region.
      // xv_i = g_i / Dt2_i is the Cauchy direction; its step length needs xv^T H xv, evaluated here from the damped
      // reduced matrix S (just assembled, not factorised) and one pass over the landmarks:
      //   xv^T H xv = xv_p^T S xv_p - mu sum Dt2 xv_p^2 + sum_l [ t^T (V_l+mu Dt2_l)^-1 t + 2 t.xv_l + xv_l^T V_l xv_l ],  t = W_l^T xv_p
      __shared__ double s_dl[2];
      __shared__ double s_sc5[5][SOLVE_THREADS / 64];
      const double mu = c.mu;
      for (int i = tid; i < Dpad; i += SOLVE_THREADS) {
        s_x[i] = i < D ? s_g[i] / s_d2[i] : 0.0;
        s_rhs[i] = i < D ? Wl.step[i] : 0.0;
      }
      __syncthreads();
      double q1 = 0, q2 = 0, Ap = 0, Al = 0, Bl = 0;
      for (int k = tid; k < D * D; k += SOLVE_THREADS) {
        const int i = k / D, j = k - i * D;
        const double v = (i >= j) ? S[LY.at(i, j)] : S[LY.at(j, i)];
        q1 += v * s_x[i] * s_x[j];
      }
      for (int i = tid; i < D; i += SOLVE_THREADS) {
        q2 += s_d2[i] * s_x[i] * s_x[i];
        Ap += s_g[i] * s_x[i];
      }
      dl_landmark_sums(Wl, acc, s_x, mu, opt, tid, &Al, &Bl);
      q1 = wave_sum_full(q1);
      q2 = wave_sum_full(q2);
      Ap = wave_sum_full(Ap);
      Al = wave_sum_full(Al);
      Bl = wave_sum_full(Bl);
      if ((tid & 63) == 0) {
        s_sc5[0][tid >> 6] = q1;
        s_sc5[1][tid >> 6] = q2;
        s_sc5[2][tid >> 6] = Ap;
        s_sc5[3][tid >> 6] = Al;
        s_sc5[4][tid >> 6] = Bl;
      }
      __syncthreads();
      if (tid == 0) {
        double a[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5; ++k)
          for (int i = 0; i < SOLVE_THREADS / 64; ++i) a[k] += s_sc5[k][i];
        double cA, beta, dln, model;
        dogleg_coefficients(a[2] + a[3], (a[0] - mu * a[1]) + a[4], c.tot_C, c.tot_E, mu, c.radius, &cA, &beta, &dln, &model);
        s_dl[0] = cA;
        s_dl[1] = beta;
        c.cA = cA;
        c.beta = beta;
        c.dl_norm = dln;
        c.pend_model = model;
        c.tot_A = a[2] + a[3];
      }
      __syncthreads();
      {
        const double cA = s_dl[0], beta = s_dl[1];
        for (int i = tid; i < Dpad; i += SOLVE_THREADS) s_x[i] = -cA * s_x[i] + beta * s_rhs[i];
      }
      __syncthreads();
      double s2 = 0, x2 = 0;
      trial_states(Wl, acc, s_x, tid, true, &s2, &x2, pre_on ? s_pre : nullptr, pre_on ? s_preoff : nullptr);
      s2 = wave_sum_full(s2);
      x2 = wave_sum_full(x2);
      if ((tid & 63) == 0) {
        s_sc5[0][tid >> 6] = s2;
        s_sc5[1][tid >> 6] = x2;
      }
      __syncthreads();
      if (tid == 0) {
        double a0 = 0, a1 = 0;
        for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
          a0 += s_sc5[0][i];
          a1 += s_sc5[1][i];
        }
        c.step2_p = a0;
        c.x2_p = a1;
        c.tr_kind = 1;
        c.spec_discard = c.explicit_next == 2;   // (the IMU terms take back what the discarded evaluation did to them: ba_imu.hpp)
        if (c.explicit_next == 1) c.iter++;   // a new iteration (after a rejection); 2 = same iteration redone
        c.explicit_next = 0;
        c.lambda = mu;
        c.pending = 1;
        *gctrl = c;
      }
      return;
    }
  }
  if constexpr (LARGE) {
    // hand the damped system to the tiled multi-workgroup solver (ba_chol_tiles.hpp).  S (HBM, block-packed)
    // holds the IMU / prior / marginalisation part plus the damping; large_export_kernel adds the Schur partials
    // of the pose part and writes the 48x48 tiles on many CUs; the trust-region state travels in ctrl
    const int nT = Wl.ct_nT, ntile = nT * (nT + 1) / 2;
    for (int i = tid; i < nT * CT_TB; i += SOLVE_THREADS) {
      Wl.ct_rhs[i] = i < Dpad ? s_rhs[i] : 0.0;
      Wl.ct_g[i] = i < D ? s_g[i] : 0.0;
      Wl.ct_d2[i] = i < D ? s_d2[i] : 0.0;
    }
    for (int i = tid; i < ntile + 1; i += SOLVE_THREADS) Wl.ct_flag[i] = 0;
    if (tid == 0) Wl.ct_flag[ntile + 3] = 0;   // progress counter of the tiled factorisation
    for (int i = tid; i < 2 * nT; i += SOLVE_THREADS) Wl.ct_flag[ntile + 4 + i] = 0;   // "partial ready" flags
    for (int i = tid; i < nT * CT_TB; i += SOLVE_THREADS)
      reinterpret_cast<unsigned long long*>(Wl.ct_x)[i] = CT_X_SENTINEL;   // no value yet (chol_backsub_task polls the values)
    __syncthreads();
    if (tid == 0) {
      *gctrl = c;
      Wl.ct_flag[ntile + 1] = (opt.dogleg && c.explicit_next) ? 2 : 1;   // 2 = tiles wanted for the dogleg scalars only
    }
    return;
  }
  // ------------------------------------------------------------------ 4b. blocked LDL^T on the matrix core (ba_ldl16.hpp)
  // The right-hand side rides along as column D of the matrix: the elimination turns it into L^-1 b.
  // (column D and the max_mu verdict were written in the damping phase, in front of the barrier every work-item has passed)
  if constexpr (!LARGE) {
    STAMP(10);
    // (diagnostics: the solver's own stamps — load | factor | back-substitution, and with -DLDL_TS_ALL wave 0's steps — as 64-bit
    // integers behind the phase stamps)
    long long* const stamps = (W.prof && blockIdx.x == 0) ? reinterpret_cast<long long*>(W.prof + 64) : nullptr;
    if constexpr (CHAIN) chain_solve<SOLVE_THREADS / 64>(S, LY, tid, s_x, &s_fail, stamps, W.ldl_comp);
    else ldl16_solve<SOLVE_THREADS / 64>(S, D, tid, s_x, &s_fail, stamps, D - Dp, W.ldl_comp);
  }
  STAMP(7);
  if (s_fail) {  // not positive definite: invalid step (handled like a rejection)
    if (tid == 0) {
      c.chol_fail++;
      c.pending = 0;
      if (opt.dogleg) {
        solve_failed_dl(&c, opt);
      } else {
        c.iter++;
        c.radius = c.radius / c.decrease_factor;
        c.decrease_factor *= 2.0;
        if (c.radius < opt.min_radius) c.done = 5 + 1;
      }
      *gctrl = c;
    }
    return;
  }
  STAMP(8);

  // ------------------------------------------------------------------ 5. scalars, trial state, ctrl
  {
    double gd = 0, ddd = 0, s2 = 0;
    for (int i = tid; i < D; i += SOLVE_THREADS) {
      const double x = s_x[i];
      Wl.step[i] = x;
      gd += s_g[i] * x;
      ddd += s_d2[i] * x * x;
      s2 += x * x;
    }
    STAMP(31);
    double x2 = 0, s2a = 0;
    trial_states(Wl, acc, s_x, tid, opt.dogleg != 0, &s2a, &x2, pre_on ? s_pre : nullptr, pre_on ? s_preoff : nullptr);
    STAMP(32);
    if (opt.dogleg) s2 = s2a;   // Ceres' step_norm is |x - x_plus_delta| over the ambient coordinates
    __shared__ double s_sc[4][SOLVE_THREADS / 64];
    gd = wave_sum_full(gd);
    ddd = wave_sum_full(ddd);
    s2 = wave_sum_full(s2);
    x2 = wave_sum_full(x2);
    if ((tid & 63) == 0) {
      s_sc[0][tid >> 6] = gd;
      s_sc[1][tid >> 6] = ddd;
      s_sc[2][tid >> 6] = s2;
      s_sc[3][tid >> 6] = x2;
    }
    __syncthreads();
    STAMP(33);
    if (tid == 0) {
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
        a0 += s_sc[0][i];
        a1 += s_sc[1][i];
        a2 += s_sc[2][i];
        a3 += s_sc[3][i];
      }
      c.gd_p = a0;
      c.ddd_p = a1;
      c.step2_p = a2;
      c.x2_p = a3;
      c.lambda = lambda;
      c.tr_kind = 0;         // dogleg: the Gauss-Newton point, launched speculatively
      c.spec_discard = 0;
      c.explicit_next = 0;
      c.have_tot = 0;
      c.iter++;
      c.pending = 1;
      *gctrl = c;
    }
  }
  STAMP(9);
#undef STAMP
}

// One reduced system through the LDS-resident solvers of solve_kernel, on its own (okvis_ba_reduced_solve: unit tests and the
// A/B timing of the two solvers): Sd = the dense D x D symmetric matrix (row-major), scattered into the solver's LDS layout the
// way the assembly does (LY.at on the lower triangle, the right-hand side as "row D"), then ldl16_solve (CHAIN = false) or
// chain_solve.  out: x[D], then ticks[0] = clock64 ticks of the solve, info[0] = 1 when a pivot was not positive.
template <bool CHAIN>
__global__ __launch_bounds__(SOLVE_THREADS) void reduced_solve_kernel(const double* __restrict__ Sd, const double* __restrict__ rhs, int D, int Dp,
                                                                      unsigned comp_mask, double* __restrict__ x, long long* __restrict__ ticks,
                                                                      int* __restrict__ info, int repeats, double* __restrict__ dump) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  typedef typename std::conditional<CHAIN, LChain, L16>::type LYT;
  const LYT LY = [&]() {
    if constexpr (CHAIN) return LChain::make(D, Dp);
    else return L16{ldl16_nb(D), D - Dp, D};
  }();
  const int nS = CHAIN ? LChain::make(D, Dp).total : ldl16_area_doubles(D);
  double* s_x = smem + nS;
  __shared__ int s_fail;
  long long total = 0;
  for (int rep = 0; rep < repeats; ++rep) {
    for (int i = tid; i < nS; i += SOLVE_THREADS) smem[i] = 0.0;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int k = tid; k < D * D; k += SOLVE_THREADS) {
      const int i = k / D, j = k - i * D;
      if (i >= j) {
        const double v = Sd[k];
        if (v != 0.0 || i == j) smem[LY.at(i, j)] = v;
      }
    }
    for (int i = tid; i < D; i += SOLVE_THREADS) smem[LY.at(D, i)] = rhs[i];
    __syncthreads();
    const long long t0 = clock64();
    if constexpr (CHAIN) chain_solve<SOLVE_THREADS / 64>(smem, LY, tid, s_x, &s_fail, nullptr, comp_mask);
    else ldl16_solve<SOLVE_THREADS / 64>(smem, D, tid, s_x, &s_fail, nullptr, D - Dp, comp_mask);
    total += clock64() - t0;
    __syncthreads();
  }
  for (int i = tid; i < D; i += SOLVE_THREADS) x[i] = s_x[i];
  if (dump)   // (diagnostics: the solver's LDS image as the solve left it)
    for (int i = tid; i < nS; i += SOLVE_THREADS) dump[i] = smem[i];
  if (tid == 0) {
    ticks[0] = total / repeats;
    info[0] = s_fail;
  }
}

// ---- large windows (D > MAX_D_LDS): solve_kernel<true> assembles and exports the damped system, the tile
// kernel factorises it on many CUs (fp64 MFMA), this kernel back-substitutes and finishes the iteration exactly
// like section 5 of solve_kernel.
// tile (ti, tj) of the damped reduced system: sum of the Schur partials (pose part) + the block-packed HBM matrix
// solve_kernel<true> assembled (IMU, priors, marginalisation prior, damping); diagonal tiles full, identity padding
__global__ __launch_bounds__(CT_THREADS) void large_export_kernel(const WinPtrs* __restrict__ wins) {
  const WinPtrs& W = wins[blockIdx.y];
  const int nT = W.ct_nT, ntile = nT * (nT + 1) / 2;
  if (nT == 0 || (int)blockIdx.x >= ntile) return;
  if (W.ct_flag[ntile + 1] == 0) return;
  const int t = blockIdx.x;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int D = W.D, Dp = W.Dp, Dpad = ((D + 5) / 6) * 6;
  const SLayout LY{Dpad / 6};
  const int nch = W.n_chunk;
  const size_t stride = W.spart_stride;
  // grid.z slices the tile: one entry per work-item, so the chunk partials of the 28 pose-part tiles are summed by 9 x as many
  // workgroups (the sums are chains of dependent loads: latency, not bandwidth)
  for (int r = blockIdx.z * CT_THREADS + threadIdx.x; r < CT_TILE; r += gridDim.z * CT_THREADS) {
    int gi = CT_TB * ti + r / CT_TB, gj = CT_TB * tj + r % CT_TB;
    const int oi = gi, oj = gj;
    double v;
    if (ti == tj && oi < oj && !W.S) {
      v = 0.0;   // upper triangle of a diagonal tile: never referenced by the factorisation
    } else if (gi < Dpad && gj < Dpad) {
      const bool owner = gi >= gj;   // the work-item that visits the lower-triangle entry itself (not its mirror image)
      if (!owner) {
        const int tmp = gi;
        gi = gj;
        gj = tmp;
      }
      const int at = LY.at(gi, gj);
      v = W.Sg[at];
      {   // IMU factors (at most two per entry): gathered here from the accepted buffer's records
        const int2 sr = W.imu_rev[at];
        const double* lin = W.imu_lin[W.ctrl->acc];
        if (sr.x >= 0) v += lin[sr.x];
        if (sr.y >= 0) v += lin[sr.y];
      }
      if (gi < Dp) {  // both indices in the pose part: add the chunk partials (block-packed, row-major lower blocks)
        const int bi = gi / 6, bj = gj / 6;
        const size_t o = (size_t)(bi * (bi + 1) / 2 + bj) * 36 + (gi - 6 * bi) * 6 + (gj - 6 * bj);
        double a = 0;
#pragma unroll 8
        for (int ch = 0; ch < nch; ++ch) a += W.spart[(size_t)ch * stride + o];
        v += a;
      }
      if (owner && !W.S) {
        // leave the matrix zero for the next assembly (solve_kernel<true> then skips its own clearing pass); inside a
        // diagonal 6x6 block the mirrored entry as well.  (Not with the debug copy on: there the mirror images are read too.)
        W.Sg[at] = 0.0;
        if (gi / 6 == gj / 6 && gi != gj) W.Sg[LY.at(gj, gi)] = 0.0;
      }
    } else {
      v = (gi == gj) ? 1.0 : 0.0;
    }
    W.ct_T[(size_t)t * CT_TILE + r] = v;
    if (W.S && oi < D && oj < D) {  // parity/debug copy of the damped system (full symmetric)
      W.S[(size_t)oi * D + oj] = v;
      if (ti > tj) W.S[(size_t)oj * D + oi] = v;
    }
  }
}

__global__ __launch_bounds__(CT_THREADS) void chol_tiles_window_kernel(const WinPtrs* __restrict__ wins) {
  extern __shared__ __attribute__((aligned(16))) double ct_smem[];
  const WinPtrs& W = wins[blockIdx.y];
  const int nT = W.ct_nT, ntile = nT * (nT + 1) / 2;
  if (nT == 0 || (int)blockIdx.x >= ntile + nT) return;
  if (W.ct_flag[ntile + 1] != 1) return;  // nothing to factorise this iteration (terminated / final pass / explicit dogleg step)
  CholTiles C;
  C.nT = nT;
  C.T = W.ct_T;
  C.Linv = W.ct_Linv;
  C.rhs = W.ct_rhs;
  C.y = W.ct_y;
  C.flag = W.ct_flag;
  C.progress = W.ct_flag + ntile + 3;
  C.pflag = W.ct_flag + ntile + 4;
  C.x = W.ct_x;
  C.tl = (W.prof && blockIdx.y == 0) ? W.prof + 64 : nullptr;   // diagnostics (debug_arrays): task timeline behind the phase stamps
  chol_tile_task(C, blockIdx.x, ct_smem);
}

__global__ __launch_bounds__(SOLVE_THREADS) void solve_large_tail_kernel(const WinPtrs* __restrict__ wins,
                                                                        const OptD* __restrict__ optp) {
  const WinPtrs& W = wins[blockIdx.x];
  const int nT = W.ct_nT, ntile = nT * (nT + 1) / 2;
  if (nT == 0 || W.ct_flag[ntile + 1] == 0) return;
  const int tid = threadIdx.x;
  if (tid == 0) W.ct_flag[ntile + 2] = W.S ? 0 : 1;   // the export of this iteration left the HBM matrix cleared (see there)
  const OptD opt = *optp;
  Ctrl* gctrl = W.ctrl;
  __shared__ Ctrl c;
  __shared__ double s_x[CT_TB * ((MAX_D + CT_TB - 1) / CT_TB)];
  __shared__ double s_sc[4][SOLVE_THREADS / 64];
  if (tid == 0) c = *gctrl;
  __syncthreads();
  if (W.ct_flag[ntile + 1] == 2) {
    // explicit dogleg step of a large window (see solve_kernel<false>): the damped reduced matrix sits un-factorised in the
    // 48x48 tiles written by large_export_kernel
    __shared__ double s_sc5[5][SOLVE_THREADS / 64];
    __shared__ double s_dl[2];
    const int D = W.D, acc = c.acc;
    const double mu = c.mu;
    for (int i = tid; i < nT * CT_TB; i += SOLVE_THREADS) s_x[i] = i < D ? W.ct_g[i] / W.ct_d2[i] : 0.0;
    __syncthreads();
    double q1 = 0, q2 = 0, Ap = 0, Al = 0, Bl = 0;
    for (int k = tid; k < D * D; k += SOLVE_THREADS) {
      int i = k / D, j = k - i * D;
      const double xx = s_x[i] * s_x[j];
      if (i < j) {
        const int t = i;
        i = j;
        j = t;
      }
      const int ti = i / CT_TB, tj = j / CT_TB;
      q1 += W.ct_T[(size_t)(ti * (ti + 1) / 2 + tj) * CT_TILE + (i - CT_TB * ti) * CT_TB + (j - CT_TB * tj)] * xx;
    }
    for (int i = tid; i < D; i += SOLVE_THREADS) {
      q2 += W.ct_d2[i] * s_x[i] * s_x[i];
      Ap += W.ct_g[i] * s_x[i];
    }
    dl_landmark_sums(W, acc, s_x, mu, opt, tid, &Al, &Bl);
    q1 = wave_sum_full(q1);
    q2 = wave_sum_full(q2);
    Ap = wave_sum_full(Ap);
    Al = wave_sum_full(Al);
    Bl = wave_sum_full(Bl);
    if ((tid & 63) == 0) {
      s_sc5[0][tid >> 6] = q1;
      s_sc5[1][tid >> 6] = q2;
      s_sc5[2][tid >> 6] = Ap;
      s_sc5[3][tid >> 6] = Al;
      s_sc5[4][tid >> 6] = Bl;
    }
    __syncthreads();
    if (tid == 0) {
      double a[5] = {0, 0, 0, 0, 0};
      for (int k = 0; k < 5; ++k)
        for (int i = 0; i < SOLVE_THREADS / 64; ++i) a[k] += s_sc5[k][i];
      double cA, beta, dln, model;
      dogleg_coefficients(a[2] + a[3], (a[0] - mu * a[1]) + a[4], c.tot_C, c.tot_E, mu, c.radius, &cA, &beta, &dln, &model);
      s_dl[0] = cA;
      s_dl[1] = beta;
      c.cA = cA;
      c.beta = beta;
      c.dl_norm = dln;
      c.pend_model = model;
      c.tot_A = a[2] + a[3];
    }
    __syncthreads();
    {
      const double cA = s_dl[0], beta = s_dl[1];
      for (int i = tid; i < D; i += SOLVE_THREADS) s_x[i] = -cA * s_x[i] + beta * W.step[i];
    }
    __syncthreads();
    double s2 = 0, x2 = 0;
    trial_states(W, acc, s_x, tid, true, &s2, &x2);
    s2 = wave_sum_full(s2);
    x2 = wave_sum_full(x2);
    if ((tid & 63) == 0) {
      s_sc5[0][tid >> 6] = s2;
      s_sc5[1][tid >> 6] = x2;
    }
    __syncthreads();
    if (tid == 0) {
      double a0 = 0, a1 = 0;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
        a0 += s_sc5[0][i];
        a1 += s_sc5[1][i];
      }
      c.step2_p = a0;
      c.x2_p = a1;
      c.tr_kind = 1;
      c.spec_discard = c.explicit_next == 2;
      if (c.explicit_next == 1) c.iter++;
      c.explicit_next = 0;
      c.lambda = mu;
      c.pending = 1;
      *gctrl = c;
    }
    return;
  }
  if (W.ct_flag[ntile] || (opt.dogleg && c.mu >= DL_MAX_MU)) {  // not positive definite: invalid step (handled like a rejection)
    if (tid == 0) {
      c.chol_fail++;
      c.pending = 0;
      if (opt.dogleg) {
        solve_failed_dl(&c, opt);
      } else {
        c.iter++;
        c.radius = c.radius / c.decrease_factor;
        c.decrease_factor *= 2.0;
        if (c.radius < opt.min_radius) c.done = 5 + 1;
      }
      *gctrl = c;
    }
    return;
  }
  // the back-substitution ran as the last tasks of the tile kernel (chol_backsub_task)
  for (int i = tid; i < nT * CT_TB; i += SOLVE_THREADS) s_x[i] = W.ct_x[i];
  __syncthreads();
  const int D = W.D, acc = c.acc;
  const double lambda = opt.dogleg ? c.mu : 1.0 / c.radius;
  double gd = 0, ddd = 0, s2 = 0, x2 = 0;
  for (int i = tid; i < D; i += SOLVE_THREADS) {
    const double x = s_x[i];
    W.step[i] = x;
    gd += W.ct_g[i] * x;
    ddd += W.ct_d2[i] * x * x;
    s2 += x * x;
  }
  {
    double s2a = 0;
    trial_states(W, acc, s_x, tid, opt.dogleg != 0, &s2a, &x2);
    if (opt.dogleg) s2 = s2a;
  }
  gd = wave_sum(gd);
  ddd = wave_sum(ddd);
  s2 = wave_sum(s2);
  x2 = wave_sum(x2);
  if ((tid & 63) == 0) {
    s_sc[0][tid >> 6] = gd;
    s_sc[1][tid >> 6] = ddd;
    s_sc[2][tid >> 6] = s2;
    s_sc[3][tid >> 6] = x2;
  }
  __syncthreads();
  if (tid == 0) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < SOLVE_THREADS / 64; ++i) {
      a0 += s_sc[0][i];
      a1 += s_sc[1][i];
      a2 += s_sc[2][i];
      a3 += s_sc[3][i];
    }
    c.gd_p = a0;
    c.ddd_p = a1;
    c.step2_p = a2;
    c.x2_p = a3;
    c.lambda = lambda;
    c.tr_kind = 0;
    c.spec_discard = 0;
    c.explicit_next = 0;
    c.have_tot = 0;
    c.iter++;
    c.pending = 1;
    *gctrl = c;
  }
}

// iteration budget of the dogleg strategy: okvis_ba_iterate(n) allows n more iterations (launch slots spent on the redo
// of a mis-speculated Gauss-Newton trial do not count as iterations, so slots and iterations can differ)
__global__ void add_budget_kernel(const WinPtrs* __restrict__ wins, int n) {
  if (threadIdx.x == 0) wins[blockIdx.x].ctrl->max_iter += n;
}

// A run that stops right behind a decision which discarded a speculative Gauss-Newton evaluation (Ctrl::explicit_next == 2:
// okvis_ba_optimize_timed out of time, or the caller fetches between two okvis_ba_iterate calls) has no further evaluation to take
// the IMU terms' preintegrations back (ba_imu.hpp, imu_factor): the record the speculative evaluation re-preintegrated would travel
// to the next frame with okvis_ba_fetch_imu_caches, one the reference never had.  This kernel does what that evaluation would have
// done — record and reference bias of every such term back from imu_cache_prev — before results are handed out.  Idempotent: a
// later evaluation finds nothing left to take back.  grid (max_imu, windows).
__global__ void imu_take_back_kernel(const WinPtrs* __restrict__ wins) {
  const WinPtrs& W = wins[blockIdx.y];
  const int f = blockIdx.x;
  if (f >= W.n_imu || W.ctrl->explicit_next != 2) return;
  auto cg = W.imu_cache + f;
  auto cp = W.imu_cache_prev + f;
  if (!cp->valid) return;
  auto src = reinterpret_cast<const BA_G double*>(cp);
  auto dst = reinterpret_cast<BA_G double*>(cg);
  for (int i = threadIdx.x; i < (int)(sizeof(ImuCacheD) / 8); i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) cp->valid = 0;
}

// the results of one window in one contiguous record for okvis_ba_fetch_results: pose[7 n_pose] | sb[9 n_sb] | lm[4 n_lm] |
// quality[n_lm] | reference bias of every IMU factor [9 n_imu]
__global__ void pack_results_kernel(const WinPtrs* __restrict__ win, int acc) {
  const WinPtrs& W = *win;
  if (acc < 0) acc = W.ctrl->acc;   // (okvis_ba_finish: the host has not read the control record yet)
  const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  double* out = W.results;
  for (int i = t; i < 7 * W.n_pose; i += nt) out[i] = W.pose[acc][i];
  out += 7 * (size_t)W.n_pose;
  for (int i = t; i < 9 * W.n_sb; i += nt) out[i] = W.sb[acc][i];
  out += 9 * (size_t)W.n_sb;
  for (int i = t; i < 4 * W.n_lm; i += nt) out[i] = W.lm[acc][i];
  out += 4 * (size_t)W.n_lm;
  for (int i = t; i < W.n_lm; i += nt) out[i] = W.quality[i];
  out += W.n_lm;
  for (int i = t; i < 9 * W.n_imu; i += nt) out[i] = W.imu_cache[i / 9].sb_ref[i % 9];
  out += 9 * (size_t)W.n_imu;
  {   // the preintegrations themselves (the records are whole doubles: Delta_q .. sb_ref, then the two flag words as one)
    constexpr int CD = (int)(sizeof(ImuCacheD) / 8);
    const double* src = reinterpret_cast<const double*>(W.imu_cache);
    for (int i = t; i < CD * W.n_imu; i += nt) out[i] = src[i];
  }
}

// landmark quality (Estimator.cpp:880-896): 3x3 eigenvalues of the un-robustified H_l of the accepted
// linearisation; quality = 0 if lambda_min < 1e-12 else sqrt(lambda_min)/sqrt(lambda_max).
__global__ void quality_kernel(const WinPtrs* __restrict__ wins) {
  const WinPtrs& W = wins[blockIdx.y];
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= W.n_lm) return;
  const double* h = W.Hq[W.ctrl->acc] + 6 * (size_t)l;
  const double v[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};
  double emin, emax;
  eig3sym_minmax(v, &emin, &emax);
  W.quality[l] = (emin < 1.0e-12) ? 0.0 : sqrt(emin) / sqrt(emax);
}

}  // namespace ba
