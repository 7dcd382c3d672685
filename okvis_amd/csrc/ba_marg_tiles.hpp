// The dense tail of the marginalisation (ba_marg.hpp) on MANY workgroups, for kept blocks that do not fit the single-workgroup
// LDS paths (more than MARG_PC_NMAX = 96 rows: a window of 20 or 50 frames, BASELINE configs[2]).  Same algebra, same proofs:
//
//   marginalizeOut, dense part      okvis_ceres/src/MarginalizationError.cpp:686-736
//   updateErrorComputation          okvis_ceres/src/MarginalizationError.cpp:806-846
//
//   marg_prior_add_kernel        previous prior into H and b, one entry per work-item
//   marg_dense_kernel(stage 7)   one workgroup: index lists, the eliminated block's V^(+1/2) (small: one frame's pose and
//                                speed/bias) — and stops
//   marg_M_kernel, marg_b0_kernel   M = W V^(+1/2), b0 = P (b_a - M V^(+1/2)^T b_b), one entry per work-item
//   marg_schur_kernel            H_a = P (U - M M^T) P, one entry per work-item                       (:736-738)
//   marg_tiles_fill_kernel       the pre-scaled, symmetrised kept block as 48 x 48 lower tiles (identity padded), rhs = P^-1 b0
//   chol_tile_kernel             ba_chol_tiles.hpp: L (fp64 matrix core), the inverses of the diagonal tiles, y = L^-1 rhs
//   marg_tiles_inverse_kernel    L^-T tile by tile (one workgroup each, the task graph of the factorisation, matrix core): ||L^-1||_F^2
//   marg_tiles_out_kernel        J = L^T P, e0 = -y
//   marg_tiles_rowsum_kernel     sum_j |A_ij|, one wave per row
//   marg_tiles_decide_kernel     the proof of full rank, as in marg_chol_inverse:  1 / ||L^-1||_F^2 > 4 eps n max_i sum_j |A_ij|
//
// When the proof fails (a rank-deficient kept block, a non-positive pivot) nothing of this is used: the caller runs the
// single-workgroup kernel, which goes on to the eigen-decomposition.
#pragma once
#include "ba_chol_tiles.hpp"
#include "ba_marg.hpp"

namespace ba {

constexpr int MARG_TILES_THREADS = 256;

struct MargTiles {
  CholTiles C;          // tiles, diagonal inverses, rhs, y, flags (x = nullptr: no back-substitution)
  double* Z;            // [ntiles] tiles of L^-T: tile (i, j) of L^-1, transposed, at ct_tile_index(i, j)
  int* zflag;           // [ntiles] tile of Z published
  double* fro;          // [ntiles] sum of the squares of each tile of L^-1 over the true rows / columns
  double* p2;           // [48 nT] scaling of the kept block
  double* rowsum;       // [48 nT] sum_j |A_ij| of the pre-scaled matrix
  int* ok;              // [0] 1 = the factor and the proof hold: J, e0 are final
};

// previous prior into the exported system (H_, b0_ of the reference's MarginalizationError): one entry per work-item
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_prior_add_kernel(const WinPtrs* __restrict__ wins, MargArgs a) {
  const WinPtrs& W = wins[0];
  const int D = W.D, pd = a.prior_dim;
  const size_t k = (size_t)blockIdx.x * MARG_TILES_THREADS + threadIdx.x;
  if (k >= (size_t)pd * pd) return;
  const int rr = (int)(k / pd), cc = (int)(k - (size_t)rr * pd);
  auto reduced = [&](int row) {   // row of the prior -> index in the reduced system (-1: a fixed block)
    int bi = 0;
    for (int q = 0; q < a.prior_nb; ++q)
      if (a.pb_off[q] <= row) bi = q;
    const int base = a.pb_type[bi] == 0 ? W.pose_off[a.pb_idx[bi]] : W.sb_off[a.pb_idx[bi]];
    return base < 0 ? -1 : base + (row - a.pb_off[bi]);
  };
  const int ri = reduced(rr), ci = reduced(cc);
  if (ri >= 0 && ci >= 0) W.S[(size_t)ri * D + ci] += a.prior_H[k];
  if (cc == 0 && ri >= 0) W.rhs[ri] += a.prior_b0[rr];
}

// M = W V^(+1/2)  (:729), one entry per work-item: the expression of marg_dense_kernel (its column gathers from H are what made the
// single workgroup slow)
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_M_kernel(const WinPtrs* __restrict__ wins, MargArgs a) {
  const WinPtrs& W = wins[0];
  const int D = W.D, na = a.out_info[0], nm = a.out_info[1];
  const double* H = W.S;
  const double* Q = a.work + (size_t)D * D;
  double* M = a.work + 2 * (size_t)D * D;
  const int* kidx = a.out_info + 8;
  const int* midx = a.out_info + 8 + na;
  const size_t k = (size_t)blockIdx.x * MARG_TILES_THREADS + threadIdx.x;
  if (k >= (size_t)na * nm) return;
  const int i = (int)(k / nm), j = (int)(k - (size_t)i * nm);
  const int ki = kidx[i];
  double s = 0;
  for (int c = 0; c < nm; ++c) s += H[(size_t)ki * D + midx[c]] / (a.p_out[ki] * a.p_out[midx[c]]) * Q[(size_t)c * nm + j];
  M[k] = s;
}

// b0 = P_a (b_a - M V^(+1/2)^T b_b)  (:732, :739); every workgroup forms V^(+1/2)^T b_b for itself
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_b0_kernel(const WinPtrs* __restrict__ wins, MargArgs a) {
  const WinPtrs& W = wins[0];
  const int D = W.D, na = a.out_info[0], nm = a.out_info[1];
  const double* Q = a.work + (size_t)D * D;
  const double* M = a.work + 2 * (size_t)D * D;
  const double* b = W.rhs;
  const int* kidx = a.out_info + 8;
  const int* midx = a.out_info + 8 + na;
  __shared__ double s_t[MAX_D];
  for (int j = threadIdx.x; j < nm; j += MARG_TILES_THREADS) {
    double s = 0;
    for (int c = 0; c < nm; ++c) s += Q[(size_t)c * nm + j] * (b[midx[c]] / a.p_out[midx[c]]);
    s_t[j] = s;
  }
  __syncthreads();
  const int i = blockIdx.x * MARG_TILES_THREADS + threadIdx.x;
  if (i >= na) return;
  const int ki = kidx[i];
  double s = b[ki] / a.p_out[ki];
  for (int c = 0; c < nm; ++c) s -= M[(size_t)i * nm + c] * s_t[c];
  a.out_b0[i] = s * a.p_out[ki];
}

// H_a = P_a (U - M M^T) P_a : the expression of marg_dense_kernel, entry by entry
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_schur_kernel(const WinPtrs* __restrict__ wins, MargArgs a) {
  const WinPtrs& W = wins[0];
  const int D = W.D, na = a.out_info[0], nm = a.out_info[1];
  const double* H = W.S;
  const double* M = a.work + 2 * (size_t)D * D;
  const int* kidx = a.out_info + 8;
  const size_t k = (size_t)blockIdx.x * MARG_TILES_THREADS + threadIdx.x;
  if (k >= (size_t)na * na) return;
  const int i = (int)(k / na), j = (int)(k - (size_t)i * na);
  const int ki = kidx[i], kj = kidx[j];
  if (nm == 0) {
    a.out_H[k] = H[(size_t)ki * D + kj];
    return;
  }
  const double pp = a.p_out[ki] * a.p_out[kj];
  double s = H[(size_t)ki * D + kj] / pp;
  for (int c = 0; c < nm; ++c) s -= M[(size_t)i * nm + c] * M[(size_t)j * nm + c];
  a.out_H[k] = s * pp;
}

// scaling of updateErrorComputation (:812-815) — needs the finished diagonal of H_a, hence a launch of its own
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_tiles_scale_kernel(MargArgs a, MargTiles T) {
  const int na = a.out_info[0];
  const int i = blockIdx.x * MARG_TILES_THREADS + threadIdx.x;
  if (i >= CT_TB * T.C.nT) return;
  double p = 1.0;
  if (i < na) {
    const double d = a.out_H[(size_t)i * na + i];
    p = d > 1.0e-9 ? sqrt(d) : 1.0e-3;
  }
  T.p2[i] = p;
  T.C.rhs[i] = i < na ? a.out_b0[i] / p : 0.0;
}

// one workgroup per lower tile: A = 0.5 (H_a + H_a^T) pre-scaled, identity outside the true part; the flags of the factorisation start at zero
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_tiles_fill_kernel(MargArgs a, MargTiles T) {
  const int na = a.out_info[0], nT = T.C.nT;
  int j = 0, rem = blockIdx.x;   // column-major task order, like chol_tile_task
  while (rem >= nT - j) {
    rem -= nT - j;
    ++j;
  }
  const int i = j + rem;
  double* t = T.C.T + (size_t)ct_tile_index(i, j) * CT_TILE;
  const double* Ha = a.out_H;
  for (int e = threadIdx.x; e < CT_TILE; e += MARG_TILES_THREADS) {
    const int r = CT_TB * i + e / CT_TB, c = CT_TB * j + e % CT_TB;
    double v = r == c ? 1.0 : 0.0;
    if (r < na && c < na) v = 0.5 * (Ha[(size_t)r * na + c] + Ha[(size_t)c * na + r]) / (T.p2[r] * T.p2[c]);
    t[e] = v;
  }
  if (blockIdx.x == 0) {
    const int nflag = nT * (nT + 1) / 2 + 1 + 2 * nT;
    for (int e = threadIdx.x; e < nflag; e += MARG_TILES_THREADS) T.C.flag[e] = 0;
    for (int e = threadIdx.x; e < nT * (nT + 1) / 2; e += MARG_TILES_THREADS) T.zflag[e] = 0;
    if (threadIdx.x == 0) T.ok[0] = 0;
  }
}

// L^-1 by forward substitution over tiles, kept TRANSPOSED (Z_ji = (L^-1)_ij^T) so that every product is the A B^T form of the
// factorisation (ct_gemm_nt):
//   Z_jj = Linv_j^T,     Z_ji = - (sum_(k = j .. i-1) Z_jk L_ik^T) Linv_i^T        (from  sum_k L_ik X_kj = delta_ij)
// One workgroup per tile (i, j), i >= j, in the column-major task order of the factorisation: tile (i, j) adds the products in
// the order k = j, j + 1, ... as the tiles Z_jk above it are published (flags; every dependency has a smaller task index, so it
// is resident or finished when a workgroup starts to wait) — the chain of a tile column is one product + one hand-over per tile
// instead of i - j products (a workgroup per column: 305 us at nT = 15).  fro[task] = the sum of the squares over the true part
// (the identity padding contributes nothing to the bound).
__global__ __launch_bounds__(CT_THREADS) void marg_tiles_inverse_kernel(MargArgs a, MargTiles T) {
  extern __shared__ __attribute__((aligned(16))) double mt_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int na = a.out_info[0], nT = T.C.nT;
  int j = 0, rem = blockIdx.x;
  while (rem >= nT - j) {
    rem -= nT - j;
    ++j;
  }
  const int i = j + rem;
  double* sA = mt_smem;                      // Z_jk, then the sum
  double* sB = mt_smem + CT_TB * CT_LD;      // L_ik, then Linv_i
  __shared__ double s_part[CT_THREADS / 64];
  __shared__ int s_ok;
  int* failflag = T.C.flag + nT * (nT + 1) / 2;
  double fs = 0.0;
  if (i == j) {   // Z_jj = Linv_j^T
    const double* Lj = T.C.Linv + (size_t)j * CT_TILE;
    double* Zjj = T.Z + (size_t)ct_tile_index(j, j) * CT_TILE;
    for (int e = tid; e < CT_TILE; e += CT_THREADS) {
      const int r = e / CT_TB, c = e - r * CT_TB;
      const double v = ct_gld(Lj + c * CT_TB + r);
      ct_gst(Zjj + e, v);
      if (CT_TB * j + r < na && CT_TB * j + c < na) fs += v * v;
    }
  } else {
    if (tid == 0) s_ok = 1;
    ct_v4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = ct_v4{0.0, 0.0, 0.0, 0.0};
    for (int k = j; k < i; ++k) {
      ct_load_tile(T.C.T + (size_t)ct_tile_index(i, k) * CT_TILE, sB, tid);    // L_ik: there since the factorisation
      if (tid == 0 && !ct_wait(T.zflag + ct_tile_index(k, j))) s_ok = 0;        // Z_jk (stored at the place of tile (k, j))
      __syncthreads();
      if (!s_ok) break;
      ct_load_tile(T.Z + (size_t)ct_tile_index(k, j) * CT_TILE, sA, tid);
      __syncthreads();
      if (wave < 3) ct_gemm_nt(acc, sA, sB, wave, lane, 1.0);
      __syncthreads();
    }
    if (s_ok) {
      if (wave < 3) ct_store_acc(acc, sA, CT_LD, wave, lane);
      ct_load_tile(T.C.Linv + (size_t)i * CT_TILE, sB, tid);
      __syncthreads();
      if (wave < 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = ct_v4{0.0, 0.0, 0.0, 0.0};
        ct_gemm_nt(acc, sA, sB, wave, lane, -1.0);
        const int col = lane & 15, r0 = lane >> 4;   // accumulator of Z_(j, i): rows 48 j + .., columns 48 i + ..
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int gr = CT_TB * j + 16 * wave + r0 + 4 * q, gc = CT_TB * i + 16 * c + col;
            if (gr < na && gc < na) fs += acc[c][q] * acc[c][q];
          }
        ct_store_acc_g(acc, T.Z + (size_t)ct_tile_index(i, j) * CT_TILE, wave, lane);
      }
    } else if (tid == 0) {
      ct_raise(failflag);   // a tile above never came: no proof (the dependants below run out through their own waits)
    }
  }
  ct_release();
  fs = wave_sum(fs);
  if (lane == 0) s_part[wave] = fs;
  __syncthreads();
  if (tid == 0) {
    double f = 0.0;
    for (int q = 0; q < CT_THREADS / 64; ++q) f += s_part[q];
    T.fro[blockIdx.x] = f;
    ct_raise(T.zflag + ct_tile_index(i, j));
  }
}

// J = L^T P (upper triangular), e0 = -L^-1 P^-1 b0 = -y.  The diagonal tiles of L are not kept by the factorisation (it keeps
// their inverses): L_jj = (Linv_j)^-1 by substitution, one column per work-item.
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_tiles_out_kernel(MargArgs a, MargTiles T) {
  const int na = a.out_info[0], nT = T.C.nT;
  int j = 0, rem = blockIdx.x;
  while (rem >= nT - j) {
    rem -= nT - j;
    ++j;
  }
  const int i = j + rem;   // tile (i, j) of L, i >= j  ->  block (j, i) of J
  __shared__ double s_L[CT_TB * CT_LD];
  const int tid = threadIdx.x;
  if (i == j) {
    __shared__ double s_X[CT_TB * CT_LD];
    const double* Lj = T.C.Linv + (size_t)j * CT_TILE;
    for (int e = tid; e < CT_TILE; e += MARG_TILES_THREADS) s_X[(e / CT_TB) * CT_LD + e % CT_TB] = ct_gld(Lj + e);
    __syncthreads();
    if (tid < CT_TB) {   // column c of L = X^-1:  L_cc = 1 / X_cc,  L_rc = -(sum_(m = c .. r-1) X_rm L_mc) / X_rr
      const int c = tid;
      for (int r = 0; r < CT_TB; ++r) {
        double v = 0.0;
        if (r == c) {
          v = 1.0 / s_X[r * CT_LD + r];
        } else if (r > c) {
          double sacc = 0.0;
          for (int m = c; m < r; ++m) sacc += s_X[r * CT_LD + m] * s_L[m * CT_LD + c];
          v = -sacc / s_X[r * CT_LD + r];
        }
        s_L[r * CT_LD + c] = v;
      }
    }
    __syncthreads();
    if (tid < CT_TB && CT_TB * j + tid < na) a.out_e0[CT_TB * j + tid] = -ct_gld(T.C.y + CT_TB * j + tid);
  } else {
    const double* Lij = T.C.T + (size_t)ct_tile_index(i, j) * CT_TILE;
    for (int e = tid; e < CT_TILE; e += MARG_TILES_THREADS) s_L[(e / CT_TB) * CT_LD + e % CT_TB] = ct_gld(Lij + e);
    __syncthreads();
  }
  // J[r][c] = L[c][r] p_c for c >= r; this tile supplies rows 48 j + .. and columns 48 i + ..; the strictly lower part of J is zero
  for (int e = tid; e < CT_TILE; e += MARG_TILES_THREADS) {
    const int rr = e / CT_TB, cc = e - rr * CT_TB;       // J row 48 j + rr, column 48 i + cc  <-  L[48 i + cc][48 j + rr]
    const int gr = CT_TB * j + rr, gc = CT_TB * i + cc;
    if (gr < na && gc < na) a.out_J[(size_t)gr * na + gc] = gc >= gr ? s_L[cc * CT_LD + rr] * T.p2[gc] : 0.0;
    if (i != j && gr < na && gc < na) a.out_J[(size_t)gc * na + gr] = 0.0;   // the mirrored block, below the diagonal
  }
}

// absolute row sums of the pre-scaled matrix, one wave per row (for the lambda_max bound)
__global__ __launch_bounds__(MARG_TILES_THREADS) void marg_tiles_rowsum_kernel(MargArgs a, MargTiles T) {
  const int na = a.out_info[0];
  const int lane = threadIdx.x & 63, r = blockIdx.x * (MARG_TILES_THREADS / 64) + (threadIdx.x >> 6);
  if (r >= na) return;
  double v = 0.0;
  for (int c = lane; c < na; c += 64)
    v += fabs(0.5 * (a.out_H[(size_t)r * na + c] + a.out_H[(size_t)c * na + r]) / (T.p2[r] * T.p2[c]));
  v = wave_sum(v);
  if (lane == 0) T.rowsum[r] = v;
}

// lambda_max bound (max absolute row sum of the pre-scaled matrix) and the decision
__global__ __launch_bounds__(MARG_THREADS) void marg_tiles_decide_kernel(MargArgs a, MargTiles T) {
  const int tid = threadIdx.x;
  const int na = a.out_info[0], nT = T.C.nT;
  __shared__ double s_red[MARG_THREADS / 64];
  double rs = 0.0;
  for (int r = tid; r < na; r += MARG_THREADS) rs = fmax(rs, T.rowsum[r]);
  rs = wave_max(rs);
  if ((tid & 63) == 0) s_red[tid >> 6] = rs;
  __syncthreads();
  if (tid == 0) {
    double rowmax = 0.0, f = 0.0;
    for (int q = 0; q < MARG_THREADS / 64; ++q) rowmax = fmax(rowmax, s_red[q]);
    for (int t = 0; t < nT * (nT + 1) / 2; ++t) f += T.fro[t];
    const int failed = T.C.flag[nT * (nT + 1) / 2];
    const bool ok = !failed && f > 0.0 && 1.0 / f > 4.0 * 2.220446049250313e-16 * na * rowmax;
    T.ok[0] = ok ? 1 : 0;
    if (ok) {
      a.out_info[2] = na;
      a.out_info[4] = 0;
    }
  }
}

}  // namespace ba
