// Dense Cholesky solve of the reduced camera system for windows too large for the single-workgroup LDS
// solver (D > MAX_D_LDS, e.g. BASELINE configs[2]: 50 keyframes, D = 750): a tiled, multi-workgroup,
// left-looking factorisation whose tile updates are fp64 MFMA GEMMs (v_mfma_f64_16x16x4_f64).
//
//   tiles      48 x 48 (8 blocks of the 6-wide pose/speed-bias granularity, 3 MFMA sub-tiles of 16), row-major,
//              lower triangle only, tile (i,j) at ((i(i+1)/2 + j) * 48 * 48
//   one workgroup per tile, launched in COLUMN-major task order.  Tile (i,j) needs tiles (i,k), (j,k), k < j,
//   and the diagonal tile (j,j): all of them have a smaller task index.  Workgroups are dispatched in index
//   order, so every dependency is resident or finished when a workgroup starts spinning on its flag: no
//   host synchronisation, no cooperative launch, no deadlock.  (Spins are bounded anyway: a stuck
//   dependency marks the factorisation as failed instead of hanging the GPU.)
//
//     off-diagonal (i,j):  C = A_ij - sum_k L_ik L_jk^T  (MFMA);  L_ij = C L_jj^-T = C (Linv_j)^T  (MFMA)
//     diagonal (j,j):      C = A_jj - sum_k L_jk L_jk^T  (MFMA);  L_jj = chol(C) in LDS (6-wide blocks),
//                          Linv_j = L_jj^-1 (published for the column's TRSMs and for the back-substitution),
//                          y_j = Linv_j (rhs_j - sum_k L_jk y_k)  (forward substitution rides along)
//   back-substitution L^T x = y: chol_backsub(), one workgroup, tile column by tile column with Linv_j^T.
#pragma once
#include "ba_device.hpp"
#include "ba_types.hpp"

namespace ba {

constexpr int CT_TB = 48;            // tile edge
constexpr int CT_LD = 49;            // LDS row stride (doubles): conflict-free column walks
constexpr int CT_THREADS = 256;
constexpr int CT_TILE = CT_TB * CT_TB;
constexpr int CT_SPIN_LIMIT = 1 << 22;

typedef double ct_v4 __attribute__((ext_vector_type(4)));

struct CholTiles {   // per-window workspace of the tiled solver (device pointers)
  int nT;            // tile rows/columns; padded dimension = 48 nT
  double* T;         // lower tiles, A on entry, L on exit
  double* Linv;      // [nT] inverse of the diagonal tiles
  double* rhs;       // [48 nT] right-hand side on entry
  double* y;         // [48 nT] L^-1 rhs
  int* flag;         // [nT(nT+1)/2 + 1]: tile done flags; last entry = failure (non-PD pivot / dependency timeout)
};

__device__ __forceinline__ int ct_tile_index(int i, int j) { return i * (i + 1) / 2 + j; }

// wait until *f != 0 (acquire); returns false on timeout
__device__ __forceinline__ bool ct_wait(const int* f) {
  for (int it = 0; it < CT_SPIN_LIMIT; ++it) {
    if (__atomic_load_n(f, __ATOMIC_ACQUIRE) != 0) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

// global tile (row-major 48x48) -> LDS (stride CT_LD)
__device__ __forceinline__ void ct_load_tile(const double* g, double* l, int tid) {
  for (int e = tid; e < CT_TILE; e += CT_THREADS) l[(e / CT_TB) * CT_LD + (e % CT_TB)] = g[e];
}

// acc(strip r of 16 rows, 3 sub-tiles of 16 columns) += sign * A[16r.., :] * B^T   with A, B 48x48 in LDS.
// v_mfma_f64_16x16x4_f64: A operand lane l = A[row l&15][k l>>4], B operand lane l = B[k l>>4][col l&15],
// C/D lane l, register i = C[row (l>>4) + 4 i][col l&15]  (cdna_hip_programming.md "Fragment layout")
__device__ __forceinline__ void ct_gemm_nt(ct_v4 acc[3], const double* A, const double* B, int strip, int lane, double sign) {
  const int rr = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < CT_TB; k0 += 4) {
    const double a = sign * A[(16 * strip + rr) * CT_LD + k0 + kk];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double b = B[(16 * c + rr) * CT_LD + k0 + kk];   // B^T[k][col] = B[col][k]
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
  }
}

// accumulator strip -> LDS / global (row-major with leading dimension ld)
__device__ __forceinline__ void ct_store_acc(const ct_v4 acc[3], double* dst, int ld, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(16 * strip + r0 + 4 * i) * ld + 16 * c + col] = acc[c][i];
}
__device__ __forceinline__ void ct_load_acc(ct_v4 acc[3], const double* src, int ld, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = src[(16 * strip + r0 + 4 * i) * ld + 16 * c + col];
}

__device__ __forceinline__ double ct_rsqrt(double x) {  // v_rsq_f64 + two Newton steps: full fp64 accuracy
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// Cholesky of the 48x48 SPD matrix in LDS (stride CT_LD, lower triangle referenced) in 6-wide block columns.
// The 6x6 diagonal block is factored in registers by one work-item (reciprocal square roots instead of
// divisions), its reciprocal diagonal is published in dinv[48] for the panel step.  A non-positive pivot sets
// *s_fail.  Afterwards the lower triangle holds L.
__device__ void ct_potrf48(double* M, double* dinv, int tid, int* s_fail) {
  for (int kb = 0; kb < 8; ++kb) {
    const int k0 = 6 * kb;
    if (tid == 0) {
      double a[6][6];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) a[r][c] = M[(k0 + r) * CT_LD + k0 + c];
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double d = a[c][c];
#pragma unroll
        for (int m = 0; m < c; ++m) d -= a[c][m] * a[c][m];
        if (!(d > 0.0)) {
          bad = true;
          d = 1.0;
        }
        const double inv = ct_rsqrt(d);
        a[c][c] = d * inv;
        dinv[k0 + c] = inv;
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
          double v = a[r][c];
#pragma unroll
          for (int m = 0; m < c; ++m) v -= a[r][m] * a[c][m];
          a[r][c] = v * inv;
        }
      }
      if (bad) *s_fail = 1;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) M[(k0 + r) * CT_LD + k0 + c] = a[r][c];
    }
    __syncthreads();
    const int nrows = CT_TB - k0 - 6;
    if (tid < nrows) {  // panel: row <- row L_kk^-T
      double* row = M + (k0 + 6 + tid) * CT_LD + k0;
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double v = row[c];
#pragma unroll
        for (int m = 0; m < c; ++m) v -= x[m] * M[(k0 + c) * CT_LD + k0 + m];
        x[c] = v * dinv[k0 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) row[c] = x[c];
    }
    __syncthreads();
    // trailing update, one work-item per entry of the lower triangle
    const int ntri = nrows * (nrows + 1) / 2;
    for (int e = tid; e < ntri; e += CT_THREADS) {
      int r = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
      while ((r + 1) * (r + 2) / 2 <= e) ++r;
      while (r * (r + 1) / 2 > e) --r;
      const int c = e - r * (r + 1) / 2;
      const double* a = M + (k0 + 6 + r) * CT_LD + k0;
      const double* b = M + (k0 + 6 + c) * CT_LD + k0;
      double s = 0;
#pragma unroll
      for (int m = 0; m < 6; ++m) s += a[m] * b[m];
      M[(k0 + 6 + r) * CT_LD + k0 + 6 + c] -= s;
    }
    __syncthreads();
  }
}

// X = L^-1 for the lower-triangular L in LDS (stride CT_LD), blocked by 6: the eight diagonal blocks are
// inverted by eight work-items, then block row by block row  X_(bi,bj) = -X_(bi,bi) sum_(k=bj..bi-1) L_(bi,k) X_(k,bj).
// X (stride CT_LD, full square, zeros above the diagonal); Tm = 6 x 48 scratch.
__device__ void ct_trinv48(const double* L, const double* dinv, double* X, double* Tm, int tid) {
  for (int e = tid; e < CT_TB * CT_TB; e += CT_THREADS) X[(e / CT_TB) * CT_LD + (e % CT_TB)] = 0.0;
  __syncthreads();
  if (tid < 8) {  // inverse of a 6x6 lower-triangular block, column by column
    const int k0 = 6 * tid;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double x[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if (r < c) {
          x[r] = 0.0;
          continue;
        }
        double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < r; ++m)
          if (m >= c) v -= L[(k0 + r) * CT_LD + k0 + m] * x[m];
        x[r] = v * dinv[k0 + r];
      }
#pragma unroll
      for (int r = c; r < 6; ++r) X[(k0 + r) * CT_LD + k0 + c] = x[r];
    }
  }
  __syncthreads();
  for (int bi = 1; bi < 8; ++bi) {
    const int ncol = 6 * bi;  // columns 0 .. 6 bi - 1 of block row bi
    for (int e = tid; e < 6 * ncol; e += CT_THREADS) {  // T = sum_k L_(bi,k) X_(k,.)
      const int r = e / ncol, c = e - r * ncol;
      const int m0 = (c / 6) * 6;  // X is lower triangular: rows >= the column's block start
      double s = 0;
      for (int m = m0; m < ncol; ++m) s += L[(6 * bi + r) * CT_LD + m] * X[m * CT_LD + c];
      Tm[r * CT_TB + c] = s;
    }
    __syncthreads();
    for (int e = tid; e < 6 * ncol; e += CT_THREADS) {  // X_(bi,.) = -X_(bi,bi) T
      const int r = e / ncol, c = e - r * ncol;
      double s = 0;
#pragma unroll
      for (int m = 0; m < 6; ++m)
        if (m <= r) s += X[(6 * bi + r) * CT_LD + 6 * bi + m] * Tm[m * CT_TB + c];
      X[(6 * bi + r) * CT_LD + c] = -s;
    }
    __syncthreads();
  }
}

// one workgroup per lower tile, blockIdx.x in column-major task order
__device__ void chol_tile_task(const CholTiles& C, int task, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nT = C.nT;
  int j = 0, rem = task;
  while (rem >= nT - j) {
    rem -= nT - j;
    ++j;
  }
  const int i = j + rem;
  double* sA = lds;                       // operand / work tiles
  double* sB = lds + CT_TB * CT_LD;
  double* sC = lds + 2 * CT_TB * CT_LD;
  __shared__ int s_ok, s_fail;
  __shared__ double s_r[CT_TB], s_dinv[CT_TB], s_tm[6 * CT_TB];
  int* failflag = C.flag + nT * (nT + 1) / 2;
  if (tid == 0) {
    s_ok = 1;
    s_fail = 0;
  }
  const bool diag = (i == j);
  ct_v4 acc[3];
  double* Tij = C.T + (size_t)ct_tile_index(i, j) * CT_TILE;
  if (wave < 3) ct_load_acc(acc, Tij, CT_TB, wave, lane);
  if (diag && tid < CT_TB) s_r[tid] = C.rhs[CT_TB * j + tid];
  __syncthreads();
  for (int k = 0; k < j; ++k) {
    if (tid == 0) {
      bool ok = ct_wait(C.flag + ct_tile_index(i, k));
      if (ok && !diag) ok = ct_wait(C.flag + ct_tile_index(j, k));
      if (!ok) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) break;
    ct_load_tile(C.T + (size_t)ct_tile_index(i, k) * CT_TILE, sA, tid);
    if (!diag) ct_load_tile(C.T + (size_t)ct_tile_index(j, k) * CT_TILE, sB, tid);
    __syncthreads();
    if (wave < 3) ct_gemm_nt(acc, sA, diag ? sA : sB, wave, lane, -1.0);
    if (diag && tid >= 192 && tid < 192 + CT_TB) {  // forward substitution rides along: r_j -= L_jk y_k
      const int r = tid - 192;
      const double* yk = C.y + CT_TB * k;
      double s = 0;
      for (int m = 0; m < CT_TB; ++m) s += sA[r * CT_LD + m] * yk[m];
      s_r[r] -= s;
    }
    __syncthreads();
  }
  if (!s_ok) {
    if (tid == 0) {
      __atomic_store_n(failflag, 1, __ATOMIC_RELEASE);
      __atomic_store_n(C.flag + ct_tile_index(i, j), 1, __ATOMIC_RELEASE);  // let the dependants run out
    }
    return;
  }
  if (diag) {
    if (wave < 3) ct_store_acc(acc, sC, CT_LD, wave, lane);
    __syncthreads();
    ct_potrf48(sC, s_dinv, tid, &s_fail);
    ct_trinv48(sC, s_dinv, sB, s_tm, tid);
    // publish L_jj (lower, zeros above), Linv_j and y_j = Linv_j r_j
    double* Linv = C.Linv + (size_t)j * CT_TILE;
    for (int e = tid; e < CT_TILE; e += CT_THREADS) {
      const int r = e / CT_TB, c = e - r * CT_TB;
      Tij[e] = (c <= r) ? sC[r * CT_LD + c] : 0.0;
      Linv[e] = sB[r * CT_LD + c];
    }
    if (tid < CT_TB) {
      double s = 0;
      for (int m = 0; m <= tid; ++m) s += sB[tid * CT_LD + m] * s_r[m];
      C.y[CT_TB * j + tid] = s;
    }
    if (tid == 0 && s_fail) __atomic_store_n(failflag, 1, __ATOMIC_RELEASE);
  } else {
    // L_ij = C Linv_j^T
    if (wave < 3) ct_store_acc(acc, sA, CT_LD, wave, lane);
    if (tid == 0 && !ct_wait(C.flag + ct_tile_index(j, j))) s_ok = 0;
    __syncthreads();
    ct_load_tile(C.Linv + (size_t)j * CT_TILE, sB, tid);
    __syncthreads();
    if (wave < 3) {
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = ct_v4{0.0, 0.0, 0.0, 0.0};
      ct_gemm_nt(acc, sA, sB, wave, lane, 1.0);
      ct_store_acc(acc, Tij, CT_TB, wave, lane);
    }
    if (tid == 0 && !s_ok) __atomic_store_n(failflag, 1, __ATOMIC_RELEASE);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) __atomic_store_n(C.flag + ct_tile_index(i, j), 1, __ATOMIC_RELEASE);
}

// back-substitution L^T x = y by one workgroup (any size): x_j = Linv_j^T (y_j - sum_(i>j) L_ij^T x_i).
// x (LDS or global, 48 nT doubles) may alias nothing else; scratch needs 48 doubles of LDS.
__device__ void chol_backsub(const CholTiles& C, double* x, double* scratch, int tid, int nthreads) {
  const int nT = C.nT;
  // scratch: [48] t, then [nparts][48] partial sums
  const int nparts = nthreads / CT_TB;  // work-items (part, component): the rows of the column below are split into parts
  double* part = scratch + CT_TB;
  for (int j = nT - 1; j >= 0; --j) {
    // t = y_j - sum_(i>j) L_ij^T x_i
    const int nrows = (nT - 1 - j) * CT_TB;  // rows below the diagonal tile
    if (tid < nparts * CT_TB) {
      const int p = tid / CT_TB, c = tid - p * CT_TB;
      double a = 0;
      for (int rr = p; rr < nrows; rr += nparts) {
        const int i = j + 1 + rr / CT_TB, r = rr % CT_TB;
        a += C.T[(size_t)ct_tile_index(i, j) * CT_TILE + r * CT_TB + c] * x[CT_TB * i + r];
      }
      part[p * CT_TB + c] = a;
    }
    __syncthreads();
    for (int c = tid; c < CT_TB; c += nthreads) {
      double s = C.y[CT_TB * j + c];
      for (int p = 0; p < nparts; ++p) s -= part[p * CT_TB + c];
      scratch[c] = s;
    }
    __syncthreads();
    const double* Linv = C.Linv + (size_t)j * CT_TILE;
    for (int c = tid; c < CT_TB; c += nthreads) {
      double s = 0;
      for (int r = c; r < CT_TB; ++r) s += Linv[r * CT_TB + c] * scratch[r];
      x[CT_TB * j + c] = s;
    }
    __syncthreads();
  }
}
constexpr int CT_BACKSUB_SCRATCH(int nthreads) { return CT_TB + (nthreads / CT_TB) * CT_TB; }

constexpr int CT_SMEM_DOUBLES = 3 * CT_TB * CT_LD;

// stand-alone solve of one dense SPD system (tests / diagnostics): grid.x = number of lower tiles
__global__ __launch_bounds__(CT_THREADS) void chol_tile_kernel(CholTiles C) {
  extern __shared__ __attribute__((aligned(16))) double ct_smem[];
  chol_tile_task(C, blockIdx.x, ct_smem);
}
__global__ __launch_bounds__(CT_THREADS) void chol_backsub_kernel(CholTiles C, double* x) {
  __shared__ double scratch[CT_BACKSUB_SCRATCH(CT_THREADS)];
  chol_backsub(C, x, scratch, threadIdx.x, CT_THREADS);
}

}  // namespace ba
