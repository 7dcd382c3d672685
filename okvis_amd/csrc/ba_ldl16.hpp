// Dense solve of the reduced camera system of one window by ONE workgroup on the fp64 matrix core (gfx950):
// blocked LDL^T with 16-wide panels, the matrix resident in MFMA accumulator registers, forward substitution riding
// along as one more column, back-substitution from the register-resident factor.
//
// This is the algebra of the reduced-camera solve of SPARSE_SCHUR (Estimator.cpp:854; in-tree analogue
// MarginalizationError.cpp:617-689): S x = b with S symmetric positive definite, D <= 175.
//
// Why this shape (measured in profiles/r02_notes.md: the previous 6-wide right-looking Cholesky in LDS spent 25 block
// columns x 2.3 us, i.e. barrier + LDS round trips, not flops):
//   * 16-wide panels: ceil((D+1)/16) <= 11 dependent steps;
//   * upper-triangular 16x16 blocks, each owned for the whole factorisation by one wave and held in the accumulator
//     layout of v_mfma_f64_16x16x4_f64 (lane l, register r = element (row (l>>4)+4r, column l&15)).  Feeding the
//     accumulator registers of U as the A operand and those of V as the B operand of four MFMAs yields U^T V in
//     accumulator layout again, so the trailing update  A_IJ -= R_I^T D^-1 R_J  needs no transposition and the matrix
//     never returns to LDS; only the current panel row R (one 16 x n strip, plain and scaled by -D^-1) goes through LDS;
//   * the dependent chain — the 16x16 diagonal block — is carried by wave 0 alone: column per lane, pivot column through
//     v_readlane, Gauss elimination in LDL^T form (reciprocals, no square roots) on [A | I], so that lanes 16..31 end up
//     with the unit-lower inverse L^-1 without a single extra instruction; the panel is then a GEMM with that inverse;
//   * the right-hand side is column D of the matrix (any spare column of the last block): the elimination turns it into
//     L^-1 b for free, and with x[D] = -1 the back-substitution is one uniform sweep  t_I = -sum_J R_IJ x_J.
//
// LDS layout of the assembled system (what the caller fills): block (I <= J) at blk(I, J), 256 doubles, element
// (r, c) of the block at (r>>2)*64 + (r&3)*16 + c  (= accumulator register r>>2... of lane (r&3)*16 + c).  Only entries
// with row <= column are referenced (diagonal blocks: upper triangle).  Everything that is not written must be zero.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_device.hpp"

namespace ba {

typedef double ldl_v4 __attribute__((ext_vector_type(4)));

constexpr int LDL_MAX_NB = 11;   // D + 1 <= 176

struct L16 {
  int nb;
  __device__ __forceinline__ int blk(int I, int J) const { return (I * nb - (I * (I - 1)) / 2 + (J - I)) * 256; }   // I <= J
  // scalar index of entry (i, j) with i >= j (the lower-triangle convention of the assembly code): stored at the mirrored
  // position (j, i) of the upper block triangle
  __device__ __forceinline__ int at(int i, int j) const {
    const int I = j >> 4, J = i >> 4, r = j & 15, c = i & 15;
    return blk(I, J) + (r >> 2) * 64 + (r & 3) * 16 + c;
  }
  __device__ __forceinline__ int sym(int i, int j) const { return i >= j ? at(i, j) : at(j, i); }
  __host__ __device__ static int blocks(int nb) { return nb * (nb + 1) / 2; }
};

// doubles of LDS the solver needs from the start of the assembly buffer (it overwrites the assembled matrix)
__host__ __device__ inline int ldl16_work_doubles(int nb) { return 3 * nb * 256 + 2 * 256 + 16 * 17 + 3 * nb * 16; }
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

// LDL^T elimination of the symmetric 16x16 block held column per lane (lanes 0..15: c(i) = A[i][lane]; lanes 16..31:
// c(i) = (i == lane - 16), the identity that becomes L^-1; the other lanes idle along).  npiv <= 16 pivots.
// The column lives in c4[i >> 2][i & 3] (the register slots that hold accumulator blocks on the other waves).
// dinv_out[k] = 1 / d_k (0 beyond npiv), written by lane 0.  Returns false when a pivot is not positive.
#define LDL_C(i) c4[(i) >> 2][(i) & 3]
__device__ __forceinline__ bool ldl16_eliminate(ldl_v4 (&c4)[4], double* dinv_out, int npiv, int lane) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double rd = 0.0;
    if (k < npiv) {
      double d = readlane_f64(LDL_C(k), k);
      ok = ok && (d > 0.0);
      d = d > 0.0 ? d : 1.0;
      rd = rcp_nr(d);
      const double u = -LDL_C(k) * rd;
#pragma unroll
      for (int i = k + 1; i < 16; ++i) LDL_C(i) = fma(readlane_f64(LDL_C(i), k), u, LDL_C(i));
    }
    if (lane == 0) dinv_out[k] = rd;
  }
  return ok;
}

// NW waves (NW * 64 threads), all of which must call.  S: the assembled system (see above) for an nb = ldl16_nb(D)
// block matrix whose column D holds the right-hand side.  x_out (LDS, >= D doubles, outside the matrix area) receives the
// solution.  *s_fail (LDS int, zeroed by the caller before a barrier) is set when a pivot is not positive.
// Ends with a barrier: x_out and *s_fail are visible to every thread on return.
template <int NW>
__device__ __forceinline__ void ldl16_solve(double* S, int D, int tid, double* x_out, int* s_fail, long long* stamps = nullptr) {
  constexpr int NREG = NW - 1;                                        // waves that own blocks
  constexpr int SLOTS_ = (LDL_MAX_NB * (LDL_MAX_NB + 1) / 2 + NREG - 1) / NREG;
  constexpr int SLOTS = SLOTS_ < 4 ? 4 : SLOTS_;   // (wave 0 keeps its 16-entry column in four of them)
  const int nb = ldl16_nb(D);
  const L16 LY{nb};
  const int nblk = L16::blocks(nb);
  const int npl = D - 16 * (nb - 1);                                  // pivots of the last block (its column npl is the rhs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // work area (aliases the assembled matrix once every block sits in registers)
  double* Rp = S;                          // [nb][256] panel row R_J of the current step, accumulator layout
  double* Rn = Rp + nb * 256;              // [nb][256] -D^-1 R_J
  double* Xs = Rn + nb * 256;              // [nb][256] L_II^-1, element (m, k) at k * 16 + m
  double* dpart = Xs + nb * 256;           // [2][256] diagonal block I with the updates k <= I - 2, accumulator layout
  double* conv = dpart + 512;              // [16][17] diagonal block on its way to the column-per-lane layout
  double* dinv = conv + 16 * 17;           // [nb][16]
  double* tv = dinv + nb * 16;             // [nb][16] t_I of the back-substitution
  double* xv = tv + nb * 16;               // [nb][16] solution incl. x[D] = -1
#define LDL_STAMP(k) do { if (stamps && tid == 0) stamps[k] = clock64(); } while (0)
  LDL_STAMP(0);

  ldl_v4 acc[SLOTS];
  ldl_v4 (&c4)[4] = reinterpret_cast<ldl_v4 (&)[4]>(acc);   // wave 0: the diagonal block, column per lane
  int sI[SLOTS], sJ[SLOTS];
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int b = s * NREG + wave - 1;
      sI[s] = nb;                          // inactive
      sJ[s] = nb;
      acc[s] = ldl_v4{0, 0, 0, 0};
      if (b < nblk) {
        int I = 0, rem = b;
        while (rem >= nb - I) {
          rem -= nb - I;
          ++I;
        }
        sI[s] = I;
        sJ[s] = I + rem;
        const double* p = S + LY.blk(I, I + rem) + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = p[64 * r];
      }
    }
  } else {
    const int j = lane & 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const double v = S[(lo >> 2) * 64 + (lo & 3) * 16 + hi];       // block (0, 0) sits at offset 0
      LDL_C(i) = lane < 16 ? v : ((lane < 32 && i == j) ? 1.0 : 0.0);
    }
  }
  __syncthreads();   // every block is in registers: the matrix area is free
  if (wave > 0) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (sI[s] == 1 && sJ[s] == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dpart[256 + 64 * r + lane] = acc[s][r];
      }
    for (int i = tid - 64; i < (nb - 1) * 16; i += (NW - 1) * 64) tv[i] = 0.0;   // (the last block's t comes from wave 0)
  }
  LDL_STAMP(1);

  for (int kb = 0; kb < nb; ++kb) {
    const int npiv = (kb == nb - 1) ? npl : 16;
    if (wave == 0) {
      const bool ok = ldl16_eliminate(c4, dinv + kb * 16, npiv, lane);
      if (stamps && tid == 0 && kb < 12) stamps[16 + 4 * kb] = clock64();
      if (lane >= 16 && lane < 32) {
        double* xo = Xs + kb * 256 + (lane - 16) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) xo[i] = LDL_C(i);
      }
      if (lane == 0 && !ok) *s_fail = 1;
      if (kb == nb - 1 && lane == npl) {   // the rhs column after the elimination: L^-1 b of the last block
#pragma unroll
        for (int k = 0; k < 16; ++k) tv[kb * 16 + k] = k < npl ? LDL_C(k) : 0.0;
      }
    }
    __syncthreads();   // B1: L_kk^-1 and 1/d published; every trailing update of step kb - 1 done
    if (stamps && tid == 0 && kb < 12) stamps[17 + 4 * kb] = clock64();
    if (wave > 0 && kb + 1 < nb) {
      // panel row: R_J = L_kk^-1 A_kJ for the owned blocks (kb, J > kb)
      double a[4], dq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = Xs[kb * 256 + 64 * q + lane];
        dq[q] = -dinv[kb * 16 + (lane >> 4) + 4 * q];
      }
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (sI[s] == kb && sJ[s] > kb) {
          ldl_v4 R{0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < 4; ++q) R = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], acc[s][q], R, 0, 0, 0);
          acc[s] = R;
          double* rp = Rp + sJ[s] * 256 + lane;
          double* rn = Rn + sJ[s] * 256 + lane;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rp[64 * r] = R[r];
            rn[64 * r] = dq[r] * R[r];
          }
        }
      }
    }
    __syncthreads();   // B2: panel row published
    if (stamps && tid == 0 && kb < 12) stamps[18 + 4 * kb] = clock64();
    if (kb + 1 >= nb) break;
    if (wave == 0) {
      // the next diagonal block: its last update, then over to the column-per-lane layout (upper triangle mirrored)
      ldl_v4 P;
      const double* dp = dpart + ((kb + 1) & 1) * 256 + lane;
      const double* rp = Rp + (kb + 1) * 256 + lane;
      const double* rn = Rn + (kb + 1) * 256 + lane;
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        P[q] = dp[64 * q];
        a[q] = rp[64 * q];
        b[q] = rn[64 * q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) P = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], P, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) conv[((lane >> 4) + 4 * r) * 17 + (lane & 15)] = P[r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int j = lane & 15;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const double v = conv[lo * 17 + hi];
        LDL_C(i) = lane < 16 ? v : ((lane < 32 && i == j) ? 1.0 : 0.0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (stamps && tid == 0 && kb < 12) stamps[19 + 4 * kb] = clock64();
    } else {
      // trailing update of the owned blocks (I > kb); the diagonal block kb + 2 first: wave 0 wants it one step later.
      // Block (kb + 1, kb + 1) is completed by wave 0 itself (above).
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const int I = sI[s], J = sJ[s];
          const bool first = (I == kb + 2 && J == kb + 2);
          if (I <= kb || I >= nb || (I == kb + 1 && J == kb + 1) || first != (pass == 0)) continue;
          const double* rp = Rp + I * 256 + lane;
          const double* rn = Rn + J * 256 + lane;
          double a[4], b[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            a[q] = rp[64 * q];
            b[q] = rn[64 * q];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[s], 0, 0, 0);
          if (first) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dpart[(I & 1) * 256 + 64 * r + lane] = acc[s][r];
          }
        }
      }
    }
  }
  LDL_STAMP(2);
  // ---- back-substitution: x_I = L_II^-T D_I^-1 t_I,  t_K -= R_KI x_I for the owners of (K < I, I)
  for (int I = nb - 1; I >= 0; --I) {
    if (wave == 0) {
      if (lane < 16) {
        const double* xc = Xs + I * 256 + lane * 16;   // column `lane` of L^-1: entries (i, lane), i = 0..15
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          s0 = fma(xc[i], dinv[I * 16 + i] * tv[I * 16 + i], s0);
          s1 = fma(xc[i + 1], dinv[I * 16 + i + 1] * tv[I * 16 + i + 1], s1);
        }
        double x = s0 + s1;
        if (I == nb - 1) x = lane == npl ? -1.0 : (lane < npl ? x : 0.0);
        xv[I * 16 + lane] = x;
        const int gi = I * 16 + lane;
        if (gi < D) x_out[gi] = x;
      }
    }
    __syncthreads();
    if (wave > 0 && I > 0) {
      const double xj = xv[I * 16 + (lane & 15)];
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (sJ[s] == I && sI[s] < I) {
          double p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = row16_sum(acc[s][r] * xj);
          if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[sI[s] * 16 + (lane >> 4) + 4 * r] -= p[r];
          }
        }
      }
    }
    __syncthreads();
  }
  LDL_STAMP(3);
#undef LDL_STAMP
}

}  // namespace ba
