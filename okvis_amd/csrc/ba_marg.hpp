// Dense tail of the marginalisation: what okvis::ceres::MarginalizationError does
// after its landmark blocks are gone.
//
//   marginalizeOut, dense part      okvis_ceres/src/MarginalizationError.cpp:686-736
//   updateErrorComputation          okvis_ceres/src/MarginalizationError.cpp:806-846
//   pseudoInverseSymmSqrt           okvis_ceres/include/okvis/ceres/implementation/MarginalizationError.hpp:215-243
//
// Input: the undamped system (H, b0) over all free pose-type / speed-bias blocks of a (sub-)window, exported
// by solve_kernel(final_only = 2) after the Schur kernel eliminated every landmark with the preconditioned
// pseudo-inverse, plus the previous prior (H_old, b0_old) over some of those blocks.  One workgroup per call.
// The symmetric eigen-decompositions (Eigen::SelfAdjointEigenSolver in the reference) are done by cyclic
// Jacobi in the round-robin parallel ordering: n/2 disjoint rotations per round, column phase then row phase.
#pragma once
#include "ba_device.hpp"
#include "ba_types.hpp"

namespace ba {

constexpr int MARG_THREADS = 1024;
constexpr int MARG_MAX_PAIRS = (MAX_D_LDS + 2) / 2;

struct MargArgs {
  const unsigned char* pose_marg;  // [n_pose] 1 = eliminate this block in the dense step
  const unsigned char* sb_marg;    // [n_sb]
  int prior_dim, prior_nb;
  const int* pb_type;   // [prior_nb] 0 = pose-type, 1 = speed/bias
  const int* pb_idx;    // [prior_nb] block index in the window
  const int* pb_off;    // [prior_nb] offset inside the prior
  const double* prior_H;   // [prior_dim^2] row-major
  const double* prior_b0;  // [prior_dim]
  double* work;            // 3 * D * D doubles
  double* out_H;           // [na*na]
  double* out_b0;          // [na]
  double* out_J;           // [na*na]
  double* out_e0;          // [na]
  int* out_info;           // [0] na, [1] nm, [2] rank, [3] Jacobi sweeps (V), [4] Jacobi sweeps (H), [8 + i] kept reduced index i
};

struct JacobiScratch {
  int p[MARG_MAX_PAIRS], q[MARG_MAX_PAIRS];
  double c[MARG_MAX_PAIRS], s[MARG_MAX_PAIRS];
  int rotated;
};

// A (n x n row-major, symmetric) -> eigenvalues on its diagonal; Q <- eigenvectors (columns).  Returns sweeps.
__device__ int jacobi_eig(double* A, double* Q, int n, int tid, JacobiScratch& js) {
  for (int k = tid; k < n * n; k += MARG_THREADS) Q[k] = (k / n == k % n) ? 1.0 : 0.0;
  __syncthreads();
  if (n < 2) return 0;
  const int m = (n + 1) & ~1, half = m / 2;
  int sweep = 0;
  for (; sweep < 60; ++sweep) {
    if (tid == 0) js.rotated = 0;
    __syncthreads();
    for (int r = 0; r < m - 1; ++r) {
      if (tid < half) {
        int a, b;
        if (tid == 0) {
          a = m - 1;
          b = r;
        } else {
          a = (r + tid) % (m - 1);
          b = (r - tid + (m - 1)) % (m - 1);
        }
        const int p = a < b ? a : b, q = a < b ? b : a;
        int pp = -1;
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double apq = A[p * n + q], app = A[p * n + p], aqq = A[q * n + q];
          if (fabs(apq) >= 1e-300 && fabs(apq) > 2.220446049250313e-16 * sqrt(fabs(app * aqq))) {
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
            pp = p;
            js.rotated = 1;
          }
        }
        js.p[tid] = pp;
        js.q[tid] = q;
        js.c[tid] = c;
        js.s[tid] = s;
      }
      __syncthreads();
      for (int it = tid; it < half * n; it += MARG_THREADS) {  // A <- A G, Q <- Q G
        const int k = it / n, i = it - k * n;
        const int p = js.p[k];
        if (p < 0) continue;
        const int q = js.q[k];
        const double c = js.c[k], s = js.s[k];
        const double aip = A[i * n + p], aiq = A[i * n + q];
        A[i * n + p] = c * aip - s * aiq;
        A[i * n + q] = s * aip + c * aiq;
        const double qip = Q[i * n + p], qiq = Q[i * n + q];
        Q[i * n + p] = c * qip - s * qiq;
        Q[i * n + q] = s * qip + c * qiq;
      }
      __syncthreads();
      for (int it = tid; it < half * n; it += MARG_THREADS) {  // A <- G^T A
        const int k = it / n, j = it - k * n;
        const int p = js.p[k];
        if (p < 0) continue;
        const int q = js.q[k];
        const double c = js.c[k], s = js.s[k];
        const double apj = A[p * n + j], aqj = A[q * n + j];
        A[p * n + j] = c * apj - s * aqj;
        A[q * n + j] = s * apj + c * aqj;
      }
      __syncthreads();
    }
    if (!js.rotated) break;
    __syncthreads();
  }
  return sweep;
}

__global__ __launch_bounds__(MARG_THREADS) void marg_dense_kernel(const WinPtrs* __restrict__ wins, int w, MargArgs a) {
  const WinPtrs& W = wins[w];
  const int tid = threadIdx.x;
  const int D = W.D;
  double* H = W.S;     // [D][D]
  double* b = W.rhs;   // [D]
  __shared__ int s_kidx[MAX_D_LDS], s_midx[MAX_D_LDS], s_ridx[MAX_MARG_DIM];
  __shared__ double s_p[MAX_D_LDS], s_t[MAX_D_LDS], s_lam[MAX_D_LDS], s_ba[MAX_D_LDS];
  __shared__ int s_na, s_nm, s_rank;
  __shared__ double s_max;
  __shared__ JacobiScratch js;
  const double EPS = 2.220446049250313e-16;

  // ---- previous prior: H_ and b0_ persist inside the reference's MarginalizationError object ----
  for (int rr = tid; rr < a.prior_dim; rr += MARG_THREADS) {
    int bi = 0;
    for (int k = 0; k < a.prior_nb; ++k)
      if (a.pb_off[k] <= rr) bi = k;
    const int base = a.pb_type[bi] == 0 ? W.pose_off[a.pb_idx[bi]] : W.sb_off[a.pb_idx[bi]];
    s_ridx[rr] = base < 0 ? -1 : base + (rr - a.pb_off[bi]);
  }
  if (tid == 0) {  // kept / eliminated reduced indices, in reduced order
    int na = 0, nm = 0;
    for (int i = 0; i < W.n_pose; ++i) {
      const int off = W.pose_off[i];
      if (off < 0) continue;
      for (int k = 0; k < 6; ++k) {
        if (a.pose_marg[i]) s_midx[nm++] = off + k; else s_kidx[na++] = off + k;
      }
    }
    // pose-type blocks come first in the reduced ordering, so both lists are still sorted after the sb blocks
    for (int i = 0; i < W.n_sb; ++i) {
      const int off = W.sb_off[i];
      if (off < 0) continue;
      for (int k = 0; k < 9; ++k) {
        if (a.sb_marg[i]) s_midx[nm++] = off + k; else s_kidx[na++] = off + k;
      }
    }
    s_na = na;
    s_nm = nm;
    a.out_info[0] = na;
    a.out_info[1] = nm;
    for (int i = 0; i < na; ++i) a.out_info[8 + i] = s_kidx[i];
  }
  __syncthreads();
  for (int k = tid; k < a.prior_dim * a.prior_dim; k += MARG_THREADS) {
    const int rr = k / a.prior_dim, cc = k - rr * a.prior_dim;
    const int ri = s_ridx[rr], ci = s_ridx[cc];
    if (ri >= 0 && ci >= 0) H[ri * D + ci] += a.prior_H[k];
  }
  for (int rr = tid; rr < a.prior_dim; rr += MARG_THREADS)
    if (s_ridx[rr] >= 0) b[s_ridx[rr]] += a.prior_b0[rr];
  __syncthreads();
  const int na = s_na, nm = s_nm;
  double* A = a.work;
  double* Q = a.work + (size_t)D * D;
  double* M = a.work + 2 * (size_t)D * D;
  int sweeps_v = 0;

  if (nm > 0) {
    // ---- dense part of marginalizeOut (:686-736) ----
    for (int i = tid; i < D; i += MARG_THREADS) s_p[i] = H[i * D + i] > 1.0e-9 ? sqrt(H[i * D + i]) : 1.0e-3;  // :689
    __syncthreads();
    for (int k = tid; k < nm * nm; k += MARG_THREADS) {  // V1 = 0.5 (V + V^T) of the scaled system (:725)
      const int i = k / nm, j = k - i * nm;
      const int mi = s_midx[i], mj = s_midx[j];
      A[k] = 0.5 * (H[mi * D + mj] / (s_p[mi] * s_p[mj]) + H[mj * D + mi] / (s_p[mj] * s_p[mi]));
    }
    __syncthreads();
    sweeps_v = jacobi_eig(A, Q, nm, tid, js);
    if (tid == 0) {
      double mx = A[0];
      for (int i = 1; i < nm; ++i) mx = fmax(mx, A[i * nm + i]);
      s_max = mx;
    }
    __syncthreads();
    {
      const double tol = EPS * nm * s_max;  // pseudoInverseSymmSqrt (:225-233)
      for (int i = tid; i < nm; i += MARG_THREADS) s_lam[i] = A[i * nm + i] > tol ? sqrt(1.0 / A[i * nm + i]) : 0.0;
    }
    __syncthreads();
    for (int k = tid; k < nm * nm; k += MARG_THREADS) Q[k] *= s_lam[k % nm];  // V^(+1/2) = Q diag(l^-1/2)
    __syncthreads();
    for (int k = tid; k < na * nm; k += MARG_THREADS) {  // M = W V^(+1/2) (:729)
      const int i = k / nm, j = k - i * nm;
      const int ki = s_kidx[i];
      double s = 0;
      for (int c = 0; c < nm; ++c) s += H[ki * D + s_midx[c]] / (s_p[ki] * s_p[s_midx[c]]) * Q[c * nm + j];
      M[k] = s;
    }
    for (int j = tid; j < nm; j += MARG_THREADS) {  // V^(+1/2)^T b_b
      double s = 0;
      for (int c = 0; c < nm; ++c) s += Q[c * nm + j] * (b[s_midx[c]] / s_p[s_midx[c]]);
      s_t[j] = s;
    }
    __syncthreads();
    for (int i = tid; i < na; i += MARG_THREADS) {  // b0 = P_a (b_a - M V^(+1/2)^T b_b) (:732,:739)
      const int ki = s_kidx[i];
      double s = b[ki] / s_p[ki];
      for (int c = 0; c < nm; ++c) s -= M[i * nm + c] * s_t[c];
      const double v = s * s_p[ki];
      a.out_b0[i] = v;
      s_ba[i] = v;
    }
    for (int k = tid; k < na * na; k += MARG_THREADS) {  // H = P_a (U - M M^T) P_a (:736-738)
      const int i = k / na, j = k - i * na;
      const int ki = s_kidx[i], kj = s_kidx[j];
      double s = H[ki * D + kj] / (s_p[ki] * s_p[kj]);
      for (int c = 0; c < nm; ++c) s -= M[i * nm + c] * M[j * nm + c];
      a.out_H[k] = s * (s_p[ki] * s_p[kj]);
    }
  } else {
    for (int i = tid; i < na; i += MARG_THREADS) {
      a.out_b0[i] = b[s_kidx[i]];
      s_ba[i] = b[s_kidx[i]];
    }
    for (int k = tid; k < na * na; k += MARG_THREADS) a.out_H[k] = H[s_kidx[k / na] * D + s_kidx[k % na]];
  }
  __syncthreads();
  if (na == 0) {
    if (tid == 0) a.out_info[2] = 0;
    return;
  }

  // ---- updateErrorComputation (:806-846) ----
  const double* Ha = a.out_H;
  for (int i = tid; i < na; i += MARG_THREADS) s_p[i] = Ha[i * na + i] > 1.0e-9 ? sqrt(Ha[i * na + i]) : 1.0e-3;
  __syncthreads();
  for (int k = tid; k < na * na; k += MARG_THREADS) {
    const int i = k / na, j = k - i * na;
    A[k] = 0.5 * (Ha[i * na + j] + Ha[j * na + i]) / (s_p[i] * s_p[j]);
  }
  __syncthreads();
  const int sweeps_h = jacobi_eig(A, Q, na, tid, js);
  if (tid == 0) {
    double mx = A[0];
    for (int i = 1; i < na; ++i) mx = fmax(mx, A[i * na + i]);
    const double tol = EPS * na * mx;
    int rank = 0;
    for (int i = 0; i < na; ++i) {
      const double l = A[i * na + i];
      if (l > tol) {
        s_lam[i] = l;
        ++rank;
      } else {
        s_lam[i] = 0.0;
      }
    }
    a.out_info[2] = rank;
    a.out_info[3] = sweeps_v;
    a.out_info[4] = sweeps_h;
  }
  __syncthreads();
  for (int k = tid; k < na * na; k += MARG_THREADS) {  // J = (P U S^1/2)^T (:832)
    const int r = k / na, c = k - r * na;
    a.out_J[k] = sqrt(s_lam[r]) * Q[c * na + r] * s_p[c];
  }
  for (int r = tid; r < na; r += MARG_THREADS) {  // e0 = -S^(+1/2) U^T P^-1 b0 (:835-837)
    double s = 0;
    for (int c = 0; c < na; ++c) s += Q[c * na + r] * (s_ba[c] / s_p[c]);
    a.out_e0[r] = s_lam[r] > 0.0 ? -sqrt(1.0 / s_lam[r]) * s : 0.0;
  }
}

}  // namespace ba
