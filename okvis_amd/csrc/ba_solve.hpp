// Kernel 3 — trust-region control + reduced-camera solve, one workgroup per window (fp64, LDS resident).
//
//   1. accept / reject the pending trial (Ceres LevenbergMarquardtStrategy semantics, ba_device.hpp)
//   2. assemble the reduced system of the accepted linearisation in LDS (packed lower triangle):
//        sum of the Schur partials (fixed chunk order)  +  IMU / prior / marginalisation-prior blocks
//   3. convergence tests of the step just accepted (gradient, function tolerance)
//   4. LM damping  S += lambda * clamp(diag U),  blocked right-looking Cholesky with 6x6 register blocks,
//      forward / backward substitution
//   5. trial poses / speed-biases  x (+) delta  (PoseLocalParameterization::plus), model-decrease scalars
//
// This is the only kernel that writes the window's Ctrl record.
#pragma once
#include "ba_device.hpp"

namespace ba {

__device__ __forceinline__ int pidx(int i, int j) { return (i * (i + 1)) / 2 + j; }  // i >= j

// accumulate J^T J (lower triangle, reduced coordinates) and J^T r of one small factor.
// J: nres x ncol row-major (ncol = sum of dims), col_off[c] = reduced index of local column c or -1.
__device__ __forceinline__ void add_small_factor(double* S, double* g, double* d2, const double* J, const double* r,
                                                 int nres, int ncol, const int* col_off, int tid, int nthreads) {
  for (int wi = tid; wi < ncol * ncol; wi += nthreads) {
    const int a = wi / ncol, b = wi - a * ncol;
    const int ra = col_off[a], rb = col_off[b];
    if (ra < 0 || rb < 0 || ra < rb) continue;
    double s = 0;
    for (int k = 0; k < nres; ++k) s += J[k * ncol + a] * J[k * ncol + b];
    S[pidx(ra, rb)] += s;
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

__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(const WinPtrs* __restrict__ wins,
                                                              const OptD* __restrict__ optp, int final_only) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const WinPtrs& W = wins[blockIdx.x];
  Ctrl* gctrl = W.ctrl;
  if (gctrl->done) return;
  const int tid = threadIdx.x;
  const OptD opt = *optp;
  const int D = W.D, Dp = W.Dp;
  const int Dpad = ((D + 5) / 6) * 6, nbk = Dpad / 6;

  double* S = smem;                                  // packed lower, Dpad
  double* s_rhs = S + (size_t)Dpad * (Dpad + 1) / 2; // Dpad
  double* s_g = s_rhs + Dpad;
  double* s_d2 = s_g + Dpad;
  double* s_x = s_d2 + Dpad;
  __shared__ Ctrl c;
  __shared__ int s_accepted, s_was_first, s_fail;
  __shared__ double s_cost_change, s_old_cost, s_lm_gmax;
  __shared__ int s_coloff[64];
  __shared__ double s_red[SOLVE_THREADS / 64];

  // ------------------------------------------------------------------ 1. decision
  if (tid < 64) {
    double sums[6] = {0, 0, 0, 0, 0, 0};
    Decision d;
    d.accept = 0; d.term = 0;
    const int pending = gctrl->pending;
    if (pending) {
      wave_trial_sums(W, 1 - gctrl->acc, tid, sums);
      decide(gctrl, &opt, sums, &d);
    }
    if (tid == 0) {
      c = *gctrl;
      s_accepted = 0;
      s_was_first = c.first;
      s_cost_change = 0;
      s_old_cost = c.cost;
      s_lm_gmax = 0;
      s_fail = 0;
      if (pending) {
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
    }
  }
  __syncthreads();
  if (c.done) {
    if (tid == 0) *gctrl = c;
    return;
  }
  const int acc = c.acc;

  // ------------------------------------------------------------------ 2. assembly
  const int npk = Dpad * (Dpad + 1) / 2;
  for (int i = tid; i < npk; i += SOLVE_THREADS) S[i] = 0.0;
  for (int i = tid; i < Dpad; i += SOLVE_THREADS) {
    s_rhs[i] = 0.0;
    s_g[i] = 0.0;
    s_d2[i] = 0.0;
    s_x[i] = 0.0;
  }
  __syncthreads();
  {
    const size_t stride = (size_t)Dp * Dp + 3 * Dp;
    const int np = Dp * (Dp + 1) / 2;
    for (int k = tid; k < np; k += SOLVE_THREADS) {
      // invert k = i(i+1)/2 + j
      int i = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= k) ++i;
      while (i * (i + 1) / 2 > k) --i;
      const int j = k - i * (i + 1) / 2;
      double s = 0;
      for (int ch = 0; ch < W.n_chunk; ++ch) s += W.spart[ch * stride + (size_t)i * Dp + j];
      S[k] = s;
    }
    for (int i = tid; i < Dp; i += SOLVE_THREADS) {
      double yb = 0, g = 0, du = 0;
      for (int ch = 0; ch < W.n_chunk; ++ch) {
        const double* sr = W.spart + ch * stride + (size_t)Dp * Dp;
        yb += sr[i];
        g += sr[Dp + i];
        du += sr[2 * Dp + i];
      }
      s_rhs[i] = yb;
      s_g[i] = g;
      s_d2[i] = du;
    }
  }
  __syncthreads();
  // ---- IMU factors: J 15x30 | r 15 ----
  for (int f = 0; f < W.n_imu; ++f) {
    if (tid < 30) {
      const int blk = tid < 6 ? 0 : (tid < 15 ? 1 : (tid < 21 ? 2 : 3));
      const int within = tid - (blk == 0 ? 0 : (blk == 1 ? 6 : (blk == 2 ? 15 : 21)));
      int off;
      if (blk == 0) off = W.pose_off[W.imu_pose0[f]];
      else if (blk == 1) off = W.sb_off[W.imu_sb0[f]];
      else if (blk == 2) off = W.pose_off[W.imu_pose1[f]];
      else off = W.sb_off[W.imu_sb1[f]];
      s_coloff[tid] = off < 0 ? -1 : off + within;
    }
    __syncthreads();
    const double* L = W.imu_lin[acc] + (size_t)f * IMU_LIN_STRIDE;
    add_small_factor(S, s_g, s_d2, L, L + 15 * 30, 15, 30, s_coloff, tid, SOLVE_THREADS);
    __syncthreads();
  }
  // ---- pose priors: J 6x6 | r 6 ----
  for (int f = 0; f < W.n_pprior; ++f) {
    if (tid < 6) {
      const int off = W.pose_off[W.pprior_pose[f]];
      s_coloff[tid] = off < 0 ? -1 : off + tid;
    }
    __syncthreads();
    const double* L = W.pp_lin[acc] + (size_t)f * 42;
    add_small_factor(S, s_g, s_d2, L, L + 36, 6, 6, s_coloff, tid, SOLVE_THREADS);
    __syncthreads();
  }
  // ---- speed/bias priors: J = -sqrtInfo (9x9 const) | r 9 ----
  for (int f = 0; f < W.n_sbprior; ++f) {
    if (tid < 9) {
      const int off = W.sb_off[W.sbprior_sb[f]];
      s_coloff[tid] = off < 0 ? -1 : off + tid;
    }
    __syncthreads();
    // J^T J and J^T r are sign-invariant / sign-flipped: use +sqrtInfo with -r
    const double* Jc = W.sbprior_sqrtinfo + (size_t)f * 81;
    const double* r = W.sbp_lin[acc] + (size_t)f * 9;
    for (int wi = tid; wi < 81; wi += SOLVE_THREADS) {
      const int a = wi / 9, b = wi - 9 * a;
      const int ra = s_coloff[a], rb = s_coloff[b];
      if (ra < 0 || rb < 0 || ra < rb) continue;
      double s = 0;
      for (int k = 0; k < 9; ++k) s += Jc[k * 9 + a] * Jc[k * 9 + b];
      S[pidx(ra, rb)] += s;
      if (a == b) s_d2[ra] += s;
    }
    if (tid < 9 && s_coloff[tid] >= 0) {
      double s = 0;
      for (int k = 0; k < 9; ++k) s -= Jc[k * 9 + tid] * r[k];
      s_g[s_coloff[tid]] += s;
    }
    __syncthreads();
  }
  // ---- relative pose factors: [J0 6x6 | J1 6x6 | r 6] stored as J 6x12 row-major | r ----
  for (int f = 0; f < W.n_rel; ++f) {
    if (tid < 12) {
      const int off = W.pose_off[tid < 6 ? W.rel_pose0[f] : W.rel_pose1[f]];
      s_coloff[tid] = off < 0 ? -1 : off + (tid % 6);
    }
    __syncthreads();
    const double* L = W.rel_lin[acc] + (size_t)f * 78;
    add_small_factor(S, s_g, s_d2, L, L + 72, 6, 12, s_coloff, tid, SOLVE_THREADS);
    __syncthreads();
  }
  // ---- marginalisation prior: H = B^T (J^T J) B, g = B^T J^T e ----
  if (W.marg_dim > 0) {
    const int Dm = W.marg_dim, nb = W.marg_nb;
    const double* M = W.marg_lin_M[acc];
    const double* JTe = W.marg_lin_e[acc] + Dm;  // [e | J^T e]
    for (int wi = tid; wi < Dm * Dm; wi += SOLVE_THREADS) {
      const int rr = wi / Dm, cc = wi - rr * Dm;
      int bi = 0, bj = 0;
      for (int b = 0; b < nb; ++b) {
        if (W.marg_block_off[b] <= rr) bi = b;
        if (W.marg_block_off[b] <= cc) bj = b;
      }
      const int oi = W.marg_block_off[bi], oj = W.marg_block_off[bj];
      const int li = rr - oi, lj = cc - oj;
      const int Ri = W.marg_block_type[bi] == 0 ? W.pose_off[W.marg_block_idx[bi]] : W.sb_off[W.marg_block_idx[bi]];
      const int Rj = W.marg_block_type[bj] == 0 ? W.pose_off[W.marg_block_idx[bj]] : W.sb_off[W.marg_block_idx[bj]];
      if (Ri < 0 || Rj < 0 || Ri + li < Rj + lj) continue;
      const bool roti = (W.marg_block_type[bi] == 0) && li >= 3;
      const bool rotj = (W.marg_block_type[bj] == 0) && lj >= 3;
      double s = 0;
      if (!roti && !rotj) {
        s = W.marg_H0[(size_t)rr * Dm + cc];
      } else {
        for (int a = 0; a < (roti ? 3 : 1); ++a) {
          const int r2 = roti ? oi + 3 + a : rr;
          const double wa = roti ? M[9 * bi + 3 * a + (li - 3)] : 1.0;
          for (int b = 0; b < (rotj ? 3 : 1); ++b) {
            const int c2 = rotj ? oj + 3 + b : cc;
            const double wb = rotj ? M[9 * bj + 3 * b + (lj - 3)] : 1.0;
            s += wa * W.marg_H0[(size_t)r2 * Dm + c2] * wb;
          }
        }
      }
      S[pidx(Ri + li, Rj + lj)] += s;
      if (Ri + li == Rj + lj) s_d2[Ri + li] += s;
    }
    for (int rr = tid; rr < Dm; rr += SOLVE_THREADS) {
      int bi = 0;
      for (int b = 0; b < nb; ++b)
        if (W.marg_block_off[b] <= rr) bi = b;
      const int oi = W.marg_block_off[bi], li = rr - oi;
      const int Ri = W.marg_block_type[bi] == 0 ? W.pose_off[W.marg_block_idx[bi]] : W.sb_off[W.marg_block_idx[bi]];
      if (Ri < 0) continue;
      double s;
      if (W.marg_block_type[bi] == 0 && li >= 3) {
        s = 0;
        for (int a = 0; a < 3; ++a) s += M[9 * bi + 3 * a + (li - 3)] * JTe[oi + 3 + a];
      } else {
        s = JTe[rr];
      }
      s_g[Ri + li] += s;
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ 3. convergence of the accepted step
  {
    double m = 0;
    for (int i = tid; i < D; i += SOLVE_THREADS) m = fmax(m, fabs(s_g[i]));
    m = wave_max(m);
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      double gm = s_lm_gmax;
      for (int i = 0; i < SOLVE_THREADS / 64; ++i) gm = fmax(gm, s_red[i]);
      if (s_accepted) {
        c.grad_max = gm;
        if (s_was_first) {
          c.initial_cost = c.cost;
          c.abs_grad_tol = opt.gradient_tolerance * fmax(gm, 2.220446049250313e-16);
        }
        if (opt.gradient_tolerance > 0 && gm <= c.abs_grad_tol) {
          c.done = 2 + 1;
        } else if (!s_was_first && opt.function_tolerance > 0 &&
                   fabs(s_cost_change) < opt.function_tolerance * s_old_cost) {
          c.done = 1 + 1;
        }
        c.first = 0;
      }
    }
    __syncthreads();
  }
  if (W.grad)
    for (int i = tid; i < D; i += SOLVE_THREADS) W.grad[i] = s_g[i];
  if (c.done || final_only) {
    if (tid == 0) *gctrl = c;
    return;
  }

  // ------------------------------------------------------------------ 4. damping + Cholesky
  const double lambda = 1.0 / c.radius;
  for (int i = tid; i < Dpad; i += SOLVE_THREADS) {
    if (i < D) {
      const double d2 = clampd(s_d2[i], opt.min_lm_diag2, opt.max_lm_diag2);
      s_d2[i] = d2;
      S[pidx(i, i)] += lambda * d2;
      s_rhs[i] = s_rhs[i] - s_g[i];
    } else {
      S[pidx(i, i)] = 1.0;  // identity padding up to a multiple of 6
      s_rhs[i] = 0.0;
    }
  }
  __syncthreads();
  if (W.S) {  // parity/debug copy of the damped system
    for (int k = tid; k < D * D; k += SOLVE_THREADS) {
      const int i = k / D, j = k - i * D;
      W.S[k] = (i >= j) ? S[pidx(i, j)] : S[pidx(j, i)];
    }
    for (int i = tid; i < D; i += SOLVE_THREADS) {
      W.rhs[i] = s_rhs[i];
      W.Dp2[i] = s_d2[i];
    }
  }
  for (int kb = 0; kb < nbk; ++kb) {
    const int k0 = kb * 6;
    // (1) factor the 6x6 diagonal block in registers of wave 0
    if (tid < 64) {
      const int i = tid / 6, j = tid - 6 * i;
      const bool in = tid < 36 && i >= j;
      double a = in ? S[pidx(k0 + i, k0 + j)] : 0.0;
      int bad = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double dk = __shfl(a, k * 6 + k, 64);
        if (!(dk > 0.0)) bad = 1;
        const double sd = sqrt(dk > 0.0 ? dk : 1.0);
        if (in && j == k) a = (i == k) ? sd : a / sd;
        const double aik = __shfl(a, (tid < 36 ? i : 0) * 6 + k, 64);
        const double ajk = __shfl(a, (tid < 36 ? j : 0) * 6 + k, 64);
        if (in && j > k) a -= aik * ajk;
      }
      if (in) S[pidx(k0 + i, k0 + j)] = a;
      if (bad && tid == 0) s_fail = 1;
    }
    __syncthreads();
    // (2) panel: rows below, L_ik = A_ik L_kk^-T  (plus the rhs row: forward substitution for free)
    for (int r = k0 + 6 + tid; r < Dpad; r += SOLVE_THREADS) {
      double x[6];
#pragma unroll
      for (int cix = 0; cix < 6; ++cix) {
        double v = S[pidx(r, k0 + cix)];
        for (int m = 0; m < cix; ++m) v -= x[m] * S[pidx(k0 + cix, k0 + m)];
        x[cix] = v / S[pidx(k0 + cix, k0 + cix)];
      }
#pragma unroll
      for (int cix = 0; cix < 6; ++cix) S[pidx(r, k0 + cix)] = x[cix];
    }
    if (tid == SOLVE_THREADS - 1) {  // y_k = L_kk^-1 (rhs_k - ...), the rest of rhs is updated in (3)
      double y[6];
      for (int cix = 0; cix < 6; ++cix) {
        double v = s_rhs[k0 + cix];
        for (int m = 0; m < cix; ++m) v -= y[m] * S[pidx(k0 + cix, k0 + m)];
        y[cix] = v / S[pidx(k0 + cix, k0 + cix)];
      }
      for (int cix = 0; cix < 6; ++cix) s_rhs[k0 + cix] = y[cix];
    }
    __syncthreads();
    // (3) trailing update with 6x6 register blocks: A_(bi,bj) -= L_(bi,k) L_(bj,k)^T
    const int nt = nbk - kb - 1;
    for (int wi = tid; wi < nt * (nt + 1) / 2; wi += SOLVE_THREADS) {
      int bi = (int)((sqrt(8.0 * wi + 1.0) - 1.0) * 0.5);
      while ((bi + 1) * (bi + 2) / 2 <= wi) ++bi;
      while (bi * (bi + 1) / 2 > wi) --bi;
      const int bj = wi - bi * (bi + 1) / 2;
      const int r0 = (kb + 1 + bi) * 6, c0 = (kb + 1 + bj) * 6;
      double Li[36], Lj[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          Li[6 * r + m] = S[pidx(r0 + r, k0 + m)];
          Lj[6 * r + m] = S[pidx(c0 + r, k0 + m)];
        }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cix = 0; cix < 6; ++cix) {
          if (bi == bj && cix > r) continue;
          double s = 0;
#pragma unroll
          for (int m = 0; m < 6; ++m) s += Li[6 * r + m] * Lj[6 * cix + m];
          S[pidx(r0 + r, c0 + cix)] -= s;
        }
    }
    // rhs rows below: rhs_i -= L_(i,k) y_k
    for (int r = k0 + 6 + tid; r < Dpad; r += SOLVE_THREADS) {
      double v = s_rhs[r];
#pragma unroll
      for (int m = 0; m < 6; ++m) v -= S[pidx(r, k0 + m)] * s_rhs[k0 + m];
      s_rhs[r] = v;
    }
    __syncthreads();
  }
  if (s_fail) {  // not positive definite: invalid step (handled like a rejection)
    if (tid == 0) {
      c.iter++;
      c.chol_fail++;
      c.radius = c.radius / c.decrease_factor;
      c.decrease_factor *= 2.0;
      c.pending = 0;
      if (c.radius < opt.min_radius) c.done = 5 + 1;
      *gctrl = c;
    }
    return;
  }
  // back substitution  L^T x = y, blocked from the last block up
  for (int kb = nbk - 1; kb >= 0; --kb) {
    const int k0 = kb * 6;
    if (tid == 0) {
      double x[6];
      for (int cix = 5; cix >= 0; --cix) {
        double v = s_rhs[k0 + cix];
        for (int m = cix + 1; m < 6; ++m) v -= S[pidx(k0 + m, k0 + cix)] * x[m];
        x[cix] = v / S[pidx(k0 + cix, k0 + cix)];
      }
      for (int cix = 0; cix < 6; ++cix) s_x[k0 + cix] = x[cix];
    }
    __syncthreads();
    for (int j = tid; j < k0; j += SOLVE_THREADS) {
      double v = s_rhs[j];
#pragma unroll
      for (int m = 0; m < 6; ++m) v -= S[pidx(k0 + m, j)] * s_x[k0 + m];
      s_rhs[j] = v;
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ 5. scalars, trial state, ctrl
  {
    double gd = 0, ddd = 0, s2 = 0;
    for (int i = tid; i < D; i += SOLVE_THREADS) {
      const double x = s_x[i];
      W.step[i] = x;
      gd += s_g[i] * x;
      ddd += s_d2[i] * x * x;
      s2 += x * x;
    }
    double x2 = 0;
    const int trial = 1 - acc;
    for (int b = tid; b < W.n_pose; b += SOLVE_THREADS) {
      const double* xp = W.pose[acc] + 7 * (size_t)b;
      double* xt = W.pose[trial] + 7 * (size_t)b;
      const int off = W.pose_off[b];
      if (off >= 0) {
        double xin[7], xo[7];
        for (int k = 0; k < 7; ++k) {
          xin[k] = xp[k];
          x2 += xp[k] * xp[k];
        }
        pose_oplus(xin, s_x + off, xo);
        for (int k = 0; k < 7; ++k) xt[k] = xo[k];
      } else {
        for (int k = 0; k < 7; ++k) xt[k] = xp[k];
      }
    }
    for (int b = tid; b < W.n_sb; b += SOLVE_THREADS) {
      const double* xp = W.sb[acc] + 9 * (size_t)b;
      double* xt = W.sb[trial] + 9 * (size_t)b;
      const int off = W.sb_off[b];
      for (int k = 0; k < 9; ++k) {
        const double v = xp[k];
        if (off >= 0) x2 += v * v;
        xt[k] = off >= 0 ? v + s_x[off + k] : v;
      }
    }
    __shared__ double s_sc[4][SOLVE_THREADS / 64];
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
      c.iter++;
      c.pending = 1;
      *gctrl = c;
    }
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
