// Kernel 2 — landmark Schur reduction into per-chunk partial reduced-camera systems (fp64).
//
//   S_pp' = U_pp' - sum_l W_pl (V_l + lambda D_l^2)^-1 W_p'l^T ,  rhs_p = -g_p + sum_l W_pl Vinv_l b_l
//
// (the algebra Ceres' SPARSE_SCHUR performs, Estimator.cpp:854; in-tree analogue
// MarginalizationError.cpp:617-689).  A workgroup owns (chunk of groups) x (96x96 tile pair of the pose
// part); every work-item owns one 6x6 block of the tile in registers (36 fp64 accumulators) and sweeps
// the chunk's landmarks, which are staged SCHUR_LM_BATCH at a time as dense [tile rows][3] tables in LDS
// (Y = W Vinv for the row tile, W for the column tile).  Partials go to HBM per chunk and are summed in a
// fixed order by the solve kernel: deterministic, no atomics.
//
// The accept/reject decision for the pending trial is recomputed here by wave 0 (bit-identical to the
// solve kernel, see ba_device.hpp) because the reduction must read the buffer that is about to become the
// accepted one and the damping 1/radius that follows from the decision.
#pragma once
#include "ba_device.hpp"

namespace ba {

constexpr int TILE_DIM = SCHUR_TILE_BLOCKS * 6;  // 96

__global__ __launch_bounds__(SCHUR_THREADS) void schur_kernel(const WinPtrs* __restrict__ wins,
                                                              const OptD* __restrict__ optp) {
  const WinPtrs& W = wins[blockIdx.y];
  const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
  const int bx = blockIdx.x;
  if (bx >= W.n_chunk * n_tp) return;
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;

  __shared__ double s_Y[SCHUR_LM_BATCH][TILE_DIM][3];
  __shared__ double s_W[SCHUR_LM_BATCH][TILE_DIM][3];
  __shared__ double s_vinv[SCHUR_LM_BATCH][6];
  __shared__ double s_b[SCHUR_LM_BATCH][3];
  __shared__ int s_dec[2];
  __shared__ double s_lambda;

  const int tid = threadIdx.x;
  const OptD opt = *optp;
  // ---- decision (wave 0) ----
  if (tid < 64) {
    int acc = ctrl->acc, term = 0;
    double radius = ctrl->radius;
    if (ctrl->pending) {
      double sums[6];
      wave_trial_sums(W, 1 - acc, tid, sums);
      Decision d;
      decide(ctrl, &opt, sums, &d);
      if (d.accept) acc = 1 - acc;
      radius = d.radius;
      term = d.term;
    }
    if (tid == 0) {
      s_dec[0] = acc;
      s_dec[1] = term;
      s_lambda = 1.0 / radius;
    }
  }
  __syncthreads();
  if (s_dec[1]) return;  // terminated by the decision; the solve kernel records it
  const int acc = s_dec[0];
  const double lambda = s_lambda;

  // ---- which chunk / tile pair ----
  const int chunk = bx / n_tp;
  int tp = bx - chunk * n_tp;
  int ti = 0;
  while (tp >= ti + 1) {  // lower-triangular enumeration: (0,0) (1,0) (1,1) (2,0) ...
    tp -= ti + 1;
    ++ti;
  }
  const int tj = tp;
  const int nblk = W.Dp / 6;
  const int row0 = ti * SCHUR_TILE_BLOCKS, col0 = tj * SCHUR_TILE_BLOCKS;
  const int nrow = min(SCHUR_TILE_BLOCKS, nblk - row0), ncol = min(SCHUR_TILE_BLOCKS, nblk - col0);
  const int bi = tid / SCHUR_TILE_BLOCKS, bj = tid % SCHUR_TILE_BLOCKS;
  const bool active = (bi < nrow) && (bj < ncol);

  double accS[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) accS[i] = 0.0;
  double accR[6] = {0, 0, 0, 0, 0, 0};   // sum Y b   (only bj == 0 of column tile 0)
  double accG[6] = {0, 0, 0, 0, 0, 0};   // sum g     (reprojection part of the gradient)
  double accD[6] = {0, 0, 0, 0, 0, 0};   // diag(U)   (reprojection part, undamped)

  const Chunk C = W.chunks[chunk];
  const int lm_begin = W.groups[C.group_begin].lm_begin;
  const int lm_end = W.groups[C.group_end - 1].lm_end;
  const double* Vb = W.V[acc];
  const double* bb = W.bl[acc];
  const double* Wb = W.W[acc];

  for (int l0 = lm_begin; l0 < lm_end; l0 += SCHUR_LM_BATCH) {
    const int nb = min(SCHUR_LM_BATCH, lm_end - l0);
    // zero the tables
    for (int i = tid; i < SCHUR_LM_BATCH * TILE_DIM * 3; i += SCHUR_THREADS) {
      (&s_Y[0][0][0])[i] = 0.0;
      (&s_W[0][0][0])[i] = 0.0;
    }
    if (tid < nb) {
      const int l = l0 + tid;
      const double* Vl = Vb + 6 * (size_t)l;
      double v[6] = {Vl[0], Vl[1], Vl[2], Vl[3], Vl[4], Vl[5]};
      v[0] += lambda * clampd(v[0], opt.min_lm_diag2, opt.max_lm_diag2);
      v[3] += lambda * clampd(v[3], opt.min_lm_diag2, opt.max_lm_diag2);
      v[5] += lambda * clampd(v[5], opt.min_lm_diag2, opt.max_lm_diag2);
      double vi[6];
      inv3sym(v, vi);
#pragma unroll
      for (int e = 0; e < 6; ++e) s_vinv[tid][e] = vi[e];
      s_b[tid][0] = bb[3 * (size_t)l];
      s_b[tid][1] = bb[3 * (size_t)l + 1];
      s_b[tid][2] = bb[3 * (size_t)l + 2];
    }
    __syncthreads();
    // fill: one work-item per (pair, row a)
    const int p0 = W.lm_pair_begin[l0], p1 = W.lm_pair_begin[l0 + nb];
    for (int wi = tid; wi < (p1 - p0) * 6; wi += SCHUR_THREADS) {
      const int p = p0 + wi / 6, a = wi % 6;
      const int slot = W.pair_off[p] / 6;
      const int lb = W.pair_lm[p] - l0;
      const double* Wp = Wb + (size_t)p * 18 + 3 * a;
      const double w0 = Wp[0], w1 = Wp[1], w2 = Wp[2];
      if (slot >= row0 && slot < row0 + nrow) {
        const double* vi = s_vinv[lb];
        double* y = s_Y[lb][(slot - row0) * 6 + a];
        y[0] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        y[1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        y[2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      }
      if (slot >= col0 && slot < col0 + ncol) {
        double* w = s_W[lb][(slot - col0) * 6 + a];
        w[0] = w0;
        w[1] = w1;
        w[2] = w2;
      }
    }
    __syncthreads();
    if (active) {
      for (int lb = 0; lb < nb; ++lb) {
        double y[18], w[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          y[i] = (&s_Y[lb][bi * 6][0])[i];
          w[i] = (&s_W[lb][bj * 6][0])[i];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c)
            accS[6 * r + c] -= y[3 * r] * w[3 * c] + y[3 * r + 1] * w[3 * c + 1] + y[3 * r + 2] * w[3 * c + 2];
        if (tj == 0 && bj == 0) {
#pragma unroll
          for (int r = 0; r < 6; ++r)
            accR[r] += y[3 * r] * s_b[lb][0] + y[3 * r + 1] * s_b[lb][1] + y[3 * r + 2] * s_b[lb][2];
        }
      }
    }
    __syncthreads();
  }

  // ---- add the per-group block partials of this chunk (U_pp, U_pe, g_p) ----
  if (active) {
    const double* gp = W.gpart[acc];
    const int my_row = (row0 + bi) * 6, my_col = (col0 + bj) * 6;
    for (int g = C.group_begin; g < C.group_end; ++g) {
      const Group G = W.groups[g];
      for (int t = G.task_begin; t < G.task_end; ++t) {
        const Task T = W.tasks[t];
        const double* o = gp + T.out;
        if (T.type < 2) {
          if (T.off_a == my_row && T.off_a == my_col) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int c = 0; c < 6; ++c) accS[6 * r + c] += (r <= c) ? o[ut6(r, c)] : o[ut6(c, r)];
          }
          if (tj == 0 && bj == 0 && T.off_a == my_row) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
              accG[r] += o[21 + r];
              accD[r] += o[ut6(r, r)];
            }
          }
        } else {
          if (T.off_a == my_row && T.off_b == my_col) {
#pragma unroll
            for (int i = 0; i < 36; ++i) accS[i] += o[i];
          } else if (T.off_b == my_row && T.off_a == my_col) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int c = 0; c < 6; ++c) accS[6 * r + c] += o[6 * c + r];
          }
        }
      }
    }
    // ---- write the partial ----
    const int Dp = W.Dp;
    double* sp = W.spart + (size_t)chunk * ((size_t)Dp * Dp + 3 * Dp);
    const int r0 = (row0 + bi) * 6, c0 = (col0 + bj) * 6;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) sp[(size_t)(r0 + r) * Dp + c0 + c] = accS[6 * r + c];
    if (tj == 0 && bj == 0) {
      double* sr = sp + (size_t)Dp * Dp;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        sr[r0 + r] = accR[r];
        sr[Dp + r0 + r] = accG[r];
        sr[2 * Dp + r0 + r] = accD[r];
      }
    }
  }
}

}  // namespace ba
