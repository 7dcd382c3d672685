// Kernel 2 — landmark Schur reduction into per-chunk partial reduced-camera systems (fp64).
//
//   S_pp' = U_pp' - sum_l W_pl (V_l + lambda D_l^2)^-1 W_p'l^T ,  rhs_p = -g_p + sum_l W_pl Vinv_l b_l
//
// (the algebra Ceres' SPARSE_SCHUR performs, Estimator.cpp:854; in-tree analogue
// MarginalizationError.cpp:617-689).  This kernel produces the landmark part  -sum Y W^T  and  sum Y b;
// the U_pp / g_p parts are summed by the solve kernel from the linearise kernel's per-group partials.
//
// A workgroup owns (chunk of groups) x (tile pair of <=16x16 pose blocks).  Only the lower triangle is
// computed.  The 256 work-items are (6x6 block pair) x (landmark slice): every work-item keeps its 6x6
// block in 36 fp64 accumulators and sweeps every n_slice-th landmark of the current batch; the slices sit
// in adjacent lanes and are combined with two xor-shuffles at the end.  Landmarks are staged
// SCHUR_LM_BATCH at a time as dense [tile rows][3] tables in LDS (Y = W Vinv for the row tile, W for the
// column tile).  Partials go to HBM per chunk in the solve kernel's block-packed layout and are summed
// there in fixed chunk order: deterministic, no atomics.
//
// The accept/reject decision for the pending trial is recomputed here by wave 0 (bit-identical to the
// solve kernel, see ba_device.hpp) because the reduction must read the buffer that is about to become the
// accepted one and the damping 1/radius that follows from the decision.
#pragma once
#include "ba_device.hpp"

namespace ba {

constexpr int TILE_DIM = SCHUR_TILE_BLOCKS * 6;  // 96

__global__ __launch_bounds__(SCHUR_THREADS) void schur_kernel(const WinPtrs* __restrict__ wins,
                                                              const OptD* __restrict__ optp, int tile_rows) {
  const WinPtrs& W = wins[blockIdx.y];
  const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
  const int bx = blockIdx.x;
  if (bx >= W.n_chunk * n_tp) return;
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;

  // dynamic LDS: [SCHUR_LM_BATCH][tile_rows][3] for Y and W, tile_rows = min(96, Dp) of the batch
  extern __shared__ __attribute__((aligned(16))) double sch_smem[];
  __shared__ double s_vinv[SCHUR_LM_BATCH][6];
  __shared__ double s_b[SCHUR_LM_BATCH][3];
  __shared__ int s_dec[2];
  __shared__ double s_lambda;

  const int tid = threadIdx.x;
  const OptD opt = *optp;
  const int trows = tile_rows;
  double* s_Y = sch_smem;
  double* s_W = sch_smem + (size_t)SCHUR_LM_BATCH * trows * 3;
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
  const bool diag_tile = (ti == tj);
  // block pairs of this tile pair and the landmark slicing
  const int npairs = diag_tile ? nrow * (nrow + 1) / 2 : nrow * ncol;
  int nslice = 1;
  while (nslice < 4 && npairs * nslice * 2 <= SCHUR_THREADS) nslice *= 2;
  const int pi = tid / nslice, slice = tid - pi * nslice;
  const bool active = pi < npairs;
  int bi = 0, bj = 0;
  if (active) {
    if (diag_tile) {
      bi = (int)((sqrtf(8.0f * pi + 1.0f) - 1.0f) * 0.5f);
      while ((bi + 1) * (bi + 2) / 2 <= pi) ++bi;
      while (bi * (bi + 1) / 2 > pi) --bi;
      bj = pi - bi * (bi + 1) / 2;
    } else {
      bi = pi / ncol;
      bj = pi - bi * ncol;
    }
  }
  const bool do_rhs = active && diag_tile && bi == bj;  // every row block of a diagonal tile exactly once

  double accS[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) accS[i] = 0.0;
  double accR[6] = {0, 0, 0, 0, 0, 0};  // sum Y b

  const Chunk C = W.chunks[chunk];
  const int lm_begin = W.groups[C.group_begin].lm_begin;
  const int lm_end = W.groups[C.group_end - 1].lm_end;
  const double* Vb = W.V[acc];
  const double* bb = W.bl[acc];
  const double* Wb = W.W[acc];

  for (int l0 = lm_begin; l0 < lm_end; l0 += SCHUR_LM_BATCH) {
    const int nb = min(SCHUR_LM_BATCH, lm_end - l0);
    // zero the tables (missing (landmark, block) pairs contribute nothing)
    {
      double2* zy = reinterpret_cast<double2*>(s_Y);
      double2* zw = reinterpret_cast<double2*>(s_W);
      for (int i = tid; i < SCHUR_LM_BATCH * trows * 3 / 2; i += SCHUR_THREADS) {
        zy[i] = make_double2(0.0, 0.0);
        zw[i] = make_double2(0.0, 0.0);
      }
    }
    if (tid < nb) {
      const int l = l0 + tid;
      const double* Vl = Vb + 6 * (size_t)l;
      double v[6] = {Vl[0], Vl[1], Vl[2], Vl[3], Vl[4], Vl[5]};
      double vi[6];
      if (opt.marg_mode) {
        pinv3sym_precond(v, vi);   // MarginalizationError::marginalizeOut landmark path (no damping)
      } else {
        v[0] += lambda * clampd(v[0], opt.min_lm_diag2, opt.max_lm_diag2);
        v[3] += lambda * clampd(v[3], opt.min_lm_diag2, opt.max_lm_diag2);
        v[5] += lambda * clampd(v[5], opt.min_lm_diag2, opt.max_lm_diag2);
        inv3sym(v, vi);
      }
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
        double* y = s_Y + ((size_t)lb * trows + (slot - row0) * 6 + a) * 3;
        y[0] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        y[1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        y[2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
      }
      if (slot >= col0 && slot < col0 + ncol) {
        double* w = s_W + ((size_t)lb * trows + (slot - col0) * 6 + a) * 3;
        w[0] = w0;
        w[1] = w1;
        w[2] = w2;
      }
    }
    __syncthreads();
    if (active) {
      for (int lb = slice; lb < nb; lb += nslice) {
        double y[18], w[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          y[i] = s_Y[((size_t)lb * trows + bi * 6) * 3 + i];
          w[i] = s_W[((size_t)lb * trows + bj * 6) * 3 + i];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c < 6; ++c)
            accS[6 * r + c] -= y[3 * r] * w[3 * c] + y[3 * r + 1] * w[3 * c + 1] + y[3 * r + 2] * w[3 * c + 2];
        if (do_rhs) {
#pragma unroll
          for (int r = 0; r < 6; ++r)
            accR[r] += y[3 * r] * s_b[lb][0] + y[3 * r + 1] * s_b[lb][1] + y[3 * r + 2] * s_b[lb][2];
        }
      }
    }
    __syncthreads();
  }

  // ---- combine the landmark slices (adjacent lanes) in a fixed order ----
  for (int o = 1; o < nslice; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 36; ++i) accS[i] += __shfl_xor(accS[i], o, 64);
#pragma unroll
    for (int i = 0; i < 6; ++i) accR[i] += __shfl_xor(accR[i], o, 64);
  }
  // ---- write the partial (pose part, row-major block-packed lower triangle | Y b | g | diag U) ----
  if (active && slice == 0) {
    const int gbi = row0 + bi, gbj = col0 + bj;  // gbi >= gbj
    double accG[6] = {0, 0, 0, 0, 0, 0}, accD[6] = {0, 0, 0, 0, 0, 0};
    const double* gp = W.gpart[acc];
    if (gbi == gbj) {
      // U_pp and g_p of this chunk's groups for pose block gbi (host-built list, fixed order)
      const int lb = W.chunk_diag_begin[chunk * nblk + gbi], le = W.chunk_diag_begin[chunk * nblk + gbi + 1];
      for (int k = lb; k < le; ++k) {
        const double* o = gp + W.chunk_diag_out[k];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = 0; c <= r; ++c) accS[6 * r + c] += o[c * 6 - (c * (c - 1)) / 2 + (r - c)];
          accG[r] += o[21 + r];
          accD[r] += o[r * 6 - (r * (r - 1)) / 2];
        }
      }
    } else {
      // pose x extrinsics cross blocks J_pose^T J_ext
      for (int k = W.chunk_cross_begin[chunk]; k < W.chunk_cross_begin[chunk + 1]; ++k) {
        const int oa = W.chunk_cross[3 * k], ob = W.chunk_cross[3 * k + 1];
        const double* o = gp + W.chunk_cross[3 * k + 2];
        if (oa == gbi * 6 && ob == gbj * 6) {
#pragma unroll
          for (int i = 0; i < 36; ++i) accS[i] += o[i];
        } else if (ob == gbi * 6 && oa == gbj * 6) {
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) accS[6 * r + c] += o[6 * c + r];
        }
      }
    }
    double* sp = W.spart + (size_t)chunk * W.spart_stride;
    double* blk = sp + (size_t)(gbi * (gbi + 1) / 2 + gbj) * 36;
#pragma unroll
    for (int i = 0; i < 36; ++i) blk[i] = accS[i];
    if (do_rhs) {
      double* sr = sp + (size_t)(nblk * (nblk + 1) / 2) * 36;
      const int Dp = nblk * 6;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        sr[gbi * 6 + r] = accR[r];
        sr[Dp + gbi * 6 + r] = accG[r];
        sr[2 * Dp + gbi * 6 + r] = accD[r];
      }
    }
  }
}

}  // namespace ba
