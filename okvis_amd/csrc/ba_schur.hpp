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

// pair-list offsets of the landmark batches of a chunk: prefetched into registers (stand-alone kernel: the loads overlap with
// the decision) or read where they are needed (fused call: nothing to hide behind)
struct SchurPairBeginRegs {
  int v[SCHUR_CHUNK_LM_MAX / SCHUR_LM_BATCH + 1];
  __device__ __forceinline__ int begin(int ib) const {
    int r = 0;
#pragma unroll
    for (int i = 0; i <= SCHUR_CHUNK_LM_MAX / SCHUR_LM_BATCH; ++i)   // static indexing keeps the values in registers
      if (i == ib) r = v[i];
    return r;
  }
};
struct SchurPairBeginGlobal {
  const BA_G int* lm_pair_begin;
  int lm_begin, lm_end;
  __device__ __forceinline__ int begin(int ib) const { return lm_pair_begin[min(lm_begin + ib * SCHUR_LM_BATCH, lm_end)]; }
};

// The reduction of one chunk for one tile pair: everything of schur_kernel after the decision, also called by the linearise
// kernel for its own group right after it has written V, b, W and the per-group J^T J partials (fused mode, ba_linearize.hpp).
// LDS: tables = 2 x SCHUR_LM_BATCH x trows x 3 doubles, vinv [SCHUR_CHUNK_LM_MAX][6], bvec [SCHUR_CHUNK_LM_MAX][3],
// boff [SCHUR_THREADS].  PB supplies the pair-list offset of the ib-th landmark batch (the stand-alone kernel prefetches them).
// All SCHUR_THREADS work-items of the workgroup call it (it contains barriers).
template <class PB>
__device__ __forceinline__ void schur_reduce_chunk(const WinPtrs& W, const OptD& opt, int chunk, int tp, int lm_begin, int lm_end,
                                                   int acc, double lambda, int trows, double* sch_smem, double (*s_vinv)[6],
                                                   double (*s_b)[3], int* s_boff, const PB& pb, bool stamps) {
  const int tid = threadIdx.x;
#undef SSTAMP
#define SSTAMP(k) do { if (stamps && W.prof && tid == 0) W.prof[k] = (double)clock64(); } while (0)
  double* s_Y = sch_smem;
  double* s_W = sch_smem + (size_t)SCHUR_LM_BATCH * trows * 3;
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

  const double* Vb = W.V[acc];
  const double* bb = W.bl[acc];
  const double* Wb = W.W[acc];

  // (V_l + lambda D_l^2)^-1 and b_l of every landmark of the chunk: one memory round trip for the whole chunk
  for (int i = tid; i < lm_end - lm_begin; i += SCHUR_THREADS) {
    const int l = lm_begin + i;
    const double* Vl = Vb + 6 * (size_t)l;
    double v[6] = {Vl[0], Vl[1], Vl[2], Vl[3], Vl[4], Vl[5]};
    const double b0 = bb[3 * (size_t)l], b1 = bb[3 * (size_t)l + 1], b2 = bb[3 * (size_t)l + 2];
    double vi[6];
    if (opt.marg_mode) {
      pinv3sym_precond(v, vi);   // MarginalizationError::marginalizeOut landmark path (no damping)
    } else {
      double sc[3] = {1.0, 1.0, 1.0};
      if (opt.dogleg) {
        const double* sl = W.lm_scale + 3 * (size_t)l;
        sc[0] = sl[0], sc[1] = sl[1], sc[2] = sl[2];
      }
      v[0] += lambda * damp_diag(v[0], sc[0], opt);
      v[3] += lambda * damp_diag(v[3], sc[1], opt);
      v[5] += lambda * damp_diag(v[5], sc[2], opt);
      inv3sym(v, vi);
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) s_vinv[i][e] = vi[e];
    s_b[i][0] = b0;
    s_b[i][1] = b1;
    s_b[i][2] = b2;
  }

  int ib = 0;
  for (int l0 = lm_begin; l0 < lm_end; l0 += SCHUR_LM_BATCH, ++ib) {
    const int nb = min(SCHUR_LM_BATCH, lm_end - l0);
    const int lc0 = l0 - lm_begin;
    // fill operands of this batch: one work-item per (landmark, block) pair; requested before the tables are
    // zeroed so that the loads are in flight meanwhile
    const int p0 = pb.begin(ib), p1 = pb.begin(ib + 1);
    const int pp = p0 + tid;
    double wp[18];
    int f_slot = 0, f_lb = 0;
    if (pp < p1) {
      const double* Wp = Wb + (size_t)pp * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) wp[i] = Wp[i];
      f_slot = W.pair_off[pp] / 6;
      f_lb = W.pair_lm[pp] - l0;
    }
    // zero the tables (missing (landmark, block) pairs contribute nothing)
    {
      double2* zy = reinterpret_cast<double2*>(s_Y);
      double2* zw = reinterpret_cast<double2*>(s_W);
      for (int i = tid; i < SCHUR_LM_BATCH * trows * 3 / 2; i += SCHUR_THREADS) {
        zy[i] = make_double2(0.0, 0.0);
        zw[i] = make_double2(0.0, 0.0);
      }
    }
    __syncthreads();
    if (l0 == lm_begin) SSTAMP(18);
    auto fill = [&](const double (&w18)[18], int slot, int lb) {
      if (slot >= row0 && slot < row0 + nrow) {
        const double* vi = s_vinv[lc0 + lb];
        const double v0 = vi[0], v1 = vi[1], v2 = vi[2], v3 = vi[3], v4 = vi[4], v5 = vi[5];
        double* y = s_Y + ((size_t)lb * trows + (slot - row0) * 6) * 3;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double w0 = w18[3 * a], w1 = w18[3 * a + 1], w2 = w18[3 * a + 2];
          y[3 * a] = w0 * v0 + w1 * v1 + w2 * v2;
          y[3 * a + 1] = w0 * v1 + w1 * v3 + w2 * v4;
          y[3 * a + 2] = w0 * v2 + w1 * v4 + w2 * v5;
        }
      }
      if (slot >= col0 && slot < col0 + ncol) {
        double* w = s_W + ((size_t)lb * trows + (slot - col0) * 6) * 3;
#pragma unroll
        for (int i = 0; i < 18; ++i) w[i] = w18[i];
      }
    };
    if (pp < p1) fill(wp, f_slot, f_lb);
    for (int q = pp + SCHUR_THREADS; q < p1; q += SCHUR_THREADS) {   // more than 256 pairs in the batch
      const double* Wp = Wb + (size_t)q * 18;
      double w2[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) w2[i] = Wp[i];
      fill(w2, W.pair_off[q] / 6, W.pair_lm[q] - l0);
    }
    __syncthreads();
    if (l0 == lm_begin) SSTAMP(19);
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
            accR[r] += y[3 * r] * s_b[lc0 + lb][0] + y[3 * r + 1] * s_b[lc0 + lb][1] + y[3 * r + 2] * s_b[lc0 + lb][2];
        }
      }
    }
    __syncthreads();
    if (l0 == lm_begin) SSTAMP(20);
  }
  SSTAMP(21);

  // ---- U_pp / g_p (diagonal blocks) and pose x extrinsics cross blocks of this chunk's groups: the host-built
  //      lists are split over the landmark slices (adjacent lanes), so that the slice reduction below sums them
  double accG[6] = {0, 0, 0, 0, 0, 0}, accD[6] = {0, 0, 0, 0, 0, 0};
  const int gbi = row0 + bi, gbj = col0 + bj;  // gbi >= gbj
  if (active) {
    const double* gp = W.gpart[acc];
    if (gbi == gbj) {
      const int lb = W.chunk_diag_begin[chunk * nblk + gbi], le = W.chunk_diag_begin[chunk * nblk + gbi + 1];
      for (int k = lb + slice; k < le; k += nslice) {
        const double* o = gp + W.chunk_diag_out[k];
        double ov[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) ov[i] = o[i];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = 0; c <= r; ++c) accS[6 * r + c] += ov[c * 6 - (c * (c - 1)) / 2 + (r - c)];
          accG[r] += ov[21 + r];
          accD[r] += ov[r * 6 - (r * (r - 1)) / 2];
        }
      }
    } else {
      for (int k = W.chunk_cross_begin[chunk] + slice; k < W.chunk_cross_begin[chunk + 1]; k += nslice) {
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
  }
  SSTAMP(22);
  // ---- combine the slices (adjacent lanes of a quad: DPP, no LDS crossbar) in a fixed order ----
  if (nslice >= 2) {
#pragma unroll
    for (int i = 0; i < 36; ++i) accS[i] += quad_xchg<0xB1>(accS[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      accR[i] += quad_xchg<0xB1>(accR[i]);
      accG[i] += quad_xchg<0xB1>(accG[i]);
      accD[i] += quad_xchg<0xB1>(accD[i]);
    }
  }
  if (nslice == 4) {
#pragma unroll
    for (int i = 0; i < 36; ++i) accS[i] += quad_xchg<0x4E>(accS[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      accR[i] += quad_xchg<0x4E>(accR[i]);
      accG[i] += quad_xchg<0x4E>(accG[i]);
      accD[i] += quad_xchg<0x4E>(accD[i]);
    }
  }
  // ---- write the partial (pose part, row-major block-packed lower triangle | Y b | g | diag U) ----
  if (active && slice == 0) {
    double* sp = W.spart + (size_t)acc * W.spart_buf_stride + (size_t)chunk * W.spart_stride;
    // the 6x6 blocks go out through LDS (the landmark tables are free now) so that the stores are coalesced
    s_boff[pi] = (gbi * (gbi + 1) / 2 + gbj) * 36;
#pragma unroll
    for (int i = 0; i < 36; ++i) sch_smem[pi * 36 + i] = accS[i];
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
  __syncthreads();
  {
    double* sp = W.spart + (size_t)acc * W.spart_buf_stride + (size_t)chunk * W.spart_stride;
    for (int i = tid; i < npairs * 36; i += SCHUR_THREADS) sp[s_boff[i / 36] + (i % 36)] = sch_smem[i];
  }
  SSTAMP(23);
}

// The accept / reject decision for the pending trial, recomputed by wave 0 of every Schur workgroup (bit-identical to the
// solve kernel, ba_device.hpp): which linearisation buffer to reduce, whether anything is to be reduced at all, and the
// regulariser.  s_ctrl / s_dec / s_lambda are LDS; the caller synchronises the workgroup afterwards.
// What wave 0 requests for the decision: the control record (one double per lane) and the per-lane partial sums of BOTH
// linearisation buffers (the record names the one of the pending trial): one memory round trip, issued as early as the caller
// likes (schur_decision_issue) and consumed by schur_decision_finish.
struct SchurDecisionLoads {
  double part0[6], part1[6], cval;
};
__device__ __forceinline__ void schur_decision_issue(const WinPtrs& W, const Ctrl* ctrl, int tid, SchurDecisionLoads& L) {
  L.cval = 0.0;
  if (tid < 64) {
    wave_trial_partials(W, 0, tid, L.part0);
    wave_trial_partials(W, 1, tid, L.part1);
    if (tid < (int)(sizeof(Ctrl) / 8)) L.cval = reinterpret_cast<const double*>(ctrl)[tid];
  }
}
__device__ __forceinline__ void schur_decision_finish(const WinPtrs& W, const OptD& opt, const SchurDecisionLoads& L, int final_call, int bx, int tid,
                                                      Ctrl& s_ctrl, int* s_dec, double& s_lambda) {
  if (tid < 64) {
    const double (&part0)[6] = L.part0;
    const double (&part1)[6] = L.part1;
    if (tid < (int)(sizeof(Ctrl) / 8)) reinterpret_cast<double*>(&s_ctrl)[tid] = L.cval;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
    int acc = s_ctrl.acc, term = 0;
    auto trial_sums = [&](int buf, double sums[6]) {
      double part[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) part[k] = buf ? part1[k] : part0[k];
      wave_trial_reduce(part, sums);
    };
    double lam;
    if (opt.dogleg) {
      // dogleg: the regulariser of the Gauss-Newton solve is mu * diagonal^2; a decision that asks for an explicit
      // dogleg step (rejected step / mis-speculated Gauss-Newton trial) needs no new reduction at all
      double mu = s_ctrl.mu;
      int expl = s_ctrl.explicit_next;
      if (s_ctrl.pending) {
        double sums[6];
        trial_sums(1 - acc, sums);
        DecisionDL d;
        decide_dl_inl(&s_ctrl, &opt, sums, final_call, &d);
        if (bx == 0 && tid == 0) {   // published for the solve kernel (it would compute exactly this)
          auto o = W.dec;
          for (int k = 0; k < 6; ++k) o[DEC_SUMS + k] = sums[k];
          o[DEC_DL + 0] = d.accept; o[DEC_DL + 1] = d.term; o[DEC_DL + 2] = d.explicit_next; o[DEC_DL + 3] = d.judged;
          o[DEC_DL + 4] = d.invalid_steps; o[DEC_DL + 5] = d.have_tot; o[DEC_DL + 6] = d.radius; o[DEC_DL + 7] = d.mu;
          o[DEC_DL + 8] = d.rho; o[DEC_DL + 9] = d.model_change; o[DEC_DL + 10] = d.tot_C; o[DEC_DL + 11] = d.tot_E;
          o[DEC_VALID] = 1.0;
        }
        if (d.accept) acc = 1 - acc;
        term = d.term;
        mu = d.mu;
        expl = d.explicit_next ? (d.judged ? 1 : 2) : 0;
      } else if (!final_call && expl != 2 && s_ctrl.iter >= s_ctrl.max_iter) {
        term = 6;
      }
      if (expl) term = 7;   // nothing to reduce in this slot
      lam = mu;
    } else {
      double radius = s_ctrl.radius;
      if (s_ctrl.pending) {
        double sums[6];
        trial_sums(1 - acc, sums);
        Decision d;
        decide_inl(&s_ctrl, &opt, sums, &d);
        if (bx == 0 && tid == 0) {
          auto o = W.dec;
          for (int k = 0; k < 6; ++k) o[DEC_SUMS + k] = sums[k];
          o[DEC_LM + 0] = d.accept; o[DEC_LM + 1] = d.term; o[DEC_LM + 2] = d.radius; o[DEC_LM + 3] = d.decrease_factor;
          o[DEC_LM + 4] = d.rho; o[DEC_LM + 5] = d.model_change;
          o[DEC_VALID] = 1.0;
        }
        if (d.accept) acc = 1 - acc;
        radius = d.radius;
        term = d.term;
      }
      lam = 1.0 / radius;
    }
    if (tid == 0) {
      s_dec[0] = acc;
      s_dec[1] = term;
      s_lambda = lam;
    }
  }
}

__device__ __forceinline__ void schur_decision(const WinPtrs& W, const OptD& opt, const Ctrl* ctrl, int final_call, int bx, int tid,
                                               Ctrl& s_ctrl, int* s_dec, double& s_lambda) {
  SchurDecisionLoads L;
  schur_decision_issue(W, ctrl, tid, L);
  schur_decision_finish(W, opt, L, final_call, bx, tid, s_ctrl, s_dec, s_lambda);
}

__global__ __launch_bounds__(SCHUR_THREADS) void schur_kernel(const WinPtrs* __restrict__ wins,
                                                              const OptD* __restrict__ optp, int tile_rows, int final_call) {
  const WinPtrs& W = wins[blockIdx.y];
  const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
  const int bx = blockIdx.x;
  if (bx >= W.n_chunk * n_tp) return;
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;

  // dynamic LDS: [SCHUR_LM_BATCH][tile_rows][3] for Y and W, tile_rows = min(96, Dp) of the batch
  extern __shared__ __attribute__((aligned(16))) double sch_smem[];
  __shared__ double s_vinv[SCHUR_CHUNK_LM_MAX][6];   // (V_l + lambda D_l^2)^-1 of every landmark of the chunk
  __shared__ double s_b[SCHUR_CHUNK_LM_MAX][3];
  __shared__ int s_boff[SCHUR_THREADS];               // output offset of block pair pi
  __shared__ int s_dec[2];
  __shared__ double s_lambda;

  const int tid = threadIdx.x;
#undef SSTAMP
#define SSTAMP(k) do { if (W.prof && tid == 0 && bx == 0 && blockIdx.y == 0) W.prof[k] = (double)clock64(); } while (0)
  SSTAMP(16);
  const OptD opt = *optp;
  const int trows = tile_rows;
  // static structure of this workgroup's chunk: requested before the decision (scalar loads, three dependent
  // round trips that overlap with wave 0's reduction instead of following it)
  const int chunk = bx / n_tp;
  const Chunk C = W.chunks[chunk];
  const int lm_begin = W.groups[C.group_begin].lm_begin;
  const int lm_end = W.groups[C.group_end - 1].lm_end;
  const int nbatch = (lm_end - lm_begin + SCHUR_LM_BATCH - 1) / SCHUR_LM_BATCH;
  SchurPairBeginRegs pb;
#pragma unroll
  for (int i = 0; i <= SCHUR_CHUNK_LM_MAX / SCHUR_LM_BATCH; ++i)
    pb.v[i] = (i <= nbatch) ? W.lm_pair_begin[min(lm_begin + i * SCHUR_LM_BATCH, lm_end)] : 0;
  // ---- decision (wave 0) ----
  __shared__ Ctrl s_ctrl;   // the control record is fetched with ONE coalesced load; decide() then reads the LDS copy
  schur_decision(W, opt, ctrl, final_call, bx, tid, s_ctrl, s_dec, s_lambda);
  __syncthreads();
  SSTAMP(17);
  if (s_dec[1]) return;  // terminated by the decision; the solve kernel records it
  schur_reduce_chunk(W, opt, chunk, bx - chunk * n_tp, lm_begin, lm_end, s_dec[0], s_lambda, trows, sch_smem, s_vinv, s_b, s_boff, pb,
                     bx == 0 && blockIdx.y == 0);
}

}  // namespace ba
