// Kernel 2 on the fp64 matrix core — the landmark Schur reduction of windows WITHOUT free extrinsics as a GEMM
// (v_mfma_f64_16x16x4_f64); windows with pose x extrinsics cross blocks keep schur_kernel (ba_schur.hpp).
//
//   S~ = [Y; (V^-1 b)^T] W^T ,  Y = W (V_l + lambda D_l^2)^-1     (MarginalizationError.cpp:617-689 / SPARSE_SCHUR, Estimator.cpp:854)
//
// Same decomposition, same output as schur_kernel: a workgroup owns (chunk of landmarks) x (tile pair of <= 16 x 16 pose
// blocks = 96 x 96 rows), lower triangle only, partials in the solve kernel's block-packed layout.  The landmarks of the
// chunk are staged `nlb` at a time as two dense row-major LDS tiles [rows][3 nlb + 1] (contraction index = 3 x landmark +
// coordinate; missing (landmark, block) pairs are zeros; odd row stride: the 16 rows a wave reads per operand fall into 16
// banks), one for the row blocks holding Y — plus, on diagonal tile pairs, one extra row holding V^-1 b, so that Y b falls
// out of the same product — one for the column blocks holding W.  The 16 x 16 output tiles are spread over the four waves and
// stay in accumulator registers across the batches (A lane l = Y[row l & 15][k + (l >> 4)], B lane l = W[col l & 15][k + (l >> 4)]).
// Off-diagonal 6x6 blocks go from the accumulators to global memory; diagonal blocks and Y b meet the chunk's J^T J / J^T r
// partial lists in LDS first (one work-item per entry, list order) and leave as coalesced stores.
//
// 48 KB of LDS and 150 registers: three workgroups per CU (schur_kernel: two, at 222 registers), and the block products of
// a 48-landmark chunk take 1.5 k matrix-core cycles per wave instead of 3 x 3 us of LDS-bound 6x6 block products.
#pragma once
#include "ba_imu.hpp"
#include "ba_schur.hpp"

namespace ba {

// 16x16 output tiles per wave (template argument MAXT): 9 = ceil(36 / 4) for an off-diagonal pair of 96-row tiles; windows whose
// pose part fits 64 rows (Dp <= 63: one diagonal tile pair, 10 tiles) get the instantiation with 3, which fits three workgroups
// per CU without spills
constexpr int SCH2_MAXT_SMALL_ROWS = 64;
typedef double sch2_v4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int sch2_pad16(int n) { return (n + 15) / 16 * 16; }
// landmarks per LDS batch so that both tiles fit `budget` doubles (a multiple of 4: the contraction runs in steps of 4)
__host__ __device__ constexpr int sch2_nlb(int trows, int budget) {
  return 4 * ((budget / (sch2_pad16(trows + 1) + sch2_pad16(trows)) - 1) / 12);
}
__host__ __device__ constexpr int sch2_tile_doubles(int trows, int nlb) { return (sch2_pad16(trows + 1) + sch2_pad16(trows)) * (3 * nlb + 1); }

template <int SCH2_MAXT>
__device__ __forceinline__ void schur_mfma_body(const WinPtrs* __restrict__ wins, const OptD* __restrict__ optp, int tile_rows, int final_call, int nlb,
                                                const CtrlSlot* __restrict__ ctrls, int nodec, const int bx) {
  const WinPtrs& W = wins[blockIdx.y];
  // The control record's address comes from the kernel arguments (CtrlSlot, ba_types.hpp; == W.ctrl): its first words — accepted
  // buffer, pending, first, done — arrive together with the window record, and with them the index of the linearisation buffer
  // that is reduced if the pending trial is accepted (the common case).  Everything the reduction reads first from that buffer
  // — the (landmark, block) rows of the first batch, the landmark blocks V | b — is requested from the speculated buffer next
  // to the decision's own loads instead of one memory round trip behind the decision (round 5; the solve kernel does the same).
  const Ctrl* ctrl = &ctrls[blockIdx.y].c;
  typedef int sch2_i4 __attribute__((ext_vector_type(4)));
  const sch2_i4 chead = *reinterpret_cast<const sch2_i4*>(ctrl);   // acc, pending, first, done
  const int spec = __builtin_amdgcn_readfirstlane(chead.y ? 1 - chead.x : chead.x);
  const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
  if (bx >= W.n_chunk * n_tp) return;

  extern __shared__ __attribute__((aligned(16))) double sch_smem[];   // tY [pad16(tile_rows + 1)][3 nlb + 1] | tW [pad16(tile_rows)][3 nlb + 1]
  __shared__ double s_vinv[SCHUR_CHUNK_LM_MAX][6];   // (V_l + lambda D_l^2)^-1 of every landmark of the chunk
  __shared__ double s_vb[SCHUR_CHUNK_LM_MAX][3];     // V^-1 b
  __shared__ double s_diag[SCHUR_TILE_BLOCKS][36];   // diagonal 6x6 blocks of Y W^T (lower triangle)
  __shared__ double s_yb[TILE_DIM];
  __shared__ int s_dec[2];
  __shared__ double s_lambda;
  __shared__ Ctrl s_ctrl;
  __shared__ int2 s_tb[SCH2_MAXT > 3 ? 2 * SCHUR_CHUNK_LM_MAX : 2];   // several tiles per dimension: pair range of (landmark, row | column tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#undef SSTAMP
#define SSTAMP(k) do { if (W.prof && tid == 0 && bx == 0 && blockIdx.y == 0) W.prof[k] = (double)clock64(); } while (0)
  SSTAMP(16);
  // wave 0's loads for the decision go out first; the static structure below is fetched while they are in flight
  // nodec (batches that keep one set of partials per linearisation buffer, okvis_ba_upload: DOGLEG and fixed-radius runs behind a
  // separate Schur launch): no decision at the head.  The launch reduces what the fused linearise launch would reduce — the trial
  // buffer if a trial is pending, with the damping the next solve will use if the trial is accepted (it does not depend on the
  // costs in these modes), else the accepted buffer — into that buffer's own set of partials; the solve kernel takes the
  // decision, sums the set of the buffer it accepts and, after a rejection, has the other set from the launch that reduced it.
  // The trust-region decision with its two dependent loads and its barrier (5.7 of a workgroup's 29 us, in every workgroup)
  // leaves the chain, and nothing in this launch waits for the costs of the IMU / prior factors any more.
  SchurDecisionLoads dl;
  if (!nodec) schur_decision_issue(W, ctrl, tid, dl);
  const double c_mu = nodec ? ctrl->mu : 0.0, c_radius = nodec ? ctrl->radius : 1.0;
  const OptD opt = *optp;
  const int chunk = bx / n_tp;
  const int* cd = W.chunk_desc + (size_t)chunk * SCHUR_DESC_INTS;   // lm_begin, lm_end, pair ranges: one record
  const int lm_begin = cd[0], lm_end = cd[1];
  const int nl = lm_end - lm_begin;

  // ---- static structure of this workgroup, requested before the decision so that its round trips overlap with wave 0's ----
  int tp = bx - chunk * n_tp, ti = 0;
  while (tp >= ti + 1) {   // lower-triangular enumeration of the tile pairs: (0,0) (1,0) (1,1) (2,0) ...
    tp -= ti + 1;
    ++ti;
  }
  const int tj = tp;
  const int nblk = W.Dp / 6, Dp = W.Dp;
  const int row0 = ti * SCHUR_TILE_BLOCKS, col0 = tj * SCHUR_TILE_BLOCKS;
  const int nrow = min(SCHUR_TILE_BLOCKS, nblk - row0), ncol = min(SCHUR_TILE_BLOCKS, nblk - col0);
  const bool diag = ti == tj;
  const int rowsA = 6 * nrow + (diag ? 1 : 0), rowsB = 6 * ncol;
  const int RT = (rowsA + 15) / 16, CT = (rowsB + 15) / 16;
  const int K = 3 * nlb, KP = K + 1;
  double* tY = sch_smem;
  double* tW = sch_smem + (size_t)sch2_pad16(tile_rows + 1) * KP;
  // the per-group J^T J / J^T r partials (task outputs) that sum into the diagonal blocks of this tile pair: work-item e <
  // 36 nrow owns entry e of the blocks (up to SCH2_DI items), work-item r < 6 nrow row r of Y b | g | diag U.  Their list
  // heads (up to SCH2_LL entries; longer lists are walked in global memory) are static structure
  constexpr int SCH2_DI = (SCHUR_TILE_BLOCKS * 36 + SCHUR_THREADS - 1) / SCHUR_THREADS, SCH2_LL = 6;
  int d_q0[SCH2_DI], d_n[SCH2_DI], d_out[SCH2_DI][SCH2_LL];
  int r_q0 = 0, r_n = 0, r_out[SCH2_LL];
#pragma unroll
  for (int q = 0; q < SCH2_LL; ++q) r_out[q] = 0;
#pragma unroll
  for (int it = 0; it < SCH2_DI; ++it) {
    d_q0[it] = d_n[it] = 0;
#pragma unroll
    for (int q = 0; q < SCH2_LL; ++q) d_out[it][q] = 0;
  }
  if (diag) {
    const int* lb_ = W.chunk_diag_begin + (size_t)chunk * nblk + row0;
#pragma unroll
    for (int it = 0; it < SCH2_DI; ++it) {
      const int e = tid + it * SCHUR_THREADS;
      if (e < nrow * 36) {
        d_q0[it] = lb_[e / 36];
        d_n[it] = lb_[e / 36 + 1] - d_q0[it];
      }
    }
    if (tid < 6 * nrow) {
      r_q0 = lb_[tid / 6];
      r_n = lb_[tid / 6 + 1] - r_q0;
    }
#pragma unroll
    for (int it = 0; it < SCH2_DI; ++it)
#pragma unroll
      for (int q = 0; q < SCH2_LL; ++q)
        if (q < d_n[it]) d_out[it][q] = W.chunk_diag_out[d_q0[it] + q];
#pragma unroll
    for (int q = 0; q < SCH2_LL; ++q)
      if (q < r_n) r_out[q] = W.chunk_diag_out[r_q0 + q];
  }
  const bool tiled = SCH2_MAXT > 3 && W.n_tile > 1;
  if constexpr (SCH2_MAXT > 3) {
    if (tiled && tid < 2 * nl) {   // (visible behind the barrier of the decision)
      const int* tb = W.lm_tile_begin + (size_t)(lm_begin + (tid >> 1)) * (W.n_tile + 1) + ((tid & 1) ? tj : ti);
      s_tb[tid] = make_int2(tb[0], tb[1]);
    }
  }
  // pair ranges of the landmark batches (uniform)
  constexpr int SCH2_NB = SCHUR_CHUNK_LM_MAX / 4;   // batches of >= 4 landmarks
  int pbv[SCH2_NB + 1];
#pragma unroll
  for (int ib = 0; ib <= SCH2_NB; ++ib) pbv[ib] = cd[2 + min(ib * (nlb / 4), SCH2_NB)];
  auto pb_at = [&](int ib) {
    int r = 0;
#pragma unroll
    for (int q = 0; q <= SCH2_NB; ++q)   // static indexing keeps the values in (scalar) registers
      if (q == ib) r = pbv[q];
    return r;
  };

  // ---- what the reduction reads first, from the speculated buffer: the (landmark, block) rows of the first batch and the
  //      landmark blocks (one landmark per work-item: a chunk has at most SCHUR_CHUNK_LM_MAX <= SCHUR_THREADS of them) ----
  const double* Wb = W.W[spec];
  double wp[18];
  int f_slot = -1, f_lb = 0;
  auto load_batch = [&](int ib) {   // one work-item per (landmark, block) pair of batch ib
    const int pp = pb_at(ib) + tid;
    f_slot = -1;
    if (!tiled && pp < pb_at(ib + 1)) {
      const double* Wp = Wb + (size_t)pp * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) wp[i] = Wp[i];
      f_slot = W.pair_off[pp] / 6;
      f_lb = W.pair_lm[pp] - (lm_begin + ib * nlb);
    }
  };
  static_assert(SCHUR_CHUNK_LM_MAX <= SCHUR_THREADS, "one landmark per work-item");
  double lv[6] = {1, 0, 0, 1, 0, 1}, lb3[3] = {0, 0, 0};
  auto load_landmark = [&](int buf) {
    if (tid < nl) {
      const double* Vl = W.V[buf] + 6 * (size_t)(lm_begin + tid);
      const double* bl = W.bl[buf] + 3 * (size_t)(lm_begin + tid);
#pragma unroll
      for (int e = 0; e < 6; ++e) lv[e] = Vl[e];
      lb3[0] = bl[0], lb3[1] = bl[1], lb3[2] = bl[2];
    }
  };
  load_batch(0);
  load_landmark(spec);
  if (!nodec) schur_decision_finish(W, opt, dl, final_call, bx, tid, s_ctrl, s_dec, s_lambda);
  __syncthreads();
  SSTAMP(17);
  if (nodec ? chead.w != 0 : (s_ctrl.done || s_dec[1])) return;   // finished earlier / terminated by the decision (the solve kernel records it)
  const int acc = nodec ? spec : __builtin_amdgcn_readfirstlane(s_dec[0]);
  // (nodec: the rule of the fused linearise launch, ba_linearize2.hpp — the damping of the solve that follows an accepted trial)
  const double lam_next = opt.dogleg ? ((chead.z || opt.gauss_newton) ? c_mu : fmax(DL_MIN_MU, 2.0 * c_mu / DL_MU_INCREASE)) : 1.0 / c_radius;
  const double lambda = nodec ? (chead.y ? lam_next : (opt.dogleg ? c_mu : 1.0 / c_radius)) : s_lambda;
  if (acc != spec) {   // (a rejected trial: the other buffer is the one to reduce — requested again)
    Wb = W.W[acc];
    load_batch(0);
    load_landmark(acc);
  }
  const double* gp = W.gpart[acc];
  double d_u[SCH2_DI], r_g = 0.0, r_du = 0.0;
  if (diag) {
#pragma unroll
    for (int it = 0; it < SCH2_DI; ++it) {
      d_u[it] = 0.0;
      const int e = tid + it * SCHUR_THREADS;
      if (e < nrow * 36) {
        const int k = e % 36, ii = k / 6, jj = k - 6 * ii;
        if (jj <= ii) {   // U_pp: lower triangle, in list order
          const int uk = jj * 6 - (jj * (jj - 1)) / 2 + (ii - jj);
          double v[SCH2_LL];
#pragma unroll
          for (int q = 0; q < SCH2_LL; ++q) v[q] = q < d_n[it] ? gp[d_out[it][q] + uk] : 0.0;
#pragma unroll
          for (int q = 0; q < SCH2_LL; ++q) d_u[it] += v[q];
          for (int q = SCH2_LL; q < d_n[it]; ++q) d_u[it] += gp[W.chunk_diag_out[d_q0[it] + q] + uk];
        }
      }
    }
    if (tid < 6 * nrow) {
      const int a = tid % 6, dk = a * 6 - (a * (a - 1)) / 2;
      double vg[SCH2_LL], vd[SCH2_LL];
#pragma unroll
      for (int q = 0; q < SCH2_LL; ++q) {
        vg[q] = q < r_n ? gp[r_out[q] + 21 + a] : 0.0;
        vd[q] = q < r_n ? gp[r_out[q] + dk] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < SCH2_LL; ++q) r_g += vg[q], r_du += vd[q];
      for (int q = SCH2_LL; q < r_n; ++q) {
        const double* o = gp + W.chunk_diag_out[r_q0 + q];
        r_g += o[21 + a];
        r_du += o[dk];
      }
    }
  }
  // (V_l + lambda D_l^2)^-1 and V^-1 b of every landmark of the chunk (work-item i: landmark i, its blocks are in registers)
  if (tid < nl) {
    const int i = tid;
    const int l = lm_begin + i;
    double v[6] = {lv[0], lv[1], lv[2], lv[3], lv[4], lv[5]};
    const double b0 = lb3[0], b1 = lb3[1], b2 = lb3[2];
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
    s_vb[i][0] = vi[0] * b0 + vi[1] * b1 + vi[2] * b2;
    s_vb[i][1] = vi[1] * b0 + vi[3] * b1 + vi[4] * b2;
    s_vb[i][2] = vi[2] * b0 + vi[4] * b1 + vi[5] * b2;
  }

  // ---- my output tiles: tt = wave + 4 t; diagonal tile pair: tile (I, J) with J <= min(I, CT - 1), else all RT x CT ----
  int ntiles;
  if (diag) {
    ntiles = 0;
    for (int I = 0; I < RT; ++I) ntiles += min(I, CT - 1) + 1;
  } else {
    ntiles = RT * CT;
  }
  auto tile_of = [&](int tt, int& I, int& J) {
    if (diag) {
      I = 0;
      int rem = tt;
      while (rem > min(I, CT - 1)) {
        rem -= min(I, CT - 1) + 1;
        ++I;
      }
      J = rem;
    } else {
      I = tt / CT;
      J = tt - I * CT;
    }
  };
  sch2_v4 accv[SCH2_MAXT];
#pragma unroll
  for (int t = 0; t < SCH2_MAXT; ++t) accv[t] = sch2_v4{0.0, 0.0, 0.0, 0.0};

  int ib = 0;
  for (int l0 = lm_begin; l0 < lm_end; l0 += nlb, ++ib) {
    const int nb = min(nlb, lm_end - l0);
    const int p0 = pb_at(ib), p1 = pb_at(ib + 1);
    {
      double2* z = reinterpret_cast<double2*>(sch_smem);
      const int n2 = (sch2_pad16(tile_rows + 1) + sch2_pad16(tile_rows)) * KP / 2;
      for (int i = tid; i < n2; i += SCHUR_THREADS) z[i] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (l0 == lm_begin) SSTAMP(18);
    auto fill = [&](const double (&w18)[18], int slot, int lb) {
      if (slot >= row0 && slot < row0 + nrow) {
        const double* vi = s_vinv[l0 - lm_begin + lb];
        const double v0 = vi[0], v1 = vi[1], v2 = vi[2], v3 = vi[3], v4 = vi[4], v5 = vi[5];
        double* y = tY + (size_t)(slot - row0) * 6 * KP + 3 * lb;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double w0 = w18[3 * a], w1 = w18[3 * a + 1], w2 = w18[3 * a + 2];
          y[a * KP] = w0 * v0 + w1 * v1 + w2 * v2;
          y[a * KP + 1] = w0 * v1 + w1 * v3 + w2 * v4;
          y[a * KP + 2] = w0 * v2 + w1 * v4 + w2 * v5;
        }
      }
      if (slot >= col0 && slot < col0 + ncol) {
        double* w = tW + (size_t)(slot - col0) * 6 * KP + 3 * lb;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          w[a * KP] = w18[3 * a];
          w[a * KP + 1] = w18[3 * a + 1];
          w[a * KP + 2] = w18[3 * a + 2];
        }
      }
    };
    if (f_slot >= 0) fill(wp, f_slot, f_lb);
    if constexpr (SCH2_MAXT > 3) {
      if (tiled) {
        // only the pairs of this tile pair: item = (landmark of the batch, row | column tile, block of the tile)
        for (int it = tid; it < nb * 32; it += SCHUR_THREADS) {
          const int lb = it >> 5, side = (it >> 4) & 1;
          if (diag && side) continue;   // (row tile = column tile: the row side fills both operands)
          const int2 r = s_tb[2 * (l0 - lm_begin + lb) + side];
          const int q = r.x + (it & 15);
          if (q < r.y) {
            const double* Wp = Wb + (size_t)q * 18;
            double w2[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) w2[i] = Wp[i];
            fill(w2, W.pair_off[q] / 6, lb);
          }
        }
      }
    }
    for (int q = p0 + tid + SCHUR_THREADS; !tiled && q < p1; q += SCHUR_THREADS) {   // more than 256 pairs in the batch
      const double* Wp = Wb + (size_t)q * 18;
      double w2[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) w2[i] = Wp[i];
      fill(w2, W.pair_off[q] / 6, W.pair_lm[q] - l0);
    }
    if (diag && tid < nb) {   // row 6 nrow of the row tile: V^-1 b (gives Y b = W V^-1 b in that row of the product)
      double* y = tY + (size_t)(6 * nrow) * KP + 3 * tid;
      const double* vb = s_vb[l0 - lm_begin + tid];
      y[0] = vb[0], y[1] = vb[1], y[2] = vb[2];
    }
    // the next batch's rows are requested now: in flight during the products
    if (l0 + nlb < lm_end) load_batch(ib + 1);
    __syncthreads();
    if (l0 == lm_begin) SSTAMP(19);
#pragma unroll
    for (int t = 0; t < SCH2_MAXT; ++t) {
      const int tt = wave + 4 * t;
      if (tt < ntiles) {
        int I, J;
        tile_of(tt, I, J);
        const double* pa = tY + (size_t)(16 * I + (lane & 15)) * KP + (lane >> 4);
        const double* pb = tW + (size_t)(16 * J + (lane & 15)) * KP + (lane >> 4);
        sch2_v4 a = accv[t];
        for (int k = 0; k < K; k += 12) {   // (K = 3 nlb is a multiple of 12) six operands in flight, then three products
          const double a0 = pa[k], b0 = pb[k], a1 = pa[k + 4], b1 = pb[k + 4], a2 = pa[k + 8], b2 = pb[k + 8];
          a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, a, 0, 0, 0);
        }
        accv[t] = a;
      }
    }
    if (l0 + nlb < lm_end) __syncthreads();
    if (l0 == lm_begin) SSTAMP(20);
  }
  SSTAMP(21);

  double* sp = W.spart + (size_t)acc * W.spart_buf_stride + (size_t)chunk * W.spart_stride;
  double* sr = sp + (size_t)(nblk * (nblk + 1) / 2) * 36;
  if constexpr (SCH2_MAXT <= 3) {
    // ---- out (pose part within 64 rows): the product leaves the accumulators as a dense matrix in LDS (the tiles are done
    //      with), then every entry of the block-packed partial is one work-item: coalesced stores, no branch per value ----
    __syncthreads();
    double* dn = sch_smem;
    const int DS = 16 * CT + 1;
#pragma unroll
    for (int t = 0; t < SCH2_MAXT; ++t) {
      const int tt = wave + 4 * t;
      if (tt < ntiles) {
        int I, J;
        tile_of(tt, I, J);
#pragma unroll
        for (int r = 0; r < 4; ++r) dn[(16 * I + (lane >> 4) + 4 * r) * DS + 16 * J + (lane & 15)] = accv[t][r];
      }
    }
    if (diag) {   // U_pp of the diagonal blocks (lower triangle; zero above)
#pragma unroll
      for (int it = 0; it < SCH2_DI; ++it) {
        const int e = tid + it * SCHUR_THREADS;
        if (e < nrow * 36) s_diag[e / 36][e % 36] = d_u[it];
      }
    }
    __syncthreads();
    SSTAMP(22);
    const int npairs = diag ? nrow * (nrow + 1) / 2 : nrow * ncol;
    for (int e = tid; e < npairs * 36; e += SCHUR_THREADS) {
      const int pi = e / 36, k = e - 36 * pi, ii = k / 6, jj = k - 6 * ii;
      int bi, bj;
      if (diag) {
        bi = (int)((sqrtf(8.0f * pi + 1.0f) - 1.0f) * 0.5f);
        while ((bi + 1) * (bi + 2) / 2 <= pi) ++bi;
        while (bi * (bi + 1) / 2 > pi) --bi;
        bj = pi - bi * (bi + 1) / 2;
      } else {
        bi = pi / ncol;
        bj = pi - bi * ncol;
      }
      const bool dblk = diag && bi == bj;
      const bool upper = dblk && jj > ii;   // Y W^T is symmetric: the upper part of a diagonal block mirrors the lower one
      const double yw = dn[(6 * bi + (upper ? jj : ii)) * DS + 6 * bj + (upper ? ii : jj)];
      const double u = dblk ? s_diag[bi][k] : 0.0;
      const int gbi = row0 + bi, gbj = col0 + bj;
      sp[(size_t)(gbi * (gbi + 1) / 2 + gbj) * 36 + k] = u - yw;
    }
    if (diag && tid < 6 * nrow) {
      const int gr = row0 * 6 + tid;
      sr[gr] = dn[(6 * nrow) * DS + tid];
      sr[Dp + gr] = r_g;
      sr[2 * Dp + gr] = r_du;
    }
  } else {
    // ---- out: off-diagonal 6x6 blocks straight from the accumulators; diagonal blocks and the Y b row through LDS ----
  #pragma unroll
    for (int t = 0; t < SCH2_MAXT; ++t) {
      const int tt = wave + 4 * t;
      if (tt < ntiles) {
        int I, J;
        tile_of(tt, I, J);
        const int j = 16 * J + (lane & 15);
        const int bj = j / 6, jj = j - 6 * bj;
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + (lane >> 4) + 4 * r;
          const double v = accv[t][r];
          if (j >= rowsB) continue;
          if (diag && i == 6 * nrow) {
            s_yb[j] = v;
          } else if (i < 6 * nrow) {
            const int bi = i / 6, ii = i - 6 * bi;
            if (diag && bi == bj) {
              if (jj <= ii) s_diag[bi][6 * ii + jj] = v;
            } else if (!diag || bj < bi) {
              const int gbi = row0 + bi, gbj = col0 + bj;
              sp[(size_t)(gbi * (gbi + 1) / 2 + gbj) * 36 + 6 * ii + jj] = -v;
            }
          }
        }
      }
    }
    SSTAMP(22);
    if (diag) {   // (uniform per workgroup)
      __syncthreads();
      // diagonal blocks: U_pp (lower triangle) - Y W^T (symmetric); Y b | g | diag U of the row blocks
  #pragma unroll
      for (int it = 0; it < SCH2_DI; ++it) {
        const int e = tid + it * SCHUR_THREADS;
        if (e < nrow * 36) {
          const int bi = e / 36, k = e - 36 * bi, ii = k / 6, jj = k - 6 * ii;
          const double yw = jj <= ii ? s_diag[bi][6 * ii + jj] : s_diag[bi][6 * jj + ii];
          const int gbi = row0 + bi;
          sp[(size_t)(gbi * (gbi + 1) / 2 + gbi) * 36 + k] = d_u[it] - yw;
        }
      }
      if (tid < 6 * nrow) {
        const int gr = row0 * 6 + tid;
        sr[gr] = s_yb[tid];
        sr[Dp + gr] = r_g;
        sr[2 * Dp + gr] = r_du;
      }
    }
  }
  SSTAMP(23);
#undef SSTAMP
}


template <int SCH2_MAXT>
__global__ __launch_bounds__(SCHUR_THREADS, SCH2_MAXT <= 3 ? 3 : 2) void schur_mfma_kernel(const WinPtrs* __restrict__ wins, const OptD* __restrict__ optp,
                                                                      int tile_rows, int final_call, int nlb, const CtrlSlot* __restrict__ ctrls, int nodec) {
  schur_mfma_body<SCH2_MAXT>(wins, optp, tile_rows, final_call, nlb, ctrls, nodec, (int)blockIdx.x);
}

// The decision-free Schur launch with the EVALUATION of the IMU / prior factors of the same trial riding along (round 6): workgroups
// 0 .. n_small - 1 of a window are the second half of the factor workgroups (imu_factor<2>, small_factors: 14 KB of LDS, no
// re-preintegration code — the records were brought up to date by small_prepare_kernel right behind the solve launch), the others the
// Schur workgroups.  Neither kind waits for the other — the reduction does not need the factors' costs when it takes no decision —
// and both only have to be through before the next solve launch.  (Round 5 let the WHOLE factor workgroups ride: 255 registers and
// 62 KB of LDS of the re-preintegration path put the launch at two workgroups per CU and cost the line 6 %; this half keeps three.)
template <int SCH2_MAXT>
__global__ __launch_bounds__(SCHUR_THREADS, 3) void schur_ride_kernel(const WinPtrs* __restrict__ wins, const OptD* __restrict__ optp, int tile_rows,
                                                                      int final_call, int nlb, const CtrlSlot* __restrict__ ctrls, int nodec, int n_small) {
  static_assert(SCHUR_THREADS == IMU_THREADS && SCHUR_THREADS == LIN_THREADS, "one block size for both kinds of workgroup");
  if ((int)blockIdx.x < n_small) {
    extern __shared__ __attribute__((aligned(16))) double sch_smem[];
    small_body<2>(wins[blockIdx.y], 0, (int)blockIdx.x, sch_smem);
    return;
  }
  schur_mfma_body<SCH2_MAXT>(wins, optp, tile_rows, final_call, nlb, ctrls, nodec, (int)blockIdx.x - n_small);
}

}  // namespace ba
