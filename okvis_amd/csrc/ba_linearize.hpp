// Kernel 1 — back-substitute + (+)update + re-linearise the reprojection factors (fp64).
//
// One workgroup (256 work-items, 4 waves) owns one *group*: a run of whole landmarks with at most
// GROUP_OBS observations, so every per-landmark and per-(landmark,block) sum is closed inside the
// workgroup and is reduced in LDS in a fixed order (deterministic, no atomics).
//
//   phase A  delta_l = -(V_l + lambda D_l^2)^-1 (b_l + sum_p W_pl^T delta_p)   [from the ACCEPTED buffer]
//            landmark_trial = landmark_acc + delta_l                            (HomogeneousPoint plus)
//   phase B  one work-item per observation: coalesced 32-byte record, poses via L1/L2, residual and the
//            2x15 Jacobian in VGPRs (ReprojectionError.hpp:87-242), Cauchy corrector, staged to LDS
//   phase C  LDS reductions: V_l, b_l, un-robustified H_l (Map::getLhs), W_(block,l), per-block
//            J^T J / J^T r partials, pose-extrinsics cross blocks, cost
//
// HBM traffic per observation: the 32-byte record in, nothing out (W is per pair, not per observation).
//
// REAL = double: the reference's arithmetic.  REAL = float: BASELINE configs[4], "fp32 Jacobian/Hessian build
// with fp64 reduced-camera solve" — phase B evaluates residual and Jacobians in fp32 (reproj_linearize_mixed),
// the staged tiles and every phase-C accumulator are fp32; state, back-substitution (phase A) and everything
// downstream (Schur, solve) stay fp64.
#pragma once
#include <type_traits>

#include "ba_device.hpp"
#include "ba_imu.hpp"

namespace ba {

enum { ST_R = 0, ST_JP = 2, ST_JL = 14, ST_IRHO = 20, ST_COST = 21, ST_JE = 22 };
template <bool EXT, class REAL = double>
struct LinCfg {
  static constexpr int STRIDE = EXT ? 35 : 23;
  // the per-observation stage (REAL) occupies this many doubles of the dynamic LDS
  static constexpr int STAGE_DOUBLES = (GROUP_OBS * STRIDE * (int)sizeof(REAL) + 7) / 8;
  static constexpr int SMEM_DOUBLES = STAGE_DOUBLES + GROUP_LM * 4 + GROUP_PAIRS * 3 + GROUP_LM * 16 + 1024;
};

__device__ __forceinline__ int ut6(int a, int b) { return a * 6 - (a * (a - 1)) / 2 + (b - a); }

// grid.x = n_small + (number of groups): the first n_small = max_imu + 1 workgroups evaluate the IMU / prior
// factors (small_body, ba_imu.hpp; they start first because a re-preintegration is the longest workgroup of
// the launch), the others one linearise group each.  Two workgroups per CU (LDS), hence at most 256 registers.
template <bool EXT, class REAL>
__global__ __launch_bounds__(LIN_THREADS, 2) void linearize_kernel(const WinPtrs* __restrict__ wins,
                                                                   const OptD* __restrict__ optp, int init, int n_small) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const WinPtrs& W = wins[blockIdx.y];
  if ((int)blockIdx.x < n_small) {
    small_body(W, init, blockIdx.x, smem);
    return;
  }
  const int g = blockIdx.x - n_small;
  if (g >= W.n_group) return;
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;
  if (!init && !ctrl->pending) return;  // the solve produced no valid step: nothing to evaluate
  const int acc = ctrl->acc, trial = 1 - acc;
  const double lambda = ctrl->lambda;
  const OptD opt = *optp;

  constexpr int STRIDE = LinCfg<EXT, REAL>::STRIDE;
  REAL* s_stage = reinterpret_cast<REAL*>(smem);
  double* s_lm = smem + LinCfg<EXT, REAL>::STAGE_DOUBLES;
  double* s_pair = s_lm + GROUP_LM * 4;
  double* s_lmres = s_pair + GROUP_PAIRS * 3;
  double* s_step = s_lmres + GROUP_LM * 16;

  const Group G = W.groups[g];
  const int tid = threadIdx.x;
  const int nlm = G.lm_end - G.lm_begin;
  const int nobs = G.obs_end - G.obs_begin;
  const int npair = G.pair_end - G.pair_begin;

  // ------------------------------------------------------------------ phase A: back-substitution
  double sc_gd = 0, sc_ddd = 0, sc_s2 = 0, sc_x2 = 0;
  if (!init) {
    for (int i = tid; i < W.D; i += LIN_THREADS) s_step[i] = W.step[i];
    __syncthreads();
    const double* Wacc = W.W[acc];
    for (int p = tid; p < npair; p += LIN_THREADS) {
      const double* Wp = Wacc + (size_t)(G.pair_begin + p) * 18;
      const double* d = s_step + W.pair_off[G.pair_begin + p];
      double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        t0 += Wp[3 * i] * d[i];
        t1 += Wp[3 * i + 1] * d[i];
        t2 += Wp[3 * i + 2] * d[i];
      }
      s_pair[3 * p] = t0;
      s_pair[3 * p + 1] = t1;
      s_pair[3 * p + 2] = t2;
    }
    __syncthreads();
    if (tid < nlm) {
      const int l = G.lm_begin + tid;
      const double* b = W.bl[acc] + 3 * (size_t)l;
      double t[3] = {b[0], b[1], b[2]};
      for (int p = W.lm_pair_begin[l]; p < W.lm_pair_begin[l + 1]; ++p) {
        const double* sp = s_pair + 3 * (p - G.pair_begin);
        t[0] += sp[0];
        t[1] += sp[1];
        t[2] += sp[2];
      }
      const double* Vl = W.V[acc] + 6 * (size_t)l;
      double v[6] = {Vl[0], Vl[1], Vl[2], Vl[3], Vl[4], Vl[5]};
      const double d0 = clampd(v[0], opt.min_lm_diag2, opt.max_lm_diag2);
      const double d1 = clampd(v[3], opt.min_lm_diag2, opt.max_lm_diag2);
      const double d2 = clampd(v[5], opt.min_lm_diag2, opt.max_lm_diag2);
      v[0] += lambda * d0;
      v[3] += lambda * d1;
      v[5] += lambda * d2;
      double vi[6];
      inv3sym(v, vi);
      const double dl0 = -(vi[0] * t[0] + vi[1] * t[1] + vi[2] * t[2]);
      const double dl1 = -(vi[1] * t[0] + vi[3] * t[1] + vi[4] * t[2]);
      const double dl2 = -(vi[2] * t[0] + vi[4] * t[1] + vi[5] * t[2]);
      const double* x = W.lm[acc] + 4 * (size_t)l;
      const double x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
      double* xt = W.lm[trial] + 4 * (size_t)l;
      const double n0 = x0 + dl0, n1 = x1 + dl1, n2 = x2 + dl2;
      xt[0] = n0; xt[1] = n1; xt[2] = n2; xt[3] = x3;
      s_lm[4 * tid] = n0; s_lm[4 * tid + 1] = n1; s_lm[4 * tid + 2] = n2; s_lm[4 * tid + 3] = x3;
      sc_gd = b[0] * dl0 + b[1] * dl1 + b[2] * dl2;
      sc_ddd = d0 * dl0 * dl0 + d1 * dl1 * dl1 + d2 * dl2 * dl2;
      sc_s2 = dl0 * dl0 + dl1 * dl1 + dl2 * dl2;
      sc_x2 = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
    }
  } else if (tid < nlm) {
    const double* x = W.lm[trial] + 4 * (size_t)(G.lm_begin + tid);
    s_lm[4 * tid] = x[0]; s_lm[4 * tid + 1] = x[1]; s_lm[4 * tid + 2] = x[2]; s_lm[4 * tid + 3] = x[3];
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase B: one observation per lane
  if (tid < nobs) {
    const int o = G.obs_begin + tid;
    const ObsRec rec = W.obs[o];
    const int l = (int)(rec.lm_cam & 0xFFFFFFu);
    const int cam = (int)(rec.lm_cam >> 24);
    const double* pose = W.pose[trial] + 7 * (size_t)rec.pose;
    const double* ext = W.pose[trial] + 7 * (size_t)rec.ext;
    double P[7], E[7], intr[12];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      P[i] = pose[i];
      E[i] = ext[i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) intr[i] = W.cam_intr[12 * cam + i];
    const double* lm = s_lm + 4 * (l - G.lm_begin);
    const double L4[4] = {lm[0], lm[1], lm[2], lm[3]};
    const bool ext_free = EXT && (W.pose_off[rec.ext] >= 0);
    ReprojLinT<REAL> J;
    if constexpr (std::is_same<REAL, double>::value) {
      ReprojLin Jd;
      reproj_linearize(P, E, L4, intr, W.cam_model[cam], rec.u, rec.v, rec.sw, ext_free, &Jd);
      J.r[0] = Jd.r[0];
      J.r[1] = Jd.r[1];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        J.Jp[i] = Jd.Jp[i];
        J.Je[i] = Jd.Je[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) J.Jl[i] = Jd.Jl[i];
    } else {
      reproj_linearize_mixed<REAL>(P, E, L4, intr, W.cam_model[cam], rec.u, rec.v, rec.sw, ext_free, &J);
    }
    if (W.obs_r[trial]) {
      W.obs_r[trial][2 * (size_t)o] = J.r[0];
      W.obs_r[trial][2 * (size_t)o + 1] = J.r[1];
    }
    // Cauchy corrector (Ceres Corrector with rho'' <= 0: scale r and J by sqrt(rho'))
    const REAL s = J.r[0] * J.r[0] + J.r[1] * J.r[1];
    REAL sr = REAL(1), irho = REAL(1), cost = REAL(0.5) * s;
    if (W.cauchy_b > 0) {
      const REAL bb = REAL(W.cauchy_b * W.cauchy_b);
      const REAL sum = REAL(1) + s / bb;
      const REAL rho1 = REAL(1) / sum;
      cost = REAL(0.5) * bb * log(sum);
      sr = sqrt(rho1);
      irho = sum;
    }
    REAL* st = s_stage + (size_t)tid * STRIDE;
    st[ST_R] = sr * J.r[0];
    st[ST_R + 1] = sr * J.r[1];
    const bool pose_free = W.pose_off[rec.pose] >= 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) st[ST_JP + i] = pose_free ? sr * J.Jp[i] : REAL(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) st[ST_JL + i] = sr * J.Jl[i];
    st[ST_IRHO] = irho;
    st[ST_COST] = cost;
    if (EXT) {
#pragma unroll
      for (int i = 0; i < 12; ++i) st[ST_JE + i] = ext_free ? sr * J.Je[i] : REAL(0);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase C: LDS reductions
  // (a) per landmark: V(6) b(3) Hq(6) cost(1)
  for (int wi = tid; wi < nlm * 16; wi += LIN_THREADS) {
    const int ll = wi >> 4, e = wi & 15;
    const int l = G.lm_begin + ll;
    const int o0 = W.lm_obs_begin[l] - G.obs_begin, o1 = W.lm_obs_begin[l + 1] - G.obs_begin;
    REAL a = 0;
    if (e < 6) {
      const int i = (e < 3) ? 0 : (e < 5 ? 1 : 2);
      const int j = (e < 3) ? e : (e < 5 ? e - 2 : 2);
      for (int o = o0; o < o1; ++o) {
        const REAL* st = s_stage + (size_t)o * STRIDE + ST_JL;
        a += st[i] * st[j] + st[3 + i] * st[3 + j];
      }
      W.V[trial][6 * (size_t)l + e] = a;
    } else if (e < 9) {
      const int i = e - 6;
      for (int o = o0; o < o1; ++o) {
        const REAL* st = s_stage + (size_t)o * STRIDE;
        a += st[ST_JL + i] * st[ST_R] + st[ST_JL + 3 + i] * st[ST_R + 1];
      }
      W.bl[trial][3 * (size_t)l + i] = a;
    } else if (e < 15) {
      const int ee = e - 9;
      const int i = (ee < 3) ? 0 : (ee < 5 ? 1 : 2);
      const int j = (ee < 3) ? ee : (ee < 5 ? ee - 2 : 2);
      for (int o = o0; o < o1; ++o) {
        const REAL* st = s_stage + (size_t)o * STRIDE;
        a += (st[ST_JL + i] * st[ST_JL + j] + st[ST_JL + 3 + i] * st[ST_JL + 3 + j]) * st[ST_IRHO];
      }
      W.Hq[trial][6 * (size_t)l + ee] = a;
    } else {
      for (int o = o0; o < o1; ++o) a += s_stage[(size_t)o * STRIDE + ST_COST];
    }
    s_lmres[wi] = a;
  }
  // (b) per (landmark, block) pair: W = sum J_block^T J_l, one work-item per row
  for (int wi = tid; wi < npair * 6; wi += LIN_THREADS) {
    const int pp = wi / 6, a = wi - 6 * pp;
    const int p = G.pair_begin + pp;
    const int jofs = (EXT && W.pair_role[p]) ? ST_JE : ST_JP;
    REAL w0 = 0, w1 = 0, w2 = 0;
    for (int k = W.pair_list_begin[p]; k < W.pair_list_begin[p + 1]; ++k) {
      const REAL* st = s_stage + (size_t)W.pair_list[k] * STRIDE;
      const REAL j0 = st[jofs + a], j1 = st[jofs + 6 + a];
      w0 += j0 * st[ST_JL] + j1 * st[ST_JL + 3];
      w1 += j0 * st[ST_JL + 1] + j1 * st[ST_JL + 4];
      w2 += j0 * st[ST_JL + 2] + j1 * st[ST_JL + 5];
    }
    double* Wt = W.W[trial] + (size_t)p * 18 + 3 * a;
    Wt[0] = w0;
    Wt[1] = w1;
    Wt[2] = w2;
  }
  // (c) per-block J^T J / J^T r partials and pose-extrinsics cross blocks.  These are the long reductions of the
  //     group (every observation of one pose block): four lanes share one (task, row) item, each takes every
  //     fourth observation of the task's list, the partial sums are combined with two xor-shuffles
  //     (8 lanes per item measured slower: 109 vs 101 us on the 64-window launch).
  const int ntask = G.task_end - G.task_begin;
  for (int wi = tid; wi < ntask * 24; wi += LIN_THREADS) {
    const int tt = wi / 24, a = (wi >> 2) % 6, part = wi & 3;
    const Task T = W.tasks[G.task_begin + tt];
    double* out = W.gpart[trial] + T.out;
    REAL acc6[6] = {0, 0, 0, 0, 0, 0};
    REAL ga = 0;
    if (T.type < 2) {
      const int jofs = (T.type == 1) ? ST_JE : ST_JP;
      for (int k = T.list_begin + part; k < T.list_end; k += 4) {
        const REAL* st = s_stage + (size_t)W.task_list[k] * STRIDE;
        const REAL j0 = st[jofs + a], j1 = st[jofs + 6 + a];
#pragma unroll
        for (int b = 0; b < 6; ++b) acc6[b] += j0 * st[jofs + b] + j1 * st[jofs + 6 + b];
        ga += j0 * st[ST_R] + j1 * st[ST_R + 1];
      }
    } else if (EXT) {
      for (int k = T.list_begin + part; k < T.list_end; k += 4) {
        const REAL* st = s_stage + (size_t)W.task_list[k] * STRIDE;
        const REAL j0 = st[ST_JP + a], j1 = st[ST_JP + 6 + a];
#pragma unroll
        for (int b = 0; b < 6; ++b) acc6[b] += j0 * st[ST_JE + b] + j1 * st[ST_JE + 6 + b];
      }
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      acc6[b] += __shfl_xor(acc6[b], 1);
      acc6[b] += __shfl_xor(acc6[b], 2);
    }
    ga += __shfl_xor(ga, 1);
    ga += __shfl_xor(ga, 2);
    if (part == 0) {
      if (T.type < 2) {
        for (int b = a; b < 6; ++b) out[ut6(a, b)] = acc6[b];
        out[21 + a] = ga;
      } else if (EXT) {
        for (int b = 0; b < 6; ++b) out[6 * a + b] = acc6[b];
      }
    }
  }
  __syncthreads();
  // (d) group scalars by wave 0
  if (tid < 64) {
    double cost = 0, gm = 0;
    if (tid < nlm) {
      cost = s_lmres[16 * tid + 15];
      gm = fmax(fabs(s_lmres[16 * tid + 6]), fmax(fabs(s_lmres[16 * tid + 7]), fabs(s_lmres[16 * tid + 8])));
    }
    cost = wave_sum(cost);
    gm = wave_max(gm);
    sc_gd = wave_sum(sc_gd);
    sc_ddd = wave_sum(sc_ddd);
    sc_s2 = wave_sum(sc_s2);
    sc_x2 = wave_sum(sc_x2);
    if (tid == 0) {
      double* gs = W.gscal[trial] + (size_t)g * GS_COUNT;
      gs[GS_COST] = cost;
      gs[GS_GD] = sc_gd;
      gs[GS_DDD] = sc_ddd;
      gs[GS_STEP2] = sc_s2;
      gs[GS_X2] = sc_x2;
      gs[GS_GMAX] = gm;
    }
  }
}

}  // namespace ba
