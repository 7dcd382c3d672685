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
#include "ba_schur.hpp"

namespace ba {

enum { ST_R = 0, ST_JP = 2, ST_JL = 14, ST_IRHO = 20, ST_COST = 21, ST_JE = 22 };
template <bool EXT, class REAL = double>
struct LinCfg {
  static constexpr int STRIDE = EXT ? 35 : 23;
  // the per-observation stage (REAL) occupies this many doubles of the dynamic LDS
  static constexpr int STAGE_DOUBLES = (GROUP_OBS * STRIDE * (int)sizeof(REAL) + 7) / 8;
  static constexpr int SMEM_DOUBLES = STAGE_DOUBLES + GROUP_LM * 4 + GROUP_PAIRS * 3 + GROUP_LM * 16 + 1024;
};

constexpr int LIN_TASK_CACHE = 64;   // Task records of a group kept in LDS (more: read from global memory)
static_assert((GROUP_PAIRS + 1 + GROUP_LM + 1 + LIN_TASK_CACHE * 6) * 4 + (2 + 3) * GROUP_OBS * 2 + 3 * GROUP_PAIRS <=
                  GROUP_PAIRS * 3 * 8, "index cache must fit the s_pair area");
static_assert(sizeof(Task) == 24, "Task is cached as 6 ints");

__device__ __forceinline__ int ut6(int a, int b) { return a * 6 - (a * (a - 1)) / 2 + (b - a); }

// ---- fused mode, fast path (WinPtrs::fuse_fast; no free extrinsics, <= FUSE_MAX_TASKS reduction tasks and <= FUSE_WIT x
// LIN_THREADS (pair, row) items per group): the group's part of the reduced camera system straight from what phase C has in
// LDS and registers, on the fp64 matrix core.
//   S~ = [Y; (V^-1 b)^T] W^T  with  Y = W V^-1  (rows = reduced pose offsets, contraction index = 3 x landmark + coordinate)
// Y and W are dense row-major tiles [R][K + 1] in the dead observation stage (missing (landmark, block) pairs are zeros), a
// lower-triangular grid of 16x16 output tiles is spread over the four waves (v_mfma_f64_16x16x4_f64: A lane l = Y[row l & 15]
// [k l >> 4], B lane l = W[col l & 15][k l >> 4]), landmarks in batches of FuseCfg::nlb; the partial leaves in the block-packed
// layout of the Schur kernel with the group's own J^T J / J^T r blocks (phase C(c), kept in LDS) added on the way out.
constexpr int FUSE_WIT = 4;         // (pair, row) items of phase C(b) one work-item keeps in registers
constexpr int FUSE_MAX_TASKS = 12;  // J^T J / J^T r blocks of one group kept in LDS
typedef double lin_v4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int fuse_rows(int Dp) { return ((Dp + 1 + 15) / 16) * 16; }
// landmarks per batch so that two [rows][3 nlb + 1] tiles fit `stage` doubles (a multiple of 4: 3 nlb is a multiple of 4)
__host__ __device__ constexpr int fuse_nlb(int Dp, int stage) { return 4 * (((stage / (2 * fuse_rows(Dp))) - 1) / 12); }

template <int NQ>
struct FuseItemsT {     // what phase C(b) hands over: W rows in registers
  double w[NQ][3];
  int row[NQ];          // reduced offset of the row (pose part), -1 = none
  int lm[NQ];           // landmark of the row, group-local
};
typedef FuseItemsT<FUSE_WIT> FuseItems;

template <int STAGE_DOUBLES, int NQ>
__device__ __forceinline__ void fused_reduce_fast(const WinPtrs& W, const OptD& opt, int g, int buf, double lam, int nlm, bool init,
                                                  const double (&pf_sc)[3], const FuseItemsT<NQ>& it, double* tiles,
                                                  const double* s_lmres, double* aux) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define FSTAMP(k) do { if (W.prof && tid == 0 && g == 0 && blockIdx.y == 0) W.prof[k] = (double)clock64(); } while (0)
  const int Dp = W.Dp, R = fuse_rows(Dp), RT = R / 16;
  const int nlb = fuse_nlb(Dp, STAGE_DOUBLES), K = 3 * nlb, KP = K + 1;
  double* tY = tiles;
  double* tW = tiles + (size_t)R * KP;
  double(*vinv)[6] = reinterpret_cast<double(*)[6]>(aux);
  double(*vb)[3] = reinterpret_cast<double(*)[3]>(aux + GROUP_LM * 6);
  const double* s_U = aux + GROUP_LM * 9;
  const int* blktask = reinterpret_cast<const int*>(aux + GROUP_LM * 9 + FUSE_MAX_TASKS * 36);   // filled when the index lists were parked
  // ---- (V_l + lambda D_l^2)^-1 and V^-1 b per landmark (one work-item each, wave 1: wave 0 is still writing the group scalars)
  const int lt = tid - 64;   // landmark of this work-item
  if (lt >= 0 && lt < nlm) {
    const double* r = s_lmres + 16 * lt;
    double v[6] = {r[0], r[1], r[2], r[3], r[4], r[5]}, vi[6];
    if (opt.marg_mode) {
      pinv3sym_precond(v, vi);
    } else {
      double sc[3] = {1.0, 1.0, 1.0};
      if (opt.dogleg) {
        if (init) {   // the scale estimated from this very linearisation (phase C(a) has stored the same expression)
          sc[0] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(v[0])) : 1.0;
          sc[1] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(v[3])) : 1.0;
          sc[2] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(v[5])) : 1.0;
        } else {
          sc[0] = pf_sc[0], sc[1] = pf_sc[1], sc[2] = pf_sc[2];
        }
      }
      v[0] += lam * damp_diag(v[0], sc[0], opt);
      v[3] += lam * damp_diag(v[3], sc[1], opt);
      v[5] += lam * damp_diag(v[5], sc[2], opt);
      inv3sym(v, vi);
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) vinv[lt][e] = vi[e];
    vb[lt][0] = vi[0] * r[6] + vi[1] * r[7] + vi[2] * r[8];
    vb[lt][1] = vi[1] * r[6] + vi[3] * r[7] + vi[4] * r[8];
    vb[lt][2] = vi[2] * r[6] + vi[4] * r[7] + vi[5] * r[8];
  }
  constexpr int MAXT = 7;   // output tiles per wave: 7 x 4 >= 28 = the lower triangle of 7 x 7 tiles (R <= 112)
  lin_v4 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = lin_v4{0.0, 0.0, 0.0, 0.0};
  const int ntiles = RT * (RT + 1) / 2;
  // ---- tiles and products, one landmark batch at a time
  for (int l0 = 0; l0 < nlm; l0 += nlb) {
    {
      double2* z = reinterpret_cast<double2*>(tiles);
      const int n2 = R * KP;   // 2 R KP doubles
      for (int i = tid; i < n2; i += LIN_THREADS) z[i] = make_double2(0.0, 0.0);
    }
    if (l0 == 0) FSTAMP(18);
    __syncthreads();
    if (l0 == 0) FSTAMP(19);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int lb = it.lm[q] - l0;
      if (it.row[q] >= 0 && lb >= 0 && lb < nlb) {
        const double w0 = it.w[q][0], w1 = it.w[q][1], w2 = it.w[q][2];
        const double* vi = vinv[it.lm[q]];
        double* y = tY + (size_t)it.row[q] * KP + 3 * lb;
        double* w = tW + (size_t)it.row[q] * KP + 3 * lb;
        y[0] = w0 * vi[0] + w1 * vi[1] + w2 * vi[2];
        y[1] = w0 * vi[1] + w1 * vi[3] + w2 * vi[4];
        y[2] = w0 * vi[2] + w1 * vi[4] + w2 * vi[5];
        w[0] = w0, w[1] = w1, w[2] = w2;
      }
    }
    if (lt >= l0 && lt < min(nlm, l0 + nlb)) {   // row Dp of Y: V^-1 b (gives Y b = W V^-1 b in row Dp of the product)
      double* y = tY + (size_t)Dp * KP + 3 * (lt - l0);
      y[0] = vb[lt][0], y[1] = vb[lt][1], y[2] = vb[lt][2];
    }
    __syncthreads();
    if (l0 == 0) FSTAMP(20);
    {
      int I = 0, rem = wave;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int tt = wave + 4 * t;
        if (tt < ntiles) {
          while (rem > I) {
            rem -= I + 1;
            ++I;
          }
          const int J = rem;
          const double* pa = tY + (size_t)(16 * I + (lane & 15)) * KP + (lane >> 4);
          const double* pb = tW + (size_t)(16 * J + (lane & 15)) * KP + (lane >> 4);
          lin_v4 a = acc[t];
          for (int k = 0; k < K; k += 4) a = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], pb[k], a, 0, 0, 0);
          acc[t] = a;
          rem += 4;
        }
      }
    }
    if (l0 + nlb < nlm) __syncthreads();
  }
  FSTAMP(21);
  // ---- out: block-packed lower triangle | Y b | g | diag U.  Off-diagonal 6x6 blocks go straight from the accumulators to
  //      global memory; the diagonal ones meet the group's own J^T J blocks in LDS first (the inverse landmark blocks are no
  //      longer needed: every wave is past the barrier behind the last fill) and leave as whole blocks
  const int nblk = Dp / 6;
  double* sp = W.spart + (size_t)buf * W.spart_buf_stride + (size_t)g * W.spart_stride;
  double* sr = sp + (size_t)(nblk * (nblk + 1) / 2) * 36;
  double* s_diag = aux;   // [nblk][36], nblk <= 16
  {
    int I = 0, rem = wave;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int tt = wave + 4 * t;
      if (tt < ntiles) {
        while (rem > I) {
          rem -= I + 1;
          ++I;
        }
        const int J = rem;
        const int j = 16 * J + (lane & 15), bj = j / 6, jj = j - 6 * bj;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + (lane >> 4) + 4 * r;
          const double v = acc[t][r];
          if (j < Dp && i == Dp) {
            sr[j] = v;
          } else if (i < Dp && j <= i) {
            const int bi = i / 6, ii = i - 6 * bi;
            if (bi == bj) s_diag[bi * 36 + 6 * ii + jj] = v;
            else sp[(bi * (bi + 1) / 2 + bj) * 36 + 6 * ii + jj] = -v;
          }
        }
        rem += 4;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < nblk * 36; e += LIN_THREADS) {
    const int bi = e / 36, k = e - 36 * bi, ii = k / 6, jj = k - 6 * ii;
    if (jj <= ii) {
      const int tk = blktask[bi];
      sp[(bi * (bi + 1) / 2 + bi) * 36 + k] = (tk >= 0 ? s_U[tk * 36 + ut6(jj, ii)] : 0.0) - s_diag[e];
    }
  }
  if (tid < Dp) {
    const int bi = tid / 6, a = tid - 6 * bi, tk = blktask[bi];
    sr[Dp + tid] = tk >= 0 ? s_U[tk * 36 + 21 + a] : 0.0;
    sr[2 * Dp + tid] = tk >= 0 ? s_U[tk * 36 + ut6(a, a)] : 0.0;
  }
  FSTAMP(22);
  FSTAMP(23);
#undef FSTAMP
}

// grid.x = n_small + (number of groups): the first n_small = max_imu + 1 workgroups evaluate the IMU / prior
// factors (small_body, ba_imu.hpp; they start first because a re-preintegration is the longest workgroup of
// the launch), the others one linearise group each.  Two workgroups per CU (LDS), hence at most 256 registers.
// FUSE: fused mode (see below); a separate instantiation so that the plain kernel carries none of its registers.
template <bool EXT, class REAL, bool FUSE>
__global__ __launch_bounds__(LIN_THREADS, 2) void linearize_kernel(const WinPtrs* __restrict__ wins,
                                                                   const OptD* __restrict__ optp, int init, int n_small) {
  constexpr bool fuse = FUSE;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const WinPtrs& W = wins[blockIdx.y];
  if ((int)blockIdx.x < n_small) {
    small_body(W, init, blockIdx.x, smem);
    return;
  }
  const int g = blockIdx.x - n_small;
  if (g >= W.n_group) return;
#define LSTAMP(k) do { if (W.prof && threadIdx.x == 0 && g == 0) W.prof[k] = (double)clock64(); } while (0)
  LSTAMP(40);
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;
  const OptD opt = *optp;
  // ---- fused mode (fuse != 0; the host picks it for DOGLEG / fixed-radius runs whose groups fit): this
  //      workgroup also reduces its own group — chunk g of the window — to the partial reduced-camera system right after
  //      linearising it, with the regulariser the NEXT solve will use if the trial is accepted (a rejected dogleg trial needs
  //      no new solve).  There is no Schur launch in this mode. ----
  static_assert(LIN_THREADS == SCHUR_THREADS, "the fused reduction runs in the linearise workgroup");
  auto reduce_own_group = [&](int buf, double lam) {
    const Group Gr = W.groups[g];
    const int trows = min(TILE_DIM, W.Dp);
    // LDS: the observation stage is dead (tables), so is the step vector of phase A (inverse landmark blocks, offsets)
    double* tables = smem;
    double* aux = smem + LinCfg<EXT, REAL>::STAGE_DOUBLES + GROUP_LM * 4 + GROUP_PAIRS * 3 + GROUP_LM * 16;
    double(*vinv)[6] = reinterpret_cast<double(*)[6]>(aux);
    double(*bvec)[3] = reinterpret_cast<double(*)[3]>(aux + SCHUR_CHUNK_LM_MAX * 6);
    int* boff = reinterpret_cast<int*>(aux + SCHUR_CHUNK_LM_MAX * 9);
    static_assert(SCHUR_CHUNK_LM_MAX * 9 + SCHUR_THREADS / 2 <= 1024, "aux area of the fused reduction");
    const SchurPairBeginGlobal pb{W.lm_pair_begin, Gr.lm_begin, Gr.lm_end};
    const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
    for (int tp = 0; tp < n_tp; ++tp) {
      schur_reduce_chunk(W, opt, g, tp, Gr.lm_begin, Gr.lm_end, buf, lam, trows, tables, vinv, bvec, boff, pb, g == 0 && blockIdx.y == 0);
      __syncthreads();
    }
  };
  if (!init && !ctrl->pending) {
    // the solve produced no valid step: nothing to evaluate.  Fused mode: the accepted linearisation is reduced again with the
    // regulariser the failed solve has raised
    if constexpr (FUSE) reduce_own_group(ctrl->acc, opt.dogleg ? ctrl->mu : 1.0 / ctrl->radius);
    return;
  }
  const int acc = ctrl->acc, trial = 1 - acc;
  const double lambda = ctrl->lambda;
  // regulariser of the solve that follows an accepted trial (decide_dl / decide, ba_device.hpp)
  const double lam_next = opt.dogleg ? ((init || ctrl->first || opt.gauss_newton) ? ctrl->mu : fmax(DL_MIN_MU, 2.0 * ctrl->mu / DL_MU_INCREASE))
                                     : 1.0 / ctrl->radius;
  const bool dl_explicit = opt.dogleg && ctrl->tr_kind == 1;
  const double dl_cA = ctrl->cA, dl_beta = ctrl->beta;

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
  const int ntask = G.task_end - G.task_begin;
  const int nplist = G.plist_end - G.plist_begin;   // <= 2 nobs
  const int ntlist = G.tlist_end - G.tlist_begin;   // <= 3 nobs

  // The group's index lists (static structure) are fetched NOW into registers — nothing but G is needed for the
  // addresses — and parked in LDS after phase A (in the s_pair area, dead by then), so that the reduction loops
  // of phase C chase indices through LDS instead of through dependent global loads.
  int* s_plb = reinterpret_cast<int*>(s_pair);                                // [GROUP_PAIRS + 1] pair list begins
  int* s_lob = s_plb + GROUP_PAIRS + 1;                                       // [GROUP_LM + 1] landmark obs begins
  int* s_task = s_lob + GROUP_LM + 1;                                         // [LIN_TASK_CACHE] Task records
  uint16_t* s_plist = reinterpret_cast<uint16_t*>(s_task + LIN_TASK_CACHE * 6);  // [2 GROUP_OBS]
  uint16_t* s_tlist = s_plist + 2 * GROUP_OBS;                                // [3 GROUP_OBS]
  uint8_t* s_prole = reinterpret_cast<uint8_t*>(s_tlist + 3 * GROUP_OBS);     // [GROUP_PAIRS]
  uint8_t* s_ppoff = s_prole + GROUP_PAIRS;                                   // [GROUP_PAIRS] reduced block offset / 6 (fast fused path)
  uint8_t* s_pplm = s_ppoff + GROUP_PAIRS;                                    // [GROUP_PAIRS] group-local landmark
  int pf_plb[2], pf_role[2], pf_lob = 0;
  uint16_t pf_pl[2], pf_tl[3];
  Task pf_task;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + j * LIN_THREADS;
    pf_plb[j] = (i < npair) ? W.pair_list_begin[G.pair_begin + i] : 0;
    pf_role[j] = (EXT && i < npair) ? W.pair_role[G.pair_begin + i] : 0;
    pf_pl[j] = (i < nplist) ? W.pair_list[G.plist_begin + i] : (uint16_t)0;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + j * LIN_THREADS;
    pf_tl[j] = (i < ntlist) ? W.task_list[G.tlist_begin + i] : (uint16_t)0;
  }
  if (tid <= nlm) pf_lob = W.lm_obs_begin[G.lm_begin + tid];
  const bool fast = FUSE && W.fuse_fast;
  int pf_poff[2] = {0, 0}, pf_plm[2] = {0, 0};
  double pf_sc[3] = {1.0, 1.0, 1.0};   // Jacobi scale of this work-item's landmark (fast fused path; estimated in this launch when init)
  if (fast && !init && tid >= 64 && tid - 64 < nlm) {   // (the landmark work of the reduction is done by wave 1)
    const double* sl = W.lm_scale + 3 * (size_t)(G.lm_begin + tid - 64);
    pf_sc[0] = sl[0], pf_sc[1] = sl[1], pf_sc[2] = sl[2];
  }
  if (fast) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + j * LIN_THREADS;
      if (i < npair) {
        pf_poff[j] = W.pair_off[G.pair_begin + i];
        pf_plm[j] = W.pair_lm[G.pair_begin + i] - G.lm_begin;
      }
    }
  }
  const bool tasks_cached = ntask <= LIN_TASK_CACHE;
  if (tasks_cached && tid < ntask) pf_task = W.tasks[G.task_begin + tid];

  LSTAMP(41);
  // ------------------------------------------------------------------ phase A: back-substitution
  double sc_gd = 0, sc_ddd = 0, sc_s2 = 0, sc_x2 = 0;
  if (!init) {
    // every operand that does not depend on the step is requested before the first barrier: the pair rows of the
    // first pass (W, 18 doubles) and the landmark lanes' b, V, x — one memory round trip instead of three
    const double* Wacc = W.W[acc];
    double Wp0[18];
    int poff0 = 0;
    if (tid < npair) {
      const double* Wp = Wacc + (size_t)(G.pair_begin + tid) * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) Wp0[i] = Wp[i];
      poff0 = W.pair_off[G.pair_begin + tid];
    }
    double bb[3] = {0, 0, 0}, v[6] = {1, 0, 0, 1, 0, 1}, xx[4] = {0, 0, 0, 0};
    int lp0 = 0, lp1 = 0;
    if (tid < nlm) {
      const int l = G.lm_begin + tid;
      const double* b = W.bl[acc] + 3 * (size_t)l;
      const double* Vl = W.V[acc] + 6 * (size_t)l;
      const double* x = W.lm[acc] + 4 * (size_t)l;
#pragma unroll
      for (int i = 0; i < 3; ++i) bb[i] = b[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = Vl[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) xx[i] = x[i];
      lp0 = W.lm_pair_begin[l] - G.pair_begin;
      lp1 = W.lm_pair_begin[l + 1] - G.pair_begin;
    }
    for (int i = tid; i < W.D; i += LIN_THREADS) s_step[i] = W.step[i];
    __syncthreads();
    if (tid < npair) {
      const double* d = s_step + poff0;
      double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        t0 += Wp0[3 * i] * d[i];
        t1 += Wp0[3 * i + 1] * d[i];
        t2 += Wp0[3 * i + 2] * d[i];
      }
      s_pair[3 * tid] = t0;
      s_pair[3 * tid + 1] = t1;
      s_pair[3 * tid + 2] = t2;
    }
    for (int p = tid + LIN_THREADS; p < npair; p += LIN_THREADS) {
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
      double t[3] = {bb[0], bb[1], bb[2]};
      for (int p = lp0; p < lp1; ++p) {
        const double* sp = s_pair + 3 * p;
        t[0] += sp[0];
        t[1] += sp[1];
        t[2] += sp[2];
      }
      double sc[3] = {1.0, 1.0, 1.0};
      if (opt.dogleg) {
        const double* sl = W.lm_scale + 3 * (size_t)l;
        sc[0] = sl[0], sc[1] = sl[1], sc[2] = sl[2];
      }
      const double d0 = damp_diag(v[0], sc[0], opt);
      const double d1 = damp_diag(v[3], sc[1], opt);
      const double d2 = damp_diag(v[5], sc[2], opt);
      v[0] += lambda * d0;
      v[3] += lambda * d1;
      v[5] += lambda * d2;
      double vi[6];
      inv3sym(v, vi);
      // (dogleg: this is the landmark part of the Gauss-Newton point; the scalars g.delta and delta^T D^2 delta below
      //  always refer to it, they decide whether the point lies inside the trust region)
      const double dl0 = -(vi[0] * t[0] + vi[1] * t[1] + vi[2] * t[2]);
      const double dl1 = -(vi[1] * t[0] + vi[3] * t[1] + vi[4] * t[2]);
      const double dl2 = -(vi[2] * t[0] + vi[4] * t[1] + vi[5] * t[2]);
      double st0 = dl0, st1 = dl1, st2 = dl2;
      if (dl_explicit) {   // explicit dogleg step  -cA xv + beta dGN,  xv_l = b_l / Dt2_l
        st0 = -dl_cA * (bb[0] / d0) + dl_beta * dl0;
        st1 = -dl_cA * (bb[1] / d1) + dl_beta * dl1;
        st2 = -dl_cA * (bb[2] / d2) + dl_beta * dl2;
      }
      const double x0 = xx[0], x1 = xx[1], x2 = xx[2], x3 = xx[3];
      double* xt = W.lm[trial] + 4 * (size_t)l;
      const double n0 = x0 + st0, n1 = x1 + st1, n2 = x2 + st2;
      xt[0] = n0; xt[1] = n1; xt[2] = n2; xt[3] = x3;
      s_lm[4 * tid] = n0; s_lm[4 * tid + 1] = n1; s_lm[4 * tid + 2] = n2; s_lm[4 * tid + 3] = x3;
      sc_gd = bb[0] * dl0 + bb[1] * dl1 + bb[2] * dl2;
      sc_ddd = d0 * dl0 * dl0 + d1 * dl1 * dl1 + d2 * dl2 * dl2;
      sc_s2 = st0 * st0 + st1 * st1 + st2 * st2;
      sc_x2 = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
    }
  } else if (tid < nlm) {
    const double* x = W.lm[trial] + 4 * (size_t)(G.lm_begin + tid);
    s_lm[4 * tid] = x[0]; s_lm[4 * tid + 1] = x[1]; s_lm[4 * tid + 2] = x[2]; s_lm[4 * tid + 3] = x[3];
  }
  __syncthreads();
  LSTAMP(42);
  // park the prefetched index lists (s_pair is dead: its last reader was before the barrier above)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + j * LIN_THREADS;
    if (i < npair) s_plb[i] = pf_plb[j] - G.plist_begin;
    if (EXT && i < npair) s_prole[i] = (uint8_t)pf_role[j];
    if (fast && i < npair) {
      s_ppoff[i] = (uint8_t)(pf_poff[j] / 6);
      s_pplm[i] = (uint8_t)pf_plm[j];
    }
    if (i < nplist) s_plist[i] = pf_pl[j];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + j * LIN_THREADS;
    if (i < ntlist) s_tlist[i] = pf_tl[j];
  }
  if (tid == 0) s_plb[npair] = nplist;
  if (tid <= nlm) s_lob[tid] = pf_lob - G.obs_begin;
  if (fast) {   // which of the group's tasks holds the J^T J block of pose block bi (s_step is free after phase A)
    int* blktask = reinterpret_cast<int*>(s_step + GROUP_LM * 9 + FUSE_MAX_TASKS * 36);
    if (tid >= 64 && tid < 64 + 32) blktask[tid - 64] = -1;
  }
  if (tasks_cached && tid < ntask) {
    int* t = s_task + 6 * tid;
    t[0] = pf_task.type; t[1] = pf_task.off_a; t[2] = pf_task.off_b;
    t[3] = pf_task.list_begin - G.tlist_begin; t[4] = pf_task.list_end - G.tlist_begin; t[5] = pf_task.out;
  }

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
    LSTAMP(43);
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
  LSTAMP(44);
  if (fast && tid < ntask) reinterpret_cast<int*>(s_step + GROUP_LM * 9 + FUSE_MAX_TASKS * 36)[pf_task.off_a / 6] = tid;   // (type 0 tasks only on this path)

  // ------------------------------------------------------------------ phase C: LDS reductions
  // (a) per landmark: V(6) b(3) Hq(6) cost(1).  Four lanes per (landmark, entry), each takes every fourth
  //     observation; every entry is the same expression  sum (x_A x_B + x_C x_D) m  with per-lane operand offsets
  //     (no divergence between the entry classes, which cost four serialised loops per wave before).
  {
    const int e = (tid >> 2) & 15, part = tid & 3;   // loop invariant: LIN_THREADS is a multiple of 64
    const bool is_cost = e == 15, use_irho = e >= 9 && e < 15, is_b = e >= 6 && e < 9;
    const int ee = e < 6 ? e : (use_irho ? e - 9 : 0);
    const int vi = (ee < 3) ? 0 : (ee < 5 ? 1 : 2);
    const int vj = (ee < 3) ? ee : (ee < 5 ? ee - 2 : 2);
    const int oA = is_cost ? ST_COST : ST_JL + (is_b ? e - 6 : vi);
    const int oB = is_b ? ST_R : ST_JL + vj;
    const int oC = ST_JL + 3 + (is_b ? e - 6 : vi);
    const int oD = is_b ? ST_R + 1 : ST_JL + 3 + vj;
    double* obase = e < 6 ? W.V[trial] : (is_b ? W.bl[trial] : W.Hq[trial]);
    const int ostride = is_b ? 3 : 6, ooff = e < 6 ? e : (is_b ? e - 6 : e - 9);
    for (int wi = tid; wi < nlm * 64; wi += LIN_THREADS) {
      const int ll = wi >> 6;
      const int o0 = s_lob[ll], o1 = s_lob[ll + 1];
      REAL a = 0;
      auto term = [&](int o) -> REAL {
        const REAL* st = s_stage + (size_t)o * STRIDE;
        const REAL t1 = st[oA] * (is_cost ? REAL(1) : st[oB]);
        const REAL t2 = is_cost ? REAL(0) : st[oC] * st[oD];
        const REAL m = use_irho ? st[ST_IRHO] : REAL(1);
        return (t1 + t2) * m;
      };
      int o = o0 + part;
      for (; o + 4 < o1; o += 8) {   // two observations per trip: both sets of LDS reads in flight together
        const REAL ta = term(o), tb = term(o + 4);
        a += ta;
        a += tb;
      }
      if (o < o1) a += term(o);
      a = quad_sum(a);
      if (part == 0) {
        if (!is_cost) obase[ostride * (size_t)(G.lm_begin + ll) + ooff] = a;
        s_lmres[16 * ll + e] = a;
        // first linearisation of an optimize() call: Jacobi scale of the landmark columns (Ceres EstimateScale,
        // 1 / (1 + sqrt(diag J^T J)); e = 0, 3, 5 are the diagonal entries of V)
        if (init && opt.dogleg && (e == 0 || e == 3 || e == 5))
          W.lm_scale[3 * (size_t)(G.lm_begin + ll) + (e == 0 ? 0 : (e == 3 ? 1 : 2))] =
              opt.jacobi_scaling ? 1.0 / (1.0 + sqrt((double)a)) : 1.0;
      }
    }
  }
  LSTAMP(45);
  // (b) per (landmark, block) pair: W = sum J_block^T J_l, one work-item per row (fast fused path: the first FUSE_WIT rows of
  //     every work-item — all of them, the host checks — stay in registers for the reduction behind phase C)
  FuseItems fit;
#pragma unroll
  for (int q = 0; q < FUSE_WIT; ++q) fit.row[q] = -1, fit.lm[q] = 0, fit.w[q][0] = fit.w[q][1] = fit.w[q][2] = 0.0;
  auto w_row = [&](int wi, REAL& w0, REAL& w1, REAL& w2) {
    const int pp = wi / 6, a = wi - 6 * pp;
    const int jofs = (EXT && s_prole[pp]) ? ST_JE : ST_JP;
    w0 = 0, w1 = 0, w2 = 0;
    for (int k = s_plb[pp]; k < s_plb[pp + 1]; ++k) {
      const REAL* st = s_stage + (size_t)s_plist[k] * STRIDE;
      const REAL j0 = st[jofs + a], j1 = st[jofs + 6 + a];
      w0 += j0 * st[ST_JL] + j1 * st[ST_JL + 3];
      w1 += j0 * st[ST_JL + 1] + j1 * st[ST_JL + 4];
      w2 += j0 * st[ST_JL + 2] + j1 * st[ST_JL + 5];
    }
    double* Wt = W.W[trial] + (size_t)(G.pair_begin + pp) * 18 + 3 * a;
    Wt[0] = w0;
    Wt[1] = w1;
    Wt[2] = w2;
  };
  if (fast) {
#pragma unroll
    for (int q = 0; q < FUSE_WIT; ++q) {
      const int wi = tid + q * LIN_THREADS;
      if (wi < npair * 6) {
        REAL w0, w1, w2;
        w_row(wi, w0, w1, w2);
        const int pp = wi / 6;
        fit.w[q][0] = w0, fit.w[q][1] = w1, fit.w[q][2] = w2;
        fit.row[q] = 6 * s_ppoff[pp] + (wi - 6 * pp);
        fit.lm[q] = s_pplm[pp];
      }
    }
  } else {
    for (int wi = tid; wi < npair * 6; wi += LIN_THREADS) {
      REAL w0, w1, w2;
      w_row(wi, w0, w1, w2);
    }
  }
  LSTAMP(46);
  // (c) per-block J^T J / J^T r partials and pose-extrinsics cross blocks.  These are the long reductions of the
  //     group (every observation of one pose block): four lanes share one (task, row) item, each takes every
  //     fourth observation of the task's list, the partial sums are combined with two xor-shuffles
  //     (8 lanes per item measured slower: 109 vs 101 us on the 64-window launch).
  for (int wi = tid; wi < ntask * 24; wi += LIN_THREADS) {
    const int tt = wi / 24, a = (wi >> 2) % 6, part = wi & 3;
    Task T;
    if (tasks_cached) {
      const int* t = s_task + 6 * tt;
      T.type = t[0]; T.off_a = t[1]; T.off_b = t[2]; T.list_begin = t[3]; T.list_end = t[4]; T.out = t[5];
    } else {
      T = W.tasks[G.task_begin + tt];
      T.list_begin -= G.tlist_begin;
      T.list_end -= G.tlist_begin;
    }
    double* out = W.gpart[trial] + T.out;
    REAL acc6[6] = {0, 0, 0, 0, 0, 0};
    REAL ga = 0;
    // jr = row operand block (J_a), jc = column operand block: pose/extrinsics Hessian  J_x^T J_x (+ J_x^T r),
    // cross block J_pose^T J_ext
    const int jr = (T.type == 1) ? ST_JE : ST_JP;
    const int jc = (T.type == 0) ? ST_JP : ST_JE;
    const bool with_g = T.type < 2;
    auto term = [&](int o, REAL t6[6], REAL& tg) {
      const REAL* st = s_stage + (size_t)o * STRIDE;
      const REAL j0 = st[jr + a], j1 = st[jr + 6 + a];
#pragma unroll
      for (int b = 0; b < 6; ++b) t6[b] = j0 * st[jc + b] + j1 * st[jc + 6 + b];
      tg = j0 * st[ST_R] + j1 * st[ST_R + 1];
    };
    if (EXT || T.type == 0) {
      int k = T.list_begin + part;
      for (; k + 4 < T.list_end; k += 8) {   // two observations per trip (LDS reads of both in flight together)
        const int oa = s_tlist[k], ob = s_tlist[k + 4];
        REAL ta[6], tb[6], ga_a, ga_b;
        term(oa, ta, ga_a);
        term(ob, tb, ga_b);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          acc6[b] += ta[b];
          acc6[b] += tb[b];
        }
        ga += ga_a;
        ga += ga_b;
      }
      if (k < T.list_end) {
        REAL ta[6], ga_a;
        term(s_tlist[k], ta, ga_a);
#pragma unroll
        for (int b = 0; b < 6; ++b) acc6[b] += ta[b];
        ga += ga_a;
      }
    }
    if (!with_g) ga = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) acc6[b] = quad_sum(acc6[b]);
    ga = quad_sum(ga);
    if (part == 0) {
      if (fast) {   // (type 0 only on this path) the group's own J^T J / J^T r blocks for the reduction below
        double* u = s_step + GROUP_LM * 9 + tt * 36;
        for (int b = a; b < 6; ++b) u[ut6(a, b)] = acc6[b];
        u[21 + a] = ga;
      }
      if (T.type < 2) {
        for (int b = a; b < 6; ++b) out[ut6(a, b)] = acc6[b];
        out[21 + a] = ga;
      } else if (EXT) {
        for (int b = 0; b < 6; ++b) out[6 * a + b] = acc6[b];
      }
    }
  }
  LSTAMP(47);
  __syncthreads();
  LSTAMP(48);
  LSTAMP(49);
  auto group_scalars = [&]() {   // (d) group scalars by wave 0
  if (tid < 64) {
      double cost = 0, gm = 0;
      if (tid < nlm) {
        cost = s_lmres[16 * tid + 15];
        gm = fmax(fabs(s_lmres[16 * tid + 6]), fmax(fabs(s_lmres[16 * tid + 7]), fabs(s_lmres[16 * tid + 8])));
      }
      cost = wave_sum_full(cost);
      gm = wave_max_full(gm);
      sc_gd = wave_sum_full(sc_gd);
      sc_ddd = wave_sum_full(sc_ddd);
      sc_s2 = wave_sum_full(sc_s2);
      sc_x2 = wave_sum_full(sc_x2);
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
  };
  if constexpr (FUSE) {
    // (the barrier above has V, b, W, the per-group partials in global memory and the last reads of the stage behind it)
    // the scalars first, by wave 0; the landmark work of the fast reduction belongs to wave 1, so the two overlap
    group_scalars();
    if (fast)
      fused_reduce_fast<LinCfg<EXT, REAL>::STAGE_DOUBLES, FUSE_WIT>(W, opt, g, trial, lam_next, nlm, init != 0, pf_sc, fit, smem, s_lmres, s_step);
    else
      reduce_own_group(trial, lam_next);
  }
  LSTAMP(50);
  if constexpr (!FUSE) group_scalars();
  LSTAMP(51);
}

}  // namespace ba
