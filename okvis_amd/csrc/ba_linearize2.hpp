// Kernel 1, piece path — back-substitute + (+)update + re-linearise the reprojection factors of windows WITHOUT free
// extrinsics (the stock configuration; windows with online extrinsics calibration keep ba_linearize.hpp).
//
// Same interface, same outputs as linearize_kernel (V, b, un-robustified H_l, W per (landmark, block) pair, per-group
// J^T J / J^T r partials, group scalars, fused group reduction), different reduction scheme.  ba_linearize.hpp stages the
// 2x15 Jacobian of every observation in LDS (47 KB per workgroup) and lets gather loops chase index lists through it: two
// workgroups per CU, waves parked on LDS round trips two thirds of their time.  Here nothing per-observation is staged:
//
//   * With T_SC fixed, the pose Jacobian of an observation is its landmark Jacobian times a 3x6 matrix that depends only
//     on (landmark, pose):  J_pose = J_l M,  M = [ -w I | rows e_i x d ],  d = hp_W - r_WS w
//     (implementation/ReprojectionError.hpp:156-167 against :188-206).  Hence for all observations of one (landmark, pose)
//     pair  sum J_p^T J_l = M^T Vp,  sum J_p^T J_p = M^T Vp M,  sum J_p^T r = M^T bp  with  Vp = sum J_l^T J_l (6 unique
//     entries), bp = sum J_l^T r: the pair blocks follow from NINE sums instead of 45.
//   * Observations are sorted by (landmark, pose, camera): the observations of a pair sit in adjacent lanes.  A PIECE is
//     one or two adjacent lanes of one pair inside one DPP row of 16 (greedy from the start of the run; the host
//     enumerates pieces with the same rule).  The 16 per-observation products (V 6, b 3, H_l 6, cost) are merged across
//     the two lanes of a piece with one DPP row shift and the piece heads store them as 16-value records in LDS
//     (<= LIN2_PIECES records).  Landmark sums, pair sums (a pair split by a row boundary has several pieces) and, after
//     the pair lanes have formed W / U / g in registers and stored the 27-entry block records in block order, the
//     per-block sums over the group's pairs are short contiguous fixed-order LDS sums: deterministic, no atomics.
//   * The window's poses and intrinsics are staged in LDS together with the other operands: phase B has no dependent
//     global load left.  (Several groups per workgroup with the next group's operands requested one group ahead were
//     measured slower, 104-118 against 83 us per 64-window launch: as a loop the body costs some hundred spilled registers,
//     the compiler hoists the constants and addresses of every phase in front of it.)
//
// LDS per workgroup: 27 KB records + 2 KB landmarks + 8 KB landmark results + 4 KB poses / intrinsics + 3 KB index lists
// + the pose part of the step: ~48 KB instead of 78 KB, 115 registers instead of 255 (three workgroups per CU), and no
// 2x12 pose Jacobian in registers.
#pragma once
#include "ba_linearize.hpp"

namespace ba {

constexpr int LIN2_REC = 27;     // entries of a pair's block record: 21 J^T J (upper triangle) + 6 J^T r
constexpr int LIN2_POSES = 64;   // poses of a window staged in LDS (more: phase B reads them from global memory)
constexpr int LIN2_CAMS = 8;     // cameras staged in LDS
constexpr int LIN2_IDX_INTS = 2 * (((GROUP_LM + 1) + 2 * LIN2_PIECES + LIN_TASK_CACHE * 6 + LIN2_PIECES / 2 + LIN2_PIECES / 4 + LIN2_CAMS + 1) / 2);

// UB: entries of the block records that go through LDS per round: all 27 (one round), or 14 (two rounds: the record area then
// is the 16 KB of the piece records, and four workgroups instead of three fit a CU)
template <class REAL, bool FUSE, int UB = LIN2_REC>
struct Lin2Cfg {
  // piece records [pieces][16], then block records [pairs][27]; the fused launch puts the tiles / tables of the group
  // reduction here afterwards (at least the observation stage of ba_linearize.hpp, which the host checked them against)
  static constexpr int PAIR_REC_DOUBLES = (LIN2_PIECES * (UB > 16 ? UB : 16) * (int)sizeof(REAL) + 7) / 8;
  static constexpr int STAGE = LinCfg<false, REAL>::STAGE_DOUBLES;
  static constexpr int REC_DOUBLES = FUSE ? (STAGE > PAIR_REC_DOUBLES ? STAGE : PAIR_REC_DOUBLES) : PAIR_REC_DOUBLES;
  static constexpr int FIXED_DOUBLES = REC_DOUBLES + GROUP_LM * 4 + GROUP_LM * 16 + 4 * 64 + LIN2_POSES * 7 + LIN2_CAMS * 12 + LIN2_IDX_INTS / 2;
  static constexpr int MIN_STEP_DOUBLES = FUSE ? 1024 : 0;   // aux area behind the step (fused: inverse landmark blocks, J^T J blocks, offsets)
};
static_assert(LIN2_PIECES * 3 * 2 <= LIN2_PIECES * 16, "phase A's pair sums alias the record area");
static_assert(LIN2_PIECES <= LIN_THREADS, "one pair per work-item");

// value of lane + 1 of the same DPP row (0 for the last lane of a row): row_shl:1
template <class T>
__device__ __forceinline__ T row_next(T v) { return quad_xchg<0x101>(v); }
__device__ __forceinline__ int row_next_i(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x101, 0xF, 0xF, false); }
__device__ __forceinline__ int row_prev_i(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xF, 0xF, false); }

// what a workgroup fetches for one group before it can start on it
struct Lin2Ops {     // records and index lists (20 registers)
  ObsRec rec;
  int lpb, pp, poff, plm, pblock, slot;
  Task task;
};
struct Lin2Heavy {   // operands of the back-substitution (54 registers)
  double Wp0[18];                      // pair lane: W row block of the accepted linearisation
  double bb[3], v[6], xx[4], sl[3];    // landmark lane: b, V, x of the accepted buffer, Jacobi scale
  int lp0, lp1;
};

// SMALL: the first n_small workgroups evaluate the IMU / prior factors (as in ba_linearize.hpp: one launch, one window's
// latency); without it the small factors have their own launch (small_kernel) and this kernel's register budget is its
// own.  OCC: workgroups per CU the kernel is compiled for (512 / OCC registers per work-item).
template <class REAL, bool FUSE, bool SMALL, int OCC = 2, int UB = LIN2_REC>
__global__ __launch_bounds__(LIN_THREADS, OCC) void linearize2_kernel(const WinPtrs* __restrict__ wins, const OptD* __restrict__ optp, int init,
                                                                      int n_small, int step_doubles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const WinPtrs& W = wins[blockIdx.y];
  int wg = blockIdx.x;
  if constexpr (SMALL) {
    if ((int)blockIdx.x < n_small) {
      small_body(W, init, blockIdx.x, smem);
      return;
    }
    wg -= n_small;
  }
  const int g = wg;
  if (g >= W.n_group) return;
#define LSTAMP(k) do { if (W.prof && threadIdx.x == 0 && g == 0) W.prof[k] = (double)clock64(); } while (0)
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;
  const OptD opt = *optp;
  constexpr int RECD = Lin2Cfg<REAL, FUSE, UB>::REC_DOUBLES;
  REAL* s_rec = reinterpret_cast<REAL*>(smem);   // [pieces][16] piece records; then [pairs][27] block records in slot order
  double* s_pair = smem;                         // phase A: [pairs][3]
  double* s_lm = smem + RECD;                    // [GROUP_LM][4] trial landmarks
  double* s_lmres = s_lm + GROUP_LM * 4;         // [GROUP_LM][16] landmark sums (V 6, b 3, H_l 6, cost)
  double* s_sc = s_lmres + GROUP_LM * 16;        // [4][64] per-landmark step scalars of phase A
  double* s_pose = s_sc + 4 * 64;                // [LIN2_POSES][7] trial poses of the window
  double* s_cam = s_pose + LIN2_POSES * 7;       // [LIN2_CAMS][12]
  int* s_lpb = reinterpret_cast<int*>(s_cam + LIN2_CAMS * 12);   // [GROUP_LM + 1] first piece of each landmark
  int* s_pp = s_lpb + GROUP_LM + 1;              // [LIN2_PIECES] pair -> first piece | count << 16
  int* s_poff = s_pp + LIN2_PIECES;              // [LIN2_PIECES] reduced offset of the pair's block
  int* s_task = s_poff + LIN2_PIECES;            // [LIN_TASK_CACHE][6]
  int* s_cmodel = s_task + LIN_TASK_CACHE * 6;   // [LIN2_CAMS]
  uint16_t* s_slot = reinterpret_cast<uint16_t*>(s_cmodel + LIN2_CAMS);   // [LIN2_PIECES] pair -> slot of its block record
  uint8_t* s_plm = reinterpret_cast<uint8_t*>(s_slot + LIN2_PIECES);      // [LIN2_PIECES] group-local landmark of a pair
  double* s_step = reinterpret_cast<double*>(s_lpb + LIN2_IDX_INTS);      // [step_doubles]: pose part of the step | fused: aux area
  double* s_aux = s_step + (step_doubles - Lin2Cfg<REAL, FUSE, UB>::MIN_STEP_DOUBLES);   // (fused: inverse landmark blocks, J^T J blocks, offsets)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- fused mode: see ba_linearize.hpp ----
  auto reduce_own_group = [&](int g, int buf, double lam) {
    const Group Gr = W.groups[g];
    const int trows = min(TILE_DIM, W.Dp);
    double* tables = smem;
    double* aux = s_aux;
    double(*vinv)[6] = reinterpret_cast<double(*)[6]>(aux);
    double(*bvec)[3] = reinterpret_cast<double(*)[3]>(aux + SCHUR_CHUNK_LM_MAX * 6);
    int* boff = reinterpret_cast<int*>(aux + SCHUR_CHUNK_LM_MAX * 9);
    const SchurPairBeginGlobal pb{W.lm_pair_begin, Gr.lm_begin, Gr.lm_end};
    const int n_tp = W.n_tile * (W.n_tile + 1) / 2;
    for (int tp = 0; tp < n_tp; ++tp) {
      schur_reduce_chunk(W, opt, g, tp, Gr.lm_begin, Gr.lm_end, buf, lam, trows, tables, vinv, bvec, boff, pb, g == 0 && blockIdx.y == 0);
      __syncthreads();
    }
  };
  if (!init && !ctrl->pending) {
    if constexpr (FUSE)
      reduce_own_group(g, ctrl->acc, opt.dogleg ? ctrl->mu : 1.0 / ctrl->radius);
    return;
  }
  const int acc = ctrl->acc, trial = 1 - acc;
  const double lambda = ctrl->lambda;
  const double lam_next = opt.dogleg ? ((init || ctrl->first || opt.gauss_newton) ? ctrl->mu : fmax(DL_MIN_MU, 2.0 * ctrl->mu / DL_MU_INCREASE))
                                     : 1.0 / ctrl->radius;
  const bool dl_explicit = opt.dogleg && ctrl->tr_kind == 1;
  const double dl_cA = ctrl->cA, dl_beta = ctrl->beta;
  const bool fast = FUSE && W.fuse_fast;
  const bool poses_staged = W.n_pose <= LIN2_POSES, cams_staged = W.n_cam <= LIN2_CAMS;

  // ---- once per workgroup: the window's trial poses and intrinsics into LDS (visible behind the first barrier) ----
  {
    const double* ps = W.pose[trial];
    const int np7 = 7 * min(W.n_pose, LIN2_POSES);
    for (int i = tid; i < np7; i += LIN_THREADS) s_pose[i] = ps[i];
    const int nc = min(W.n_cam, LIN2_CAMS);
    if (tid < 12 * nc) s_cam[tid] = W.cam_intr[tid];
    if (tid < nc) s_cmodel[tid] = W.cam_model[tid];
  }

  // everything a group needs that only depends on its Group record: requested in one go
  auto issue_loads = [&](const Group& G, Lin2Ops& o) {
    const int nlm = G.lm_end - G.lm_begin, nobs = G.obs_end - G.obs_begin, npair = G.pair_end - G.pair_begin, ntask = G.task_end - G.task_begin;
    o.rec.lm_cam = 0; o.rec.pose = 0; o.rec.ext = 0; o.rec.u = 0; o.rec.v = 0; o.rec.sw = 0;
    if (tid < nobs) o.rec = W.obs[G.obs_begin + tid];
    o.lpb = o.pp = o.poff = o.plm = o.pblock = o.slot = 0;
    if (tid <= nlm) o.lpb = W.lm_piece_begin[G.lm_begin + tid] - G.piece_begin;
    if (tid < npair) {
      o.pp = W.pair_piece[G.pair_begin + tid];
      o.poff = W.pair_off[G.pair_begin + tid];
      o.plm = W.pair_lm[G.pair_begin + tid] - G.lm_begin;
      o.pblock = W.pair_block[G.pair_begin + tid];
      o.slot = W.task_list[G.tlist_begin + tid];
    }
    if (ntask <= LIN_TASK_CACHE && tid < ntask) o.task = W.tasks[G.task_begin + tid];
  };
  // ... and the operands of the back-substitution (54 registers: requested at the top of the group's own turn)
  auto issue_heavy = [&](const Group& G, Lin2Heavy& o) {
    const int nlm = G.lm_end - G.lm_begin, npair = G.pair_end - G.pair_begin;
    o.sl[0] = o.sl[1] = o.sl[2] = 1.0;
    o.lp0 = o.lp1 = 0;
    if (!init) {
      if (tid < npair) {
        const double* Wp = W.W[acc] + (size_t)(G.pair_begin + tid) * 18;
#pragma unroll
        for (int i = 0; i < 18; ++i) o.Wp0[i] = Wp[i];
      }
      if (tid < nlm) {
        const int l = G.lm_begin + tid;
        const double* b = W.bl[acc] + 3 * (size_t)l;
        const double* Vl = W.V[acc] + 6 * (size_t)l;
        const double* x = W.lm[acc] + 4 * (size_t)l;
#pragma unroll
        for (int i = 0; i < 3; ++i) o.bb[i] = b[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) o.v[i] = Vl[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) o.xx[i] = x[i];
        o.lp0 = W.lm_pair_begin[l] - G.pair_begin;
        o.lp1 = W.lm_pair_begin[l + 1] - G.pair_begin;
        if (opt.dogleg) {
          const double* sl = W.lm_scale + 3 * (size_t)l;
          o.sl[0] = sl[0], o.sl[1] = sl[1], o.sl[2] = sl[2];
        }
      }
    } else if (tid < nlm) {
      const double* x = W.lm[trial] + 4 * (size_t)(G.lm_begin + tid);
#pragma unroll
      for (int i = 0; i < 4; ++i) o.xx[i] = x[i];
    }
  };

  const Group G = W.groups[g];
  Lin2Ops ops;
  issue_loads(G, ops);
  if (!init)
    for (int i = tid; i < W.Dp; i += LIN_THREADS) s_step[i] = W.step[i];   // (pairs only refer to pose blocks)

  {
    LSTAMP(40);
    Lin2Heavy hv;
    issue_heavy(G, hv);
    const int nlm = G.lm_end - G.lm_begin;
    const int nobs = G.obs_end - G.obs_begin;
    const int npair = G.pair_end - G.pair_begin;
    const int ntask = G.task_end - G.task_begin;
    const bool has_obs = tid < nobs, has_pair = tid < npair;
    const bool tasks_cached = ntask <= LIN_TASK_CACHE;
    const ObsRec rec = ops.rec;
    const int pf_poff = ops.poff, pf_plm = ops.plm, pf_pblock = ops.pblock;
    double pf_sc[3] = {1.0, 1.0, 1.0};
    if (fast && !init && tid >= 64 && tid - 64 < nlm) {   // (the landmark work of the group reduction is done by wave 1)
      const double* sl = W.lm_scale + 3 * (size_t)(G.lm_begin + tid - 64);
      pf_sc[0] = sl[0], pf_sc[1] = sl[1], pf_sc[2] = sl[2];
    }
    LSTAMP(41);
    // ------------------------------------------------------------------ phase A: back-substitution
    if (!init) {
      __syncthreads();   // the step is in LDS
      if (has_pair) {
        const double* d = s_step + pf_poff;
        double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          t0 += hv.Wp0[3 * i] * d[i];
          t1 += hv.Wp0[3 * i + 1] * d[i];
          t2 += hv.Wp0[3 * i + 2] * d[i];
        }
        s_pair[3 * tid] = t0;
        s_pair[3 * tid + 1] = t1;
        s_pair[3 * tid + 2] = t2;
      }
      __syncthreads();
      double sc_gd = 0, sc_ddd = 0, sc_s2 = 0, sc_x2 = 0;
      if (tid < nlm) {
        const int l = G.lm_begin + tid;
        double t[3] = {hv.bb[0], hv.bb[1], hv.bb[2]};
        for (int p = hv.lp0; p < hv.lp1; ++p) {
          const double* sp = s_pair + 3 * p;
          t[0] += sp[0];
          t[1] += sp[1];
          t[2] += sp[2];
        }
        double v[6] = {hv.v[0], hv.v[1], hv.v[2], hv.v[3], hv.v[4], hv.v[5]};
        const double d0 = damp_diag(v[0], hv.sl[0], opt);
        const double d1 = damp_diag(v[3], hv.sl[1], opt);
        const double d2 = damp_diag(v[5], hv.sl[2], opt);
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
          st0 = -dl_cA * (hv.bb[0] / d0) + dl_beta * dl0;
          st1 = -dl_cA * (hv.bb[1] / d1) + dl_beta * dl1;
          st2 = -dl_cA * (hv.bb[2] / d2) + dl_beta * dl2;
        }
        const double x0 = hv.xx[0], x1 = hv.xx[1], x2 = hv.xx[2], x3 = hv.xx[3];
        double* xt = W.lm[trial] + 4 * (size_t)l;
        const double n0 = x0 + st0, n1 = x1 + st1, n2 = x2 + st2;
        xt[0] = n0; xt[1] = n1; xt[2] = n2; xt[3] = x3;
        s_lm[4 * tid] = n0; s_lm[4 * tid + 1] = n1; s_lm[4 * tid + 2] = n2; s_lm[4 * tid + 3] = x3;
        sc_gd = hv.bb[0] * dl0 + hv.bb[1] * dl1 + hv.bb[2] * dl2;
        sc_ddd = d0 * dl0 * dl0 + d1 * dl1 * dl1 + d2 * dl2 * dl2;
        sc_s2 = st0 * st0 + st1 * st1 + st2 * st2;
        sc_x2 = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      }
      if (tid < 64) {
        s_sc[tid] = sc_gd;
        s_sc[64 + tid] = sc_ddd;
        s_sc[128 + tid] = sc_s2;
        s_sc[192 + tid] = sc_x2;
      }
    } else {
      if (tid < nlm) {
        s_lm[4 * tid] = hv.xx[0]; s_lm[4 * tid + 1] = hv.xx[1]; s_lm[4 * tid + 2] = hv.xx[2]; s_lm[4 * tid + 3] = hv.xx[3];
      }
      if (tid < 64) s_sc[tid] = s_sc[64 + tid] = s_sc[128 + tid] = s_sc[192 + tid] = 0.0;
    }
    // park the index lists of this group
    if (tid <= nlm) s_lpb[tid] = ops.lpb;
    if (has_pair) {
      s_pp[tid] = ops.pp;
      s_poff[tid] = pf_poff;
      s_plm[tid] = (uint8_t)pf_plm;
      s_slot[tid] = (uint16_t)ops.slot;
    }
    if (tasks_cached && tid < ntask) {
      int* t = s_task + 6 * tid;
      t[0] = ops.task.type; t[1] = ops.task.off_a; t[2] = ops.task.off_b;
      t[3] = ops.task.list_begin - G.tlist_begin; t[4] = ops.task.list_end - G.tlist_begin; t[5] = ops.task.out;
    }
    if (fast) {   // which of the group's tasks holds the J^T J block of pose block bi (behind the pose part of the step)
      int* blktask = reinterpret_cast<int*>(s_aux + GROUP_LM * 9 + FUSE_MAX_TASKS * 36);
      if (tid >= 64 && tid < 64 + 32) blktask[tid - 64] = -1;
    }
    __syncthreads();
    LSTAMP(42);
    if (fast && tid < ntask) reinterpret_cast<int*>(s_aux + GROUP_LM * 9 + FUSE_MAX_TASKS * 36)[ops.task.off_a / 6] = tid;

    // ------------------------------------------------------------------ phase B: one observation per lane
    // a[0..5] V, a[6..8] b, a[9..14] un-robustified H_l, a[15] cost of this observation
    REAL a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = REAL(0);
    const int key_l = has_obs ? (int)(rec.lm_cam & 0xFFFFFFu) : -1;
    const int key_p = has_obs ? (int)rec.pose : -1;
    if (has_obs) {
      const int o = G.obs_begin + tid;
      const int cam = (int)(rec.lm_cam >> 24);
      double P[7], E[7], intr[12];
      int cam_model;
      if (poses_staged) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          P[i] = s_pose[7 * (int)rec.pose + i];
          E[i] = s_pose[7 * (int)rec.ext + i];
        }
      } else {
        const double* pose = W.pose[trial] + 7 * (size_t)rec.pose;
        const double* ext = W.pose[trial] + 7 * (size_t)rec.ext;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          P[i] = pose[i];
          E[i] = ext[i];
        }
      }
      if (cams_staged) {
#pragma unroll
        for (int i = 0; i < 12; ++i) intr[i] = s_cam[12 * cam + i];
        cam_model = s_cmodel[cam];
      } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) intr[i] = W.cam_intr[12 * cam + i];
        cam_model = W.cam_model[cam];
      }
      const double* lm = s_lm + 4 * (key_l - G.lm_begin);
      const double L4[4] = {lm[0], lm[1], lm[2], lm[3]};
      REAL r0, r1, jl[6];
      if constexpr (std::is_same<REAL, double>::value) {
        ReprojLin Jd;
        reproj_linearize(P, E, L4, intr, cam_model, rec.u, rec.v, rec.sw, false, &Jd);
        r0 = Jd.r[0], r1 = Jd.r[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) jl[i] = Jd.Jl[i];
      } else {
        ReprojLinT<REAL> J;
        reproj_linearize_mixed<REAL>(P, E, L4, intr, cam_model, rec.u, rec.v, rec.sw, false, &J);
        r0 = J.r[0], r1 = J.r[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) jl[i] = J.Jl[i];
      }
      if (W.obs_r[trial]) {
        W.obs_r[trial][2 * (size_t)o] = r0;
        W.obs_r[trial][2 * (size_t)o + 1] = r1;
      }
      // Cauchy corrector (Ceres Corrector with rho'' <= 0: scale r and J by sqrt(rho'))
      const REAL s = r0 * r0 + r1 * r1;
      REAL sr = REAL(1), irho = REAL(1), cost = REAL(0.5) * s;
      if (W.cauchy_b > 0) {
        const REAL bb = REAL(W.cauchy_b * W.cauchy_b);
        const REAL sum = REAL(1) + s / bb;
        const REAL rho1 = REAL(1) / sum;
        cost = REAL(0.5) * bb * log(sum);
        sr = sqrt(rho1);
        irho = sum;
      }
      r0 *= sr, r1 *= sr;
#pragma unroll
      for (int i = 0; i < 6; ++i) jl[i] *= sr;
      a[0] = jl[0] * jl[0] + jl[3] * jl[3];
      a[1] = jl[0] * jl[1] + jl[3] * jl[4];
      a[2] = jl[0] * jl[2] + jl[3] * jl[5];
      a[3] = jl[1] * jl[1] + jl[4] * jl[4];
      a[4] = jl[1] * jl[2] + jl[4] * jl[5];
      a[5] = jl[2] * jl[2] + jl[5] * jl[5];
      a[6] = jl[0] * r0 + jl[3] * r1;
      a[7] = jl[1] * r0 + jl[4] * r1;
      a[8] = jl[2] * r0 + jl[5] * r1;
#pragma unroll
      for (int e = 0; e < 6; ++e) a[9 + e] = a[e] * irho;
      a[15] = cost;
    }
    LSTAMP(43);
    // ---- pieces: runs of one (landmark, pose) inside a DPP row, cut into pairs of lanes from the start of the run ----
    // (the lane exchanges first, with every lane active: a DPP read inside a short-circuited condition would see disabled lanes)
    const int prev_l = row_prev_i(key_l, -2), prev_p = row_prev_i(key_p, -2);
    const int next_l = row_next_i(key_l, -2), next_p = row_next_i(key_p, -2);
    const bool brk = ((lane & 15) == 0) | (key_l != prev_l) | (key_p != prev_p);
    const bool same_next = has_obs & ((lane & 15) != 15) & (key_l == next_l) & (key_p == next_p);
    const unsigned long long brk_mask = __ballot(brk);
    const unsigned long long below = brk_mask & ((2ull << lane) - 1ull);   // (lane 0 of every row is a break: never empty)
    const int run_start = 63 - __builtin_clzll(below);
    const bool head = has_obs && (((lane - run_start) & 1) == 0);
    const bool merge = head && same_next;
    const unsigned long long head_mask = __ballot(head);
    const int pw = wave == 0 ? 0 : (wave == 1 ? G.pw1 : (wave == 2 ? G.pw2 : G.pw3));
    const int piece = pw + __builtin_amdgcn_mbcnt_hi((unsigned)(head_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)head_mask, 0));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const REAL t = row_next(a[e]);
      a[e] += merge ? t : REAL(0);
    }
    if (head) {
      REAL* rp = s_rec + 16 * piece;
#pragma unroll
      for (int e = 0; e < 16; ++e) rp[e] = a[e];
    }
    __syncthreads();
    LSTAMP(44);

    // ------------------------------------------------------------------ phase C
    // (a) per landmark: 16 lanes, one per entry, sum the landmark's piece records in piece order
    {
      const int e = tid & 15;
      const bool is_cost = e == 15, is_b = e >= 6 && e < 9;
      double* obase = e < 6 ? W.V[trial] : (is_b ? W.bl[trial] : W.Hq[trial]);
      const int ostride = is_b ? 3 : 6, ooff = e < 6 ? e : (is_b ? e - 6 : e - 9);
      for (int wi = tid; wi < nlm * 16; wi += LIN_THREADS) {
        const int ll = wi >> 4;
        const int p0 = s_lpb[ll], p1 = s_lpb[ll + 1];
        REAL s0 = 0, s1 = 0;
        int p = p0;
        for (; p + 1 < p1; p += 2) {
          s0 += s_rec[16 * p + e];
          s1 += s_rec[16 * (p + 1) + e];
        }
        if (p < p1) s0 += s_rec[16 * p + e];
        const REAL sum = s0 + s1;
        if (!is_cost) obase[ostride * (size_t)(G.lm_begin + ll) + ooff] = sum;
        s_lmres[16 * ll + e] = sum;
        // first linearisation of an optimize() call: Jacobi scale of the landmark columns (Ceres EstimateScale)
        if (init && opt.dogleg && (e == 0 || e == 3 || e == 5))
          W.lm_scale[3 * (size_t)(G.lm_begin + ll) + (e == 0 ? 0 : (e == 3 ? 1 : 2))] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt((double)sum)) : 1.0;
      }
    }
    LSTAMP(45);
    // (b) per (landmark, block) pair, one lane each: Vp, bp from the pair's pieces; d and w from the trial state
    REAL vp[6] = {0, 0, 0, 0, 0, 0}, bp[3] = {0, 0, 0};
    REAL d0 = 0, d1 = 0, d2 = 0, hw = 0;
    int my_slot = 0;
    if (has_pair) {
      const int pp = s_pp[tid];
      const int p0 = pp & 0xFFFF, p1 = p0 + (pp >> 16);
      for (int p = p0; p < p1; ++p) {
        const REAL* rp = s_rec + 16 * p;
#pragma unroll
        for (int e = 0; e < 6; ++e) vp[e] += rp[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) bp[e] += rp[6 + e];
      }
      double pt[3];
      if (poses_staged) {
        pt[0] = s_pose[7 * pf_pblock], pt[1] = s_pose[7 * pf_pblock + 1], pt[2] = s_pose[7 * pf_pblock + 2];
      } else {
        const double* t = W.pose[trial] + 7 * (size_t)pf_pblock;
        pt[0] = t[0], pt[1] = t[1], pt[2] = t[2];
      }
      const double* x = s_lm + 4 * pf_plm;
      const double w = x[3];
      d0 = REAL(x[0] - pt[0] * w);
      d1 = REAL(x[1] - pt[1] * w);
      d2 = REAL(x[2] - pt[2] * w);
      hw = REAL(w);
      my_slot = (int)s_slot[tid];
    }
    __syncthreads();   // the piece records are free, the landmark sums are complete
    LSTAMP(46);
    // (d) group scalars, by wave 3 (no pair lives there) in the shadow of the pair lanes' work
    if (wave == 3) {
      double cost = 0, gm = 0, sc_gd = 0, sc_ddd = 0, sc_s2 = 0, sc_x2 = 0;
      if (lane < nlm) {
        cost = s_lmres[16 * lane + 15];
        gm = fmax(fabs(s_lmres[16 * lane + 6]), fmax(fabs(s_lmres[16 * lane + 7]), fabs(s_lmres[16 * lane + 8])));
        sc_gd = s_sc[lane], sc_ddd = s_sc[64 + lane], sc_s2 = s_sc[128 + lane], sc_x2 = s_sc[192 + lane];
      }
      cost = wave_sum_full(cost);
      gm = wave_max_full(gm);
      sc_gd = wave_sum_full(sc_gd);
      sc_ddd = wave_sum_full(sc_ddd);
      sc_s2 = wave_sum_full(sc_s2);
      sc_x2 = wave_sum_full(sc_x2);
      if (lane == 0) {
        double* gs = W.gscal[trial] + (size_t)g * GS_COUNT;
        gs[GS_COST] = cost;
        gs[GS_GD] = sc_gd;
        gs[GS_DDD] = sc_ddd;
        gs[GS_STEP2] = sc_s2;
        gs[GS_X2] = sc_x2;
        gs[GS_GMAX] = gm;
      }
    }
    // W = M^T Vp (6x3), U = M^T Vp M (upper triangle, 21), g = M^T bp (6);  M = [ -w I | e_i x d ]
    REAL Wm[18];
    {
      const REAL V3[3][3] = {{vp[0], vp[1], vp[2]}, {vp[1], vp[3], vp[4]}, {vp[2], vp[4], vp[5]}};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Wm[0 + c] = -hw * V3[0][c];
        Wm[3 + c] = -hw * V3[1][c];
        Wm[6 + c] = -hw * V3[2][c];
        Wm[9 + c] = d2 * V3[1][c] - d1 * V3[2][c];
        Wm[12 + c] = d0 * V3[2][c] - d2 * V3[0][c];
        Wm[15 + c] = d1 * V3[0][c] - d0 * V3[1][c];
      }
    }
    if (has_pair) {
      double* Wt = W.W[trial] + (size_t)(G.pair_begin + tid) * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) Wt[i] = Wm[i];
    }
    // (c) per-block J^T J / J^T r partials.  The pair's record goes to its SLOT — the records of one block (task) are contiguous
    //     there — UB entries per round, and every (block, entry) is one contiguous sum in slot (= list) order
    auto urec_k = [&](int kk) -> REAL {   // entry kk of this pair's block record: k < 21: U[ra][b] at k = ut6(ra, b); 21 + a: g[a]
      REAL v = 0;
      int k = 0;
#pragma unroll
      for (int ra = 0; ra < 7; ++ra)
#pragma unroll
        for (int b = (ra < 6 ? ra : 0); b < 6; ++b, ++k)
          if (k == kk) {
            const REAL w0 = ra < 6 ? Wm[3 * (ra < 6 ? ra : 0)] : bp[0], w1 = ra < 6 ? Wm[3 * (ra < 6 ? ra : 0) + 1] : bp[1],
                       w2 = ra < 6 ? Wm[3 * (ra < 6 ? ra : 0) + 2] : bp[2];
            v = b == 0 ? -hw * w0 : b == 1 ? -hw * w1 : b == 2 ? -hw * w2 : b == 3 ? w1 * d2 - w2 * d1 : b == 4 ? w2 * d0 - w0 * d2 : w0 * d1 - w1 * d0;
          }
      return v;
    };
#pragma unroll
    for (int r0 = 0; r0 < LIN2_REC; r0 += UB) {
      const int ne = (LIN2_REC - r0 < UB) ? LIN2_REC - r0 : UB;
      if (r0 > 0) __syncthreads();   // the previous round's sums are done with the records
      if (has_pair) {
        REAL* ur = s_rec + UB * my_slot;
#pragma unroll
        for (int k = 0; k < UB; ++k)
          if (k < ne) ur[k] = urec_k(r0 + k);
      }
      __syncthreads();
      if (r0 == 0) LSTAMP(47);
      for (int wi = tid; wi < ntask * ne; wi += LIN_THREADS) {
        const int tt = wi / ne, k = wi - tt * ne;
        int lb, le, out;
        if (tasks_cached) {
          const int* t = s_task + 6 * tt;
          lb = t[3], le = t[4], out = t[5];
        } else {
          const Task T = W.tasks[G.task_begin + tt];
          lb = T.list_begin - G.tlist_begin, le = T.list_end - G.tlist_begin, out = T.out;
        }
        REAL s0 = 0, s1 = 0;
        int j = lb;
        for (; j + 1 < le; j += 2) {
          s0 += s_rec[j * UB + k];
          s1 += s_rec[(j + 1) * UB + k];
        }
        if (j < le) s0 += s_rec[j * UB + k];
        const REAL sum = s0 + s1;
        W.gpart[trial][out + r0 + k] = sum;
        if (fast) s_aux[GROUP_LM * 9 + tt * 36 + r0 + k] = sum;   // the group's own J^T J / J^T r blocks for the reduction below
      }
    }
    LSTAMP(48);
    LSTAMP(49);
    if constexpr (FUSE) {
      __syncthreads();   // block records consumed (the tiles of the reduction take their place), J^T J blocks complete
      if (fast) {
        FuseItemsT<6> fit;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          fit.row[q] = has_pair ? pf_poff + q : -1;
          fit.lm[q] = pf_plm;
          fit.w[q][0] = Wm[3 * q], fit.w[q][1] = Wm[3 * q + 1], fit.w[q][2] = Wm[3 * q + 2];
        }
        fused_reduce_fast<RECD, 6>(W, opt, g, trial, lam_next, nlm, init != 0, pf_sc, fit, smem, s_lmres, s_aux);
      } else {
        reduce_own_group(g, trial, lam_next);
      }
    }
    LSTAMP(50);
    LSTAMP(51);
  }
#undef LSTAMP
}

// the IMU / prior factors of every window in a launch of their own (the companion of linearize2_kernel<.., SMALL = false>)
__global__ __launch_bounds__(LIN_THREADS, 2) void small_kernel(const WinPtrs* __restrict__ wins, int init) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  small_body(wins[blockIdx.y], init, blockIdx.x, smem);
}
// ... and its first half alone (imu_factor<1>: what may change an IMU term's preintegration record), where the evaluation rides in
// the decision-free Schur launch of the same slot (schur_ride_kernel, ba_schur2.hpp); grid: the IMU terms only
__global__ __launch_bounds__(LIN_THREADS, 2) void small_prepare_kernel(const WinPtrs* __restrict__ wins) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  small_body<1>(wins[blockIdx.y], 0, blockIdx.x, smem);
}

// First preintegration of IMU terms that arrive without one (okvis_ba_window::imu_sb_ref_valid = 0), started by okvis_ba_upload /
// okvis_ba_patch_window BEFORE the host builds the window's index lists, so that the 0.1 ms recursion runs while the host works
// instead of stretching the first linearise launch of the next optimisation.  Workgroup k integrates the one term of the
// stand-in window record mini[k] (its IMU arrays, noise parameters and cache pointer; everything else unset) at the bias sb0s[9 k ..]
// — imu_redo itself, so the record is the one the first evaluation would have built at that bias; imu_pre_place_kernel then
// copies it over the term's (empty) record in the uploaded window, behind the arena copy on the same stream.
__global__ __launch_bounds__(IMU_THREADS, 2) void imu_pre_kernel(const WinPtrs* __restrict__ mini, const double* __restrict__ sb0s) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  imu_redo(mini[blockIdx.x], 0, sb0s + 9 * (size_t)blockIdx.x, smem, threadIdx.x);
}
__global__ void imu_pre_place_kernel(const WinPtrs* __restrict__ wins, const int2* __restrict__ where, const ImuCacheD* __restrict__ src) {
  const int2 wf = where[blockIdx.x];
  const double* s = reinterpret_cast<const double*>(src + blockIdx.x);
  auto d = reinterpret_cast<BA_G double*>(wins[wf.x].imu_cache + wf.y);
  for (int i = threadIdx.x; i < (int)(sizeof(ImuCacheD) / 8) - 1; i += blockDim.x) d[i] = s[i];
  if (threadIdx.x == 0) {   // (the last double holds the two counters: one preintegration so far, valid at the bias it was built at —
                            // 3: imu_maybe_redo takes it for the first evaluation's only if that evaluation sees the same bias)
    wins[wf.x].imu_cache[wf.y].valid = 3;
    wins[wf.x].imu_cache[wf.y].redo_count = 1;
  }
}

}  // namespace ba
