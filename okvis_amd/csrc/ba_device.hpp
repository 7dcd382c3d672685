// Wave/workgroup helpers and the trust-region decision shared by all kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_math.hpp"
#include "ba_types.hpp"

namespace ba {

// A pointer that is read from memory (every pointer of WinPtrs) is a generic-address-space pointer for the compiler: its
// accesses become FLAT instructions, which count on the LDS counter as well, so that the next LDS wait also waits for
// them (a global store followed by an LDS read costs a full memory round trip).  as_global() states that the pointer
// refers to HBM: global_load / global_store, which only the vector-memory counter tracks.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T* as_global(T* p) {
  return (__attribute__((address_space(1))) T*)p;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// Lane exchange inside a quad with DPP (VALU speed; __shfl_xor goes through the LDS crossbar, ds_bpermute).
// CTRL 0xB1 = quad_perm:[1,0,3,2] (xor 1), 0x4E = quad_perm:[2,3,0,1] (xor 2).
template <int CTRL>
__device__ __forceinline__ double quad_xchg(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float quad_xchg(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the four lanes of a quad (all four lanes obtain it; all four must be active)
template <class T>
__device__ __forceinline__ T quad_sum(T v) {
  v += quad_xchg<0xB1>(v);
  v += quad_xchg<0x4E>(v);
  return v;
}

__device__ __forceinline__ double rsqrt_nr(double d) {
  // 1/sqrt(d): hardware estimate + two Newton steps (short dependency chain, no fp64 divide / sqrt macro)
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}

// PoseLocalParameterization::plus (PoseLocalParameterization.cpp:60-87, Transformation.hpp:246-258) as the solve kernel
// evaluates it twice per launch on its critical path: the same function as pose_oplus (ba_math.hpp) without the library
// calls.  Square roots and divisions become v_rsq_f64 + Newton steps; sin(h) / h and cos(h) of the half angle are Taylor
// polynomials of h^2 for h <= 0.5 (truncation < 5e-17: a rotation update of up to one radian), the library functions
// beyond.  About 100 instructions instead of about 700; agrees with pose_oplus to the last bit or two.
__device__ __forceinline__ void pose_oplus_dev(const double* x, const double* d, double* out) {
  out[0] = x[0] + d[0];
  out[1] = x[1] + d[1];
  out[2] = x[2] + d[2];
  const double qn2 = x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6];
  const double qi = rsqrt_nr(qn2);
  const double q[4] = {x[3] * qi, x[4] * qi, x[5] * qi, x[6] * qi};
  const double h2 = 0.25 * (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);   // (half angle)^2
  double sc, cs;
  if (h2 <= 0.25) {
    sc = fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0), -1.0 / 39916800.0),
                                                     1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
    cs = fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, fma(h2, 1.0 / 20922789888000.0, -1.0 / 87178291200.0), 1.0 / 479001600.0),
                                                              -1.0 / 3628800.0), 1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
  } else {
    const double h = sqrt(h2);
    sc = sin(h) / h;
    cs = cos(h);
  }
  const double s = 0.5 * sc;
  const double dq[4] = {s * d[3], s * d[4], s * d[5], cs};
  double qn[4];
  qmul(dq, q, qn);
  const double ni = rsqrt_nr(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
  out[3] = qn[0] * ni; out[4] = qn[1] * ni; out[5] = qn[2] * ni; out[6] = qn[3] * ni;
}

// Whole-wave reductions on the VALU: two quad permutes, row_half_mirror (0x141) and row_mirror (0x140) leave every
// lane with the total of its row of 16, four v_readlane combine the rows.  ALL 64 LANES MUST BE ACTIVE.  Every lane
// obtains the same bits; ~25 instructions instead of six ds_bpermute round trips (the butterfly of wave_sum).
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over the 16 lanes of a DPP row (all of them obtain it; all 16 must be active)
__device__ __forceinline__ double row16_sum(double v) {
  v += quad_xchg<0xB1>(v);
  v += quad_xchg<0x4E>(v);
  v += quad_xchg<0x141>(v);
  v += quad_xchg<0x140>(v);
  return v;
}
__device__ __forceinline__ double wave_sum_full(double v) {
  v += quad_xchg<0xB1>(v);
  v += quad_xchg<0x4E>(v);
  v += quad_xchg<0x141>(v);
  v += quad_xchg<0x140>(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_full(double v) {
  v = fmax(v, quad_xchg<0xB1>(v));
  v = fmax(v, quad_xchg<0x4E>(v));
  v = fmax(v, quad_xchg<0x141>(v));
  v = fmax(v, quad_xchg<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Sums of the per-group / per-factor partials of linearisation buffer `buf`, computed by ONE full wave
// in a fixed order (lane-strided accumulation + wave_sum_full) so that the Schur kernel and the solve
// kernel obtain bit-identical values.  out: cost, g.delta (landmarks), delta^T D^2 delta, |delta|^2,
// |x|^2, max|g_l|.
// (two halves: the lane-strided partial sums, which only need memory, and the wave reductions; the Schur kernels request the
// partials of BOTH buffers before the control record has arrived and reduce the one the record names)
__device__ __forceinline__ void wave_trial_partials(const WinPtrs& W, int buf, int lane, double part[6]) {
  double cost = 0, gd = 0, ddd = 0, s2 = 0, x2 = 0, gm = 0;
  const double* gs = W.gscal[buf];
  if (W.n_group <= 64 && W.n_imu <= 64) {
    // one group and one IMU factor per lane at most (every window up to configs[1]'s size): all loads leave together — as three
    // loops they were three dependent memory round trips at the head of the solve kernel and of every Schur workgroup.  The
    // same sums in the same order as the loops below (an absent term adds +0.0).
    double v[6] = {0, 0, 0, 0, 0, 0}, ic = 0, sc = 0;
    if (lane < W.n_group) {
      const double* p = gs + (size_t)lane * GS_COUNT;
      v[0] = p[GS_COST], v[1] = p[GS_GD], v[2] = p[GS_DDD], v[3] = p[GS_STEP2], v[4] = p[GS_X2], v[5] = p[GS_GMAX];
    }
    if (lane < W.n_imu) ic = W.imu_lin[buf][(size_t)lane * IMU_LIN_STRIDE + IMU_COST];
    if (lane == 0) sc = W.small_cost[buf][0];
    cost = (0.0 + v[0]) + ic;
    if (lane == 0) cost += sc;
    part[0] = cost, part[1] = 0.0 + v[1], part[2] = 0.0 + v[2], part[3] = 0.0 + v[3], part[4] = 0.0 + v[4], part[5] = fmax(0.0, v[5]);
    return;
  }
  for (int g = lane; g < W.n_group; g += 64) {
    const double* p = gs + (size_t)g * GS_COUNT;
    cost += p[GS_COST];
    gd += p[GS_GD];
    ddd += p[GS_DDD];
    s2 += p[GS_STEP2];
    x2 += p[GS_X2];
    gm = fmax(gm, p[GS_GMAX]);
  }
  for (int f = lane; f < W.n_imu; f += 64) cost += W.imu_lin[buf][(size_t)f * IMU_LIN_STRIDE + IMU_COST];
  if (lane == 0) cost += W.small_cost[buf][0];
  part[0] = cost, part[1] = gd, part[2] = ddd, part[3] = s2, part[4] = x2, part[5] = gm;
}
__device__ __forceinline__ void wave_trial_reduce(const double part[6], double out[6]) {
  out[0] = wave_sum_full(part[0]);
  out[1] = wave_sum_full(part[1]);
  out[2] = wave_sum_full(part[2]);
  out[3] = wave_sum_full(part[3]);
  out[4] = wave_sum_full(part[4]);
  out[5] = wave_max_full(part[5]);
}
__device__ __forceinline__ void wave_trial_sums(const WinPtrs& W, int buf, int lane, double out[6]) {
  double part[6];
  wave_trial_partials(W, buf, lane, part);
  wave_trial_reduce(part, out);
}

struct Decision {
  int accept;     // 1 = the pending trial becomes the accepted state
  int term;       // 0 = continue, else termination reason (3 parameter tol, 4 radius, 5 numeric)
  double radius, decrease_factor;
  double rho, model_change;
};

// Accept/reject + trust-region radius update for the pending trial (Ceres LevenbergMarquardtStrategy
// StepAccepted/StepRejected + TrustRegionMinimizer step evaluation, restated; DESIGN.md "solver policy").
// Contraction-free so that every kernel evaluating it gets the same bits (no fast-math anywhere: without contraction the
// operations are plain IEEE whatever they are inlined into).  The Schur kernels inline it (as a real call it cost them ~1 us
// of scratch traffic for the callee-saved registers); the solve kernel, which has no registers to spare, calls it (decide /
// decide_dl below).
__device__ __forceinline__ void decide_inl(const Ctrl* c, const OptD* o, const double sums[6], Decision* d) {
#pragma clang fp contract(off)
  d->accept = 0;
  d->term = 0;
  d->radius = c->radius;
  d->decrease_factor = c->decrease_factor;
  d->rho = 0;
  d->model_change = 0;
  if (c->first || o->gauss_newton) {
    d->accept = 1;
    return;
  }
  const double gd = c->gd_p + sums[1];
  const double ddd = c->ddd_p + sums[2];
  const double step2 = c->step2_p + sums[3];
  const double x2 = c->x2_p + sums[4];
  const double model = -0.5 * gd + 0.5 * c->lambda * ddd;
  d->model_change = model;
  if (!(model > 0.0)) {  // invalid step: handled like a rejected one
    d->radius = c->radius / c->decrease_factor;
    d->decrease_factor = c->decrease_factor * 2.0;
    if (d->radius < o->min_radius) d->term = 5;
    return;
  }
  if (o->parameter_tolerance > 0.0 &&
      sqrt(step2) <= o->parameter_tolerance * (sqrt(x2) + o->parameter_tolerance)) {
    d->term = 3;
    return;
  }
  const double rho = (c->cost - sums[0]) / model;
  d->rho = rho;
  if (rho > o->min_relative_decrease) {
    d->accept = 1;
    const double t = 2.0 * rho - 1.0;
    double r = c->radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
    d->radius = fmin(o->max_radius, r);
    d->decrease_factor = 2.0;
  } else {
    d->radius = c->radius / c->decrease_factor;
    d->decrease_factor = c->decrease_factor * 2.0;
    if (d->radius < o->min_radius) d->term = 4;
  }
}

// ---- dogleg strategy (the reference's configuration, Estimator.cpp:858) -----------------------------------------
// Decision for the pending trial under Ceres' TrustRegionMinimizer + DoglegStrategy rules (restated; DESIGN.md
// "solver policy").  sums = wave_trial_sums of the trial buffer.  Evaluated bit-identically by the Schur kernel and
// the solve kernel (not inlined, contraction-free).
struct DecisionDL {
  int accept;         // the pending trial becomes the accepted state
  int term;           // 0 = continue, else termination reason (1 function tol, 3 parameter tol, 4 radius, 5 invalid steps,
                      // 6 = iteration budget of this call used up)
  int explicit_next;  // the next trial is an explicit dogleg step from the stored Gauss-Newton point (no new solve)
  int judged;         // an iteration was completed (accepted / rejected / invalid); 0 = mis-speculation or first
  int invalid_steps, have_tot;
  double radius, mu, rho, model_change, tot_C, tot_E;
};
__device__ __forceinline__ void decide_dl_inl(const Ctrl* c, const OptD* o, const double sums[6], int final_call, DecisionDL* d) {
#pragma clang fp contract(off)
  d->accept = 0;
  d->term = 0;
  d->explicit_next = 0;
  d->judged = 0;
  d->invalid_steps = c->invalid_steps;
  d->have_tot = c->have_tot;
  d->radius = c->radius;
  d->mu = c->mu;
  d->rho = 0;
  d->model_change = 0;
  d->tot_C = c->tot_C;
  d->tot_E = c->tot_E;
  if (c->first) {
    d->accept = 1;
  } else if (o->gauss_newton) {
    d->accept = 1;
    d->judged = 1;
  } else {
    double model, dl_norm;
    bool have = true;
    if (c->tr_kind == 0) {  // the speculative Gauss-Newton point: is it inside the trust region?
      const double C = c->gd_p + sums[1], E = c->ddd_p + sums[2];
      d->tot_C = C;
      d->tot_E = E;
      d->have_tot = 1;
      dl_norm = sqrt(E);
      model = -0.5 * C + 0.5 * c->mu * E;
      if (dl_norm > c->radius) {
        d->explicit_next = 1;  // no: replace the trial by the proper dogleg step, nothing judged
        have = false;
      }
    } else {
      model = c->pend_model;
      dl_norm = c->dl_norm;
    }
    if (have) {
      d->judged = 1;
      d->model_change = model;
      if (model < 0.0) {  // invalid step (StepIsInvalid)
        d->invalid_steps = c->invalid_steps + 1;
        if (d->invalid_steps >= o->max_invalid) d->term = 5;
        d->mu = c->mu * DL_MU_INCREASE;
      } else {
        d->invalid_steps = 0;
        const double step2 = c->step2_p + sums[3], x2 = c->x2_p + sums[4];
        const double cost_change = c->cost - sums[0];
        if (o->parameter_tolerance > 0.0 && sqrt(step2) <= o->parameter_tolerance * (sqrt(x2) + o->parameter_tolerance)) {
          d->term = 3;
        } else if (o->function_tolerance > 0.0 && fabs(cost_change) < o->function_tolerance * c->cost) {
          d->term = 1;  // Ceres <= 1.10 returns here without taking the step
        } else {
          const double rho = cost_change / model;
          d->rho = rho;
          if (rho > o->min_relative_decrease) {  // StepAccepted
            d->accept = 1;
            if (rho < 0.25) d->radius = c->radius * 0.5;
            if (rho > 0.75) d->radius = fmax(c->radius, 3.0 * dl_norm);
            d->mu = fmax(DL_MIN_MU, 2.0 * c->mu / DL_MU_INCREASE);
            d->have_tot = 0;
          } else {  // StepRejected: only the interpolation is redone
            d->radius = c->radius * 0.5;
            d->explicit_next = 1;
          }
          if (d->radius < o->min_radius) d->term = 4;
        }
      }
    }
  }
  // a NEW iteration would start now (anything but the redo of a mis-speculated trial): is there budget left?
  if (!final_call && !d->term && !(d->explicit_next && !d->judged) && c->iter >= c->max_iter) d->term = 6;
}

__device__ __noinline__ void decide(const Ctrl* c, const OptD* o, const double sums[6], Decision* d) { decide_inl(c, o, sums, d); }
__device__ __noinline__ void decide_dl(const Ctrl* c, const OptD* o, const double sums[6], int final_call, DecisionDL* d) {
  decide_dl_inl(c, o, sums, final_call, d);
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Damping diagonal of one column.  LM: clamp(h) with h = diag(J^T J).  Dogleg: Ceres works on the column-scaled
// Jacobian (scale s, estimated at the first linearisation): diagonal_^2 = clamp(s^2 h); in unscaled variables the
// regulariser mu * diagonal_^2 becomes mu * clamp(s^2 h) / s^2.
__device__ __forceinline__ double damp_diag(double h, double s, const OptD& o) {
  if (!o.dogleg) return clampd(h, o.min_lm_diag2, o.max_lm_diag2);
  const double s2 = s * s;
  return clampd(s2 * h, o.min_lm_diag2, o.max_lm_diag2) / s2;
}

}  // namespace ba
