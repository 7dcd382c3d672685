// IMU factors, small priors and the marginalisation prior at the trial state (fp64) — the first
// (max_imu + 1) workgroups of the linearise launch (ba_linearize.hpp).
//
// n_imu workgroups (one per ImuError) + 1 workgroup for all PoseError / SpeedAndBiasError /
// RelativePoseError / MarginalizationError terms of the window.
//
// IMU workgroup (ImuError::EvaluateWithMinimalJacobians, ImuError.cpp:514-685):
//   * bias check |b_g - b_g,ref| * dt > 1e-4 (or first use) -> on-device re-preintegration
//     (ImuError::redoPreintegration, ImuError.cpp:76-284), parallel over the raw samples:
//       per-sample quantities in parallel, the ordered prefix products/sums by one work-item per sample
//       (O(n^2) adds but only n deep), the 15x15 covariance recursion P <- F P F^T + Q with one
//       work-item per matrix entry (2 barriers per sample), then P^-1 and its Cholesky factor in LDS.
//     The cache (the reference's `mutable` members) lives in HBM and persists across iterations,
//     including across rejected steps — exactly the reference's behaviour.
//   * residual (15) and the 15x30 minimal Jacobian  sqrtInfo * [F0 | F1]
#pragma once
#include "ba_device.hpp"

namespace ba {

constexpr int IMU_N = 32;  // integration steps per LDS chunk of the re-preintegration (factor length is unbounded)
// LDS layout (doubles) of the re-preintegration scratch, SoA over steps
struct ImuLds {
  static constexpr int DQ = 0;                   // 4N
  static constexpr int DT = DQ + 4 * IMU_N;      // N
  static constexpr int AB = DT + IMU_N;          // 3N   bias-corrected mean acceleration
  static constexpr int JRDT = AB + 3 * IMU_N;    // 9N   rightJacobian(omega dt) * dt
  static constexpr int RINV = JRDT + 9 * IMU_N;  // 9N   R(dq^-1); later aliased by dp_term
  static constexpr int DQP = RINV + 9 * IMU_N;   // 4(N+1) prefix products Delta_q_k
  static constexpr int CINT = DQP + 4 * (IMU_N + 1);  // 9N  0.5 (C+C1) dt
  static constexpr int AINT = CINT + 9 * IMU_N;  // 3N
  static constexpr int DAL = AINT + 3 * IMU_N;   // 9N   C1 Jr dt
  static constexpr int C1 = DAL + 9 * IMU_N;     // 9N
  static constexpr int ADBL = C1 + 9 * IMU_N;    // 3N
  static constexpr int CDBL = ADBL + 3 * IMU_N;  // 9N
  static constexpr int B012 = CDBL + 9 * IMU_N;  // 9N
  static constexpr int GG = B012 + 9 * IMU_N;    // 9N
  static constexpr int DVT = GG + 9 * IMU_N;     // 9N
  static constexpr int SG2 = DVT + 9 * IMU_N;    // N
  static constexpr int SA2 = SG2 + IMU_N;        // N
  static constexpr int PM = SA2 + IMU_N;         // 225
  static constexpr int TM = PM + 225;            // 225
  static constexpr int FM = TM + 225;            // 450  F = [F0 | F1]
  static constexpr int EV = FM + 450;            // 16   error vector
  static constexpr int CA = EV + 16;             // cache copy: see below (300)
  // exclusive prefixes (index k = value BEFORE step k; index ns = carry into the next chunk)
  static constexpr int CINTP = CA + 320;                 // 9(N+1)
  static constexpr int AINTP = CINTP + 9 * (IMU_N + 1);  // 3(N+1)
  static constexpr int CROSSP = AINTP + 3 * (IMU_N + 1); // 9(N+1)
  static constexpr int DVP = CROSSP + 9 * (IMU_N + 1);   // 9(N+1)
  static constexpr int TSBUF = DVP + 9 * (IMU_N + 1);    // (N+8) timestamps staged for the loop control (as long long)
  // coefficient block of the covariance transition F_k = I + N_k per step (signs folded in, so that every matrix entry
  // of F P and (F P) F^T is a plain sum of coefficient * entry):  -[adbl]x (9) | dt | dp_term (9) | b012 (9) | -dt C1 (9)
  // | -[aint]x (9) | dv_term (9) | -C_int (9) | noise diagonal of p, alpha, v, b_g, b_a (5) | 0
  static constexpr int CB = TSBUF + IMU_N + 8;
  static constexpr int CB_STRIDE = 70;
  static constexpr int TOTAL = CB + CB_STRIDE * IMU_N;
};
typedef double imu_v4 __attribute__((ext_vector_type(4)));
enum { CB_A1 = 0, CB_DT = 9, CB_DPT = 10, CB_B012 = 19, CB_DTC1 = 28, CB_SKI = 37, CB_DVT = 46, CB_CINT = 55, CB_NOISE = 64, CB_ZERO = 69 };
// compact LDS layout of the evaluate kernel (no re-preintegration scratch): J | F | e | cache copy
struct EvalLds {
  static constexpr int PM = 0;      // 450: J = sqrtInfo * F
  static constexpr int FM = 450;    // 450: F = [F0 | F1]
  static constexpr int EV = 900;    // 16
  static constexpr int CA = 916;    // 320
  static constexpr int TOTAL = 1236;
};
// cache copy offsets inside CA
enum { CA_DQ = 0, CA_CI = 4, CA_CD = 13, CA_AI = 22, CA_AD = 25, CA_DA = 28, CA_DV = 37, CA_DP = 46, CA_SI = 55 };

__device__ __forceinline__ void ld9(const double* p, int k, double* o) {
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = p[9 * k + i];
}
__device__ __forceinline__ void st9(double* p, int k, const double* o) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[9 * k + i] = o[i];
}

// Cholesky of a 15x15 SPD matrix by ONE wave.  Lane j (< 15) holds column j of the full symmetric matrix in c[0..14].
// Step k: the pivot column is read from lane k with v_readlane (wave-uniform l_i = A_ik / sqrt(A_kk)); the lane's own
// L_jk is its entry of row k (= A_jk by symmetry, the whole trailing matrix is kept up to date), so the update
// c_i -= l_i L_jk needs nothing else.  Lane t >= k stores L_tk to out[rs t + cs k]; invd[k] = 1 / L_kk on every lane.
__device__ __forceinline__ void chol15_wave(double (&c)[15], double (&invd)[15], double* out, int rs, int cs, int lane, bool act) {
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    const double d = readlane_f64(c[k], k);
    const double inv = rsqrt_nr(d > 0.0 ? d : 1.0);
    invd[k] = inv;
    const double mine = c[k] * inv;   // L_jk for lanes j >= k
    if (act && lane >= k) out[rs * lane + cs * k] = mine;
#pragma unroll
    for (int i = k + 1; i < 15; ++i) {
      const double li = readlane_f64(c[i], k) * inv;
      c[i] -= li * mine;
    }
  }
}

// diagnostics (debug_arrays): cycles per stage of the re-preintegration of factor 0, accumulated over the chunks
#define RSTAMP(k) do { if (W.prof && f == 0 && tid == 0) { const long long t_ = clock64(); W.prof[50 + (k)] += (double)(t_ - rs_t); rs_t = t_; } } while (0)
__device__ void imu_redo(const WinPtrs& W, int f, const double* sb0, double* lds, int tid) {
  long long rs_t = clock64();
  if (W.prof && f == 0 && tid < 12) W.prof[50 + tid] = 0.0;
  // The integration steps are processed in chunks of IMU_N (the LDS scratch), with the running state
  // (Delta_q, the integrals, the cross matrix, dv/db_g, the covariance) carried from chunk to chunk, so a
  // factor may span any number of raw samples (the reference copies the whole deque into every ImuError).
  __shared__ int s_it[IMU_N], s_flag[IMU_N];
  __shared__ long long s_ts[IMU_N], s_tn[IMU_N];
  __shared__ int s_nsteps, s_next_it, s_started, s_finished, s_first;
  __shared__ long long s_time;
  __shared__ double c_Dq[4];  // carry of Delta_q; the other carries sit at index 0 of the prefix arrays
  __shared__ double t_Cdbl[9], t_adbl[3], t_dal[9], t_dp[9];              // running totals
  const int n = W.imu_s_count[f];
  const long long* ts = W.imu_s_t + W.imu_s_begin[f];
  const double* gyr = W.imu_s_gyr + 3 * (size_t)W.imu_s_begin[f];
  const double* acc = W.imu_s_acc + 3 * (size_t)W.imu_s_begin[f];
  const long long t0 = W.imu_t0[f], t1 = W.imu_t1[f];
  const ImuParamsD prm = W.imu;
  const double bg[3] = {sb0[3], sb0[4], sb0[5]}, ba[3] = {sb0[6], sb0[7], sb0[8]};
  double* P = lds + ImuLds::PM;
  double* T = lds + ImuLds::TM;
  if (tid == 0) {
    s_next_it = 0;
    s_started = 0;
    s_finished = 0;
    s_time = t0;
    c_Dq[0] = c_Dq[1] = c_Dq[2] = 0.0;
    c_Dq[3] = 1.0;
  }
  if (tid < 9) {
    lds[ImuLds::CINTP + tid] = 0; lds[ImuLds::CROSSP + tid] = 0; lds[ImuLds::DVP + tid] = 0;
    t_Cdbl[tid] = 0; t_dal[tid] = 0; t_dp[tid] = 0;
  }
  if (tid < 3) {
    lds[ImuLds::AINTP + tid] = 0; t_adbl[tid] = 0;
  }
  if (tid < 225) P[tid] = 0.0;
  if (tid == 0) s_first = n;
  __syncthreads();
  // the leading samples the reference skips with `continue` (nexttime - time <= 0 while time is still t0,
  // ImuError.cpp:128-130): find the first step that advances, in parallel (same result as the serial scan)
  {
    int first = n;
    for (int it = tid; it < n; it += IMU_THREADS) {
      long long nexttime = (it + 1 == n) ? t1 : ts[it + 1];
      if (t1 < nexttime) nexttime = t1;
      if (nexttime - t0 > 0) {
        first = it;
        break;
      }
    }
    if (first < n) atomicMin(&s_first, first);
  }
  __syncthreads();
  if (tid == 0) s_next_it = s_first;
  long long* tsbuf = reinterpret_cast<long long*>(lds + ImuLds::TSBUF);
  __syncthreads();

  for (;;) {
    // ---- stage 0: loop control of ImuError.cpp:113-150,259-260 (integer time logic) by one work-item, on
    //      timestamps staged in LDS
    const int base = s_next_it;
    for (int j = tid; j < IMU_N + 8; j += IMU_THREADS) tsbuf[j] = (base + j < n) ? ts[base + j] : t1;
    __syncthreads();
    // fast path (wave 0, one lane per candidate step): when every candidate step of this chunk advances the
    // time (strictly increasing timestamps — the normal case) step j is simply sample base + j, so the serial
    // scan below is not needed; any non-advancing step falls back to it (identical results by construction)
    bool fast_done = false;
    if (tid < 64) {
      const int j = tid;
      const int it = base + j;
      const bool cand = j < IMU_N && it < n && !s_finished;
      long long nexttime = t1, before = s_time;
      int flag = 0;
      if (cand) {
        nexttime = (it + 1 == n) ? t1 : tsbuf[j + 1];
        if (t1 < nexttime) {
          nexttime = t1;
          flag |= 1;
        }
        if (j > 0) {
          before = tsbuf[j];
          if (t1 < before) before = t1;
        }
        if (j == 0 && !s_started) flag |= 2;
      }
      const bool adv = cand && (nexttime - before > 0);
      const bool hit_end = cand && (nexttime == t1);
      const unsigned long long m_cand = __ballot(cand), m_adv = __ballot(adv), m_end = __ballot(hit_end);
      // steps are executed up to and including the first one that reaches t1
      int ncand = __popcll(m_cand);
      const int first_end = m_end ? (__ffsll((long long)m_end) - 1) : 64;
      const int nsteps = first_end < ncand ? first_end + 1 : ncand;
      const unsigned long long need = (nsteps >= 64) ? ~0ULL : ((1ULL << nsteps) - 1ULL);
      const bool ok = nsteps > 0 && ((m_adv & need) == need);
      if (ok) {
        if (j < nsteps) {
          s_it[j] = it;
          s_flag[j] = flag;
          s_ts[j] = before;
          s_tn[j] = nexttime;
        }
        if (j == nsteps - 1) {
          s_nsteps = nsteps;
          s_next_it = it + 1;
          s_time = nexttime;
          s_started = 1;
          s_finished = (nexttime == t1 || it + 1 >= n) ? 1 : 0;
        }
      }
      fast_done = ok;
      fast_done = __shfl(fast_done ? 1 : 0, 0) != 0;
    }
    if (tid == 0 && !fast_done) {
      long long time = s_time;
      bool started = s_started != 0;
      int k = 0, it = base;
      bool fin = s_finished != 0;
      for (; it < n && k < IMU_N && !fin; ++it) {
        const int jn = it + 1 - base;
        long long nexttime = (it + 1 == n) ? t1 : (jn < IMU_N + 8 ? tsbuf[jn] : ts[it + 1]);
        int flag = 0;
        if (t1 < nexttime) {
          nexttime = t1;
          flag |= 1;  // interpolate the second sample to t1
        }
        if (nexttime - time <= 0) continue;
        if (!started) {
          started = true;
          flag |= 2;  // interpolate the first sample to t0
        }
        s_it[k] = it;
        s_flag[k] = flag;
        s_ts[k] = time;
        s_tn[k] = nexttime;
        ++k;
        time = nexttime;
        if (nexttime == t1) fin = true;
      }
      if (it >= n) fin = true;
      s_nsteps = k;
      s_next_it = it;
      s_time = time;
      s_started = started ? 1 : 0;
      s_finished = fin ? 1 : 0;
    }
    __syncthreads();
    const int ns = s_nsteps;
    if (ns == 0) break;
    RSTAMP(0);
    // ---- stage 1: per-step quantities
    if (tid < ns) {
      const int it = s_it[tid];
      const int nx = (it + 1 < n) ? it + 1 : it;
      double w0[3] = {gyr[3 * it], gyr[3 * it + 1], gyr[3 * it + 2]};
      double a0[3] = {acc[3 * it], acc[3 * it + 1], acc[3 * it + 2]};
      double w1[3] = {gyr[3 * nx], gyr[3 * nx + 1], gyr[3 * nx + 2]};
      double a1[3] = {acc[3 * nx], acc[3 * nx + 1], acc[3 * nx + 2]};
      const double dt = ns_to_sec(s_tn[tid] - s_ts[tid]);
      if (s_flag[tid] & 1) {
        const long long raw_next = (it + 1 == n) ? t1 : ts[it + 1];
        const double interval = ns_to_sec(raw_next - ts[it]);
        const double r = dt / interval;
        for (int c = 0; c < 3; ++c) {
          w1[c] = (1.0 - r) * w0[c] + r * w1[c];
          a1[c] = (1.0 - r) * a0[c] + r * a1[c];
        }
      }
      if (s_flag[tid] & 2) {
        const double r = dt / ns_to_sec(s_tn[tid] - ts[it]);
        for (int c = 0; c < 3; ++c) {
          w0[c] = r * w0[c] + (1.0 - r) * w1[c];
          a0[c] = r * a0[c] + (1.0 - r) * a1[c];
        }
      }
      double sg = prm.sigma_g_c, sa = prm.sigma_a_c;
      bool gs = false, as = false;
      for (int c = 0; c < 3; ++c) {
        gs = gs || fabs(w0[c]) > prm.g_max || fabs(w1[c]) > prm.g_max;
        as = as || fabs(a0[c]) > prm.a_max || fabs(a1[c]) > prm.a_max;
      }
      if (gs) sg *= 100;
      if (as) sa *= 100;
      double om[3], ab[3];
      for (int c = 0; c < 3; ++c) {
        om[c] = 0.5 * (w0[c] + w1[c]) - bg[c];
        ab[c] = 0.5 * (a0[c] + a1[c]) - ba[c];
      }
      const double th = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]) * 0.5 * dt;
      const double sc = sinc(th), ct = cos(th);
      double dq[4] = {sc * om[0] * 0.5 * dt, sc * om[1] * 0.5 * dt, sc * om[2] * 0.5 * dt, ct};
      double phi[3] = {om[0] * dt, om[1] * dt, om[2] * dt};
      double Jr[9], dqi[4], Ri[9];
      right_jacobian(phi, Jr);
      for (int c = 0; c < 9; ++c) Jr[c] *= dt;
      qinv(dq, dqi);
      qrot(dqi, Ri);
      for (int c = 0; c < 4; ++c) lds[ImuLds::DQ + 4 * tid + c] = dq[c];
      lds[ImuLds::DT + tid] = dt;
      for (int c = 0; c < 3; ++c) lds[ImuLds::AB + 3 * tid + c] = ab[c];
      st9(lds + ImuLds::JRDT, tid, Jr);
      st9(lds + ImuLds::RINV, tid, Ri);
      lds[ImuLds::SG2 + tid] = dt * sg * sg;
      lds[ImuLds::SA2 + tid] = dt * sa * sa;
    }
    __syncthreads();
    RSTAMP(1);
    // ---- stage 2: Delta_q_k = Delta_q_carry (x) dq_0 (x) ... (x) dq_(k-1), the reference's product order, by
    //      one work-item; meanwhile (another wave) the recursion cross_(k+1) = R(dq_k)^T cross_k + Jr_k dt_k,
    //      one work-item per column of `cross`
    if (tid == 0) {
      double q[4] = {c_Dq[0], c_Dq[1], c_Dq[2], c_Dq[3]};
      for (int c = 0; c < 4; ++c) lds[ImuLds::DQP + c] = q[c];
      for (int j = 0; j < ns; ++j) {
        double t[4];
        qmul(q, lds + ImuLds::DQ + 4 * j, t);
        q[0] = t[0]; q[1] = t[1]; q[2] = t[2]; q[3] = t[3];
        for (int c = 0; c < 4; ++c) lds[ImuLds::DQP + 4 * (j + 1) + c] = q[c];
      }
    } else if (tid >= 64 && tid < 67) {
      const int col = tid - 64;
      double x0 = lds[ImuLds::CROSSP + col], x1 = lds[ImuLds::CROSSP + 3 + col], x2 = lds[ImuLds::CROSSP + 6 + col];
      for (int j = 0; j < ns; ++j) {
        const double* Ri = lds + ImuLds::RINV + 9 * j;
        const double* Jd = lds + ImuLds::JRDT + 9 * j;
        const double y0 = Ri[0] * x0 + Ri[1] * x1 + Ri[2] * x2 + Jd[col];
        const double y1 = Ri[3] * x0 + Ri[4] * x1 + Ri[5] * x2 + Jd[3 + col];
        const double y2 = Ri[6] * x0 + Ri[7] * x1 + Ri[8] * x2 + Jd[6 + col];
        x0 = y0; x1 = y1; x2 = y2;
        lds[ImuLds::CROSSP + 9 * (j + 1) + col] = x0;
        lds[ImuLds::CROSSP + 9 * (j + 1) + 3 + col] = x1;
        lds[ImuLds::CROSSP + 9 * (j + 1) + 6 + col] = x2;
      }
    }
    __syncthreads();
    RSTAMP(2);
    // ---- stage 3 (every stage keeps its temporaries local and re-reads what it needs from LDS: nothing but
    //      `tid` stays live in registers across the barriers — the launch is capped at 256 registers)
    if (tid < ns) {
      double C[9], CC[9], ab[3], C1m[9];
      qrot(lds + ImuLds::DQP + 4 * tid, C);
      qrot(lds + ImuLds::DQP + 4 * (tid + 1), C1m);
      const double dt = lds[ImuLds::DT + tid];
      for (int c = 0; c < 3; ++c) ab[c] = lds[ImuLds::AB + 3 * tid + c];
      for (int c = 0; c < 9; ++c) CC[c] = C[c] + C1m[c];
      double t9[9], t3[3], h[9];
      for (int c = 0; c < 9; ++c) h[c] = 0.5 * CC[c];
      for (int c = 0; c < 9; ++c) t9[c] = h[c] * dt;
      st9(lds + ImuLds::CINT, tid, t9);
      mat3_vec(h, ab, t3);
      for (int c = 0; c < 3; ++c) lds[ImuLds::AINT + 3 * tid + c] = t3[c] * dt;
      double Jr[9];
      ld9(lds + ImuLds::JRDT, tid, Jr);
      mat3_mul(C1m, Jr, t9);
      st9(lds + ImuLds::DAL, tid, t9);
      st9(lds + ImuLds::C1, tid, C1m);
    }
    __syncthreads();
    RSTAMP(3);
    // ---- stage 4: ordered prefix sums of the integrals, one work-item per component
    if (tid < 12) {
      const int base_p = tid < 9 ? ImuLds::CINTP + tid : ImuLds::AINTP + (tid - 9);
      const int base_i = tid < 9 ? ImuLds::CINT + tid : ImuLds::AINT + (tid - 9);
      const int st = tid < 9 ? 9 : 3;
      double a = lds[base_p];
      for (int j = 0; j < ns; ++j) {
        a += lds[base_i + st * j];
        lds[base_p + st * (j + 1)] = a;
      }
    }
    __syncthreads();
    RSTAMP(4);
    // ---- stage 5
    if (tid < ns) {
      double G[9], C[9], CC[9], ab[3];
      const double dt = lds[ImuLds::DT + tid];
      qrot(lds + ImuLds::DQP + 4 * tid, C);
      for (int c = 0; c < 3; ++c) ab[c] = lds[ImuLds::AB + 3 * tid + c];
      for (int c = 0; c < 9; ++c) CC[c] = C[c] + lds[ImuLds::C1 + 9 * tid + c];
      double Cint[9], aint[3], cross[9], cross1[9];
      ld9(lds + ImuLds::CINTP, tid, Cint);
      for (int c = 0; c < 3; ++c) aint[c] = lds[ImuLds::AINTP + 3 * tid + c];
      ld9(lds + ImuLds::CROSSP, tid, cross);
      ld9(lds + ImuLds::CROSSP, tid + 1, cross1);
      double q[9], t3[3], t9[9];
      for (int c = 0; c < 9; ++c) q[c] = 0.25 * CC[c];
      mat3_vec(q, ab, t3);
      double adbl_k[3], sk[9];
      for (int c = 0; c < 3; ++c) {
        adbl_k[c] = aint[c] * dt + t3[c] * dt * dt;
        lds[ImuLds::ADBL + 3 * tid + c] = adbl_k[c];
      }
      cross_mx(adbl_k, sk);
      {
        double* cb = lds + ImuLds::CB + ImuLds::CB_STRIDE * tid;
        for (int c = 0; c < 9; ++c) cb[CB_A1 + c] = -sk[c];
        const double ai_k[3] = {lds[ImuLds::AINT + 3 * tid], lds[ImuLds::AINT + 3 * tid + 1], lds[ImuLds::AINT + 3 * tid + 2]};
        cross_mx(ai_k, sk);
        for (int c = 0; c < 9; ++c) cb[CB_SKI + c] = -sk[c];
      }
      for (int c = 0; c < 9; ++c) t9[c] = Cint[c] * dt + q[c] * dt * dt;
      st9(lds + ImuLds::CDBL, tid, t9);
      for (int c = 0; c < 9; ++c) t9[c] = -Cint[c] * dt + q[c] * dt * dt;
      st9(lds + ImuLds::B012, tid, t9);
      double ax[9], C1m[9], u[9], v[9];
      cross_mx(ab, ax);
      ld9(lds + ImuLds::C1, tid, C1m);
      mat3_mul(C, ax, u);
      mat3_mul(u, cross, v);
      for (int c = 0; c < 9; ++c) G[c] = v[c];
      mat3_mul(C1m, ax, u);
      mat3_mul(u, cross1, v);
      for (int c = 0; c < 9; ++c) G[c] += v[c];
      st9(lds + ImuLds::GG, tid, G);
      for (int c = 0; c < 9; ++c) t9[c] = 0.5 * dt * G[c];
      st9(lds + ImuLds::DVT, tid, t9);
    }
    __syncthreads();
    RSTAMP(5);
    // ---- stage 6: prefix of dv/db_g (one work-item per component), then dp_term (aliases RINV, no longer needed)
    if (tid < 9) {
      double a = lds[ImuLds::DVP + tid];
      for (int j = 0; j < ns; ++j) {
        a += lds[ImuLds::DVT + 9 * j + tid];
        lds[ImuLds::DVP + 9 * (j + 1) + tid] = a;
      }
    }
    __syncthreads();
    if (tid < ns) {
      double t9[9];
      const double dt = lds[ImuLds::DT + tid];
      for (int c = 0; c < 9; ++c)
        t9[c] = dt * lds[ImuLds::DVP + 9 * tid + c] + 0.25 * dt * dt * lds[ImuLds::GG + 9 * tid + c];
      st9(lds + ImuLds::RINV, tid, t9);
      // the rest of this step's coefficient block (ImuError.cpp:209-249); operands first, stores afterwards
      double* cb = lds + ImuLds::CB + ImuLds::CB_STRIDE * tid;
      double b9[9], c9[9], d9[9], i9[9];
      ld9(lds + ImuLds::B012, tid, b9);
      ld9(lds + ImuLds::C1, tid, c9);
      ld9(lds + ImuLds::DVT, tid, d9);
      ld9(lds + ImuLds::CINT, tid, i9);
      const double s2a = lds[ImuLds::SG2 + tid], s2v = lds[ImuLds::SA2 + tid];
      cb[CB_DT] = dt;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        cb[CB_DPT + c] = t9[c];
        cb[CB_B012 + c] = b9[c];
        cb[CB_DTC1 + c] = -(dt * c9[c]);
        cb[CB_DVT + c] = d9[c];
        cb[CB_CINT + c] = -i9[c];
      }
      cb[CB_NOISE + 0] = 0.5 * dt * dt * s2v;
      cb[CB_NOISE + 1] = s2a;
      cb[CB_NOISE + 2] = s2v;
      cb[CB_NOISE + 3] = dt * prm.sigma_gw_c * prm.sigma_gw_c;
      cb[CB_NOISE + 4] = dt * prm.sigma_aw_c * prm.sigma_aw_c;
      cb[CB_ZERO] = 0.0;
    }
    __syncthreads();
    RSTAMP(6);
    // ---- stage 8: covariance recursion over this chunk on the fp64 matrix core, ONE wave, no barrier and no LDS
    //      round trip inside the chain.  P (15x15 padded to 16x16) lives in the accumulator layout of
    //      v_mfma_f64_16x16x4_f64 (lane l, register r = entry [(l >> 4) + 4 r][l & 15]).  Feeding register r of a
    //      matrix U in that layout as the A operand and register r of V as the B operand of the r-th k-block gives
    //      sum_k U[k][m] V[k][n] = (U^T V)[m][n], again in accumulator layout.  With G = F^T:
    //          Z = P^T G = (F P)^T,     P' = Z^T G + Q = F P F^T + Q          (ImuError.cpp:228-249)
    //      so the recursion never leaves the registers and needs no transposition (and no symmetry assumption).
    //      F = I + N with the 3x3 blocks of the coefficient record of the step; every lane fetches its four entries of
    //      G (and its noise entry, if it holds a diagonal element) through per-lane offsets into that record.
    if ((tid >> 6) == 0) {
      const int col = tid & 15, rb = tid >> 4;
      int go[4], qo[4];
      double one[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = col, j = rb + 4 * r;   // G[j][i] = F[i][j]
        int o = CB_ZERO;
        if (i < 3) {
          if (j >= 3 && j < 6) o = CB_A1 + 3 * i + (j - 3);
          else if (j == 6 + i) o = CB_DT;
          else if (j >= 9 && j < 12) o = CB_DPT + 3 * i + (j - 9);
          else if (j >= 12 && j < 15) o = CB_B012 + 3 * i + (j - 12);
        } else if (i < 6) {
          if (j >= 9 && j < 12) o = CB_DTC1 + 3 * (i - 3) + (j - 9);
        } else if (i < 9) {
          if (j >= 3 && j < 6) o = CB_SKI + 3 * (i - 6) + (j - 3);
          else if (j >= 9 && j < 12) o = CB_DVT + 3 * (i - 6) + (j - 9);
          else if (j >= 12 && j < 15) o = CB_CINT + 3 * (i - 6) + (j - 12);
        }
        go[r] = o;
        one[r] = (i == j && i < 15) ? 1.0 : 0.0;
        qo[r] = (i == j && i < 15) ? CB_NOISE + i / 3 : CB_ZERO;
      }
      imu_v4 Pc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + 4 * r;
        Pc[r] = (row < 15 && col < 15) ? P[15 * row + col] : 0.0;
      }
      const double* cb = lds + ImuLds::CB;
      double g[4], q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g[r] = cb[go[r]] + one[r];
        q[r] = cb[qo[r]];
      }
      for (int k = 0; k < ns; ++k) {
        double gn[4], qn[4];   // next step's coefficients, requested before this step's chain starts
        const double* cbn = cb + ImuLds::CB_STRIDE * (k + 1 < ns ? k + 1 : k);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gn[r] = cbn[go[r]];
          qn[r] = cbn[qo[r]];
        }
        asm volatile("" ::: "memory");   // (keeps the reads above the chain: hipcc otherwise sinks them to their first use)
        // two accumulator chains of two k-blocks each per product: a dependent v_mfma_f64_16x16x4 costs ~114 cycles,
        // an independent one 64
        const imu_v4 zero = {0.0, 0.0, 0.0, 0.0};
        imu_v4 Za = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[0], g[0], zero, 0, 0, 0);
        imu_v4 Zb = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[2], g[2], zero, 0, 0, 0);
        Za = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[1], g[1], Za, 0, 0, 0);
        Zb = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[3], g[3], Zb, 0, 0, 0);
        const imu_v4 Z = Za + Zb;
        const imu_v4 qv = {q[0], q[1], q[2], q[3]};
        imu_v4 Pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Z[0], g[0], qv, 0, 0, 0);
        imu_v4 Pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Z[2], g[2], zero, 0, 0, 0);
        Pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Z[1], g[1], Pa, 0, 0, 0);
        Pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Z[3], g[3], Pb, 0, 0, 0);
        Pc = Pa + Pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          g[r] = gn[r] + one[r];
          q[r] = qn[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + 4 * r;
        if (row < 15 && col < 15) P[15 * row + col] = Pc[r];
      }
    } else {
      // ---- stage 7, by the other waves while wave 0 runs the recursion (disjoint LDS regions): running totals (ordered
      //      sums, one work-item per component); the carries of the prefix arrays move from index ns to index 0
      const int u = tid - 64;
      int base_i = -1, stride = 0;
      double* tot = nullptr;
      if (u < 9) { base_i = ImuLds::CDBL + u; stride = 9; tot = t_Cdbl + u; }
      else if (u < 12) { base_i = ImuLds::ADBL + (u - 9); stride = 3; tot = t_adbl + (u - 9); }
      else if (u < 21) { base_i = ImuLds::DAL + (u - 12); stride = 9; tot = t_dal + (u - 12); }
      else if (u < 30) { base_i = ImuLds::RINV + (u - 21); stride = 9; tot = t_dp + (u - 21); }
      if (base_i >= 0) {
        double add = 0;
        for (int k = 0; k < ns; ++k) add += lds[base_i + stride * k];
        *tot += add;
      }
      if (u >= 32 && u < 36) c_Dq[u - 32] = lds[ImuLds::DQP + 4 * ns + (u - 32)];
      if (u >= 64 && u < 73) {
        const int c = u - 64;
        lds[ImuLds::CINTP + c] = lds[ImuLds::CINTP + 9 * ns + c];
        lds[ImuLds::CROSSP + c] = lds[ImuLds::CROSSP + 9 * ns + c];
        lds[ImuLds::DVP + c] = lds[ImuLds::DVP + 9 * ns + c];
        if (c < 3) lds[ImuLds::AINTP + c] = lds[ImuLds::AINTP + 3 * ns + c];
      }
    }
    __syncthreads();
    RSTAMP(7);
    RSTAMP(8);
    if (s_finished) break;
  }
  // ---- cache copy in LDS
  double* ca = lds + ImuLds::CA;
  if (tid < 9) {
    ca[CA_CI + tid] = lds[ImuLds::CINTP + tid];
    ca[CA_CD + tid] = t_Cdbl[tid];
    ca[CA_DA + tid] = t_dal[tid];
    ca[CA_DV + tid] = lds[ImuLds::DVP + tid];
    ca[CA_DP + tid] = t_dp[tid];
  }
  if (tid < 3) {
    ca[CA_AI + tid] = lds[ImuLds::AINTP + tid];
    ca[CA_AD + tid] = t_adbl[tid];
  }
  if (tid < 4) ca[CA_DQ + tid] = c_Dq[tid];
  __syncthreads();
  RSTAMP(9);
  // ---- stage 9: information = sym(P)^-1, sqrtInfo = chol(sym(information))^T  (ImuError.cpp:268-279), by ONE wave
  //      without barriers: lane j holds column j of the (full, symmetric) matrix in registers; a right-looking
  //      Cholesky step reads the pivot column through v_readlane (wave-uniform values), the lane's own entry of the
  //      pivot ROW is its L_jk by symmetry, and every lane updates the rest of its column (see chol15_wave).
  if (tid < 225) ca[CA_SI + tid] = 0.0;   // the strictly lower part of sqrtInfo
  __syncthreads();
  if ((tid >> 6) == 0) {
    const int j = tid < 15 ? tid : 14;    // (lanes 15..63 shadow lane 14 and never store)
    const bool act = tid < 15;
    double c[15], invd[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) c[i] = 0.5 * P[15 * i + j] + 0.5 * P[15 * j + i];
    chol15_wave(c, invd, T, 15, 1, tid, act);          // T[15 i + k] = L_ik
    __builtin_amdgcn_wave_barrier();
    // X = L^-1, column j by lane j:  x_i = ((i == j) - sum_{m < i} L_im x_m) / L_ii  (zero above the diagonal by itself)
    double x[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      double lrow[15];
#pragma unroll
      for (int m = 0; m < i; ++m) lrow[m] = T[15 * i + m];   // wave-uniform addresses
      double sacc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < i; ++m) sacc -= lrow[m] * x[m];
      x[i] = sacc * invd[i];
    }
    __builtin_amdgcn_wave_barrier();
    if (act) {
#pragma unroll
      for (int m = 0; m < 15; ++m) P[15 * m + j] = x[m];     // P := X (lower triangular)
    }
    __builtin_amdgcn_wave_barrier();
    // information = X^T X: column j on lane j, info_ij = sum_{m >= max(i,j)} X_mi X_mj  (x_m = 0 for m < j)
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      double xc[15];
#pragma unroll
      for (int m = i; m < 15; ++m) xc[m] = P[15 * m + i];    // wave-uniform addresses
      double sacc = 0.0;
#pragma unroll
      for (int m = i; m < 15; ++m) sacc += xc[m] * x[m];
      c[i] = sacc;
    }
    chol15_wave(c, invd, ca + CA_SI, 1, 15, tid, act);  // sqrtInfo = L^T: entry (k, t) = L_tk
  }
  __syncthreads();
  // ---- write the cache back to HBM
  ImuCacheD* cg = W.imu_cache + f;
  if (tid < 4) cg->Delta_q[tid] = ca[CA_DQ + tid];
  if (tid < 9) {
    cg->C_integral[tid] = ca[CA_CI + tid];
    cg->C_doubleintegral[tid] = ca[CA_CD + tid];
    cg->dalpha_db_g[tid] = ca[CA_DA + tid];
    cg->dv_db_g[tid] = ca[CA_DV + tid];
    cg->dp_db_g[tid] = ca[CA_DP + tid];
    cg->sb_ref[tid] = sb0[tid];
  }
  if (tid < 3) {
    cg->acc_integral[tid] = ca[CA_AI + tid];
    cg->acc_doubleintegral[tid] = ca[CA_AD + tid];
  }
  if (tid < 225) cg->sqrt_info[tid] = ca[CA_SI + tid];
  if (tid == 0) {
    cg->valid = 1;
    cg->redo_count += 1;
  }
  __syncthreads();
  RSTAMP(10);
}
#undef RSTAMP

// MODE 0: the whole factor.  MODE 1 / 2 (round 6): its two halves in two launches of one slot — 1 = what may CHANGE the record (take
// back a discarded evaluation's, the bias check, the re-preintegration; runs in small_kernel right behind the solve launch), 2 = the
// evaluation at a record that is final for this slot (rides in the decision-free Schur launch in front of the next solve launch,
// ba_schur2.hpp: the light half — 10 KB of LDS, no re-preintegration code — next to the Schur workgroups at their occupancy).
template <int MODE = 0>
__device__ void imu_factor(const WinPtrs& W, int f, int trial, double* lds, int tid, int spec_discard = 0) {
  __shared__ double s_db[6];
  // the trial states p0[7] | p1[7] | b0[9] | b1[9] and the cache's reference bias [9]
  __shared__ double s_st[32 + 9];
  __shared__ int s_valid, s_prev;
  if (W.prof && f == 0 && tid == 0 && blockIdx.y == 0) W.prof[62] = (double)clock64();   // diagnostics: length of one IMU workgroup
  double* ca = lds + EvalLds::CA;
  const double* p0 = W.pose[trial] + 7 * (size_t)W.imu_pose0[f];
  const double* p1 = W.pose[trial] + 7 * (size_t)W.imu_pose1[f];
  const double* b0 = W.sb[trial] + 9 * (size_t)W.imu_sb0[f];
  const double* b1 = W.sb[trial] + 9 * (size_t)W.imu_sb1[f];
  const double Dt = ns_to_sec(W.imu_t1[f] - W.imu_t0[f]);
  ImuCacheD* cg = W.imu_cache + f;
  ImuCacheD* cp = W.imu_cache_prev + f;
  // Everything the factor reads from HBM — the preintegration record, the four state blocks, the record's reference bias and its
  // state word — is requested in one go, every work-item its share, and waits at ONE barrier.  (Before: work-item 0 fetched the
  // bias and the reference for the bias check, barrier, then the record came in, barrier, then work-item 0 fetched the poses
  // inside its serial part: three dependent round trips to memory in every IMU workgroup.)
  auto stage = [&]() {
    if constexpr (MODE != 1) {   // (the first half only looks at the states and the record's reference bias)
      if (tid < 4) ca[CA_DQ + tid] = cg->Delta_q[tid];
      if (tid < 9) {
        ca[CA_CI + tid] = cg->C_integral[tid];
        ca[CA_CD + tid] = cg->C_doubleintegral[tid];
        ca[CA_DA + tid] = cg->dalpha_db_g[tid];
        ca[CA_DV + tid] = cg->dv_db_g[tid];
        ca[CA_DP + tid] = cg->dp_db_g[tid];
      }
      if (tid < 3) {
        ca[CA_AI + tid] = cg->acc_integral[tid];
        ca[CA_AD + tid] = cg->acc_doubleintegral[tid];
      }
      if (tid < 225) ca[CA_SI + tid] = cg->sqrt_info[tid];
    }
    const int u = tid - 192;   // (the last wave has the fewest of the loads above)
    if (u >= 0 && u < 7) s_st[u] = p0[u];
    else if (u >= 7 && u < 14) s_st[u] = p1[u - 7];
    else if (u >= 14 && u < 23) s_st[u] = b0[u - 14];
    else if (u >= 23 && u < 32) s_st[u] = b1[u - 23];
    else if (u >= 32 && u < 41) s_st[u] = cg->sb_ref[u - 32];
    else if (u == 41) s_valid = cg->valid;
    else if (u == 42) s_prev = cp->valid;
  };
  stage();
  __syncthreads();
  // ---- the record follows the reference's sequence of evaluations.  ImuError's preintegration members are `mutable`: every
  // Evaluate call may redo them and what it leaves stays, whether the step is accepted or not (ImuError.cpp:541-558).  This
  // backend makes one kind of evaluation Ceres never makes: the Gauss-Newton point, tried speculatively before its length is
  // known, when it turns out to lie outside the trust region (Ctrl::spec_discard; the explicit dogleg step replaces it).  If
  // that evaluation re-preintegrated — a Gauss-Newton point far away moves the gyro bias past the threshold — the record it
  // overwrote comes back here, before this evaluation looks at its own bias.  (Until round 5 it stayed: the explicit trial then
  // saw a reference bias further away than the threshold and integrated once more, at ITS bias, where the reference still
  // has the preintegration of the last real evaluation — costs 1e-9 ... 1e-5 apart in the middle of radius-limited runs,
  // two re-preintegrations of 104 us for none.  Found with the long double referee, tools/gpu_cost_consistency.py.)
  if constexpr (MODE != 2) {
  if (s_prev) {   // (uniform; rare: the previous evaluation of this term re-preintegrated)
    if (spec_discard) {
      const double* src = reinterpret_cast<const double*>(cp);
      double* dst = reinterpret_cast<double*>(cg);
      for (int i = tid; i < (int)(sizeof(ImuCacheD) / 8); i += IMU_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (tid == 0) cp->valid = 0;
    if (spec_discard) {
      stage();
      __syncthreads();
    }
  }
  // ---- bias check of ImuError::EvaluateWithMinimalJacobians (ImuError.cpp:541-558), by every work-item from the staged values
  // (a uniform verdict, no second barrier): re-preintegrate on first use or when |b_g - b_g,ref| * dt > 1e-4 (rarely taken)
  {
    const int v = s_valid;
    double db[3];
    for (int i = 0; i < 3; ++i) db[i] = s_st[14 + 3 + i] - s_st[32 + 3 + i];
    const double nbg = sqrt(db[0] * db[0] + db[1] * db[1] + db[2] * db[2]);
    // valid == 3: the record was built ahead of time at the bias of the uploaded window (imu_pre_kernel).  It stands for the
    // preintegration of the first evaluation (redo_ = true, ImuError.cpp:62) only if that evaluation sees the very same bias;
    // okvis_ba_set_state may have changed it since: then the term is integrated again, at the bias it is evaluated at, and the
    // early integration does not count.
    bool pre_stale = false;
    if (v == 3) {
      for (int i = 0; i < 9; ++i) pre_stale = pre_stale || !(s_st[14 + i] == s_st[32 + i]);
      if (tid == 0) {
        if (pre_stale) cg->redo_count = 0;
        else cg->valid = 1;
      }
    }
    const bool redo = (!v) || pre_stale || (nbg * Dt > 0.0001);  // ImuError.cpp:549
    // a cache inherited from a previous optimize() call (only its reference bias travels): rebuild it at that
    // reference unless the bias moved past the threshold anyway
    const bool redo_ref = !redo && v == 2;
    if (redo || redo_ref) {
      double sbx[9];
      for (int i = 0; i < 9; ++i) sbx[i] = redo_ref ? s_st[32 + i] : s_st[14 + i];
      __syncthreads();   // (every work-item has read what it needs of the staged values: the re-preintegration reuses the LDS)
      if (v == 1 || v == 2) {   // keep what is about to be overwritten: the next evaluation may have to take it back (see above)
        const double* src = reinterpret_cast<const double*>(cg);
        double* dst = reinterpret_cast<double*>(cp);
        for (int i = tid; i < (int)(sizeof(ImuCacheD) / 8); i += IMU_THREADS) dst[i] = src[i];
      }
      imu_redo(W, f, sbx, lds, tid);   // updates the HBM cache in place; ends with a barrier
      if constexpr (MODE == 1) return;
      stage();
      __syncthreads();
    }
  }
  }   // MODE != 2
  if constexpr (MODE == 1) return;
  if (W.prof && f == 0 && tid == 0 && blockIdx.y == 0) W.prof[36] = (double)clock64();
  // the cache is valid now and, if it was just redone, its reference equals sb0 so that Delta_b is
  // exactly zero (ImuError.cpp:553)
  if (tid < 6) s_db[tid] = s_st[14 + 3 + tid] - s_st[32 + 3 + tid];
  if (W.prof && f == 0 && tid == 0 && blockIdx.y == 0) W.prof[37] = (double)clock64();
  // ---- F = [F0 | F1] and the error vector (ImuError.cpp:561-601), by the first work-item of each of the four waves: every one
  // of them forms the common quantities (normalised quaternions, C0, the position / velocity differences, Delta_q corrected for
  // the bias change) with the same expressions — the same bits — and then its share of the blocks; every entry is computed by
  // the expression one work-item used to compute it with.  (One work-item alone spent 4.4 of the workgroup's 11 us here: some
  // 1300 instructions at one instruction every 4-8 cycles.)
  double* F = lds + EvalLds::FM;
  double* ev = lds + EvalLds::EV;
  for (int i = tid; i < 450; i += IMU_THREADS) F[i] = 0.0;
  __syncthreads();
  static_assert(IMU_THREADS == 256, "four waves, four shares");
  if ((tid & 63) == 0) {
    const int part = tid >> 6;
    const double *p0 = s_st, *p1 = s_st + 7, *sb0 = s_st + 14, *b1 = s_st + 23;   // (the staged states)
    double q0[4] = {p0[3], p0[4], p0[5], p0[6]}, q1[4] = {p1[3], p1[4], p1[5], p1[6]};
    qnormalize(q0);
    qnormalize(q1);
    double C0[9];
    qrot(q0, C0);
    const double g = W.imu.g;
    const double v0[3] = {sb0[0], sb0[1], sb0[2]};
    const double v1[3] = {b1[0], b1[1], b1[2]};
    double dp[3], dv[3];
    for (int c = 0; c < 3; ++c) {
      const double gw = (c == 2) ? g : 0.0;
      dp[c] = p0[c] - p1[c] + v0[c] * Dt - 0.5 * gw * Dt * Dt;
      dv[c] = v0[c] - v1[c] - gw * Dt;
    }
    const double* da = ca + CA_DA;
    double dal[3];
    for (int c = 0; c < 3; ++c) dal[c] = -(da[3 * c] * s_db[0] + da[3 * c + 1] * s_db[1] + da[3 * c + 2] * s_db[2]);
    const double hn = 0.5 * sqrt(dal[0] * dal[0] + dal[1] * dal[1] + dal[2] * dal[2]);
    const double sc = sinc(hn) * 0.5;
    const double dqa[4] = {sc * dal[0], sc * dal[1], sc * dal[2], cos(hn)};
    double Dq[4];
    qmul(dqa, ca + CA_DQ, Dq);
    double q1i[4];
    qinv(q1, q1i);
    auto Fset = [&](int r, int cidx, const double* M, double sgn) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F[30 * (r + i) + cidx + j] = sgn * M[3 * i + j];
    };
    double C0T[9], t9[9], cx[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) C0T[3 * i + j] = C0[3 * j + i];
    if (part == 0) {
      // d e_q / d q1 (columns 18..20 of rows 3..5)
      double A[16], B[16], Cq[16], AB[16];
      qplus44(Dq, A);
      qoplus44(q0, B);
      qplus44(q1i, Cq);
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          double s = 0;
          for (int m = 0; m < 4; ++m) s += A[4 * i + m] * B[4 * m + j];
          AB[4 * i + j] = s;
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int m = 0; m < 4; ++m) s += AB[4 * i + m] * Cq[4 * m + j];
          t9[3 * i + j] = s;
        }
      Fset(3, 18, t9, -1.0);
    } else if (part == 1) {
      // d e_q / d b_g (rows 3..5, columns 9..11) and d e_p / d q0 (rows 0..2, columns 3..5)
      double A[16], B[16], qb[4], M3[9];
      qmul(q1i, q0, qb);
      qoplus44(qb, A);
      qoplus44(Dq, B);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int m = 0; m < 4; ++m) s += A[4 * i + m] * B[4 * m + j];
          M3[3 * i + j] = s;
        }
      mat3_mul(M3, da, t9);
      Fset(3, 9, t9, -1.0);
      cross_mx(dp, cx);
      mat3_mul(C0T, cx, t9);
      Fset(0, 3, t9, 1.0);
    } else if (part == 2) {
      // d e_q / d q0 (rows 3..5, columns 3..5), d e_v / d q0 (rows 6..8, columns 3..5) and the copies of C0^T
      double qa[4], A[16], B[16];
      qmul(Dq, q1i, qa);
      qplus44(qa, A);
      qoplus44(q0, B);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int m = 0; m < 4; ++m) s += A[4 * i + m] * B[4 * m + j];
          t9[3 * i + j] = s;
        }
      Fset(3, 3, t9, 1.0);
      cross_mx(dv, cx);
      mat3_mul(C0T, cx, t9);
      Fset(6, 3, t9, 1.0);
      Fset(0, 0, C0T, 1.0);
      for (int c = 0; c < 9; ++c) t9[c] = C0T[c] * Dt;
      Fset(0, 6, t9, 1.0);
      Fset(6, 6, C0T, 1.0);
      Fset(0, 15, C0T, -1.0);
      Fset(6, 21, C0T, -1.0);
    } else {
      // the blocks that are copies of the preintegration's Jacobians, the identity parts that no block covers (rows 9..14)
      // and the error vector (:597-601)
      Fset(0, 9, ca + CA_DP, 1.0);
      Fset(0, 12, ca + CA_CD, -1.0);
      Fset(6, 9, ca + CA_DV, 1.0);
      Fset(6, 12, ca + CA_CI, -1.0);
      for (int i = 9; i < 15; ++i) {
        F[30 * i + i] = 1.0;
        F[30 * i + 15 + i] = -1.0;
      }
      double e0[3], e2[3];
      mat3_vec(C0T, dp, e0);
      mat3_vec(C0T, dv, e2);
      for (int i = 0; i < 3; ++i) {
        double s0 = e0[i] + ca[CA_AD + i], s2 = e2[i] + ca[CA_AI + i];
        for (int m = 0; m < 6; ++m) {
          // (F[30 i + 9 + m] and F[30 (6 + i) + 9 + m]: the copies this work-item has just written, read from where they came from)
          const double f0 = m < 3 ? 1.0 * ca[CA_DP + 3 * i + m] : -1.0 * ca[CA_CD + 3 * i + (m - 3)];
          const double f2 = m < 3 ? 1.0 * ca[CA_DV + 3 * i + m] : -1.0 * ca[CA_CI + 3 * i + (m - 3)];
          s0 += f0 * s_db[m];
          s2 += f2 * s_db[m];
        }
        ev[i] = s0;
        ev[6 + i] = s2;
      }
      double qe[4], qt[4];
      qmul(q1i, q0, qt);
      qmul(Dq, qt, qe);
      ev[3] = 2 * qe[0];
      ev[4] = 2 * qe[1];
      ev[5] = 2 * qe[2];
      for (int i = 0; i < 6; ++i) ev[9 + i] = sb0[3 + i] - b1[3 + i];
    }
  }
  __syncthreads();
  if (W.prof && f == 0 && tid == 0 && blockIdx.y == 0) W.prof[39] = (double)clock64();
  // ---- J = sqrtInfo (upper) * F (kept in LDS), r = sqrtInfo * e, then H = J^T J and g = J^T r
  double* out = W.imu_lin[trial] + (size_t)f * IMU_LIN_STRIDE;
  const double* SI = ca + CA_SI;
  double* Jl = lds + EvalLds::PM;
  for (int wi = tid; wi < 450; wi += IMU_THREADS) {
    const int i = wi / 30, j = wi - 30 * i;
    double s = 0;
    for (int m = i; m < 15; ++m) s += SI[15 * i + m] * F[30 * m + j];
    Jl[wi] = s;
  }
  __shared__ double s_r[16];
  if (tid < 15) {
    double s = 0;
    for (int m = tid; m < 15; ++m) s += SI[15 * tid + m] * ev[m];
    out[IMU_R + tid] = s;
    s_r[tid] = s;
  }
  __syncthreads();
  if (W.prof && f == 0 && tid == 0 && blockIdx.y == 0) W.prof[59] = (double)clock64();
  // (an entry's place in the record: host-built, W.imu_pos — the order in which the solve kernel adds the record to its system)
  const BA_G int* pos = W.imu_pos + (size_t)f * IMU_LIN_STRIDE;
  for (int wi = tid; wi < 465 + 30; wi += IMU_THREADS) {
    const int at_rec = pos[wi];
    if (wi < 465) {
      int a = (int)((sqrtf(8.0f * wi + 1.0f) - 1.0f) * 0.5f);
      while ((a + 1) * (a + 2) / 2 <= wi) ++a;
      while (a * (a + 1) / 2 > wi) --a;
      const int b = wi - a * (a + 1) / 2;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s += Jl[30 * k + a] * Jl[30 * k + b];
      out[at_rec] = s;
    } else {
      const int a = wi - 465;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s += Jl[30 * k + a] * s_r[k];
      out[at_rec] = s;
    }
  }
  if (tid == 0) {
    double s = 0;
    for (int i = 0; i < 15; ++i) s += s_r[i] * s_r[i];
    out[IMU_COST] = 0.5 * s;
    if (W.prof && f == 0 && blockIdx.y == 0) W.prof[63] = (double)clock64();
  }
}

// all PoseError / SpeedAndBiasError / RelativePoseError / MarginalizationError terms of one window
__device__ void small_factors(const WinPtrs& W, int trial, double* lds, int tid) {
  __shared__ double s_cost[IMU_THREADS / 64];
  double cost = 0;
  const int np = W.n_pprior, nsb = W.n_sbprior, nr = W.n_rel;
  for (int f = tid; f < np; f += IMU_THREADS) {
    // PoseError.cpp:91-118
    const double* x = W.pose[trial] + 7 * (size_t)W.pprior_pose[f];
    const double* m = W.pprior_meas + 7 * (size_t)f;
    const double* SI = W.pprior_sqrtinfo + 36 * (size_t)f;
    double q[4] = {x[3], x[4], x[5], x[6]}, qm[4] = {m[3], m[4], m[5], m[6]};
    // (the *_strict forms: a pose at its prior gives dq.xyz = 0 exactly, as the reference's arithmetic does — ba_math.hpp)
    qnormalize_strict(q);
    qnormalize_strict(qm);
    double qi[4], dq[4];
    qinv_strict(q, qi);
    qnormalize_strict(qi);
    qmul_strict(qm, qi, dq);
    qnormalize_strict(dq);
    const double e[6] = {m[0] - x[0], m[1] - x[1], m[2] - x[2], 2 * dq[0], 2 * dq[1], 2 * dq[2]};
    double J0[36];
    for (int i = 0; i < 36; ++i) J0[i] = 0;
    J0[0] = J0[7] = J0[14] = -1.0;
    double P3[9];
    qplus33(dq, P3);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J0[6 * (3 + i) + 3 + j] = -P3[3 * i + j];
    double* out = W.pp_lin[trial] + 42 * (size_t)f;
    for (int i = 0; i < 6; ++i) {
      double r = 0;
      for (int k = 0; k < 6; ++k) r += SI[6 * i + k] * e[k];
      out[36 + i] = r;
      cost += 0.5 * r * r;
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += SI[6 * i + k] * J0[6 * k + j];
        out[6 * i + j] = s;
      }
    }
  }
  for (int f = tid; f < nsb; f += IMU_THREADS) {
    // SpeedAndBiasError.cpp:93-101
    const double* x = W.sb[trial] + 9 * (size_t)W.sbprior_sb[f];
    const double* m = W.sbprior_meas + 9 * (size_t)f;
    const double* SI = W.sbprior_sqrtinfo + 81 * (size_t)f;
    double* out = W.sbp_lin[trial] + 9 * (size_t)f;
    for (int i = 0; i < 9; ++i) {
      double r = 0;
      for (int k = 0; k < 9; ++k) r += SI[9 * i + k] * (m[k] - x[k]);
      out[i] = r;
      cost += 0.5 * r * r;
    }
  }
  for (int f = tid; f < nr; f += IMU_THREADS) {
    // RelativePoseError.cpp:88-159
    const double* x0 = W.pose[trial] + 7 * (size_t)W.rel_pose0[f];
    const double* x1 = W.pose[trial] + 7 * (size_t)W.rel_pose1[f];
    const double* SI = W.rel_sqrtinfo + 36 * (size_t)f;
    double q0[4] = {x0[3], x0[4], x0[5], x0[6]}, q1[4] = {x1[3], x1[4], x1[5], x1[6]};
    qnormalize_strict(q0);
    qnormalize_strict(q1);
    double qi[4], dq[4];
    qinv_strict(q0, qi);
    qnormalize_strict(qi);
    qmul_strict(q1, qi, dq);
    qnormalize_strict(dq);
    const double e[6] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2], 2 * dq[0], 2 * dq[1], 2 * dq[2]};
    double P3[9], O3[9];
    qplus33(dq, P3);
    qoplus33(dq, O3);
    double J[72];  // 6 x 12 = [J0 | J1] before weighting
    for (int i = 0; i < 72; ++i) J[i] = 0;
    for (int i = 0; i < 3; ++i) {
      J[12 * i + i] = -1.0;
      J[12 * i + 6 + i] = 1.0;
      for (int j = 0; j < 3; ++j) {
        J[12 * (3 + i) + 3 + j] = -P3[3 * i + j];
        J[12 * (3 + i) + 9 + j] = O3[3 * i + j];
      }
    }
    double* out = W.rel_lin[trial] + 78 * (size_t)f;
    for (int i = 0; i < 6; ++i) {
      double r = 0;
      for (int k = 0; k < 6; ++k) r += SI[6 * i + k] * e[k];
      out[72 + i] = r;
      cost += 0.5 * r * r;
      for (int j = 0; j < 12; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += SI[6 * i + k] * J[12 * k + j];
        out[12 * i + j] = s;
      }
    }
  }
  // ---- marginalisation prior: e = e0 + J dchi, J^T e, rotation blocks (MarginalizationError.cpp:867-946)
  if (W.marg_dim > 0) {
    const int Dm = W.marg_dim, nb = W.marg_nb;
    double* dchi = lds;          // Dm
    double* ee = lds + Dm;       // Dm
    for (int i = tid; i < Dm; i += IMU_THREADS) dchi[i] = 0.0;
    __syncthreads();
    for (int b = tid; b < nb; b += IMU_THREADS) {
      const int idx = W.marg_block_idx[b], o = W.marg_block_off[b];
      const double* xl = W.marg_lin + 9 * (size_t)b;
      double* M = W.marg_lin_M[trial] + 9 * (size_t)b;
      if (W.marg_block_type[b] == 0) {
        const double* x = W.pose[trial] + 7 * (size_t)idx;
        if (W.pose_off[idx] >= 0) {
          double d[6];
          pose_ominus(xl, x, d);
          for (int k = 0; k < 6; ++k) dchi[o + k] = d[k];
          const double qli[4] = {-xl[3], -xl[4], -xl[5], xl[6]};
          double qd[4];
          qmul(x + 3, qli, qd);
          qoplus33(qd, M);
        } else {
          for (int k = 0; k < 9; ++k) M[k] = (k % 4 == 0) ? 1.0 : 0.0;
        }
      } else {
        const double* x = W.sb[trial] + 9 * (size_t)idx;
        if (W.sb_off[idx] >= 0)
          for (int k = 0; k < 9; ++k) dchi[o + k] = x[k] - xl[k];
      }
    }
    __syncthreads();
    for (int r = tid; r < Dm; r += IMU_THREADS) {
      double s = W.marg_e0[r];
      const double* Jr = W.marg_J + (size_t)r * Dm;
      for (int cidx = 0; cidx < Dm; ++cidx) s += Jr[cidx] * dchi[cidx];
      ee[r] = s;
      W.marg_lin_e[trial][r] = s;
      cost += 0.5 * s * s;
    }
    __syncthreads();
    for (int cidx = tid; cidx < Dm; cidx += IMU_THREADS) {
      double s = 0;
      for (int r = 0; r < Dm; ++r) s += W.marg_J[(size_t)r * Dm + cidx] * ee[r];
      W.marg_lin_e[trial][Dm + cidx] = s;
    }
  }
  cost = wave_sum(cost);
  if ((tid & 63) == 0) s_cost[tid >> 6] = cost;
  __syncthreads();
  if (tid == 0) {
    double s = 0;
    for (int i = 0; i < IMU_THREADS / 64; ++i) s += s_cost[i];
    W.small_cost[trial][0] = s;
  }
}

// Body of one "small factor" workgroup: bx < n_imu -> ImuError bx, bx == n_imu -> all priors + marginalisation
// prior.  Runs inside the linearise launch (ba_linearize.hpp): the IMU / prior factors and the reprojection
// factors both depend only on the trial state of the solve kernel, so they share one launch and a slow
// re-preintegration overlaps with the (wide) reprojection work instead of holding a kernel boundary.
template <int MODE = 0>   // (imu_factor's; the priors are evaluated with the second half)
__device__ __forceinline__ void small_body(const WinPtrs& W, int init, int bx, double* smem) {
  if (bx > W.n_imu) return;
  const Ctrl* ctrl = W.ctrl;
  if (ctrl->done) return;
  if (!init && !ctrl->pending) return;
  const int trial = 1 - ctrl->acc;
  if (bx < W.n_imu)
    imu_factor<MODE>(W, bx, trial, smem, threadIdx.x, init ? 0 : ctrl->spec_discard);
  else if (MODE != 1)
    small_factors(W, trial, smem, threadIdx.x);
}

}  // namespace ba
