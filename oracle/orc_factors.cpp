// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
// CPU restatement of the okvis_ceres error terms; citations relative to /root/reference.
#include "orc_factors.hpp"

#include <limits>

namespace orc {

// ===================================================================================================
// Distortion models
// ===================================================================================================
bool distort(const Camera& c, const real u[2], real out[2], real J[4]) {
  const real u0 = u[0], u1 = u[1];
  switch (c.model) {
    case DIST_NONE: {  // NoDistortion: identity
      out[0] = u0;
      out[1] = u1;
      if (J) {
        J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1;
      }
      return true;
    }
    case DIST_RADTAN: {
      // okvis_cv/include/okvis/cameras/implementation/RadialTangentialDistortion.hpp:105-151
      const real k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3];
      const real mx_u = u0 * u0, my_u = u1 * u1, mxy_u = u0 * u1;
      const real rho_u = mx_u + my_u;
      const real rad_dist_u = k1 * rho_u + k2 * rho_u * rho_u;
      out[0] = u0 + u0 * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho_u + 2.0 * mx_u);
      out[1] = u1 + u1 * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho_u + 2.0 * my_u);
      if (J) {
        J[0] = 1 + rad_dist_u + k1 * 2.0 * mx_u + k2 * rho_u * 4 * mx_u + 2.0 * p1 * u1 + 6 * p2 * u0;
        J[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho_u * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
        J[1] = J[2];
        J[3] = 1 + rad_dist_u + k1 * 2.0 * my_u + k2 * rho_u * 4 * my_u + 6 * p1 * u1 + 2.0 * p2 * u0;
      }
      return true;
    }
    case DIST_EQUI: {
      // okvis_cv/include/okvis/cameras/implementation/EquidistantDistortion.hpp:105-206
      const real k1 = c.d[0], k2 = c.d[1], k3 = c.d[2], k4 = c.d[3];
      const real r = std::sqrt(u0 * u0 + u1 * u1);
      const real theta = std::atan(r);
      const real theta2 = theta * theta, theta4 = theta2 * theta2;
      const real theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const real thetad = theta * (1 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
      const real scaling = (r > 1e-8) ? thetad / r : 1.0;
      out[0] = scaling * u0;
      out[1] = scaling * u1;
      if (J) {
        if (r > 1e-8) {
          real t2, t3, t4, t6, t7, t8, t9, t11, t17, t18, t19, t20, t25;
          t2 = u0 * u0;
          t3 = u1 * u1;
          t4 = t2 + t3;
          t6 = std::atan(std::sqrt(t4));
          t7 = t6 * t6;
          t8 = 1.0 / std::sqrt(t4);
          t9 = t7 * t7;
          t11 = 1.0 / ((t2 + t3) + 1.0);
          t17 = (((k1 * t7 + k2 * t9) + k3 * t7 * t9) + k4 * (t9 * t9)) + 1.0;
          t18 = 1.0 / t4;
          t19 = 1.0 / std::sqrt(t4 * t4 * t4);
          t20 = t6 * t8 * t17;
          t25 = ((k2 * t6 * t7 * t8 * t11 * u1 * 4.0 + k3 * t6 * t8 * t9 * t11 * u1 * 6.0) +
                 k4 * t6 * t7 * t8 * t9 * t11 * u1 * 8.0) +
                k1 * t6 * t8 * t11 * u1 * 2.0;
          t4 = ((k2 * t6 * t7 * t8 * t11 * u0 * 4.0 + k3 * t6 * t8 * t9 * t11 * u0 * 6.0) +
                k4 * t6 * t7 * t8 * t9 * t11 * u0 * 8.0) +
               k1 * t6 * t8 * t11 * u0 * 2.0;
          t7 = t11 * t17 * t18 * u0 * u1;
          J[1] = (t7 + t6 * t8 * t25 * u0) - t6 * t17 * t19 * u0 * u1;                        // J(0,1)
          J[3] = ((t20 - t3 * t6 * t17 * t19) + t3 * t11 * t17 * t18) + t6 * t8 * t25 * u1;   // J(1,1)
          J[0] = ((t20 - t2 * t6 * t17 * t19) + t2 * t11 * t17 * t18) + t6 * t8 * t4 * u0;    // J(0,0)
          J[2] = (t7 + t6 * t8 * t4 * u1) - t6 * t17 * t19 * u0 * u1;                         // J(1,0)
        } else {
          J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1;
        }
      }
      return true;
    }
    case DIST_RADTAN8: {
      // okvis_cv/include/okvis/cameras/implementation/RadialTangentialDistortion8.hpp:125-150
      const real k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3];
      const real k3 = c.d[4], k4 = c.d[5], k5 = c.d[6], k6 = c.d[7];
      const real mx_u = u0 * u0, my_u = u1 * u1, mxy_u = u0 * u1;
      const real rho_u = mx_u + my_u;
      if (rho_u > 9.0) return false;  // reference returns false with outputs unset
      const real cc = rho_u * (k4 + rho_u * (k5 + k6 * rho_u)) + 1.0;
      const real c2 = cc * cc;
      const real rad_dist_u = (1.0 + ((k3 * rho_u + k2) * rho_u + k1) * rho_u) /
                                (1.0 + ((k6 * rho_u + k5) * rho_u + k4) * rho_u);
      out[0] = u0 * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho_u + 2.0 * mx_u);
      out[1] = u1 * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho_u + 2.0 * my_u);
      if (J) {
        const real num = rho_u * (k1 + rho_u * (k2 + k3 * rho_u)) + 1.0;
        const real den = rho_u * (k4 + rho_u * (k5 + k6 * rho_u)) + 1.0;
        // d(num)/du_a and d(den)/du_a as written (expanded) in the reference
        const real dnum0 = rho_u * (u0 * (k2 + k3 * rho_u) * 2.0 + k3 * u0 * rho_u * 2.0) +
                             u0 * (k1 + rho_u * (k2 + k3 * rho_u)) * 2.0;
        const real dnum1 = rho_u * (u1 * (k2 + k3 * rho_u) * 2.0 + k3 * u1 * rho_u * 2.0) +
                             u1 * (k1 + rho_u * (k2 + k3 * rho_u)) * 2.0;
        const real dden0 = rho_u * (u0 * (k5 + k6 * rho_u) * 2.0 + k6 * u0 * rho_u * 2.0) +
                             u0 * (k4 + rho_u * (k5 + k6 * rho_u)) * 2.0;
        const real dden1 = rho_u * (u1 * (k5 + k6 * rho_u) * 2.0 + k6 * u1 * rho_u * 2.0) +
                             u1 * (k4 + rho_u * (k5 + k6 * rho_u)) * 2.0;
        J[0] = p1 * u1 * 2.0 + p2 * u0 * 6.0 + num / den + (u0 * dnum0) / den - u0 * dden0 * num * 1.0 / c2;
        J[1] = p1 * u0 * 2.0 + p2 * u1 * 2.0 + (u0 * dnum1) / den - u0 * dden1 * num * 1.0 / c2;
        J[2] = p1 * u0 * 2.0 + p2 * u1 * 2.0 + (u1 * dnum0) / den - u1 * dden0 * num * 1.0 / c2;
        J[3] = p1 * u1 * 6.0 + p2 * u0 * 2.0 + num / den + (u1 * dnum1) / den - u1 * dden1 * num * 1.0 / c2;
      }
      return true;
    }
  }
  return false;
}

// PinholeCamera<D>::project with point Jacobian (implementation/PinholeCamera.hpp:148-226)
bool project(const Camera& c, const V3& point, real kp[2], Mat<2, 3>* Jout) {
  if (std::fabs(point[2]) < 1.0e-12) return false;  // :155-157 ProjectionStatus::Invalid, outputs unset
  const real rz = 1.0 / point[2];
  const real rz2 = rz * rz;
  real u[2] = {point[0] * rz, point[1] * rz};
  real d[2], Jd[4];
  if (!distort(c, u, d, Jd)) return false;
  if (Jout) {
    Mat<2, 3>& J = *Jout;  // :196-206
    J(0, 0) = c.fu * Jd[0] * rz;
    J(0, 1) = c.fu * Jd[1] * rz;
    J(0, 2) = -c.fu * (point[0] * Jd[0] + point[1] * Jd[1]) * rz2;
    J(1, 0) = c.fv * Jd[2] * rz;
    J(1, 1) = c.fv * Jd[3] * rz;
    J(1, 2) = -c.fv * (point[0] * Jd[2] + point[1] * Jd[3]) * rz2;
  }
  kp[0] = c.fu * d[0] + c.cu;  // :209-210
  kp[1] = c.fv * d[1] + c.cv;
  return true;
}

// PinholeCamera<D>::projectHomogeneous (implementation/PinholeCamera.hpp:357-378)
bool projectHomogeneous(const Camera& c, const V4& hp, real kp[2], Mat<2, 4>* J) {
  V3 head = vec3(hp[0], hp[1], hp[2]);
  Mat<2, 3> J3;
  bool ok;
  if (hp[3] < 0) {
    ok = project(c, -head, kp, J ? &J3 : nullptr);  // Jacobian sign NOT flipped (quirk d)
  } else {
    ok = project(c, head, kp, J ? &J3 : nullptr);
  }
  if (!ok) return false;
  if (J) {
    for (int i = 0; i < 2; ++i) {
      for (int j = 0; j < 3; ++j) (*J)(i, j) = J3(i, j);
      (*J)(i, 3) = 0.0;
    }
  }
  return true;
}

// ===================================================================================================
// PoseLocalParameterization
// ===================================================================================================
void pose_plus(const real x[7], const real delta[6], real out[7]) {
  Transformation T = Transformation::fromParams(x);  // PoseLocalParameterization.cpp:66-69
  T.oplus(delta);                                    // :72 -> Transformation::oplus
  T.toParams(out);
}
void pose_minus(const real x[7], const real xp[7], real delta[6]) {
  delta[0] = xp[0] - x[0];
  delta[1] = xp[1] - x[1];
  delta[2] = xp[2] - x[2];
  Quat qp{xp[3], xp[4], xp[5], xp[6]};
  Quat q{x[3], x[4], x[5], x[6]};
  Quat d = qmul(qp, qinv(q));  // :112-115
  delta[3] = 2 * d.x;
  delta[4] = 2 * d.y;
  delta[5] = 2 * d.z;
}
void pose_lift_jacobian(const real x[7], real J[42]) {
  // PoseLocalParameterization.cpp:131-145
  for (int i = 0; i < 42; ++i) J[i] = 0;
  J[0 * 7 + 0] = 1;
  J[1 * 7 + 1] = 1;
  J[2 * 7 + 2] = 1;
  Quat q_inv{-x[3], -x[4], -x[5], x[6]};
  M4 Qp = qoplusMat(q_inv);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) J[(3 + i) * 7 + 3 + j] = 2.0 * Qp(i, j);
}
void pose_plus_jacobian(const real x[7], real J[42]) {
  // Transformation::oplusJacobian (implementation/Transformation.hpp:273-286): 7x6
  for (int i = 0; i < 42; ++i) J[i] = 0;
  J[0 * 6 + 0] = 1;
  J[1 * 6 + 1] = 1;
  J[2 * 6 + 2] = 1;
  Transformation T = Transformation::fromParams(x);
  M4 Qo = qoplusMat(T.q);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) J[(3 + i) * 6 + 3 + j] = 0.5 * Qo(i, j);
}

// ===================================================================================================
// ReprojectionError
// ===================================================================================================
void reprojection_error(const real pose[7], const real point[4], const real extr[7],
                        const Camera& cam, const real uv[2], const real sqrtInfo[4], bool jac,
                        ReprojOut* out) {
  // implementation/ReprojectionError.hpp:95-121
  const V3 t_WS_W = vec3(pose[0], pose[1], pose[2]);
  const Quat q_WS{pose[3], pose[4], pose[5], pose[6]};
  V4 hp_W;
  for (int i = 0; i < 4; ++i) hp_W[i] = point[i];
  const V3 t_SC_S = vec3(extr[0], extr[1], extr[2]);
  const Quat q_SC{extr[3], extr[4], extr[5], extr[6]};

  const M3 C_SC = qrot(q_SC);
  const M3 C_CS = C_SC.t();
  M4 T_CS = M4::Identity();
  T_CS.setBlock(0, 0, C_CS);
  T_CS.setBlock(0, 3, -1.0 * (C_CS * t_SC_S));
  const M3 C_WS = qrot(q_WS);
  const M3 C_SW = C_WS.t();
  M4 T_SW = M4::Identity();
  T_SW.setBlock(0, 0, C_SW);
  T_SW.setBlock(0, 3, -1.0 * (C_SW * t_WS_W));
  const V4 hp_S = T_SW * hp_W;
  const V4 hp_C = T_CS * hp_S;

  // :124-141
  real kp[2];
  Mat<2, 4> Jh;
  Mat<2, 2> sqrtI;
  for (int i = 0; i < 4; ++i) sqrtI[i] = sqrtInfo[i];
  out->defined = projectHomogeneous(cam, hp_C, kp, jac ? &Jh : nullptr);
  out->Jp = Mat<2, 6>::Zero();
  out->Jl = Mat<2, 3>::Zero();
  out->Je = Mat<2, 6>::Zero();
  if (!out->defined) {
    // Reference: outputs of project() unset and status ignored (quirk e) -> undefined behaviour.
    // Oracle definition: zero residual, zero Jacobians.
    out->r[0] = out->r[1] = 0.0;
    out->valid = false;
    return;
  }
  Mat<2, 1> error;
  error[0] = uv[0] - kp[0];
  error[1] = uv[1] - kp[1];
  Mat<2, 1> werr = sqrtI * error;
  out->r[0] = werr[0];
  out->r[1] = werr[1];

  // :143-151
  bool valid = true;
  if (std::fabs(hp_C[3]) > 1.0e-8) {
    if (hp_C[2] / hp_C[3] < 0.2) valid = false;
  }
  out->valid = valid;
  if (!jac) return;
  Mat<2, 4> Jh_weighted = sqrtI * Jh;

  if (valid) {
    {  // :156-167
      V3 p = vec3(hp_W[0], hp_W[1], hp_W[2]) - hp_W[3] * t_WS_W;
      Mat<4, 6> J = Mat<4, 6>::Zero();
      J.setBlock(0, 0, hp_W[3] * C_SW);
      J.setBlock(0, 3, -1.0 * (C_SW * crossMx(p)));
      out->Jp = Jh_weighted * T_CS * J;
    }
    {  // :188-206
      M4 T_CW = T_CS * T_SW;
      Mat<2, 4> J1 = -1.0 * (Jh_weighted * T_CW);
      out->Jl = J1.block<2, 3>(0, 0);
    }
    {  // :208-219
      V3 p = vec3(hp_S[0], hp_S[1], hp_S[2]) - hp_S[3] * t_SC_S;
      Mat<4, 6> J = Mat<4, 6>::Zero();
      J.setBlock(0, 0, hp_S[3] * C_CS);
      J.setBlock(0, 3, -1.0 * (C_CS * crossMx(p)));
      out->Je = Jh_weighted * J;
    }
  }
}

// ===================================================================================================
// ImuError
// ===================================================================================================
namespace {
struct PreintState {
  Quat Delta_q;
  M3 C_integral, C_doubleintegral;
  V3 acc_integral, acc_doubleintegral;
  M3 cross, dalpha_db_g, dv_db_g, dp_db_g;
  Mat<15, 15> P_delta;
  real Delta_t;
};

// The integration loop shared (textually duplicated in the reference) by redoPreintegration
// (ImuError.cpp:113-261) and propagation (:327-468).  `staticVariant` selects the two places where the
// copies differ: dalpha_db_g (quirk b, :200 vs :412) and sigma2_v (quirk c, :234 vs :438).
int integrate(const ImuSamples& s, const ImuParams& prm, int64_t t_start, int64_t t_end,
              const real sb[9], bool staticVariant, bool withCov, PreintState* st) {
  int64_t time = t_start;
  const int64_t end = t_end;
  st->Delta_q = Quat{0, 0, 0, 1};
  st->C_integral = M3::Zero();
  st->C_doubleintegral = M3::Zero();
  st->acc_integral = vec3(0, 0, 0);
  st->acc_doubleintegral = vec3(0, 0, 0);
  st->cross = M3::Zero();
  st->dalpha_db_g = M3::Zero();
  st->dv_db_g = M3::Zero();
  st->dp_db_g = M3::Zero();
  st->P_delta = Mat<15, 15>::Zero();
  st->Delta_t = 0;
  bool hasStarted = false;
  int i = 0;
  const V3 bg = vec3(sb[3], sb[4], sb[5]);
  const V3 ba = vec3(sb[6], sb[7], sb[8]);
  for (int it = 0; it < s.n; ++it) {
    V3 omega_S_0 = vec3(s.gyr[3 * it], s.gyr[3 * it + 1], s.gyr[3 * it + 2]);
    V3 acc_S_0 = vec3(s.acc[3 * it], s.acc[3 * it + 1], s.acc[3 * it + 2]);
    // the reference dereferences (it+1) before the end check (quirk f); with the precondition
    // back().timeStamp >= t_end the loop breaks before it == n-1, so the value is never used.
    const int nx = (it + 1 < s.n) ? it + 1 : it;
    V3 omega_S_1 = vec3(s.gyr[3 * nx], s.gyr[3 * nx + 1], s.gyr[3 * nx + 2]);
    V3 acc_S_1 = vec3(s.acc[3 * nx], s.acc[3 * nx + 1], s.acc[3 * nx + 2]);

    int64_t nexttime = (it + 1 == s.n) ? t_end : s.t[it + 1];
    real dt = nsToSec(nexttime - time);
    if (end < nexttime) {
      real interval = nsToSec(nexttime - s.t[it]);
      nexttime = t_end;
      dt = nsToSec(nexttime - time);
      const real r = dt / interval;
      omega_S_1 = (1.0 - r) * omega_S_0 + r * omega_S_1;
      acc_S_1 = (1.0 - r) * acc_S_0 + r * acc_S_1;
    }
    if (dt <= 0.0) continue;
    st->Delta_t += dt;
    if (!hasStarted) {
      hasStarted = true;
      const real r = dt / nsToSec(nexttime - s.t[it]);
      omega_S_0 = r * omega_S_0 + (1.0 - r) * omega_S_1;
      acc_S_0 = r * acc_S_0 + (1.0 - r) * acc_S_1;
    }
    // saturation (:153-173)
    real sigma_g_c = prm.sigma_g_c;
    real sigma_a_c = prm.sigma_a_c;
    bool gsat = false, asat = false;
    for (int k = 0; k < 3; ++k) {
      if (std::fabs(omega_S_0[k]) > prm.g_max || std::fabs(omega_S_1[k]) > prm.g_max) gsat = true;
      if (std::fabs(acc_S_0[k]) > prm.a_max || std::fabs(acc_S_1[k]) > prm.a_max) asat = true;
    }
    if (gsat) sigma_g_c *= 100;
    if (asat) sigma_a_c *= 100;

    // orientation (:177-185)
    const V3 omega_S_true = 0.5 * (omega_S_0 + omega_S_1) - bg;
    const real theta_half = omega_S_true.norm() * 0.5 * dt;
    const real sinc_theta_half = sinc(theta_half);
    const real cos_theta_half = std::cos(theta_half);
    Quat dq;
    dq.x = sinc_theta_half * omega_S_true[0] * 0.5 * dt;
    dq.y = sinc_theta_half * omega_S_true[1] * 0.5 * dt;
    dq.z = sinc_theta_half * omega_S_true[2] * 0.5 * dt;
    dq.w = cos_theta_half;
    const Quat Delta_q_1 = qmul(st->Delta_q, dq);
    // rotation matrix integrals (:186-197)
    const M3 C = qrot(st->Delta_q);
    const M3 C_1 = qrot(Delta_q_1);
    const V3 acc_S_true = 0.5 * (acc_S_0 + acc_S_1) - ba;
    const M3 CC = C + C_1;
    const M3 C_integral_1 = st->C_integral + (0.5 * CC) * dt;
    const V3 acc_integral_1 = st->acc_integral + (0.5 * CC) * acc_S_true * dt;
    st->C_doubleintegral = st->C_doubleintegral + st->C_integral * dt + (0.25 * CC) * dt * dt;
    st->acc_doubleintegral =
        st->acc_doubleintegral + st->acc_integral * dt + (0.25 * CC) * acc_S_true * dt * dt;

    // Jacobian parts (:199-207 / :411-417)
    const M3 Jr = rightJacobian(omega_S_true * dt);
    if (staticVariant)
      st->dalpha_db_g = st->dalpha_db_g + dt * C_1;
    else
      st->dalpha_db_g = st->dalpha_db_g + (C_1 * Jr) * dt;
    const M3 cross_1 = qrot(qinv(dq)) * st->cross + Jr * dt;
    const M3 acc_S_x = crossMx(acc_S_true);
    const M3 G = C * acc_S_x * st->cross + C_1 * acc_S_x * cross_1;
    const M3 dv_db_g_1 = st->dv_db_g + (0.5 * dt) * G;
    const M3 dp_old_term = dt * st->dv_db_g + (0.25 * dt * dt) * G;
    st->dp_db_g = st->dp_db_g + dp_old_term;

    if (withCov) {  // covariance propagation (:209-249)
      Mat<15, 15> F = Mat<15, 15>::Identity();
      F.setBlock(0, 3, -1.0 * crossMx(st->acc_integral * dt + (0.25 * CC) * acc_S_true * dt * dt));
      F.setBlock(0, 6, dt * M3::Identity());
      F.setBlock(0, 9, dp_old_term);
      F.setBlock(0, 12, (-1.0 * st->C_integral) * dt + (0.25 * CC) * dt * dt);
      F.setBlock(3, 9, (-dt) * C_1);
      F.setBlock(6, 3, -1.0 * crossMx((0.5 * CC) * acc_S_true * dt));
      F.setBlock(6, 9, (0.5 * dt) * G);
      F.setBlock(6, 12, (-0.5 * CC) * dt);
      st->P_delta = F * st->P_delta * F.t();
      const real sigma2_dalpha = dt * sigma_g_c * sigma_g_c;
      const real sigma2_v = staticVariant ? dt * sigma_a_c * prm.sigma_a_c : dt * sigma_a_c * sigma_a_c;
      const real sigma2_p = 0.5 * dt * dt * sigma2_v;
      const real sigma2_b_g = dt * prm.sigma_gw_c * prm.sigma_gw_c;
      const real sigma2_b_a = dt * prm.sigma_aw_c * prm.sigma_aw_c;
      for (int k = 0; k < 3; ++k) {
        st->P_delta(3 + k, 3 + k) += sigma2_dalpha;
        st->P_delta(6 + k, 6 + k) += sigma2_v;
        st->P_delta(0 + k, 0 + k) += sigma2_p;
        st->P_delta(9 + k, 9 + k) += sigma2_b_g;
        st->P_delta(12 + k, 12 + k) += sigma2_b_a;
      }
    }
    // memory shift (:251-257)
    st->Delta_q = Delta_q_1;
    st->C_integral = C_integral_1;
    st->acc_integral = acc_integral_1;
    st->cross = cross_1;
    st->dv_db_g = dv_db_g_1;
    time = nexttime;
    ++i;
    if (nexttime == t_end) break;
  }
  return i;
}
}  // namespace

int imu_redo_preintegration(const ImuSamples& s, const ImuParams& p, int64_t t0, int64_t t1,
                            const real sb[9], ImuCache* c) {
  if (s.n == 0 || !(s.t[s.n - 1] >= t1)) return -1;  // :87-89
  PreintState st;
  int i = integrate(s, p, t0, t1, sb, /*static*/ false, /*cov*/ true, &st);
  c->Delta_q = st.Delta_q;
  c->C_integral = st.C_integral;
  c->C_doubleintegral = st.C_doubleintegral;
  c->acc_integral = st.acc_integral;
  c->acc_doubleintegral = st.acc_doubleintegral;
  c->cross = st.cross;
  c->dalpha_db_g = st.dalpha_db_g;
  c->dv_db_g = st.dv_db_g;
  c->dp_db_g = st.dp_db_g;
  for (int k = 0; k < 9; ++k) c->sb_ref[k] = sb[k];  // :264
  // :268-279
  c->P_delta = 0.5 * st.P_delta + 0.5 * st.P_delta.t();
  inverse_lu(c->P_delta.a, 15, c->information.a);
  c->information = 0.5 * c->information + 0.5 * c->information.t();
  sqrt_information_upper(c->information.a, 15, c->sqrtInfo.a);
  return i;
}

void imu_evaluate(const ImuSamples& s, const ImuParams& p, int64_t t0, int64_t t1, ImuCache* c,
                  const real pose0[7], const real sb0[9], const real pose1[7], const real sb1[9],
                  real r[15], real* J0, real* J1, real* J2, real* J3) {
  // ImuError.cpp:520-539
  const Transformation T_WS_0 = Transformation::fromParams(pose0);
  const Transformation T_WS_1 = Transformation::fromParams(pose1);
  const M3 C_WS_0 = T_WS_0.C;
  const M3 C_S0_W = C_WS_0.t();
  // :541-558
  const real Delta_t = nsToSec(t1 - t0);
  Mat<6, 1> Delta_b;
  for (int k = 0; k < 6; ++k) Delta_b[k] = sb0[3 + k] - c->sb_ref[3 + k];
  const real nbg = std::sqrt(Delta_b[0] * Delta_b[0] + Delta_b[1] * Delta_b[1] + Delta_b[2] * Delta_b[2]);
  c->redo = c->redo || (nbg * Delta_t > 0.0001);
  if (c->redo) {
    imu_redo_preintegration(s, p, t0, t1, sb0, c);
    c->redoCounter++;
    for (int k = 0; k < 6; ++k) Delta_b[k] = 0;
    c->redo = false;
  }
  // :561-601
  const V3 g_W = vec3(0, 0, p.g);  // imuParameters_.g * (0,0,6371009).normalized()
  const V3 v0 = vec3(sb0[0], sb0[1], sb0[2]);
  const V3 v1 = vec3(sb1[0], sb1[1], sb1[2]);
  Mat<15, 15> F0 = Mat<15, 15>::Identity();
  const V3 delta_p_est_W = T_WS_0.r - T_WS_1.r + v0 * Delta_t - (0.5 * Delta_t * Delta_t) * g_W;
  const V3 delta_v_est_W = v0 - v1 - g_W * Delta_t;
  const V3 dbg = vec3(Delta_b[0], Delta_b[1], Delta_b[2]);
  const Quat Dq = qmul(deltaQ(-1.0 * (c->dalpha_db_g * dbg)), c->Delta_q);
  F0.setBlock(0, 0, C_S0_W);
  F0.setBlock(0, 3, C_S0_W * crossMx(delta_p_est_W));
  F0.setBlock(0, 6, C_S0_W * Delta_t);
  F0.setBlock(0, 9, c->dp_db_g);
  F0.setBlock(0, 12, -1.0 * c->C_doubleintegral);
  const Quat q1inv = qinv(T_WS_1.q);
  F0.setBlock(3, 3, (qplusMat(qmul(Dq, q1inv)) * qoplusMat(T_WS_0.q)).block<3, 3>(0, 0));
  F0.setBlock(3, 9, (qoplusMat(qmul(q1inv, T_WS_0.q)) * qoplusMat(Dq)).block<3, 3>(0, 0) *
                        (-1.0 * c->dalpha_db_g));
  F0.setBlock(6, 3, C_S0_W * crossMx(delta_v_est_W));
  F0.setBlock(6, 6, C_S0_W);
  F0.setBlock(6, 9, c->dv_db_g);
  F0.setBlock(6, 12, -1.0 * c->C_integral);

  Mat<15, 15> F1 = -1.0 * Mat<15, 15>::Identity();
  F1.setBlock(0, 0, -1.0 * C_S0_W);
  F1.setBlock(3, 3, -1.0 * (qplusMat(Dq) * qoplusMat(T_WS_0.q) * qplusMat(q1inv)).block<3, 3>(0, 0));
  F1.setBlock(6, 6, -1.0 * C_S0_W);

  Mat<15, 1> error;
  {
    V3 e0 = C_S0_W * delta_p_est_W + c->acc_doubleintegral + F0.block<3, 6>(0, 9) * Delta_b;
    Quat qe = qmul(Dq, qmul(q1inv, T_WS_0.q));
    V3 e2 = C_S0_W * delta_v_est_W + c->acc_integral + F0.block<3, 6>(6, 9) * Delta_b;
    for (int k = 0; k < 3; ++k) {
      error[k] = e0[k];
      error[6 + k] = e2[k];
    }
    error[3] = 2 * qe.x;
    error[4] = 2 * qe.y;
    error[5] = 2 * qe.z;
    for (int k = 0; k < 6; ++k) error[9 + k] = sb0[3 + k] - sb1[3 + k];
  }
  Mat<15, 1> werr = c->sqrtInfo * error;
  for (int k = 0; k < 15; ++k) r[k] = werr[k];

  // :608-682
  if (J0) {
    Mat<15, 6> J = c->sqrtInfo * F0.block<15, 6>(0, 0);
    std::memcpy(J0, J.a, sizeof(J.a));
  }
  if (J1) {
    Mat<15, 9> J = c->sqrtInfo * F0.block<15, 9>(0, 6);
    std::memcpy(J1, J.a, sizeof(J.a));
  }
  if (J2) {
    Mat<15, 6> J = c->sqrtInfo * F1.block<15, 6>(0, 0);
    std::memcpy(J2, J.a, sizeof(J.a));
  }
  if (J3) {
    Mat<15, 9> J = c->sqrtInfo * F1.block<15, 9>(0, 6);
    std::memcpy(J3, J.a, sizeof(J.a));
  }
}

int imu_propagation(const ImuSamples& s, const ImuParams& p, real T_WS_io[7], real sb[9],
                    int64_t t_start, int64_t t_end, real* cov, real* jac) {
  if (s.n == 0 || !(s.t[s.n - 1] >= t_end)) return -1;  // ImuError.cpp:301-302
  const Transformation T_WS = Transformation::fromParams(T_WS_io);
  const V3 r_0 = T_WS.r;
  const Quat q_WS_0 = T_WS.q;
  const M3 C_WS_0 = T_WS.C;
  PreintState st;
  int i = integrate(s, p, t_start, t_end, sb, /*static*/ true, /*cov*/ cov != nullptr, &st);
  // :470-477
  const V3 g_W = vec3(0, 0, p.g);
  const V3 v = vec3(sb[0], sb[1], sb[2]);
  const real Dt = st.Delta_t;
  Transformation Tn(r_0 + v * Dt + C_WS_0 * st.acc_doubleintegral - (0.5 * Dt * Dt) * g_W,
                    qmul(q_WS_0, st.Delta_q));
  Tn.toParams(T_WS_io);
  V3 vn = v + C_WS_0 * st.acc_integral - g_W * Dt;
  sb[0] = vn[0];
  sb[1] = vn[1];
  sb[2] = vn[2];
  if (jac) {  // :480-491
    Mat<15, 15> F = Mat<15, 15>::Identity();
    F.setBlock(0, 3, -1.0 * crossMx(C_WS_0 * st.acc_doubleintegral));
    F.setBlock(0, 6, Dt * M3::Identity());
    F.setBlock(0, 9, C_WS_0 * st.dp_db_g);
    F.setBlock(0, 12, -1.0 * (C_WS_0 * st.C_doubleintegral));
    F.setBlock(3, 9, -1.0 * (C_WS_0 * st.dalpha_db_g));
    F.setBlock(6, 3, -1.0 * crossMx(C_WS_0 * st.acc_integral));
    F.setBlock(6, 9, C_WS_0 * st.dv_db_g);
    F.setBlock(6, 12, -1.0 * (C_WS_0 * st.C_integral));
    std::memcpy(jac, F.a, sizeof(F.a));
  }
  if (cov) {  // :494-502
    Mat<15, 15> T = Mat<15, 15>::Identity();
    T.setBlock(0, 0, C_WS_0);
    T.setBlock(3, 3, C_WS_0);
    T.setBlock(6, 6, C_WS_0);
    Mat<15, 15> P = T * st.P_delta * T.t();
    std::memcpy(cov, P.a, sizeof(P.a));
  }
  return i;
}

// ===================================================================================================
// Priors
// ===================================================================================================
void pose_error(const real pose[7], const real meas[7], const real sqrtInfo[36], real r[6],
                real* Jmin) {
  // PoseError.cpp:91-104
  const Transformation T_WS = Transformation::fromParams(pose);
  const Transformation T_m = Transformation::fromParams(meas);
  const Transformation dp = T_m * T_WS.inverse();
  Mat<6, 1> error;
  for (int k = 0; k < 3; ++k) error[k] = T_m.r[k] - T_WS.r[k];
  error[3] = 2 * dp.q.x;
  error[4] = 2 * dp.q.y;
  error[5] = 2 * dp.q.z;
  Mat<6, 6> S;
  std::memcpy(S.a, sqrtInfo, sizeof(S.a));
  Mat<6, 1> w = S * error;
  for (int k = 0; k < 6; ++k) r[k] = w[k];
  if (Jmin) {  // :107-118
    Mat<6, 6> J0 = -1.0 * Mat<6, 6>::Identity();
    J0.setBlock(3, 3, -1.0 * qplusMat(dp.q).block<3, 3>(0, 0));
    Mat<6, 6> J = S * J0;
    std::memcpy(Jmin, J.a, sizeof(J.a));
  }
}

void speedbias_error(const real sb[9], const real meas[9], const real sqrtInfo[81], real r[9],
                     real* Jmin) {
  // SpeedAndBiasError.cpp:93-114
  Mat<9, 9> S;
  std::memcpy(S.a, sqrtInfo, sizeof(S.a));
  Mat<9, 1> e;
  for (int k = 0; k < 9; ++k) e[k] = meas[k] - sb[k];
  Mat<9, 1> w = S * e;
  for (int k = 0; k < 9; ++k) r[k] = w[k];
  if (Jmin) {
    Mat<9, 9> J = -1.0 * S;
    std::memcpy(Jmin, J.a, sizeof(J.a));
  }
}

void relative_pose_error(const real pose0[7], const real pose1[7], const real sqrtInfo[36],
                         real r[6], real* J0min, real* J1min) {
  // RelativePoseError.cpp:88-107
  const Transformation T0 = Transformation::fromParams(pose0);
  const Transformation T1 = Transformation::fromParams(pose1);
  const Transformation dp = T1 * T0.inverse();
  Mat<6, 1> error;
  for (int k = 0; k < 3; ++k) error[k] = T1.r[k] - T0.r[k];
  error[3] = 2 * dp.q.x;
  error[4] = 2 * dp.q.y;
  error[5] = 2 * dp.q.z;
  Mat<6, 6> S;
  std::memcpy(S.a, sqrtInfo, sizeof(S.a));
  Mat<6, 1> w = S * error;
  for (int k = 0; k < 6; ++k) r[k] = w[k];
  if (J0min) {  // :110-133
    Mat<6, 6> J0 = -1.0 * Mat<6, 6>::Identity();
    J0.setBlock(3, 3, -1.0 * qplusMat(dp.q).block<3, 3>(0, 0));
    Mat<6, 6> J = S * J0;
    std::memcpy(J0min, J.a, sizeof(J.a));
  }
  if (J1min) {  // :134-158
    Mat<6, 6> J1 = Mat<6, 6>::Identity();
    J1.setBlock(3, 3, qoplusMat(dp.q).block<3, 3>(0, 0));
    Mat<6, 6> J = S * J1;
    std::memcpy(J1min, J.a, sizeof(J.a));
  }
}

}  // namespace orc
