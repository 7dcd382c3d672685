// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// CPU restatement of the okvis_ceres error terms on the hot path of okvis::Estimator::optimize.
// Every function cites the reference lines it follows (paths relative to /root/reference).
#pragma once
#include "orc_math.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------------
// Camera: PinholeCamera<DISTORTION> (okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp)
// ---------------------------------------------------------------------------------------------------
struct Camera {
  real fu, fv, cu, cv;
  int model;    // OKVIS_BA_DIST_*
  real d[8];  // distortion coefficients
};
enum { DIST_NONE = 0, DIST_RADTAN = 1, DIST_EQUI = 2, DIST_RADTAN8 = 3 };

// D::distort(pointUndistorted, pointDistorted, pointJacobian).  Returns false where the reference
// returns false WITHOUT writing its outputs (RadialTangentialDistortion8, rho>9).
bool distort(const Camera& c, const real u[2], real out[2], real J[4]);

// PinholeCamera<D>::project(point, imagePoint, pointJacobian) (implementation/PinholeCamera.hpp:148-226).
// Returns false in the two cases where the reference leaves its outputs unset (|z|<1e-12, :155-157;
// distortion failure): the oracle then DEFINES kp=(nan) -> caller zeroes residual and Jacobians.
bool project(const Camera& c, const V3& p, real kp[2], Mat<2, 3>* J);

// PinholeCamera<D>::projectHomogeneous (:357-378): negates the point (not the Jacobian) when w<0,
// pads a zero 4th Jacobian column.
bool projectHomogeneous(const Camera& c, const V4& hp, real kp[2], Mat<2, 4>* J);

// ---------------------------------------------------------------------------------------------------
// PoseLocalParameterization (okvis_ceres/src/PoseLocalParameterization.cpp:60-145)
// ---------------------------------------------------------------------------------------------------
void pose_plus(const real x[7], const real delta[6], real out[7]);          // :60-87
void pose_minus(const real x[7], const real x_plus_delta[7], real delta[6]);  // :103-116
void pose_lift_jacobian(const real x[7], real J[6 * 7]);                      // :131-145 (row-major 6x7)
void pose_plus_jacobian(const real x[7], real J[7 * 6]);                      // :119-128 -> Transformation::oplusJacobian

// ---------------------------------------------------------------------------------------------------
// ReprojectionError<GEOMETRY>::EvaluateWithMinimalJacobians
// (okvis_ceres/include/okvis/ceres/implementation/ReprojectionError.hpp:87-242)
// ---------------------------------------------------------------------------------------------------
struct ReprojOut {
  real r[2];      // weighted residual (NOT robustified)
  Mat<2, 6> Jp;     // minimal Jacobian w.r.t. T_WS
  Mat<2, 3> Jl;     // minimal Jacobian w.r.t. landmark (Euclidean part)
  Mat<2, 6> Je;     // minimal Jacobian w.r.t. T_SC
  bool valid;       // false => Jacobians zeroed (:143-151,166,193,218)
  bool defined;     // false => reference behaviour undefined (projection outputs unset); oracle zeroes all
};
// sqrtInfo: 2x2 row-major upper-triangular squareRootInformation_ (:65-73)
void reprojection_error(const real pose[7], const real point[4], const real extr[7],
                        const Camera& cam, const real uv[2], const real sqrtInfo[4], bool jac,
                        ReprojOut* out);

// ---------------------------------------------------------------------------------------------------
// ImuError (okvis_ceres/src/ImuError.cpp)
// ---------------------------------------------------------------------------------------------------
struct ImuParams {
  real sigma_g_c, sigma_a_c, sigma_gw_c, sigma_aw_c, g, g_max, a_max;
};
struct ImuSamples {  // the ImuMeasurementDeque copied into the factor (ImuError.hpp:151-153)
  int n;
  const int64_t* t;   // ns
  const real* gyr;  // [n][3]
  const real* acc;  // [n][3]
};
struct ImuCache {  // the `mutable` preintegration members (ImuError.hpp:248-276)
  Quat Delta_q;
  M3 C_integral, C_doubleintegral;
  V3 acc_integral, acc_doubleintegral;
  M3 cross, dalpha_db_g, dv_db_g, dp_db_g;
  Mat<15, 15> P_delta, information, sqrtInfo;
  real sb_ref[9];
  bool redo;        // redo_ = true initially (ImuError.hpp:271)
  int redoCounter;
  ImuCache() : redo(true), redoCounter(0) {
    for (int i = 0; i < 9; ++i) sb_ref[i] = 0;
  }
};
// ImuError::redoPreintegration (:76-284). Returns number of integration steps or -1.
int imu_redo_preintegration(const ImuSamples& s, const ImuParams& p, int64_t t0, int64_t t1,
                            const real sb[9], ImuCache* c);
// ImuError::EvaluateWithMinimalJacobians (:514-685). J0 15x6, J1 15x9, J2 15x6, J3 15x9 row-major;
// any may be NULL.  The cache is updated in place exactly like the reference's mutable members.
void imu_evaluate(const ImuSamples& s, const ImuParams& p, int64_t t0, int64_t t1, ImuCache* c,
                  const real pose0[7], const real sb0[9], const real pose1[7], const real sb1[9],
                  real r[15], real* J0, real* J1, real* J2, real* J3);
// static ImuError::propagation (:287-504). T_WS[7], sb[9] in/out; cov/jac 15x15 row-major or NULL.
int imu_propagation(const ImuSamples& s, const ImuParams& p, real T_WS[7], real sb[9],
                    int64_t t_start, int64_t t_end, real* cov, real* jac);

// ---------------------------------------------------------------------------------------------------
// small priors
// ---------------------------------------------------------------------------------------------------
// PoseError::EvaluateWithMinimalJacobians (PoseError.cpp:86-138). sqrtInfo 6x6 row-major, J 6x6.
void pose_error(const real pose[7], const real meas[7], const real sqrtInfo[36], real r[6],
                real* Jmin);
// SpeedAndBiasError::EvaluateWithMinimalJacobians (SpeedAndBiasError.cpp:89-118). J 9x9.
void speedbias_error(const real sb[9], const real meas[9], const real sqrtInfo[81], real r[9],
                     real* Jmin);
// RelativePoseError::EvaluateWithMinimalJacobians (RelativePoseError.cpp:84-163). J0,J1 6x6.
void relative_pose_error(const real pose0[7], const real pose1[7], const real sqrtInfo[36],
                         real r[6], real* J0min, real* J1min);

// ---------------------------------------------------------------------------------------------------
// robust loss: Ceres CauchyLoss(b) + Corrector, semantics mirrored in-tree at
// MarginalizationError.cpp:325-365.  rho[0..2] = rho, rho', rho''.
// ---------------------------------------------------------------------------------------------------
inline void cauchy_loss(real b, real s, real rho[3]) {
  const real bb = b * b, c = 1.0 / bb;
  const real sum = 1.0 + s * c;
  const real inv = 1.0 / sum;
  rho[0] = bb * std::log(sum);
  rho[1] = (inv > 0.0) ? inv : 0.0;  // max(numeric_limits::min(), inv) in Ceres; inv>0 always here
  rho[2] = -c * (inv * inv);
}

}  // namespace orc
