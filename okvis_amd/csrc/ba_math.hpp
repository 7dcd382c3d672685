// Device math for the gfx950 sliding-window BA kernels (fp64, register-resident, closed form).
//
// Everything here is written for one work-item holding one observation / one factor in VGPRs: no
// generic matrix class, no temporaries in memory.  BA_HD expands to __host__ __device__ under hipcc and
// to nothing under a plain host compiler, which lets tests/host_probe compile the per-item math for the
// CPU and compare it with the oracle without a GPU (the kernels themselves have no CPU path).
//
// Reference semantics followed (paths relative to the okvis tree):
//   quaternion/pose ops   okvis_kinematics/include/okvis/kinematics/operators.hpp:62-112,
//                         implementation/Transformation.hpp:45-82,246-258
//   pinhole + distortion  okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:148-226,357-378,
//                         RadialTangentialDistortion.hpp:105-151, EquidistantDistortion.hpp:105-206,
//                         RadialTangentialDistortion8.hpp:125-150
//   reprojection factor   okvis_ceres/include/okvis/ceres/implementation/ReprojectionError.hpp:87-242
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BA_HD __host__ __device__ __forceinline__
#else
#define BA_HD inline
#endif

namespace ba {

enum { DIST_NONE = 0, DIST_RADTAN = 1, DIST_EQUI = 2, DIST_RADTAN8 = 3 };

// ---- quaternions (x,y,z,w), Hamilton ---------------------------------------------------------------
BA_HD void qmul(const double* a, const double* b, double* r) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  r[0] = aw * bx + ax * bw + ay * bz - az * by;
  r[1] = aw * by + ay * bw + az * bx - ax * bz;
  r[2] = aw * bz + az * bw + ax * by - ay * bx;
  r[3] = aw * bw - ax * bx - ay * by - az * bz;
}
// Eigen toRotationMatrix() formula, no normalisation (ReprojectionError uses the raw parameters).
// R row-major.
BA_HD void qrot(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
BA_HD void qnormalize(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// Eigen Quaternion::inverse(): conjugate / squaredNorm
BA_HD void qinv(const double* q, double* r) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  r[0] = -q[0] / n2; r[1] = -q[1] / n2; r[2] = -q[2] / n2; r[3] = q[3] / n2;
}
// qnormalize / qinv / qmul with every product and sum rounded on its own (no fused multiply-add), for the places where the
// reference's result is an EXACT cancellation: PoseError / RelativePoseError form dq = q_m (x) q^-1 through Transformation's
// normalising constructor (implementation/Transformation.hpp:110-117, :170-173, :216-220); a pose that sits at its prior gives
// dq.xyz = w x - x w + ... = 0 exactly when the products are rounded one by one, and 1e-17 when the compiler contracts them —
// times the prior's 1e16 a gradient entry of 0.04 where the reference has none (found by the long double referee: the reduced
// gradient 3e-8 of its largest entry off, b0 of a marginalisation 1.6e-7; the step does not see it, the entry's pivot is 1e16).
#if defined(__clang__)
#define BA_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define BA_NO_CONTRACT
#endif
BA_HD void qnormalize_strict(double* q) {
  BA_NO_CONTRACT
  const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
BA_HD void qinv_strict(const double* q, double* r) {
  BA_NO_CONTRACT
  const double n2 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  r[0] = -q[0] / n2; r[1] = -q[1] / n2; r[2] = -q[2] / n2; r[3] = q[3] / n2;
}
BA_HD void qmul_strict(const double* a, const double* b, double* r) {
  BA_NO_CONTRACT
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  r[0] = ((aw * bx + ax * bw) + ay * bz) - az * by;
  r[1] = ((aw * by + ay * bw) + az * bx) - ax * bz;
  r[2] = ((aw * bz + az * bw) + ax * by) - ay * bx;
  r[3] = ((aw * bw - ax * bx) - ay * by) - az * bz;
}
BA_HD double sinc(double x) {
  if (fabs(x) > 1e-6) return sin(x) / x;
  const double x2 = x * x, x4 = x2 * x2, x6 = x2 * x2 * x2;
  return 1.0 - (1.0 / 6.0) * x2 + (1.0 / 120.0) * x4 - (1.0 / 5040.0) * x6;
}
// top-left 3x3 of okvis::kinematics::plus(q) / oplus(q) (operators.hpp:92-112), row-major
BA_HD void qplus33(const double* q, double* M) {
  M[0] = q[3];  M[1] = -q[2]; M[2] = q[1];
  M[3] = q[2];  M[4] = q[3];  M[5] = -q[0];
  M[6] = -q[1]; M[7] = q[0];  M[8] = q[3];
}
BA_HD void qoplus33(const double* q, double* M) {
  M[0] = q[3];  M[1] = q[2];  M[2] = -q[1];
  M[3] = -q[2]; M[4] = q[3];  M[5] = q[0];
  M[6] = q[1];  M[7] = -q[0]; M[8] = q[3];
}
// full 4x4 plus / oplus matrices, row-major
BA_HD void qplus44(const double* q, double* Q) {
  Q[0] = q[3];  Q[1] = -q[2]; Q[2] = q[1];  Q[3] = q[0];
  Q[4] = q[2];  Q[5] = q[3];  Q[6] = -q[0]; Q[7] = q[1];
  Q[8] = -q[1]; Q[9] = q[0];  Q[10] = q[3]; Q[11] = q[2];
  Q[12] = -q[0]; Q[13] = -q[1]; Q[14] = -q[2]; Q[15] = q[3];
}
BA_HD void qoplus44(const double* q, double* Q) {
  Q[0] = q[3];  Q[1] = q[2];  Q[2] = -q[1]; Q[3] = q[0];
  Q[4] = -q[2]; Q[5] = q[3];  Q[6] = q[0];  Q[7] = q[1];
  Q[8] = q[1];  Q[9] = -q[0]; Q[10] = q[3]; Q[11] = q[2];
  Q[12] = -q[0]; Q[13] = -q[1]; Q[14] = -q[2]; Q[15] = q[3];
}

// ---- 3x3 helpers (row-major) ----------------------------------------------------------------------
BA_HD void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
BA_HD void mat3_mulT(const double* A, const double* B, double* C) {  // A * B^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
BA_HD void mat3_Tmul(const double* A, const double* B, double* C) {  // A^T * B
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
BA_HD void mat3_vec(const double* A, const double* v, double* r) {
  r[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  r[1] = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  r[2] = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
}
BA_HD void mat3_Tvec(const double* A, const double* v, double* r) {
  r[0] = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
  r[1] = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
  r[2] = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
}
BA_HD void cross_mx(const double* v, double* C) {  // okvis::kinematics::crossMx (operators.hpp:62-76)
  C[0] = 0;     C[1] = -v[2]; C[2] = v[1];
  C[3] = v[2];  C[4] = 0;     C[5] = -v[0];
  C[6] = -v[1]; C[7] = v[0];  C[8] = 0;
}
// rightJacobian (implementation/Transformation.hpp:69-82)
BA_HD void right_jacobian(const double* phi, double* J) {
  const double Phi = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  double X[9], X2[9];
  cross_mx(phi, X);
  mat3_mul(X, X, X2);
  double a, b;
  if (Phi < 1.0e-4) {
    a = -0.5;
    b = 1.0 / 6.0;
  } else {
    const double Phi2 = Phi * Phi, Phi3 = Phi2 * Phi;
    a = -(1.0 - cos(Phi)) / Phi2;
    b = (Phi - sin(Phi)) / Phi3;
  }
  for (int i = 0; i < 9; ++i) J[i] = a * X[i] + b * X2[i];
  J[0] += 1.0; J[4] += 1.0; J[8] += 1.0;
}
// inverse of symmetric 3x3 (upper-tri 00 01 02 11 12 22) by cofactors
BA_HD void inv3sym(const double* v, double* o) {
  const double a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double id = 1.0 / (a * c00 + b * c01 + c * c02);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}
// eigenvalues of a symmetric 3x3 by cyclic Jacobi; returns min and max (Estimator.cpp:880-896 needs
// only those two)
BA_HD void eig3sym_minmax(const double* v, double* emin, double* emax) {
  double A[3][3] = {{v[0], v[1], v[2]}, {v[1], v[3], v[4]}, {v[2], v[4], v[5]}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
      }
  }
  const double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  *emin = fmin(e0, fmin(e1, e2));
  *emax = fmax(e0, fmax(e1, e2));
}

// Preconditioned pseudo-inverse of a symmetric PSD 3x3 (upper-tri 00 01 02 11 12 22): what
// MarginalizationError::marginalizeOut applies to ONE landmark block (MarginalizationError.cpp:617-684 with
// pseudoInverseSymmSqrt, implementation/MarginalizationError.hpp:215-243), written in unscaled variables:
//   p_i = sqrt(V_ii) if V_ii > 1e-9 else 1e-3 (:619);  Vs = P^-1 V P^-1 = Q diag(l) Q^T;
//   Vs^+ keeps the eigenvalues l > eps * 3 * l_max;  result = P^-1 Vs^+ P^-1
// so that W result W^T equals the reference's unscale(M M^T), M = (P_a^-1 W P^-1) Q diag(l^-1/2): the kept-side
// scaling P_a cancels.  Eigen-decomposition by cyclic Jacobi.
BA_HD void pinv3sym_precond(const double* v, double* o) {
  const double p[3] = {v[0] > 1.0e-9 ? sqrt(v[0]) : 1.0e-3, v[3] > 1.0e-9 ? sqrt(v[3]) : 1.0e-3,
                       v[5] > 1.0e-9 ? sqrt(v[5]) : 1.0e-3};
  double A[3][3] = {{v[0] / (p[0] * p[0]), v[1] / (p[0] * p[1]), v[2] / (p[0] * p[2])},
                    {v[1] / (p[0] * p[1]), v[3] / (p[1] * p[1]), v[4] / (p[1] * p[2])},
                    {v[2] / (p[0] * p[2]), v[4] / (p[1] * p[2]), v[5] / (p[2] * p[2])}};
  double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-300) break;
    for (int a = 0; a < 2; ++a)
      for (int b = a + 1; b < 3; ++b) {
        if (A[a][b] == 0.0) continue;
        const double theta = (A[b][b] - A[a][a]) / (2.0 * A[a][b]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double aka = A[k][a], akb = A[k][b];
          A[k][a] = c * aka - s * akb;
          A[k][b] = s * aka + c * akb;
          const double qka = Q[k][a], qkb = Q[k][b];
          Q[k][a] = c * qka - s * qkb;
          Q[k][b] = s * qka + c * qkb;
        }
        for (int k = 0; k < 3; ++k) {
          const double aak = A[a][k], abk = A[b][k];
          A[a][k] = c * aak - s * abk;
          A[b][k] = s * aak + c * abk;
        }
      }
  }
  const double lmax = fmax(A[0][0], fmax(A[1][1], A[2][2]));
  const double tol = 2.220446049250313e-16 * 3.0 * lmax;
  double li[3];
  for (int k = 0; k < 3; ++k) li[k] = A[k][k] > tol ? 1.0 / A[k][k] : 0.0;
  const int ut[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  for (int e = 0; e < 6; ++e) {
    const int i = ut[e][0], j = ut[e][1];
    double s = 0;
    for (int k = 0; k < 3; ++k) s += Q[i][k] * li[k] * Q[j][k];
    o[e] = s / (p[i] * p[j]);
  }
}

// ---- pose (+) / (-) ----------------------------------------------------------------------------------
// Transformation::oplus via PoseLocalParameterization::plus (PoseLocalParameterization.cpp:60-87):
// constructs Transformation(r, q) (normalises q), r += d[0:3], q = normalise(dq(d[3:6]) (x) q)
BA_HD void pose_oplus(const double* x, const double* d, double* out) {
  double q[4] = {x[3], x[4], x[5], x[6]};
  qnormalize(q);
  out[0] = x[0] + d[0];
  out[1] = x[1] + d[1];
  out[2] = x[2] + d[2];
  const double halfnorm = 0.5 * sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  const double s = sinc(halfnorm) * 0.5;
  const double dq[4] = {s * d[3], s * d[4], s * d[5], cos(halfnorm)};
  double qn[4];
  qmul(dq, q, qn);
  qnormalize(qn);
  out[3] = qn[0]; out[4] = qn[1]; out[5] = qn[2]; out[6] = qn[3];
}
// PoseLocalParameterization::minus (:103-116): delta = [xp.r - x.r ; 2 vec(q_xp (x) q_x^-1)]
BA_HD void pose_ominus(const double* x, const double* xp, double* d) {
  d[0] = xp[0] - x[0];
  d[1] = xp[1] - x[1];
  d[2] = xp[2] - x[2];
  double qi[4], dq[4];
  qinv_strict(x + 3, qi);   // (xp = x gives exactly zero, as in the reference: see qmul_strict)
  qmul_strict(xp + 3, qi, dq);
  d[3] = 2 * dq[0]; d[4] = 2 * dq[1]; d[5] = 2 * dq[2];
}

// ---- camera ------------------------------------------------------------------------------------------
// D::distort(u) -> d, Jd (2x2 row-major). Returns false where the reference leaves outputs unset.
BA_HD bool distort(int model, const double* k, double u0, double u1, double* d, double* J) {
  if (model == DIST_RADTAN) {
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
    const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    const double rad = k1 * rho + k2 * rho * rho;
    d[0] = u0 + u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
    d[1] = u1 + u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
    J[0] = 1 + rad + k1 * 2.0 * mx + k2 * rho * 4 * mx + 2.0 * p1 * u1 + 6 * p2 * u0;
    J[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
    J[1] = J[2];
    J[3] = 1 + rad + k1 * 2.0 * my + k2 * rho * 4 * my + 6 * p1 * u1 + 2.0 * p2 * u0;
    return true;
  } else if (model == DIST_EQUI) {
    const double k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3];
    const double r2 = u0 * u0 + u1 * u1;
    const double r = sqrt(r2);
    const double th = atan(r);
    const double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    const double poly = 1 + k1 * th2 + k2 * th4 + k3 * th6 + k4 * th8;
    const double thd = th * poly;
    if (r > 1e-8) {
      const double sc = thd / r;
      d[0] = sc * u0;
      d[1] = sc * u1;
      // d(thd)/d(th) * d(th)/dr, th = atan r
      const double dpoly = 1 + 3 * k1 * th2 + 5 * k2 * th4 + 7 * k3 * th6 + 9 * k4 * th8;
      const double dthd_dr = dpoly / (1.0 + r2);
      // d_i = u_i * thd/r  =>  dd_i/du_j = delta_ij thd/r + u_i u_j (dthd_dr - thd/r)/r^2
      const double g = (dthd_dr - sc) / r2;
      J[0] = sc + u0 * u0 * g;
      J[1] = u0 * u1 * g;
      J[2] = J[1];
      J[3] = sc + u1 * u1 * g;
    } else {
      d[0] = u0;
      d[1] = u1;
      J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1;
    }
    return true;
  } else if (model == DIST_RADTAN8) {
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    if (rho > 9.0) return false;
    const double num = 1.0 + ((k3 * rho + k2) * rho + k1) * rho;
    const double den = 1.0 + ((k6 * rho + k5) * rho + k4) * rho;
    const double rad = num / den;
    d[0] = u0 * rad + 2.0 * p1 * mxy + p2 * (rho + 2.0 * mx);
    d[1] = u1 * rad + 2.0 * p2 * mxy + p1 * (rho + 2.0 * my);
    // d(rad)/d(rho)
    const double dnum = k1 + rho * (2 * k2 + 3 * k3 * rho);
    const double dden = k4 + rho * (2 * k5 + 3 * k6 * rho);
    const double drad = (dnum * den - num * dden) / (den * den);
    J[0] = rad + 2 * mx * drad + 2.0 * p1 * u1 + 6.0 * p2 * u0;
    J[1] = 2 * mxy * drad + 2.0 * p1 * u0 + 2.0 * p2 * u1;
    J[2] = J[1];
    J[3] = rad + 2 * my * drad + 6.0 * p1 * u1 + 2.0 * p2 * u0;
    return true;
  }
  d[0] = u0;
  d[1] = u1;
  J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1;
  return true;
}

struct ReprojLin {
  double r[2];     // weighted residual, NOT robustified
  double Jp[12];   // 2x6 minimal Jacobian w.r.t. T_WS
  double Jl[6];    // 2x3 minimal Jacobian w.r.t. the landmark
  double Je[12];   // 2x6 minimal Jacobian w.r.t. T_SC
};

// ReprojectionError<G>::EvaluateWithMinimalJacobians with information = sw^2 * I2.
// intr = fu fv cu cv d0..d7.  Returns the `valid` flag (Jacobians are zero when false).
BA_HD bool reproj_linearize(const double* pose, const double* ext, const double* lm, const double* intr,
                            int model, double mu, double mv, double sw, bool want_ext, ReprojLin* o) {
  double C_WS[9], C_SC[9];
  qrot(pose + 3, C_WS);
  qrot(ext + 3, C_SC);
  const double w = lm[3];
  // hp_S = T_SW hp_W, hp_C = T_CS hp_S (implementation/ReprojectionError.hpp:110-121)
  const double dW[3] = {lm[0] - pose[0] * w, lm[1] - pose[1] * w, lm[2] - pose[2] * w};
  double pS[3];
  mat3_Tvec(C_WS, dW, pS);
  const double eS[3] = {pS[0] - ext[0] * w, pS[1] - ext[1] * w, pS[2] - ext[2] * w};
  double pC[3];
  mat3_Tvec(C_SC, eS, pC);
  // projectHomogeneous (PinholeCamera.hpp:357-378): w<0 -> project(-head), Jacobian sign kept
  double x = pC[0], y = pC[1], z = pC[2];
  if (w < 0) {
    x = -x; y = -y; z = -z;
  }
  for (int i = 0; i < 12; ++i) o->Jp[i] = 0.0;
  for (int i = 0; i < 6; ++i) o->Jl[i] = 0.0;
  for (int i = 0; i < 12; ++i) o->Je[i] = 0.0;
  bool defined = fabs(z) >= 1.0e-12;  // PinholeCamera.hpp:155-157
  double dd[2] = {0, 0}, Jd[4] = {1, 0, 0, 1};
  double rz = 0, rz2 = 0;
  if (defined) {
    rz = 1.0 / z;
    rz2 = rz * rz;
    defined = distort(model, intr + 4, x * rz, y * rz, dd, Jd);
  }
  if (!defined) {  // reference behaviour undefined (outputs unset, status ignored): zero everything
    o->r[0] = 0.0;
    o->r[1] = 0.0;
    return false;
  }
  const double fu = intr[0], fv = intr[1];
  o->r[0] = sw * (mu - (fu * dd[0] + intr[2]));
  o->r[1] = sw * (mv - (fv * dd[1] + intr[3]));
  // validity (:143-151)
  bool valid = true;
  if (fabs(w) > 1.0e-8)
    if (pC[2] / w < 0.2) valid = false;
  if (!valid) return false;
  // Jw = sw * J_project (2x3) (PinholeCamera.hpp:196-206)
  double Jw[6];
  Jw[0] = sw * fu * Jd[0] * rz;
  Jw[1] = sw * fu * Jd[1] * rz;
  Jw[2] = -sw * fu * (x * Jd[0] + y * Jd[1]) * rz2;
  Jw[3] = sw * fv * Jd[2] * rz;
  Jw[4] = sw * fv * Jd[3] * rz;
  Jw[5] = -sw * fv * (x * Jd[2] + y * Jd[3]) * rz2;
  // B = Jw * C_CS = Jw * C_SC^T (2x3)
  double B[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      B[3 * i + j] = Jw[3 * i] * C_SC[3 * j] + Jw[3 * i + 1] * C_SC[3 * j + 1] + Jw[3 * i + 2] * C_SC[3 * j + 2];
  // A = B * C_SW = B * C_WS^T;  J_lm = -A  (:188-206)
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      o->Jl[3 * i + j] = -(B[3 * i] * C_WS[3 * j] + B[3 * i + 1] * C_WS[3 * j + 1] + B[3 * i + 2] * C_WS[3 * j + 2]);
  // J_pose = A [I w, -[dW]x]  (:156-167):  translation = -w * Jl ; rotation rows = Jl_row x dW
  for (int i = 0; i < 2; ++i) {
    const double a0 = o->Jl[3 * i], a1 = o->Jl[3 * i + 1], a2 = o->Jl[3 * i + 2];
    o->Jp[6 * i + 0] = -w * a0;
    o->Jp[6 * i + 1] = -w * a1;
    o->Jp[6 * i + 2] = -w * a2;
    o->Jp[6 * i + 3] = a1 * dW[2] - a2 * dW[1];
    o->Jp[6 * i + 4] = a2 * dW[0] - a0 * dW[2];
    o->Jp[6 * i + 5] = a0 * dW[1] - a1 * dW[0];
  }
  if (want_ext) {
    // J_ext = B [I w_S, -[eS]x]  (:208-219): rotation rows = -(B_row x eS) = eS x B_row
    for (int i = 0; i < 2; ++i) {
      const double b0 = B[3 * i], b1 = B[3 * i + 1], b2 = B[3 * i + 2];
      o->Je[6 * i + 0] = w * b0;
      o->Je[6 * i + 1] = w * b1;
      o->Je[6 * i + 2] = w * b2;
      o->Je[6 * i + 3] = eS[1] * b2 - eS[2] * b1;
      o->Je[6 * i + 4] = eS[2] * b0 - eS[0] * b2;
      o->Je[6 * i + 5] = eS[0] * b1 - eS[1] * b0;
    }
  }
  return true;
}

// ---- reduced-precision linearisation (BASELINE configs[4]: fp32 Jacobian/Hessian build, fp64 solve) ------
// Same algebra as distort()/reproj_linearize() above with the arithmetic in T (float).  The only fp64
// operations are the two translations hp_W - r_WS w and p_S - r_SC w_S: subtracting world-scale coordinates
// in fp32 would lose the digits the residual lives in; everything after the points are expressed relative to
// the sensor/camera (rotations, projection, distortion, Jacobians) is T.
template <class T>
BA_HD void qrot_t(const T* q, T* R) {
  const T tx = T(2) * q[0], ty = T(2) * q[1], tz = T(2) * q[2];
  const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = T(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = T(1) - (txx + tyy);
}
template <class T>
BA_HD bool distort_t(int model, const T* k, T u0, T u1, T* d, T* J) {
  if (model == DIST_RADTAN) {
    const T k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3];
    const T mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    const T rad = k1 * rho + k2 * rho * rho;
    d[0] = u0 + u0 * rad + T(2) * p1 * mxy + p2 * (rho + T(2) * mx);
    d[1] = u1 + u1 * rad + T(2) * p2 * mxy + p1 * (rho + T(2) * my);
    J[0] = T(1) + rad + k1 * T(2) * mx + k2 * rho * T(4) * mx + T(2) * p1 * u1 + T(6) * p2 * u0;
    J[2] = k1 * T(2) * u0 * u1 + k2 * T(4) * rho * u0 * u1 + p1 * T(2) * u0 + T(2) * p2 * u1;
    J[1] = J[2];
    J[3] = T(1) + rad + k1 * T(2) * my + k2 * rho * T(4) * my + T(6) * p1 * u1 + T(2) * p2 * u0;
    return true;
  } else if (model == DIST_EQUI) {
    const T k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3];
    const T r2 = u0 * u0 + u1 * u1;
    const T r = sqrt(r2);
    const T th = atan(r);
    const T th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
    const T thd = th * (T(1) + k1 * th2 + k2 * th4 + k3 * th6 + k4 * th8);
    if (r > T(1e-8)) {
      const T sc = thd / r;
      d[0] = sc * u0;
      d[1] = sc * u1;
      const T dpoly = T(1) + T(3) * k1 * th2 + T(5) * k2 * th4 + T(7) * k3 * th6 + T(9) * k4 * th8;
      const T g = (dpoly / (T(1) + r2) - sc) / r2;
      J[0] = sc + u0 * u0 * g;
      J[1] = u0 * u1 * g;
      J[2] = J[1];
      J[3] = sc + u1 * u1 * g;
    } else {
      d[0] = u0;
      d[1] = u1;
      J[0] = T(1); J[1] = T(0); J[2] = T(0); J[3] = T(1);
    }
    return true;
  } else if (model == DIST_RADTAN8) {
    const T k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const T mx = u0 * u0, my = u1 * u1, mxy = u0 * u1, rho = mx + my;
    if (rho > T(9)) return false;
    const T num = T(1) + ((k3 * rho + k2) * rho + k1) * rho;
    const T den = T(1) + ((k6 * rho + k5) * rho + k4) * rho;
    const T rad = num / den;
    d[0] = u0 * rad + T(2) * p1 * mxy + p2 * (rho + T(2) * mx);
    d[1] = u1 * rad + T(2) * p2 * mxy + p1 * (rho + T(2) * my);
    const T dnum = k1 + rho * (T(2) * k2 + T(3) * k3 * rho);
    const T dden = k4 + rho * (T(2) * k5 + T(3) * k6 * rho);
    const T drad = (dnum * den - num * dden) / (den * den);
    J[0] = rad + T(2) * mx * drad + T(2) * p1 * u1 + T(6) * p2 * u0;
    J[1] = T(2) * mxy * drad + T(2) * p1 * u0 + T(2) * p2 * u1;
    J[2] = J[1];
    J[3] = rad + T(2) * my * drad + T(6) * p1 * u1 + T(2) * p2 * u0;
    return true;
  }
  d[0] = u0;
  d[1] = u1;
  J[0] = T(1); J[1] = T(0); J[2] = T(0); J[3] = T(1);
  return true;
}

template <class T>
struct ReprojLinT {
  T r[2], Jp[12], Jl[6], Je[12];
};

template <class T>
BA_HD bool reproj_linearize_mixed(const double* pose, const double* ext, const double* lm, const double* intr,
                                  int model, double mu, double mv, double sw, bool want_ext, ReprojLinT<T>* o) {
  const T qp[4] = {T(pose[3]), T(pose[4]), T(pose[5]), T(pose[6])};
  const T qe[4] = {T(ext[3]), T(ext[4]), T(ext[5]), T(ext[6])};
  T C_WS[9], C_SC[9];
  qrot_t(qp, C_WS);
  qrot_t(qe, C_SC);
  const double wd = lm[3];
  const T w = T(wd);
  const T dW[3] = {T(lm[0] - pose[0] * wd), T(lm[1] - pose[1] * wd), T(lm[2] - pose[2] * wd)};  // fp64 difference
  T pS[3];
  for (int j = 0; j < 3; ++j) pS[j] = C_WS[j] * dW[0] + C_WS[3 + j] * dW[1] + C_WS[6 + j] * dW[2];
  const T eS[3] = {pS[0] - T(ext[0]) * w, pS[1] - T(ext[1]) * w, pS[2] - T(ext[2]) * w};  // |r_SC| ~ 0.1 m: T is enough
  T pC[3];
  for (int j = 0; j < 3; ++j) pC[j] = C_SC[j] * eS[0] + C_SC[3 + j] * eS[1] + C_SC[6 + j] * eS[2];
  T x = pC[0], y = pC[1], z = pC[2];
  if (wd < 0) {
    x = -x; y = -y; z = -z;
  }
  for (int i = 0; i < 12; ++i) o->Jp[i] = T(0);
  for (int i = 0; i < 6; ++i) o->Jl[i] = T(0);
  for (int i = 0; i < 12; ++i) o->Je[i] = T(0);
  bool defined = fabs(z) >= T(1.0e-12);
  T dd[2] = {T(0), T(0)}, Jd[4] = {T(1), T(0), T(0), T(1)};
  T rz = T(0), rz2 = T(0);
  T kk[8];
  for (int i = 0; i < 8; ++i) kk[i] = T(intr[4 + i]);
  if (defined) {
    rz = T(1) / z;
    rz2 = rz * rz;
    defined = distort_t<T>(model, kk, x * rz, y * rz, dd, Jd);
  }
  if (!defined) {
    o->r[0] = T(0);
    o->r[1] = T(0);
    return false;
  }
  const T fu = T(intr[0]), fv = T(intr[1]), s = T(sw);
  // the measurement minus the principal point is formed in fp64 (pixel coordinates ~1e2..1e3)
  o->r[0] = s * (T(mu - intr[2]) - fu * dd[0]);
  o->r[1] = s * (T(mv - intr[3]) - fv * dd[1]);
  bool valid = true;
  if (fabs(wd) > 1.0e-8)
    if (pC[2] / w < T(0.2)) valid = false;
  if (!valid) return false;
  T Jw[6];
  Jw[0] = s * fu * Jd[0] * rz;
  Jw[1] = s * fu * Jd[1] * rz;
  Jw[2] = -s * fu * (x * Jd[0] + y * Jd[1]) * rz2;
  Jw[3] = s * fv * Jd[2] * rz;
  Jw[4] = s * fv * Jd[3] * rz;
  Jw[5] = -s * fv * (x * Jd[2] + y * Jd[3]) * rz2;
  T B[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      B[3 * i + j] = Jw[3 * i] * C_SC[3 * j] + Jw[3 * i + 1] * C_SC[3 * j + 1] + Jw[3 * i + 2] * C_SC[3 * j + 2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      o->Jl[3 * i + j] = -(B[3 * i] * C_WS[3 * j] + B[3 * i + 1] * C_WS[3 * j + 1] + B[3 * i + 2] * C_WS[3 * j + 2]);
  for (int i = 0; i < 2; ++i) {
    const T a0 = o->Jl[3 * i], a1 = o->Jl[3 * i + 1], a2 = o->Jl[3 * i + 2];
    o->Jp[6 * i + 0] = -w * a0;
    o->Jp[6 * i + 1] = -w * a1;
    o->Jp[6 * i + 2] = -w * a2;
    o->Jp[6 * i + 3] = a1 * dW[2] - a2 * dW[1];
    o->Jp[6 * i + 4] = a2 * dW[0] - a0 * dW[2];
    o->Jp[6 * i + 5] = a0 * dW[1] - a1 * dW[0];
  }
  if (want_ext) {
    for (int i = 0; i < 2; ++i) {
      const T b0 = B[3 * i], b1 = B[3 * i + 1], b2 = B[3 * i + 2];
      o->Je[6 * i + 0] = w * b0;
      o->Je[6 * i + 1] = w * b1;
      o->Je[6 * i + 2] = w * b2;
      o->Je[6 * i + 3] = eS[1] * b2 - eS[2] * b1;
      o->Je[6 * i + 4] = eS[2] * b0 - eS[0] * b2;
      o->Je[6 * i + 5] = eS[0] * b1 - eS[1] * b0;
    }
  }
  return true;
}

// okvis::Duration::toSec() of a signed ns difference (Duration.hpp:111-113, Duration.cpp:55-73)
BA_HD double ns_to_sec(long long ns) {
  long long sec = ns / 1000000000LL;
  long long nsec = ns % 1000000000LL;
  if (nsec < 0) {
    nsec += 1000000000LL;
    --sec;
  }
  return (double)sec + 1e-9 * (double)nsec;
}

}  // namespace ba
