"""A second, independent statement of the factor VALUES on the hot path, in matrix-form numpy/scipy.

TEST INFRASTRUCTURE ONLY.  Purpose: the reference holds no golden vectors and cannot be built in this
environment, so the fixtures in this directory are produced by the C++ oracle — this module
is the cross-check that the oracle's numbers are not an artefact of one implementation.  It is written
from the equations (which cites the reference lines) with different building blocks
than the oracle: scipy Rotation for every rotation, dense 15x15 numpy algebra for the covariance, numpy's
inverse/Cholesky for the weighting.  `make_golden.py` refuses to write a fixture unless both agree.

Only residual values are re-stated here; Jacobians are pinned by central differences in
tests/test_oracle_factors.py (the way the reference's own tests pin them).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation as R


def rot(q_xyzw):
    q = np.asarray(q_xyzw, float)
    return R.from_quat(q / np.linalg.norm(q)).as_matrix()


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], float)


def distort(model, k, u):
    x, y = u
    rho = x * x + y * y
    if model == 0:
        return np.array([x, y])
    if model == 1:  # radial-tangential, implementation/RadialTangentialDistortion.hpp:105-151
        k1, k2, p1, p2 = k[:4]
        rad = 1 + k1 * rho + k2 * rho ** 2
        return np.array([x * rad + 2 * p1 * x * y + p2 * (rho + 2 * x * x),
                         y * rad + 2 * p2 * x * y + p1 * (rho + 2 * y * y)])
    if model == 2:  # equidistant, implementation/EquidistantDistortion.hpp:105-206
        k1, k2, k3, k4 = k[:4]
        r = np.sqrt(rho)
        if r <= 1e-8:
            return np.array([x, y])
        th = np.arctan(r)
        thd = th * (1 + k1 * th ** 2 + k2 * th ** 4 + k3 * th ** 6 + k4 * th ** 8)
        return np.array([x, y]) * (thd / r)
    if model == 3:  # 8-parameter radial-tangential, implementation/RadialTangentialDistortion8.hpp:125-150
        k1, k2, p1, p2, k3, k4, k5, k6 = k[:8]
        rad = (1 + k1 * rho + k2 * rho ** 2 + k3 * rho ** 3) / (1 + k4 * rho + k5 * rho ** 2 + k6 * rho ** 3)
        return np.array([x * rad + 2 * p1 * x * y + p2 * (rho + 2 * x * x),
                         y * rad + 2 * p2 * x * y + p1 * (rho + 2 * y * y)])
    raise ValueError(model)


def reprojection_residual(pose, point, extr, intr, model, uv, sqrtw):
    """r = L (z - project(T_CS T_SW hp_W))."""
    pose, point, extr, intr = (np.asarray(a, float) for a in (pose, point, extr, intr))
    T_WS = np.eye(4); T_WS[:3, :3] = rot(pose[3:]); T_WS[:3, 3] = pose[:3]
    T_SC = np.eye(4); T_SC[:3, :3] = rot(extr[3:]); T_SC[:3, 3] = extr[:3]
    hp_C = np.linalg.inv(T_SC) @ np.linalg.inv(T_WS) @ point
    p = hp_C[:3] * (-1.0 if hp_C[3] < 0 else 1.0)   # PinholeCamera.hpp:363-369
    d = distort(model, intr[4:], p[:2] / p[2])
    kp = np.array([intr[0] * d[0] + intr[2], intr[1] * d[1] + intr[3]])
    return sqrtw * (np.asarray(uv, float) - kp)


def pose_error_residual(pose, meas, sqrt_info):
    """Appendix A.5: e = [r_meas - r ; 2 vec(q_meas (x) q^-1)]."""
    dq = _qmul(np.asarray(meas[3:], float), _qinv(np.asarray(pose[3:], float)))
    e = np.concatenate([np.asarray(meas[:3]) - np.asarray(pose[:3]), 2 * dq[:3]])
    return np.asarray(sqrt_info).reshape(6, 6) @ e


def speedbias_error_residual(sb, meas, sqrt_info):
    return np.asarray(sqrt_info).reshape(9, 9) @ (np.asarray(meas, float) - np.asarray(sb, float))


def _qmul(a, b):
    av, aw, bv, bw = a[:3], a[3], b[:3], b[3]
    return np.concatenate([aw * bv + bw * av + np.cross(av, bv), [aw * bw - av @ bv]])


def _qinv(q):
    return np.concatenate([-q[:3], [q[3]]]) / (q @ q)


def _right_jacobian(phi):
    """SO(3) right Jacobian (kinematics/implementation/Transformation.hpp:69-82), series form."""
    th = np.linalg.norm(phi)
    K = skew(phi)
    if th < 1e-5:
        return np.eye(3) - 0.5 * K + K @ K / 6.0
    return np.eye(3) - (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K


def imu_preintegrate(t_ns, gyr, acc, prm, t0, t1, sb):
    """Appendix A.4 recursion over [t0, t1] with end-point interpolation; returns the integrals and P."""
    t_ns = np.asarray(t_ns, np.int64)
    gyr = np.asarray(gyr, float).reshape(-1, 3)
    acc = np.asarray(acc, float).reshape(-1, 3)
    bg, ba = np.asarray(sb[3:6], float), np.asarray(sb[6:9], float)
    sec = lambda ns: float(ns) * 1e-9
    q = np.array([0, 0, 0, 1.0])
    Ci = np.zeros((3, 3)); Cii = np.zeros((3, 3)); ai = np.zeros(3); aii = np.zeros(3)
    X = np.zeros((3, 3)); dal = np.zeros((3, 3)); dv = np.zeros((3, 3)); dp = np.zeros((3, 3))
    P = np.zeros((15, 15))
    time, started, n = int(t0), False, 0
    for i in range(len(t_ns) - 1):
        w0, w1, a0, a1 = gyr[i].copy(), gyr[i + 1].copy(), acc[i].copy(), acc[i + 1].copy()
        nxt = int(t_ns[i + 1])
        dt = sec(nxt - time)
        if t1 < nxt:
            interval = sec(nxt - int(t_ns[i]))
            nxt = int(t1)
            dt = sec(nxt - time)
            r = dt / interval
            w1 = (1 - r) * w0 + r * w1
            a1 = (1 - r) * a0 + r * a1
        if dt <= 0:
            continue
        if not started:
            started = True
            r = dt / sec(nxt - int(t_ns[i]))
            w0 = r * w0 + (1 - r) * w1
            a0 = r * a0 + (1 - r) * a1
        sg, sa = prm.sigma_g_c, prm.sigma_a_c
        if max(np.abs(w0).max(), np.abs(w1).max()) > prm.g_max:
            sg *= 100
        if max(np.abs(a0).max(), np.abs(a1).max()) > prm.a_max:
            sa *= 100
        w = 0.5 * (w0 + w1) - bg
        a = 0.5 * (a0 + a1) - ba
        dR = R.from_rotvec(w * dt)
        C = rot(q)
        q1 = _qmul(q, dR.as_quat())
        C1 = rot(q1)
        S = C + C1
        Jr = _right_jacobian(w * dt)
        X1 = dR.as_matrix().T @ X + Jr * dt
        G = C @ skew(a) @ X + C1 @ skew(a) @ X1
        F = np.eye(15)
        F[0:3, 3:6] = -skew(ai * dt + 0.25 * S @ a * dt * dt)
        F[0:3, 6:9] = np.eye(3) * dt
        F[0:3, 9:12] = dt * dv + 0.25 * dt * dt * G
        F[0:3, 12:15] = -Ci * dt + 0.25 * S * dt * dt
        F[3:6, 9:12] = -dt * C1
        F[6:9, 3:6] = -skew(0.5 * S @ a * dt)
        F[6:9, 9:12] = 0.5 * dt * G
        F[6:9, 12:15] = -0.5 * S * dt
        s2v = dt * sa * sa
        Q = np.diag(np.repeat([0.5 * dt * dt * s2v, dt * sg * sg, s2v,
                               dt * prm.sigma_gw_c ** 2, dt * prm.sigma_aw_c ** 2], 3))
        P = F @ P @ F.T + Q
        Cii = Cii + Ci * dt + 0.25 * S * dt * dt
        aii = aii + ai * dt + 0.25 * S @ a * dt * dt
        dp = dp + dt * dv + 0.25 * dt * dt * G
        dv = dv + 0.5 * dt * G
        dal = dal + C1 @ Jr * dt
        Ci = Ci + 0.5 * S * dt
        ai = ai + 0.5 * S @ a * dt
        q, X, time = q1, X1, nxt
        n += 1
        if nxt == t1:
            break
    return dict(q=q, Ci=Ci, Cii=Cii, ai=ai, aii=aii, dal=dal, dv=dv, dp=dp, P=P, n=n)


def imu_residual_fresh(t_ns, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1):
    """Appendix A.4 residual with the preintegration done at sb0 (Delta b = 0). Returns (r, sqrt_info, n)."""
    pre = imu_preintegrate(t_ns, gyr, acc, prm, t0, t1, sb0)
    pose0, sb0, pose1, sb1 = (np.asarray(a, float) for a in (pose0, sb0, pose1, sb1))
    Dt = float(t1 - t0) * 1e-9
    g = np.array([0, 0, prm.g])
    C0T = rot(pose0[3:]).T
    dphat = pose0[:3] - pose1[:3] + sb0[:3] * Dt - 0.5 * g * Dt * Dt
    dvhat = sb0[:3] - sb1[:3] - g * Dt
    qe = _qmul(pre["q"], _qmul(_qinv(pose1[3:]), pose0[3:]))
    e = np.concatenate([C0T @ dphat + pre["aii"], 2 * qe[:3], C0T @ dvhat + pre["ai"], sb0[3:] - sb1[3:]])
    P = 0.5 * (pre["P"] + pre["P"].T)
    info = np.linalg.inv(P)
    info = 0.5 * (info + info.T)
    L = np.linalg.cholesky(info).T
    return L @ e, L, pre["n"]


# ---------------------------------------------------------------------------------------------------
# Marginalisation, matrix form with scipy's symmetric eigen-solver
# ---------------------------------------------------------------------------------------------------
def _pinv_sqrt(V):
    import scipy.linalg as sl
    lam, Q = sl.eigh(V)
    tol = np.finfo(float).eps * V.shape[0] * lam.max()
    s = np.where(lam > tol, 1.0 / np.sqrt(np.where(lam > tol, lam, 1.0)), 0.0)
    return Q * s


def schur_marginalize(H, b0, marg_idx, landmark_blocks=False):
    """One marginalizeOut step: diagonal pre-scaling p (1e-3 where diag <= 1e-9), pseudo-inverse of the
    eliminated block (per 3x3 block when landmark_blocks), Schur complement, un-scaling."""
    n = H.shape[0]
    marg_idx = np.asarray(marg_idx, int)
    keep = np.setdiff1d(np.arange(n), marg_idx)
    d = np.diag(H)
    p = np.where(d > 1e-9, np.sqrt(np.where(d > 1e-9, d, 1.0)), 1e-3)
    Hs = H / np.outer(p, p)
    bs = b0 / p
    U, W, V = Hs[np.ix_(keep, keep)], Hs[np.ix_(keep, marg_idx)], Hs[np.ix_(marg_idx, marg_idx)]
    ba, bb = bs[keep], bs[marg_idx]
    if landmark_blocks:
        Vis = np.zeros_like(V)
        for i in range(0, len(marg_idx), 3):
            Vis[i:i + 3, i:i + 3] = _pinv_sqrt(V[i:i + 3, i:i + 3])
    else:
        Vis = _pinv_sqrt(0.5 * (V + V.T))
    M = W @ Vis
    Hn = (U - M @ M.T) * np.outer(p[keep], p[keep])
    bn = (ba - M @ (Vis.T @ bb)) * p[keep]
    return Hn, bn, keep


def error_computation(H, b0):
    """updateErrorComputation: J, e0 with J^T J = H (eigenvalues below eps*n*max dropped)."""
    import scipy.linalg as sl
    d = np.diag(H)
    p = np.where(d > 1e-9, np.sqrt(np.where(d > 1e-9, d, 1.0)), 1e-3)
    lam, Q = sl.eigh(0.5 * (H + H.T) / np.outer(p, p))
    tol = np.finfo(float).eps * H.shape[0] * lam.max()
    S = np.where(lam > tol, lam, 0.0)
    Sp = np.where(lam > tol, 1.0 / np.where(lam > tol, lam, 1.0), 0.0)
    J = (p[:, None] * Q * np.sqrt(S)).T
    e0 = -(np.sqrt(Sp)[:, None] * Q.T / p) @ b0
    return J, e0, int((lam > tol).sum())
