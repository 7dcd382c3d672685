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


# ---------------------------------------------------------------------------------------------------------------------
# Third statement of the trust-region policy the reference configures (Estimator.cpp:854-873: TRUST_REGION, DOGLEG with
# Ceres 1.9 defaults: traditional dogleg, Jacobi scaling) - numpy, dense, on the FULL normal equations (no Schur complement),
# written from the description of Ceres 1.9's TrustRegionMinimizer / DoglegStrategy, not from the oracle's code.  It drives
# any object with  cost() -> float,  full_system() -> (H, b0 = -gradient, [(type, index)] with 0 pose / 1 speed-bias /
# 2 landmark in column order),  get_state() / set_state(pose, sb, lm);  tests/test_dogleg_policy.py hands it the reference's
# own factors (tests/ref_lib.RefWindow).
def _oplus_blocks(state, blocks, delta):
    pose, sb, lm = (a.copy() for a in state)
    o = 0
    for typ, idx in blocks:
        if typ == 0:
            d = delta[o:o + 6]
            dq_v = d[3:]                      # quaternion increment of the reference: dq = (sinc(|da|/2) da/2, cos(|da|/2))
            n = np.linalg.norm(dq_v)
            half = 0.5 * n
            s = 0.5 if n < 1e-12 else np.sin(half) / n
            dq = np.r_[s * dq_v, np.cos(half)]
            pose[idx, :3] += d[:3]
            q = _qmul(dq, pose[idx, 3:])
            pose[idx, 3:] = q / np.linalg.norm(q)
            o += 6
        elif typ == 1:
            sb[idx] += delta[o:o + 9]
            o += 9
        else:
            lm[idx, :3] += delta[o:o + 3]
            o += 3
    return pose, sb, lm


def _ambient(state, blocks):
    pose, sb, lm = state
    return np.concatenate([pose[i] if t == 0 else sb[i] if t == 1 else lm[i] for t, i in blocks])


def dogleg_minimize(win, max_iter, initial_radius=1e4, jacobi_scaling=True, function_tolerance=1e-6, gradient_tolerance=1e-10,
                    parameter_tolerance=1e-8, min_relative_decrease=1e-3, max_invalid=5, min_radius=1e-32, max_radius=1e16):
    """returns dict(iterations, successful_steps, termination, final_cost, initial_cost, final_radius)
    termination: 0 iteration limit, 1 function tolerance, 2 gradient tolerance, 3 parameter tolerance, 4 radius, 5 invalid steps"""
    MIN_DIAG, MAX_DIAG, MIN_MU, MAX_MU, MU_UP = 1e-6, 1e32, 1e-8, 1.0, 10.0
    x = win.get_state()
    cost = win.cost()
    H, b0, blocks = win.full_system()
    g = -b0
    n = g.size
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0.0))) if jacobi_scaling else np.ones(n)
    out = dict(initial_cost=cost, iterations=0, successful_steps=0, termination=0)

    def gradient_max_norm(x, g, blocks):
        xm = _oplus_blocks(x, blocks, -g)
        return np.abs(_ambient(x, blocks) - _ambient(xm, blocks)).max()

    radius, mu, reuse, invalid = float(initial_radius), MIN_MU, False, 0
    if gradient_max_norm(x, g, blocks) <= gradient_tolerance:
        out.update(termination=2, final_cost=cost, final_radius=radius)
        return out
    gn = cauchy_alpha = ghat = d = None
    while out["iterations"] < max_iter:
        out["iterations"] += 1
        Ht = H * np.outer(scale, scale)          # J S
        gt = g * scale
        ok = True
        if not reuse:
            d = np.sqrt(np.clip(np.diag(Ht), MIN_DIAG, MAX_DIAG))
            ghat = gt / d                        # gradient in the diagonally scaled space
            v = ghat / d
            cauchy_alpha = (ghat @ ghat) / (v @ Ht @ v)
            gn = None
            while True:                          # Gauss-Newton point with a little regularisation mu D^2, raised on failure
                try:
                    L = np.linalg.cholesky(Ht + mu * np.diag(d * d))
                    sol = np.linalg.solve(L.T, np.linalg.solve(L, gt))
                    if np.all(np.isfinite(sol)):
                        gn = -sol * d
                        break
                except np.linalg.LinAlgError:
                    pass
                mu *= MU_UP
                if mu > MAX_MU:
                    break
            ok = gn is not None
        if ok:
            gn_norm, g_norm = np.linalg.norm(gn), np.linalg.norm(ghat)
            if gn_norm <= radius:
                dl, dl_norm = gn, gn_norm
            elif cauchy_alpha * g_norm >= radius:
                dl, dl_norm = -(radius / g_norm) * ghat, radius
            else:
                b_dot_a = -cauchy_alpha * (ghat @ gn)
                a2 = (cauchy_alpha * g_norm) ** 2
                bma2 = a2 - 2.0 * b_dot_a + gn_norm ** 2
                c = b_dot_a - a2
                dd = np.sqrt(c * c + bma2 * (radius ** 2 - a2))
                beta = (dd - c) / bma2 if c <= 0 else (radius ** 2 - a2) / (dd + c)
                dl, dl_norm = (-cauchy_alpha * (1.0 - beta)) * ghat + beta * gn, radius
            step_t = dl / d                      # Jacobi-scaled variables
            model_change = -(gt @ step_t + 0.5 * step_t @ Ht @ step_t)
            ok = np.all(np.isfinite(step_t)) and model_change > 0.0
        if not ok:                               # DoglegStrategy::StepIsInvalid: mu goes up, nothing is re-used, the radius stays
            invalid += 1
            if invalid >= max_invalid:
                out["termination"] = 5
                break
            mu *= MU_UP
            reuse = False
            continue
        invalid = 0
        delta = step_t * scale
        x_new = _oplus_blocks(x, blocks, delta)
        xa, xb = _ambient(x, blocks), _ambient(x_new, blocks)
        if np.linalg.norm(xb - xa) <= parameter_tolerance * (np.linalg.norm(xa) + parameter_tolerance):
            out["termination"] = 3
            break
        win.set_state(*x_new)
        new_cost = win.cost()
        change = cost - new_cost
        if abs(change) < function_tolerance * cost:      # Ceres <= 1.10: returns WITHOUT taking the step
            win.set_state(*x)
            out["termination"] = 1
            break
        rho = change / model_change
        if rho > min_relative_decrease:
            x, cost = x_new, new_cost
            out["successful_steps"] += 1
            H, b0, blocks = win.full_system()
            g = -b0
            if rho < 0.25:
                radius *= 0.5
            if rho > 0.75:
                radius = min(max_radius, max(radius, 3.0 * dl_norm))
            mu = max(MIN_MU, 2.0 * mu / MU_UP)
            reuse = False
            if gradient_max_norm(x, g, blocks) <= gradient_tolerance:
                out["termination"] = 2
                break
        else:
            win.set_state(*x)
            radius *= 0.5
            reuse = True
        if radius < min_radius:
            out["termination"] = 4
            break
    out.update(final_cost=cost, final_radius=radius)
    return out
