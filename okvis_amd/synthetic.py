"""Deterministic synthetic sliding windows of the BASELINE.json shapes.

The generator only *creates inputs*: a smooth truth trajectory, IMU samples, landmarks, noisy keypoint
measurements and a perturbed initial state, laid out as :class:`okvis_amd.window.Window`.  Window
construction mirrors what ``okvis::Estimator::addStates`` sets up for each new frame
(reference okvis_ceres/src/Estimator.cpp:110-343): first-pose prior with information
diag(1e8,1e8,1e8,0,0,1e8) (:240-243), speed/bias prior (:269-284), one ImuError per consecutive frame pair
(:288-307), extrinsics fixed / priored / chained by RelativePoseError (:199-214, :247-268, :310-336), and
observations with information 64/size^2 (implementation/Estimator.hpp:62-65) under CauchyLoss(1)
(Estimator.cpp:60).  Camera intrinsics/extrinsics and IMU noise are the EuRoC values of
reference config/config_fpga_p2_euroc.yaml:2-46.

Everything is seeded (numpy PCG64 Generator) — no libc rand like the reference tests.
"""
from __future__ import annotations

import numpy as np

from .window import (DIST_EQUIDISTANT, DIST_NONE, DIST_RADTAN, DIST_RADTAN8, ImuParams, Window)

# --- EuRoC calibration, reference config/config_fpga_p2_euroc.yaml:2-23 --------------------------------
EUROC_T_SC = [
    np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
              [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
              [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
              [0, 0, 0, 1.0]]),
    np.array([[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
              [0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024],
              [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038],
              [0, 0, 0, 1.0]]),
]
EUROC_INTR = np.array([
    [458.654880721, 457.296696463, 367.215803962, 248.37534061,
     -0.28340811217, 0.0739590738929, 0.000193595028569, 1.76187114545e-05, 0, 0, 0, 0],
    [457.587426604, 456.13442556, 379.99944652, 255.238185386,
     -0.283683654496, 0.0745128430929, -0.000104738949098, -3.55590700274e-05, 0, 0, 0, 0],
])
IMAGE_W, IMAGE_H = 752, 480
# PinholeCamera<D>::createTestObject (okvis_cv/include/okvis/cameras/PinholeCamera.hpp:287-297) with
# EquidistantDistortion::testObject (EquidistantDistortion.hpp:104-107): used by TestEstimator.cpp:113-116
TEST_INTR_EQUI = np.array([350.0, 360.0, 378.0, 238.0, -0.21, 0.14, 0.0006, 0.0003, 0, 0, 0, 0])
TEST_INTR_RADTAN = np.array([350.0, 360.0, 378.0, 238.0, -0.16, 0.15, 0.0003, 0.0002, 0, 0, 0, 0])


# ---------------------------------------------------------------------------------------------------
# small numpy geometry (x,y,z,w Hamilton quaternions, as in the reference README.md:23-25)
# ---------------------------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def qrot(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def delta_q(dalpha):
    h = 0.5 * np.linalg.norm(dalpha)
    s = 0.5 * (np.sin(h) / h if h > 1e-6 else 1.0 - h * h / 6.0)
    return np.array([s * dalpha[0], s * dalpha[1], s * dalpha[2], np.cos(h)])


def pose_oplus(pose, delta):
    """Left perturbation r+=dr, q = dq(dalpha) (x) q  (okvis Transformation::oplus)."""
    out = np.array(pose, dtype=np.float64)
    out[:3] += delta[:3]
    q = qmul(delta_q(delta[3:6]), out[3:7])
    out[3:7] = q / np.linalg.norm(q)
    return out


def T_to_pose(T):
    return np.concatenate([T[:3, 3], rot_to_quat(T[:3, :3])])


def sqrt_information_eigen_llt(info):
    """squareRootInformation = LLT(info).matrixL().transpose() as Eigen's unblocked LLT computes it,
    INCLUDING the early exit on a non-positive pivot (reference PoseError.cpp:70-76 applied to the
    rank-deficient information of Estimator.cpp:240-242)."""
    A = np.array(info, dtype=np.float64)
    n = A.shape[0]
    for k in range(n):
        x = A[k, k] - np.dot(A[k, :k], A[k, :k])
        if x <= 0.0:
            break
        x = np.sqrt(x)
        A[k, k] = x
        for i in range(k + 1, n):
            A[i, k] = (A[i, k] - np.dot(A[i, :k], A[k, :k])) / x
    return np.tril(A).T.copy()


def project_points(intr, model, p_C):
    """Vectorised forward pinhole projection with distortion (no Jacobians) — generator use only."""
    fu, fv, cu, cv = intr[:4]
    d = intr[4:12]
    z = p_C[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u0 = p_C[:, 0] / z
        u1 = p_C[:, 1] / z
    ok = z > 0.2
    if model == DIST_NONE:
        d0, d1 = u0, u1
    elif model == DIST_RADTAN:
        k1, k2, p1, p2 = d[:4]
        rho = u0 * u0 + u1 * u1
        rad = k1 * rho + k2 * rho * rho
        d0 = u0 + u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0)
        d1 = u1 + u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1)
        ok &= rho < 1.2  # stay in the monotonic region of the EuRoC radtan model
    elif model == DIST_EQUIDISTANT:
        k1, k2, k3, k4 = d[:4]
        r = np.sqrt(u0 * u0 + u1 * u1)
        th = np.arctan(r)
        th2 = th * th
        thd = th * (1 + k1 * th2 + k2 * th2 ** 2 + k3 * th2 ** 3 + k4 * th2 ** 4)
        sc = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
        d0, d1 = sc * u0, sc * u1
    elif model == DIST_RADTAN8:
        k1, k2, p1, p2, k3, k4, k5, k6 = d
        rho = u0 * u0 + u1 * u1
        rad = (1 + ((k3 * rho + k2) * rho + k1) * rho) / (1 + ((k6 * rho + k5) * rho + k4) * rho)
        d0 = u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0)
        d1 = u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1)
        ok &= rho < 9.0
    else:
        raise ValueError(model)
    uv = np.stack([fu * d0 + cu, fv * d1 + cv], 1)
    ok &= (uv[:, 0] >= 0) & (uv[:, 0] < IMAGE_W) & (uv[:, 1] >= 0) & (uv[:, 1] < IMAGE_H)
    return uv, ok


# ---------------------------------------------------------------------------------------------------
# truth trajectory
# ---------------------------------------------------------------------------------------------------
# body axes in the world at rest: x_S up, y_S = -y_W, z_S = +x_W (the EuRoC cameras look along +z_S)
_R_WS0 = np.array([[0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]])
_POS_A = np.array([0.35, 0.9, 0.15])
_POS_W = np.array([0.55, 0.45, 0.8])
_POS_P = np.array([0.3, -0.4, 1.1])
_ANG_A = np.array([0.25, 0.05, 0.04])   # yaw, pitch, roll amplitudes [rad]; yaw rate <= 0.3*... rad/s
_ANG_W = np.array([0.9, 0.7, 1.1])
_ANG_P = np.array([0.1, 0.5, -0.2])


def truth_at(t):
    """Return (p_W, v_W, a_W, R_WS, omega_S) of the analytic truth trajectory at time t [s]."""
    p = _POS_A * np.sin(_POS_W * t + _POS_P)
    v = _POS_A * _POS_W * np.cos(_POS_W * t + _POS_P)
    a = -_POS_A * _POS_W ** 2 * np.sin(_POS_W * t + _POS_P)
    ang = _ANG_A * np.sin(_ANG_W * t + _ANG_P)
    dang = _ANG_A * _ANG_W * np.cos(_ANG_W * t + _ANG_P)
    psi, th, ph = ang
    dpsi, dth, dph = dang
    cz, sz, cy, sy, cx, sx = np.cos(psi), np.sin(psi), np.cos(th), np.sin(th), np.cos(ph), np.sin(ph)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    R_dyn = Rz @ Ry @ Rx
    # body rates of a ZYX Euler rotation
    w_dyn = np.array([dph - dpsi * sy, dth * cx + dpsi * sx * cy, -dth * sx + dpsi * cx * cy])
    R_WS = R_dyn @ _R_WS0
    omega_S = _R_WS0.T @ w_dyn
    return p, v, a, R_WS, omega_S


# ---------------------------------------------------------------------------------------------------
# window generator
# ---------------------------------------------------------------------------------------------------
def make_window(num_keyframes=10, num_landmarks=400, visibility=1.0, seed=20240923,
                imu_rate_hz=200, frame_dt=0.5, estimate_extrinsics="fixed", pixel_noise=1.0,
                keypoint_size=8.0, cam_model=DIST_RADTAN, pose_noise=(0.05, np.deg2rad(0.5)),
                landmark_noise=0.10, speed_noise=0.02, with_imu=True, frame_offset_s=0.0017):
    """Build one synthetic window.

    estimate_extrinsics: "fixed"    sigma_* = 0 (EuRoC config): shared constant T_SC blocks
                         "shared"   sigma_absolute > 0, sigma_c_relative = 0: one estimated block per camera
                                    with a PoseError prior (TestEstimator.cpp case c=1)
                         "perframe" sigma_absolute, sigma_c_relative > 0: one block per frame and camera,
                                    chained by RelativePoseError (TestEstimator.cpp case c=3)
    """
    rng = np.random.default_rng(seed)
    K, L, NC = int(num_keyframes), int(num_landmarks), 2
    prm = ImuParams()
    if cam_model == DIST_RADTAN:
        intr = EUROC_INTR.copy()
    elif cam_model == DIST_EQUIDISTANT:
        intr = np.stack([TEST_INTR_EQUI, TEST_INTR_EQUI])
    elif cam_model == DIST_NONE:
        intr = EUROC_INTR.copy()
        intr[:, 4:] = 0
    elif cam_model == DIST_RADTAN8:
        intr = EUROC_INTR.copy()
        intr[:, 8:12] = [[0.01, 0.002, -0.001, 0.0005], [0.012, 0.0015, -0.0008, 0.0004]]
    else:
        raise ValueError(cam_model)
    T_SC_true = [T.copy() for T in EUROC_T_SC]

    # ---- frame times and truth states ----
    t_frame = frame_offset_s + frame_dt * np.arange(K)
    t_frame_ns = np.round(t_frame * 1e9).astype(np.int64)
    t_frame = t_frame_ns * 1e-9
    pose_true = np.zeros((K, 7))
    sb_true = np.zeros((K, 9))
    R_true = []
    for k in range(K):
        p, v, a, R, w = truth_at(t_frame[k])
        pose_true[k, :3] = p
        pose_true[k, 3:] = rot_to_quat(R)
        sb_true[k, :3] = v
        R_true.append(R)

    # ---- IMU samples (truth + white noise at the config densities, zero true bias) ----
    dt_imu = 1.0 / imu_rate_hz
    n_samp = int(np.ceil((t_frame[-1] + 2 * dt_imu) / dt_imu)) + 2
    s_t_ns = (np.arange(n_samp, dtype=np.int64) * int(round(dt_imu * 1e9)))
    gyr = np.zeros((n_samp, 3))
    acc = np.zeros((n_samp, 3))
    g_W = np.array([0.0, 0.0, prm.g])
    for j in range(n_samp):
        p, v, a, R, w = truth_at(s_t_ns[j] * 1e-9)
        gyr[j] = w
        acc[j] = R.T @ (a + g_W)
    gyr += rng.standard_normal((n_samp, 3)) * prm.sigma_g_c / np.sqrt(dt_imu)
    acc += rng.standard_normal((n_samp, 3)) * prm.sigma_a_c / np.sqrt(dt_imu)

    # ---- landmarks: rejection-sample points 3-15 m in front of the trajectory seen by every view ----
    def views(points_W):
        vis = np.zeros((points_W.shape[0], K, NC), dtype=bool)
        uvs = np.zeros((points_W.shape[0], K, NC, 2))
        for k in range(K):
            p_S = (points_W - pose_true[k, :3]) @ R_true[k]          # R^T (p - r)
            for c in range(NC):
                Rc, tc = T_SC_true[c][:3, :3], T_SC_true[c][:3, 3]
                p_C = (p_S - tc) @ Rc
                uv, ok = project_points(intr[c], cam_model, p_C)
                vis[:, k, c] = ok
                uvs[:, k, c] = uv
        return vis, uvs

    lms = np.zeros((0, 3))
    while lms.shape[0] < L:
        n = 4 * L
        depth = rng.uniform(3.0, 15.0, n)
        lat = rng.uniform(-0.9, 0.9, n) * depth * 0.55
        ver = rng.uniform(-0.9, 0.9, n) * depth * 0.35
        cand = np.stack([depth, lat, ver], 1) + pose_true[K // 2, :3]   # world x is "forward"
        vis, _ = views(cand)
        keep = vis.all(axis=(1, 2)) if visibility >= 1.0 else (vis.sum(axis=(1, 2)) >= 2 * NC)
        lms = np.concatenate([lms, cand[keep]])[:L]
    vis, uvs = views(lms)
    if visibility < 1.0:
        drop = rng.uniform(size=vis.shape[:2]) > visibility        # drop whole (landmark, frame) views
        vis &= ~drop[:, :, None]
        # keep every landmark observed from at least two frames
        for l in range(L):
            if vis[l].any(axis=1).sum() < 2:
                ks = rng.choice(K, 2, replace=False)
                vis[l, ks] = views(lms[l:l + 1])[0][0, ks]

    # ---- pose blocks: K body poses, then extrinsics ----
    ext_nominal = [pose_oplus(T_to_pose(T), np.concatenate([rng.normal(0, 1e-3, 3), rng.normal(0, 1e-4, 3)]))
                   if estimate_extrinsics != "fixed" else T_to_pose(T) for T in T_SC_true]
    if estimate_extrinsics in ("fixed", "shared"):
        ext_blocks = np.array(ext_nominal)                          # [NC,7]
        ext_index = lambda k, c: K + c                              # noqa: E731
    elif estimate_extrinsics == "perframe":
        ext_blocks = np.array([ext_nominal[c] for k in range(K) for c in range(NC)])
        ext_index = lambda k, c: K + k * NC + c                     # noqa: E731
    else:
        raise ValueError(estimate_extrinsics)
    pose_init = np.array([pose_oplus(pose_true[k], np.concatenate([rng.normal(0, pose_noise[0], 3),
                                                                  rng.normal(0, pose_noise[1], 3)]))
                          for k in range(K)])
    pose = np.concatenate([pose_init, ext_blocks])
    pose_fixed = np.zeros(pose.shape[0], np.uint8)
    if estimate_extrinsics == "fixed":
        pose_fixed[K:] = 1                                          # Estimator.cpp:264-267
    sb = sb_true.copy()
    sb[:, :3] += rng.normal(0, speed_noise, (K, 3))
    lm = np.concatenate([lms + rng.normal(0, landmark_noise, lms.shape), np.ones((L, 1))], 1)

    # ---- observations ----
    l_idx, k_idx, c_idx = np.nonzero(vis)
    meas = uvs[l_idx, k_idx, c_idx] + rng.standard_normal((l_idx.size, 2)) * pixel_noise
    meas = meas.astype(np.float32).astype(np.float64)               # cv::KeyPoint stores floats
    obs_ext = np.array([ext_index(k, c) for k, c in zip(k_idx, c_idx)], np.int32)
    w = Window(
        pose=pose, pose_fixed=pose_fixed, sb=sb, sb_fixed=np.zeros(K, np.uint8), lm=lm,
        cam_intr=intr, cam_model=np.full(NC, cam_model, np.int32),
        obs_lm=l_idx.astype(np.int32), obs_pose=k_idx.astype(np.int32), obs_ext=obs_ext,
        obs_cam=c_idx.astype(np.int32), obs_uv=meas,
        obs_sqrtw=np.full(l_idx.size, 8.0 / keypoint_size), cauchy_b=1.0, imu_params=prm)
    w.sort_observations()

    # ---- IMU factors ----
    if with_imu and K > 1:
        b, cnt = [], []
        for k in range(1, K):
            j0 = int(np.searchsorted(s_t_ns, t_frame_ns[k - 1], side="right")) - 1   # last sample <= t0
            j1 = int(np.searchsorted(s_t_ns, t_frame_ns[k], side="left"))            # first sample >= t1
            j0 = max(j0 - 1, 0)                                     # one sample of margin like the frontend
            j1 = min(j1 + 1, n_samp - 1)
            b.append(j0)
            cnt.append(j1 - j0 + 1)
        w.imu_pose0 = np.arange(0, K - 1, dtype=np.int32)
        w.imu_sb0 = np.arange(0, K - 1, dtype=np.int32)
        w.imu_pose1 = np.arange(1, K, dtype=np.int32)
        w.imu_sb1 = np.arange(1, K, dtype=np.int32)
        w.imu_t0 = t_frame_ns[:-1].copy()
        w.imu_t1 = t_frame_ns[1:].copy()
        w.imu_s_begin = np.array(b, np.int32)
        w.imu_s_count = np.array(cnt, np.int32)
        w.imu_s_t, w.imu_s_gyr, w.imu_s_acc = s_t_ns, gyr, acc

    # ---- priors (Estimator.cpp:238-285) ----
    info0 = np.diag([1e8, 1e8, 1e8, 0.0, 0.0, 1e8])
    pp_pose, pp_meas, pp_si = [0], [pose[0].copy()], [sqrt_information_eigen_llt(info0).reshape(-1)]
    if estimate_extrinsics != "fixed":
        s_abs_t, s_abs_r = 1.0e-3, 1.0e-4                           # TestEstimator.cpp:107-110
        for c in range(NC):
            pp_pose.append(ext_index(0, c))
            pp_meas.append(pose[ext_index(0, c)].copy())
            pp_si.append(sqrt_information_eigen_llt(
                np.diag([1 / s_abs_t ** 2] * 3 + [1 / s_abs_r ** 2] * 3)).reshape(-1))
    w.pprior_pose = np.array(pp_pose, np.int32)
    w.pprior_meas = np.array(pp_meas)
    w.pprior_sqrtinfo = np.array(pp_si)
    w.sbprior_sb = np.array([0], np.int32)
    w.sbprior_meas = sb[0:1].copy()
    w.sbprior_sqrtinfo = sqrt_information_eigen_llt(
        np.diag([1.0] * 3 + [1 / prm.sigma_bg ** 2] * 3 + [1 / prm.sigma_ba ** 2] * 3)).reshape(1, -1)
    if estimate_extrinsics == "perframe":
        s_rel_t, s_rel_r = 1e-8, 1e-7                               # TestEstimator.cpp:111-114
        r0, r1, si = [], [], []
        for k in range(1, K):
            dt = (t_frame_ns[k] - t_frame_ns[k - 1]) * 1e-9
            for c in range(NC):
                r0.append(ext_index(k - 1, c))
                r1.append(ext_index(k, c))
                si.append(sqrt_information_eigen_llt(
                    np.diag([1 / (s_rel_t ** 2 * dt)] * 3 + [1 / (s_rel_r ** 2 * dt)] * 3)).reshape(-1))
        w.rel_pose0, w.rel_pose1, w.rel_sqrtinfo = np.array(r0, np.int32), np.array(r1, np.int32), np.array(si)
    w.meta = dict(pose_true=pose_true, sb_true=sb_true, lm_true=lms, K=K, L=L, seed=seed,
                  visibility=visibility, extrinsics=estimate_extrinsics, t_frame_ns=t_frame_ns)
    w.validate()
    return w


# the BASELINE.json configs
def config_A(seed=20240923, visibility=1.0, **kw):
    """configs[1]: 10 keyframes / 2 cams / 400 landmarks / 100-sample IMU factors."""
    return make_window(10, 400, visibility, seed, **kw)


def config_C(seed=20240923, visibility=1.0, **kw):
    """configs[2]: 50 keyframes / 2000 landmarks (large Schur reduce)."""
    return make_window(50, 2000, visibility, seed, frame_dt=kw.pop("frame_dt", 0.1), **kw)


def config_batch(n_windows=64, seed=20240923, **kw):
    """configs[3]: independent copies of A with seeds seed+i."""
    return [config_A(seed + i, **kw) for i in range(n_windows)]


def small_window(seed=1, K=4, L=40, **kw):
    """A small case the CPU oracle finishes in milliseconds (parity tests)."""
    return make_window(K, L, kw.pop("visibility", 0.7), seed, **kw)
