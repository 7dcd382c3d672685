"""Deterministic call sequences against an `Estimator` (okvis_amd.estimator surface), shared by the CPU test that runs them
on the reference's own okvis::Estimator (tests/ref_lib.RefEstimator) and the GPU test that runs them on both back-ends
and compares the trace.  Every random number is drawn up front, so two back-ends see bit-identical inputs."""
import numpy as np

from okvis_amd import estimator as E, synthetic
from okvis_amd.window import DIST_EQUIDISTANT, DIST_RADTAN, ImuParams


def sliding_window(make_estimator, make_frame, n_frames=12, num_keyframes=5, num_imu_frames=3, iters=5, seed=7,
                   model=DIST_EQUIDISTANT, extrinsics_sigmas=(0, 0, 0, 0), marginalize=True, fail_marginalization_at=(),
                   fail_pending_marginalization_at=()):
    """What ThreadedKFVio does per frame (ThreadedKFVio.cpp:736-765): addStates, addLandmark / addObservation for the
    visible wall points, optimize(iters), applyMarginalizationStrategy(numKeyframes, numImuFrames).  Returns a trace:
    one dict per frame with the states of every frame in the window, a sample of landmarks, counts and removed ids."""
    rng = np.random.default_rng(seed)
    IMU_RATE, FRAME_DT = 100.0, 0.5
    DT = 1.0 / IMU_RATE
    DURATION = n_frames * FRAME_DT
    prm = ImuParams(sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, g=9.81, g_max=1000.0,
                    a_max=1000.0)
    speed = np.array([0.0, 1.0, 0.0])
    n_imu = int(DURATION * IMU_RATE) + 2
    t_imu = (np.arange(n_imu) * int(round(DT * 1e9))).astype(np.int64) + 1_000_000_000
    gyr = rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_g_c * np.sqrt(DT)
    acc = np.array([0, 0, prm.g]) + rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_a_c * np.sqrt(DT)
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0, 0.1, 0, 0, 0, 0, 1.0]])
    k_intr = synthetic.TEST_INTR_EQUI if model == DIST_EQUIDISTANT else synthetic.TEST_INTR_RADTAN
    intr = np.stack([k_intr, k_intr])
    pts = np.array([[3.0, y, z, 1.0] for y in np.arange(-6.0, DURATION + 6.0, 0.75) for z in np.arange(-6.0, 6.0 + 1e-9, 0.75)])
    # ids far away from anything the reference's process-wide IdProvider (IdProvider.cpp:46-52, a counter from 1) hands out for its
    # speed/bias and extrinsics blocks: a frame id it has already used makes okvis::Estimator::addStates refuse the frame
    ids = 5_000_000 + np.arange(len(pts))
    lm_noise = rng.normal(size=(len(pts), 3)) * 0.05
    px_noise = rng.uniform(-1, 1, (n_frames, 2, len(pts), 2))

    est = make_estimator()
    est.addCamera(*extrinsics_sigmas)
    est.addCamera(*extrinsics_sigmas)
    est.addImu(E.imu_param_vector(prm))
    added, all_removed, frames, trace = set(), [], [], []
    prev_t = None
    for k in range(n_frames):
        t_k = 1_000_000_000 + int(round(k * FRAME_DT * 1e9))
        r_k = speed * k * FRAME_DT
        f = make_frame(1_000_000 + k, t_k, T_SC, intr, [model] * 2)
        frames.append(f)
        lo = np.searchsorted(t_imu, (prev_t if k else t_k) - 20_000_000)
        hi = np.searchsorted(t_imu, t_k + 20_000_000) + 1
        assert est.addStates(f, t_imu[lo:hi], gyr[lo:hi], acc[lo:hi], k % 3 == 0), (type(est).__name__, k, est.last_error() if hasattr(est, 'last_error') else None)
        prev_t = t_k
        n_obs = 0
        for i in range(2):
            p_C = pts[:, :3] - r_k - T_SC[i, :3]
            if model == DIST_EQUIDISTANT:
                uv, ok = synthetic.project_points(intr[i], model, p_C)
                near = ok & (np.abs(pts[:, 1] - r_k[1]) < 5.0)
            else:   # pinhole + radtan looks along +z: put the wall in front by swapping axes
                uv, ok = synthetic.project_points(intr[i], model, p_C[:, [1, 2, 0]] * np.array([1, 1, 1.0]))
                near = ok & (np.abs(pts[:, 1] - r_k[1]) < 2.5)
            for j in np.flatnonzero(near):
                lid = int(ids[j])
                if lid in all_removed:
                    continue
                if lid not in added:
                    assert est.addLandmark(lid, pts[j] + np.r_[lm_noise[j], 0])
                    added.add(lid)
                m = uv[j] + px_noise[k, i, j]
                kp = f.add_keypoint(i, m[0], m[1], 8.0)
                assert est.addObservation(lid, f.id, i, kp) != 0
                n_obs += 1
        if k in fail_pending_marginalization_at:
            # the numbers of the marginalisation the previous frame enqueued never arrive (injected): the call that waits for them
            # throws once, the prior is dropped, and the estimator goes on
            est.debugFailPendingMarginalization()
            late = False
            try:
                est.optimize(iters, 2, False)
            except Exception:
                late = True
            assert late and est.priorInfo()[0] == 0
        s = est.optimize(iters, 2, False)
        removed = []
        if marginalize and k in fail_marginalization_at:
            # the numerics of this call fail (injected): the estimator has to be exactly where it was before the call
            before = (est.numFrames(), est.numLandmarks(), est.priorInfo())
            est.debugFailNextMarginalization()
            failed = False
            try:
                est.applyMarginalizationStrategy(num_keyframes, num_imu_frames, removed)
            except Exception:
                failed = True
            assert failed and removed == [] and before == (est.numFrames(), est.numLandmarks(), est.priorInfo())
        if marginalize:
            assert est.applyMarginalizationStrategy(num_keyframes, num_imu_frames, removed)
        all_removed += removed
        rec = dict(frame=k, n_obs=n_obs, summary=s, removed=sorted(removed), n_frames=est.numFrames(),
                   n_landmarks=est.numLandmarks(), prior=est.priorInfo(), poses={}, sbs={}, keyframe={}, in_imu={},
                   landmarks={})
        for age in range(est.numFrames()):
            fid = est.frameIdByAge(age)
            rec["poses"][fid] = est.get_T_WS(fid)
            rec["keyframe"][fid] = est.isKeyframe(fid)
            rec["in_imu"][fid] = est.isInImuWindow(fid)
            if rec["in_imu"][fid]:
                rec["sbs"][fid] = est.getSpeedAndBias(fid)
        alive = sorted(added - set(all_removed))
        for lid in alive[::7]:
            rec["landmarks"][lid] = est.getLandmark(lid)[0]
        trace.append(rec)
    truth = dict(speed=speed, r_last=speed * (n_frames - 1) * FRAME_DT, last_id=frames[-1].id)
    est.close()
    return trace, truth
