"""GPU: the reference's integration test re-stated on the C++ host class (okvis_ceres/test/TestEstimator.cpp
:52-238): 2 equidistant cameras (baseline 0.1 m), a wall of landmarks at x = 3 m, constant-velocity motion,
IMU at 100 Hz, pixel noise U(-1,1), keypoint size 8, optimize(10,4,false) after every frame; final errors
||d speed&bias|| < 0.04, rotation < 1e-2, translation < 1e-1 (TestEstimator.cpp:229-236).
applyMarginalizationStrategy(2, 3) + a last optimize as in TestEstimator.cpp:207-213."""
import numpy as np
import pytest

from okvis_amd import estimator, synthetic
from okvis_amd.window import DIST_EQUIDISTANT, ImuParams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [0, 1, 2, 3])
def test_estimator_constant_velocity(c):
    rng = np.random.default_rng(100 + c)
    DURATION, IMU_RATE = 10.0, 100.0
    DT = 1.0 / IMU_RATE
    prm = ImuParams(sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, g=9.81,
                    g_max=1000.0, a_max=1000.0)
    speed = np.array([0.0, 1.0, 0.0])
    n_imu = int(DURATION * IMU_RATE) + 1
    t_imu = (np.arange(n_imu) * int(round(DT * 1e9))).astype(np.int64) + 1_000_000_000
    gyr = rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_g_c * np.sqrt(DT)
    acc = np.array([0, 0, prm.g]) + rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_a_c * np.sqrt(DT)
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0, 0.1, 0, 0, 0, 0, 1.0]])
    intr = np.stack([synthetic.TEST_INTR_EQUI, synthetic.TEST_INTR_EQUI])
    est = estimator.Estimator(0)
    sig = (1.0e-3 * (c % 2), 1.0e-4 * (c % 2), 1e-8 * (c // 2), 1e-7 * (c // 2))
    est.addCamera(*sig)
    est.addCamera(*sig)
    est.addImu(estimator.imu_param_vector(prm))
    # landmark wall at x = 3 m, the reference's 43 x 41 grid (TestEstimator.cpp:138-144)
    pts, ids = [], []
    nid = 1000
    for y in np.arange(-10.0, DURATION * 0.1 + 10.0 + 1e-9, 0.5):
        for z in np.arange(-10.0, 10.0 + 1e-9, 0.5):
            pts.append([3.0, y, z, 1.0]); ids.append(nid); nid += 1
            assert est.addLandmark(ids[-1], pts[-1])
    assert not est.addLandmark(ids[0], pts[0])          # duplicate id -> false (Map.cpp:297-299)
    pts = np.array(pts)
    assert len(ids) == 1763
    # per-frame extrinsics blocks (c >= 2) grow the reduced system by 12 per frame: fewer frames there
    K = 6 if c < 2 else 4
    frames = []
    last_id = None
    for k in range(K + 1):
        t_k = 1_000_000_000 + int(round(k * DURATION / K * 1e9))
        r_k = speed * k * DURATION / K
        f = estimator.Frame(10 + k, t_k, T_SC, intr, [DIST_EQUIDISTANT] * 2)
        n_obs = 0
        cams = []
        for i in range(2):
            # identity rotations as in the reference: the fisheye camera looks along +z_W, the wall is to
            # its side; points behind / outside the image are rejected by the projection status
            p_C = pts[:, :3] - r_k - T_SC[i, :3]
            uv, ok = synthetic.project_points(intr[i], DIST_EQUIDISTANT, p_C)
            cams.append((uv, ok))
        frames.append(f)
        assert est.addStates(f, t_imu, gyr, acc, k % 3 == 0)
        last_id = f.id
        for j in range(len(ids)):
            for i in range(2):
                uv, ok = cams[i]
                if ok[j]:
                    m = uv[j] + rng.uniform(-1, 1, 2)
                    kp = f.add_keypoint(i, m[0], m[1], 8.0)
                    assert est.addObservation(ids[j], f.id, i, kp) != 0
                    n_obs += 1
        assert n_obs > 20
        s = est.optimize(10, 4, False)
        assert s["final_cost"] <= s["initial_cost"] * (1 + 1e-9)
    # a duplicate observation returns NULL/0 (implementation/Estimator.hpp:52-56)
    # try out the marginalization strategy, then the last optimization (TestEstimator.cpp:207-213)
    removed = []
    n_before, l_before = est.numFrames(), est.numLandmarks()
    assert est.applyMarginalizationStrategy(2, 3, removed)
    older_kf = sum(1 for k in range(K + 1 - 3) if k % 3 == 0)
    assert est.numFrames() == 3 + min(2, older_kf)        # newest 3 + up to 2 keyframes kept among the older ones
    assert est.numFrames() < n_before and est.numLandmarks() == l_before - len(removed)
    dim, nb = est.priorInfo()
    assert dim > 0 and nb >= 3                            # kept keyframe poses + the oldest IMU-window pose/speed-bias
    s = est.optimize(10, 4, False)
    assert s["final_cost"] <= s["initial_cost"] * (1 + 1e-9)
    T = est.get_T_WS(last_id)
    sb = est.getSpeedAndBias(last_id)
    r_true = speed * DURATION
    assert np.linalg.norm(sb - np.r_[speed, np.zeros(6)]) < 0.04
    assert 2 * np.linalg.norm(T[3:6]) < 1e-2
    assert np.linalg.norm(T[:3] - r_true) < 1e-1
    est.close()


def test_sliding_window_with_marginalization_every_frame():
    """What ThreadedKFVio does per frame (ThreadedKFVio.cpp:736-765): optimize, then
    applyMarginalizationStrategy(numKeyframes=5, numImuFrames=3) — 20 frames, so the prior is re-linearised
    and re-marginalised many times (chained H_/b0_, first-estimate linearisation points kept per block)."""
    rng = np.random.default_rng(7)
    IMU_RATE, N_FRAMES, FRAME_DT = 100.0, 20, 0.5
    DT = 1.0 / IMU_RATE
    DURATION = N_FRAMES * FRAME_DT
    prm = ImuParams(sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, g=9.81,
                    g_max=1000.0, a_max=1000.0)
    speed = np.array([0.0, 1.0, 0.0])
    n_imu = int(DURATION * IMU_RATE) + 2
    t_imu = (np.arange(n_imu) * int(round(DT * 1e9))).astype(np.int64) + 1_000_000_000
    gyr = rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_g_c * np.sqrt(DT)
    acc = np.array([0, 0, prm.g]) + rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_a_c * np.sqrt(DT)
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0, 0.1, 0, 0, 0, 0, 1.0]])
    intr = np.stack([synthetic.TEST_INTR_EQUI, synthetic.TEST_INTR_EQUI])
    est = estimator.Estimator(0)
    est.addCamera(0, 0, 0, 0)
    est.addCamera(0, 0, 0, 0)
    est.addImu(estimator.imu_param_vector(prm))
    pts = np.array([[3.0, y, z, 1.0] for y in np.arange(-6.0, DURATION + 6.0, 0.75) for z in np.arange(-6.0, 6.0 + 1e-9, 0.75)])
    ids = 5000 + np.arange(len(pts))
    added, all_removed = set(), []
    frames = []
    prev_t = None
    for k in range(N_FRAMES):
        t_k = 1_000_000_000 + int(round(k * FRAME_DT * 1e9))
        r_k = speed * k * FRAME_DT
        f = estimator.Frame(100 + k, t_k, T_SC, intr, [DIST_EQUIDISTANT] * 2)
        frames.append(f)
        lo = np.searchsorted(t_imu, (prev_t if k else t_k) - 20_000_000)
        hi = np.searchsorted(t_imu, t_k + 20_000_000) + 1
        assert est.addStates(f, t_imu[lo:hi], gyr[lo:hi], acc[lo:hi], k % 3 == 0)
        prev_t = t_k
        n_obs = 0
        for i in range(2):
            p_C = pts[:, :3] - r_k - T_SC[i, :3]
            uv, ok = synthetic.project_points(intr[i], DIST_EQUIDISTANT, p_C)
            near = ok & (np.abs(pts[:, 1] - r_k[1]) < 5.0)
            for j in np.flatnonzero(near):
                lid = int(ids[j])
                if lid in all_removed:
                    continue
                if lid not in added:
                    assert est.addLandmark(lid, pts[j] + np.r_[rng.normal(size=3) * 0.05, 0])
                    added.add(lid)
                m = uv[j] + rng.uniform(-1, 1, 2)
                kp = f.add_keypoint(i, m[0], m[1], 8.0)
                assert est.addObservation(lid, f.id, i, kp) != 0
                n_obs += 1
        assert n_obs > 50
        s = est.optimize(5, 2, False)
        assert np.isfinite(s["final_cost"]) and s["final_cost"] <= s["initial_cost"] * (1 + 1e-9)
        removed = []
        assert est.applyMarginalizationStrategy(5, 3, removed)
        all_removed += removed
        assert est.numFrames() <= 8
        if k >= 3:
            dim, nb = est.priorInfo()
            assert dim >= 6 + 9 and nb >= 2
        # the newest three frames keep their speed/bias block, older ones do not (Estimator.cpp:485-554)
        for age in range(est.numFrames()):
            assert est.isInImuWindow(est.frameIdByAge(age)) == (age < 3)
    assert len(all_removed) > 0 and len(set(all_removed)) == len(all_removed)
    for lid in all_removed[:5]:
        with pytest.raises(estimator.EstimatorError):
            est.getLandmark(lid)
    T = est.get_T_WS(frames[-1].id)
    sb = est.getSpeedAndBias(frames[-1].id)
    r_true = speed * (N_FRAMES - 1) * FRAME_DT
    assert np.linalg.norm(sb - np.r_[speed, np.zeros(6)]) < 0.04
    assert 2 * np.linalg.norm(T[3:6]) < 1e-2
    assert np.linalg.norm(T[:3] - r_true) < 1e-1
    est.close()


def test_failed_marginalization_is_rolled_back():
    """applyMarginalizationStrategy interleaves decisions with deletions (Estimator.cpp:485-725); when its numerics fail the
    book-keeping is rolled back (strong guarantee).  A run in which the call fails at three frames (injected) and is then
    repeated must be indistinguishable from a run without failures: same windows, same removed landmarks, same states."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import estimator_scenarios as S
    kw = dict(n_frames=16, num_keyframes=3, num_imu_frames=2, iters=4, seed=11)
    clean, _ = S.sliding_window(lambda: estimator.Estimator(0), estimator.Frame, **kw)
    rough, _ = S.sliding_window(lambda: estimator.Estimator(0), estimator.Frame, fail_marginalization_at=(4, 9, 13, 14, 15), **kw)
    assert all(clean[k]["n_frames"] < k + 1 for k in (4, 9, 13, 14, 15)), "no frame leaves the window at the failing frames"
    assert any(clean[k]["removed"] for k in (13, 14, 15)), "no landmark is marginalised at the failing frames"
    for a, b in zip(clean, rough):
        assert (a["removed"], a["n_frames"], a["n_landmarks"], a["prior"]) == (b["removed"], b["n_frames"], b["n_landmarks"], b["prior"])
        for fid in a["poses"]:
            assert np.array_equal(a["poses"][fid], b["poses"][fid])
        for fid in a["sbs"]:
            assert np.array_equal(a["sbs"][fid], b["sbs"][fid])
        for lid in a["landmarks"]:
            assert np.array_equal(a["landmarks"][lid], b["landmarks"][lid])


def test_a_marginalisation_that_fails_late_drops_the_prior():
    """applyMarginalizationStrategy enqueues the marginalisation and returns; its numbers are waited for where they are read next
    (okvis_ba_marginalize_begin / _end).  When they never arrive (injected at two frames) the waiting call throws once, the prior
    is dropped — blocks without numbers must not reach a window — and the estimator carries on: every later frame optimises, the
    next marginalisation builds a new prior."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import estimator_scenarios as S
    kw = dict(n_frames=14, num_keyframes=3, num_imu_frames=2, iters=4, seed=11)
    clean, _ = S.sliding_window(lambda: estimator.Estimator(0), estimator.Frame, **kw)
    rough, _ = S.sliding_window(lambda: estimator.Estimator(0), estimator.Frame, fail_pending_marginalization_at=(6, 10), **kw)
    assert clean[5]["prior"][0] > 0 and clean[9]["prior"][0] > 0, "no prior at the frames whose numbers fail to arrive"
    for k, (a, b) in enumerate(zip(clean, rough)):
        assert (a["removed"], a["n_frames"], a["n_landmarks"]) == (b["removed"], b["n_frames"], b["n_landmarks"])   # the book-keeping is the same
        assert b["summary"]["iterations"] > 0 and np.isfinite(b["summary"]["final_cost"])
        if k < 6:
            assert a["prior"] == b["prior"]
            for fid in a["poses"]:
                assert np.array_equal(a["poses"][fid], b["poses"][fid])
    assert rough[-1]["prior"][0] > 0   # a new prior has been built since


@pytest.mark.parametrize("sigmas", [(0, 0, 0, 0), (0.01, 0.01, 1e-3, 1e-3)])
def test_patched_windows_iterate_like_flattened_ones(sigmas):
    """optimize() patches the window the solver holds with the edits since the last call; a second estimator flattens and uploads
    every frame (setUsePatch(False), the round-3 route).  The two hold the same window landmark for landmark — the blocks that
    stay carry the device's values, which are the values the other one uploads — so every state of every frame agrees bit for
    bit.  With relative extrinsics noise every frame has extrinsics blocks and relative-pose terms of its own."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import estimator_scenarios as S
    made = {}

    def make(patch):
        def f():
            e = estimator.Estimator(0)
            e.setUsePatch(patch)
            made[patch] = e
            return e
        return f

    routes = {}
    orig = estimator.Estimator.optimize

    def traced(self, *a, **k):
        r = orig(self, *a, **k)
        routes.setdefault(id(self), []).append(self.lastOptimizeWasPatch())
        return r

    estimator.Estimator.optimize = traced
    try:
        kw = dict(n_frames=14, num_keyframes=3, num_imu_frames=2, iters=5, seed=23, extrinsics_sigmas=sigmas)
        a, _ = S.sliding_window(make(True), estimator.Frame, **kw)
        b, _ = S.sliding_window(make(False), estimator.Frame, **kw)
    finally:
        estimator.Estimator.optimize = orig
    ra, rb = routes[id(made[True])], routes[id(made[False])]
    assert ra[0] is False and all(ra[1:]) and not any(rb), (ra, rb)
    assert any(x["removed"] for x in a)
    for x, y in zip(a, b):
        assert (x["removed"], x["n_frames"], x["n_landmarks"], x["prior"]) == (y["removed"], y["n_frames"], y["n_landmarks"], y["prior"])
        assert x["summary"] == y["summary"], (x["summary"], y["summary"])
        for fid in x["poses"]:
            assert np.array_equal(x["poses"][fid], y["poses"][fid])
        for fid in x["sbs"]:
            assert np.array_equal(x["sbs"][fid], y["sbs"][fid])
        for lid in x["landmarks"]:
            assert np.array_equal(x["landmarks"][lid], y["landmarks"][lid])


def test_add_states_says_why_it_refuses():
    """okvis::Estimator::addStates logs a reason and returns false (Estimator.cpp:121-163); the drop-in returns false and keeps the
    reason for okvis_est_last_error."""
    prm = ImuParams()
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0, 0.1, 0, 0, 0, 0, 1.0]])
    intr = np.stack([synthetic.TEST_INTR_EQUI, synthetic.TEST_INTR_EQUI])
    t = (np.arange(60) * 10_000_000).astype(np.int64) + 1_000_000_000
    gyr = np.zeros((60, 3)); acc = np.tile([0.0, 0.0, prm.g], (60, 1))
    est = estimator.Estimator(0)
    f0 = estimator.Frame(7, int(t[5]), T_SC, intr, [DIST_EQUIDISTANT] * 2)
    assert not est.addStates(f0, t[:10], gyr[:10], acc[:10], True) and "IMU parameters" in est.last_error()
    est.addCamera(0, 0, 0, 0); est.addCamera(0, 0, 0, 0)
    est.addImu(estimator.imu_param_vector(prm))
    assert not est.addStates(f0, t[:0], gyr[:0], acc[:0], True) and "initPoseFromImu" in est.last_error()
    assert est.addStates(f0, t[:10], gyr[:10], acc[:10], True)
    f1 = estimator.Frame(8, int(t[40]), T_SC, intr, [DIST_EQUIDISTANT] * 2)
    assert not est.addStates(f1, t[:20], gyr[:20], acc[:20], False)          # measurements end before the frame
    assert "propagation used -1 of 20" in est.last_error()
    again = estimator.Frame(7, int(t[40]), T_SC, intr, [DIST_EQUIDISTANT] * 2)
    assert not est.addStates(again, t[:50], gyr[:50], acc[:50], False) and "was used before" in est.last_error()
    older = estimator.Frame(3, int(t[40]), T_SC, intr, [DIST_EQUIDISTANT] * 2)
    assert not est.addStates(older, t[:50], gyr[:50], acc[:50], False) and "does not follow" in est.last_error()
    assert est.addStates(f1, t[:50], gyr[:50], acc[:50], False) and est.numFrames() == 2
    est.close()
