"""CPU tests of the C++ host layer (okvis_amd::Estimator): the static helpers the reference keeps on the
CPU (ImuError::propagation, Estimator::initPoseFromImu) against the oracle, and the loud failure without
a GPU."""
import numpy as np
import pytest

from okvis_amd import estimator, synthetic


def test_propagation_matches_oracle(oracle):
    w = synthetic.make_window(3, 10, 1.0, 31)
    prm = estimator.imu_param_vector(w.imu_params)
    for f in range(w.n_imu):
        b, n = w.imu_s_begin[f], w.imu_s_count[f]
        t, g, a = w.imu_s_t[b:b + n], w.imu_s_gyr[b:b + n], w.imu_s_acc[b:b + n]
        T0, s0 = w.meta["pose_true"][f], w.meta["sb_true"][f] + np.r_[0, 0, 0, 1e-3, -2e-3, 1e-3, 0.01, 0.02, -0.01]
        T, s, k = estimator.propagation(t, g, a, prm, T0, s0, w.imu_t0[f], w.imu_t1[f])
        To, so, _, _, ko = oracle.imu_propagation(t, g, a, w.imu_params, T0, s0, w.imu_t0[f], w.imu_t1[f])
        assert k == ko and k >= 99
        assert np.abs(T - To).max() < 1e-13 and np.abs(s - so).max() < 1e-13
    # measurements not covering the interval -> -1 (ImuError.cpp:301-302)
    assert estimator.propagation(t[:20], g[:20], a[:20], prm, T0, s0, w.imu_t0[-1], w.imu_t1[-1])[2] == -1


def test_init_pose_from_imu_aligns_gravity():
    # Estimator.cpp:811-840: the measured mean acceleration is rotated onto +z_W
    rng = np.random.default_rng(0)
    for _ in range(20):
        R = synthetic.qrot(synthetic.delta_q(rng.uniform(-1, 1, 3)))
        acc = np.tile(R.T @ np.array([0, 0, 9.81]), (10, 1)) + rng.normal(0, 1e-3, (10, 3))
        T, ok = estimator.init_pose_from_imu(acc)
        assert ok and np.allclose(T[:3], 0)
        C_WS = synthetic.qrot(T[3:])
        assert np.allclose(C_WS @ acc.mean(0) / np.linalg.norm(acc.mean(0)), [0, 0, 1], atol=1e-9)
    assert estimator.init_pose_from_imu(np.zeros((0, 3)))[1] is False


def test_estimator_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(estimator.EstimatorError, match="no HIP device"):
        estimator.Estimator(0)
