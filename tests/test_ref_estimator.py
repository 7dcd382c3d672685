"""The reference's OWN okvis::Estimator (okvis_ceres/src/Estimator.cpp compiled unmodified into oracle/_ref, with Map,
MarginalizationError, the error terms, MultiFrame / NCameraSystem; ::ceres::Solve = oracle/ref/ceres_shim_solve.cpp) driven
through the call sequence of its integration test (okvis_ceres/test/TestEstimator.cpp:52-238) and of the per-frame loop of
ThreadedKFVio (ThreadedKFVio.cpp:736-765) — on the CPU.  It shows that the stand-in build is a working Estimator (the pin the
GPU comparison in tests/test_gpu_estimator_vs_reference.py relies on) and records the reference's own behaviour of
applyMarginalizationStrategy (which frames / landmarks leave, size of the prior)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_lib as R  # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built and /root/reference absent")

import estimator_scenarios as S  # noqa: E402


def test_reference_estimator_sliding_window():
    trace, truth = S.sliding_window(R.RefEstimator, R.RefFrame, n_frames=16, num_keyframes=3)
    last = trace[-1]
    # TestEstimator.cpp:229-236 tolerances on the newest state
    T, sb = last["poses"][truth["last_id"]], last["sbs"][truth["last_id"]]
    assert np.linalg.norm(sb - np.r_[truth["speed"], np.zeros(6)]) < 0.04
    assert 2 * np.linalg.norm(T[3:6]) < 1e-2
    assert np.linalg.norm(T[:3] - truth["r_last"]) < 1e-1
    # window management (Estimator.cpp:434-773): at most numKeyframes + numImuFrames frames, the newest three keep their
    # speed/bias block, a prior exists once frames were marginalised, removed landmarks never come back
    removed_all = sum((r["removed"] for r in trace), [])
    assert len(removed_all) == len(set(removed_all)) > 0
    for r in trace:
        assert r["n_frames"] <= 6
        assert np.isfinite(r["summary"]["final_cost"]) and r["summary"]["final_cost"] <= r["summary"]["initial_cost"] * (1 + 1e-9)
        ages = list(r["poses"].keys())
        assert [r["in_imu"][f] for f in ages] == [i < 3 for i in range(len(ages))]
    assert trace[-1]["prior"][0] >= 6 + 9
