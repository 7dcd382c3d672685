"""GPU: okvis_amd::Estimator (the MI355X backend behind the okvis::Estimator method set) against the reference's OWN
okvis::Estimator on identical call sequences.

The reference side is okvis_ceres/src/Estimator.cpp + Map.cpp + MarginalizationError.cpp + the error terms + MultiFrame /
NCameraSystem, compiled unmodified into oracle/_ref (stand-in Eigen / Ceres / glog / OpenCV headers; ::ceres::Solve =
oracle/ref/ceres_shim_solve.cpp, the DOGLEG policy stated a second time).  Both sides run what ThreadedKFVio does per frame
(ThreadedKFVio.cpp:736-765): addStates, addLandmark / addObservation, optimize(n), applyMarginalizationStrategy.  Compared
after EVERY frame: which frames are in the window, keyframe / IMU-window flags, which landmarks were removed, size and
block count of the marginalisation prior (all exact), and poses / speed-biases / landmarks (tolerances below).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_lib as R  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not available")]

import estimator_scenarios as S  # noqa: E402
from okvis_amd import estimator as E  # noqa: E402
from okvis_amd.window import DIST_EQUIDISTANT  # noqa: E402


def _compare(kw, settle=5, pos_tol=1e-5, rot_tol=1e-5, sb_tol=1e-4, lm_tol=2e-3, cost_tol=1e-6):
    """States are compared tightly from frame `settle` on.  The first windows of this scenario (TestEstimator.cpp's: one or
    two frames, 0.1 m stereo baseline, a wall 3 m away) are so weakly constrained that the iterates of two correct solvers
    drift apart at the 1e-3 level while the cost still falls (measured: frame 1 after 20 iterations, cost 42.786 vs 42.777);
    once the window holds 4-5 frames the minimum is sharp and both agree to 1e-7 ... 1e-8."""
    tr_r, _ = S.sliding_window(R.RefEstimator, R.RefFrame, **kw)
    tr_g, truth = S.sliding_window(lambda: E.Estimator(0), E.Frame, **kw)
    assert len(tr_r) == len(tr_g)
    worst = dict(pos=0.0, rot=0.0, sb=0.0, lm=0.0, cost=0.0)
    early = dict(pos=0.0, rot=0.0, sb=0.0, lm=0.0, cost=0.0)
    for a, b in zip(tr_r, tr_g):
        k = a["frame"]
        # ---- discrete behaviour: exact ----
        assert a["n_obs"] == b["n_obs"]
        assert (a["n_frames"], a["n_landmarks"]) == (b["n_frames"], b["n_landmarks"]), k
        assert a["removed"] == b["removed"], k
        # size of the prior: exact.  Number of blocks: the reference also lists the FIXED extrinsics blocks its marginalised
        # reprojection errors touch (minimal dimension 0, MarginalizationError.hpp ParameterBlockInfo), the backend only free ones
        assert a["prior"][0] == b["prior"][0], (k, a["prior"], b["prior"])
        assert b["prior"][1] <= a["prior"][1] <= b["prior"][1] + 2, (k, a["prior"], b["prior"])
        assert list(a["poses"].keys()) == list(b["poses"].keys()), k
        assert a["keyframe"] == b["keyframe"] and a["in_imu"] == b["in_imu"], k
        assert sorted(a["landmarks"].keys()) == sorted(b["landmarks"].keys())
        # ---- states ----
        w = worst if k >= settle else early
        for fid in a["poses"]:
            w["pos"] = max(w["pos"], np.abs(a["poses"][fid][:3] - b["poses"][fid][:3]).max())
            w["rot"] = max(w["rot"], np.abs(a["poses"][fid][3:] - b["poses"][fid][3:]).max())
        for fid in a["sbs"]:
            w["sb"] = max(w["sb"], np.abs(a["sbs"][fid] - b["sbs"][fid]).max())
        for lid in a["landmarks"]:
            w["lm"] = max(w["lm"], np.abs(a["landmarks"][lid] - b["landmarks"][lid]).max())
        ca, cb = a["summary"]["final_cost"], b["summary"]["final_cost"]
        w["cost"] = max(w["cost"], abs(ca - cb) / ca)
    assert worst["pos"] <= pos_tol and worst["rot"] <= rot_tol and worst["sb"] <= sb_tol and worst["lm"] <= lm_tol, (worst, early)
    assert worst["cost"] <= cost_tol, (worst, early)   # the north_star tolerance on the final cost, after every optimize()
    assert early["pos"] <= 1e-2 and early["sb"] <= 5e-2 and early["cost"] <= 1e-2, early
    return worst, tr_g, truth


def test_sliding_window_matches_reference_estimator():
    worst, tr, truth = _compare(dict(n_frames=16, num_keyframes=3, num_imu_frames=3, iters=5, seed=7))
    last = tr[-1]
    T, sb = last["poses"][truth["last_id"]], last["sbs"][truth["last_id"]]
    assert np.linalg.norm(sb - np.r_[truth["speed"], np.zeros(6)]) < 0.04
    assert np.linalg.norm(T[:3] - truth["r_last"]) < 1e-1
    assert sum(len(r["removed"]) for r in tr) > 0
    print("worst deviations from the reference Estimator:", worst)


def test_sliding_window_with_estimated_extrinsics():
    _compare(dict(n_frames=10, num_keyframes=3, num_imu_frames=3, iters=4, seed=11, extrinsics_sigmas=(1e-3, 1e-4, 1e-8, 1e-7)))


def test_growing_window_without_marginalization():
    _compare(dict(n_frames=8, iters=8, seed=13, marginalize=False))
