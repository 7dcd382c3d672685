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


def _compare(kw, settle=5, pos_tol=1e-5, rot_tol=3e-6, sb_tol=5e-5, lm_tol=1e-4, cost_tol=1e-6):
    """States are compared tightly from frame `settle` on: tolerances = ten times the worst gap measured over the three
    scenarios below (round 3: pos 1.3e-6 m, rot 2.6e-7, speed/bias 5.1e-6, landmarks 1.0e-5 m, cost 3.2e-8).  The first
    windows of this scenario (TestEstimator.cpp's: one or two frames, 0.1 m stereo baseline, a wall 3 m away) are weakly
    constrained: test_early_windows_gpu_oracle_and_dense_reference below runs GPU, oracle and the dense reference solve on
    the SAME flattened windows and finds, per optimize() call, 1e-5 (poses) / 1e-4 (landmarks) between any two of the three
    in frames 0-1 and <= 4e-9 from frame 2 on.  Along two separately evolving estimators those first differences are carried
    through the weak directions until the window holds 4-5 frames (measured worst before `settle`: pos 2.3e-3, speed/bias
    1.2e-2, cost 2.4e-4), hence the looser bound there."""
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
    print("frames >= %d:" % settle, worst, " earlier frames:", early)
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


# ---- which side moves in the early windows: three solvers on the very same flattened windows ----------------------------
class _ThreeSolvers:
    """Observer of okvis_amd::Estimator::optimize (Estimator::setWindowObserver): every window the backend flattens is also
    handed to the oracle (landmark Schur + reduced solve, the algorithm of the backend on the CPU) and to the reference's own
    Map + error terms with the dense DOGLEG of oracle/ref/ceres_shim_solve.cpp.  Records the pairwise gaps after `iters`."""

    def __init__(self, iters):
        import ctypes as C

        import oracle_lib as O
        from okvis_amd.window import OptionsC, WindowC
        self.iters, self.rows, self.errors = iters, [], []
        self._O, self._C, self._WindowC, self._OptionsC = O, C, WindowC, OptionsC
        O.lib()
        self._cb = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)(self._call)
        self._clones = None

    def attach(self, est):
        C = self._C
        fn = est._api.okvis_est_set_window_observer
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert fn(est._h, C.cast(self._cb, C.c_void_p), None) > 0
        self._est = est
        return est

    def _call(self, wptr, stage, _user):
        try:
            C = self._C
            w = C.cast(wptr, C.POINTER(self._WindowC))
            if stage == 0:
                self._clones = (self._O.OracleWindow.from_c(w), R.RefWindow.from_c(w), self._O.OracleWindow.from_c(w, extended=True))
                return
            orc, ref, ld = self._clones
            wc = w.contents
            opt = self._OptionsC()
            get = self._est._api.okvis_est_get_options
            get.argtypes = [C.c_void_p, C.POINTER(self._OptionsC)]
            assert get(self._est._h, C.byref(opt)) > 0
            g = [np.ctypeslib.as_array(p, shape=s).copy() if s[0] else np.zeros(s) for p, s in
                 ((wc.pose, (wc.n_pose, 7)), (wc.sb, (wc.n_sb, 9)), (wc.lm, (wc.n_lm, 4)))]
            so = orc.optimize(self.iters, opt)
            sr = ref.optimize(self.iters, opt, dogleg=True)
            sl = ld.optimize(self.iters, opt)    # the referee: the oracle's sources in long double (tests/test_oracle_referee.py)
            o, r, l = orc.get_state(), ref.get_state(), ld.get_state()
            gap = lambda a, b: [float(np.abs(x - y).max()) if x.size else 0.0 for x, y in zip(a, b)]  # noqa: E731
            self.rows.append(dict(n_pose=wc.n_pose, n_lm=wc.n_lm, cost_oracle=so["final_cost"], cost_dense=sr["final_cost"],
                                  cost_referee=sl["final_cost"], gpu_oracle=gap(g, o), gpu_dense=gap(g, r), oracle_dense=gap(o, r),
                                  gpu_referee=gap(g, l), oracle_referee=gap(o, l), dense_referee=gap(r, l), oracle=so, referee=sl))
        except Exception as e:   # exceptions do not cross the C boundary: keep them for the test
            self.errors.append(repr(e))


def test_early_windows_gpu_oracle_and_dense_reference():
    """The first windows of the TestEstimator.cpp scenario (frames < 5) were compared at 1e-2 against the reference.  On the
    very same flattened windows: the backend and the oracle (same algorithm, CPU) agree to 1e-7 in every frame, early ones
    included — the gap to the reference side in those frames is between the Schur-complement solvers and the dense solve
    of the reference Map over weakly constrained directions, not between the GPU and its CPU statement."""
    iters = 5
    probe = _ThreeSolvers(iters)
    costs = []

    def make():
        return probe.attach(E.Estimator(0))

    tr, _ = S.sliding_window(make, E.Frame, n_frames=9, num_keyframes=3, num_imu_frames=3, iters=iters, seed=7)
    assert not probe.errors, probe.errors
    assert len(probe.rows) == len(tr) == 9
    for k, (row, rec) in enumerate(zip(probe.rows, tr)):
        cg = rec["summary"]["final_cost"]
        costs.append((k, abs(cg - row["cost_oracle"]) / cg, abs(cg - row["cost_dense"]) / cg))
        print("frame %d poses %d landmarks %d  cost gpu-oracle %.1e gpu-dense %.1e | pose/sb/lm gpu-oracle %s gpu-dense %s "
              "oracle-dense %s" % (k, row["n_pose"], row["n_lm"], costs[-1][1], costs[-1][2],
                                   ["%.1e" % x for x in row["gpu_oracle"]], ["%.1e" % x for x in row["gpu_dense"]],
                                   ["%.1e" % x for x in row["oracle_dense"]]))
        assert (rec["summary"]["iterations"], rec["summary"]["successful_steps"]) == \
               (row["oracle"]["iterations"], row["oracle"]["successful_steps"]), k
        print("         against the long double referee: cost gpu %.1e oracle %.1e dense %.1e | pose/sb/lm gpu %s oracle %s dense %s" % (
            abs(cg - row["cost_referee"]) / cg, abs(row["cost_oracle"] - row["cost_referee"]) / cg, abs(row["cost_dense"] - row["cost_referee"]) / cg,
            ["%.1e" % x for x in row["gpu_referee"]], ["%.1e" % x for x in row["oracle_referee"]], ["%.1e" % x for x in row["dense_referee"]]))
    # measured (round 3): from frame 2 on all three agree: cost <= 7e-12, states <= 4e-9.  Frames 0 and 1 (one or two frames in
    # the window): cost 3.5e-8 ... 9.3e-7, poses 5e-8 ... 1e-5, landmarks 8e-7 ... 1.4e-4 BETWEEN ANY TWO of the three
    for k, row in enumerate(probe.rows):
        if k >= 2:
            assert costs[k][1] <= 1e-10 and costs[k][2] <= 1e-10, costs[k]
            for key in ("gpu_oracle", "gpu_dense", "oracle_dense", "gpu_referee"):
                assert max(row[key]) <= 4e-8, (k, key, row[key])
            assert abs(tr[k]["summary"]["final_cost"] - row["cost_referee"]) <= 1e-10 * row["cost_referee"], k   # (measured <= 5e-12)
        else:
            assert costs[k][1] <= 1e-6 and costs[k][2] <= 1e-5, costs[k]
            assert max(row["gpu_oracle"][:2]) <= 1e-4 and row["gpu_oracle"][2] <= 1e-3, (k, row["gpu_oracle"])
            # the CPU statement of the same algorithm is not closer to the dense solve than the GPU is by more than a small factor:
            # the spread is the conditioning of the window, not a defect of one solver
            # ... and the referee (round 6: the oracle's sources in long double on the same flattened windows) says which side that
            # conditioning moves: the reference's DENSE solve is the one next to it (cost 3e-10, states 2e-7), the two Schur-complement
            # solvers — GPU and fp64 oracle alike — sit 1e-5 (poses) / 2e-4 (landmarks) / 1e-6 (cost) away in frame 0, 7e-8 / 1e-6 /
            # 2e-8 in frame 1, and from frame 2 on everything agrees to 1e-9.  Bounds: 10 x measured (profiles/r06_early_windows_referee.txt)
            cgr = abs(tr[k]["summary"]["final_cost"] - row["cost_referee"]) / row["cost_referee"]
            assert cgr <= (1e-5 if k == 0 else 2e-7), (k, cgr)
            lim = ([1.1e-4, 1e-6, 1.7e-3] if k == 0 else [7e-7, 3.3e-6, 1.1e-5])
            assert all(g <= l for g, l in zip(row["gpu_referee"], lim)), (k, row["gpu_referee"])
            assert max(row["gpu_referee"]) <= 5 * max(row["oracle_referee"]), (k, row)   # (measured 2.4x and 0.7x: not a defect of the device code)
            assert max(row["gpu_dense"]) <= 20 * max(row["oracle_dense"]), (k, row)   # (measured: 1.6x with the staged linearise kernel, 7.5x with the piece path: rounding of a window whose reduced matrix has condition 1e15)
