"""GPU parity of the SEPARATE Schur launch — the path of every batch of more than 48 windows, i.e. of the headline bench line — on
small batches (options.reserved0 bit 2 keeps the separate launch where the fused linearise + reduce launch would be chosen).

Round 5: in DOGLEG and fixed-radius runs that launch takes no trust-region decision any more (schur_mfma_kernel, nodec): it reduces
the trial buffer into that buffer's own set of partials and the solve kernel decides, as in fused mode.  The tuning flag
OKVIS_BA_TUNE_SCHUR_DECIDES keeps the decision at the head of the Schur launch; both routes are compared with the oracle here, and with
each other."""
import os

import numpy as np
import pytest

from okvis_amd import solver, synthetic
from okvis_amd.window import TUNE_SCHUR_DECIDES, default_options

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opts(mode, **kw):
    o = default_options()
    o.reserved0 = 4                      # no fused launch: Schur kernel + solve kernel + linearise launch
    if mode == "gn":
        o.gauss_newton = 1
    elif mode == "lm":
        o.strategy = 1                   # OKVIS_BA_STRATEGY_LM: the damping depends on the decision, which stays in the Schur launch
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _windows():
    # (far starts: rejected steps.  Not seed 41 of test_dogleg_rejected_steps, whose cost follows the rounding at the 1e-6 level)
    return [synthetic.small_window(seed=60 + i, K=4 + i % 3, L=50 + 15 * i) for i in range(3)] + \
           [synthetic.small_window(seed=sd, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8) for sd in (42, 43, 44)]


@pytest.mark.parametrize("mode", ["dogleg", "gn", "lm"])
def test_separate_launch_matches_oracle(oracle, mode):
    ws = _windows()
    kw = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    b = solver.WindowBatch(ws, options=_opts(mode, **kw))
    for n in (1, 4, 12):
        sg = b.optimize(n)
        for i, w in enumerate(ws):
            o = oracle.OracleWindow(w)
            ref = None
            for m in (1, 4, 12):         # the batch keeps its states between the calls: the oracle repeats the calls
                ref = o.optimize(m, _opts(mode, **kw))
                if m == n:
                    break
            tol = 1e-8 if i >= 3 else 1e-9
            assert abs(sg[i]["final_cost"] - ref["final_cost"]) <= tol * ref["final_cost"], (mode, n, i, sg[i], ref)
            assert (sg[i]["iterations"], sg[i]["successful_steps"]) == (ref["iterations"], ref["successful_steps"]), (mode, n, i, sg[i], ref)
    if mode == "dogleg":
        assert any(s["successful_steps"] < s["iterations"] for s in sg), "no rejected step in the far-start windows"
    b.close()


def _both_modes(flags):
    res = {}
    for mode in ("dogleg", "gn"):
        ws = _windows()
        o = _opts(mode)
        o.tuning.flags = flags
        b = solver.WindowBatch(ws, options=o)
        route = b.launch_route()
        assert route["fused"] == 0 and route["decision_free_schur"] == (0 if flags & TUNE_SCHUR_DECIDES else 1), route
        sm = b.optimize(10)
        res[mode + "_cost"] = np.array([x["final_cost"] for x in sm])
        res[mode + "_iter"] = np.array([x["iterations"] for x in sm])
        res[mode + "_succ"] = np.array([x["successful_steps"] for x in sm])
        res[mode + "_pose"] = np.concatenate([b.get_state(i)[0].reshape(-1) for i in range(len(ws))])
        b.close()
    return res


def test_decision_free_schur_launch_against_the_deciding_one():
    """The two routes: the same accepted / rejected steps; fixed-radius runs (every step accepted: the same
    partials summed in the same order) agree bit for bit, and so do DOGLEG runs without a rejected step; after a rejection the solve
    kernel corrects the sums it took from the trial's set by the difference of the two sets where the deciding launch reduces
    the accepted buffer again: the far-start windows then differ by rounding (measured 4e-12 on the cost)."""
    out = {"spec": _both_modes(0), "dec": _both_modes(TUNE_SCHUR_DECIDES)}
    a, b = out["spec"], out["dec"]
    for k in ("dogleg_iter", "dogleg_succ", "gn_iter", "gn_succ"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["gn_cost"], b["gn_cost"]) and np.array_equal(a["gn_pose"], b["gn_pose"])
    # (a rejected step or a Gauss-Newton trial replaced by an explicit dogleg step: the correction path; the other windows agree bit for bit)
    assert np.count_nonzero(a["dogleg_cost"] == b["dogleg_cost"]) >= len(a["dogleg_cost"]) // 2
    assert np.abs(a["dogleg_cost"] - b["dogleg_cost"]).max() <= 1e-10 * np.abs(b["dogleg_cost"]).max()
    assert np.abs(a["dogleg_pose"] - b["dogleg_pose"]).max() <= 1e-8


def _ride_run(mode, flags, radius=None):
    ws = _windows()
    kw = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    if radius is not None:
        kw["initial_radius"] = radius
    o = _opts(mode, **kw)
    o.tuning.split_small_min = 1          # the IMU / prior factors in a launch of their own, as from 40 windows on
    o.tuning.flags = flags
    b = solver.WindowBatch(ws, options=o)
    route = b.launch_route()
    out = dict(route=route, runs=[])
    for n in (1, 3, 8):
        sm = b.optimize(n)
        out["runs"].append(dict(cost=[float(x["final_cost"]).hex() for x in sm], it=[(x["iterations"], x["successful_steps"]) for x in sm],
                                pose=[b.get_state(i)[0].tobytes() for i in range(len(ws))],
                                sb_ref=[b.array("IMU_SB_REF", i).tobytes() for i in range(len(ws))],
                                redo=[b.array("IMU_REDO_COUNT", i).tobytes() for i in range(len(ws))], sm=sm))
    b.close()
    return ws, o, out


@pytest.mark.parametrize("mode, radius", [("dogleg", None), ("dogleg", 30.0), ("gn", None)])
def test_factor_evaluation_riding_in_the_schur_launch(oracle, mode, radius):
    """Round 6: where the IMU / prior factors have a launch of their own and the Schur launch takes no decision, that launch carries
    their EVALUATION (schur_ride_kernel) and small_prepare_kernel behind the solve launch only keeps the preintegration records up to date
    (take-back of a discarded speculative evaluation's, bias check, re-preintegration).  Against OKVIS_BA_TUNE_NO_SMALL_RIDE (the whole
    factors in small_kernel): the same bits everywhere — costs, book-keeping, poses, every term's reference bias and its count of
    re-preintegrations, from near and far starts, with a radius that makes mis-speculated Gauss-Newton trials — and against the oracle."""
    from okvis_amd.window import TUNE_NO_SMALL_RIDE
    ws, o, ride = _ride_run(mode, 0, radius)
    _, _, plain = _ride_run(mode, TUNE_NO_SMALL_RIDE, radius)
    assert ride["route"]["small_rides"] == 1 and ride["route"]["split_small"] == 1 and ride["route"]["decision_free_schur"] == 1, ride["route"]
    assert plain["route"]["small_rides"] == 0 and plain["route"]["split_small"] == 1, plain["route"]
    for a, b in zip(ride["runs"], plain["runs"]):
        for k in ("cost", "it", "pose", "sb_ref", "redo"):
            assert a[k] == b[k], k
    if mode == "dogleg":
        assert any(np.frombuffer(r, np.float64).max() >= 2 for r in ride["runs"][-1]["redo"]), "no term re-preintegrated: the first half is not exercised"
    # ... and the oracle (the batch keeps its states between the calls: the oracle repeats the calls)
    for i, w in enumerate(ws):
        ow = oracle.OracleWindow(w)
        for run, n in zip(ride["runs"], (1, 3, 8)):
            ref = ow.optimize(n, o)
            g = run["sm"][i]
            assert (g["iterations"], g["successful_steps"]) == (ref["iterations"], ref["successful_steps"]), (i, n)
            assert abs(g["final_cost"] - ref["final_cost"]) <= (1e-8 if i >= 3 else 1e-9) * ref["final_cost"], (i, n, g["final_cost"], ref["final_cost"])


def test_riding_evaluation_without_imu_terms_and_in_a_ragged_batch(oracle):
    """The riding launch when there is nothing to prepare: windows without IMU terms (only the prior workgroup rides, no prepare launch at all)
    and a batch that mixes them with windows of 3 - 5 terms (workgroups beyond a window's own count leave)."""
    from okvis_amd.window import TUNE_NO_SMALL_RIDE
    kw = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    no_imu = [synthetic.small_window(seed=70 + i, K=4, L=50, with_imu=False) for i in range(3)]
    mixed = no_imu[:2] + [synthetic.small_window(seed=80 + i, K=4 + i, L=45 + 10 * i) for i in range(3)]
    for ws in (no_imu, mixed):
        runs = []
        for flags in (0, TUNE_NO_SMALL_RIDE):
            o = _opts("dogleg", **kw)
            o.tuning.split_small_min = 1
            o.tuning.flags = flags
            b = solver.WindowBatch(ws, options=o)
            assert b.launch_route()["small_rides"] == (0 if flags else 1)
            sm = b.optimize(6)
            runs.append(([float(x["final_cost"]).hex() for x in sm], [(x["iterations"], x["successful_steps"]) for x in sm],
                         [b.get_state(i)[0].tobytes() for i in range(len(ws))], sm))
            b.close()
        assert runs[0][:3] == runs[1][:3]
        for i, w in enumerate(ws):
            ref = oracle.OracleWindow(w).optimize(6, _opts("dogleg", **kw))
            g = runs[0][3][i]
            assert (g["iterations"], g["successful_steps"]) == (ref["iterations"], ref["successful_steps"]), i
            # (a window without IMU terms has no scale: six DOGLEG iterations along its free direction end 4e-8 apart between two correct
            #  solvers — measured; north_star's tolerance there, 1e-9 where the IMU terms fix the gauge)
            tol = 1e-6 if w.n_imu == 0 else 1e-9
            assert abs(g["final_cost"] - ref["final_cost"]) <= tol * ref["final_cost"], (i, g["final_cost"], ref["final_cost"])
