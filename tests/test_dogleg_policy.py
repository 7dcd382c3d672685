"""The reference's configured trust-region policy (Estimator.cpp:854-873: DOGLEG, traditional, Jacobi scaling) — CPU.

Two statements of it are compared on identical windows:
  * oracle/orc_window.cpp `dogleg_loop`: landmark Schur complement, unscaled variables (what the GPU mirrors);
  * oracle/ref/ceres_shim_solve.cpp behind `okvis::ceres::Map::solve()`: generic ::ceres::Problem, FULL normal equations by
    dense Cholesky in column-scaled variables, every residual / Jacobian evaluated by the okvis reference's own error
    terms (oracle/_ref, compiled unmodified) and chained through the reference's local parameterisations.
Neither is Ceres (not in the tree); what they pin is that the policy as stated is implemented consistently
and that the oracle's Schur-based algebra equals the direct one.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import ref_lib as R  # noqa: E402

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built and /root/reference absent")

import make_golden as G  # noqa: E402
from okvis_amd import synthetic  # noqa: E402
from okvis_amd.window import STRATEGY_DOGLEG, STRATEGY_LM, default_options  # noqa: E402


def _both(oracle, w, iters, opt, dogleg=True):
    o = oracle.OracleWindow(w)
    so = o.optimize(iters, opt)
    r = R.RefWindow(w)
    sr = r.optimize(iters, opt, dogleg=dogleg)
    return so, sr, o.get_state(), r.get_state()


@pytest.mark.parametrize("case", range(len(G.SMALL)))
@pytest.mark.parametrize("radius", [1e4, 30.0, 1.0])
def test_dogleg_iterates_match(oracle, case, radius):
    """radius 1e4: Gauss-Newton point inside the region (case 1 of the dogleg); 30: interpolation between the Cauchy
    and the Gauss-Newton point (case 3); 1: scaled Cauchy steps first (case 2), radius tripled after good steps."""
    w = synthetic.small_window(**G.SMALL[case])
    opt = default_options(STRATEGY_DOGLEG)
    opt.initial_radius = radius
    so, sr, xo, xr = _both(oracle, w, 10, opt)
    assert abs(so["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    for k in ("iterations", "successful_steps", "termination"):
        assert so[k] == sr[k], (k, so, sr)
    assert abs(so["final_radius"] - sr["final_radius"]) <= 1e-6 * sr["final_radius"]
    for a, b in zip(xo, xr):
        assert np.abs(a - b).max() <= 1e-8
    if radius == 1.0:
        assert so["final_radius"] > 1e3     # 3x per accepted boundary step


def test_dogleg_terminates_by_function_tolerance_without_taking_the_step(oracle):
    """Ceres <= 1.10 returns on |cost change| < function_tolerance * cost BEFORE accepting the step."""
    w = synthetic.small_window(**G.SMALL[0])
    opt = default_options(STRATEGY_DOGLEG)
    opt.function_tolerance = 1e-3
    so, sr, xo, xr = _both(oracle, w, 50, opt)
    assert so["termination"] == 1 and sr["termination"] == 1
    assert so["iterations"] == sr["iterations"] and so["successful_steps"] == sr["successful_steps"]
    assert so["iterations"] == so["successful_steps"] + 1      # the last iteration computed a step and dropped it
    assert abs(so["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    for a, b in zip(xo, xr):
        assert np.abs(a - b).max() <= 1e-8


def test_dogleg_without_jacobi_scaling(oracle):
    w = synthetic.small_window(**G.SMALL[1])
    opt = default_options(STRATEGY_DOGLEG)
    opt.jacobi_scaling = 0
    opt.initial_radius = 100.0
    so, sr, xo, xr = _both(oracle, w, 8, opt)
    assert abs(so["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    assert so["iterations"] == sr["iterations"] and so["successful_steps"] == sr["successful_steps"]


def test_dogleg_with_rejected_steps(oracle):
    """badly perturbed windows: Gauss-Newton steps overshoot and are rejected, the radius halves and only the interpolation
    is redone (reuse_), identical in both statements"""
    found = False
    for seed in (41, 42, 43, 44):
        w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
        opt = default_options(STRATEGY_DOGLEG)
        opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
        so, sr, xo, xr = _both(oracle, w, 20, opt)
        found = found or so["successful_steps"] < so["iterations"]
        assert so["iterations"] == sr["iterations"] and so["successful_steps"] == sr["successful_steps"], (so, sr)
        assert abs(so["final_cost"] - sr["final_cost"]) <= 1e-7 * sr["final_cost"]
        assert abs(so["final_radius"] - sr["final_radius"]) <= 1e-5 * sr["final_radius"]
    assert found, "no rejected step in any of the seeds: the scenario does not exercise the path"


def test_dogleg_config_A_costs(oracle):
    """BASELINE configs[1] (10 KF x 2 cam x 400 landmarks): the two statements agree on the 10-iteration result"""
    w = synthetic.config_A()
    opt = default_options(STRATEGY_DOGLEG)
    so, sr, xo, xr = _both(oracle, w, 5, opt)
    assert abs(so["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    assert so["iterations"] == sr["iterations"] and so["successful_steps"] == sr["successful_steps"]


@pytest.mark.parametrize("case", range(len(G.SMALL)))
@pytest.mark.parametrize("radius", [1e4, 30.0, 1.0])
def test_numpy_statement_of_the_policy(oracle, case, radius):
    """Third statement: the policy in numpy on the full dense normal equations (tests/golden/independent.py::dogleg_minimize,
    written from the description of Ceres 1.9's minimiser), fed by the REFERENCE's own factors through RefWindow
    (cost / full_system / set_state).  It has to agree with the oracle (Schur complement, C++) and with the ceres-shim."""
    import independent as I
    w = synthetic.small_window(**G.SMALL[case])
    opt = default_options(STRATEGY_DOGLEG)
    opt.initial_radius = radius
    so = oracle.OracleWindow(w).optimize(10, opt)
    r = R.RefWindow(w)
    sn = I.dogleg_minimize(r, 10, initial_radius=radius)
    assert abs(sn["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"], (sn, so)
    assert (sn["iterations"], sn["successful_steps"], sn["termination"]) == (so["iterations"], so["successful_steps"], so["termination"]), (sn, so)
    assert abs(sn["final_radius"] - so["final_radius"]) <= 1e-5 * so["final_radius"]
    assert abs(sn["initial_cost"] - so["initial_cost"]) <= 1e-12 * so["initial_cost"]


def test_numpy_statement_with_rejected_steps(oracle):
    import independent as I
    found = False
    for seed in (41, 42, 43, 44):
        w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
        opt = default_options(STRATEGY_DOGLEG)
        opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
        so = oracle.OracleWindow(w).optimize(20, opt)
        sn = I.dogleg_minimize(R.RefWindow(w), 20, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        found = found or sn["successful_steps"] < sn["iterations"]
        assert (sn["iterations"], sn["successful_steps"]) == (so["iterations"], so["successful_steps"]), (sn, so)
        assert abs(sn["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]
    assert found
