"""GPU: the chain solver of the reduced camera system (okvis_amd/csrc/ba_chain.hpp, okvis_ba_tuning::solve_mode) — the speed/bias
blocks eliminated along the IMU chain, then the dense pose system — (i) on its own through okvis_ba_reduced_solve against numpy and
against the dense blocked LDL^T on the same systems, (ii) inside the optimisation: windows of every kind through both solvers against
the oracle (which keeps Ceres' landmark-only ordering and a dense Cholesky of the whole reduced system, Estimator.cpp:854)."""
import ctypes as C

import numpy as np
import pytest

from okvis_amd import _lib, solver, synthetic
from okvis_amd.window import SOLVE_CHAIN, SOLVE_DENSE, STRATEGY_LM, default_options, set_options
from tests.chain_emulation import chain_solve, chain_structured_system

pytestmark = pytest.mark.gpu
_dp = C.POINTER(C.c_double)


def reduced_solve(H, g, Dp, mode, comp_mask=0, repeats=1, dump=None):
    D = H.shape[0]
    Hc = np.ascontiguousarray(H, np.float64)
    gc = np.ascontiguousarray(g, np.float64)
    x = np.zeros(D)
    ticks, info = C.c_int64(), C.c_int32()
    _lib.check(_lib.lib().okvis_ba_reduced_solve(0, D, Dp, mode, comp_mask, Hc.ctypes.data_as(_dp), gc.ctypes.data_as(_dp), x.ctypes.data_as(_dp),
                                                 C.byref(ticks), C.byref(info), repeats,
                                                 None if dump is None else dump.ctypes.data_as(_dp), 0 if dump is None else dump.size), "reduced_solve")
    return x, ticks.value, info.value


@pytest.mark.parametrize("n_pose,n_sb", [(10, 10), (8, 3), (4, 4), (1, 1), (2, 2), (5, 1), (3, 5), (12, 10), (6, 11), (10, 9), (16, 2), (3, 17)])
def test_reduced_solve_chain_against_numpy_and_the_dense_solver(n_pose, n_sb):
    rng = np.random.default_rng(1000 * n_pose + n_sb)
    for prior in (0.0, 1e10):
        H, g, Dp = chain_structured_system(rng, n_pose, n_sb, pose_prior=prior)
        xr = np.linalg.solve(H, g)
        xe = chain_solve(H, g, Dp)
        xc, _, fc = reduced_solve(H, g, Dp, SOLVE_CHAIN)
        xd, _, fd = reduced_solve(H, g, Dp, SOLVE_DENSE)
        assert fc == 0 and fd == 0
        res = lambda v: np.abs(H @ v - g).max() / (np.abs(H).max() * np.abs(v).max() + np.abs(g).max())
        assert res(xc) < 1e-13 and res(xd) < 1e-13, (res(xc), res(xd))
        tol = 1e-11 if prior == 0.0 else 1e-6
        sc = np.abs(xr).max()
        assert np.abs(xc - xr).max() <= tol * sc and np.abs(xd - xr).max() <= tol * sc and np.abs(xc - xe).max() <= tol * sc


def test_reduced_solve_reports_a_pivot_that_is_not_positive():
    rng = np.random.default_rng(5)
    H, g, Dp = chain_structured_system(rng, 4, 4, pose_prior=0.0)
    D = H.shape[0]
    for where in (Dp + 3, Dp + 9 * 2 + 1, D - 1, 2):     # in the left sweep, the middle block, the right sweep, the pose system
        Hb = H.copy()
        Hb[where, where] = -1.0
        _, _, f = reduced_solve(Hb, g, Dp, SOLVE_CHAIN)
        assert f == 1, where
    _, _, f = reduced_solve(H, g, Dp, SOLVE_CHAIN)
    assert f == 0


def test_reduced_solve_timing_record(capsys):
    """not an assertion on speed: the ticks of the two solvers on configs[1]'s shape, printed for the log (profiles/)"""
    rng = np.random.default_rng(9)
    H, g, Dp = chain_structured_system(rng, 10, 10, pose_prior=1e10)
    out = {}
    for name, mode in (("dense", SOLVE_DENSE), ("chain", SOLVE_CHAIN)):
        _, t, _ = reduced_solve(H, g, Dp, mode, repeats=20)
        out[name] = t
    with capsys.disabled():
        print(f"\n  reduced solve D = 150 (10 poses + 10 speed/bias blocks), device ticks per solve: {out}")
    assert out["dense"] > 0 and out["chain"] > 0


def _opts(mode, **kw):
    return set_options(default_options(), tuning_solve_mode=mode, **kw)


def _windows():
    far = dict(pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
    ws = [synthetic.config_A(),
          synthetic.small_window(seed=3, K=4, L=60),
          synthetic.small_window(seed=11, K=5, L=60, **far),
          synthetic.small_window(seed=41, K=5, L=60, **far),
          synthetic.make_window(7, 120, 0.6, 77),
          synthetic.make_window(11, 80, 0.7, seed=78, frame_dt=0.2),           # D = 165
          synthetic.make_window(3, 50, 1.0, 79),
          synthetic.make_window(2, 40, 1.0, 80)]
    return ws


@pytest.mark.parametrize("mode", [SOLVE_CHAIN, SOLVE_DENSE])
@pytest.mark.parametrize("trust", ["dogleg", "gn", "lm"])
def test_windows_through_both_solvers_against_the_oracle(oracle, mode, trust):
    kw = {}
    if trust == "gn":
        kw["gauss_newton"] = 1
    elif trust == "lm":
        kw["strategy"] = STRATEGY_LM
    for i, w in enumerate(_windows()):
        o = _opts(mode, **kw)
        b = solver.WindowBatch([w], options=o)
        want = SOLVE_CHAIN if mode == SOLVE_CHAIN else SOLVE_DENSE        # (every one of them fits the chain solver, D = 165 included)
        assert (solver.index_lists(w, o)["chain"] > 0) == (want == SOLVE_CHAIN)
        assert b.launch_route()["solve_mode"] == want, (i, b.launch_route())
        sg = b.optimize(8)[0]
        ow = oracle.OracleWindow(w)
        sr = ow.optimize(8, o)
        tol = 1e-7 if i in (2, 3) else 1e-9          # (the far starts: interpolated dogleg steps behind a 1e16 prior)
        assert abs(sg["final_cost"] - sr["final_cost"]) <= tol * sr["final_cost"], (i, mode, trust, sg, sr)
        assert (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"]), (i, sg, sr)
        pg, sbg, lg = b.get_state()
        pr, sbr, lr = ow.get_state()
        st = 1e-7 * max(1.0, tol / 1e-9)
        assert np.abs(pg - pr).max() < st and np.abs(sbg - sbr).max() < st and np.abs(lg - lr).max() < 10 * st, i
        b.close()


def test_step_of_the_chain_solver_equals_the_dense_solvers(oracle):
    """one iteration from the same state: the reduced step of the two solvers (OKVIS_BA_ARR_STEP) against each other and against
    the oracle's, the damped reduced matrix they factorise read back through both layouts bit for bit"""
    w = synthetic.config_A()
    out = {}
    for mode in (SOLVE_CHAIN, SOLVE_DENSE):
        o = _opts(mode, debug_arrays=1, use_graph=0)
        b = solver.WindowBatch([w], options=o)
        b.begin()
        b.iterate(1)
        b.synchronize()
        out[mode] = (b.array("STEP"), b.array("REDUCED_S"), b.array("REDUCED_RHS"))
        b.finish()
        b.close()
    assert np.array_equal(out[SOLVE_CHAIN][1], out[SOLVE_DENSE][1]) and np.array_equal(out[SOLVE_CHAIN][2], out[SOLVE_DENSE][2])
    sc, sd = out[SOLVE_CHAIN][0], out[SOLVE_DENSE][0]
    D = w.reduced_dim()
    S = out[SOLVE_DENSE][1].reshape(D, D)
    xr = np.linalg.solve(S, out[SOLVE_DENSE][2])
    scale = np.abs(xr).max()
    assert np.abs(sc - xr).max() <= 1e-8 * scale and np.abs(sd - xr).max() <= 1e-8 * scale
    assert np.abs(sc - sd).max() <= 1e-8 * scale


@pytest.mark.parametrize("mode", [SOLVE_CHAIN, SOLVE_DENSE])
def test_batches_and_marginalisation_prior_through_both_solvers(oracle, mode):
    """a batch of mixed shapes (fused route), a batch behind a separate Schur launch, and windows that carry a marginalisation prior
    over poses and ONE speed/bias block (what the sliding window's prior looks like)"""
    ws = [synthetic.make_window(4 + i % 4, 60 + 10 * i, 0.8, 500 + i) for i in range(10)]
    for sep in (0, 4):
        o = _opts(mode, reserved0=sep)
        b = solver.WindowBatch(ws, options=o)
        sg = b.optimize(6)
        for i, w in enumerate(ws):
            sr = oracle.OracleWindow(w).optimize(6, o)
            assert abs(sg[i]["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"], (sep, i, sg[i], sr)
            assert (sg[i]["iterations"], sg[i]["successful_steps"]) == (sr["iterations"], sr["successful_steps"])
        b.close()
    rng = np.random.default_rng(6)
    for pair in ((0, 0), (0, 1)):        # prior over two poses and: one speed/bias block | two neighbours
        wm = synthetic.small_window(seed=6, K=5, L=70)
        sbs = sorted(set(pair))
        Dm = 12 + 9 * len(sbs)
        wm.marg_J = np.triu(rng.standard_normal((Dm, Dm))) * 3.0
        wm.marg_e0 = rng.standard_normal(Dm) * 0.1
        wm.marg_block_type = np.array([0, 0] + [1] * len(sbs), np.int32)
        wm.marg_block_idx = np.array([0, 1] + sbs, np.int32)
        wm.marg_block_off = np.array([0, 6] + [12 + 9 * k for k in range(len(sbs))], np.int32)
        lin = np.zeros((2 + len(sbs), 9))
        lin[0, :7], lin[1, :7] = wm.pose[0], wm.pose[1]
        for k, sbi in enumerate(sbs):
            lin[2 + k] = wm.sb[sbi]
        wm.marg_lin = lin
        o = _opts(mode)
        b = solver.WindowBatch([wm], options=o)
        assert b.launch_route()["solve_mode"] == mode
        sg = b.optimize(6)[0]
        sr = oracle.OracleWindow(wm).optimize(6, o)
        assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"], (pair, sg, sr)
        assert (sg["iterations"], sg["successful_steps"]) == (sr["iterations"], sr["successful_steps"])
        b.close()
