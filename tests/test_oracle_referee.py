"""The oracle in extended precision as a referee (VERDICT r4, item 8 i).

oracle/liboracle_ld.so is oracle/liboracle.so's sources built with orc::real = long double (x87: 64-bit mantissa, 11 bits more
than fp64); the C entry points take and return double in both.  It answers which side of a disagreement between the GPU and the
fp64 oracle moved: the far-start DOGLEG case of test_gpu_dogleg.py::test_dogleg_rejected_steps (seed 41: ten accepted and ten
rejected steps along a flat valley) used to sit 6e-7 ... 1.7e-6 from the fp64 oracle depending on a tuning knob of the index
build.  Measured (tools/gpu_referee_spread.py, 12 one-ulp perturbations of the landmark start values, each run against the
referee of the same input): fp64 oracle 3.5e-9 (median) / 1.4e-8 (max) from the referee; the GPU 2.9e-7 / 7.1e-7 before the
compensated elimination of the prior blocks (ba_ldl16.hpp, WinPtrs::ldl_comp), 1e-9 ... 1.3e-8 with it.  The oracle had been right.

CPU tests: the referee builds, agrees with the fp64 build where nothing is ill-conditioned, and the fp64 oracle stays within 5e-8 of it
on the far-start cases.  GPU tests: the GPU against the referee."""
import numpy as np
import pytest

from okvis_amd import synthetic
from okvis_amd.window import SOLVE_DENSE, STRATEGY_DOGLEG, TUNE_NO_LDL_COMP, default_options

FAR = dict(K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)


def _opts(**kw):
    o = default_options(kw.pop("strategy", STRATEGY_DOGLEG))
    o.function_tolerance = 0.0
    o.gradient_tolerance = 0.0
    o.parameter_tolerance = 0.0
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_referee_matches_fp64_build_on_a_well_conditioned_window(oracle):
    w = synthetic.small_window(seed=3, K=4, L=60)
    a = oracle.OracleWindow(w)
    b = oracle.OracleWindow(w, extended=True)
    assert abs(a.linearize() - b.linearize()) <= 1e-13 * b.linearize()
    for name in ("OBS_RESIDUAL", "LM_V", "LM_B", "PAIR_W", "IMU_RESIDUAL"):
        x, y = a.array(name), b.array(name)
        assert x.shape == y.shape and np.abs(x - y).max() <= 1e-12 * max(np.abs(y).max(), 1e-300), name
    sa, sb = a.optimize(6, default_options()), b.optimize(6, default_options())
    assert (sa["iterations"], sa["successful_steps"], sa["termination"]) == (sb["iterations"], sb["successful_steps"], sb["termination"])
    assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e-9 * sb["final_cost"]


@pytest.mark.parametrize("seed", [41, 42, 43, 44])
def test_fp64_oracle_against_referee_far_start(oracle, seed):
    w = synthetic.small_window(seed=seed, **FAR)
    a = oracle.OracleWindow(w).optimize(20, _opts())
    b = oracle.OracleWindow(w, extended=True).optimize(20, _opts())
    assert (a["iterations"], a["successful_steps"]) == (b["iterations"], b["successful_steps"])
    assert abs(a["final_cost"] - b["final_cost"]) <= 5e-8 * b["final_cost"], (a["final_cost"], b["final_cost"])   # measured <= 3.8e-9


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [41, 42, 43, 44])
def test_gpu_against_referee_far_start(oracle, seed):
    from okvis_amd import solver
    w = synthetic.small_window(seed=seed, **FAR)
    ref = oracle.OracleWindow(w, extended=True)
    r = ref.optimize(20, _opts())
    b = solver.WindowBatch([w], options=_opts())
    g = b.optimize(20)[0]
    assert (g["iterations"], g["successful_steps"], g["termination"]) == (r["iterations"], r["successful_steps"], r["termination"])
    assert abs(g["final_cost"] - r["final_cost"]) <= 5e-8 * r["final_cost"], (g["final_cost"], r["final_cost"])   # measured <= 1.5e-9
    pg, sbg, lg = b.get_state()
    pr, sbr, lr = ref.get_state()
    # (the state moves further than the cost along the valley: seed 41 poses 2e-8)
    assert np.abs(pg - pr).max() < 1e-7 and np.abs(sbg - sbr).max() < 1e-7 and np.abs(lg - lr).max() < 1e-6
    b.close()


@pytest.mark.gpu
def test_gpu_spread_under_one_ulp_perturbations(oracle):
    """the 20-iteration cost of seed 41 under one-ulp perturbations of the landmark start values, each against the referee of the
    same input: no run further than 5e-8 (before the compensated elimination: 7e-7; the fp64 oracle: 1.4e-8; measured 1.2e-9 ... 1.3e-8
    over the builds of round 5 — the second iteration takes the cost from 124 k to 1.6 k and multiplies whatever the first left)"""
    from okvis_amd import solver
    w = synthetic.small_window(seed=41, **FAR)
    rng = np.random.default_rng(0)
    worst = 0.0
    for t in range(6):
        lm2 = w.lm.copy()
        if t:
            lm2[:, :3] = np.nextafter(w.lm[:, :3], w.lm[:, :3] + rng.choice([-1.0, 1.0], size=w.lm[:, :3].shape))
        ref = oracle.OracleWindow(w, extended=True)
        ref.set_state(lm=lm2)
        r = ref.optimize(20, _opts())
        b = solver.WindowBatch([w], options=_opts())
        b.set_state(0, lm=lm2)
        g = b.optimize(20)[0]
        b.close()
        assert g["successful_steps"] == r["successful_steps"]
        worst = max(worst, abs(g["final_cost"] - r["final_cost"]) / r["final_cost"])
    assert worst <= 5e-8, worst


@pytest.mark.gpu
def test_switching_the_compensated_elimination_off_brings_the_distance_back(oracle):
    """OKVIS_BA_TUNE_NO_LDL_COMP (okvis_ba_options::tuning, read when a window is uploaded): the same inputs, the plain elimination of the prior blocks — the
    20-iteration cost of seed 41 is several 1e-7 from the referee again on at least one of four inputs one ulp apart (measured
    6e-7), and within 5e-8 on all of them with the default.  Keeps the switch honest and the defect on record."""
    from okvis_amd import solver
    w = synthetic.small_window(seed=41, **FAR)
    rng = np.random.default_rng(0)
    inputs = [w.lm.copy()]
    for _ in range(4):
        lm2 = w.lm.copy()
        lm2[:, :3] = np.nextafter(w.lm[:, :3], w.lm[:, :3] + rng.choice([-1.0, 1.0], size=w.lm[:, :3].shape))
        inputs.append(lm2)
    worst = {}
    for off in (False, True):
        og = _opts()
        og.tuning.flags = TUNE_NO_LDL_COMP if off else 0
        og.tuning.solve_mode = SOLVE_DENSE      # (the switch belongs to the dense blocked LDL^T)
        worst[off] = 0.0
        for lm2 in inputs:
            ref = oracle.OracleWindow(w, extended=True)
            ref.set_state(lm=lm2)
            r = ref.optimize(20, _opts())
            b = solver.WindowBatch([w], options=og)
            b.set_state(0, lm=lm2)
            g = b.optimize(20)[0]
            b.close()
            worst[off] = max(worst[off], abs(g["final_cost"] - r["final_cost"]) / r["final_cost"])
    assert worst[False] <= 5e-8, worst
    assert worst[True] >= 1e-7, worst


# ---- windows above the LDS solver's size: the tiled solver (ba_chol_tiles.hpp) is not compensated ------------------------
LARGE = [("near", dict(seed=33), "dogleg_r40", 1e-9), ("near", dict(seed=33), "dogleg", 1e-11), ("near", dict(seed=33), "lm", 1e-11),
         ("far", dict(seed=34, pose_noise=(0.3, np.deg2rad(4.0)), landmark_noise=0.5), "dogleg_r40", 5e-8),
         ("far", dict(seed=34, pose_noise=(0.3, np.deg2rad(4.0)), landmark_noise=0.5), "dogleg", 1e-10),
         ("far", dict(seed=34, pose_noise=(0.3, np.deg2rad(4.0)), landmark_noise=0.5), "lm", 1e-10)]


@pytest.mark.gpu
@pytest.mark.parametrize("name, kw, mode, bound", LARGE, ids=[f"{c[0]}-{c[2]}" for c in LARGE])
def test_gpu_tiled_solver_against_referee_with_a_pose_prior(oracle, name, kw, mode, bound):
    """D = 300 (20 keyframes: the multi-workgroup tiled Cholesky, whose elimination is NOT compensated) with the window's 1e16 yaw
    prior, eight iterations from a near and from a far start in DOGLEG (default radius and radius 40: rejected and interpolated
    steps) and LM mode, against the long double referee.  Bounds = 10 x measured (profiles/r05_referee_large.txt: 9.6e-11 / 2.5e-13 /
    3.4e-13 near, 4.4e-9 / 4.4e-12 / 3.6e-12 far; the fp64 oracle's own distance in the worst case: 1.0e-8) — the asymmetric rounding
    that cost the LDS solver 3e-7 before its compensation does not show here."""
    from okvis_amd import solver
    from okvis_amd.window import STRATEGY_LM
    w = synthetic.make_window(20, 200, 1.0, frame_dt=0.1, **kw)
    assert w.reduced_dim() == 300 and len(w.pprior_pose) >= 1 and np.abs(np.asarray(w.pprior_sqrtinfo)).max() >= 1e7   # (sqrt of 1e16)
    o = _opts(strategy=STRATEGY_LM) if mode == "lm" else (_opts(initial_radius=40.0) if mode == "dogleg_r40" else _opts())
    r = oracle.OracleWindow(w, extended=True).optimize(8, o)
    b = solver.WindowBatch([w], options=o)
    assert b.launch_route()["solve_tiled"] == 1
    g = b.optimize(8)[0]
    b.close()
    assert (g["iterations"], g["successful_steps"]) == (r["iterations"], r["successful_steps"])
    assert abs(g["final_cost"] - r["final_cost"]) <= bound * r["final_cost"], (g["final_cost"], r["final_cost"])
