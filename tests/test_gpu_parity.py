"""GPU parity tests (pytest -m gpu): the HIP path through the C-ABI against the CPU oracle on the same
seeded inputs.  Tolerances: intermediate arrays 1e-9 relative to the array's max magnitude (fp64, different
summation order), final cost 1e-9 relative (north_star asks 1e-6), identical iteration bookkeeping."""
import numpy as np
import pytest

from okvis_amd import synthetic
from okvis_amd.window import DIST_EQUIDISTANT, DIST_NONE, DIST_RADTAN, DIST_RADTAN8, default_options

pytestmark = pytest.mark.gpu


def _batch(ws, **kw):
    from okvis_amd import solver
    opt = default_options()
    for k, v in kw.items():
        setattr(opt, k, v)
    return solver.WindowBatch(ws, options=opt)


def _close(a, b, tol=1e-9):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    scale = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize("ext", ["fixed", "shared", "perframe"])
@pytest.mark.parametrize("model", [DIST_RADTAN, DIST_EQUIDISTANT])
def test_linearisation_arrays_match_oracle(oracle, ext, model):
    w = synthetic.small_window(seed=21, K=4, L=60, estimate_extrinsics=ext, cam_model=model)
    b = _batch([w], debug_arrays=1, use_graph=0)
    o = oracle.OracleWindow(w)
    c_ref = o.linearize()
    b.begin()
    s = b.finish()[0]
    assert abs(s["final_cost"] - c_ref) <= 1e-12 * c_ref
    for name in ("OBS_RESIDUAL", "LM_V", "LM_B", "LM_HQ", "PAIR_W", "IMU_RESIDUAL"):
        _close(b.array(name), o.array(name))
    _close(b.array("LM_QUALITY"), o.array("LM_QUALITY"), 1e-7)
    assert np.array_equal(b.pairs()[0], o.pairs()[0]) and np.array_equal(b.pairs()[1], o.pairs()[1])
    # reduced system and step of the first iteration
    b.begin()
    b.iterate(1)
    opt = default_options()
    assert o.solve(opt.initial_radius, opt) == 0
    S_g, S_r = b.array("REDUCED_S"), o.array("REDUCED_S")
    _close(S_g, S_r, 1e-12)
    _close(b.array("REDUCED_RHS"), o.array("REDUCED_RHS"), 1e-9)   # (1e-6 until round 5: the pose prior's residual, ba_math.hpp qmul_strict)
    _close(b.array("STEP"), o.array("STEP"), 1e-8)
    b.close()


@pytest.mark.parametrize("model", [DIST_NONE, DIST_RADTAN8])
def test_other_distortion_models(oracle, model):
    w = synthetic.small_window(seed=22, K=3, L=40, cam_model=model)
    b = _batch([w], debug_arrays=1)
    o = oracle.OracleWindow(w)
    o.linearize()
    b.begin(); b.finish()
    for name in ("OBS_RESIDUAL", "LM_V", "PAIR_W"):
        _close(b.array(name), o.array(name))
    b.close()


@pytest.mark.parametrize("cfg", [dict(K=4, L=40, ext="fixed"), dict(K=4, L=40, ext="shared"),
                                 dict(K=3, L=30, ext="perframe"), dict(K=6, L=150, ext="fixed", vis=0.35)])
def test_optimize_matches_oracle(oracle, cfg):
    w = synthetic.make_window(cfg["K"], cfg["L"], cfg.get("vis", 0.7), seed=23, estimate_extrinsics=cfg["ext"])
    for n in (1, 3, 10, 40):
        b = _batch([w])
        sg = b.optimize(n)[0]
        o = oracle.OracleWindow(w)
        sr = o.optimize(n)
        assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"], (n, sg, sr)
        assert (sg["iterations"], sg["successful_steps"], sg["termination"]) == \
               (sr["iterations"], sr["successful_steps"], sr["termination"]), (n, sg, sr)
        pg, sbg, lg = b.get_state()
        pr, sbr, lr = o.get_state()
        assert np.abs(pg - pr).max() < 1e-7 and np.abs(sbg - sbr).max() < 1e-7 and np.abs(lg - lr).max() < 1e-6
        b.close()


def test_config_A_full_size_and_batch(oracle):
    # BASELINE configs[1] at full size, as a batch of 4 different windows through the hipGraph path
    ws = [synthetic.config_A(seed=20240923 + i) for i in range(4)]
    b = _batch(ws)
    sg = b.optimize(10)
    for i, w in enumerate(ws):
        sr = oracle.OracleWindow(w).optimize(10)
        assert abs(sg[i]["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
        assert sg[i]["iterations"] == sr["iterations"] and sg[i]["successful_steps"] == sr["successful_steps"]
    # size-independent properties at full size: the cost never increases over accepted steps,
    # optimisation is idempotent at the optimum (function tolerance), landmark w is untouched
    s2 = b.optimize(50)
    for i in range(4):
        assert s2[i]["final_cost"] <= sg[i]["final_cost"] * (1 + 1e-12)
        assert s2[i]["termination"] == 1
    c = b.evaluate_cost()
    s3 = b.optimize(5)
    for i in range(4):
        assert abs(s3[i]["final_cost"] - c[i]) <= 2e-6 * c[i]
        assert np.array_equal(b.get_state(i)[2][:, 3], ws[i].lm[:, 3])
    b.close()


def test_eager_and_graph_paths_agree(oracle):
    w = synthetic.small_window(seed=24)
    a = _batch([w], use_graph=1).optimize(12)[0]
    c = _batch([w], use_graph=0).optimize(12)[0]
    assert a == c     # bitwise: same kernels, same order


def test_gauss_newton_mode_matches_oracle(oracle):
    w = synthetic.small_window(seed=25, K=4, L=50)
    opt = dict(gauss_newton=1, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    b = _batch([w], **opt)
    sg = b.optimize(8)[0]
    o = oracle.OracleWindow(w)
    op = default_options()
    for k, v in opt.items():
        setattr(op, k, v)
    sr = o.optimize(8, op)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    assert sg["successful_steps"] == sr["successful_steps"] == 8


def test_set_state_and_restart(oracle):
    w = synthetic.small_window(seed=26)
    b = _batch([w])
    b.optimize(5)
    p, s, l = b.get_state()
    b.set_state(0, w.pose, w.sb, w.lm)          # Estimator::set_T_WS / setSpeedAndBias / setLandmark
    c0 = b.evaluate_cost()[0]
    o = oracle.OracleWindow(w)
    assert abs(c0 - o.linearize()) <= 1e-12 * c0
    b.close()


def test_edge_cases(oracle):
    from okvis_amd import solver
    # landmark without observations, a single landmark, and a ragged window
    w = synthetic.small_window(seed=27, K=3, L=20)
    keep = w.obs_lm != 5
    for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_uv", "obs_sqrtw"):
        setattr(w, n, getattr(w, n)[keep])
    b = _batch([w])
    sg = b.optimize(6)[0]
    sr = oracle.OracleWindow(w).optimize(6)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    assert np.array_equal(b.get_state()[2][5], w.lm[5])      # unobserved landmark does not move
    b.close()
    # point behind / too close to a camera: residual kept, Jacobians zeroed (ReprojectionError.hpp:143-151)
    w2 = synthetic.small_window(seed=28, K=3, L=20)
    w2.lm = w2.lm.copy()
    w2.lm[0, :3] = w2.pose[0, :3] + 0.01
    b = _batch([w2], debug_arrays=1)
    o = oracle.OracleWindow(w2)
    c = o.linearize()
    b.begin()
    s = b.finish()[0]
    assert abs(s["final_cost"] - c) <= 1e-12 * c
    _close(b.array("LM_V"), o.array("LM_V"))
    _close(b.array("OBS_RESIDUAL"), o.array("OBS_RESIDUAL"))
    b.close()
    # negative homogeneous scale (PinholeCamera.hpp:363-367)
    w3 = synthetic.small_window(seed=29, K=3, L=20)
    w3.lm = w3.lm.copy(); w3.lm[1] *= -1.0
    b = _batch([w3], debug_arrays=1)
    o = oracle.OracleWindow(w3)
    c = o.linearize()
    b.begin(); s = b.finish()[0]
    assert abs(s["final_cost"] - c) <= 1e-12 * c
    _close(b.array("PAIR_W"), o.array("PAIR_W"))
    b.close()


def test_time_limited_optimize_runs_min_iterations():
    w = synthetic.small_window(seed=30)
    b = _batch([w], function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    s = b.optimize_timed(10, 3, 0.0)[0]        # zero budget: exactly the minimum iterations
    assert s["iterations"] == 3
    s = b.optimize_timed(7, 3, -1.0)[0]        # negative limit: no limit -> max iterations
    assert s["iterations"] == 7
    b.close()


def test_long_imu_factors_are_chunked(oracle):
    # 400+ integration steps per factor: the on-device re-preintegration streams over LDS-sized chunks
    w = synthetic.make_window(3, 40, 1.0, seed=31, imu_rate_hz=800)
    assert w.imu_s_count.max() > 400
    b = _batch([w], debug_arrays=1)
    o = oracle.OracleWindow(w)
    c = o.linearize()
    b.begin()
    s = b.finish()[0]
    assert abs(s["final_cost"] - c) <= 1e-10 * c
    _close(b.array("IMU_RESIDUAL"), o.array("IMU_RESIDUAL"), 1e-8)
    sg = b.optimize(8)[0]
    sr = oracle.OracleWindow(w).optimize(8)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
    b.close()


def test_marginalisation_prior_evaluation(oracle):
    # a synthetic dense prior e = e0 + J dchi over two poses and one speed/bias block
    # (MarginalizationError::EvaluateWithMinimalJacobians, MarginalizationError.cpp:893-946)
    rng = np.random.default_rng(32)
    w = synthetic.small_window(seed=32, K=4, L=40)
    Dm = 6 + 9 + 6
    A = rng.standard_normal((Dm, Dm))
    w.marg_J = np.triu(A) * 3.0
    w.marg_e0 = rng.standard_normal(Dm) * 0.1
    w.marg_block_type = np.array([0, 1, 0], np.int32)
    w.marg_block_idx = np.array([0, 0, 1], np.int32)
    w.marg_block_off = np.array([0, 6, 15], np.int32)
    lin = np.zeros((3, 9))
    lin[0, :7] = synthetic.pose_oplus(w.pose[0], rng.normal(0, 0.02, 6))
    lin[1] = w.sb[0] + rng.normal(0, 0.01, 9)
    lin[2, :7] = synthetic.pose_oplus(w.pose[1], rng.normal(0, 0.02, 6))
    w.marg_lin = lin
    for exact in (1,):
        b = _batch([w], debug_arrays=1)
        o = oracle.OracleWindow(w)
        o.set_marg_exact(exact)
        c = o.linearize()
        b.begin()
        s = b.finish()[0]
        assert abs(s["final_cost"] - c) <= 1e-11 * c
        b.begin(); b.iterate(1)
        opt = default_options()
        assert o.solve(opt.initial_radius, opt) == 0
        _close(b.array("REDUCED_S"), o.array("REDUCED_S"), 1e-12)
        _close(b.array("STEP"), o.array("STEP"), 1e-7)
        sg = b.optimize(10)[0]
        sr = o.optimize(10)
        assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"]
        b.close()


def test_large_reduced_system_uses_hbm_resident_solve(oracle):
    # D = 300 > 174: the block matrix of the reduced system lives in HBM/L2 (solve_kernel<true>)
    w = synthetic.make_window(20, 200, 1.0, seed=33, frame_dt=0.1)
    assert w.reduced_dim() == 300
    b = _batch([w], debug_arrays=1)
    o = oracle.OracleWindow(w)
    o.linearize()
    b.begin(); b.iterate(1)
    opt = default_options()
    assert o.solve(opt.initial_radius, opt) == 0
    _close(b.array("REDUCED_S"), o.array("REDUCED_S"), 1e-12)
    _close(b.array("STEP"), o.array("STEP"), 1e-7)
    for n in (3, 12):
        sg = _batch([w]).optimize(n)[0]
        sr = oracle.OracleWindow(w).optimize(n)
        assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"], (n, sg, sr)
        assert sg["iterations"] == sr["iterations"] and sg["successful_steps"] == sr["successful_steps"]
    b.close()


def test_config_C_full_size(oracle):
    # BASELINE configs[2]: 50 keyframes / 2000 landmarks / 200 000 observations, D = 750
    w = synthetic.config_C()
    assert w.n_obs == 200000 and w.reduced_dim() == 750
    b = _batch([w])
    sg = b.optimize(10)[0]
    sr = oracle.OracleWindow(w).optimize(10)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-9 * sr["final_cost"], (sg, sr)
    assert (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"])
    pg, sbg, lg = b.get_state()
    o = oracle.OracleWindow(w)
    o.optimize(10)
    pr, sbr, lr = o.get_state()
    assert np.abs(pg - pr).max() < 1e-7 and np.abs(sbg - sbr).max() < 1e-7 and np.abs(lg - lr).max() < 1e-6
    # mixed batch: a small and a large window side by side (both solve instantiations in one launch pair)
    ws = [synthetic.small_window(seed=34), synthetic.make_window(20, 100, 1.0, seed=35, frame_dt=0.1)]
    bb = _batch(ws)
    s2 = bb.optimize(6)
    for i, wi in enumerate(ws):
        r = oracle.OracleWindow(wi).optimize(6)
        assert abs(s2[i]["final_cost"] - r["final_cost"]) <= 1e-9 * r["final_cost"]
    b.close(); bb.close()


def test_prior_numbers_set_after_the_hand_over():
    """okvis_ba_set_marg_prior_values: a window uploaded with the blocks of its marginalisation prior and stand-in numbers, the
    real J and e0 set afterwards, is the window uploaded with them — same optimisation bit for bit, same container — also for a
    prior whose H0 = J^T J is formed on the device (more than 128 rows); a window without a prior refuses the call."""
    import copy
    import ctypes as C
    from okvis_amd import solver
    dp = C.POINTER(C.c_double)
    for K, nposeb, nsbb in ((6, 4, 2), (26, 16, 10)):     # 42 rows | 186 rows
        w = synthetic.make_window(K, 120, 0.8, seed=70 + K, frame_dt=0.25)
        rng = np.random.default_rng(K)
        bt = [0] * nposeb + [1] * nsbb
        bi = list(range(nposeb)) + list(range(nsbb))
        off, bo = 0, []
        for t in bt:
            bo.append(off); off += 6 if t == 0 else 9
        w.marg_block_type, w.marg_block_idx, w.marg_block_off = (np.array(a, np.int32) for a in (bt, bi, bo))
        w.marg_J = np.triu(rng.standard_normal((off, off))) * 3.0
        w.marg_e0 = rng.standard_normal(off) * 0.01
        w.marg_lin = np.array([np.r_[w.pose[i], 0, 0] if t == 0 else w.sb[i] for t, i in zip(bt, bi)])
        a = solver.WindowBatch([w], options=default_options(), patchable=True)
        standin = copy.deepcopy(w)
        standin.marg_J = np.zeros_like(w.marg_J); standin.marg_e0 = np.zeros_like(w.marg_e0)
        b = solver.WindowBatch([standin], options=default_options(), patchable=True)
        J, e0 = np.ascontiguousarray(w.marg_J), np.ascontiguousarray(w.marg_e0)
        assert b._L.okvis_ba_set_marg_prior_values(b._h, 0, J.ctypes.data_as(dp), e0.ctypes.data_as(dp)) == 0
        sa, sb_ = a.optimize(5)[0], b.optimize(5)[0]
        assert sa == sb_ and sa["iterations"] > 0
        for x, y in zip(a.get_state(0), b.get_state(0)):
            assert np.array_equal(x, y)
        v = b.patched_view(0)
        assert np.array_equal(v.marg_J, w.marg_J) and np.array_equal(v.marg_e0, w.marg_e0)
        a.close(); b.close()
    plain = solver.WindowBatch([synthetic.small_window(seed=3)], options=default_options())
    z = np.zeros(4)
    assert plain._L.okvis_ba_set_marg_prior_values(plain._h, 0, z.ctypes.data_as(dp), z.ctypes.data_as(dp)) == -1
    plain.close()


def test_imu_preintegration_travels_with_the_term():
    """The reference's ImuError object keeps its preintegration between optimize() calls.  A window can hand over the record
    okvis_ba_fetch_imu_caches gave out for the same term (flag 2): nothing is re-preintegrated on first use, and the numbers are the
    ones the rebuild at the reference bias (flag 1) produces — bit for bit, through a whole optimisation; a bias beyond the threshold
    still re-preintegrates (ImuError.cpp:549); a record of a term that was never evaluated is refused."""
    w = synthetic.small_window(seed=62, K=5, L=60)
    n = w.n_imu
    b0 = _batch([w], debug_arrays=1)
    b0.optimize(5)
    res = b0.fetch_results(0)
    caches = b0.fetch_imu_caches(0)
    assert caches.shape == (n, 290) and np.array_equal(caches[:, 280:289], res["imu_sb_ref"])   # (the record ends with its reference bias)
    b0.close()
    import copy
    nxt = copy.deepcopy(w)
    nxt.pose, nxt.sb, nxt.lm = res["pose"], res["sb"], res["lm"]
    nxt.imu_sb_ref = res["imu_sb_ref"]
    runs = {}
    for flag in (1, 2):
        wi = copy.deepcopy(nxt)
        wi.imu_sb_ref_valid = np.full(n, flag, np.uint8)
        if flag == 2:
            wi.imu_cache = caches
        b = _batch([wi], debug_arrays=1)
        b.begin()
        redo_first = b.array("IMU_REDO_COUNT").copy()
        lin = b.array("IMU_RESIDUAL").copy()
        b.finish()
        s = b.optimize(6)[0]
        runs[flag] = (redo_first, lin, s, b.get_state(0), b.fetch_imu_caches(0))
        b.close()
    assert np.array_equal(runs[1][0], np.ones(n)) and np.array_equal(runs[2][0], np.zeros(n))   # rebuilt at the reference | kept
    assert np.array_equal(runs[1][1], runs[2][1]) and runs[1][2] == runs[2][2]
    for a, c in zip(runs[1][3], runs[2][3]):
        assert np.array_equal(a, c)
    assert np.array_equal(runs[1][4][:, :289], runs[2][4][:, :289])       # (the last double holds the flag word and the redo counter)
    # a gyro bias beyond the threshold re-preintegrates although the record is there
    far = copy.deepcopy(nxt)
    far.imu_sb_ref_valid = np.full(n, 2, np.uint8); far.imu_cache = caches
    far.sb = far.sb.copy(); far.sb[:, 3:6] += 5e-3
    b = _batch([far], debug_arrays=1)
    b.begin()
    assert np.array_equal(b.array("IMU_REDO_COUNT"), np.ones(n))
    b.finish(); b.close()
    # a record that was never filled (flag word 0) must not travel with flag 2
    bad = copy.deepcopy(nxt)
    bad.imu_sb_ref_valid = np.full(n, 2, np.uint8); bad.imu_cache = np.zeros((n, 290))
    with pytest.raises(Exception):
        _batch([bad])


def test_inherited_imu_reference_bias(oracle):
    """ImuError's preintegration cache survives optimize() calls in the reference (speedAndBiases_ref_): a window
    can hand over the reference bias of every factor.  Within the 1e-4 threshold the factor is evaluated with the
    first-order bias correction around that reference (no re-preintegration at the current bias); beyond it the
    cache is rebuilt (ImuError.cpp:541-558)."""
    w = synthetic.small_window(seed=61, K=4, L=40)
    n = w.n_imu
    for scale, expect_same_ref in ((2e-5, True), (5e-3, False)):
        ref = np.array([w.sb[w.imu_sb0[f]] for f in range(n)])
        ref[:, 3:6] += scale * np.array([1.0, -1.0, 0.5])
        w.imu_sb_ref = ref
        w.imu_sb_ref_valid = np.ones(n, np.uint8)
        b = _batch([w], debug_arrays=1)
        o = oracle.OracleWindow(w)
        c_ref = o.linearize()
        b.begin()
        s = b.finish()[0]
        assert abs(s["final_cost"] - c_ref) <= 1e-11 * c_ref
        _close(b.array("IMU_RESIDUAL"), o.array("IMU_RESIDUAL"), 1e-8)
        gref = b.array("IMU_SB_REF").reshape(n, 9)
        _close(gref, o.array("IMU_SB_REF").reshape(n, 9), 1e-15)
        assert np.array_equal(gref, ref) == expect_same_ref
        # and it changes the numbers: a fresh factor (no reference) evaluates differently when the reference is kept
        w2 = synthetic.small_window(seed=61, K=4, L=40)
        c_fresh = oracle.OracleWindow(w2).linearize()
        assert (c_fresh != c_ref) == expect_same_ref
        so = o.optimize(6)
        sg = b.optimize(6)[0]
        assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
        b.close()
