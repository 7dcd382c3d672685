"""diagnostics (not a test): the bounds of the pivoted-Cholesky rank decision on gauge-deficient marginalisations"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from okvis_amd import solver, synthetic
from okvis_amd.window import Window, default_options
import oracle_lib
def flags(w, poses=(), sbs=()):
    pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8)
    pm[list(poses)] = 1; sm[list(sbs)] = 1
    return pm, sm
def run(w, pm, sm, prior=None, tag=""):
    o = default_options(); o.debug_arrays = 2
    b = solver.WindowBatch([w], options=o)
    g = b.marginalize(0, pm, sm, prior)
    p = b.array("PROF")
    b.close()
    r = oracle_lib.OracleWindow(w).marginalize(pm, sm, prior)
    H = r["H"]; n = H.shape[0]; d = np.diag(H); sc = np.where(d > 1e-9, np.sqrt(np.abs(d)), 1e-3)
    ev = np.linalg.eigvalsh(H / np.outer(sc, sc)); tau = np.finfo(float).eps * n * ev.max()
    small = ev[ev < 1e4 * tau] / tau
    print(f"{tag:28s} dim {n:3d} rank gpu {g['rank']:3d} oracle {r['rank']:3d} sweeps {g['sweeps']}  kept/tau_hi {p[30]:9.3g} dropped-bound/tau_hi {p[31]:9.3g} "
          f"trace(R)/tau_hi {p[34]:8.3g} cw {p[35]:5.2f} r {int(p[32])}/{int(p[33])}  eigenvalues/tau below 1e4: {np.round(small, 2)}")
    return r
for seed in range(20):
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(4, 8)); Lm = int(rng.integers(12, 80))
    ext = ["fixed", "shared", "perframe"][int(rng.integers(0, 3))]
    w = synthetic.make_window(K, Lm, float(rng.uniform(0.4, 1.0)), seed=9100 + seed, estimate_extrinsics=ext)
    keep = [i for i in range(len(w.pprior_pose)) if w.pprior_pose[i] != 0]
    w.pprior_pose = w.pprior_pose[keep]; w.pprior_meas = w.pprior_meas[keep]; w.pprior_sqrtinfo = w.pprior_sqrtinfo[keep]
    n_p = int(rng.integers(1, 3))
    pm, sm = flags(w, list(range(n_p)), list(range(int(rng.integers(1, 3)))))
    if w.reduced_dim() > 174: continue
    r = run(w, pm, sm, tag=f"seed {seed} {ext} stage 1")
    if seed % 2: continue
    w2 = Window(pose=w.pose, pose_fixed=w.pose_fixed, sb=w.sb, sb_fixed=w.sb_fixed, lm=np.zeros((0, 4)), cam_intr=w.cam_intr, cam_model=w.cam_model,
                obs_lm=np.zeros(0, np.int32), obs_pose=np.zeros(0, np.int32), obs_ext=np.zeros(0, np.int32), obs_cam=np.zeros(0, np.int32),
                obs_uv=np.zeros((0, 2)), obs_sqrtw=np.zeros(0), imu_params=w.imu_params)
    prior = dict(block_type=r["block_type"], block_idx=r["block_idx"], H=r["H"], b0=r["b0"])
    in_prior = [int(i) for t, i in zip(r["block_type"], r["block_idx"]) if t == 0 and i < K]
    run(w2, *flags(w2, [in_prior[0]], []), prior=prior, tag=f"seed {seed} {ext} stage 2")
