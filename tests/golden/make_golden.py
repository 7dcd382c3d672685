#!/usr/bin/env python3
"""Generate the committed fixtures tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  The reference (ethz-asl/okvis) holds no golden vectors for this path.  The vectors are
produced by the C++ oracle (`oracle/`) and every factor value is cross-checked, before it is written, against
(a) the reference's OWN classes compiled unmodified into oracle/_ref (tests/ref_lib.py; required to 1e-12 when the
reference tree is present — it is in the authoring container) and (b) the independent numpy/scipy statement in
`independent.py` (1e-11).  tests/test_oracle_vs_ref.py re-checks (a) on every CPU run.  The window-level vectors
are results of the DOGLEG policy (the default, Estimator.cpp:858).  Re-run with

    python tests/golden/make_golden.py

after a deliberate change of the oracle; `tests/test_golden.py` (CPU: oracle vs fixtures; GPU: HIP path vs
fixtures) fails on any unintended drift.
"""
from __future__ import annotations

import dataclasses
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import independent as ind  # noqa: E402
import oracle_lib as O  # noqa: E402
import ref_lib as R  # noqa: E402

HAVE_REF = R.available()
from okvis_amd import synthetic  # noqa: E402
from okvis_amd.window import ImuParams, Window, default_options  # noqa: E402

REL = 1e-11


def _agree(a, b, what):
    a, b = np.asarray(a, float), np.asarray(b, float)
    err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert err < REL, f"{what}: oracle and independent statement differ by {err:.3e}"
    return err


def rand_pose(rng, tmax=1.0, rmax=np.pi):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(-rmax, rmax)
    q = np.concatenate([np.sin(ang / 2) * ax, [np.cos(ang / 2)]])
    return np.concatenate([rng.uniform(-tmax, tmax, 3), q])


INTR = {  # PinholeCamera<D>::testObject constants (PinholeCamera.hpp:287-297 + each distortion's testObject)
    0: [350, 360, 378, 238, 0, 0, 0, 0, 0, 0, 0, 0],
    1: [350, 360, 378, 238, -0.16, 0.15, 0.0003, 0.0002, 0, 0, 0, 0],
    2: [350, 360, 378, 238, -0.21, 0.14, 0.0006, 0.0003, 0, 0, 0, 0],
    3: [350, 360, 378, 238, -0.16, 0.15, 0.0003, 0.0002, 0.01, 0.02, -0.01, 0.005],
}


def factor_vectors():
    rng = np.random.default_rng(20240923)
    out = {}
    # ---- reprojection: 8 generic cases per distortion model + the reference's edge cases ----
    cases = []
    for model in (0, 1, 2, 3):
        for _ in range(8):
            pose = rand_pose(rng)
            extr = rand_pose(rng, 0.2, 0.3)
            p_C = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(2, 12)])
            Rws, Rsc = ind.rot(pose[3:]), ind.rot(extr[3:])
            p_W = Rws @ (Rsc @ p_C + extr[:3]) + pose[:3]
            w = rng.choice([1.0, 0.5, 2.0])
            cases.append((model, pose, np.concatenate([p_W * w, [w]]), extr, rng.uniform(0, 700, 2),
                          rng.choice([1.0, 0.5, 8.0 / 6.0])))
    # negative w (PinholeCamera.hpp:363-369), a point closer than 0.2 m (ReprojectionError.hpp:143-151)
    m, pose, pt, extr, uv, sw = cases[9]
    cases.append((m, pose, -pt, extr, uv, sw))
    pose = rand_pose(rng)
    extr = np.array([0, 0, 0, 0, 0, 0, 1.0])
    p_W = ind.rot(pose[3:]) @ np.array([0.01, 0.02, 0.1]) + pose[:3]
    cases.append((1, pose, np.concatenate([p_W, [1.0]]), extr, np.array([300.0, 200.0]), 1.0))
    rp = {k: [] for k in ("model", "pose", "point", "extr", "intr", "uv", "sqrtw", "r", "Jp", "Jl", "Je", "valid")}
    for model, pose, pt, extr, uv, sw in cases:
        intr = np.array(INTR[model], float)
        r, Jp, Jl, Je, valid, _defined = O.reprojection(pose, pt, extr, intr, model, uv, sw * np.eye(2))
        _agree(r, ind.reprojection_residual(pose, pt, extr, intr, model, uv, sw), "reprojection residual")
        if HAVE_REF:
            rr = R.reprojection(pose, pt, extr, intr, model, uv, sw * np.eye(2))
            for a, b, n in zip((r, Jp, Jl, Je), rr, ("r", "Jp", "Jl", "Je")):
                assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), f"reprojection {n} vs reference"
        for k, v in zip(rp, (model, pose, pt, extr, intr, uv, sw, r, Jp, Jl, Je, valid)):
            rp[k].append(v)
    out.update({"reproj_" + k: np.array(v) for k, v in rp.items()})

    # ---- small priors ----
    pp = {k: [] for k in ("pose", "meas", "sqrtinfo", "r", "J")}
    for _ in range(4):
        pose, meas = rand_pose(rng), None
        meas = synthetic.pose_oplus(pose, rng.normal(size=6) * 0.05)
        A = rng.normal(size=(6, 6))
        si = O.sqrt_information(A @ A.T + np.eye(6))
        r, J = O.pose_error(pose, meas, si)
        _agree(r, ind.pose_error_residual(pose, meas, si), "pose error")
        for k, v in zip(pp, (pose, meas, si, r, J)):
            pp[k].append(v)
    out.update({"poseerr_" + k: np.array(v) for k, v in pp.items()})
    sp = {k: [] for k in ("sb", "meas", "sqrtinfo", "r", "J")}
    for _ in range(4):
        sb, meas = rng.normal(size=9), rng.normal(size=9)
        A = rng.normal(size=(9, 9))
        si = O.sqrt_information(A @ A.T + np.eye(9))
        r, J = O.speedbias_error(sb, meas, si)
        _agree(r, ind.speedbias_error_residual(sb, meas, si), "speed/bias error")
        for k, v in zip(sp, (sb, meas, si, r, J)):
            sp[k].append(v)
    out.update({"sberr_" + k: np.array(v) for k, v in sp.items()})
    rl = {k: [] for k in ("p0", "p1", "sqrtinfo", "r", "J0", "J1")}
    for _ in range(4):
        p0 = rand_pose(rng)
        p1 = synthetic.pose_oplus(p0, rng.normal(size=6) * 0.05)
        A = rng.normal(size=(6, 6))
        si = O.sqrt_information(A @ A.T + np.eye(6))
        r, J0, J1 = O.relative_pose_error(p0, p1, si)
        for k, v in zip(rl, (p0, p1, si, r, J0, J1)):
            rl[k].append(v)
    out.update({"relpose_" + k: np.array(v) for k, v in rl.items()})
    # first-pose prior weighting quirk (Estimator.cpp:240-242 through Eigen LLT, SURVEY.md §7 quirk a)
    info = np.diag([1e8, 1e8, 1e8, 0.0, 0.0, 1e8])
    out["firstpose_information"] = info
    out["firstpose_sqrtinfo"] = O.sqrt_information(info)

    # ---- IMU factor: fresh preintegration + evaluation; generic, unaligned end points, saturated ----
    prm = ImuParams()
    im = {k: [] for k in ("t0", "t1", "pose0", "sb0", "pose1", "sb1", "r", "J0", "J1", "J2", "J3", "sqrtinfo", "n")}
    streams = []
    for case in range(3):
        rate, dur = (200.0, 0.5) if case != 1 else (200.0, 0.1037)
        n = int(round((dur + 0.02) * rate)) + 2
        t = (np.arange(n) * (1e9 / rate)).astype(np.int64) + 1_000_000_000
        tt = (t - t[0]) * 1e-9
        gyr = np.stack([0.3 * np.sin(2 * tt), 0.2 * np.cos(3 * tt), 0.1 + 0.05 * tt], 1) + rng.normal(size=(n, 3)) * 1e-3
        acc = np.stack([0.5 * np.cos(tt), 0.4 * np.sin(2 * tt), 9.81 + 0.1 * np.sin(5 * tt)], 1) + rng.normal(size=(n, 3)) * 1e-2
        if case == 2:
            gyr[7, 1] = 9.0     # > g_max: sigma x100 for the two intervals touching it (ImuError.cpp:153-173)
            acc[20, 0] = 200.0  # > a_max
        t0 = int(t[0] + (2_500_000 if case == 1 else 0))
        t1 = int(t0 + round(dur * 1e9))
        pose0, pose1 = rand_pose(rng, 1.0, 0.5), None
        pose1 = synthetic.pose_oplus(pose0, np.concatenate([rng.normal(size=3) * 0.2, rng.normal(size=3) * 0.1]))
        sb0 = np.concatenate([rng.normal(size=3) * 0.5, rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.05])
        sb1 = sb0 + np.concatenate([rng.normal(size=3) * 0.1, rng.normal(size=6) * 1e-3])
        r, Js, si, cnt = O.imu_evaluate_fresh(t, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1)
        r2, L2, n2 = ind.imu_residual_fresh(t, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1)
        assert cnt == 1 and n2 >= 20, (cnt, n2)   # one re-preintegration; n2 = integration steps
        # L is a Cholesky factor of an inverse with cond ~1e10: compare the information it encodes and the residual
        _agree(si.T @ si / np.abs(si.T @ si).max(), L2.T @ L2 / np.abs(L2.T @ L2).max(), "IMU information")
        e = np.abs(r - r2).max() / max(1.0, np.abs(r2).max())
        assert e < 1e-10, f"IMU residual: {e:.3e}"   # measured ~1e-13 (covariance cond ~1e9)
        if HAVE_REF:
            rr, Jr, sir, _ = R.imu_evaluate_fresh(t, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1)
            assert np.abs(r - rr).max() <= 1e-12 * max(1.0, np.abs(rr).max()), "IMU residual vs reference"
            for a, b in zip(Js, Jr):
                assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), "IMU Jacobian vs reference"
        streams.append((t, gyr, acc))
        for k, v in zip(im, (t0, t1, pose0, sb0, pose1, sb1, r, *Js, si, cnt)):
            im[k].append(v)
    out.update({"imu_" + k: np.array(v) for k, v in im.items()})
    for i, (t, g, a) in enumerate(streams):
        out[f"imu{i}_t"], out[f"imu{i}_gyr"], out[f"imu{i}_acc"] = t, g, a
    return out


WINDOW_FIELDS = [f.name for f in dataclasses.fields(Window) if f.name not in ("imu_params", "meta", "cauchy_b")]


def pack_window(prefix, w, out):
    for k in WINDOW_FIELDS:
        out[prefix + k] = np.asarray(getattr(w, k))
    out[prefix + "cauchy_b"] = np.float64(w.cauchy_b)
    out[prefix + "imu_params"] = np.array([getattr(w.imu_params, f.name) for f in dataclasses.fields(ImuParams)])


def unpack_window(prefix, z):
    kw = {k: z[prefix + k] for k in WINDOW_FIELDS if prefix + k in z}   # fields added later keep their defaults
    prm = ImuParams(**{f.name: float(v) for f, v in zip(dataclasses.fields(ImuParams), z[prefix + "imu_params"])})
    return Window(cauchy_b=float(z[prefix + "cauchy_b"]), imu_params=prm, **kw)


def window_digest(w):
    h = hashlib.sha256()
    for k in WINDOW_FIELDS:
        h.update(np.ascontiguousarray(getattr(w, k)).tobytes())
    return h.hexdigest()


def run_oracle(w, iters, out, prefix):
    ow = O.OracleWindow(w)
    ow.linearize()
    out[prefix + "initial_cost"] = np.float64(ow.cost())
    out[prefix + "obs_residual0"] = ow.array("OBS_RESIDUAL")
    out[prefix + "imu_residual0"] = ow.array("IMU_RESIDUAL")
    s = ow.optimize(iters)
    pose, sb, lm = ow.get_state()
    out[prefix + "summary"] = np.array([s["iterations"], s["successful_steps"], s["termination"]], np.int64)
    out[prefix + "final_cost"] = np.float64(s["final_cost"])
    out[prefix + "final_pose"], out[prefix + "final_sb"], out[prefix + "final_lm"] = pose, sb, lm
    out[prefix + "lm_quality"] = ow.array("LM_QUALITY")
    return s


SMALL = [dict(seed=1, K=4, L=40, estimate_extrinsics="fixed"),
         dict(seed=2, K=4, L=40, estimate_extrinsics="shared"),
         dict(seed=3, K=5, L=30, estimate_extrinsics="perframe", cam_model=2),
         dict(seed=4, K=3, L=24, estimate_extrinsics="fixed", visibility=0.5)]


def window_vectors():
    out = {}
    for i, kw in enumerate(SMALL):
        w = synthetic.small_window(**kw)
        pack_window(f"w{i}_", w, out)
        s = run_oracle(w, 10, out, f"w{i}_")
        print(f"  small window {i}: {kw} -> cost {out[f'w{i}_initial_cost']:.6f} -> {s['final_cost']:.9f} "
              f"in {s['iterations']} it ({s['successful_steps']} ok, term {s['termination']})")
    out["n_small"] = np.int64(len(SMALL))
    # BASELINE configs[1] at full size: inputs are regenerated from the seed (2 MB would not be a "small
    # fixture"); the digest pins the generator, the scalars pin the solve.
    w = synthetic.config_A()
    o2 = {}
    s = run_oracle(w, 10, o2, "A_")
    out["A_digest"] = np.array(window_digest(w))
    for k in ("A_initial_cost", "A_final_cost", "A_summary", "A_final_pose", "A_final_sb"):
        out[k] = o2[k]
    print(f"  config A: cost {out['A_initial_cost']:.6f} -> {s['final_cost']:.9f} in {s['iterations']} it")
    return out


MARG = [dict(seed=71, K=5, L=40, estimate_extrinsics="fixed", pose=[0], sb=[0, 1]),
        dict(seed=72, K=4, L=30, estimate_extrinsics="shared", pose=[0], sb=[0]),
        dict(seed=73, K=4, L=24, estimate_extrinsics="perframe", pose=[0, 4, 5], sb=[0]),
        # no first-pose prior: the kept block is singular along the gauge directions (rank < dim), as every prior of
        # the running pipeline is — the case the GPU decides with its pivoted-Cholesky path
        dict(seed=74, K=5, L=40, estimate_extrinsics="fixed", pose=[0], sb=[0], drop_pose_prior=True)]


def marg_flags(w, kw):
    pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8)
    pm[kw["pose"]] = 1; sm[kw["sb"]] = 1
    return pm, sm


def marg_window(kw):
    """window + flags of one MARG case"""
    kw = dict(kw)
    pose, sb, drop = kw.pop("pose"), kw.pop("sb"), kw.pop("drop_pose_prior", False)
    w = synthetic.small_window(**kw)
    if drop:
        w.pprior_pose = np.zeros(0, np.int32); w.pprior_meas = np.zeros((0, 7)); w.pprior_sqrtinfo = np.zeros((0, 36))
    return w, marg_flags(w, dict(pose=pose, sb=sb))


def marginalization_vectors():
    """MarginalizationError numerics: oracle output, cross-checked against the scipy-eigh statement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_marginalization import numpy_marginalize
    out = {"n": np.int64(len(MARG))}
    for i, kw in enumerate(MARG):
        w, (pm, sm) = marg_window(kw)
        r = O.OracleWindow(w).marginalize(pm, sm)
        Hn, bn = numpy_marginalize(O.OracleWindow(w), w, pm, sm)
        _agree(r["H"] / np.abs(Hn).max(), Hn / np.abs(Hn).max(), "marginalised H")
        assert np.abs(r["b0"] - bn).max() <= 1e-9 * np.abs(bn).max(), "marginalised b0"
        Jn, e0n, rank = ind.error_computation(Hn, bn)
        assert rank == r["rank"]
        for k in ("H", "b0", "block_type", "block_idx", "block_off"):
            out[f"m{i}_{k}"] = r[k]
        out[f"m{i}_JtJ"] = r["J"].T @ r["J"]
        out[f"m{i}_Jte0"] = r["J"].T @ r["e0"]
        out[f"m{i}_rank"] = np.int64(r["rank"])
        out[f"m{i}_digest"] = np.array(window_digest(w))
        print(f"  marginalisation {i}: {kw} -> dim {r['dim']} rank {r['rank']}")
    return out


def main():
    f = factor_vectors()
    np.savez_compressed(os.path.join(HERE, "factors.npz"), **f)
    w = window_vectors()
    np.savez_compressed(os.path.join(HERE, "windows.npz"), **w)
    m = marginalization_vectors()
    np.savez_compressed(os.path.join(HERE, "marginalization.npz"), **m)
    for n in ("factors.npz", "windows.npz", "marginalization.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
