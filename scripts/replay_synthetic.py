#!/usr/bin/env python3
"""Synthetic stand-in for the EuRoC replay (SURVEY.md §8f rank 4 needs OpenCV/BRISK and the dataset, both
absent): drives okvis_amd::Estimator frame by frame the way ThreadedKFVio does (ThreadedKFVio.cpp:501-533,
733-765) — addStates, addObservation for every visible landmark, optimize(numIter), applyMarginalizationStrategy(5, 3)
— and reports the per-frame backend latency INCLUDING window flattening, PCIe upload and download.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis_amd import estimator, synthetic
from okvis_amd.window import DIST_EQUIDISTANT, ImuParams


def main(n_frames=40, num_iter=10, grid=0.5, use_graph=None):
    rng = np.random.default_rng(11)
    IMU_RATE, FRAME_DT = 200.0, 0.25
    DT = 1.0 / IMU_RATE
    DURATION = n_frames * FRAME_DT
    prm = ImuParams(sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, g=9.81, g_max=1000.0, a_max=1000.0)
    speed = np.array([0.0, 1.0, 0.0])
    n_imu = int(DURATION * IMU_RATE) + 4
    t_imu = (np.arange(n_imu) * int(round(DT * 1e9))).astype(np.int64) + 1_000_000_000
    gyr = rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_g_c * np.sqrt(DT)
    acc = np.array([0, 0, prm.g]) + rng.uniform(-1, 1, (n_imu, 3)) * prm.sigma_a_c * np.sqrt(DT)
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0, 0.1, 0, 0, 0, 0, 1.0]])
    intr = np.stack([synthetic.TEST_INTR_EQUI, synthetic.TEST_INTR_EQUI])
    est = estimator.Estimator(0)
    if use_graph is not None:
        est.setUseGraph(use_graph)
    est.addCamera(0, 0, 0, 0); est.addCamera(0, 0, 0, 0)
    est.addImu(estimator.imu_param_vector(prm))
    pts = np.array([[3.0, y, z, 1.0] for y in np.arange(-6.0, DURATION + 6.0, grid) for z in np.arange(-4.0, 4.0 + 1e-9, grid)])
    ids = 5000 + np.arange(len(pts))
    added, gone = set(), set()
    rows, prev_t = [], None
    for k in range(n_frames):
        t_k = 1_000_000_000 + int(round(k * FRAME_DT * 1e9))
        r_k = speed * k * FRAME_DT
        f = estimator.Frame(100 + k, t_k, T_SC, intr, [DIST_EQUIDISTANT] * 2)
        lo = np.searchsorted(t_imu, (prev_t if k else t_k) - 20_000_000)
        hi = np.searchsorted(t_imu, t_k + 20_000_000) + 1
        t0 = time.perf_counter()
        assert est.addStates(f, t_imu[lo:hi], gyr[lo:hi], acc[lo:hi], k % 4 == 0)
        prev_t = t_k
        n_obs = 0
        for i in range(2):
            uv, ok = synthetic.project_points(intr[i], DIST_EQUIDISTANT, pts[:, :3] - r_k - T_SC[i, :3])
            near = ok & (np.abs(pts[:, 1] - r_k[1]) < 4.0)
            for j in np.flatnonzero(near):
                lid = int(ids[j])
                if lid in gone:
                    continue
                if lid not in added:
                    est.addLandmark(lid, pts[j] + np.r_[rng.normal(size=3) * 0.05, 0]); added.add(lid)
                m = uv[j] + rng.uniform(-1, 1, 2)
                est.addObservation(lid, f.id, i, f.add_keypoint(i, m[0], m[1], 8.0)); n_obs += 1
        t1 = time.perf_counter()
        s = est.optimize(num_iter, 2, False)
        t2 = time.perf_counter()
        tm = est.lastOptimizeTimings()
        removed = []
        est.applyMarginalizationStrategy(5, 3, removed)
        gone.update(removed)
        t3 = time.perf_counter()
        mi = est.lastMarginalizationInfo()
        rows.append(dict(frame=k, new_obs=n_obs, landmarks=est.numLandmarks(), frames=est.numFrames(), iterations=s["iterations"],
                         add_ms=(t1 - t0) * 1e3, optimize_ms=(t2 - t1) * 1e3, marginalize_ms=(t3 - t2) * 1e3, prior_dim=est.priorInfo()[0],
                         flatten_ms=tm[0], upload_ms=tm[1], iterate_ms=tm[2], download_ms=tm[3],
                         mflatten_ms=mi[0], mupload_ms=mi[1], mgpu_ms=mi[2], sweeps_v=mi[3], sweeps_h=mi[4], marg_D=mi[5]))
    T = est.get_T_WS(100 + n_frames - 1)
    err = float(np.linalg.norm(T[:3] - speed * (n_frames - 1) * FRAME_DT))
    est.close()
    steady = rows[10:]
    med = lambda key: float(np.median([r[key] for r in steady]))
    print(json.dumps({"use_graph": use_graph, "frames": n_frames, "num_iter": num_iter, "final_position_error_m": err,
                      "steady_state_median": {"observations_added_per_frame": med("new_obs"), "landmarks_in_window": med("landmarks"),
                                              "frames_in_window": med("frames"), "prior_dim": med("prior_dim"),
                                              "optimize_ms": med("optimize_ms"), "marginalize_ms": med("marginalize_ms"),
                                              "iterations": med("iterations"), "optimize_split_ms": {k: med(k + "_ms") for k in ("flatten", "upload", "iterate", "download")},
                                              "marginalize_split_ms": {k: med("m" + k + "_ms") for k in ("flatten", "upload", "gpu")},
                                              "marginalize_window_dim": med("marg_D"), "jacobi_sweeps": [med("sweeps_v"), med("sweeps_h")]},
                      "note": "wall clock around the C++ host calls: optimize() = flatten + okvis_ba_upload (host index build + H2D) + "
                              "iterations on the GPU + state/quality/IMU-reference download; addStates/addObservation time is dominated by the Python test harness"}))


if __name__ == "__main__":
    main(grid=float(sys.argv[1]) if len(sys.argv) > 1 else 0.5, use_graph=int(sys.argv[2]) if len(sys.argv) > 2 else None)
