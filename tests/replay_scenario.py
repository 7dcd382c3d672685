"""The frame loop of okvis_amd/csrc/host/replay.cpp::replay (what ThreadedKFVio does per frame, ThreadedKFVio.cpp:501-533 /
736-765) stated over the flat estimator surface of okvis_amd.estimator, so that the SAME recording can be driven through the
reference's own okvis::Estimator (tests/ref_lib.RefEstimator) and through the MI355X backend, call for call.  The inputs are
what the C++ readers read (okvis_replay_read), not a second parse of the files."""
import ctypes as C
import os

import numpy as np

from okvis_amd import recording


def read(path, imu_as_float=True):
    info = recording.probe(path, imu_as_float)
    L = recording._host_lib()
    n_imu, n_cam, n_fr, n_obs, n_lm = (info[k] for k in ("n_imu", "n_cameras", "n_frames", "n_observations", "n_landmarks"))
    r = dict(imu_t=np.zeros(n_imu, np.int64), imu_ga=np.zeros((n_imu, 6)), T_SC=np.zeros((n_cam, 7)), intr=np.zeros((n_cam, 12)),
             model=np.zeros(n_cam, np.int32), imu_params=np.zeros(13), frames=np.zeros((n_fr, 3), np.int64),
             obs_i=np.zeros((n_obs, 3), np.int64), obs_f=np.zeros((n_obs, 3), np.float32), lm_i=np.zeros((n_lm, 2), np.int64),
             lm_hp=np.zeros((n_lm, 4)))
    L.okvis_replay_read.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 11
    order = ("imu_t", "imu_ga", "T_SC", "intr", "model", "imu_params", "frames", "obs_i", "obs_f", "lm_i", "lm_hp")
    if L.okvis_replay_read(os.fsencode(path), int(imu_as_float), *[r[k].ctypes.data for k in order]) < 0:
        raise RuntimeError(L.okvis_est_last_error().decode())
    return r


def _rotate(q, v):   # q = xyzw
    x, y, z, w = q
    t = 2 * np.array([y * v[2] - z * v[1], z * v[0] - x * v[2], x * v[1] - y * v[0]])
    return v + w * t + np.array([y * t[2] - z * t[1], z * t[0] - x * t[2], x * t[1] - y * t[0]])


def replay(rec, make_estimator, make_frame, num_keyframes=5, num_imu_frames=3, num_iterations=10, max_frames=0,
           imu_overlap=0.02, sample_landmarks=5):
    """One dict per frame: pose / speed-bias of the newest frame, window composition, removed landmark ids, prior size,
    the states of every frame in the window and every `sample_landmarks`-th landmark alive."""
    est = make_estimator()
    for _ in range(len(rec["T_SC"])):
        est.addCamera(0, 0, 0, 0)                     # fixed extrinsics (EuRoC config)
    est.addImu(rec["imu_params"])
    lm_row = {int(i): k for k, i in enumerate(rec["lm_i"][:, 0])}
    frame_id_at = {int(t): int(i) for t, i, _ in rec["frames"]}
    overlap = int(round(imu_overlap * 1e9))
    gone, added, trace, keep = set(), set(), [], []
    obs_at, last_t = 0, 0
    imu_t, obs_i, obs_f = rec["imu_t"], rec["obs_i"], rec["obs_f"]
    n_frames = min(max_frames, len(rec["frames"])) if max_frames > 0 else len(rec["frames"])
    for k in range(n_frames):
        t_ns, fid, kf = (int(x) for x in rec["frames"][k])
        f = make_frame(fid, t_ns, rec["T_SC"], rec["intr"], list(rec["model"]))
        keep.append(f)
        t0, t1 = (last_t if k else t_ns) - overlap, t_ns + overlap
        lo, hi = np.searchsorted(imu_t, t0, "left"), np.searchsorted(imu_t, t1, "right")
        if not est.addStates(f, imu_t[lo:hi], rec["imu_ga"][lo:hi, :3], rec["imu_ga"][lo:hi, 3:], bool(kf)):
            raise RuntimeError(f"addStates failed at frame {fid}")
        last_t = t_ns
        in_window = {est.frameIdByAge(a) for a in range(est.numFrames())}
        n_obs = 0
        while obs_at < len(obs_i) and obs_i[obs_at, 0] <= t_ns:
            ot, cam, lid = (int(x) for x in obs_i[obs_at])
            u, v, size = obs_f[obs_at]
            obs_at += 1
            if ot != t_ns or lid in gone:
                continue
            if lid not in added:
                row = lm_row[lid]
                src = frame_id_at[int(rec["lm_i"][row, 1])]
                if src not in in_window:
                    continue                           # its frame has left the window: the track is not started
                T = est.get_T_WS(src)                  # hp_W = T_WS(frame of the triangulation, current estimate) hp_S
                hp = rec["lm_hp"][row]
                est.addLandmark(lid, np.r_[_rotate(T[3:], hp[:3]) + hp[3] * T[:3], hp[3]])
                added.add(lid)
            kp = f.add_keypoint(cam, float(u), float(v), float(size))
            if est.addObservation(lid, fid, cam, kp) != 0:
                n_obs += 1
        s = est.optimize(num_iterations, 2, False)
        removed = []
        est.applyMarginalizationStrategy(num_keyframes, num_imu_frames, removed)
        gone.update(removed)
        added.difference_update(removed)
        r = dict(frame=k, id=fid, T_WS=est.get_T_WS(fid), sb=est.getSpeedAndBias(fid), n_obs=n_obs, summary=s,
                 removed=sorted(removed), n_frames=est.numFrames(), n_landmarks=est.numLandmarks(), prior=est.priorInfo(),
                 poses={}, sbs={}, landmarks={})
        for age in range(est.numFrames()):
            i = est.frameIdByAge(age)
            r["poses"][i] = est.get_T_WS(i)
            if est.isInImuWindow(i):
                r["sbs"][i] = est.getSpeedAndBias(i)
        for lid in sorted(added)[::sample_landmarks]:
            r["landmarks"][lid] = est.getLandmark(lid)[0]
        trace.append(r)
    est.close()
    return trace
