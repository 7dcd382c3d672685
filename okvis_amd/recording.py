"""ASL-folder recordings for the backend replay (okvis_amd/csrc/host/replay.hpp).

`write_synthetic_recording` writes what a EuRoC sequence plus the OKVIS frontend would hand to the backend — IMU CSV, camera
`sensor.yaml`s, ground truth, and the recorded tracks (frames / landmarks / observations) — from the analytic truth
trajectory of `synthetic.py`, with the raggedness of real data: landmarks enter and leave the images as the sensor moves,
detections drop out at random, tracks start on a stereo match and continue monocular, each image keeps at most
`max_keypoints` keypoints, landmark estimates start with depth-dependent triangulation noise.

`probe` / `run_replay` call the C++ readers and the replay loop (`okvis_replay_probe`, `okvis_replay_run` of
libokvis_amd_estimator.so); no dataset logic lives in Python.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import synthetic
from .window import DIST_RADTAN, ImuParams

_DP = C.POINTER(C.c_double)


def _real(x):
    """repr with a '.' in the mantissa ("4.0e-06"): a real for cv::FileStorage either way, and for YAML 1.1 readers only so"""
    t = repr(float(x))
    m, _, e = t.partition("e")
    return (m if "." in m else m + ".0") + ("e" + e if e else "")


def _yaml_list(v):
    return "[" + ", ".join(repr(float(x)) for x in v) + "]"


def write_synthetic_recording(path, duration_s=10.0, frame_rate_hz=10.0, imu_rate_hz=200.0, n_points=1100, seed=3,
                              detect_prob=0.85, max_keypoints=220, pixel_noise=0.7, keyframe_every=4,
                              gyro_bias=(0.003, -0.002, 0.001), acc_bias=(0.02, -0.015, 0.01), start_s=0.25,
                              ids_by_first_sighting=False):
    """Returns a dict with the truth (frame times, poses) and counts.  Deterministic in `seed`.  ids_by_first_sighting: landmark ids
    grow with the time a track starts (what OKVIS' IdProvider does) instead of following the point cloud."""
    rng = np.random.default_rng(seed)
    prm = ImuParams()
    for d in ("imu0", "cam0", "cam1", "state_groundtruth_estimate0", "okvis_amd_tracks"):
        os.makedirs(os.path.join(path, d), exist_ok=True)
    t0_ns = 1_403_636_579_000_000_000  # EuRoC-like epoch: exercises the int64 path of the readers
    # ---- IMU ----
    dt = 1.0 / imu_rate_hz
    n_imu = int(round((duration_s + start_s + 0.1) * imu_rate_hz))
    ts = np.arange(n_imu) * dt
    t_imu_ns = t0_ns + np.round(ts * 1e9).astype(np.int64)
    gyr, acc = np.zeros((n_imu, 3)), np.zeros((n_imu, 3))
    gt = np.zeros((n_imu, 16))
    g_W = np.array([0.0, 0.0, prm.g])
    for j in range(n_imu):
        p, v, a, R, w = synthetic.truth_at(ts[j])
        gyr[j] = w + gyro_bias
        acc[j] = R.T @ (a + g_W) + acc_bias
        q = synthetic.rot_to_quat(R)  # xyzw
        gt[j] = np.r_[p, q[3], q[0], q[1], q[2], v, gyro_bias, acc_bias]
    gyr += rng.standard_normal((n_imu, 3)) * prm.sigma_g_c / np.sqrt(dt)
    acc += rng.standard_normal((n_imu, 3)) * prm.sigma_a_c / np.sqrt(dt)
    with open(os.path.join(path, "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        for j in range(n_imu):
            f.write(f"{t_imu_ns[j]}," + ",".join(repr(float(x)) for x in np.r_[gyr[j], acc[j]]) + "\n")
    with open(os.path.join(path, "imu0", "sensor.yaml"), "w") as f:
        f.write("#Default imu sensor yaml file\nsensor_type: imu\ncomment: synthetic\n\n# Sensor extrinsics wrt. the body-frame.\nT_BS:\n"
                "  cols: 4\n  rows: 4\n  data: [1.0, 0.0, 0.0, 0.0,\n         0.0, 1.0, 0.0, 0.0,\n         0.0, 0.0, 1.0, 0.0,\n         0.0, 0.0, 0.0, 1.0]\n"
                f"rate_hz: {int(imu_rate_hz)}\n\n# inertial sensor noise model parameters (static)\n"
                f"gyroscope_noise_density: {prm.sigma_g_c!r}     # [ rad / s / sqrt(Hz) ]\n"
                f"gyroscope_random_walk: {prm.sigma_gw_c!r}       # [ rad / s^2 / sqrt(Hz) ]\n"
                f"accelerometer_noise_density: {prm.sigma_a_c!r}  # [ m / s^2 / sqrt(Hz) ]\n"
                f"accelerometer_random_walk: {prm.sigma_aw_c!r}    # [ m / s^3 / sqrt(Hz) ]\n")
    with open(os.path.join(path, "state_groundtruth_estimate0", "data.csv"), "w") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], v_RS_R_x [m s^-1], "
                "v_RS_R_y [m s^-1], v_RS_R_z [m s^-1], b_w_RS_S_x [rad s^-1], b_w_RS_S_y [rad s^-1], b_w_RS_S_z [rad s^-1], "
                "b_a_RS_S_x [m s^-2], b_a_RS_S_y [m s^-2], b_a_RS_S_z [m s^-2]\n")
        for j in range(n_imu):
            f.write(f"{t_imu_ns[j]}," + ",".join(repr(float(x)) for x in gt[j]) + "\n")
    # ---- cameras ----
    intr = synthetic.EUROC_INTR
    for c in range(2):
        T = synthetic.EUROC_T_SC[c]
        rows = [", ".join(repr(float(x)) for x in T[r]) for r in range(4)]
        with open(os.path.join(path, f"cam{c}", "sensor.yaml"), "w") as f:
            f.write("# General sensor definitions.\nsensor_type: camera\ncomment: synthetic\n\n# Sensor extrinsics wrt. the body-frame.\nT_BS:\n"
                    "  cols: 4\n  rows: 4\n  data: [" + ",\n         ".join(rows) + "]\n\n# Camera specific definitions.\nrate_hz: "
                    f"{int(frame_rate_hz)}\nresolution: [{synthetic.IMAGE_W}, {synthetic.IMAGE_H}]\ncamera_model: pinhole\n"
                    f"intrinsics: {_yaml_list(intr[c, :4])} #fu, fv, cu, cv\ndistortion_model: radial-tangential\n"
                    f"distortion_coefficients: {_yaml_list(intr[c, 4:8])}\n")
    # ---- frames ----
    n_frames = int(round(duration_s * frame_rate_hz))
    t_frame = start_s + np.arange(n_frames) / frame_rate_hz + 0.0017   # not on an IMU sample
    t_frame_ns = t0_ns + np.round(t_frame * 1e9).astype(np.int64)
    t_frame = (t_frame_ns - t0_ns) * 1e-9
    frame_ids = 1000 + 3 * np.arange(n_frames)                          # increasing, not contiguous (IdProvider is shared)
    keyframe = (np.arange(n_frames) % keyframe_every) == 0
    # ---- a room of points in front of the (oscillating) sensor; what each camera sees in each frame ----
    pts = np.stack([rng.uniform(2.0, 14.0, n_points), rng.uniform(-9.0, 9.0, n_points), rng.uniform(-4.0, 4.0, n_points)], 1)
    vis = np.zeros((n_points, n_frames, 2), bool)
    uvs = np.zeros((n_points, n_frames, 2, 2))
    poses = np.zeros((n_frames, 7))
    R_frames = []
    for k in range(n_frames):
        p, v, a, R, w = synthetic.truth_at(t_frame[k])
        poses[k] = np.r_[p, synthetic.rot_to_quat(R)]
        R_frames.append(R)
        p_S = (pts - p) @ R
        for c in range(2):
            Rc, tc = synthetic.EUROC_T_SC[c][:3, :3], synthetic.EUROC_T_SC[c][:3, 3]
            uv, ok = synthetic.project_points(intr[c], DIST_RADTAN, (p_S - tc) @ Rc)
            vis[:, k, c] = ok & (rng.uniform(size=n_points) < detect_prob)
            uvs[:, k, c] = uv
    started = np.zeros(n_points, bool)
    lid_of = 5000 + np.arange(n_points)
    n_started = 0
    obs_rows, lm_rows = [], []
    n_obs_frame = []
    for k in range(n_frames):
        new = ~started & vis[:, k, 0] & vis[:, k, 1]          # a track starts on a stereo match
        for l in np.flatnonzero(new):
            p_S = R_frames[k].T @ (pts[l] - poses[k, :3])                   # in the sensor frame of the triangulating frame
            depth = np.linalg.norm(p_S)
            if ids_by_first_sighting:
                lid_of[l] = 5000 + n_started
                n_started += 1
            lm_rows.append((lid_of[l], t_frame_ns[k], p_S + p_S / depth * rng.standard_normal() * 0.004 * depth ** 2
                            + rng.standard_normal(3) * 0.002 * depth))
        started |= new
        cnt = 0
        for c in range(2):
            cand = np.flatnonzero(started & vis[:, k, c])
            if cand.size > max_keypoints:                      # the detector's keypoint budget
                cand = np.sort(rng.choice(cand, max_keypoints, replace=False))
            for l in rng.permutation(cand):                    # any order inside a frame
                m = (uvs[l, k, c] + rng.standard_normal(2) * pixel_noise).astype(np.float32)
                obs_rows.append((t_frame_ns[k], c, m[0], m[1], 8.0 if l % 3 else 12.0, lid_of[l]))
                cnt += 1
        n_obs_frame.append(cnt)
    with open(os.path.join(path, "okvis_amd_tracks", "frames.csv"), "w") as f:
        f.write("#timestamp [ns],frame_id,is_keyframe\n")
        for k in range(n_frames):
            f.write(f"{t_frame_ns[k]},{frame_ids[k]},{int(keyframe[k])}\n")
    with open(os.path.join(path, "okvis_amd_tracks", "landmarks.csv"), "w") as f:
        f.write("#landmark_id,timestamp [ns],x_S,y_S,z_S,w\n")
        for lid, t, p in lm_rows:
            f.write(f"{lid},{t}," + ",".join(repr(float(x)) for x in p) + ",1.0\n")
    with open(os.path.join(path, "okvis_amd_tracks", "observations.csv"), "w") as f:
        f.write("#timestamp [ns],cam,u,v,size,landmark_id\n")
        for t, c, u, v, s, lid in obs_rows:
            f.write(f"{t},{c},{float(u)!r},{float(v)!r},{s},{lid}\n")
    return dict(t_frame_ns=t_frame_ns, frame_ids=frame_ids, keyframe=keyframe, poses=poses, n_imu=n_imu, n_frames=n_frames,
                n_observations=len(obs_rows), n_landmarks=len(lm_rows), observations_per_frame=np.array(n_obs_frame),
                first_imu=np.r_[float(t_imu_ns[0]), gyr[0], acc[0]])


def write_okvis_config(file, frame_rate_hz=10.0, imu_rate_hz=200.0, num_keyframes=5, num_imu_frames=3, min_iterations=3,
                       max_iterations=10, time_limit=0.035, sigma_absolute_translation=0.0, sigma_absolute_orientation=0.0,
                       prm=None, distortion_type="radialtangential"):
    """A configuration file in the format of the reference's applications (`okvis_app_synchronous <config> <dataset>`; the keys of
    reference config/config_fpga_p2_euroc.yaml) for the synthetic rig of `write_synthetic_recording`: OpenCV-FileStorage YAML 1.0,
    the cameras as a block sequence of flow mappings whose T_SC spans lines.  Returns the values written."""
    prm = prm or ImuParams()
    intr = synthetic.EUROC_INTR
    lines = ["%YAML:1.0", "cameras:"]
    for c in range(2):
        T = synthetic.EUROC_T_SC[c]
        rows = [", ".join(repr(float(x)) for x in T[r]) for r in range(4)]
        lines += ["     - {T_SC:", "        [ " + ",\n          ".join(rows) + "],",
                  f"        image_dimension: [{synthetic.IMAGE_W}, {synthetic.IMAGE_H}],",
                  f"        distortion_coefficients: {_yaml_list(intr[c, 4:8])},",
                  f"        distortion_type: {distortion_type},",
                  f"        focal_length: {_yaml_list(intr[c, 0:2])},",
                  f"        principal_point: {_yaml_list(intr[c, 2:4])}}}", ""]
    lines += ["", "camera_params:",
              f"    camera_rate: {int(frame_rate_hz)} # frames per second expected",
              f"    sigma_absolute_translation: {_real(sigma_absolute_translation)} # [m]",
              f"    sigma_absolute_orientation: {_real(sigma_absolute_orientation)} # [rad]",
              "    sigma_c_relative_translation: 0.0 # [m]",
              "    sigma_c_relative_orientation: 0.0 # [rad]",
              "    timestamp_tolerance: 0.005 # [s]", "",
              "imu_params:",
              f"    a_max: {_real(prm.a_max)} # [m/s^2]", f"    g_max: {_real(prm.g_max)} # [rad/s]",
              f"    sigma_g_c: {_real(prm.sigma_g_c)}", f"    sigma_a_c: {_real(prm.sigma_a_c)}",
              f"    sigma_bg: {_real(prm.sigma_bg)}", f"    sigma_ba: {_real(prm.sigma_ba)}",
              f"    sigma_gw_c: {_real(prm.sigma_gw_c)}", f"    sigma_aw_c: {_real(prm.sigma_aw_c)}",
              f"    tau: {_real(getattr(prm, 'tau', 3600.0))}", f"    g: {_real(prm.g)}", "    a0: [ 0.0, 0.0, 0.0 ]",
              f"    imu_rate: {int(imu_rate_hz)}", "    # transform Body-Sensor (IMU)", "    T_BS:",
              "        [1.0000, 0.0000, 0.0000, 0.0000,", "         0.0000, 1.0000, 0.0000, 0.0000,",
              "         0.0000, 0.0000, 1.0000, 0.0000,", "         0.0000, 0.0000, 0.0000, 1.0000]", "",
              "# Estimator parameters", f"numKeyframes: {int(num_keyframes)} # keyframes in the window",
              f"numImuFrames: {int(num_imu_frames)} # frames linked by the most recent IMU terms", "",
              "ceres_options:", f"    minIterations: {int(min_iterations)}   # always performed",
              f"    maxIterations: {int(max_iterations)}  # never more", f"    timeLimit: {_real(time_limit)}   # [s]", "",
              "detection_options:", "    threshold: 40.0", "    octaves: 0", "    maxNoKeypoints: 400", "",
              "imageDelay: 0.0  # [s]", "", "displayImages: false", "useDriver: false", "",
              "publishing_options:", "    publish_rate: 200", "    publishLandmarks: true", "    trackedBodyFrame: B",
              "    velocitiesFrame: Wc", ""]
    with open(file, "w") as f:
        f.write("\n".join(lines))
    return dict(num_keyframes=num_keyframes, num_imu_frames=num_imu_frames, min_iterations=min_iterations, max_iterations=max_iterations,
                time_limit=time_limit, camera_rate=int(frame_rate_hz), imu_rate=int(imu_rate_hz))


def write_image_folders(path, t_frame_ns, extra_before=2, period_ns=50_000_000):
    """<path>/cam<i>/data/<t_ns>.png (empty files: the backend replay only reads the names, as okvis_app_synchronous.cpp:264-318 does
    before it decodes them) and <path>/cam<i>/data.csv: one image per recorded frame plus a few before the first frame."""
    times = [int(t_frame_ns[0]) - (k + 1) * period_ns for k in range(extra_before)][::-1] + [int(t) for t in t_frame_ns]
    for c in range(2):
        d = os.path.join(path, f"cam{c}", "data")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(path, f"cam{c}", "data.csv"), "w") as f:
            f.write("#timestamp [ns],filename\n")
            for t in times:
                open(os.path.join(d, f"{t}.png"), "w").close()
                f.write(f"{t},{t}.png\n")
    return np.array(times, np.int64)


def _host_lib():
    from . import estimator
    return estimator.lib()._L


def probe(path, imu_as_float=True):
    """Counts and first records as the C++ readers see the folder (no GPU needed)."""
    L = _host_lib()
    L.okvis_replay_probe.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), _DP, _DP, _DP, C.POINTER(C.c_int), _DP]
    L.okvis_est_last_error.restype = C.c_char_p
    counts = (C.c_longlong * 6)()
    imu, T, intr, prm = np.zeros(7), np.zeros(7), np.zeros(12), np.zeros(4)
    model = C.c_int()
    ok = L.okvis_replay_probe(os.fsencode(path), int(imu_as_float), counts, imu.ctypes.data_as(_DP), T.ctypes.data_as(_DP),
                              intr.ctypes.data_as(_DP), C.byref(model), prm.ctypes.data_as(_DP))
    if ok < 0:
        raise RuntimeError(L.okvis_est_last_error().decode())
    return dict(n_imu=counts[0], n_cameras=counts[1], n_ground_truth=counts[2], n_frames=counts[3], n_observations=counts[4],
                n_landmarks=counts[5], first_imu=imu, cam0_T_SC=T, cam0_intr=intr, cam0_model=model.value, imu_noise=prm)


def run_replay(path, device=0, num_keyframes=5, num_imu_frames=3, num_iterations=10, num_threads=2, max_frames=0,
               min_observations_per_landmark=0, imu_as_float=True, imu_overlap=0.02, trajectory_csv=None):
    """okvis_replay_run: a fresh okvis_amd::Estimator over the whole recording (needs the GPU)."""
    L = _host_lib()
    L.okvis_replay_run.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, _DP]
    L.okvis_est_last_error.restype = C.c_char_p
    opts = (C.c_int * 7)(num_keyframes, num_imu_frames, num_iterations, num_threads, max_frames, min_observations_per_landmark,
                         int(imu_as_float))
    stats = np.zeros(8)
    ok = L.okvis_replay_run(os.fsencode(path), device, opts, imu_overlap,
                            os.fsencode(trajectory_csv) if trajectory_csv else None, stats.ctypes.data_as(_DP))
    if ok < 0:
        raise RuntimeError(L.okvis_est_last_error().decode())
    return dict(frames=int(stats[0]), landmarks_removed=int(stats[1]), has_ground_truth=bool(stats[2]), rms_position=stats[3],
                final_position=stats[4], final_rotation=stats[5], ms_optimize=stats[6], ms_marginalize=stats[7])


def _check(L, ok):
    if ok < 0:
        L.okvis_est_last_error.restype = C.c_char_p
        raise RuntimeError(L.okvis_est_last_error().decode())
    return ok


def yaml_to_json(text):
    """The document as the C++ parser of okvis_config.hpp sees it: nested lists / dicts, scalars as ("i" | "r" | "s", text)."""
    import json
    L = _host_lib()
    L.okvis_yaml_to_json.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    n = _check(L, L.okvis_yaml_to_json(text.encode(), None, 0))
    buf = C.create_string_buffer(n + 1)
    _check(L, L.okvis_yaml_to_json(text.encode(), buf, n + 1))

    def conv(x):
        if x is None:
            return None
        if isinstance(x, list):
            return [conv(e) for e in x]
        if "m" in x:
            return {k: conv(v) for k, v in x["m"].items()}
        (kind, val), = x.items()
        return (kind, val)
    return conv(json.loads(buf.value.decode()))


def read_config(file, max_cameras=4):
    """okvis_config_read: the values of a reference configuration file that reach the backend."""
    L = _host_lib()
    L.okvis_config_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), _DP]
    ints = (C.c_int * (8 + 3 * max_cameras))()
    reals = np.zeros(36 + 19 * max_cameras)
    n = _check(L, L.okvis_config_read(os.fsencode(file), max_cameras, ints, reals.ctypes.data_as(_DP)))
    cams = [dict(width=ints[8 + 3 * k], height=ints[9 + 3 * k], model=ints[10 + 3 * k], T_SC=reals[36 + 19 * k:43 + 19 * k].copy(),
                 intr=reals[43 + 19 * k:55 + 19 * k].copy()) for k in range(min(n, max_cameras))]
    return dict(num_keyframes=ints[0], num_imu_frames=ints[1], min_iterations=ints[2], max_iterations=ints[3], camera_rate=ints[4],
                imu_rate=ints[5], n_cameras=n, time_limit=reals[0], image_delay=reals[1], timestamp_tolerance=reals[2],
                extrinsics=reals[3:7].copy(), imu=reals[7:20].copy(), T_BS=reals[20:36].reshape(4, 4).copy(), cameras=cams)


def list_images(path, cam):
    L = _host_lib()
    L.okvis_asl_list_images.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.c_int]
    n = _check(L, L.okvis_asl_list_images(os.fsencode(path), cam, None, 0))
    t = (C.c_longlong * max(n, 1))()
    _check(L, L.okvis_asl_list_images(os.fsencode(path), cam, t, n))
    return np.array(t[:n], np.int64)


def read_image_csv(file):
    L = _host_lib()
    L.okvis_asl_read_image_csv.argtypes = [C.c_char_p, C.POINTER(C.c_longlong), C.c_int]
    n = _check(L, L.okvis_asl_read_image_csv(os.fsencode(file), None, 0))
    t = (C.c_longlong * max(n, 1))()
    _check(L, L.okvis_asl_read_image_csv(os.fsencode(file), t, n))
    return np.array(t[:n], np.int64)


def probe_config(path, config, imu_as_float=True):
    """The folder as readRecording(path, config) sees it: calibration and IMU parameters from the configuration file."""
    L = _host_lib()
    L.okvis_replay_probe_config.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_longlong), _DP, _DP, C.POINTER(C.c_int), _DP, _DP]
    counts = (C.c_longlong * 6)()
    T, intr, prm, ext = np.zeros(7), np.zeros(12), np.zeros(13), np.zeros(4)
    model = C.c_int()
    _check(L, L.okvis_replay_probe_config(os.fsencode(path), os.fsencode(config), int(imu_as_float), counts, T.ctypes.data_as(_DP),
                                          intr.ctypes.data_as(_DP), C.byref(model), prm.ctypes.data_as(_DP), ext.ctypes.data_as(_DP)))
    return dict(n_imu=counts[0], n_cameras=counts[1], n_ground_truth=counts[2], n_frames=counts[3], n_observations=counts[4],
                n_landmarks=counts[5], cam0_T_SC=T, cam0_intr=intr, cam0_model=model.value, imu=prm, extrinsics=ext)


def run_replay_config(path, config, device=0, num_keyframes=-1, num_imu_frames=-1, num_iterations=-1, num_threads=-1, max_frames=0,
                      min_observations_per_landmark=0, imu_as_float=True, imu_overlap=0.02, use_time_limit=False, trajectory_csv=None):
    """okvis_replay_run_config: `okvis_app_synchronous <config> <dataset folder>` on the backend (needs the GPU); values < 0 keep the
    configuration file's."""
    L = _host_lib()
    L.okvis_replay_run_config.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_int, C.c_char_p, _DP]
    opts = (C.c_int * 7)(num_keyframes, num_imu_frames, num_iterations, num_threads, max_frames, min_observations_per_landmark,
                         int(imu_as_float))
    stats = np.zeros(8)
    _check(L, L.okvis_replay_run_config(os.fsencode(path), os.fsencode(config), device, opts, imu_overlap, int(use_time_limit),
                                        os.fsencode(trajectory_csv) if trajectory_csv else None, stats.ctypes.data_as(_DP)))
    return dict(frames=int(stats[0]), landmarks_removed=int(stats[1]), has_ground_truth=bool(stats[2]), rms_position=stats[3],
                final_position=stats[4], final_rotation=stats[5], ms_optimize=stats[6], ms_marginalize=stats[7])
