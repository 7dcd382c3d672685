"""CPU: the ASL / recorded-track readers of okvis_amd/csrc/host/replay.cpp (what okvis_app_synchronous.cpp:233-379 parses,
plus the recorded frontend output) — values, the std::stof quirk, comment / blank / CRLF handling and every error path with
its file:line message.  No GPU: only the readers run (okvis_replay_probe)."""
import os
import shutil

import numpy as np
import pytest

from okvis_amd import recording, synthetic


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("asl"))
    info = recording.write_synthetic_recording(d, duration_s=1.5, n_points=300, seed=5)
    return d, info


def test_counts_and_values(folder):
    d, info = folder
    p = recording.probe(d)
    assert (p["n_imu"], p["n_cameras"], p["n_ground_truth"], p["n_frames"], p["n_observations"], p["n_landmarks"]) == \
           (info["n_imu"], 2, info["n_imu"], info["n_frames"], info["n_observations"], info["n_landmarks"])
    # okvis_app_synchronous.cpp:337-349 reads the six IMU values with std::stof: they arrive rounded to float
    assert p["first_imu"][0] == info["first_imu"][0]
    assert np.array_equal(p["first_imu"][1:], info["first_imu"][1:].astype(np.float32).astype(np.float64))
    assert not np.array_equal(p["first_imu"][1:], info["first_imu"][1:])
    q = recording.probe(d, imu_as_float=False)
    assert np.array_equal(q["first_imu"], info["first_imu"])
    # camera 0: T_BS -> (r, q xyzw), intrinsics + radial-tangential coefficients
    T = synthetic.T_to_pose(synthetic.EUROC_T_SC[0])
    assert np.allclose(p["cam0_T_SC"][:3], T[:3], atol=1e-15)
    assert min(np.abs(p["cam0_T_SC"][3:] - T[3:]).max(), np.abs(p["cam0_T_SC"][3:] + T[3:]).max()) < 1e-12
    assert np.array_equal(p["cam0_intr"][:8], synthetic.EUROC_INTR[0, :8]) and p["cam0_model"] == 1
    assert np.array_equal(p["imu_noise"], [12e-4, 4e-6, 8e-3, 4e-5])


def _copy(folder, tmp_path):
    d = str(tmp_path / "copy")
    shutil.copytree(folder[0], d)
    return d


def _edit(path, fn):
    with open(path) as f:
        lines = f.read().split("\n")
    lines = fn(lines)
    with open(path, "w") as f:
        f.write("\n".join(lines))


def test_comments_blank_lines_and_crlf(folder, tmp_path):
    d = _copy(folder, tmp_path)
    _edit(os.path.join(d, "imu0", "data.csv"), lambda L: [L[0], "", "# a comment", "   "] + [x + "\r" for x in L[1:]])
    _edit(os.path.join(d, "okvis_amd_tracks", "frames.csv"), lambda L: ["", "#x"] + L)
    assert recording.probe(d)["n_imu"] == folder[1]["n_imu"]


@pytest.mark.parametrize("case", ["imu_field", "imu_order", "imu_short", "imu_empty", "no_imu", "frames_order", "frame_ids",
                                  "obs_landmark", "obs_time", "obs_cam", "lm_twice", "lm_frame", "yaml_model", "yaml_list",
                                  "no_camera"])
def test_errors_name_file_and_line(folder, tmp_path, case):
    d = _copy(folder, tmp_path)
    imu = os.path.join(d, "imu0", "data.csv")
    tr = os.path.join(d, "okvis_amd_tracks")
    expect = None
    if case == "imu_field":
        _edit(imu, lambda L: L[:5] + [L[5].replace(",", ",x", 1)] + L[6:])
        expect = ("imu0/data.csv:6", "not a finite number")
    elif case == "imu_order":
        _edit(imu, lambda L: L[:3] + [L[2]] + L[3:])
        expect = ("imu0/data.csv:4", "timestamps must increase")
    elif case == "imu_short":
        _edit(imu, lambda L: L[:2] + [",".join(L[2].split(",")[:5])] + L[3:])
        expect = ("imu0/data.csv:3", "expected at least 7 fields")
    elif case == "imu_empty":
        _edit(imu, lambda L: L[:1])
        expect = ("imu0/data.csv", "no imu messages present")
    elif case == "no_imu":
        os.remove(imu)
        expect = ("imu0/data.csv", "cannot open")
    elif case == "frames_order":
        _edit(os.path.join(tr, "frames.csv"), lambda L: [L[0], L[2], L[1]] + L[3:])
        expect = ("frames.csv:3", "frame timestamps must increase")
    elif case == "frame_ids":
        def f(L):
            a = L[2].split(",")
            a[1] = L[1].split(",")[1]
            return L[:2] + [",".join(a)] + L[3:]
        _edit(os.path.join(tr, "frames.csv"), f)
        expect = ("frames.csv:3", "frame ids must increase")
    elif case == "obs_landmark":
        _edit(os.path.join(tr, "observations.csv"), lambda L: L[:1] + [",".join(L[1].split(",")[:5] + ["42"])] + L[2:])
        expect = ("observations.csv:2", "unknown landmark")
    elif case == "obs_time":
        _edit(os.path.join(tr, "observations.csv"), lambda L: L[:1] + ["7," + L[1].split(",", 1)[1]] + L[2:])
        expect = ("observations.csv:2", "no frame")
    elif case == "obs_cam":
        def f(L):
            a = L[1].split(",")
            a[1] = "2"
            return L[:1] + [",".join(a)] + L[2:]
        _edit(os.path.join(tr, "observations.csv"), f)
        expect = ("observations.csv:2", "camera index out of range")
    elif case == "lm_twice":
        _edit(os.path.join(tr, "landmarks.csv"), lambda L: L[:2] + [L[1]] + L[2:])
        expect = ("landmarks.csv:3", "appears twice")
    elif case == "lm_frame":
        def f(L):
            a = L[1].split(",")
            a[1] = "5"
            return L[:1] + [",".join(a)] + L[2:]
        _edit(os.path.join(tr, "landmarks.csv"), f)
        expect = ("landmarks.csv:2", "frame that does not exist")
    elif case == "yaml_model":
        _edit(os.path.join(d, "cam1", "sensor.yaml"), lambda L: [x.replace("radial-tangential", "fisheye624") for x in L])
        expect = ("cam1/sensor.yaml", "unknown distortion_model 'fisheye624'")
    elif case == "yaml_list":
        _edit(os.path.join(d, "cam0", "sensor.yaml"), lambda L: [x for x in L if not x.strip().startswith("intrinsics")])
        expect = ("cam0/sensor.yaml", "missing list 'intrinsics'")
    elif case == "no_camera":
        os.remove(os.path.join(d, "cam0", "sensor.yaml"))
        expect = ("cam0/sensor.yaml", "no camera calibration")
    with pytest.raises(RuntimeError) as e:
        recording.probe(d)
    assert expect[0] in str(e.value) and expect[1] in str(e.value), str(e.value)


def test_other_distortion_models_and_optional_files(folder, tmp_path):
    d = _copy(folder, tmp_path)
    _edit(os.path.join(d, "cam0", "sensor.yaml"), lambda L: [x.replace("radial-tangential", "equidistant") for x in L])
    os.remove(os.path.join(d, "imu0", "sensor.yaml"))                       # defaults of ImuParameters then
    shutil.rmtree(os.path.join(d, "state_groundtruth_estimate0"))
    p = recording.probe(d)
    assert p["cam0_model"] == 2 and p["n_ground_truth"] == 0
    assert np.array_equal(p["imu_noise"], [12e-4, 4e-6, 8e-3, 4e-5])
