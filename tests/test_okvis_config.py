"""CPU: the reader of the reference's configuration file (okvis_amd/csrc/host/okvis_config.cpp; what
okvis_common/src/VioParametersReader.cpp reads of config/config_fpga_p2_euroc.yaml) and the EuRoC image enumeration of
okvis_app_synchronous.cpp:264-318 — the YAML subset against PyYAML, the values against what was written, the reference's defaults
and refusals, and readRecording(path, config).  No GPU."""
import os
import shutil

import numpy as np
import pytest
import yaml

from okvis_amd import recording, synthetic


def _plain(x):
    """the C++ parser's tree with scalars typed the way cv::FileNode types them"""
    if x is None:
        return None
    if isinstance(x, list):
        return [_plain(e) for e in x]
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    kind, text = x
    if kind == "i":
        return int(text)
    if kind == "r":
        return float(text)
    return {"true": True, "false": False}.get(text, text)


def _pyyaml(text):
    return yaml.safe_load("\n".join(l for l in text.split("\n") if not l.startswith("%")))


DOCS = {
    "block_maps": "a: 1\nb:\n  c: 2.5\n  d:\n    e: x\n    f: -3\ng: last\n",
    "block_sequence_of_scalars": "k:\n  - 1\n  - 2.0\n  - three\nz: 0\n",
    "sequence_in_the_key_column": "k:\n- 1\n- 2\nz: 0\n",
    "sequence_of_block_maps": "cams:\n  - name: a\n    v: [1, 2]\n  - name: b\n    v: [3, 4]\nn: 2\n",
    "flow_in_flow": "m: {a: [1, [2, 3], {b: 4.5}], c: {d: e}}\n",
    "flow_over_lines": "T:\n   [1.0, 0.0,\n    0.0, 1.0]\nm: {a:\n     [1,\n      2], b: 3}\n",
    "comments_and_quotes": "# head\na: 'x # not a comment'   # a comment\nb: \"y: z\"\n\n# between\nc: 4 # four\n",
    "top_level_sequence": "- 1\n- {a: 2}\n- [3, 4]\n",
    "empty_value": "a:\nb: 1\n",
    "crlf": "a: 1\r\nb:\r\n  c: [1,\r\n      2]\r\n",
}


@pytest.mark.parametrize("name", sorted(DOCS))
def test_yaml_subset_agrees_with_pyyaml(name):
    assert _plain(recording.yaml_to_json(DOCS[name])) == _pyyaml(DOCS[name])


def test_scalar_types_follow_cv_filenode():
    t = recording.yaml_to_json("a: 176\nb: 176.0\nc: 12.0e-4\nd: 1e5\ne: radialtangential\nf: '5'\ng: -3\nh: .5\ni: 0x10\n")
    assert {k: v[0] for k, v in t.items()} == dict(a="i", b="r", c="r", d="r", e="s", f="s", g="i", h="r", i="s")


@pytest.mark.parametrize("text, what", [
    ("a: [1, 2\nb: 3\n", "expected in a flow sequence"),
    ("a: {b: 1\n", "unterminated"),
    ("a: 1\n   b: 2\n", "bad indentation"),
    ("a: 1\na: 2\n", "appears twice"),
    ("a: [1, 2] x\n", "unexpected text after a flow collection"),
    ("a:\n  - 1\n    - 2\n", "bad indentation"),
])
def test_malformed_documents_are_refused_with_the_line(text, what):
    with pytest.raises(RuntimeError, match="<text>:[0-9]+: .*" + what):
        recording.yaml_to_json(text)


@pytest.fixture(scope="module")
def config(tmp_path_factory):
    f = str(tmp_path_factory.mktemp("cfg") / "config.yaml")
    return f, recording.write_okvis_config(f, frame_rate_hz=10, num_keyframes=4, num_imu_frames=2, min_iterations=2, max_iterations=7,
                                           time_limit=0.02, sigma_absolute_translation=1e-3, sigma_absolute_orientation=2e-3)


def test_whole_configuration_parses_like_pyyaml(config):
    text = open(config[0]).read()
    assert _plain(recording.yaml_to_json(text)) == _pyyaml(text)


def test_values_that_reach_the_backend(config):
    c = recording.read_config(config[0])
    assert (c["num_keyframes"], c["num_imu_frames"], c["min_iterations"], c["max_iterations"], c["camera_rate"], c["imu_rate"]) == (4, 2, 2, 7, 10, 200)
    assert c["time_limit"] == 0.02 and c["image_delay"] == 0.0 and c["timestamp_tolerance"] == 0.005
    assert np.array_equal(c["extrinsics"], [1e-3, 2e-3, 0, 0])
    # a_max g_max sigma_g_c sigma_a_c sigma_bg sigma_ba sigma_gw_c sigma_aw_c tau g a0
    assert np.array_equal(c["imu"], [176.0, 7.8, 12e-4, 8e-3, 0.03, 0.1, 4e-6, 4e-5, 3600.0, 9.81007, 0, 0, 0])
    assert np.array_equal(c["T_BS"], np.eye(4))
    assert c["n_cameras"] == 2
    for k in range(2):
        cam = c["cameras"][k]
        T = synthetic.T_to_pose(synthetic.EUROC_T_SC[k])
        assert (cam["width"], cam["height"], cam["model"]) == (synthetic.IMAGE_W, synthetic.IMAGE_H, 1)
        assert np.allclose(cam["T_SC"][:3], T[:3], atol=1e-15)
        assert min(np.abs(cam["T_SC"][3:] - T[3:]).max(), np.abs(cam["T_SC"][3:] + T[3:]).max()) < 1e-12
        assert np.array_equal(cam["intr"][:8], synthetic.EUROC_INTR[k, :8]) and not cam["intr"][8:].any()


REFERENCE_CONFIG = "/root/reference/config/config_fpga_p2_euroc.yaml"


@pytest.mark.skipif(not os.path.exists(REFERENCE_CONFIG), reason="the reference tree is not here")
def test_the_references_own_configuration_file():
    text = open(REFERENCE_CONFIG).read()
    ours, theirs = _plain(recording.yaml_to_json(text)), _pyyaml(text)
    assert ours == theirs
    c = recording.read_config(REFERENCE_CONFIG)
    assert (c["num_keyframes"], c["num_imu_frames"], c["min_iterations"], c["max_iterations"]) == (
        theirs["numKeyframes"], theirs["numImuFrames"], theirs["ceres_options"]["minIterations"], theirs["ceres_options"]["maxIterations"])
    assert c["time_limit"] == theirs["ceres_options"]["timeLimit"] and c["camera_rate"] == theirs["camera_params"]["camera_rate"]
    ip = theirs["imu_params"]
    assert np.array_equal(c["imu"], [ip[k] for k in ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c",
                                                      "tau", "g")] + list(ip["a0"]))
    assert c["n_cameras"] == len(theirs["cameras"]) == 2
    for cam, ref in zip(c["cameras"], theirs["cameras"]):
        T = np.array(ref["T_SC"]).reshape(4, 4)
        assert np.array_equal(cam["T_SC"][:3], T[:3, 3])
        x, y, z, w = cam["T_SC"][3:]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.abs(R - T[:3, :3]).max() < 1e-9   # the file's rotation is orthonormal to its printed digits
        assert np.array_equal(cam["intr"][:8], ref["focal_length"] + ref["principal_point"] + ref["distortion_coefficients"])
        assert (cam["width"], cam["height"]) == tuple(ref["image_dimension"]) and cam["model"] == 1


def _edited(config, tmp_path, fn):
    f = str(tmp_path / "edited.yaml")
    lines = open(config[0]).read().split("\n")
    with open(f, "w") as o:
        o.write("\n".join(fn(lines)))
    return f


def test_defaults_of_the_reference(config, tmp_path):
    """VioParametersReader.cpp:88-128 (5 / 2 / 1 / 10 / no limit), :197-201 (0.2 / camera_rate), :205-236 (0.0)"""
    drop = ("numKeyframes", "numImuFrames", "ceres_options", "minIterations", "maxIterations", "timeLimit", "timestamp_tolerance", "sigma_absolute")
    f = _edited(config, tmp_path, lambda L: [l for l in L if not l.strip().startswith(drop)])
    c = recording.read_config(f)
    assert (c["num_keyframes"], c["num_imu_frames"], c["min_iterations"], c["max_iterations"], c["time_limit"]) == (5, 2, 1, 10, -1.0)
    assert c["timestamp_tolerance"] == 0.2 / 10 and not c["extrinsics"].any()


@pytest.mark.parametrize("edit, message", [
    (lambda L: [l for l in L if "a_max" not in l], "'imu_params: a_max' parameter missing in configuration file."),
    # cv::FileNode::isReal() is false for "176": the reference's assertion fires, so does this reader
    (lambda L: [l.replace("a_max: 176.0", "a_max: 176") for l in L], "'imu_params: a_max' parameter missing in configuration file."),
    (lambda L: [l for l in L if "imu_rate" not in l], "'imu_params: imu_rate' parameter missing"),
    (lambda L: [l for l in L if "a0:" not in l], "'imu_params: a0' parameter missing"),
    (lambda L: [l for l in L if "imageDelay" not in l], "'imageDelay' parameter missing"),
    (lambda L: [l for l in L if "camera_rate" not in l], "'camera_params: camera_rate' parameter missing"),
    (lambda L: [l.replace("timestamp_tolerance: 0.005", "timestamp_tolerance: 0.06") for l in L], "larger than half the time between frames"),
    (lambda L: [l for l in L if "principal_point" not in l or L.index(l) > 20] , "incomplete calibration in configuration file for camera 0"),
    (lambda L: [l.replace("distortion_type: radialtangential", "distortion_type: fisheye") for l in L], "unrecognized distortion type fisheye"),
    (lambda L: [l.replace("T_BS:", "T_BX:") for l in L], "'T_BS' parameter missing in the configuration file or in the wrong format."),
])
def test_refusals_carry_the_references_wording(config, tmp_path, edit, message):
    f = _edited(config, tmp_path, edit)
    if "incomplete" in message:   # dropping the last entry of camera 0's flow mapping: close the mapping on the line before
        lines = open(f).read().split("\n")
        k = next(i for i, l in enumerate(lines) if "focal_length" in l)
        lines[k] = lines[k].rstrip(",") + "}"
        open(f, "w").write("\n".join(lines))
    with pytest.raises(RuntimeError, match=message.replace("(", r"\(")):
        recording.read_config(f)


def test_missing_file():
    with pytest.raises(RuntimeError, match="Could not open config file"):
        recording.read_config("/nonexistent/config.yaml")


def test_equidistant_and_eight_coefficient_models(tmp_path):
    f = str(tmp_path / "c.yaml")
    recording.write_okvis_config(f, distortion_type="equidistant")
    assert [c["model"] for c in recording.read_config(f)["cameras"]] == [2, 2]   # OKVIS_BA_DIST_EQUIDISTANT
    recording.write_okvis_config(f, distortion_type="radialtangential8")
    with pytest.raises(RuntimeError, match="needs 8 distortion_coefficients"):
        recording.read_config(f)
    recording.write_okvis_config(f, distortion_type="plumb_bob")
    assert [c["model"] for c in recording.read_config(f)["cameras"]] == [1, 1]


# ---- the dataset side ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dataset(tmp_path_factory, config):
    d = str(tmp_path_factory.mktemp("euroc"))
    info = recording.write_synthetic_recording(d, duration_s=1.2, n_points=250, seed=6)
    info["image_times"] = recording.write_image_folders(d, info["t_frame_ns"])
    return d, info


def test_image_folder_as_okvis_app_synchronous_lists_it(dataset):
    d, info = dataset
    os.makedirs(os.path.join(d, "cam0", "data", "a_directory"), exist_ok=True)   # directories are skipped (okvis_app_synchronous.cpp:268)
    for cam in range(2):
        t = recording.list_images(d, cam)
        assert np.array_equal(t, info["image_times"]) and np.all(np.diff(t) > 0)
        assert np.array_equal(recording.read_image_csv(os.path.join(d, f"cam{cam}", "data.csv")), info["image_times"])
    with pytest.raises(RuntimeError, match="cannot open the image folder"):
        recording.list_images(d, 2)


def test_image_names_must_be_timestamps(dataset, tmp_path):
    d = str(tmp_path / "bad")
    os.makedirs(os.path.join(d, "cam0", "data"))
    open(os.path.join(d, "cam0", "data", "frame_000000000001.png"), "w").close()
    with pytest.raises(RuntimeError, match="an image name must be"):
        recording.list_images(d, 0)


def test_recording_with_the_configuration_files_calibration(dataset, config, tmp_path):
    d = str(tmp_path / "copy")
    shutil.copytree(dataset[0], d)
    # okvis_app_synchronous never opens the sensor.yaml files: without them the configuration file is enough
    for s in ("cam0", "cam1", "imu0"):
        os.remove(os.path.join(d, s, "sensor.yaml"))
    with pytest.raises(RuntimeError, match="no camera calibration found"):
        recording.probe(d)
    p = recording.probe_config(d, config[0])
    info = dataset[1]
    assert (p["n_imu"], p["n_cameras"], p["n_frames"], p["n_observations"], p["n_landmarks"]) == (
        info["n_imu"], 2, info["n_frames"], info["n_observations"], info["n_landmarks"])
    assert np.array_equal(p["cam0_intr"][:8], synthetic.EUROC_INTR[0, :8]) and p["cam0_model"] == 1
    assert np.array_equal(p["extrinsics"], [1e-3, 2e-3, 0, 0]) and p["imu"][0] == 176.0
    # a recorded frame without its image (beyond timestamp_tolerance) is refused
    t = int(info["t_frame_ns"][3])
    os.remove(os.path.join(d, "cam1", "data", f"{t}.png"))
    with pytest.raises(RuntimeError, match="has no image of camera 1 within the timestamp tolerance"):
        recording.probe_config(d, config[0])
    # ... and one that is off by less than the tolerance (5 ms) is the frame's image
    open(os.path.join(d, "cam1", "data", f"{t + 3_000_000}.png"), "w").close()
    assert recording.probe_config(d, config[0])["n_frames"] == info["n_frames"]
