"""GPU tests of solver routes that are chosen once per process (environment switches read at first use): the same windows through
two processes, the results compared bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, name, **env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r04_switch_check.py"), "run", out], check=True, env=e, timeout=300,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return np.load(out)


@pytest.mark.parametrize("windows,frames", [(1, 6), (2, 4), (4, 3)])
def test_first_preintegration_started_at_upload_changes_nothing(tmp_path, windows, frames):
    """okvis_ba_upload starts the first preintegration of up to eight new IMU terms before it builds the index lists
    (imu_pre_kernel); OKVIS_BA_NO_PRE leaves it to the first linearise launch.  Same arithmetic at the same bias: every state and
    every summary of a DOGLEG and a fixed-radius optimisation is the same bit for bit, also with the terms spread over windows."""
    a = _run(tmp_path, "pre", CHECK_WINDOWS=windows, CHECK_K=frames)
    b = _run(tmp_path, "nopre", CHECK_WINDOWS=windows, CHECK_K=frames, OKVIS_BA_NO_PRE=1)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["dogleg_iter"].min() > 0


@pytest.mark.parametrize("windows,frames", [(1, 6), (3, 4)])
def test_bias_set_between_upload_and_first_evaluation(tmp_path, windows, frames):
    """okvis_ba_set_state(sb) between okvis_ba_upload and the first evaluation: the preintegration started at upload was built
    at the uploaded bias and must not stand for the one of the first evaluation (redo_ = true, ImuError.cpp:62, integrates at the
    bias it is evaluated at).  Every state, every summary and every count of preintegrations equals the run that leaves the
    first preintegration to the first linearise launch, bit for bit."""
    a = _run(tmp_path, "pre_set", CHECK_WINDOWS=windows, CHECK_K=frames, CHECK_SET_STATE=1)
    b = _run(tmp_path, "nopre_set", CHECK_WINDOWS=windows, CHECK_K=frames, CHECK_SET_STATE=1, OKVIS_BA_NO_PRE=1)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["dogleg_iter"].min() > 0 and a["dogleg_redo"].min() >= 1
