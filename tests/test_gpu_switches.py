"""GPU tests of solver routes chosen by okvis_ba_options::tuning (ABI 7; environment variables read once per process up to ABI 6, which
is why these comparisons used to need two processes): the same windows through two solvers of one process, bit for bit."""
import numpy as np
import pytest

from okvis_amd import solver, synthetic
from okvis_amd.window import TUNE_NO_EARLY_PREINTEGRATION, default_options

pytestmark = pytest.mark.gpu


def _run(n, frames, flags, set_state=False):
    """n windows of `frames` frames (frames - 1 IMU terms each) through optimize() in both trust-region modes"""
    wins = [synthetic.make_window(frames, 120 + 10 * i, 0.7, 9_300_000 + i) for i in range(n)]
    res = {}
    for mode in ("dogleg", "gn"):
        o = default_options()
        o.gauss_newton = 1 if mode == "gn" else 0
        o.tuning.flags = flags
        b = solver.WindowBatch(wins, device=0, options=o)
        if set_state:   # the biases change between the upload and the first evaluation
            for w in range(n):
                sb = b.get_state(w)[1].copy()
                sb[:, 3:6] += 2.0e-3 * (1 + w % 3)
                sb[:, 6:9] -= 1.0e-2
                b.set_state(w, sb=sb)
        sm = b.optimize(12)
        st = [b.get_state(w) for w in range(n)]
        res[mode + "_cost"] = np.array([x["final_cost"] for x in sm])
        res[mode + "_iter"] = np.array([x["iterations"] for x in sm])
        res[mode + "_succ"] = np.array([x["successful_steps"] for x in sm])
        for k, name in enumerate(("pose", "sb", "lm")):
            res[f"{mode}_{name}"] = np.concatenate([x[k].reshape(-1) for x in st])
        res[mode + "_redo"] = np.concatenate([np.asarray(b.array("IMU_REDO_COUNT", w)).reshape(-1) for w in range(n)])
        b.close()
    return res


@pytest.mark.parametrize("windows,frames", [(1, 6), (2, 4), (4, 3)])
def test_first_preintegration_started_at_upload_changes_nothing(windows, frames):
    """okvis_ba_upload starts the first preintegration of up to eight new IMU terms before it builds the index lists
    (imu_pre_kernel); OKVIS_BA_TUNE_NO_EARLY_PREINTEGRATION leaves it to the first linearise launch.  Same arithmetic at the same
    bias: every state and every summary of a DOGLEG and a fixed-radius optimisation is the same bit for bit, also with the terms
    spread over windows."""
    a = _run(windows, frames, 0)
    b = _run(windows, frames, TUNE_NO_EARLY_PREINTEGRATION)
    assert sorted(a) == sorted(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["dogleg_iter"].min() > 0


@pytest.mark.parametrize("windows,frames", [(1, 6), (3, 4)])
def test_bias_set_between_upload_and_first_evaluation(windows, frames):
    """okvis_ba_set_state(sb) between okvis_ba_upload and the first evaluation: the preintegration started at upload was built
    at the uploaded bias and must not stand for the one of the first evaluation (redo_ = true, ImuError.cpp:62, integrates at the
    bias it is evaluated at).  Every state, every summary and every count of preintegrations equals the run that leaves the
    first preintegration to the first linearise launch, bit for bit."""
    a = _run(windows, frames, 0, set_state=True)
    b = _run(windows, frames, TUNE_NO_EARLY_PREINTEGRATION, set_state=True)
    assert sorted(a) == sorted(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["dogleg_iter"].min() > 0 and a["dogleg_redo"].min() >= 1
