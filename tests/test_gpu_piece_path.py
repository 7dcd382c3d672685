"""GPU parity of the piece path of the linearise launch (okvis_amd/csrc/ba_linearize2.hpp; reference algebra:
okvis_ceres/include/okvis/ceres/implementation/ReprojectionError.hpp:87-242) on the window shapes that exercise its piece
enumeration — the host (count_pieces, ba_capi.hip) and the kernel (DPP run detection) must cut the observations of a group into
the same pieces — and of its fall-back to the staged kernel:

  * monocular windows: every observation is a piece of its own, groups close at 128 pieces (half the lanes),
  * landmarks with odd observation counts: pairs start at odd lanes and straddle the 16-lane rows (one pair = two pieces),
  * three and four observations of one (landmark, pose): pieces of 2 + 1 and 2 + 2,
  * a landmark with more than 128 pieces: the batch goes back to the staged kernel,
  * the two kernels against each other on the same window.

Everything is compared with the CPU oracle through the C-ABI, like tests/test_gpu_parity.py."""
import copy

import numpy as np
import pytest

from okvis_amd import solver, synthetic
from okvis_amd.window import default_options

pytestmark = pytest.mark.gpu


def _keep(w, mask):
    for name in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_sqrtw", "obs_uv"):
        setattr(w, name, np.asarray(getattr(w, name))[mask].copy())
    return w


def _check(oracle, w, n=6, tol=1e-9, **opt):
    o = default_options()
    for k, v in opt.items():
        setattr(o, k, v)
    b = solver.WindowBatch([w], options=o)
    sg = b.optimize(n)[0]
    ow = oracle.OracleWindow(w)
    sr = ow.optimize(n, o)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= tol * sr["final_cost"], (sg, sr)
    assert (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"])
    pg, sbg, lg = b.get_state()
    pr, sbr, lr = ow.get_state()
    assert np.abs(pg - pr).max() < 1e-7 and np.abs(sbg - sbr).max() < 1e-7 and np.abs(lg - lr).max() < 1e-6
    b.close()
    return sg


def test_monocular_window(oracle):
    w = synthetic.make_window(8, 120, 0.9, seed=61)
    _keep(w, np.asarray(w.obs_cam) == 0)
    assert w.obs_lm.size > 500
    _check(oracle, w)
    _check(oracle, w, schur_lm_per_block=24)     # (separate Schur launch on the matrix core)


@pytest.mark.parametrize("seed", [62, 63, 64])
def test_odd_counts_and_rows_cut_through_pairs(oracle, seed):
    """random observations dropped: landmarks with 1, 3, 5 ... observations shift every later pair of the group by one lane"""
    rng = np.random.default_rng(seed)
    w = synthetic.make_window(7, 90, 0.8, seed=seed)
    _keep(w, rng.random(w.obs_lm.size) > 0.23)
    counts = np.bincount(np.asarray(w.obs_lm), minlength=w.lm.shape[0])
    assert (counts % 2 == 1).sum() > 20
    _check(oracle, w)
    _check(oracle, w, gauss_newton=1, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0, reserved0=4)


def test_three_and_four_observations_of_one_pair(oracle):
    w = synthetic.small_window(seed=65, K=5, L=50)
    rng = np.random.default_rng(65)
    n = w.obs_lm.size
    extra = np.concatenate([rng.choice(n, 30, replace=False), rng.choice(n, 30, replace=False), rng.choice(n, 20, replace=False)])
    order = np.sort(np.concatenate([np.arange(n), extra]))
    uv = np.asarray(w.obs_uv)[order].copy()
    is_dup = np.r_[False, order[1:] == order[:-1]]
    uv[is_dup] += rng.uniform(-1.5, 1.5, (int(is_dup.sum()), 2))
    for name in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_sqrtw"):
        setattr(w, name, np.asarray(getattr(w, name))[order].copy())
    w.obs_uv = uv
    runs = np.diff(np.flatnonzero(np.r_[True, (np.diff(w.obs_lm) != 0) | (np.diff(w.obs_pose) != 0), True]))
    assert (runs >= 3).sum() > 10 and (runs >= 4).sum() > 2
    _check(oracle, w, n=8)


def test_landmark_beyond_128_pieces_falls_back_to_the_staged_kernel(oracle):
    """140 keyframes seen by one camera, all but the last six fixed (D stays small): a landmark seen from more than 128 poses has
    more than 128 pieces, which the piece path refuses at upload; the batch then runs through ba_linearize.hpp"""
    w = synthetic.make_window(140, 12, 1.0, seed=66, with_imu=False, frame_dt=0.05)
    _keep(w, np.asarray(w.obs_cam) == 0)
    w.pose_fixed = np.asarray(w.pose_fixed).copy()
    w.pose_fixed[:134] = 1
    if len(w.sb_fixed):
        w.sb_fixed = np.asarray(w.sb_fixed).copy()
        w.sb_fixed[:] = 1      # (no IMU terms in this window: nothing constrains a free speed / bias block)
    assert np.bincount(np.asarray(w.obs_lm)).max() > 128
    _check(oracle, w, n=4)


def test_piece_path_and_staged_kernel_agree(oracle):
    w = synthetic.config_A(seed=67)
    res = {}
    for name, r0 in (("piece", 0), ("staged", 8)):
        o = default_options()
        o.reserved0 = r0
        b = solver.WindowBatch([w], options=o)
        res[name] = (b.optimize(10)[0], b.get_state())
        b.close()
    a, s = res["piece"], res["staged"]
    assert abs(a[0]["final_cost"] - s[0]["final_cost"]) <= 1e-11 * s[0]["final_cost"]
    assert (a[0]["iterations"], a[0]["successful_steps"]) == (s[0]["iterations"], s[0]["successful_steps"])
    assert np.abs(a[1][0] - s[1][0]).max() < 1e-9 and np.abs(a[1][2] - s[1][2]).max() < 1e-8


def test_more_poses_than_the_kernel_stages_in_lds(oracle):
    """70 keyframes, two cameras (72 pose blocks > LIN2_POSES = 64): phase B and the pair lanes read the poses from global memory
    instead of the LDS copy; 140 observations = 70 pieces per landmark keep the batch on the piece path"""
    w = synthetic.make_window(70, 10, 1.0, seed=68, with_imu=False, frame_dt=0.05)
    w.pose_fixed = np.asarray(w.pose_fixed).copy()
    w.pose_fixed[:62] = 1
    w.sb_fixed = np.asarray(w.sb_fixed).copy()
    w.sb_fixed[:] = 1
    assert w.pose.shape[0] > 64 and np.bincount(np.asarray(w.obs_lm)).max() == 140
    _check(oracle, w, n=4)
