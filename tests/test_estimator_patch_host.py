"""CPU tests of the window edits of okvis_amd::Estimator (csrc/host/estimator.cpp: patchWindow).  Between two optimize() calls
the estimator does not re-derive its window: the edits of addStates / addObservation / removeObservation /
applyMarginalizationStrategy / the setters become ONE okvis_ba_patch of the window the solver holds (the reference edits its
ceres::Problem in place the same way, okvis_ceres/src/Map.cpp:292-565).  A book-keeping-only estimator (device -1: no solver, no
numbers) applies these patches to the host container okvis_ba_patch_window uses (ba_store.hpp) and compares the result with a
freshly flattened window after every hand-over; optimize() raises when they differ."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import replay_scenario as RS  # noqa: E402
from okvis_amd import estimator as E  # noqa: E402
from okvis_amd import recording  # noqa: E402


class _Dry(E.Estimator):
    def __init__(self):
        super().__init__(-1)
        self.patched = []

    def optimize(self, *a, **k):
        s = super().optimize(*a, **k)          # (raises if the patched container differs from a fresh flatten)
        self.patched.append(self.lastOptimizeWasPatch())
        return s


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("asl_patch"))
    return d, recording.write_synthetic_recording(d, duration_s=6.0, n_points=500, seed=11)


def test_replay_windows_are_patched_and_equal_fresh_ones(folder):
    d, _ = folder
    made = []

    def make():
        made.append(_Dry())
        return made[-1]

    tr = RS.replay(RS.read(d), make, E.Frame, max_frames=45)
    assert len(tr) == 45
    flags = made[0].patched
    assert flags[0] is False and all(flags[1:]), flags          # one upload, then edits only
    assert max(r["n_landmarks"] for r in tr) > 150 and tr[-1]["n_frames"] <= 9
    assert sum(len(r["removed"]) for r in tr) > 50              # frames and landmarks did leave through the patches


def test_no_patch_switch_uploads_every_time(folder):
    d, _ = folder
    made = []

    def make():
        e = _Dry()
        e.setUsePatch(False)
        made.append(e)
        return e

    RS.replay(RS.read(d), make, E.Frame, max_frames=8)
    assert not any(made[0].patched)


def _frame_loop(rng, est, rec, n_frames, per_frame_hook):
    """a reduced replay loop with a hook between the observations of a frame and its optimize()"""
    lm_row = {int(i): k for k, i in enumerate(rec["lm_i"][:, 0])}
    frame_id_at = {int(t): int(i) for t, i, _ in rec["frames"]}
    gone, added, keep, obs_at, last_t = set(), set(), [], 0, 0
    observed = {}
    for k in range(n_frames):
        t_ns, fid, kf = (int(x) for x in rec["frames"][k])
        f = E.Frame(fid, t_ns, rec["T_SC"], rec["intr"], list(rec["model"]))
        keep.append(f)
        t0, t1 = (last_t if k else t_ns) - 20_000_000, t_ns + 20_000_000
        lo, hi = np.searchsorted(rec["imu_t"], t0, "left"), np.searchsorted(rec["imu_t"], t1, "right")
        assert est.addStates(f, rec["imu_t"][lo:hi], rec["imu_ga"][lo:hi, :3], rec["imu_ga"][lo:hi, 3:], bool(kf))
        last_t = t_ns
        in_window = {est.frameIdByAge(a) for a in range(est.numFrames())}
        while obs_at < len(rec["obs_i"]) and rec["obs_i"][obs_at, 0] <= t_ns:
            ot, cam, lid = (int(x) for x in rec["obs_i"][obs_at])
            u, v, size = rec["obs_f"][obs_at]
            obs_at += 1
            if ot != t_ns or lid in gone:
                continue
            if lid not in added:
                if frame_id_at[int(rec["lm_i"][lm_row[lid], 1])] not in in_window:
                    continue
                est.addLandmark(lid, rec["lm_hp"][lm_row[lid]])
                added.add(lid)
            kp = f.add_keypoint(cam, float(u), float(v), float(size))
            if est.addObservation(lid, fid, cam, kp) != 0:
                observed.setdefault(lid, []).append((fid, cam, kp))
        per_frame_hook(k, fid, observed, in_window)
        est.optimize(3, 1, False)
        removed = []
        est.applyMarginalizationStrategy(4, 3, removed)
        gone.update(removed)
        added.difference_update(removed)
        for lid in removed:
            observed.pop(lid, None)
        in_window = {est.frameIdByAge(a) for a in range(est.numFrames())}
        for lid in list(observed):
            observed[lid] = [o for o in observed[lid] if o[0] in in_window]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_edits_between_frames(folder, seed):
    """observations removed at random (some landmarks lose all of them and come back later), values set by the caller,
    a marginalisation that fails and is rolled back: every window the patches produce equals a freshly flattened one"""
    d, _ = folder
    rec = RS.read(d)
    rng = np.random.default_rng(seed)
    est = _Dry()
    for _ in range(len(rec["T_SC"])):
        est.addCamera(0, 0, 0, 0)
    est.addImu(rec["imu_params"])
    emptied = []

    def hook(k, fid, observed, in_window):
        lids = [l for l in observed if observed[l]]
        for lid in rng.choice(lids, size=min(12, len(lids)), replace=False) if lids else []:
            lst = observed[int(lid)]
            j = int(rng.integers(len(lst)))
            assert est.removeObservation(int(lid), *lst[j])
            lst.pop(j)
        if k % 3 == 1 and lids:                       # one landmark loses every observation (stays in the map, leaves the window)
            lid = int(rng.choice(lids))
            for o in observed[lid]:
                assert est.removeObservation(lid, *o)
            emptied.append((lid, list(observed[lid])))
            observed[lid] = []
        if k % 3 == 0 and emptied:                    # ... and is observed again later (from frames still in the window)
            lid, obs = emptied.pop(0)
            for o in obs:
                if o[0] in in_window and lid in observed:
                    if est.addObservation(lid, *o) != 0:
                        observed[lid].append(o)
        for lid in rng.choice(lids, size=min(5, len(lids)), replace=False) if lids else []:
            p, _, _ = est.getLandmark(int(lid))
            est.setLandmark(int(lid), p + np.r_[rng.normal(0, 1e-3, 3), 0])
        T = est.get_T_WS(fid)
        T[:3] += rng.normal(0, 1e-3, 3)
        est.set_T_WS(fid, T)
        sb = est.getSpeedAndBias(fid)
        est.setSpeedAndBias(fid, sb + rng.normal(0, 1e-3, 9))
        if k in (9, 14):
            est.debugFailNextMarginalization()

    failed = []
    orig = est.applyMarginalizationStrategy

    def marg(nk, ni, removed):
        try:
            return orig(nk, ni, removed)
        except E.EstimatorError:
            failed.append(1)                           # rolled back: the next call does the work
            return orig(nk, ni, removed)

    est.applyMarginalizationStrategy = marg
    _frame_loop(rng, est, rec, 30, hook)
    assert len(failed) == 2
    assert est.patched[0] is False and all(est.patched[1:]), est.patched   # (a roll-back takes its entries of the edit logs back)
    assert est.debugCheckWindow() != ""          # (the last marginalisation's edits are still to be handed over)
    est.optimize(1, 1, False)
    assert est.debugCheckWindow() == ""
    est.close()


def test_per_frame_extrinsics_and_their_terms(folder):
    """cameras with relative extrinsics noise get a block per frame and a relative-pose term between consecutive ones
    (Estimator.cpp:191-218, 310-336): blocks, priors and terms travel through the patch"""
    d, _ = folder
    rec = RS.read(d)
    est = _Dry()
    for _ in range(len(rec["T_SC"])):
        est.addCamera(0.01, 0.01, 1e-3, 1e-3)
    est.addImu(rec["imu_params"])
    _frame_loop(np.random.default_rng(0), est, rec, 20, lambda *a: None)
    assert est.patched[0] is False and all(est.patched[1:])
    est.close()


def test_observation_table_stays_bounded_under_a_long_lived_observation():
    """A camera standing still: the old keyframes (and their observations) stay while the observations of every passing frame
    come and go.  The handle-indexed table drops dead slots only at its front; it must not grow with the number of observations
    ever made (~72 bytes per dead slot, 300 KB/s at 10 Hz stereo)."""
    from okvis_amd import synthetic
    from okvis_amd.window import DIST_EQUIDISTANT, ImuParams
    prm = ImuParams(sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, g=9.81, g_max=1000.0, a_max=1000.0)
    est = _Dry()
    est.addCamera(0, 0, 0, 0)
    est.addImu(E.imu_param_vector(prm))
    T_SC = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    intr = np.stack([synthetic.TEST_INTR_EQUI])
    t = (np.arange(120) * 10_000_000).astype(np.int64) + 1_000_000_000
    gyr = np.zeros((120, 3)); acc = np.tile([0, 0, 9.81], (120, 1))
    frames = []
    for k in range(2):
        f = E.Frame(1_000_000 + k, 1_000_000_000 + k * 500_000_000, T_SC, intr, [DIST_EQUIDISTANT])
        for i in range(8):
            f.add_keypoint(0, 100.0 + 10 * i, 120.0, 8.0)
        frames.append(f)
        lo, hi = (0, 4) if k == 0 else (0, 56)
        assert est.addStates(f, t[lo:hi], gyr[lo:hi], acc[lo:hi], True), est.last_error()
    assert est.addLandmark(7_000_001, [3.0, 0.0, 0.0, 1.0]) and est.addLandmark(7_000_002, [3.0, 0.5, 0.0, 1.0])
    first = est.addObservation(7_000_001, frames[0].id, 0, 0)      # stays for good
    assert first
    api = est._api
    api.okvis_est_debug_obs_slots.restype = __import__("ctypes").c_longlong
    api.okvis_est_debug_obs_slots.argtypes = [__import__("ctypes").c_void_p]
    n = 40_000
    for i in range(n):
        h = est.addObservation(7_000_002, frames[1].id, 0, 1 + i % 7)
        assert h > first
        assert est.removeObservation(7_000_002, frames[1].id, 0, 1 + i % 7)
    slots = api.okvis_est_debug_obs_slots(est._h)
    assert slots < 5000, slots                                    # not ~n
    # the old observation is still there and can be removed; a new one still lands
    h = est.addObservation(7_000_002, frames[1].id, 0, 3)
    assert h and est.removeObservation(7_000_001, frames[0].id, 0, 0) and est.removeObservation(7_000_002, frames[1].id, 0, 3)
    assert api.okvis_est_debug_obs_slots(est._h) < 5000
