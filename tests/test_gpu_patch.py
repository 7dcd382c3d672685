"""okvis_ba_patch_window on the GPU: a patched solver and a solver that uploads the same edits from scratch give bit-identical
results, the kept blocks carry the device's values over, and the patched window agrees with the oracle."""
import numpy as np
import pytest

from okvis_amd import solver
from okvis_amd.window import Patch, default_options
from tests.patch_helpers import patch_between, sliding_pair, windows_differ

pytestmark = pytest.mark.gpu


def _state(b):
    return [np.asarray(a).copy() for a in b.get_state(0)]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_patched_solver_equals_fresh_upload_bitwise(oracle, seed):
    A, B = sliding_pair(seed=40 + seed, K=5 + seed, L=90)
    pa = solver.WindowBatch([A.window()], options=default_options(), patchable=True)
    s0 = pa.optimize(4)[0]
    pose_a, sb_a, lm_a = _state(pa)
    p = patch_between(A, B)
    pa.patch(0, p)
    v = pa.patched_view(0)                        # the container: B's structure with the values the device held
    want = B.window()
    assert windows_differ(v, want) == ["pose", "sb", "lm"]          # structure of B ...
    keep_pose = [n for n, i in enumerate(A.pose_ids) if i in B.pose_ids]
    keep_sb = [n for n, k in enumerate(A.sb_ids) if k in B.sb_ids]
    keep_lm = [n for n, l in enumerate(A.lm_ids) if l in B.lm_ids]
    assert np.array_equal(v.pose[:len(keep_pose)], pose_a[keep_pose])      # ... optimised values of what stayed ...
    assert np.array_equal(v.sb[:len(keep_sb)], sb_a[keep_sb]) and np.array_equal(v.lm[:len(keep_lm)], lm_a[keep_lm])
    assert np.array_equal(v.pose[len(keep_pose):], want.pose[len(keep_pose):])     # ... the given values of what arrived
    assert np.array_equal(v.imu_sb_ref_valid, [2] * (v.n_imu - 1) + [0])           # kept IMU terms keep their preintegration (flag 2)
    fresh = solver.WindowBatch([v], options=default_options())
    s1, s2 = pa.optimize(5)[0], fresh.optimize(5)[0]
    assert s1 == s2 and s1["iterations"] > 0 and s1["initial_cost"] != s0["final_cost"]
    for a, b in zip(_state(pa), _state(fresh)):
        assert np.array_equal(a, b)
    qa, qb = pa.array("LM_QUALITY"), fresh.array("LM_QUALITY")
    assert np.array_equal(qa, qb)
    so = oracle.OracleWindow(v).optimize(5)
    assert so["iterations"] == s1["iterations"] and abs(so["final_cost"] - s1["final_cost"]) <= 1e-9 * so["final_cost"]
    pa.close(); fresh.close()


def test_new_landmarks_in_place_equal_a_fresh_upload_bitwise():
    """add_lm_before (what okvis_amd::Estimator sends: new landmarks take their places by id among the ones that stay): the
    landmarks that stay keep the device's values at their new indices; the patched solver and a fresh upload of its container
    iterate bit for bit alike"""
    A, B = sliding_pair(seed=70, K=6, L=90, n_new_lm=14)
    A.lm_ids, B.lm_ids = sorted(A.lm_ids), sorted(B.lm_ids)
    pa = solver.WindowBatch([A.window()], options=default_options(), patchable=True)
    pa.optimize(4)
    lm_a = _state(pa)[2]
    p = patch_between(A, B)
    assert len(p.add_lm_before) == len(p.add_lm) > 0 and p.add_lm_before[0] < len(B.lm_ids) - len(p.add_lm)
    pa.patch(0, p)
    v = pa.patched_view(0)
    want = B.window()
    assert windows_differ(v, want) == ["pose", "sb", "lm"]
    ia = {l: n for n, l in enumerate(A.lm_ids)}
    for n, l in enumerate(B.lm_ids):
        assert np.array_equal(v.lm[n], lm_a[ia[l]] if l in ia else want.lm[n])
    fresh = solver.WindowBatch([v], options=default_options())
    assert pa.optimize(5)[0] == fresh.optimize(5)[0]
    for a, b in zip(_state(pa), _state(fresh)):
        assert np.array_equal(a, b)
    pa.close(); fresh.close()


def test_patch_in_a_batch_and_state_errors():
    A, B = sliding_pair(seed=50, K=5, L=60)
    A2, _ = sliding_pair(seed=51, K=5, L=60)
    plain = solver.WindowBatch([A.window()], options=default_options())
    pc, keep = patch_between(A, B).as_c()
    import ctypes as C
    assert plain._L.okvis_ba_patch_window(plain._h, 0, C.byref(pc)) == -2         # not patchable
    plain.close()
    b = solver.WindowBatch([A.window(), A2.window()], options=default_options(), patchable=True)
    b.optimize(3)
    st_other = [np.asarray(a).copy() for a in b.get_state(1)]
    assert b._L.okvis_ba_patch_window(b._h, 2, C.byref(pc)) == -1
    bad, kb = Patch(remove_lm=[10 ** 6]).as_c()
    assert b._L.okvis_ba_patch_window(b._h, 0, C.byref(bad)) == -1
    b.patch(0, patch_between(A, B))
    for a, c in zip(b.get_state(1), st_other):     # the other window of the batch keeps its optimised state through the rebuild
        assert np.array_equal(np.asarray(a), c)
    v0, v1 = b.patched_view(0), b.patched_view(1)
    fresh = solver.WindowBatch([v0, v1], options=default_options())
    assert b.optimize(4) == fresh.optimize(4)
    b.close(); fresh.close()


def test_patch_beyond_a_structure_limit_leaves_the_solver_as_it_was():
    """The edit itself is legal, its result is not (one landmark with more observations than okvis_ba_limits::max_obs_per_lm): the
    status of the upload check comes back and the solver still holds the old window with its optimised values."""
    import ctypes as C
    A, _ = sliding_pair(seed=60, K=5, L=60)
    w = A.window()
    b = solver.WindowBatch([w], options=default_options(), patchable=True)
    b.optimize(3)
    before = [np.asarray(a).copy() for a in b.get_state(0)]
    n = solver.limits()["max_obs_per_lm"] + 1
    too_many = Patch(add_obs_lm=np.zeros(n, np.int32), add_obs_pose=np.zeros(n, np.int32), add_obs_ext=np.full(n, 5, np.int32),
                     add_obs_cam=np.zeros(n, np.int32), add_obs_uv=np.zeros((n, 2)), add_obs_sqrtw=np.ones(n))
    pc, keep = too_many.as_c()
    assert b._L.okvis_ba_patch_window(b._h, 0, C.byref(pc)) == -3
    assert windows_differ(b.patched_view(0), w) == ["pose", "sb", "lm"]       # structure untouched, values = the device's
    for a, c in zip(b.get_state(0), before):
        assert np.array_equal(np.asarray(a), c)
    s1 = b.optimize(2)[0]
    assert s1["iterations"] >= 1 and np.isfinite(s1["final_cost"])
    b.close()
