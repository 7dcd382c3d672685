"""Host side of okvis_ba_patch (include/okvis_amd_ba.h, "incremental structure updates"; reference semantics: Map::addParameterBlock /
addResidualBlock / removeResidualBlock / removeParameterBlock, okvis_ceres/src/Map.cpp:292-565): the window container okvis_ba_store_*
against windows built directly from the same data.  No GPU."""
import numpy as np
import pytest

from okvis_amd import solver, synthetic
from okvis_amd.window import PATCH_MARG_PRIOR, Patch
from tests.patch_helpers import Carving, make_marg, patch_between, sliding_pair, windows_differ


def structure(w):
    """What the index build of okvis_ba_upload makes of a window (the container keeps one copy of the raw IMU samples per term, a
    caller may share them between terms: the arena size is not compared)."""
    st = solver.check_window(w)
    st.pop("arena_bytes")
    return st


def test_store_holds_the_window_it_was_given():
    for ext in ("fixed", "shared", "perframe"):
        w = synthetic.small_window(seed=3, K=5, L=40, estimate_extrinsics=ext)
        st = solver.WindowStore(w)
        assert windows_differ(st.view(), w) == []
        assert structure(st.view()) == structure(w)
        st.close()


@pytest.mark.parametrize("seed", range(6))
def test_one_frame_of_the_sliding_window(seed):
    """Oldest frame out (its observations, IMU term and priors go with it), landmarks out, single observations out, a frame with its
    IMU term / observations / landmarks in, priors moved, dense prior replaced: the container equals the window built from scratch."""
    A, B = sliding_pair(seed=10 + seed, K=4 + seed % 3, L=60 + 10 * seed)
    st = solver.WindowStore(A.window())
    p = patch_between(A, B)
    assert len(p.remove_obs) > 0 and len(p.add_obs_lm) > 0 and len(p.remove_lm) > 0 and len(p.add_imu_pose0) == 1
    assert st.patch(p) == 0
    got, want = st.view(), B.window()
    assert windows_differ(got, want) == []
    got.validate()                                            # sorted by (lm, pose, cam)
    assert structure(got) == structure(want)
    assert np.array_equal(got.imu_sb_ref_valid, np.zeros(got.n_imu, np.uint8))
    st.close()


@pytest.mark.parametrize("seed", range(4))
def test_new_landmarks_inserted_in_id_order(seed):
    """add_lm_before: the new landmarks take their places among the ones that stay (a flattened okvis::PointMap is ordered by id)
    instead of the end; observations and sparse values use the resulting indices"""
    A, B = sliding_pair(seed=30 + seed, K=5, L=70, n_new_lm=15)
    A.lm_ids, B.lm_ids = sorted(A.lm_ids), sorted(B.lm_ids)
    assert B.lm_ids != [l for l in B.lm_ids if l in A.lm_ids] + [l for l in B.lm_ids if l not in A.lm_ids]   # (really interleaved)
    st = solver.WindowStore(A.window())
    p = patch_between(A, B)
    assert len(p.add_lm_before) == len(p.add_lm) > 0
    want = B.window()
    p.set_lm_idx = np.array([0, want.n_lm - 1], np.int32)
    p.set_lm = want.lm[[0, want.n_lm - 1]] + 0.25
    want.lm[[0, want.n_lm - 1]] += 0.25
    assert st.patch(p) == 0
    got = st.view()
    assert windows_differ(got, want) == []
    got.validate()
    assert structure(got) == structure(want)
    # places that are not ascending, or beyond the landmarks that stay, are refused and change nothing
    for bad in (p.add_lm_before[::-1].copy(), p.add_lm_before + want.n_lm):
        q = patch_between(A, B)
        q.add_lm_before = bad
        st2 = solver.WindowStore(A.window())
        if not np.array_equal(bad, p.add_lm_before):
            assert st2.patch(q) != 0
            assert windows_differ(st2.view(), A.window()) == []
        st2.close()
    st.close()


def test_several_frames_in_a_row():
    K, L = 5, 70
    W = synthetic.make_window(K + 3, L, 0.8, seed=31, frame_dt=0.2)
    ext = [("e", 0), ("e", 1)]
    lms = list(range(L))
    cur = Carving(W, [("f", k) for k in range(K)] + ext, list(range(K)), lms[:50])
    st = solver.WindowStore(cur.window())
    pose_ids, sb_ids, lm_ids = list(cur.pose_ids), list(cur.sb_ids), list(cur.lm_ids)
    for step in range(3):
        pose_ids = [i for i in pose_ids if i != ("f", step)] + [("f", K + step)]
        sb_ids = [k for k in sb_ids if k != step] + [K + step]
        lm_ids = [l for l in lm_ids if l % 7 != step] + lms[50 + 5 * step:55 + 5 * step]
        nxt = Carving(W, pose_ids, sb_ids, lm_ids, pose_prior_on=[("f", step + 1)], sb_prior_on=[step + 1])
        assert st.patch(patch_between(cur, nxt)) == 0
        assert windows_differ(st.view(), nxt.window()) == []
        cur = nxt
    st.close()


def test_sparse_values_and_equal_keys():
    w = synthetic.small_window(seed=5, K=4, L=30)
    st = solver.WindowStore(w)
    # a second keypoint of the same image on the same landmark (legal: implementation/Estimator.hpp:52-56) goes behind the first
    i = 17
    p = Patch(add_obs_lm=[w.obs_lm[i]], add_obs_pose=[w.obs_pose[i]], add_obs_ext=[w.obs_ext[i]], add_obs_cam=[w.obs_cam[i]],
              add_obs_uv=[[1.5, 2.5]], add_obs_sqrtw=[0.5], set_lm_idx=[3, 0], set_lm=[[1, 2, 3, 1], [4, 5, 6, 1]],
              set_pose_idx=[1], set_pose=[w.pose[2]], set_sb_idx=[2], set_sb=[np.arange(9.0)])
    assert st.patch(p) == 0
    v = st.view()
    assert v.n_obs == w.n_obs + 1 and np.array_equal(v.obs_uv[i + 1], [1.5, 2.5]) and np.array_equal(v.obs_uv[i], w.obs_uv[i])
    assert np.array_equal(v.lm[3], [1, 2, 3, 1]) and np.array_equal(v.lm[0], [4, 5, 6, 1]) and np.array_equal(v.lm[1], w.lm[1])
    assert np.array_equal(v.pose[1], w.pose[2]) and np.array_equal(v.sb[2], np.arange(9.0))
    st.close()


def test_rejected_patches_leave_the_window_untouched():
    rng = np.random.default_rng(8)
    W = synthetic.make_window(5, 30, 1.0, seed=8, frame_dt=0.2)
    A = Carving(W, [("f", k) for k in range(5)] + [("e", 0), ("e", 1)], list(range(5)), list(range(30)),
                marg=make_marg(W, [("p", ("f", 0)), ("s", 0)], rng))
    w = A.window()
    st = solver.WindowStore(w)
    bad = [
        Patch(remove_obs=[5, 5]), Patch(remove_obs=[9, 3]), Patch(remove_lm=[30]), Patch(remove_pose=[-1]), Patch(remove_sb=[7]),
        Patch(remove_imu=[4]),
        Patch(remove_pose=[0]),                                   # the dense prior would lose a block
        Patch(remove_sb=[0]),
        Patch(add_obs_lm=[30], add_obs_pose=[0], add_obs_ext=[5], add_obs_cam=[0], add_obs_uv=[[0, 0]], add_obs_sqrtw=[1]),
        Patch(add_obs_lm=[0], add_obs_pose=[7], add_obs_ext=[5], add_obs_cam=[0], add_obs_uv=[[0, 0]], add_obs_sqrtw=[1]),
        Patch(add_obs_lm=[0], add_obs_pose=[0], add_obs_ext=[5], add_obs_cam=[2], add_obs_uv=[[0, 0]], add_obs_sqrtw=[1]),
        Patch(set_lm_idx=[30], set_lm=[[0, 0, 0, 1]]),
        Patch(add_imu_pose0=[0], add_imu_sb0=[0], add_imu_pose1=[1], add_imu_sb1=[9], add_imu_t0=[0], add_imu_t1=[1],
              add_imu_s_begin=[0], add_imu_s_count=[2], add_imu_s_t=[0, 1], add_imu_s_gyr=np.zeros((2, 3)), add_imu_s_acc=np.zeros((2, 3))),
        Patch(add_imu_pose0=[0], add_imu_sb0=[0], add_imu_pose1=[1], add_imu_sb1=[1], add_imu_t0=[0], add_imu_t1=[1],
              add_imu_s_begin=[1], add_imu_s_count=[2], add_imu_s_t=[0, 1], add_imu_s_gyr=np.zeros((2, 3)), add_imu_s_acc=np.zeros((2, 3))),
        Patch(replace=1, pprior_pose=[9], pprior_meas=np.zeros((1, 7)), pprior_sqrtinfo=np.zeros((1, 36))),
    ]
    for p in bad:
        assert st.patch(p) == -1
        assert windows_differ(st.view(), w) == []
    # the same removal is fine once the prior is replaced (here: dropped) in the same patch
    assert st.patch(Patch(remove_pose=[0], remove_sb=[0], replace=PATCH_MARG_PRIOR)) == 0
    v = st.view()
    assert v.n_pose == w.n_pose - 1 and v.n_sb == 4 and v.n_imu == w.n_imu - 1 and np.asarray(v.marg_e0).size == 0
    assert len(v.pprior_pose) == 0 and len(v.sbprior_sb) == 0          # W's priors sat on frame 0
    assert v.n_obs == int((np.asarray(w.obs_pose) != 0).sum()) and v.obs_pose.min() == 0 and v.obs_ext.min() == 4   # renumbered
    st.close()


def test_c_abi_rejects_null_arguments():
    import ctypes as C
    from okvis_amd import _lib
    L = _lib.lib()
    assert L.okvis_ba_store_create(None, None) == -1
    assert L.okvis_ba_store_patch(None, None) == -1 and L.okvis_ba_store_view(None, None) == -1
    L.okvis_ba_store_destroy(None)
    assert L.okvis_ba_set_patchable(None, 1) == -1 and L.okvis_ba_patch_window(None, 0, None) == -1
    assert L.okvis_ba_patched_view(None, 0, None) == -1


@pytest.mark.parametrize("seed", range(10))
def test_random_edits(seed):
    """Random subsets of frames / speed-bias blocks / landmarks / observations leave, random ones arrive (blocks in the middle of the
    window, not only the oldest): the container equals the window built from scratch, and stays a valid upload."""
    rng = np.random.default_rng(1000 + seed)
    K, L = int(rng.integers(5, 9)), int(rng.integers(30, 90))
    W = synthetic.make_window(K, L, float(rng.uniform(0.5, 1.0)), seed=2000 + seed, frame_dt=0.2)
    ext = [("e", 0), ("e", 1)]
    frames = list(range(K))
    a_frames = sorted(rng.choice(frames, int(rng.integers(3, K)), replace=False).tolist())
    gone = [k for k in a_frames if rng.uniform() < 0.3]
    arrive = [k for k in frames if k not in a_frames and rng.uniform() < 0.7]
    b_frames = [k for k in a_frames if k not in gone] + arrive
    if not b_frames:
        b_frames = [a_frames[0]]
    lms = list(range(L))
    a_lm = sorted(rng.choice(lms, int(rng.integers(L // 3, L)), replace=False).tolist())
    b_lm = [l for l in a_lm if rng.uniform() > 0.25] + [l for l in lms if l not in a_lm and rng.uniform() < 0.5]
    A = Carving(W, [("f", k) for k in a_frames] + ext, a_frames, a_lm)
    cand = [(int(W.obs_lm[i]), int(W.obs_pose[i]), int(W.obs_cam[i])) for i in range(W.n_obs)
            if W.obs_lm[i] in a_lm and W.obs_lm[i] in b_lm and W.obs_pose[i] in a_frames and W.obs_pose[i] in b_frames]
    drop = [cand[i] for i in rng.choice(len(cand), min(len(cand), int(rng.integers(0, 12))), replace=False)] if cand else []
    # speed/bias blocks may leave without their frame (Estimator::applyMarginalizationStrategy drops the speed/bias of old keyframes)
    b_sb = [k for k in b_frames if k in arrive or rng.uniform() > 0.2]
    kept_f = [k for k in a_frames if k in b_frames]
    B = Carving(W, [("f", k) for k in kept_f] + ext + [("f", k) for k in arrive],
                [k for k in a_frames if k in b_sb] + [k for k in arrive if k in b_sb], b_lm, drop_obs=drop,
                pose_prior_on=[("f", b_frames[0])], sb_prior_on=b_sb[:1] if b_sb else [],
                marg=make_marg(W, [("p", ("f", b_frames[0]))] + ([("s", b_sb[0])] if b_sb else []), rng))
    st = solver.WindowStore(A.window())
    assert st.patch(patch_between(A, B)) == 0
    got, want = st.view(), B.window()
    assert windows_differ(got, want) == []
    assert structure(got) == structure(want)
    st.close()


def test_per_frame_extrinsics_leave_with_their_relative_pose_terms():
    """A window with one extrinsics block per frame and camera, chained by RelativePoseError terms (TestEstimator.cpp case c = 3):
    removing frame 0 and its two extrinsics blocks takes its observations, its IMU term, the priors on those blocks and the two
    relative-pose terms that touch them along; everything else is renumbered in place."""
    K, NC = 5, 2
    w = synthetic.small_window(seed=21, K=K, L=40, estimate_extrinsics="perframe")
    gone = [0, K + 0, K + 1]                                   # T_WS of frame 0, T_SC0 and T_SC1 of frame 0
    st = solver.WindowStore(w)
    assert st.patch(Patch(remove_pose=gone, remove_sb=[0])) == 0
    v = st.view()
    new = -np.ones(w.n_pose, np.int64)
    new[[i for i in range(w.n_pose) if i not in gone]] = np.arange(w.n_pose - 3)
    keep_o = ~np.isin(w.obs_pose, gone) & ~np.isin(w.obs_ext, gone)
    assert not keep_o.all() and v.n_obs == int(keep_o.sum())
    assert np.array_equal(v.obs_pose, new[w.obs_pose[keep_o]]) and np.array_equal(v.obs_ext, new[w.obs_ext[keep_o]])
    assert np.array_equal(v.obs_uv, w.obs_uv[keep_o]) and np.array_equal(v.obs_lm, w.obs_lm[keep_o])
    keep_r = ~np.isin(w.rel_pose0, gone) & ~np.isin(w.rel_pose1, gone)
    assert int((~keep_r).sum()) == NC
    assert np.array_equal(v.rel_pose0, new[w.rel_pose0[keep_r]]) and np.array_equal(v.rel_pose1, new[w.rel_pose1[keep_r]])
    assert np.array_equal(v.rel_sqrtinfo, w.rel_sqrtinfo[keep_r])
    keep_p = ~np.isin(w.pprior_pose, gone)
    assert np.array_equal(v.pprior_pose, new[w.pprior_pose[keep_p]]) and len(v.sbprior_sb) == 0
    assert v.n_imu == w.n_imu - 1 and np.array_equal(v.imu_pose0, new[w.imu_pose0[1:]]) and np.array_equal(v.imu_sb0, w.imu_sb0[1:] - 1)
    assert np.array_equal(v.pose, np.delete(w.pose, gone, axis=0)) and np.array_equal(v.sb, w.sb[1:])
    assert np.array_equal(v.lm, w.lm)                          # landmarks stay, observed or not
    v.validate()
    assert solver.check_window(v)["D"] == solver.check_window(w)["D"] - 6 * 3 - 9
    st.close()


def test_store_rejects_out_of_range_indices():
    """okvis_ba_store_create checks every index array (the container is usable without a device: okvis_ba_upload's validation
    never sees these windows, and apply() indexes its remap tables with the values)."""
    import copy
    from okvis_amd._lib import BackendError
    base = synthetic.small_window(seed=5, K=4, L=30)

    def broken(field, idx, value):
        w = copy.deepcopy(base)
        a = np.array(getattr(w, field)).copy()
        a[idx] = value
        setattr(w, field, a)
        return w

    cases = [("obs_lm", 0, base.n_lm), ("obs_lm", 3, -1), ("obs_pose", 1, base.n_pose), ("obs_ext", 2, -2), ("obs_cam", 0, len(base.cam_model)),
             ("imu_pose0", 0, base.n_pose), ("imu_sb1", 0, base.n_sb), ("imu_s_count", 0, 2 ** 31 - 1), ("imu_s_begin", 0, -1)]
    if len(base.pprior_pose):
        cases.append(("pprior_pose", 0, base.n_pose))
    if len(base.sbprior_sb):
        cases.append(("sbprior_sb", 0, -1))
    for field, idx, value in cases:
        with pytest.raises(BackendError):
            solver.WindowStore(broken(field, idx, value))
    solver.WindowStore(base).close()   # the untouched window is still accepted


def test_preintegration_records_stay_with_their_terms():
    """okvis_ba_window::imu_cache (flag 2): the container keeps a term's preintegration record and its reference bias through edits —
    the term that leaves takes its record with it, the one that arrives has none — and refuses flags it cannot honour."""
    A, B = sliding_pair(seed=77, K=5, L=60)
    w = A.window()
    n = w.n_imu
    rng = np.random.default_rng(5)
    w.imu_sb_ref = rng.standard_normal((n, 9))
    w.imu_cache = rng.standard_normal((n, 290))
    w.imu_sb_ref_valid = np.array([2, 1, 0, 2][:n] + [2] * max(0, n - 4), np.uint8)
    st = solver.WindowStore(w)
    v = st.view()
    assert np.array_equal(v.imu_sb_ref_valid, w.imu_sb_ref_valid)
    for f in range(n):
        if w.imu_sb_ref_valid[f]:
            assert np.array_equal(v.imu_sb_ref[f], w.imu_sb_ref[f])
        if w.imu_sb_ref_valid[f] == 2:
            assert np.array_equal(v.imu_cache[f], w.imu_cache[f])
    assert st.patch(patch_between(A, B)) == 0
    got = st.view()
    kept = [f for f in range(n) if imu_term_key(w, f) in {imu_term_key(got, g) for g in range(got.n_imu)}]
    assert 0 < len(kept) < n and got.n_imu == len(kept) + 1
    for g in range(got.n_imu):
        src = [f for f in kept if imu_term_key(w, f) == imu_term_key(got, g)]
        if not src:
            assert got.imu_sb_ref_valid[g] == 0                       # the new term: its first evaluation integrates
            continue
        f = src[0]
        assert got.imu_sb_ref_valid[g] == w.imu_sb_ref_valid[f]
        if w.imu_sb_ref_valid[f]:
            assert np.array_equal(got.imu_sb_ref[g], w.imu_sb_ref[f])
        if w.imu_sb_ref_valid[f] == 2:
            assert np.array_equal(got.imu_cache[g], w.imu_cache[f])
    st.close()
    bad = A.window()
    bad.imu_sb_ref = np.zeros((n, 9)); bad.imu_sb_ref_valid = np.full(n, 2, np.uint8)    # flag 2 without the records
    with pytest.raises(Exception):
        solver.WindowStore(bad)
    bad.imu_cache = np.zeros((n, 290)); bad.imu_sb_ref_valid = np.full(n, 3, np.uint8)   # no such flag
    with pytest.raises(Exception):
        solver.WindowStore(bad)


def imu_term_key(w, f):
    return (int(w.imu_t0[f]), int(w.imu_t1[f]))
