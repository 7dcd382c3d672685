"""Helpers of the okvis_ba_patch tests: carve sub-windows out of one big synthetic window by IDENTITY (frame number,
extrinsics camera, landmark number) and derive the patch that turns one carving into another.  The expected result of a patch is
built here from the big window directly, never by re-implementing the patch: the C++ container (okvis_amd/csrc/ba_store.hpp) and
this file only share the documented semantics (include/okvis_amd_ba.h, "incremental structure updates")."""
import numpy as np

from okvis_amd import synthetic
from okvis_amd.window import PATCH_MARG_PRIOR, PATCH_POSE_PRIORS, PATCH_SB_PRIORS, Patch, Window


class Carving:
    """pose_ids: list of ('f', k) / ('e', c); sb_ids: list of frame numbers; lm_ids: list of landmark numbers of W;
    drop_obs: set of (lm, frame, cam) identities left out (removeObservation of the frontend)."""

    def __init__(self, W, pose_ids, sb_ids, lm_ids, drop_obs=(), pose_prior_on=None, sb_prior_on=None, marg=None):
        self.W, self.pose_ids, self.sb_ids, self.lm_ids = W, list(pose_ids), list(sb_ids), list(lm_ids)
        self.drop_obs = set(drop_obs)
        self.pose_prior_on = pose_prior_on    # list of pose identities that carry an absolute prior (None: W's own, if included)
        self.sb_prior_on = sb_prior_on
        self.marg = marg                      # None or dict(ids=[('p', pose_id) | ('s', frame)], J, e0, lin)
        self.imu_first = []                   # IMU terms of W that come first, in this order (the ones an edited window already had)

    def _pose_block(self, ident):
        K = self.W.meta["K"]
        return ident[1] if ident[0] == "f" else K + ident[1]

    def window(self) -> Window:
        W = self.W
        K = W.meta["K"]
        pmap = {self._pose_block(i): n for n, i in enumerate(self.pose_ids)}      # block of W -> index here
        smap = {k: n for n, k in enumerate(self.sb_ids)}
        lmap = {l: n for n, l in enumerate(self.lm_ids)}
        pb = [self._pose_block(i) for i in self.pose_ids]
        keep = [i for i in range(W.n_obs)
                if W.obs_lm[i] in lmap and W.obs_pose[i] in pmap and W.obs_ext[i] in pmap
                and (int(W.obs_lm[i]), int(W.obs_pose[i]), int(W.obs_cam[i])) not in self.drop_obs]
        keep = np.array(keep, np.int64)
        w = Window(
            pose=W.pose[pb], pose_fixed=W.pose_fixed[pb], sb=W.sb[self.sb_ids], sb_fixed=W.sb_fixed[self.sb_ids], lm=W.lm[self.lm_ids],
            cam_intr=W.cam_intr, cam_model=W.cam_model,
            obs_lm=np.array([lmap[int(l)] for l in W.obs_lm[keep]], np.int32), obs_pose=np.array([pmap[int(p)] for p in W.obs_pose[keep]], np.int32),
            obs_ext=np.array([pmap[int(p)] for p in W.obs_ext[keep]], np.int32), obs_cam=W.obs_cam[keep].astype(np.int32),
            obs_uv=W.obs_uv[keep].reshape(-1, 2), obs_sqrtw=W.obs_sqrtw[keep], cauchy_b=W.cauchy_b, imu_params=W.imu_params)
        w.sort_observations()
        # IMU terms of W whose four blocks are here, in W's order
        fs = [f for f in range(W.n_imu) if W.imu_pose0[f] in pmap and W.imu_pose1[f] in pmap and W.imu_sb0[f] in smap and W.imu_sb1[f] in smap]
        fs = [f for f in self.imu_first if f in fs] + [f for f in fs if f not in self.imu_first]
        self.imu_terms = fs
        w.imu_pose0 = np.array([pmap[int(W.imu_pose0[f])] for f in fs], np.int32)
        w.imu_pose1 = np.array([pmap[int(W.imu_pose1[f])] for f in fs], np.int32)
        w.imu_sb0 = np.array([smap[int(W.imu_sb0[f])] for f in fs], np.int32)
        w.imu_sb1 = np.array([smap[int(W.imu_sb1[f])] for f in fs], np.int32)
        w.imu_t0, w.imu_t1 = W.imu_t0[fs], W.imu_t1[fs]
        w.imu_s_begin, w.imu_s_count = W.imu_s_begin[fs], W.imu_s_count[fs]
        w.imu_s_t, w.imu_s_gyr, w.imu_s_acc = W.imu_s_t, W.imu_s_gyr, W.imu_s_acc
        # priors
        if self.pose_prior_on is None:
            kp = [i for i in range(len(W.pprior_pose)) if int(W.pprior_pose[i]) in pmap]
            w.pprior_pose = np.array([pmap[int(W.pprior_pose[i])] for i in kp], np.int32)
            w.pprior_meas, w.pprior_sqrtinfo = W.pprior_meas[kp].reshape(-1, 7), W.pprior_sqrtinfo[kp].reshape(-1, 36)
        else:
            w.pprior_pose, w.pprior_meas, w.pprior_sqrtinfo = self.pose_prior_arrays(pmap)
        if self.sb_prior_on is None:
            ks = [i for i in range(len(W.sbprior_sb)) if int(W.sbprior_sb[i]) in smap]
            w.sbprior_sb = np.array([smap[int(W.sbprior_sb[i])] for i in ks], np.int32)
            w.sbprior_meas, w.sbprior_sqrtinfo = W.sbprior_meas[ks].reshape(-1, 9), W.sbprior_sqrtinfo[ks].reshape(-1, 81)
        else:
            w.sbprior_sb, w.sbprior_meas, w.sbprior_sqrtinfo = self.sb_prior_arrays(smap)
        if self.marg is not None:
            bt, bi, bo = self.marg_blocks(pmap, smap)
            w.marg_block_type, w.marg_block_idx, w.marg_block_off = bt, bi, bo
            w.marg_J, w.marg_e0, w.marg_lin = self.marg["J"], self.marg["e0"], self.marg["lin"]
        w.validate()
        return w

    def pose_prior_arrays(self, pmap):
        W = self.W
        si = synthetic.sqrt_information_eigen_llt(np.diag([1e4, 1e4, 1e4, 1e2, 1e2, 1e4])).reshape(-1)
        idx = np.array([pmap[self._pose_block(i)] for i in self.pose_prior_on], np.int32)
        meas = np.array([W.pose[self._pose_block(i)] for i in self.pose_prior_on]).reshape(-1, 7)
        return idx, meas, np.tile(si, (len(idx), 1)).reshape(-1, 36)

    def sb_prior_arrays(self, smap):
        W = self.W
        si = synthetic.sqrt_information_eigen_llt(np.diag([1.0] * 3 + [1e3] * 3 + [1e2] * 3)).reshape(-1)
        idx = np.array([smap[k] for k in self.sb_prior_on], np.int32)
        return idx, W.sb[self.sb_prior_on].reshape(-1, 9), np.tile(si, (len(idx), 1)).reshape(-1, 81)

    def marg_blocks(self, pmap, smap):
        bt, bi, bo, off = [], [], [], 0
        for kind, ident in self.marg["ids"]:
            if kind == "p":
                bt.append(0); bi.append(pmap[self._pose_block(ident)]); bo.append(off); off += 6
            else:
                bt.append(1); bi.append(smap[ident]); bo.append(off); off += 9
        return np.array(bt, np.int32), np.array(bi, np.int32), np.array(bo, np.int32)


def make_marg(W, ids, rng):
    """A synthetic dense prior e = e0 + J dchi over the given blocks (MarginalizationError::EvaluateWithMinimalJacobians)."""
    K = W.meta["K"]
    dims = [6 if k == "p" else 9 for k, _ in ids]
    Dm = sum(dims)
    lin = np.zeros((len(ids), 9))
    for n, (k, ident) in enumerate(ids):
        if k == "p":
            blk = ident[1] if ident[0] == "f" else K + ident[1]
            lin[n, :7] = synthetic.pose_oplus(W.pose[blk], rng.normal(0, 1e-3, 6))
        else:
            lin[n] = W.sb[ident] + rng.normal(0, 1e-3, 9)
    return dict(ids=list(ids), J=np.triu(rng.standard_normal((Dm, Dm))) + 3.0 * np.eye(Dm), e0=rng.standard_normal(Dm) * 0.05, lin=lin)


def patch_between(A: Carving, B: Carving) -> Patch:
    """The okvis_ba_patch that turns carving A into carving B.  Requires B's kept blocks in A's relative order, new ones behind;
    new landmarks may sit anywhere among the kept ones (add_lm_before then says where)."""
    W = A.W
    p = Patch()
    keep_pose = [i for i in A.pose_ids if i in B.pose_ids]
    keep_sb = [k for k in A.sb_ids if k in B.sb_ids]
    keep_lm = [l for l in A.lm_ids if l in B.lm_ids]
    new_pose = [i for i in B.pose_ids if i not in A.pose_ids]
    new_sb = [k for k in B.sb_ids if k not in A.sb_ids]
    new_lm = [l for l in B.lm_ids if l not in A.lm_ids]
    assert B.pose_ids == keep_pose + new_pose and B.sb_ids == keep_sb + new_sb
    assert [l for l in B.lm_ids if l in A.lm_ids] == keep_lm
    lm_in_place = B.lm_ids != keep_lm + new_lm
    p.remove_pose = np.array([n for n, i in enumerate(A.pose_ids) if i not in B.pose_ids], np.int32)
    p.remove_sb = np.array([n for n, k in enumerate(A.sb_ids) if k not in B.sb_ids], np.int32)
    p.remove_lm = np.array([n for n, l in enumerate(A.lm_ids) if l not in B.lm_ids], np.int32)
    wa = A.window()
    B.imu_first = list(A.imu_terms)      # appended terms go behind the ones that stay
    wb = B.window()
    K = W.meta["K"]
    # observations by identity (lm of W, pose block of W, ext block of W, cam, u, v)
    def ident(C, w):
        pb = [C._pose_block(i) for i in C.pose_ids]
        return [(C.lm_ids[w.obs_lm[i]], pb[w.obs_pose[i]], pb[w.obs_ext[i]], int(w.obs_cam[i]), float(w.obs_uv[i, 0]), float(w.obs_uv[i, 1]))
                for i in range(w.n_obs)]
    ia, ib = ident(A, wa), ident(B, wb)
    sb_ = set(ib)
    kept_blocks = {A._pose_block(i) for i in keep_pose}
    # explicit removals: only observations whose blocks all stay (the others go with their block)
    p.remove_obs = np.array([n for n, o in enumerate(ia)
                             if o not in sb_ and o[0] in keep_lm and o[1] in kept_blocks and o[2] in kept_blocks], np.int32)
    sa = set(ia)
    add = [n for n, o in enumerate(ib) if o not in sa]
    rng = np.random.default_rng(len(add))
    add = list(rng.permutation(add))                     # "any order"
    p.add_obs_lm, p.add_obs_pose, p.add_obs_ext = wb.obs_lm[add], wb.obs_pose[add], wb.obs_ext[add]
    p.add_obs_cam, p.add_obs_uv, p.add_obs_sqrtw = wb.obs_cam[add], wb.obs_uv[add].reshape(-1, 2), wb.obs_sqrtw[add]
    pbB = [B._pose_block(i) for i in B.pose_ids]
    nkp, nks, nkl = len(keep_pose), len(keep_sb), len(keep_lm)
    p.add_pose, p.add_pose_fixed = wb.pose[nkp:].reshape(-1, 7), wb.pose_fixed[nkp:]
    p.add_sb, p.add_sb_fixed = wb.sb[nks:].reshape(-1, 9), wb.sb_fixed[nks:]
    p.add_lm = wb.lm[nkl:].reshape(-1, 4)
    if lm_in_place:
        pos = [n for n, l in enumerate(B.lm_ids) if l not in A.lm_ids]
        p.add_lm = wb.lm[pos].reshape(-1, 4)
        p.add_lm_before = np.array([n - k for k, n in enumerate(pos)], np.int32)   # kept landmarks in front of each new one
    # IMU terms: B's terms that A does not have, with their own sample arrays
    new_terms = [n for n, f in enumerate(B.imu_terms) if f not in A.imu_terms]
    assert [f for f in B.imu_terms if f in A.imu_terms] + [B.imu_terms[n] for n in new_terms] == B.imu_terms
    p.add_imu_pose0, p.add_imu_sb0 = wb.imu_pose0[new_terms], wb.imu_sb0[new_terms]
    p.add_imu_pose1, p.add_imu_sb1 = wb.imu_pose1[new_terms], wb.imu_sb1[new_terms]
    p.add_imu_t0, p.add_imu_t1 = wb.imu_t0[new_terms], wb.imu_t1[new_terms]
    st, sg, sa_, beg = [], [], [], []
    for n in new_terms:
        b, c = int(wb.imu_s_begin[n]), int(wb.imu_s_count[n])
        beg.append(sum(len(x) for x in st))
        st.append(wb.imu_s_t[b:b + c]); sg.append(wb.imu_s_gyr[b:b + c]); sa_.append(wb.imu_s_acc[b:b + c])
    p.add_imu_s_begin, p.add_imu_s_count = np.array(beg, np.int32), wb.imu_s_count[new_terms]
    if st:
        p.add_imu_s_t, p.add_imu_s_gyr, p.add_imu_s_acc = np.concatenate(st), np.concatenate(sg), np.concatenate(sa_)
    # explicit IMU removal is only needed for a term whose blocks all stay
    p.remove_imu = np.array([n for n, f in enumerate(A.imu_terms)
                             if f not in B.imu_terms and all(x in B.pose_ids for x in (("f", int(W.imu_pose0[f])), ("f", int(W.imu_pose1[f]))))
                             and int(W.imu_sb0[f]) in B.sb_ids and int(W.imu_sb1[f]) in B.sb_ids], np.int32)
    if B.pose_prior_on is not None:
        p.replace |= PATCH_POSE_PRIORS
        p.pprior_pose, p.pprior_meas, p.pprior_sqrtinfo = wb.pprior_pose, wb.pprior_meas, wb.pprior_sqrtinfo
    if B.sb_prior_on is not None:
        p.replace |= PATCH_SB_PRIORS
        p.sbprior_sb, p.sbprior_meas, p.sbprior_sqrtinfo = wb.sbprior_sb, wb.sbprior_meas, wb.sbprior_sqrtinfo
    if B.marg is not None or A.marg is not None:
        p.replace |= PATCH_MARG_PRIOR
        if B.marg is not None:
            p.marg_block_type, p.marg_block_idx, p.marg_block_off = wb.marg_block_type, wb.marg_block_idx, wb.marg_block_off
            p.marg_J, p.marg_e0, p.marg_lin = wb.marg_J, wb.marg_e0, wb.marg_lin
    return p


def imu_terms_equal(a: Window, b: Window) -> bool:
    if a.n_imu != b.n_imu:
        return False
    for n in ("imu_pose0", "imu_sb0", "imu_pose1", "imu_sb1", "imu_t0", "imu_t1", "imu_s_count"):
        if not np.array_equal(np.asarray(getattr(a, n)), np.asarray(getattr(b, n))):
            return False
    for f in range(a.n_imu):
        ba, bb, c = int(a.imu_s_begin[f]), int(b.imu_s_begin[f]), int(a.imu_s_count[f])
        for n in ("imu_s_t", "imu_s_gyr", "imu_s_acc"):
            if not np.array_equal(np.asarray(getattr(a, n))[ba:ba + c], np.asarray(getattr(b, n))[bb:bb + c]):
                return False
    return True


NON_IMU_FIELDS = ("pose", "pose_fixed", "sb", "sb_fixed", "lm", "cam_intr", "cam_model", "obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_uv",
                  "obs_sqrtw", "pprior_pose", "pprior_meas", "pprior_sqrtinfo", "sbprior_sb", "sbprior_meas", "sbprior_sqrtinfo", "rel_pose0",
                  "rel_pose1", "rel_sqrtinfo", "marg_block_type", "marg_block_idx", "marg_block_off", "marg_J", "marg_e0", "marg_lin")


def windows_differ(a: Window, b: Window):
    """Names of the fields in which two windows differ (IMU terms compared term by term)."""
    bad = [n for n in NON_IMU_FIELDS
           if not np.array_equal(np.asarray(getattr(a, n)).reshape(-1), np.asarray(getattr(b, n)).reshape(-1))]
    if a.cauchy_b != b.cauchy_b:
        bad.append("cauchy_b")
    if not imu_terms_equal(a, b):
        bad.append("imu")
    return bad


def sliding_pair(seed=7, K=6, L=80, n_new_lm=12, n_drop_lm=9, n_drop_obs=15):
    """Carvings A (frames 0..K-1) and B (frames 1..K) of one (K+1)-frame window: what one frame of the estimator changes - the oldest
    frame leaves with its terms, some landmarks leave, some observations are dropped, a frame with its IMU term, observations and
    new landmarks arrives, the priors move to the new oldest frame and a dense prior over it replaces the one A carried."""
    rng = np.random.default_rng(seed)
    W = synthetic.make_window(K + 1, L, 0.7, seed=seed, frame_dt=0.2)
    ext = [("e", 0), ("e", 1)]
    lm_all = list(range(L))
    new_lm = sorted(rng.choice(L, n_new_lm, replace=False).tolist())              # only B has them
    a_lm = [l for l in lm_all if l not in new_lm]
    drop_lm = sorted(rng.choice(a_lm, n_drop_lm, replace=False).tolist())         # only A has them
    b_lm = [l for l in a_lm if l not in drop_lm] + new_lm
    a_pose = [("f", k) for k in range(K)] + ext
    b_pose = [("f", k) for k in range(1, K)] + ext + [("f", K)]
    cand = [(int(W.obs_lm[i]), int(W.obs_pose[i]), int(W.obs_cam[i])) for i in range(W.n_obs)
            if W.obs_lm[i] in b_lm and W.obs_lm[i] in a_lm and 1 <= W.obs_pose[i] < K]
    drop = [cand[i] for i in rng.choice(len(cand), n_drop_obs, replace=False)]
    A = Carving(W, a_pose, list(range(K)), a_lm, marg=make_marg(W, [("p", ("f", 0)), ("s", 0)], rng))
    B = Carving(W, b_pose, list(range(1, K)) + [K], b_lm, drop_obs=drop, pose_prior_on=[("f", 1)], sb_prior_on=[1],
                marg=make_marg(W, [("p", ("f", 1)), ("s", 1), ("p", ("f", 2))], rng))
    return A, B
