"""CPU tests of the index build behind okvis_ba_upload (what replaces okvis::ceres::Map's book-keeping of residual and parameter
blocks, reference okvis_ceres/src/Map.cpp:292-565, for the device): the lists okvis_ba_check_window_lists hands out against a plain
Python restatement of the documented rules — reduced ordering, (landmark, block) pairs, linearise groups, the pieces of the piece
path (ba_linearize2.hpp), reduction tasks and their slot lists, Schur chunks and their per-block partial lists."""
import numpy as np
import pytest

from okvis_amd import solver, synthetic
from okvis_amd.window import default_options

GROUP_OBS, GROUP_PAIRS, LIN2_PIECES, SCHUR_DESC_INTS = 256, 512, 128, 20


def restate(w, n_windows):
    """The documented rules, landmark by landmark, without any of the product's shortcuts."""
    obs_lm, obs_pose, obs_ext = (np.asarray(a) for a in (w.obs_lm, w.obs_pose, w.obs_ext))
    n_pose, n_lm = w.n_pose, w.n_lm
    pose_off, off = [], 0
    for i in range(n_pose):
        pose_off.append(-1 if w.pose_fixed[i] else off)
        off += 0 if w.pose_fixed[i] else 6
    role = {}
    for ip, ie in zip(obs_pose, obs_ext):
        role[int(ip)], role[int(ie)] = 0, 1
    lm_obs_begin = np.searchsorted(obs_lm, np.arange(n_lm + 1))
    has_ext = any(pose_off[int(e)] >= 0 for e in obs_ext)
    pairs, lm_pair_begin = [], [0]
    for l in range(n_lm):
        o0, o1 = lm_obs_begin[l], lm_obs_begin[l + 1]
        blocks = sorted({int(b) for b in np.concatenate([obs_pose[o0:o1], obs_ext[o0:o1]]) if pose_off[int(b)] >= 0})
        pairs += [(l, b, pose_off[b], role[b]) for b in blocks]
        lm_pair_begin.append(len(pairs))

    def pieces(l, lane0):
        """[(pose, first group-local lane, observations)] of landmark l when its first observation takes lane0"""
        out, o0, o1 = [], lm_obs_begin[l], lm_obs_begin[l + 1]
        o = o0
        while o < o1:
            e = o + 1   # a run: one pose, inside one row of 16 lanes
            while e < o1 and obs_pose[e] == obs_pose[o] and (lane0 + (e - o0)) % 16 != 0:
                e += 1
            for k in range(o, e, 2):
                out.append((int(obs_pose[o]), lane0 + (k - o0), min(2, e - k)))
            o = e
        return out

    lin2 = not has_ext and all(len(pieces(l, 0)) <= LIN2_PIECES for l in range(n_lm))
    cap = 16 if n_windows <= 8 else 32
    groups, l = [], 0
    while l < n_lm:
        g = dict(lm_begin=l, no=0, np=0, npc=0, pieces=[])
        while l < n_lm:
            lo, lp = lm_obs_begin[l + 1] - lm_obs_begin[l], lm_pair_begin[l + 1] - lm_pair_begin[l]
            pc = pieces(l, g["no"]) if lin2 else []
            if l > g["lm_begin"] and (g["no"] + lo > GROUP_OBS or g["np"] + lp > GROUP_PAIRS or l - g["lm_begin"] + 1 > cap or
                                      g["npc"] + len(pc) > LIN2_PIECES):
                break
            g["no"] += lo; g["np"] += lp; g["npc"] += len(pc); g["pieces"].append((l, pc))
            l += 1
        g["lm_end"] = l
        groups.append(g)
    return dict(pose_off=pose_off, lm_obs_begin=lm_obs_begin, lm_pair_begin=np.array(lm_pair_begin), pairs=pairs, groups=groups,
                lin2=lin2, role=role)


WINDOWS = {
    "frame": lambda: synthetic.make_window(8, 430, 0.5, seed=20240924),
    "A": lambda: synthetic.config_A(),
    "sparse": lambda: synthetic.config_A(visibility=0.35, seed=3),
    "small": lambda: synthetic.small_window(seed=4, K=5, L=70),
    "mono_rows": lambda: _repeat_observations(synthetic.small_window(seed=8, K=6, L=50, visibility=0.6), 3),
    "shared": lambda: synthetic.small_window(seed=11, K=5, L=80, estimate_extrinsics="shared"),
    "perframe": lambda: synthetic.small_window(seed=12, K=5, L=80, estimate_extrinsics="perframe"),
}


def _repeat_observations(w, times):
    """Every observation `times` times (legal: several keypoints of one image matched to one landmark): runs longer than two
    observations and runs that cross a row of 16 lanes."""
    keep = None
    for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_uv", "obs_sqrtw"):
        a = np.repeat(np.asarray(getattr(w, n)), times, axis=0)
        if keep is None:   # ... of every length and parity: every fifth and every seventh observation is dropped again
            keep = (np.arange(len(a)) % 5 != 3) & (np.arange(len(a)) % 7 != 2)
        setattr(w, n, a[keep])
    return w


@pytest.mark.parametrize("name", sorted(WINDOWS))
@pytest.mark.parametrize("n_windows", [1, 64])
def test_index_lists_follow_the_documented_rules(name, n_windows):
    w = WINDOWS[name]()
    L = solver.index_lists(w, default_options(), n_windows)
    R = restate(w, n_windows)
    assert L["piece_path"] == int(R["lin2"]) == int(name not in ("shared", "perframe"))
    # pairs
    assert np.array_equal(L["lm_obs_begin"], R["lm_obs_begin"]) and np.array_equal(L["lm_pair_begin"], R["lm_pair_begin"])
    assert [tuple(int(x) for x in t) for t in zip(L["pair_lm"], L["pair_off"], L["pair_role"])] == [(p[0], p[2], p[3]) for p in R["pairs"]]
    if R["lin2"]:
        assert list(L["pair_block"]) == [p[1] for p in R["pairs"]]
    # groups
    G = L["groups"]
    assert len(G) == len(R["groups"])
    obs_pose = np.asarray(w.obs_pose)
    piece_total, task_total, gpart = 0, 0, 0
    for g, rg in zip(G, R["groups"]):
        lm_b, lm_e, ob, oe, pb, pe, tb, te, plb, ple, tlb, tle, piece_begin, pw1, pw2, pw3 = (int(x) for x in g)
        assert (lm_b, lm_e) == (rg["lm_begin"], rg["lm_end"])
        assert (ob, oe) == (R["lm_obs_begin"][lm_b], R["lm_obs_begin"][lm_e]) and (pb, pe) == (R["lm_pair_begin"][lm_b], R["lm_pair_begin"][lm_e])
        assert oe - ob <= GROUP_OBS and pe - pb <= GROUP_PAIRS and tb == task_total
        blocks = sorted({R["pairs"][p][1] for p in range(pb, pe)})
        if R["lin2"]:
            # pieces: window-wide begin per landmark, group-local first piece | count << 16 per pair, pieces before waves 1..3
            assert piece_begin == piece_total
            local, waves = 0, [0, 0, 0, 0]
            for l, pcs in rg["pieces"]:
                assert L["lm_piece_begin"][l] == piece_total + local
                for p in range(R["lm_pair_begin"][l], R["lm_pair_begin"][l + 1]):
                    mine = [i for i, pc in enumerate(pcs) if pc[0] == R["pairs"][p][1]]
                    assert mine == list(range(mine[0], mine[0] + len(mine)))   # the pieces of a pair are contiguous
                    assert L["pair_piece"][p] == (local + mine[0]) | (len(mine) << 16)
                for pc in pcs:
                    waves[pc[1] // 64] += 1
                local += len(pcs)
            assert local <= LIN2_PIECES and (pw1, pw2, pw3) == (waves[0], waves[0] + waves[1], waves[0] + waves[1] + waves[2])
            piece_total += local
            # tasks: one per free block of the group, ascending; the slot of a pair's block record in task_list
            assert te - tb == len(blocks) and tle - tlb == pe - pb
            slot = 0
            for t, b in zip(range(tb, te), blocks):
                mine = [p - pb for p in range(pb, pe) if R["pairs"][p][1] == b]
                assert tuple(L["tasks"][t]) == (0, R["pose_off"][b], -1, tlb + slot, tlb + slot + len(mine), gpart)
                assert [int(L["task_list"][tlb + p]) for p in mine] == list(range(slot, slot + len(mine)))
                slot += len(mine)
                gpart += 27
        else:
            # staged path: per pair the group-local observations that touch its block; tasks per block (own role), then one per
            # (pose, extrinsics) pair of free blocks
            obs_ext = np.asarray(w.obs_ext)
            for p in range(pb, pe):
                l, b = R["pairs"][p][0], R["pairs"][p][1]
                o0, o1 = R["lm_obs_begin"][l], R["lm_obs_begin"][l + 1]
                want = [o - ob for o in range(o0, o1) if obs_pose[o] == b or obs_ext[o] == b]
                got = L["pair_list"][L["pair_list_begin"][p]:(L["pair_list_begin"][p + 1] if p + 1 < pe else ple)]
                assert list(got) == want
            t = tb
            for b in blocks:
                want = [o - ob for o in range(ob, oe) if obs_pose[o] == b or obs_ext[o] == b]
                ty, oa, obb, lb, le, out = (int(x) for x in L["tasks"][t])
                assert (ty, oa, obb, out) == (R["role"][b], R["pose_off"][b], -1, gpart) and list(L["task_list"][lb:le]) == want
                gpart += 27
                t += 1
            cross = sorted({(int(obs_pose[o]), int(obs_ext[o])) for o in range(ob, oe)
                            if R["pose_off"][int(obs_pose[o])] >= 0 and R["pose_off"][int(obs_ext[o])] >= 0})
            for ip, ie in cross:
                want = [o - ob for o in range(ob, oe) if obs_pose[o] == ip and obs_ext[o] == ie]
                ty, oa, obb, lb, le, out = (int(x) for x in L["tasks"][t])
                assert (ty, oa, obb, out) == (2, R["pose_off"][ip], R["pose_off"][ie], gpart) and list(L["task_list"][lb:le]) == want
                gpart += 36
                t += 1
            assert t == te
        task_total = te
    if R["lin2"]:
        assert L["lm_piece_begin"][w.n_lm] == piece_total
    # chunks: consecutive groups, every group once; per chunk and pose block the partials of that block in (group, task) order
    Ck = L["chunks"]
    assert Ck[0][0] == 0 and Ck[-1][1] == len(G) and all(Ck[i][1] == Ck[i + 1][0] for i in range(len(Ck) - 1))
    st = solver.check_window(w)
    nb = st["Dp"] // 6
    for c, (g0, g1) in enumerate(Ck):
        per_block = [[] for _ in range(nb)]
        for t in range(int(G[g0][6]), int(G[g1 - 1][7])):
            if L["tasks"][t][0] < 2:
                per_block[int(L["tasks"][t][1]) // 6].append(int(L["tasks"][t][5]))
        for b in range(nb):
            lo, hi = L["chunk_diag_begin"][c * nb + b], L["chunk_diag_begin"][c * nb + b + 1]
            assert list(L["chunk_diag_out"][lo:hi]) == per_block[b]
        d = L["chunk_desc"][c * SCHUR_DESC_INTS:(c + 1) * SCHUR_DESC_INTS]
        lb, le = int(G[g0][0]), int(G[g1 - 1][1])
        assert (d[0], d[1]) == (lb, le) and [int(x) for x in d[2:19]] == [int(R["lm_pair_begin"][min(lb + 4 * i, le)]) for i in range(17)]
    if n_windows == 1 and name in ("frame", "A"):
        assert np.array_equal(Ck, np.stack([np.arange(len(G)), np.arange(len(G)) + 1], 1))   # fused mode: chunk = group
    if n_windows == 64:   # separate Schur launch: chunks of at most 48 landmarks (a group more would exceed them), at least one group
        for g0, g1 in Ck:
            nl = sum(int(G[g][1] - G[g][0]) for g in range(g0, g1))
            assert g1 > g0 and (nl <= 48 or g1 == g0 + 1)
            if g1 < len(G):
                assert nl + int(G[g1][1] - G[g1][0]) > 48


def test_groups_hold_fewer_landmarks_when_few_windows_share_the_device():
    w = synthetic.make_window(8, 430, 0.5, seed=20240924)
    few, many = solver.index_lists(w, None, 8)["groups"], solver.index_lists(w, None, 9)["groups"]
    assert (few[:, 1] - few[:, 0]).max() == 16 and (many[:, 1] - many[:, 0]).max() == 32


def test_index_lists_report_their_size_before_they_write():
    import ctypes as C
    from okvis_amd import _lib
    w = synthetic.small_window(seed=2)
    wc, keep = w.as_c()
    L = _lib.lib()
    n = C.c_int64()
    assert L.okvis_ba_check_window_lists(C.byref(wc), None, 1, 3, None, 0, C.byref(n)) == -1 and n.value > 0
    buf = (C.c_int32 * (n.value + 1))(*([-7] * (n.value + 1)))
    assert L.okvis_ba_check_window_lists(C.byref(wc), None, 1, 3, buf, n.value - 1, C.byref(n)) == -1 and buf[0] == -7
    assert L.okvis_ba_check_window_lists(C.byref(wc), None, 1, 3, buf, n.value, C.byref(n)) == 0 and buf[n.value] == -7
    assert L.okvis_ba_check_window_lists(C.byref(wc), None, 1, 99, buf, n.value, C.byref(n)) == -1
    assert L.okvis_ba_check_window_lists(C.byref(wc), None, 0, 3, buf, n.value, C.byref(n)) == -1
    assert L.okvis_ba_check_window_lists(None, None, 1, 3, buf, n.value, C.byref(n)) == -1


def _expected_ldl_comp(w, blocks, chain=False):
    """blocks: (type, index) of the blocks that carry a pose prior or the marginalisation prior; the dense solver numbers the
    speed/bias part first (L16::perm) and cuts the system into 16-wide diagonal blocks; the chain solver (ba_chain.hpp) hands
    ldl16_solve the pose system alone: its 16-blocks, pose blocks only"""
    free_p = [i for i in range(w.n_pose) if not w.pose_fixed[i]]
    free_s = [i for i in range(w.n_sb) if not w.sb_fixed[i]]
    Dp, Ds = 6 * len(free_p), 9 * len(free_s)
    if chain:
        Ds = 0
    m = 0
    for t, i in blocks:
        if t == 0 and i in free_p:
            rows = range(Ds + 6 * free_p.index(i), Ds + 6 * free_p.index(i) + 6)
        elif t == 1 and i in free_s and not chain:
            rows = range(9 * free_s.index(i), 9 * free_s.index(i) + 9)
        else:
            continue
        for r in rows:
            m |= 1 << (r >> 4)
    return m


def test_blocks_with_a_prior_are_marked_for_the_compensated_elimination():
    """WinPtrs::ldl_comp (ba_ldl16.hpp): the diagonal blocks of the dense solver that hold columns of a pose prior or of the
    marginalisation prior, in the solver's ordering; nothing for windows above the LDS solver's size; the tuning flag OKVIS_BA_TUNE_NO_LDL_COMP clears it"""
    from okvis_amd.window import SOLVE_CHAIN, SOLVE_DENSE, TUNE_LDL_COMP_ALL, TUNE_NO_LDL_COMP, default_options, set_options
    dense = set_options(default_options(), tuning_solve_mode=SOLVE_DENSE)
    chain = set_options(default_options(), tuning_solve_mode=SOLVE_CHAIN)
    wA = synthetic.config_A()
    assert solver.index_lists(wA, dense)["ldl_comp"] == _expected_ldl_comp(wA, [(0, int(p)) for p in wA.pprior_pose]) == 1 << 5
    assert solver.index_lists(wA, dense)["chain"] == 0
    # the chain solver (the default for a window whose speed/bias blocks form a chain): ten blocks, and the pose system's own block 0
    for o in (None, chain):
        il = solver.index_lists(wA, o)
        assert il["chain"] == 10 and il["ldl_comp"] == _expected_ldl_comp(wA, [(0, int(p)) for p in wA.pprior_pose], chain=True) == 1
    w = synthetic.small_window(seed=41, K=5, L=60)
    assert solver.index_lists(w, dense)["ldl_comp"] == _expected_ldl_comp(w, [(0, int(p)) for p in w.pprior_pose]) == 0b1100
    assert solver.index_lists(w)["chain"] == 0 and solver.index_lists(w, chain)["chain"] == 5 and solver.index_lists(w, chain)["ldl_comp"] == 1   # (auto: from 8 blocks on)
    # a dense prior over two poses and two speed/bias blocks that are NOT neighbours: no chain, whatever the options ask for
    rng = np.random.default_rng(6)
    wm = synthetic.small_window(seed=6, K=5, L=70)
    Dm = 6 + 9 + 6 + 9
    wm.marg_J = np.triu(rng.standard_normal((Dm, Dm))) * 3.0
    wm.marg_e0 = rng.standard_normal(Dm) * 0.1
    wm.marg_block_type = np.array([0, 1, 0, 1], np.int32)
    wm.marg_block_idx = np.array([0, 0, 1, 2], np.int32)
    wm.marg_block_off = np.array([0, 6, 15, 21], np.int32)
    lin = np.zeros((4, 9))
    lin[0, :7] = wm.pose[0]; lin[1] = wm.sb[0]; lin[2, :7] = wm.pose[1]; lin[3] = wm.sb[2]
    wm.marg_lin = lin
    want = _expected_ldl_comp(wm, [(0, int(p)) for p in wm.pprior_pose] + [(0, 0), (1, 0), (0, 1), (1, 2)])
    for o in (None, dense, chain):
        il = solver.index_lists(wm, o)
        assert il["chain"] == 0 and il["ldl_comp"] == want and bin(want).count("1") >= 3
    # ... over two NEIGHBOURS it is one: the prior's pose columns are marked in the pose system, its speed/bias columns nowhere
    wm.marg_block_idx = np.array([0, 0, 1, 1], np.int32)
    lin[3] = wm.sb[1]
    wm.marg_lin = lin
    il = solver.index_lists(wm, chain)
    assert il["chain"] == 5 and il["ldl_comp"] == _expected_ldl_comp(wm, [(0, int(p)) for p in wm.pprior_pose] + [(0, 0), (0, 1)], chain=True)
    wl = synthetic.make_window(20, 30, 1.0, 2, frame_dt=0.1)
    assert wl.reduced_dim() == 300 and solver.index_lists(wl)["ldl_comp"] == 0 and solver.index_lists(wl)["chain"] == 0
    # one speed/bias block: the chain solver on request only
    w1 = synthetic.small_window(seed=3, K=4, L=30)
    w1.sb_fixed = np.array([0, 1, 1, 1], np.uint8)
    assert solver.index_lists(w1)["chain"] == 0 and solver.index_lists(w1, chain)["chain"] == 1
    assert solver.index_lists(wA, set_options(default_options(), tuning_flags=TUNE_NO_LDL_COMP))["ldl_comp"] == 0
    assert solver.index_lists(wA, set_options(default_options(), tuning_flags=TUNE_LDL_COMP_ALL))["ldl_comp"] == 0xFFFFFFFF
