"""Host-side handle of a batch of windows on one MI355X (thin wrapper over the C-ABI).

Mirrors how ``okvis::Estimator`` drives its backend: upload the window structure (what
``addStates/addLandmark/addObservation`` build in the reference, okvis_ceres/src/Estimator.cpp:110-365,
implementation/Estimator.hpp:43-90), ``optimize(numIter)`` (Estimator.cpp:843-906), read the states back
(``get_T_WS/getSpeedAndBias/getLandmark``, Estimator.cpp:933-1200).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .window import LimitsC, OptionsC, SummaryC, Window, WindowC, default_options

ARR = dict(POSE=0, SB=1, LM=2, OBS_RESIDUAL=3, LM_V=4, LM_B=5, LM_HQ=6, PAIR_W=7, REDUCED_S=8,
           REDUCED_RHS=9, STEP=10, LM_QUALITY=11, GRADIENT=12, IMU_RESIDUAL=13, HPP=14, DAMPING=15, IMU_SB_REF=16, PROF=99, IMU_REDO_COUNT=98, CTRL=97, SLOTS=96)
_dp = C.POINTER(C.c_double)


def limits() -> dict:
    lim = LimitsC()
    _lib.lib().okvis_ba_get_limits(C.byref(lim))
    return {n: getattr(lim, n) for n, _ in lim._fields_}


def check_window(window: Window, options: OptionsC | None = None) -> dict:
    """Host-only structure check (no GPU needed): the index building okvis_ba_upload performs."""
    window.validate()
    wc, keep = window.as_c()
    st = (C.c_int64 * 8)()
    _lib.check(_lib.lib().okvis_ba_check_window(C.byref(wc), C.byref(options) if options is not None else None, st),
               "check_window")
    del keep
    return dict(D=st[0], Dp=st[1], n_pair=st[2], n_group=st[3], n_chunk=st[4], n_task=st[5], gpart=st[6],
                arena_bytes=st[7])


INDEX_LISTS = ("groups", "lm_obs_begin", "lm_pair_begin", "pair_lm", "pair_block", "pair_off", "pair_role", "lm_piece_begin",
               "pair_piece", "pair_list_begin", "pair_list", "tasks", "task_list", "chunks", "chunk_diag_begin", "chunk_diag_out",
               "chunk_desc", "piece_path", "ldl_comp", "chain")   # OKVIS_BA_LIST_* in this order


def index_lists(window: Window, options: OptionsC | None = None, n_windows: int = 1) -> dict:
    """The index lists okvis_ba_upload would build for `window` as one of `n_windows` (host only; okvis_ba_check_window_lists)."""
    wc, keep = window.as_c()
    L = _lib.lib()
    po = C.byref(options) if options is not None else None
    out = {}
    for which, name in enumerate(INDEX_LISTS):
        n = C.c_int64()
        rc = L.okvis_ba_check_window_lists(C.byref(wc), po, n_windows, which, None, 0, C.byref(n))
        if rc != 0 and not (rc == -1 and n.value > 0):
            _lib.check(rc, "check_window_lists")
        a = np.zeros(max(n.value, 1), np.int32)
        if n.value:
            _lib.check(L.okvis_ba_check_window_lists(C.byref(wc), po, n_windows, which, a.ctypes.data_as(C.POINTER(C.c_int32)), n.value,
                                                     C.byref(n)), "check_window_lists")
        out[name] = a[:n.value]
    del keep
    out["groups"] = out["groups"].reshape(-1, 16)
    out["tasks"] = out["tasks"].reshape(-1, 6)
    out["chunks"] = out["chunks"].reshape(-1, 2)
    out["piece_path"] = int(out["piece_path"][0])
    out["ldl_comp"] = int(out["ldl_comp"][0]) & 0xFFFFFFFF
    out["chain"] = int(out["chain"][0])
    return out


class WindowStore:
    """Host-side window container with O(edit) structure updates (okvis_ba_store_*; no GPU needed)."""

    def __init__(self, window: Window):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        wc, keep = window.as_c()
        _lib.check(self._L.okvis_ba_store_create(C.byref(wc), C.byref(self._h)), "store_create")
        del keep

    def patch(self, patch) -> int:
        """Apply a :class:`okvis_amd.window.Patch`; returns the status (0 = applied, otherwise the window is untouched)."""
        pc, keep = patch.as_c()
        rc = self._L.okvis_ba_store_patch(self._h, C.byref(pc))
        del keep
        return rc

    def view(self) -> Window:
        from .window import window_from_c
        wc = WindowC()
        _lib.check(self._L.okvis_ba_store_view(self._h, C.byref(wc)), "store_view")
        return window_from_c(wc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.okvis_ba_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WindowBatch:
    """A batch of independent sliding windows resident in the HBM of one GPU."""

    def __init__(self, windows: Sequence[Window], device: int = 0, options: OptionsC | None = None, patchable: bool = False):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        _lib.check(self._L.okvis_ba_create(C.byref(self._h), int(device)), "create")
        self.options = options or default_options()
        _lib.check(self._L.okvis_ba_set_options(self._h, C.byref(self.options)), "set_options")
        if patchable:   # the solver keeps a container of every uploaded window (okvis_ba_patch_window)
            _lib.check(self._L.okvis_ba_set_patchable(self._h, 1), "set_patchable")
        self.upload(windows)

    def patch(self, w: int, patch):
        """okvis_ba_patch_window: edit window w on the solver (blocks that stay keep the device's values)."""
        pc, keep = patch.as_c()
        _lib.check(self._L.okvis_ba_patch_window(self._h, int(w), C.byref(pc)), "patch_window")
        del keep
        self.windows[w] = self.patched_view(w)

    def patched_view(self, w: int) -> Window:
        """The solver's container of window w as a :class:`Window` (current device values)."""
        from .window import window_from_c
        wc = WindowC()
        _lib.check(self._L.okvis_ba_patched_view(self._h, int(w), C.byref(wc)), "patched_view")
        return window_from_c(wc)

    def upload(self, windows: Sequence[Window]):
        """okvis_ba_upload on this solver: replaces the windows.  Device allocations are kept (grow-only) and so are the captured
        launch graphs when the new windows have the shapes of the old ones."""
        self.windows = list(windows)
        arr = (WindowC * len(self.windows))()
        keep = []
        for i, w in enumerate(self.windows):
            w.validate()
            wc, k = w.as_c()
            arr[i] = wc
            keep.append(k)
        _lib.check(self._L.okvis_ba_upload(self._h, len(self.windows), arr), "upload")
        del keep

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.okvis_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.windows)

    # ---- options ----
    def set_options(self, options: OptionsC):
        _lib.check(self._L.okvis_ba_set_options(self._h, C.byref(options)), "set_options")
        self.options = options

    # ---- hot path ----
    def optimize(self, num_iter: int):
        s = (SummaryC * len(self))()
        _lib.check(self._L.okvis_ba_optimize(self._h, int(num_iter), s), "optimize")
        return [x.as_dict() for x in s]

    def optimize_timed(self, max_iter: int, min_iter: int, time_limit_s: float):
        s = (SummaryC * len(self))()
        _lib.check(self._L.okvis_ba_optimize_timed(self._h, int(max_iter), int(min_iter), float(time_limit_s), s),
                   "optimize_timed")
        return [x.as_dict() for x in s]

    def begin(self):
        _lib.check(self._L.okvis_ba_begin(self._h), "begin")

    def iterate(self, n: int):
        _lib.check(self._L.okvis_ba_iterate(self._h, int(n)), "iterate")

    def finish(self):
        s = (SummaryC * len(self))()
        _lib.check(self._L.okvis_ba_finish(self._h, s), "finish")
        return [x.as_dict() for x in s]

    def evaluate_cost(self):
        c = np.zeros(len(self))
        _lib.check(self._L.okvis_ba_evaluate_cost(self._h, c.ctypes.data_as(_dp)), "evaluate_cost")
        return c

    def marginalize(self, w: int, pose_marg, sb_marg, prior=None) -> dict:
        """okvis_ba_marginalize on window w (all its landmarks + the flagged blocks are eliminated)."""
        from .window import marg_call
        win = self.windows[w]
        st, out = marg_call(lambda sp, rs: self._L.okvis_ba_marginalize(self._h, int(w), sp, rs), win.n_pose, win.n_sb,
                            pose_marg, sb_marg, prior)
        _lib.check(st, "marginalize")
        return out

    def synchronize(self):
        _lib.check(self._L.okvis_ba_synchronize(self._h), "synchronize")

    # ---- results ----
    def get_state(self, w: int = 0):
        W = self.windows[w]
        pose, sb, lm = np.zeros((W.n_pose, 7)), np.zeros((W.n_sb, 9)), np.zeros((W.n_lm, 4))
        _lib.check(self._L.okvis_ba_get_state(self._h, w, pose.ctypes.data_as(_dp), sb.ctypes.data_as(_dp),
                                              lm.ctypes.data_as(_dp)), "get_state")
        return pose, sb, lm

    def fetch_results(self, w: int = 0):
        """okvis_ba_fetch_results: state, landmark quality and the IMU caches' reference biases with one synchronisation."""
        W = self.windows[w]
        pose, sb, lm = np.zeros((W.n_pose, 7)), np.zeros((W.n_sb, 9)), np.zeros((W.n_lm, 4))
        q, ref = np.zeros(W.n_lm), np.zeros((W.n_imu, 9))
        _lib.check(self._L.okvis_ba_fetch_results(self._h, w, pose.ctypes.data_as(_dp), sb.ctypes.data_as(_dp), lm.ctypes.data_as(_dp),
                                                  q.ctypes.data_as(_dp), ref.ctypes.data_as(_dp)), "fetch_results")
        return dict(pose=pose, sb=sb, lm=lm, quality=q, imu_sb_ref=ref)

    def fetch_imu_caches(self, w: int = 0):
        """okvis_ba_fetch_imu_caches: the preintegration records of window w's IMU terms, [n_imu, 290] (Window.imu_cache, flag 2)."""
        from .window import IMU_CACHE_DOUBLES
        out = np.zeros((self.windows[w].n_imu, IMU_CACHE_DOUBLES))
        _lib.check(self._L.okvis_ba_fetch_imu_caches(self._h, w, out.ctypes.data_as(_dp)), "fetch_imu_caches")
        return out

    def set_state(self, w: int, pose=None, sb=None, lm=None):
        def p(a):
            return None if a is None else np.ascontiguousarray(a, np.float64)
        pose, sb, lm = p(pose), p(sb), p(lm)
        _lib.check(self._L.okvis_ba_set_state(
            self._h, w, None if pose is None else pose.ctypes.data_as(_dp),
            None if sb is None else sb.ctypes.data_as(_dp), None if lm is None else lm.ctypes.data_as(_dp)), "set_state")

    def array(self, name: str, w: int = 0) -> np.ndarray:
        n = C.c_int64()
        _lib.check(self._L.okvis_ba_array_size(self._h, w, ARR[name], C.byref(n)), f"array_size {name}")
        out = np.zeros(max(n.value, 0))
        _lib.check(self._L.okvis_ba_download(self._h, w, ARR[name], out.ctypes.data_as(_dp), n.value), f"download {name}")
        return out

    def reduced_dim(self, w: int = 0) -> int:
        d = C.c_int32()
        _lib.check(self._L.okvis_ba_reduced_dim(self._h, w, C.byref(d)))
        return d.value

    def pairs(self, w: int = 0):
        n = C.c_int32()
        _lib.check(self._L.okvis_ba_pair_count(self._h, w, C.byref(n)))
        a, b = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
        ip = C.POINTER(C.c_int32)
        _lib.check(self._L.okvis_ba_pairs(self._h, w, a.ctypes.data_as(ip), b.ctypes.data_as(ip)))
        return a, b

    # ---- measurement hooks ----
    def last_iterate_ms(self) -> float:
        t = C.c_float()
        _lib.check(self._L.okvis_ba_last_iterate_ms(self._h, C.byref(t)))
        return t.value

    def profile_iterations(self, n: int):
        ms = (C.c_float * 4)()
        _lib.check(self._L.okvis_ba_profile_iterations(self._h, int(n), ms))
        return dict(schur=ms[0], solve=ms[1], small=ms[2], linearize=ms[3])

    def profile_launches(self, n: int):
        """per-iteration launch durations [n][3] in ms: schur, solve, linearise (incl. the IMU / prior workgroups)"""
        ms = (C.c_float * (4 * int(n)))()
        _lib.check(self._L.okvis_ba_profile_launches(self._h, int(n), ms))
        a = np.array(ms[:], dtype=np.float64).reshape(int(n), 4)
        return dict(schur=a[:, 0], solve=a[:, 1], linearize=a[:, 2] + a[:, 3])

    def helper_timeouts(self) -> int:
        """okvis_ba_helper_timeouts: solve launches whose helper workgroups were late (0 in a healthy run)"""
        n = C.c_int64()
        _lib.check(self._L.okvis_ba_helper_timeouts(self._h, C.byref(n)))
        return n.value

    ROUTE = ("windows", "fused", "decision_free_schur", "piece_path", "split_small", "sub_batches", "sub_batch_max_windows",
             "schur_kernel", "solve_dbuf", "solve_tiled", "solve_helpers", "graph", "max_chunks", "slots", "solve_mode", "small_rides")

    def launch_route(self) -> dict:
        """okvis_ba_launch_route: which launches this batch takes under the current options (read-only)"""
        r = (C.c_int32 * 16)()
        _lib.check(self._L.okvis_ba_launch_route(self._h, r), "launch_route")
        return {n: int(r[i]) for i, n in enumerate(self.ROUTE)}

    def algorithmic_bytes(self):
        v = [C.c_int64() for _ in range(4)]
        _lib.check(self._L.okvis_ba_algorithmic_bytes(self._h, *[C.byref(x) for x in v]))
        return dict(linearize=v[0].value, schur=v[1].value, solve=v[2].value, small=v[3].value)
