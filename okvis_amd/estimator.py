"""ctypes view of the C++ host class ``okvis_amd::Estimator`` (okvis_amd/csrc/host/estimator.hpp), the
mirror of ``okvis::Estimator`` (reference okvis_ceres/include/okvis/Estimator.hpp:77-581).  Used by the
tests that re-state the reference's integration test; C++ callers use the class directly."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .window import SummaryC

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OKVIS_AMD_LIB_DIR: a directory holding an instrumented build of the same sources — scripts/host_sanitize.sh)
LIB_PATH = os.path.join(os.environ.get("OKVIS_AMD_LIB_DIR") or os.path.join(_HERE, "lib"), "libokvis_amd_estimator.so")
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_lib = None


def declare(L, prefix="okvis_est_"):
    """ctypes signatures of the flat estimator API exported by library L under `prefix` (the reference build in
    oracle/_ref exports the same entry points as ref_est_*, see tests/ref_lib.py)."""
    f = lambda n: getattr(L, prefix + n)  # noqa: E731
    f("last_error").restype = C.c_char_p
    f("create").restype = C.c_void_p
    f("create").argtypes = [C.c_int]
    f("destroy").argtypes = [C.c_void_p]
    f("add_camera").argtypes = [C.c_void_p, _dp]
    f("add_imu").argtypes = [C.c_void_p, _dp]
    f("frame_create").restype = C.c_void_p
    f("frame_create").argtypes = [C.c_uint64, C.c_int64, C.c_int, _dp, _dp, C.POINTER(C.c_int)]
    f("frame_destroy").argtypes = [C.c_void_p]
    f("frame_add_keypoint").argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
    f("add_states").argtypes = [C.c_void_p, C.c_void_p, C.c_int, _lp, _dp, _dp, C.c_int]
    f("add_landmark").argtypes = [C.c_void_p, C.c_uint64, _dp]
    f("add_observation").argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    f("remove_observation").argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    f("optimize").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SummaryC)]
    f("set_time_limit").argtypes = [C.c_void_p, C.c_double, C.c_int]
    f("apply_marginalization").argtypes = [C.c_void_p, C.c_int, C.c_int]
    f("get_T_WS").argtypes = [C.c_void_p, C.c_uint64, _dp]
    f("get_speed_and_bias").argtypes = [C.c_void_p, C.c_uint64, _dp]
    f("get_extrinsics").argtypes = [C.c_void_p, C.c_uint64, C.c_int, _dp]
    f("get_landmark").argtypes = [C.c_void_p, C.c_uint64, _dp, _dp, C.POINTER(C.c_int)]
    f("num_frames").argtypes = [C.c_void_p]
    f("set_use_graph").argtypes = [C.c_void_p, C.c_int]
    f("last_timings").argtypes = [C.c_void_p, _dp]
    f("last_marg_info").argtypes = [C.c_void_p, _dp]
    f("apply_marginalization2").argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.c_int]
    f("prior_info").argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    f("frame_id_by_age").argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    f("is_keyframe").argtypes = [C.c_void_p, C.c_uint64]
    f("is_in_imu_window").argtypes = [C.c_void_p, C.c_uint64]
    f("num_landmarks").argtypes = [C.c_void_p]
    f("init_pose_from_imu").argtypes = [C.c_int, _dp, _dp]
    f("propagation").argtypes = [C.c_int, _lp, _dp, _dp, _dp, _dp, _dp, C.c_int64, C.c_int64]


class Api:
    """`api.okvis_est_xyz` resolves to `<prefix>xyz` of the library (one code path for both builds)."""

    def __init__(self, L, prefix="okvis_est_"):
        self._L, self._prefix = L, prefix
        declare(L, prefix)

    def __getattr__(self, name):
        return getattr(self._L, self._prefix + name[len("okvis_est_"):])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `python -m okvis_amd.build`")
        _lib = Api(C.CDLL(LIB_PATH))
    return _lib


class EstimatorError(RuntimeError):
    pass


def _chk(r, api=None):  # noqa: D401
    if r < 0:
        raise EstimatorError((api or lib()).okvis_est_last_error().decode())
    return r


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def imu_param_vector(p) -> np.ndarray:
    """okvis_amd.window.ImuParams -> the 13-vector of the C wrapper."""
    return np.array([p.a_max, p.g_max, p.sigma_g_c, p.sigma_a_c, p.sigma_bg, p.sigma_ba, p.sigma_gw_c, p.sigma_aw_c,
                     3600.0, p.g, 0.0, 0.0, 0.0])


def propagation(t, gyr, acc, prm13, T_WS, sb, t_start, t_end):
    t = np.ascontiguousarray(t, np.int64); gyr = _d(gyr); acc = _d(acc); prm13 = _d(prm13)
    T = _d(T_WS).copy(); s = _d(sb).copy()
    n = lib().okvis_est_propagation(int(t.size), t.ctypes.data_as(_lp), gyr.ctypes.data_as(_dp), acc.ctypes.data_as(_dp),
                                    prm13.ctypes.data_as(_dp), T.ctypes.data_as(_dp), s.ctypes.data_as(_dp),
                                    C.c_int64(int(t_start)), C.c_int64(int(t_end)))
    return T, s, n


def init_pose_from_imu(acc):
    acc = _d(acc).reshape(-1, 3)
    out = np.zeros(7)
    ok = lib().okvis_est_init_pose_from_imu(acc.shape[0], acc.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    return out, bool(ok)


class Frame:
    def __init__(self, frame_id, t_ns, T_SC, intr, models, api=None):
        self._api = api or lib()
        T_SC = _d(T_SC).reshape(-1, 7); intr = _d(intr).reshape(-1, 12)
        m = (C.c_int * len(models))(*[int(x) for x in models])
        self._h = self._api.okvis_est_frame_create(C.c_uint64(frame_id), C.c_int64(int(t_ns)), T_SC.shape[0],
                                               T_SC.ctypes.data_as(_dp), intr.ctypes.data_as(_dp), m)
        self.id = frame_id

    def add_keypoint(self, cam, x, y, size):
        return self._api.okvis_est_frame_add_keypoint(self._h, cam, C.c_float(x), C.c_float(y), C.c_float(size))

    def __del__(self):
        if getattr(self, "_h", None):
            self._api.okvis_est_frame_destroy(self._h)
            self._h = None


class Estimator:
    def __init__(self, device=0, api=None):
        self._api = api or lib()
        self._h = self._api.okvis_est_create(device)
        if not self._h:
            raise EstimatorError(self._api.okvis_est_last_error().decode())

    def _c(self, r):
        return _chk(r, self._api)

    def close(self):
        if getattr(self, "_h", None):
            self._api.okvis_est_destroy(self._h)
            self._h = None

    __del__ = close

    def addCamera(self, sig_abs_t, sig_abs_r, sig_rel_t, sig_rel_r):
        return self._c(self._api.okvis_est_add_camera(self._h, _d([sig_abs_t, sig_abs_r, sig_rel_t, sig_rel_r]).ctypes.data_as(_dp)))

    def addImu(self, prm13):
        return self._c(self._api.okvis_est_add_imu(self._h, _d(prm13).ctypes.data_as(_dp)))

    def addStates(self, frame: Frame, t, gyr, acc, asKeyframe):
        t = np.ascontiguousarray(t, np.int64); gyr = _d(gyr); acc = _d(acc)
        return bool(self._c(self._api.okvis_est_add_states(self._h, frame._h, int(t.size), t.ctypes.data_as(_lp),
                                                   gyr.ctypes.data_as(_dp), acc.ctypes.data_as(_dp), int(asKeyframe))))

    def last_error(self) -> str:
        """Text of the last refusal / exception of this thread's calls (okvis_est_last_error)."""
        return self._api.okvis_est_last_error().decode()

    def addLandmark(self, lm_id, hp):
        return bool(self._c(self._api.okvis_est_add_landmark(self._h, C.c_uint64(lm_id), _d(hp).ctypes.data_as(_dp))))

    def addObservation(self, lm_id, pose_id, cam, kp):
        h = C.c_uint64()
        r = self._c(self._api.okvis_est_add_observation(self._h, C.c_uint64(lm_id), C.c_uint64(pose_id), cam, kp, C.byref(h)))
        return h.value if r else 0

    def removeObservation(self, lm_id, pose_id, cam, kp):
        return bool(self._c(self._api.okvis_est_remove_observation(self._h, C.c_uint64(lm_id), C.c_uint64(pose_id), cam, kp)))

    def optimize(self, numIter, numThreads=1, verbose=False):
        s = SummaryC()
        self._c(self._api.okvis_est_optimize(self._h, numIter, numThreads, int(verbose), C.byref(s)))
        return s.as_dict()

    def setOptimizationTimeLimit(self, limit, min_iter):
        return bool(self._c(self._api.okvis_est_set_time_limit(self._h, float(limit), int(min_iter))))

    def applyMarginalizationStrategy(self, numKeyframes, numImuFrames, removed=None):
        """removed: optional list that receives the ids of the removed landmarks (okvis::MapPointVector&)."""
        n = C.c_int()
        cap = 1 << 16
        ids = (C.c_uint64 * cap)()
        ok = bool(self._c(self._api.okvis_est_apply_marginalization2(self._h, numKeyframes, numImuFrames, C.byref(n), ids, cap)))
        if removed is not None:
            removed.extend(int(ids[i]) for i in range(min(n.value, cap)))
        return ok

    def debugFailPendingMarginalization(self):
        """test hook: the marginalisation that is on its way fails where its numbers are waited for (late failure path)"""
        fn = self._api.okvis_est_debug_fail_pending_marginalization
        fn.argtypes = [C.c_void_p]
        self._c(fn(self._h))

    def debugFailNextMarginalization(self):
        """test hook: the next applyMarginalizationStrategy raises where its GPU call would be (roll-back test)"""
        fn = self._api.okvis_est_debug_fail_next_marginalization
        fn.argtypes = [C.c_void_p]
        self._c(fn(self._h))

    def lastOptimizeTimings(self):
        """ms: flatten, upload (host index build + H2D), iterations, downloads."""
        out = np.zeros(4)
        self._c(self._api.okvis_est_last_timings(self._h, out.ctypes.data_as(_dp)))
        return out

    def lastMarginalizationInfo(self):
        """ms flatten, upload, okvis_ba_marginalize; Jacobi sweeps (2, 0 = Cholesky fast path); sub-window D."""
        out = np.zeros(6)
        self._c(self._api.okvis_est_last_marg_info(self._h, out.ctypes.data_as(_dp)))
        return out

    def set_T_WS(self, pose_id, T):
        fn = self._api.okvis_est_set_T_WS
        fn.argtypes = [C.c_void_p, C.c_uint64, _dp]
        return bool(self._c(fn(self._h, int(pose_id), _d(T).ctypes.data_as(_dp))))

    def setSpeedAndBias(self, pose_id, sb):
        fn = self._api.okvis_est_set_speed_and_bias
        fn.argtypes = [C.c_void_p, C.c_uint64, _dp]
        return bool(self._c(fn(self._h, int(pose_id), _d(sb).ctypes.data_as(_dp))))

    def setCameraSensorStates(self, pose_id, cam, T):
        fn = self._api.okvis_est_set_extrinsics
        fn.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _dp]
        return bool(self._c(fn(self._h, int(pose_id), int(cam), _d(T).ctypes.data_as(_dp))))

    def setLandmark(self, lm_id, hp):
        fn = self._api.okvis_est_set_landmark
        fn.argtypes = [C.c_void_p, C.c_uint64, _dp]
        return bool(self._c(fn(self._h, int(lm_id), _d(hp).ctypes.data_as(_dp))))

    def setUsePatch(self, on):
        """True (default): optimize() patches the window the solver holds with the edits since the last call; False: flatten +
        upload every time."""
        fn = self._api.okvis_est_set_use_patch
        fn.argtypes = [C.c_void_p, C.c_int]
        self._c(fn(self._h, int(bool(on))))

    def lastOptimizeWasPatch(self):
        fn = self._api.okvis_est_last_was_patch
        fn.argtypes = [C.c_void_p]
        return bool(self._c(fn(self._h)))

    def debugCheckWindow(self):
        """'' when the window the solver holds equals a freshly flattened one, else what differs"""
        fn = self._api.okvis_est_debug_check_window
        fn.argtypes = [C.c_void_p]
        return "" if self._c(fn(self._h)) else self._api.okvis_est_last_error().decode()

    def setUseGraph(self, use_graph):
        self._c(self._api.okvis_est_set_use_graph(self._h, int(use_graph)))

    def priorInfo(self):
        d, nb = C.c_int(), C.c_int()
        self._c(self._api.okvis_est_prior_info(self._h, C.byref(d), C.byref(nb)))
        return d.value, nb.value

    def frameIdByAge(self, age):
        i = C.c_uint64()
        self._c(self._api.okvis_est_frame_id_by_age(self._h, int(age), C.byref(i)))
        return i.value

    def isKeyframe(self, frame_id):
        return bool(self._c(self._api.okvis_est_is_keyframe(self._h, C.c_uint64(frame_id))))

    def isInImuWindow(self, frame_id):
        return bool(self._c(self._api.okvis_est_is_in_imu_window(self._h, C.c_uint64(frame_id))))

    def get_T_WS(self, pose_id):
        out = np.zeros(7)
        return out if self._c(self._api.okvis_est_get_T_WS(self._h, C.c_uint64(pose_id), out.ctypes.data_as(_dp))) else None

    def getSpeedAndBias(self, pose_id):
        out = np.zeros(9)
        return out if self._c(self._api.okvis_est_get_speed_and_bias(self._h, C.c_uint64(pose_id), out.ctypes.data_as(_dp))) else None

    def getCameraSensorStates(self, pose_id, cam):
        out = np.zeros(7)
        return out if self._c(self._api.okvis_est_get_extrinsics(self._h, C.c_uint64(pose_id), cam, out.ctypes.data_as(_dp))) else None

    def getLandmark(self, lm_id):
        p = np.zeros(4); q = C.c_double(); n = C.c_int()
        self._c(self._api.okvis_est_get_landmark(self._h, C.c_uint64(lm_id), p.ctypes.data_as(_dp), C.byref(q), C.byref(n)))
        return p, q.value, n.value

    def numFrames(self):
        return self._api.okvis_est_num_frames(self._h)

    def numLandmarks(self):
        return self._api.okvis_est_num_landmarks(self._h)
