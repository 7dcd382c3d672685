"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from okvis_amd.window import (ImuParamsC, MargResultC, MargSpecC, OptionsC, SummaryC, Window, WindowC,
                              default_options, marg_call)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int32)

# array ids (include/okvis_amd_ba.h enum okvis_ba_array)
ARR = dict(POSE=0, SB=1, LM=2, OBS_RESIDUAL=3, LM_V=4, LM_B=5, LM_HQ=6, PAIR_W=7, REDUCED_S=8,
           REDUCED_RHS=9, STEP=10, LM_QUALITY=11, GRADIENT=12, IMU_RESIDUAL=13, HPP=14, IMU_SB_REF=16)


def locked_make(args, lock_dir):
    """`make` under an exclusive lock file in lock_dir: parallel test workers (pytest -n) that find the same target stale must not
    build it into each other's half-written objects."""
    import fcntl
    with open(os.path.join(lock_dir, ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s"] + args, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_oracle():
    locked_make(["-C", _ORACLE_DIR], _ORACLE_DIR)
    return os.path.join(_ORACLE_DIR, "liboracle.so")


_lib = None
_lib_ld = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path) or os.path.exists(os.path.join(_ORACLE_DIR, "Makefile")) and _stale(path):
        path = build_oracle()
    _lib = _bind(C.CDLL(path))
    return _lib


def lib_ld():
    """The same sources built with `real = long double` (x87 extended precision, 64-bit mantissa): the referee of
    tests/test_oracle_referee.py.  Same entry points, double in and out."""
    global _lib_ld
    if _lib_ld is not None:
        return _lib_ld
    path = os.path.join(_ORACLE_DIR, "liboracle_ld.so")
    if not os.path.exists(path) or os.path.exists(os.path.join(_ORACLE_DIR, "Makefile")) and _stale(path):
        locked_make(["-C", _ORACLE_DIR, "liboracle_ld.so"], _ORACLE_DIR)
    _lib_ld = _bind(C.CDLL(path))
    return _lib_ld


def _bind(L):
    L.orc_window_create.restype = C.c_void_p
    L.orc_window_create.argtypes = [C.POINTER(WindowC)]
    L.orc_window_destroy.argtypes = [C.c_void_p]
    L.orc_window_set_marg_exact.argtypes = [C.c_void_p, C.c_int]
    L.orc_window_reduced_dim.argtypes = [C.c_void_p]
    L.orc_window_pair_count.argtypes = [C.c_void_p]
    L.orc_window_pairs.argtypes = [C.c_void_p, _ip, _ip]
    L.orc_window_linearize.restype = C.c_double
    L.orc_window_linearize.argtypes = [C.c_void_p]
    L.orc_window_cost.restype = C.c_double
    L.orc_window_cost.argtypes = [C.c_void_p]
    L.orc_window_solve.argtypes = [C.c_void_p, C.c_double, C.POINTER(OptionsC)]
    L.orc_window_optimize.argtypes = [C.c_void_p, C.POINTER(OptionsC), C.c_int, C.POINTER(SummaryC)]
    L.orc_window_time_iterations.restype = C.c_double
    L.orc_window_time_iterations.argtypes = [C.c_void_p, C.POINTER(OptionsC), C.c_int]
    L.orc_window_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.orc_window_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.orc_window_array_size.restype = C.c_int64
    L.orc_window_array_size.argtypes = [C.c_void_p, C.c_int]
    L.orc_window_download.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
    L.orc_window_full_gradient.argtypes = [C.c_void_p, _dp]
    L.orc_window_marginalize.argtypes = [C.c_void_p, C.POINTER(MargSpecC), C.POINTER(MargResultC)]
    L.orc_sym_eig.argtypes = [_dp, C.c_int, _dp, _dp]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_imu_propagation.argtypes = [C.c_int, _lp, _dp, _dp, C.POINTER(ImuParamsC), _dp, _dp, C.c_int64,
                                      C.c_int64, _dp, _dp]
    return L


def _stale(so):
    t = os.path.getmtime(so)
    for f in os.listdir(_ORACLE_DIR):
        if f.endswith((".cpp", ".hpp", ".h")) and os.path.getmtime(os.path.join(_ORACLE_DIR, f)) > t:
            return True
    return os.path.getmtime(os.path.join(_ROOT, "include", "okvis_amd_ba.h")) > t


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _arr(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a if shape is None else a.reshape(shape)


# ---------------------------------------------------------------------------------------------------
# factor-level wrappers
# ---------------------------------------------------------------------------------------------------
def pose_plus(x, d):
    out = np.zeros(7)
    lib().orc_pose_plus(_p(_arr(x)), _p(_arr(d)), _p(out))
    return out


def pose_minus(x, xp):
    out = np.zeros(6)
    lib().orc_pose_minus(_p(_arr(x)), _p(_arr(xp)), _p(out))
    return out


def pose_lift_jacobian(x):
    out = np.zeros((6, 7))
    lib().orc_pose_lift_jacobian(_p(_arr(x)), _p(out))
    return out


def pose_plus_jacobian(x):
    out = np.zeros((7, 6))
    lib().orc_pose_plus_jacobian(_p(_arr(x)), _p(out))
    return out


def reprojection(pose, point, extr, intr, model, uv, sqrt_info=None, jac=True):
    si = _arr(np.eye(2) if sqrt_info is None else sqrt_info).reshape(4)
    r = np.zeros(2)
    Jp, Jl, Je = (np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 6))) if jac else (None, None, None)
    st = lib().orc_reprojection(_p(_arr(pose)), _p(_arr(point)), _p(_arr(extr)), _p(_arr(intr)), int(model),
                                _p(_arr(uv)), _p(si), _p(r), _p(Jp), _p(Jl), _p(Je))
    return r, Jp, Jl, Je, bool(st & 1), bool(st & 2)


def project(intr, model, point, jac=True):
    kp = np.zeros(2)
    J = np.zeros((2, 3)) if jac else None
    ok = lib().orc_project(_p(_arr(intr)), int(model), _p(_arr(point)), _p(kp), _p(J))
    return kp, J, bool(ok)


def _imu_args(t, gyr, acc, prm):
    t = np.ascontiguousarray(t, np.int64)
    return t, _arr(gyr), _arr(acc), prm.as_c()


def imu_evaluate_fresh(t, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    r = np.zeros(15)
    J0, J1, J2, J3 = np.zeros((15, 6)), np.zeros((15, 9)), np.zeros((15, 6)), np.zeros((15, 9))
    si = np.zeros((15, 15))
    n = lib().orc_imu_evaluate_fresh(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc),
                                     C.c_int64(int(t0)), C.c_int64(int(t1)), _p(_arr(pose0)), _p(_arr(sb0)),
                                     _p(_arr(pose1)), _p(_arr(sb1)), _p(r), _p(J0), _p(J1), _p(J2), _p(J3), _p(si))
    return r, (J0, J1, J2, J3), si, n


def imu_evaluate_at_ref(t, gyr, acc, prm, t0, t1, sb_ref, pose0, sb0, pose1, sb1, jac=True):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    r = np.zeros(15)
    Js = (np.zeros((15, 6)), np.zeros((15, 9)), np.zeros((15, 6)), np.zeros((15, 9))) if jac else (None,) * 4
    n = lib().orc_imu_evaluate_at_ref(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc),
                                      C.c_int64(int(t0)), C.c_int64(int(t1)), _p(_arr(sb_ref)), _p(_arr(pose0)),
                                      _p(_arr(sb0)), _p(_arr(pose1)), _p(_arr(sb1)), _p(r), *[_p(j) for j in Js])
    return r, Js, n


def imu_propagation(t, gyr, acc, prm, T_WS, sb, t_start, t_end, want_cov=False, want_jac=False):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    T = _arr(T_WS).copy()
    s = _arr(sb).copy()
    cov = np.zeros((15, 15)) if want_cov else None
    jac = np.zeros((15, 15)) if want_jac else None
    n = lib().orc_imu_propagation(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc), _p(T), _p(s),
                                  C.c_int64(int(t_start)), C.c_int64(int(t_end)), _p(cov), _p(jac))
    return T, s, cov, jac, n


def pose_error(pose, meas, sqrt_info):
    r, J = np.zeros(6), np.zeros((6, 6))
    lib().orc_pose_error(_p(_arr(pose)), _p(_arr(meas)), _p(_arr(sqrt_info).reshape(36)), _p(r), _p(J))
    return r, J


def speedbias_error(sb, meas, sqrt_info):
    r, J = np.zeros(9), np.zeros((9, 9))
    lib().orc_speedbias_error(_p(_arr(sb)), _p(_arr(meas)), _p(_arr(sqrt_info).reshape(81)), _p(r), _p(J))
    return r, J


def relative_pose_error(p0, p1, sqrt_info):
    r, J0, J1 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    lib().orc_relative_pose_error(_p(_arr(p0)), _p(_arr(p1)), _p(_arr(sqrt_info).reshape(36)), _p(r), _p(J0), _p(J1))
    return r, J0, J1


def sqrt_information(info):
    info = _arr(info)
    n = info.shape[0]
    out = np.zeros((n, n))
    lib().orc_sqrt_information(_p(info), n, _p(out))
    return out


# ---------------------------------------------------------------------------------------------------
# window-level wrapper
# ---------------------------------------------------------------------------------------------------
class OracleWindow:
    def __init__(self, window: Window, extended=False):
        self.window = window
        self._L = lib_ld() if extended else lib()
        self._wc, self._keep = window.as_c()
        self._h = self._L.orc_window_create(C.byref(self._wc))
        self.D = self._L.orc_window_reduced_dim(self._h)
        self.n_pair = self._L.orc_window_pair_count(self._h)

    @classmethod
    def from_c(cls, wc_ptr, extended=False):
        """from a POINTER(WindowC) somebody else owns (the library copies what it needs)"""
        import types
        self = cls.__new__(cls)
        wc = wc_ptr.contents
        self.window = types.SimpleNamespace(n_pose=wc.n_pose, n_sb=wc.n_sb, n_lm=wc.n_lm)
        self._L = lib_ld() if extended else lib()
        self._h = self._L.orc_window_create(wc_ptr)
        self.D = self._L.orc_window_reduced_dim(self._h)
        self.n_pair = self._L.orc_window_pair_count(self._h)
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_window_destroy(self._h)
            self._h = None

    def set_marg_exact(self, exact):
        self._L.orc_window_set_marg_exact(self._h, int(exact))

    def pairs(self):
        a = np.zeros(self.n_pair, np.int32)
        b = np.zeros(self.n_pair, np.int32)
        self._L.orc_window_pairs(self._h, a.ctypes.data_as(_ip), b.ctypes.data_as(_ip))
        return a, b

    def linearize(self):
        return self._L.orc_window_linearize(self._h)

    def cost(self):
        return self._L.orc_window_cost(self._h)

    def solve(self, radius, opt=None):
        opt = opt or default_options()
        return self._L.orc_window_solve(self._h, float(radius), C.byref(opt))

    def optimize(self, num_iter, opt=None):
        opt = opt or default_options()
        s = SummaryC()
        self._L.orc_window_optimize(self._h, C.byref(opt), int(num_iter), C.byref(s))
        return s.as_dict()

    def time_iterations(self, n, opt=None):
        opt = opt or default_options()
        return self._L.orc_window_time_iterations(self._h, C.byref(opt), int(n))

    def get_state(self):
        w = self.window
        pose, sb, lm = np.zeros((w.n_pose, 7)), np.zeros((w.n_sb, 9)), np.zeros((w.n_lm, 4))
        self._L.orc_window_get_state(self._h, _p(pose), _p(sb), _p(lm))
        return pose, sb, lm

    def set_state(self, pose=None, sb=None, lm=None):
        self._L.orc_window_set_state(self._h, _p(None if pose is None else _arr(pose)),
                                   _p(None if sb is None else _arr(sb)), _p(None if lm is None else _arr(lm)))

    def array(self, name):
        which = ARR[name]
        n = self._L.orc_window_array_size(self._h, which)
        out = np.zeros(max(n, 0))
        if n > 0:
            assert self._L.orc_window_download(self._h, which, _p(out), n) == 0
        return out

    def marginalize(self, pose_marg, sb_marg, prior=None):
        st, out = marg_call(lambda sp, rs: self._L.orc_window_marginalize(self._h, sp, rs), self.window.n_pose,
                            self.window.n_sb, pose_marg, sb_marg, prior)
        assert st == 0, st
        return out

    def full_gradient(self):
        g = np.zeros(self.D + 3 * self.window.n_lm)
        self._L.orc_window_full_gradient(self._h, _p(g))
        return g
