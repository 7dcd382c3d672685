"""ctypes binding of oracle/_ref/libokvis_ref.so — TEST INFRASTRUCTURE.

The library holds the okvis reference's OWN classes (ReprojectionError, ImuError, PoseError, SpeedAndBiasError,
RelativePoseError, MarginalizationError, Map, PinholeCamera<D>, the local parameterisations) compiled unmodified from
/root/reference by oracle/ref/Makefile against the stand-in headers of oracle/shim (Eigen / Ceres / glog / OpenCV are
not installed).  It is the pin for the restatement in oracle/ and for tests/golden/*.npz; only tests/ and
tests/golden/make_golden.py import this module.  The product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from okvis_amd.window import ImuParamsC, MargResultC, MargSpecC, OptionsC, SummaryC, Window, WindowC, default_options, marg_call

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REF_BUILD = os.path.join(_ROOT, "oracle", "ref")
_REF_SO = os.path.join(_ROOT, "oracle", "_ref", "libokvis_ref.so")
REFERENCE = os.environ.get("OKVIS_REFERENCE", "/root/reference")
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int32)


def available() -> bool:
    """True when the library exists or can be built (the reference tree is present)."""
    return os.path.exists(_REF_SO) or os.path.isdir(os.path.join(REFERENCE, "okvis_ceres"))


def build() -> str:
    """make -C oracle/ref (needs the reference tree; on the GPU box the prebuilt .so travels with the snapshot)."""
    if os.path.isdir(os.path.join(REFERENCE, "okvis_ceres")):
        try:
            from tests.oracle_lib import locked_make
        except ImportError:   # (imported with tests/ itself on the path, e.g. by tests/golden/make_golden.py)
            from oracle_lib import locked_make
        locked_make(["-C", _REF_BUILD, f"REF={REFERENCE}"], _REF_BUILD)
    return _REF_SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = build()
    L = C.CDLL(path)
    assert L.ref_abi_version() == 1
    L.ref_window_create.restype = C.c_void_p
    L.ref_window_create.argtypes = [C.POINTER(WindowC)]
    L.ref_window_destroy.argtypes = [C.c_void_p]
    L.ref_window_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.ref_window_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.ref_window_cost.restype = C.c_double
    L.ref_window_cost.argtypes = [C.c_void_p]
    L.ref_window_residual.argtypes = [C.c_void_p, C.c_int, _dp]
    L.ref_window_num_residual_blocks.argtypes = [C.c_void_p]
    L.ref_window_full_system.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _ip, _ip, _ip, _ip]
    L.ref_window_marginalize.argtypes = [C.c_void_p, C.POINTER(MargSpecC), C.POINTER(MargResultC)]
    L.ref_window_lm_quality.argtypes = [C.c_void_p, _dp, _dp]
    L.ref_window_imu_sb_ref.argtypes = [C.c_void_p, _dp]
    L.ref_window_optimize.argtypes = [C.c_void_p, C.POINTER(OptionsC), C.c_int, C.c_int, C.POINTER(SummaryC)]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _arr(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a if shape is None else a.reshape(shape)


# ---- factor level (same call shapes as tests/oracle_lib.py) ------------------------------------------------------
def pose_plus(x, d):
    out = np.zeros(7)
    lib().ref_pose_plus(_p(_arr(x)), _p(_arr(d)), _p(out))
    return out


def pose_minus(x, xp):
    out = np.zeros(6)
    lib().ref_pose_minus(_p(_arr(x)), _p(_arr(xp)), _p(out))
    return out


def pose_lift_jacobian(x):
    out = np.zeros((6, 7))
    lib().ref_pose_lift_jacobian(_p(_arr(x)), _p(out))
    return out


def pose_plus_jacobian(x):
    out = np.zeros((7, 6))
    lib().ref_pose_plus_jacobian(_p(_arr(x)), _p(out))
    return out


def project(intr, model, point, jac=True):
    kp = np.zeros(2)
    J = np.zeros((2, 3)) if jac else None
    st = lib().ref_project(_p(_arr(intr)), int(model), _p(_arr(point)), _p(kp), _p(J))
    return kp, J, st


def reprojection(pose, point, extr, intr, model, uv, sqrt_info=None, jac=True):
    si = _arr(np.eye(2) if sqrt_info is None else sqrt_info).reshape(4)
    r = np.zeros(2)
    Jp, Jl, Je = (np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 6))) if jac else (None, None, None)
    lib().ref_reprojection(_p(_arr(pose)), _p(_arr(point)), _p(_arr(extr)), _p(_arr(intr)), int(model), _p(_arr(uv)),
                           _p(si), _p(r), _p(Jp), _p(Jl), _p(Je))
    return r, Jp, Jl, Je


def sqrt_information(info):
    info = _arr(info)
    n = info.shape[0]
    out = np.zeros((n, n))
    assert lib().ref_sqrt_information(_p(info), n, _p(out)) == 0
    return out


def pose_error(pose, meas, sqrt_info):
    r, J = np.zeros(6), np.zeros((6, 6))
    lib().ref_pose_error(_p(_arr(pose)), _p(_arr(meas)), _p(_arr(sqrt_info).reshape(36)), _p(r), _p(J))
    return r, J


def speedbias_error(sb, meas, sqrt_info):
    r, J = np.zeros(9), np.zeros((9, 9))
    lib().ref_speedbias_error(_p(_arr(sb)), _p(_arr(meas)), _p(_arr(sqrt_info).reshape(81)), _p(r), _p(J))
    return r, J


def relative_pose_error(p0, p1, sqrt_info):
    r, J0, J1 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    lib().ref_relative_pose_error(_p(_arr(p0)), _p(_arr(p1)), _p(_arr(sqrt_info).reshape(36)), _p(r), _p(J0), _p(J1))
    return r, J0, J1


def _imu_args(t, gyr, acc, prm):
    t = np.ascontiguousarray(t, np.int64)
    return t, _arr(gyr), _arr(acc), prm.as_c()


def imu_evaluate_fresh(t, gyr, acc, prm, t0, t1, pose0, sb0, pose1, sb1):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    r = np.zeros(15)
    J0, J1, J2, J3 = np.zeros((15, 6)), np.zeros((15, 9)), np.zeros((15, 6)), np.zeros((15, 9))
    si = np.zeros((15, 15))
    n = lib().ref_imu_evaluate_fresh(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc),
                                     C.c_int64(int(t0)), C.c_int64(int(t1)), _p(_arr(pose0)), _p(_arr(sb0)),
                                     _p(_arr(pose1)), _p(_arr(sb1)), _p(r), _p(J0), _p(J1), _p(J2), _p(J3), _p(si))
    return r, (J0, J1, J2, J3), si, n


def imu_evaluate_at_ref(t, gyr, acc, prm, t0, t1, sb_ref, pose0, sb0, pose1, sb1):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    r = np.zeros(15)
    Js = (np.zeros((15, 6)), np.zeros((15, 9)), np.zeros((15, 6)), np.zeros((15, 9)))
    n = lib().ref_imu_evaluate_at_ref(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc),
                                      C.c_int64(int(t0)), C.c_int64(int(t1)), _p(_arr(sb_ref)), _p(_arr(pose0)),
                                      _p(_arr(sb0)), _p(_arr(pose1)), _p(_arr(sb1)), _p(r), *[_p(j) for j in Js])
    return r, Js, n


def imu_propagation(t, gyr, acc, prm, T_WS, sb, t_start, t_end, want_cov=False, want_jac=False):
    t, gyr, acc, pc = _imu_args(t, gyr, acc, prm)
    T = _arr(T_WS).copy()
    s = _arr(sb).copy()
    cov = np.zeros((15, 15)) if want_cov else None
    jac = np.zeros((15, 15)) if want_jac else None
    lib().ref_imu_propagation.argtypes = [C.c_int, _lp, _dp, _dp, C.POINTER(ImuParamsC), _dp, _dp, C.c_int64, C.c_int64,
                                          _dp, _dp]
    n = lib().ref_imu_propagation(int(t.size), t.ctypes.data_as(_lp), _p(gyr), _p(acc), C.byref(pc), _p(T), _p(s),
                                  C.c_int64(int(t_start)), C.c_int64(int(t_end)), _p(cov), _p(jac))
    return T, s, cov, jac, n


# ---- window level ------------------------------------------------------------------------------------------------
class RefWindow:
    """The reference's okvis::ceres::Map filled with the reference's own parameter / residual blocks."""

    def __init__(self, window: Window):
        self.window = window
        self._wc, self._keep = window.as_c()
        self._h = lib().ref_window_create(C.byref(self._wc))

    @classmethod
    def from_c(cls, wc_ptr):
        """from a POINTER(WindowC) somebody else owns (the library copies what it needs)"""
        import types
        self = cls.__new__(cls)
        wc = wc_ptr.contents
        self.window = types.SimpleNamespace(n_pose=wc.n_pose, n_sb=wc.n_sb, n_lm=wc.n_lm)
        self._h = lib().ref_window_create(wc_ptr)
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_window_destroy(self._h)
            self._h = None

    def cost(self):
        return lib().ref_window_cost(self._h)

    def get_state(self):
        w = self.window
        pose, sb, lm = np.zeros((w.n_pose, 7)), np.zeros((w.n_sb, 9)), np.zeros((w.n_lm, 4))
        lib().ref_window_get_state(self._h, _p(pose), _p(sb), _p(lm))
        return pose, sb, lm

    def set_state(self, pose=None, sb=None, lm=None):
        lib().ref_window_set_state(self._h, _p(None if pose is None else _arr(pose)),
                                   _p(None if sb is None else _arr(sb)), _p(None if lm is None else _arr(lm)))

    def residuals(self):
        """list of the un-robustified weighted residual vectors in insertion order
        (pose priors, speed/bias priors, relative pose, IMU, [marginalisation prior], reprojection)"""
        out = []
        buf = np.zeros(4096)
        for k in range(lib().ref_window_num_residual_blocks(self._h)):
            n = lib().ref_window_residual(self._h, k, _p(buf))
            out.append(buf[:n].copy())
        return out

    def full_system(self):
        """H, b0 of MarginalizationError::addResidualBlock over every residual, re-ordered to the repository's order:
        free pose blocks (by index), free speed/bias blocks, landmarks."""
        w = self.window
        cap = 6 * w.n_pose + 9 * w.n_sb + 3 * w.n_lm
        capb = w.n_pose + w.n_sb + w.n_lm
        dim, nb = C.c_int32(0), C.c_int32(0)
        H, b0 = np.zeros(cap * cap), np.zeros(cap)
        typ, idx, off = (np.zeros(capb, np.int32) for _ in range(3))
        st = lib().ref_window_full_system(self._h, cap, capb, C.byref(dim), _p(H), _p(b0), C.byref(nb),
                                          typ.ctypes.data_as(_ip), idx.ctypes.data_as(_ip), off.ctypes.data_as(_ip))
        assert st == 0
        n, nb = dim.value, nb.value
        H = H[:n * n].reshape(n, n)
        b0 = b0[:n]
        size = {0: 6, 1: 9, 2: 3}
        order = sorted(range(nb), key=lambda k: (typ[k], idx[k]))
        perm = np.concatenate([np.arange(off[k], off[k] + size[int(typ[k])]) for k in order]) if nb else np.zeros(0, int)
        blocks = [(int(typ[k]), int(idx[k])) for k in order]
        return H[np.ix_(perm, perm)], b0[perm], blocks

    def marginalize(self, pose_marg, sb_marg, prior=None):
        """consumes the window (like the reference: marginalised residuals and blocks leave the Map)"""
        st, out = marg_call(lambda sp, rs: lib().ref_window_marginalize(self._h, sp, rs), self.window.n_pose,
                            self.window.n_sb, pose_marg, sb_marg, prior)
        assert st == 0, st
        return out

    def lm_quality(self):
        q = np.zeros(self.window.n_lm)
        H = np.zeros((self.window.n_lm, 3, 3))
        lib().ref_window_lm_quality(self._h, _p(q), _p(H))
        return q, H

    def imu_sb_ref(self):
        out = np.zeros((self.window.n_imu, 9))
        lib().ref_window_imu_sb_ref(self._h, _p(out))
        return out

    def optimize(self, num_iter, opt=None, dogleg=True):
        opt = opt or default_options()
        s = SummaryC()
        lib().ref_window_optimize(self._h, C.byref(opt), int(num_iter), int(bool(dogleg)), C.byref(s))
        return s.as_dict()


# ---- the reference's okvis::Estimator behind the same Python surface as okvis_amd.estimator.Estimator -------------
def estimator_api():
    """flat estimator API of the reference build (ref_est_*), usable as `api=` of okvis_amd.estimator.Estimator / Frame"""
    from okvis_amd import estimator as E
    global _est_api
    try:
        return _est_api
    except NameError:
        _est_api = E.Api(lib(), "ref_est_")
        return _est_api


def RefEstimator():
    from okvis_amd import estimator as E
    return E.Estimator(0, api=estimator_api())


def RefFrame(frame_id, t_ns, T_SC, intr, models):
    from okvis_amd import estimator as E
    return E.Frame(frame_id, t_ns, T_SC, intr, models, api=estimator_api())
