"""ctypes view of include/okvis_amd_frontend.h: the batched reprojection pieces of the OKVIS frontend (stereo triangulation
with uncertainty, 3D-2D projection and chi-square gating) on the MI355X.  No CPU path."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

TRI_VALID, TRI_NOT_PARALLEL, TRI_CAN_INIT, TRI_RANK_DEFICIENT = 1, 2, 4, 8
PROJ_SUCCESSFUL, PROJ_OUTSIDE_IMAGE, PROJ_MASKED, PROJ_BEHIND, PROJ_INVALID = range(5)
GATE_VERIFIED, GATE_ACCEPTED, GATE_UNCERTAIN = 1, 2, 4
SYMBOLS = ["okvis_fe_create", "okvis_fe_destroy", "okvis_fe_stereo_triangulate", "okvis_fe_stereo_triangulate_gn", "okvis_fe_project_landmarks",
           "okvis_fe_gate_3d2d"]


class CameraC(C.Structure):
    """okvis_fe_camera"""
    _fields_ = [("intr", C.c_double * 12), ("model", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("reserved", C.c_int32)]


def camera(intr, model, width=752, height=480) -> CameraC:
    c = CameraC()
    k = np.zeros(12)
    k[:len(intr)] = intr
    c.intr[:] = list(k)
    c.model, c.width, c.height = int(model), int(width), int(height)
    return c


def _f32(a, cols):
    return np.ascontiguousarray(a, np.float32).reshape(-1, cols)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, np.float64)
    return a if shape is None else a.reshape(shape)


def declare(L, prefix="okvis_fe_", with_context=True):
    """signatures of the three batch entries under `prefix` (the reference build exports them as ref_fe_* without a context)"""
    vp = C.c_void_p
    ctx = [vp] if with_context else []
    cam = C.POINTER(CameraC)
    getattr(L, prefix + "stereo_triangulate").argtypes = ctx + [cam, cam, vp, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp,
                                                                C.c_int32, vp, vp, vp]
    if with_context and hasattr(L, prefix + "stereo_triangulate_gn"):
        getattr(L, prefix + "stereo_triangulate_gn").argtypes = ctx + [cam, cam, vp, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp,
                                                                       C.c_int32, vp, vp, vp, vp]
    getattr(L, prefix + "project_landmarks").argtypes = ctx + [cam, vp, vp, C.c_int32, vp, vp, vp, vp]
    getattr(L, prefix + "gate_3d2d").argtypes = ctx + [C.c_int32, vp, vp, C.c_int32, vp, C.c_int32, vp, vp, vp]


class Frontend:
    """One okvis_fe_context.  `api=(library, prefix)` swaps in another implementation of the same three entries (the tests
    pass the reference build); the default is the HIP library."""

    def __init__(self, device: int = 0, api=None):
        if api is None:
            self._L, self._prefix, self._ctx = _lib.lib(), "okvis_fe_", C.c_void_p()
            for s in SYMBOLS:
                getattr(self._L, s)
            declare(self._L)
            self._L.okvis_fe_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
            self._L.okvis_fe_destroy.argtypes = [C.c_void_p]
            _lib.check(self._L.okvis_fe_create(C.byref(self._ctx), int(device)), "okvis_fe_create")
        else:
            self._L, self._prefix = api
            self._ctx = None
            declare(self._L, self._prefix, with_context=False)

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.okvis_fe_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _call(self, name, *args):
        fn = getattr(self._L, self._prefix + name)
        rc = fn(self._ctx, *args) if self._ctx is not None else fn(*args)
        if self._ctx is not None:
            _lib.check(rc, name)
        elif rc != 0:
            raise RuntimeError(f"{self._prefix}{name} returned {rc}")

    def stereo_triangulate(self, cam_a: CameraC, cam_b: CameraC, T_AB, UOplus, kp_a, kp_b, pairs, sigma_ray=None,
                           want_uncertainty=True):
        """-> hp_A [n][4], cov [n][3][3], flags [n] (TRI_* bits)"""
        kp_a, kp_b = _f32(kp_a, 3), _f32(kp_b, 3)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = len(pairs)
        T_AB, UOplus = _f64(T_AB, 7), _f64(UOplus, (6, 6))
        sig = None if sigma_ray is None else _f64(sigma_ray, n)
        hp, cov, flags = np.zeros((n, 4)), np.zeros((n, 3, 3)), np.zeros(n, np.uint8)
        self._call("stereo_triangulate", C.byref(cam_a), C.byref(cam_b), T_AB.ctypes.data, UOplus.ctypes.data, len(kp_a),
                   kp_a.ctypes.data, len(kp_b), kp_b.ctypes.data, n, pairs.ctypes.data, None if sig is None else sig.ctypes.data,
                   int(bool(want_uncertainty)), hp.ctypes.data, cov.ctypes.data, flags.ctypes.data)
        return hp, cov, flags

    def stereo_triangulate_gn(self, cam_a: CameraC, cam_b: CameraC, T_AB, UOplus, kp_a, kp_b, pairs, sigma_ray=None):
        """okvis_fe_stereo_triangulate_gn -> hp_A, cov, flags, gn [n][9][9] (the Gauss-Newton matrix getUncertainty inverts)"""
        kp_a, kp_b = _f32(kp_a, 3), _f32(kp_b, 3)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = len(pairs)
        T_AB, UOplus = _f64(T_AB, 7), _f64(UOplus, (6, 6))
        sig = None if sigma_ray is None else _f64(sigma_ray, n)
        hp, cov, flags, gn = np.zeros((n, 4)), np.zeros((n, 3, 3)), np.zeros(n, np.uint8), np.zeros((n, 9, 9))
        self._call("stereo_triangulate_gn", C.byref(cam_a), C.byref(cam_b), T_AB.ctypes.data, UOplus.ctypes.data, len(kp_a),
                   kp_a.ctypes.data, len(kp_b), kp_b.ctypes.data, n, pairs.ctypes.data, None if sig is None else sig.ctypes.data,
                   1, hp.ctypes.data, cov.ctypes.data, flags.ctypes.data, gn.ctypes.data)
        return hp, cov, flags, gn

    def project_landmarks(self, cam_b: CameraC, T_CbW, P3, hp_W):
        """-> uv [n][2], U [n][2][2], status [n] (PROJ_*)"""
        hp_W = _f64(hp_W).reshape(-1, 4)
        n = len(hp_W)
        T_CbW, P3 = _f64(T_CbW, 7), _f64(P3, (3, 3))
        uv, U, st = np.zeros((n, 2)), np.zeros((n, 2, 2)), np.zeros(n, np.uint8)
        self._call("project_landmarks", C.byref(cam_b), T_CbW.ctypes.data, P3.ctypes.data, n, hp_W.ctypes.data, uv.ctypes.data,
                   U.ctypes.data, st.ctypes.data)
        return uv, U, st

    def gate_3d2d(self, uv, U, kp_b, pairs):
        """-> chi2 [n], flags [n] (GATE_* bits)"""
        uv, U, kp_b = _f64(uv).reshape(-1, 2), _f64(U).reshape(-1, 2, 2), _f32(kp_b, 3)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = len(pairs)
        chi2, flags = np.zeros(n), np.zeros(n, np.uint8)
        self._call("gate_3d2d", len(uv), uv.ctypes.data, U.ctypes.data, len(kp_b), kp_b.ctypes.data, n, pairs.ctypes.data,
                   chi2.ctypes.data, flags.ctypes.data)
        return chi2, flags
