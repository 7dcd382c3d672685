"""Flat struct-of-arrays description of one sliding window (host side).

Mirror of ``okvis_ba_window`` in ``include/okvis_amd_ba.h``.  It replaces the pointer graph the reference
builds in ``okvis::ceres::Map`` (reference okvis_ceres/include/okvis/ceres/Map.hpp:348-402,
okvis_ceres/src/Map.cpp:292-565) and ``Estimator::statesMap_/landmarksMap_``
(okvis_ceres/include/okvis/Estimator.hpp:555-563) by index arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

DIST_NONE, DIST_RADTAN, DIST_EQUIDISTANT, DIST_RADTAN8 = 0, 1, 2, 3
BLOCK_POSE, BLOCK_SPEEDBIAS = 0, 1

IMU_CACHE_DOUBLES = 290   # OKVIS_BA_IMU_CACHE_DOUBLES
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)


class ImuParamsC(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("sigma_g_c", "sigma_a_c", "sigma_gw_c", "sigma_aw_c", "g", "g_max", "a_max")]


class WindowC(C.Structure):
    """ctypes image of ``okvis_ba_window`` (field order must match the header)."""
    _fields_ = [
        ("n_pose", C.c_int32), ("pose", _dp), ("pose_fixed", _bp),
        ("n_sb", C.c_int32), ("sb", _dp), ("sb_fixed", _bp),
        ("n_lm", C.c_int32), ("lm", _dp),
        ("n_cam", C.c_int32), ("cam_intr", _dp), ("cam_model", _ip),
        ("n_obs", C.c_int32), ("obs_lm", _ip), ("obs_pose", _ip), ("obs_ext", _ip), ("obs_cam", _ip),
        ("obs_uv", _dp), ("obs_sqrtw", _dp), ("cauchy_b", C.c_double),
        ("n_imu", C.c_int32), ("imu_pose0", _ip), ("imu_sb0", _ip), ("imu_pose1", _ip), ("imu_sb1", _ip),
        ("imu_t0", _lp), ("imu_t1", _lp), ("imu_s_begin", _ip), ("imu_s_count", _ip),
        ("n_imu_samples", C.c_int32), ("imu_s_t", _lp), ("imu_s_gyr", _dp), ("imu_s_acc", _dp),
        ("imu_params", ImuParamsC),
        ("n_pprior", C.c_int32), ("pprior_pose", _ip), ("pprior_meas", _dp), ("pprior_sqrtinfo", _dp),
        ("n_sbprior", C.c_int32), ("sbprior_sb", _ip), ("sbprior_meas", _dp), ("sbprior_sqrtinfo", _dp),
        ("n_relpose", C.c_int32), ("rel_pose0", _ip), ("rel_pose1", _ip), ("rel_sqrtinfo", _dp),
        ("marg_dim", C.c_int32), ("marg_nblocks", C.c_int32), ("marg_block_type", _ip),
        ("marg_block_idx", _ip), ("marg_block_off", _ip), ("marg_J", _dp), ("marg_e0", _dp),
        ("marg_lin", _dp),
        ("imu_sb_ref", _dp), ("imu_sb_ref_valid", _bp), ("imu_cache", _dp),
    ]


# okvis_ba_tuning.flags (include/okvis_amd_ba.h, OKVIS_BA_TUNE_*) and okvis_ba_tuning.solve_mode (OKVIS_BA_SOLVE_*)
TUNE_SCHUR_DECIDES, TUNE_SCHUR_VALU, TUNE_SCHUR_MFMA_LARGE, TUNE_NO_LDL_COMP, TUNE_LDL_COMP_ALL = 0x1, 0x2, 0x4, 0x8, 0x10
TUNE_H0_ON_HOST, TUNE_NO_EARLY_PREINTEGRATION, TUNE_NO_MARG_TILES, TUNE_NO_SMALL_RIDE = 0x20, 0x40, 0x80, 0x100
SOLVE_AUTO, SOLVE_DENSE, SOLVE_CHAIN = 0, 1, 2


class TuningC(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("fused_max_windows", C.c_int32), ("group_lm", C.c_int32), ("group_work", C.c_int32),
        ("split_small_min", C.c_int32), ("lin2_occupancy", C.c_int32), ("stagger_us", C.c_int32), ("solve_mode", C.c_int32),
    ]


class OptionsC(C.Structure):
    _fields_ = [
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("min_relative_decrease", C.c_double), ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("use_graph", C.c_int32), ("schur_lm_per_block", C.c_int32),
        ("debug_arrays", C.c_int32), ("gauss_newton", C.c_int32),
        ("n_streams", C.c_int32), ("fp32_linearize", C.c_int32),
        ("strategy", C.c_int32), ("jacobi_scaling", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
        ("reserved0", C.c_int32), ("tuning", TuningC),
    ]


class SummaryC(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32),
        ("successful_steps", C.c_int32), ("termination", C.c_int32), ("reserved", C.c_int32),
        ("final_radius", C.c_double), ("gradient_max_norm", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class LimitsC(C.Structure):
    _fields_ = [("max_obs_per_lm", C.c_int32), ("max_reduced_dim", C.c_int32),
                ("max_marg_dim", C.c_int32), ("max_imu_samples_per_factor", C.c_int32)]


class MargSpecC(C.Structure):
    """ctypes image of ``okvis_ba_marg_spec``."""
    _fields_ = [("pose_marg", _bp), ("sb_marg", _bp), ("prior_dim", C.c_int32), ("prior_nblocks", C.c_int32),
                ("prior_block_type", _ip), ("prior_block_idx", _ip), ("prior_block_off", _ip),
                ("prior_H", _dp), ("prior_b0", _dp)]


class MargResultC(C.Structure):
    """ctypes image of ``okvis_ba_marg_result``."""
    _fields_ = [("capacity_dim", C.c_int32), ("capacity_blocks", C.c_int32), ("dim", C.c_int32),
                ("nblocks", C.c_int32), ("rank", C.c_int32), ("block_type", _ip), ("block_idx", _ip),
                ("block_off", _ip), ("H", _dp), ("b0", _dp), ("J", _dp), ("e0", _dp), ("sweeps", C.c_int32 * 2)]


def marg_call(fn, n_pose, n_sb, pose_marg, sb_marg, prior=None):
    """Marshal one okvis_ba_marginalize-shaped call ``fn(spec*, result*) -> status``.
    prior = dict(block_type, block_idx, H, b0) over blocks of the uploaded window, or None."""
    pm = np.ascontiguousarray(pose_marg, np.uint8).reshape(-1)
    sm = np.ascontiguousarray(sb_marg, np.uint8).reshape(-1)
    assert pm.size == n_pose and sm.size == n_sb
    if pm.size == 0:
        pm = np.zeros(1, np.uint8)
    if sm.size == 0:
        sm = np.zeros(1, np.uint8)
    spec = MargSpecC()
    spec.pose_marg = pm.ctypes.data_as(_bp)
    spec.sb_marg = sm.ctypes.data_as(_bp)
    keep = [pm, sm]
    if prior is not None and len(prior["block_type"]) > 0:
        bt = np.ascontiguousarray(prior["block_type"], np.int32)
        bi = np.ascontiguousarray(prior["block_idx"], np.int32)
        dims = np.where(bt == 0, 6, 9)
        bo = np.ascontiguousarray(np.concatenate([[0], np.cumsum(dims)[:-1]]), np.int32)
        pd = int(dims.sum())
        Hm = np.ascontiguousarray(prior["H"], np.float64).reshape(pd, pd)
        b0 = np.ascontiguousarray(prior["b0"], np.float64).reshape(pd)
        spec.prior_dim, spec.prior_nblocks = pd, int(bt.size)
        spec.prior_block_type, spec.prior_block_idx, spec.prior_block_off = (a.ctypes.data_as(_ip) for a in (bt, bi, bo))
        spec.prior_H, spec.prior_b0 = Hm.ctypes.data_as(_dp), b0.ctypes.data_as(_dp)
        keep += [bt, bi, bo, Hm, b0]
    cap = 6 * n_pose + 9 * n_sb
    capb = max(1, n_pose + n_sb)
    out = dict(block_type=np.zeros(capb, np.int32), block_idx=np.zeros(capb, np.int32), block_off=np.zeros(capb, np.int32),
               H=np.zeros(max(1, cap * cap)), b0=np.zeros(max(1, cap)), J=np.zeros(max(1, cap * cap)), e0=np.zeros(max(1, cap)))
    res = MargResultC()
    res.capacity_dim, res.capacity_blocks = cap, capb
    for k in ("block_type", "block_idx", "block_off"):
        setattr(res, k, out[k].ctypes.data_as(_ip))
    for k in ("H", "b0", "J", "e0"):
        setattr(res, k, out[k].ctypes.data_as(_dp))
    status = fn(C.byref(spec), C.byref(res))
    del keep
    if status != 0:
        return status, None
    n, nb = int(res.dim), int(res.nblocks)
    return 0, dict(dim=n, rank=int(res.rank), sweeps=(int(res.sweeps[0]), int(res.sweeps[1])), block_type=out["block_type"][:nb].copy(), block_idx=out["block_idx"][:nb].copy(),
                   block_off=out["block_off"][:nb].copy(), H=out["H"][:n * n].reshape(n, n).copy(), b0=out["b0"][:n].copy(),
                   J=out["J"][:n * n].reshape(n, n).copy(), e0=out["e0"][:n].copy())


STRATEGY_DOGLEG, STRATEGY_LM = 0, 1
DEFAULT_STRATEGY = STRATEGY_DOGLEG


def default_options(strategy=None) -> OptionsC:
    """Ceres 1.9 defaults restated from its documentation (not in the reference tree)."""
    return OptionsC(1e4, 1e16, 1e-32, 1e-6, 1e32, 1e-3, 1e-6, 1e-10, 1e-8, 1, 0, 0, 0, 0, 0,
                    DEFAULT_STRATEGY if strategy is None else strategy, 1, 5, 0)


def set_options(o: OptionsC, **kw) -> OptionsC:
    """o.<name> = value; names starting with `tuning_` address okvis_ba_options::tuning (tuning_group_lm=64, tuning_flags=...)."""
    for k, v in kw.items():
        if k.startswith("tuning_"):
            setattr(o.tuning, k[7:], v)
        else:
            setattr(o, k, v)
    return o


def _f64(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
    return a


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))


@dataclass
class ImuParams:
    """Subset of okvis::ImuParameters reaching the hot path (okvis_common Parameters.hpp)."""
    sigma_g_c: float = 12.0e-4
    sigma_a_c: float = 8.0e-3
    sigma_gw_c: float = 4.0e-6
    sigma_aw_c: float = 4.0e-5
    g: float = 9.81007
    g_max: float = 7.8
    a_max: float = 176.0
    sigma_bg: float = 0.03   # only used to build the first speed/bias prior (Estimator.cpp:272-277)
    sigma_ba: float = 0.1

    def as_c(self) -> ImuParamsC:
        return ImuParamsC(self.sigma_g_c, self.sigma_a_c, self.sigma_gw_c, self.sigma_aw_c, self.g,
                          self.g_max, self.a_max)


@dataclass
class Window:
    """One sliding window.  All arrays are numpy, converted to the C layout by :meth:`as_c`."""
    pose: np.ndarray                      # [n_pose,7]
    pose_fixed: np.ndarray                # [n_pose] u8
    sb: np.ndarray                        # [n_sb,9]
    sb_fixed: np.ndarray
    lm: np.ndarray                        # [n_lm,4]
    cam_intr: np.ndarray                  # [n_cam,12]
    cam_model: np.ndarray                 # [n_cam]
    obs_lm: np.ndarray
    obs_pose: np.ndarray
    obs_ext: np.ndarray
    obs_cam: np.ndarray
    obs_uv: np.ndarray                    # [n_obs,2]
    obs_sqrtw: np.ndarray                 # [n_obs]
    cauchy_b: float = 1.0
    imu_pose0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_sb0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_pose1: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_sb1: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_t0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    imu_t1: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    imu_s_begin: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_s_count: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    imu_s_t: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    imu_s_gyr: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    imu_s_acc: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    imu_params: ImuParams = field(default_factory=ImuParams)
    pprior_pose: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    pprior_meas: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    pprior_sqrtinfo: np.ndarray = field(default_factory=lambda: np.zeros((0, 36)))
    sbprior_sb: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sbprior_meas: np.ndarray = field(default_factory=lambda: np.zeros((0, 9)))
    sbprior_sqrtinfo: np.ndarray = field(default_factory=lambda: np.zeros((0, 81)))
    rel_pose0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    rel_pose1: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    rel_sqrtinfo: np.ndarray = field(default_factory=lambda: np.zeros((0, 36)))
    marg_block_type: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    marg_block_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    marg_block_off: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    marg_J: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    marg_e0: np.ndarray = field(default_factory=lambda: np.zeros(0))
    marg_lin: np.ndarray = field(default_factory=lambda: np.zeros((0, 9)))
    imu_sb_ref: np.ndarray = field(default_factory=lambda: np.zeros((0, 9)))       # [n_imu,9] or empty
    imu_sb_ref_valid: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))   # 0 / 1 (reference bias) / 2 (+ imu_cache)
    imu_cache: np.ndarray = field(default_factory=lambda: np.zeros((0, IMU_CACHE_DOUBLES)))   # [n_imu, 290] or empty (WindowBatch.fetch_imu_caches)
    meta: dict = field(default_factory=dict)   # generator bookkeeping (truth etc.); never uploaded

    # ------------------------------------------------------------------------------------------
    @property
    def n_pose(self): return int(np.asarray(self.pose).reshape(-1, 7).shape[0])
    @property
    def n_sb(self): return int(np.asarray(self.sb).reshape(-1, 9).shape[0])
    @property
    def n_lm(self): return int(np.asarray(self.lm).reshape(-1, 4).shape[0])
    @property
    def n_obs(self): return int(np.asarray(self.obs_lm).size)
    @property
    def n_imu(self): return int(np.asarray(self.imu_pose0).size)

    def reduced_dim(self) -> int:
        return 6 * int((np.asarray(self.pose_fixed) == 0).sum()) + 9 * int((np.asarray(self.sb_fixed) == 0).sum())

    def sort_observations(self) -> None:
        """Establish the (lm, pose, cam) order the C-ABI requires."""
        order = np.lexsort((self.obs_cam, self.obs_pose, self.obs_lm))
        for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_uv", "obs_sqrtw"):
            setattr(self, n, np.asarray(getattr(self, n))[order])

    def validate(self) -> None:
        """Host-side argument checks (what OKVIS_ASSERT_* would catch in the reference)."""
        npz, nl = self.n_pose, self.n_lm
        for n in ("obs_pose", "obs_ext"):
            a = np.asarray(getattr(self, n))
            if a.size and (a.min() < 0 or a.max() >= npz):
                raise ValueError(f"{n} out of range")
        a = np.asarray(self.obs_lm)
        if a.size and (a.min() < 0 or a.max() >= nl):
            raise ValueError("obs_lm out of range")
        if a.size:
            key = np.stack([self.obs_lm, self.obs_pose, self.obs_cam], 1).astype(np.int64)
            k = (key[:, 0] * (npz + 1) + key[:, 1]) * (len(self.cam_model) + 1) + key[:, 2]
            if np.any(np.diff(k) < 0):
                raise ValueError("observations must be sorted by (lm, pose, cam)")
            # equal keys are legal: one landmark matched to two keypoints of the same image gives two residual blocks in
            # the reference (implementation/Estimator.hpp:52-56 only rejects an identical KeypointIdentifier)
        md = np.asarray(self.marg_e0).size
        if md and np.asarray(self.marg_J).shape != (md, md):
            raise ValueError("marg_J must be [marg_dim, marg_dim]")

    def as_c(self):
        """Return (WindowC, keepalive list). Arrays are made contiguous with the ABI dtypes."""
        k = {}
        k["pose"] = _f64(self.pose, (-1, 7)); k["pose_fixed"] = np.ascontiguousarray(self.pose_fixed, np.uint8)
        k["sb"] = _f64(self.sb, (-1, 9)); k["sb_fixed"] = np.ascontiguousarray(self.sb_fixed, np.uint8)
        k["lm"] = _f64(self.lm, (-1, 4))
        k["cam_intr"] = _f64(self.cam_intr, (-1, 12)); k["cam_model"] = _i32(self.cam_model)
        for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "imu_pose0", "imu_sb0", "imu_pose1", "imu_sb1",
                  "imu_s_begin", "imu_s_count", "pprior_pose", "sbprior_sb", "rel_pose0", "rel_pose1",
                  "marg_block_type", "marg_block_idx", "marg_block_off"):
            k[n] = _i32(getattr(self, n))
        for n in ("imu_t0", "imu_t1", "imu_s_t"):
            k[n] = _i64(getattr(self, n))
        k["obs_uv"] = _f64(self.obs_uv, (-1, 2)); k["obs_sqrtw"] = _f64(self.obs_sqrtw, (-1,))
        k["imu_s_gyr"] = _f64(self.imu_s_gyr, (-1, 3)); k["imu_s_acc"] = _f64(self.imu_s_acc, (-1, 3))
        k["pprior_meas"] = _f64(self.pprior_meas, (-1, 7)); k["pprior_sqrtinfo"] = _f64(self.pprior_sqrtinfo, (-1, 36))
        k["sbprior_meas"] = _f64(self.sbprior_meas, (-1, 9)); k["sbprior_sqrtinfo"] = _f64(self.sbprior_sqrtinfo, (-1, 81))
        k["rel_sqrtinfo"] = _f64(self.rel_sqrtinfo, (-1, 36))
        md = int(np.asarray(self.marg_e0).size)
        k["marg_J"] = _f64(self.marg_J, (md, md)); k["marg_e0"] = _f64(self.marg_e0, (-1,))
        k["marg_lin"] = _f64(self.marg_lin, (-1, 9))
        k["imu_sb_ref"] = _f64(self.imu_sb_ref, (-1, 9))
        k["imu_sb_ref_valid"] = np.ascontiguousarray(self.imu_sb_ref_valid, np.uint8)
        k["imu_cache"] = _f64(self.imu_cache, (-1, IMU_CACHE_DOUBLES))

        def p(name, typ):
            a = k[name]
            return a.ctypes.data_as(typ) if a.size else C.cast(None, typ)

        w = WindowC()
        w.n_pose = k["pose"].shape[0]; w.pose = p("pose", _dp); w.pose_fixed = p("pose_fixed", _bp)
        w.n_sb = k["sb"].shape[0]; w.sb = p("sb", _dp); w.sb_fixed = p("sb_fixed", _bp)
        w.n_lm = k["lm"].shape[0]; w.lm = p("lm", _dp)
        w.n_cam = k["cam_intr"].shape[0]; w.cam_intr = p("cam_intr", _dp); w.cam_model = p("cam_model", _ip)
        w.n_obs = k["obs_lm"].size
        for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam"):
            setattr(w, n, p(n, _ip))
        w.obs_uv = p("obs_uv", _dp); w.obs_sqrtw = p("obs_sqrtw", _dp); w.cauchy_b = float(self.cauchy_b)
        w.n_imu = k["imu_pose0"].size
        for n in ("imu_pose0", "imu_sb0", "imu_pose1", "imu_sb1", "imu_s_begin", "imu_s_count"):
            setattr(w, n, p(n, _ip))
        w.imu_t0 = p("imu_t0", _lp); w.imu_t1 = p("imu_t1", _lp)
        w.n_imu_samples = k["imu_s_t"].size; w.imu_s_t = p("imu_s_t", _lp)
        w.imu_s_gyr = p("imu_s_gyr", _dp); w.imu_s_acc = p("imu_s_acc", _dp)
        w.imu_params = self.imu_params.as_c()
        w.n_pprior = k["pprior_pose"].size; w.pprior_pose = p("pprior_pose", _ip)
        w.pprior_meas = p("pprior_meas", _dp); w.pprior_sqrtinfo = p("pprior_sqrtinfo", _dp)
        w.n_sbprior = k["sbprior_sb"].size; w.sbprior_sb = p("sbprior_sb", _ip)
        w.sbprior_meas = p("sbprior_meas", _dp); w.sbprior_sqrtinfo = p("sbprior_sqrtinfo", _dp)
        w.n_relpose = k["rel_pose0"].size; w.rel_pose0 = p("rel_pose0", _ip); w.rel_pose1 = p("rel_pose1", _ip)
        w.rel_sqrtinfo = p("rel_sqrtinfo", _dp)
        w.marg_dim = md; w.marg_nblocks = k["marg_block_type"].size
        w.marg_block_type = p("marg_block_type", _ip); w.marg_block_idx = p("marg_block_idx", _ip)
        w.marg_block_off = p("marg_block_off", _ip)
        w.marg_J = p("marg_J", _dp); w.marg_e0 = p("marg_e0", _dp); w.marg_lin = p("marg_lin", _dp)
        if k["imu_sb_ref"].shape[0] == w.n_imu and k["imu_sb_ref_valid"].size == w.n_imu and w.n_imu > 0:
            w.imu_sb_ref = p("imu_sb_ref", _dp); w.imu_sb_ref_valid = p("imu_sb_ref_valid", _bp)
            if k["imu_cache"].shape[0] == w.n_imu:
                w.imu_cache = p("imu_cache", _dp)
        return w, k


# ------------------------------------------------------------------------------------------------------------------
# incremental structure updates (okvis_ba_patch, include/okvis_amd_ba.h)
# ------------------------------------------------------------------------------------------------------------------
PATCH_POSE_PRIORS, PATCH_SB_PRIORS, PATCH_RELPOSE, PATCH_MARG_PRIOR = 1, 2, 4, 8


class PatchC(C.Structure):
    """ctypes image of ``okvis_ba_patch`` (field order must match the header)."""
    _fields_ = [
        ("n_remove_obs", C.c_int32), ("remove_obs", _ip), ("n_remove_lm", C.c_int32), ("remove_lm", _ip),
        ("n_remove_pose", C.c_int32), ("remove_pose", _ip), ("n_remove_sb", C.c_int32), ("remove_sb", _ip),
        ("n_remove_imu", C.c_int32), ("remove_imu", _ip),
        ("n_add_pose", C.c_int32), ("add_pose", _dp), ("add_pose_fixed", _bp),
        ("n_add_sb", C.c_int32), ("add_sb", _dp), ("add_sb_fixed", _bp),
        ("n_add_lm", C.c_int32), ("add_lm", _dp),
        ("n_add_obs", C.c_int32), ("add_obs_lm", _ip), ("add_obs_pose", _ip), ("add_obs_ext", _ip), ("add_obs_cam", _ip),
        ("add_obs_uv", _dp), ("add_obs_sqrtw", _dp),
        ("n_add_imu", C.c_int32), ("add_imu_pose0", _ip), ("add_imu_sb0", _ip), ("add_imu_pose1", _ip), ("add_imu_sb1", _ip),
        ("add_imu_t0", _lp), ("add_imu_t1", _lp), ("add_imu_s_begin", _ip), ("add_imu_s_count", _ip),
        ("n_add_imu_samples", C.c_int32), ("add_imu_s_t", _lp), ("add_imu_s_gyr", _dp), ("add_imu_s_acc", _dp),
        ("replace", C.c_int32),
        ("n_pprior", C.c_int32), ("pprior_pose", _ip), ("pprior_meas", _dp), ("pprior_sqrtinfo", _dp),
        ("n_sbprior", C.c_int32), ("sbprior_sb", _ip), ("sbprior_meas", _dp), ("sbprior_sqrtinfo", _dp),
        ("n_relpose", C.c_int32), ("rel_pose0", _ip), ("rel_pose1", _ip), ("rel_sqrtinfo", _dp),
        ("marg_dim", C.c_int32), ("marg_nblocks", C.c_int32), ("marg_block_type", _ip), ("marg_block_idx", _ip),
        ("marg_block_off", _ip), ("marg_J", _dp), ("marg_e0", _dp), ("marg_lin", _dp),
        ("n_set_pose", C.c_int32), ("set_pose_idx", _ip), ("set_pose", _dp),
        ("n_set_sb", C.c_int32), ("set_sb_idx", _ip), ("set_sb", _dp),
        ("n_set_lm", C.c_int32), ("set_lm_idx", _ip), ("set_lm", _dp),
        ("add_lm_before", _ip),
    ]


def _e(dtype=np.float64, *shape):
    return field(default_factory=lambda: np.zeros(shape or (0,), dtype))


@dataclass
class Patch:
    """One batch of edits to a window (``okvis_ba_patch``): removals by current index, then appended blocks and terms in the
    new numbering, replaced prior families, sparse value updates."""
    remove_obs: np.ndarray = _e(np.int32)
    remove_lm: np.ndarray = _e(np.int32)
    remove_pose: np.ndarray = _e(np.int32)
    remove_sb: np.ndarray = _e(np.int32)
    remove_imu: np.ndarray = _e(np.int32)
    add_pose: np.ndarray = _e(np.float64, 0, 7)
    add_pose_fixed: np.ndarray = _e(np.uint8)
    add_sb: np.ndarray = _e(np.float64, 0, 9)
    add_sb_fixed: np.ndarray = _e(np.uint8)
    add_lm: np.ndarray = _e(np.float64, 0, 4)
    add_obs_lm: np.ndarray = _e(np.int32)
    add_obs_pose: np.ndarray = _e(np.int32)
    add_obs_ext: np.ndarray = _e(np.int32)
    add_obs_cam: np.ndarray = _e(np.int32)
    add_obs_uv: np.ndarray = _e(np.float64, 0, 2)
    add_obs_sqrtw: np.ndarray = _e(np.float64)
    add_imu_pose0: np.ndarray = _e(np.int32)
    add_imu_sb0: np.ndarray = _e(np.int32)
    add_imu_pose1: np.ndarray = _e(np.int32)
    add_imu_sb1: np.ndarray = _e(np.int32)
    add_imu_t0: np.ndarray = _e(np.int64)
    add_imu_t1: np.ndarray = _e(np.int64)
    add_imu_s_begin: np.ndarray = _e(np.int32)
    add_imu_s_count: np.ndarray = _e(np.int32)
    add_imu_s_t: np.ndarray = _e(np.int64)
    add_imu_s_gyr: np.ndarray = _e(np.float64, 0, 3)
    add_imu_s_acc: np.ndarray = _e(np.float64, 0, 3)
    replace: int = 0
    pprior_pose: np.ndarray = _e(np.int32)
    pprior_meas: np.ndarray = _e(np.float64, 0, 7)
    pprior_sqrtinfo: np.ndarray = _e(np.float64, 0, 36)
    sbprior_sb: np.ndarray = _e(np.int32)
    sbprior_meas: np.ndarray = _e(np.float64, 0, 9)
    sbprior_sqrtinfo: np.ndarray = _e(np.float64, 0, 81)
    rel_pose0: np.ndarray = _e(np.int32)
    rel_pose1: np.ndarray = _e(np.int32)
    rel_sqrtinfo: np.ndarray = _e(np.float64, 0, 36)
    marg_block_type: np.ndarray = _e(np.int32)
    marg_block_idx: np.ndarray = _e(np.int32)
    marg_block_off: np.ndarray = _e(np.int32)
    marg_J: np.ndarray = _e(np.float64, 0, 0)
    marg_e0: np.ndarray = _e(np.float64)
    marg_lin: np.ndarray = _e(np.float64, 0, 9)
    set_pose_idx: np.ndarray = _e(np.int32)
    set_pose: np.ndarray = _e(np.float64, 0, 7)
    set_sb_idx: np.ndarray = _e(np.int32)
    set_sb: np.ndarray = _e(np.float64, 0, 9)
    set_lm_idx: np.ndarray = _e(np.int32)
    set_lm: np.ndarray = _e(np.float64, 0, 4)
    add_lm_before: np.ndarray = _e(np.int32)   # empty = appended at the end

    _I32 = ("remove_obs", "remove_lm", "remove_pose", "remove_sb", "remove_imu", "add_obs_lm", "add_obs_pose", "add_obs_ext",
            "add_obs_cam", "add_imu_pose0", "add_imu_sb0", "add_imu_pose1", "add_imu_sb1", "add_imu_s_begin", "add_imu_s_count",
            "pprior_pose", "sbprior_sb", "rel_pose0", "rel_pose1", "marg_block_type", "marg_block_idx", "marg_block_off",
            "set_pose_idx", "set_sb_idx", "set_lm_idx", "add_lm_before")
    _I64 = ("add_imu_t0", "add_imu_t1", "add_imu_s_t")
    _U8 = ("add_pose_fixed", "add_sb_fixed")
    _F64 = dict(add_pose=7, add_sb=9, add_lm=4, add_obs_uv=2, add_obs_sqrtw=0, add_imu_s_gyr=3, add_imu_s_acc=3, pprior_meas=7,
                pprior_sqrtinfo=36, sbprior_meas=9, sbprior_sqrtinfo=81, rel_sqrtinfo=36, marg_e0=0, marg_lin=9, set_pose=7,
                set_sb=9, set_lm=4)

    def as_c(self):
        """Return (PatchC, keepalive dict)."""
        k = {}
        for n in self._I32:
            k[n] = _i32(getattr(self, n))
        for n in self._I64:
            k[n] = _i64(getattr(self, n))
        for n in self._U8:
            k[n] = np.ascontiguousarray(np.asarray(getattr(self, n), np.uint8).reshape(-1))
        for n, wdt in self._F64.items():
            k[n] = _f64(getattr(self, n), (-1, wdt) if wdt else (-1,))
        md = int(k["marg_e0"].size)
        k["marg_J"] = _f64(self.marg_J, (md, md))
        typ = {np.dtype(np.int32): _ip, np.dtype(np.int64): _lp, np.dtype(np.uint8): _bp, np.dtype(np.float64): _dp}
        p = PatchC()
        for n, a in k.items():
            setattr(p, n, a.ctypes.data_as(typ[a.dtype]) if a.size else C.cast(None, typ[a.dtype]))
        p.n_remove_obs, p.n_remove_lm, p.n_remove_pose = k["remove_obs"].size, k["remove_lm"].size, k["remove_pose"].size
        p.n_remove_sb, p.n_remove_imu = k["remove_sb"].size, k["remove_imu"].size
        p.n_add_pose, p.n_add_sb, p.n_add_lm = k["add_pose"].shape[0], k["add_sb"].shape[0], k["add_lm"].shape[0]
        p.n_add_obs, p.n_add_imu, p.n_add_imu_samples = k["add_obs_lm"].size, k["add_imu_pose0"].size, k["add_imu_s_t"].size
        p.replace = int(self.replace)
        p.n_pprior, p.n_sbprior, p.n_relpose = k["pprior_pose"].size, k["sbprior_sb"].size, k["rel_pose0"].size
        p.marg_dim, p.marg_nblocks = md, k["marg_block_type"].size
        p.n_set_pose, p.n_set_sb, p.n_set_lm = k["set_pose_idx"].size, k["set_sb_idx"].size, k["set_lm_idx"].size
        return p, k


def window_from_c(w: WindowC) -> Window:
    """Copy an ``okvis_ba_window`` (e.g. okvis_ba_store_view) into a :class:`Window`."""
    def arr(ptr, n, dtype, shape=None):
        if n == 0 or not ptr:
            return np.zeros(shape if shape is not None else (0,), dtype)
        a = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
        return a.reshape(shape) if shape is not None else a
    np_, ns, nl, no, ni = w.n_pose, w.n_sb, w.n_lm, w.n_obs, w.n_imu
    md, nb = w.marg_dim, w.marg_nblocks
    ip = w.imu_params
    out = Window(
        pose=arr(w.pose, 7 * np_, np.float64, (np_, 7)), pose_fixed=arr(w.pose_fixed, np_, np.uint8),
        sb=arr(w.sb, 9 * ns, np.float64, (ns, 9)), sb_fixed=arr(w.sb_fixed, ns, np.uint8), lm=arr(w.lm, 4 * nl, np.float64, (nl, 4)),
        cam_intr=arr(w.cam_intr, 12 * w.n_cam, np.float64, (w.n_cam, 12)), cam_model=arr(w.cam_model, w.n_cam, np.int32),
        obs_lm=arr(w.obs_lm, no, np.int32), obs_pose=arr(w.obs_pose, no, np.int32), obs_ext=arr(w.obs_ext, no, np.int32),
        obs_cam=arr(w.obs_cam, no, np.int32), obs_uv=arr(w.obs_uv, 2 * no, np.float64, (no, 2)), obs_sqrtw=arr(w.obs_sqrtw, no, np.float64),
        cauchy_b=float(w.cauchy_b),
        imu_pose0=arr(w.imu_pose0, ni, np.int32), imu_sb0=arr(w.imu_sb0, ni, np.int32), imu_pose1=arr(w.imu_pose1, ni, np.int32),
        imu_sb1=arr(w.imu_sb1, ni, np.int32), imu_t0=arr(w.imu_t0, ni, np.int64), imu_t1=arr(w.imu_t1, ni, np.int64),
        imu_s_begin=arr(w.imu_s_begin, ni, np.int32), imu_s_count=arr(w.imu_s_count, ni, np.int32),
        imu_s_t=arr(w.imu_s_t, w.n_imu_samples, np.int64), imu_s_gyr=arr(w.imu_s_gyr, 3 * w.n_imu_samples, np.float64, (w.n_imu_samples, 3)),
        imu_s_acc=arr(w.imu_s_acc, 3 * w.n_imu_samples, np.float64, (w.n_imu_samples, 3)),
        imu_params=ImuParams(ip.sigma_g_c, ip.sigma_a_c, ip.sigma_gw_c, ip.sigma_aw_c, ip.g, ip.g_max, ip.a_max),
        pprior_pose=arr(w.pprior_pose, w.n_pprior, np.int32), pprior_meas=arr(w.pprior_meas, 7 * w.n_pprior, np.float64, (w.n_pprior, 7)),
        pprior_sqrtinfo=arr(w.pprior_sqrtinfo, 36 * w.n_pprior, np.float64, (w.n_pprior, 36)),
        sbprior_sb=arr(w.sbprior_sb, w.n_sbprior, np.int32), sbprior_meas=arr(w.sbprior_meas, 9 * w.n_sbprior, np.float64, (w.n_sbprior, 9)),
        sbprior_sqrtinfo=arr(w.sbprior_sqrtinfo, 81 * w.n_sbprior, np.float64, (w.n_sbprior, 81)),
        rel_pose0=arr(w.rel_pose0, w.n_relpose, np.int32), rel_pose1=arr(w.rel_pose1, w.n_relpose, np.int32),
        rel_sqrtinfo=arr(w.rel_sqrtinfo, 36 * w.n_relpose, np.float64, (w.n_relpose, 36)),
        marg_block_type=arr(w.marg_block_type, nb if md else 0, np.int32), marg_block_idx=arr(w.marg_block_idx, nb if md else 0, np.int32),
        marg_block_off=arr(w.marg_block_off, nb if md else 0, np.int32), marg_J=arr(w.marg_J, md * md, np.float64, (md, md)),
        marg_e0=arr(w.marg_e0, md, np.float64), marg_lin=arr(w.marg_lin, 9 * (nb if md else 0), np.float64, (nb if md else 0, 9)),
        imu_sb_ref=arr(w.imu_sb_ref, 9 * ni, np.float64, (ni, 9)), imu_sb_ref_valid=arr(w.imu_sb_ref_valid, ni, np.uint8),
        imu_cache=arr(w.imu_cache, IMU_CACHE_DOUBLES * ni, np.float64, (ni, IMU_CACHE_DOUBLES)))
    return out
