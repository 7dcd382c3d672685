"""ctypes loader of the HIP shared library.  Fails loudly: there is no Python/CPU fallback path."""
from __future__ import annotations

import ctypes as C
import os

from .window import LimitsC, MargResultC, MargSpecC, OptionsC, PatchC, SummaryC, WindowC

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OKVIS_AMD_LIB_DIR: a directory holding an instrumented build of the same sources — scripts/host_sanitize.sh)
LIB_PATH = os.path.join(os.environ.get("OKVIS_AMD_LIB_DIR") or os.path.join(_HERE, "lib"), "libokvis_amd_ba.so")

# every symbol include/okvis_amd_ba.h declares
SYMBOLS = [
    "okvis_ba_abi_version", "okvis_ba_get_limits", "okvis_ba_default_options", "okvis_ba_error_string",
    "okvis_ba_create", "okvis_ba_destroy", "okvis_ba_upload", "okvis_ba_check_window", "okvis_ba_check_window_lists", "okvis_ba_set_state", "okvis_ba_set_options",
    "okvis_ba_optimize", "okvis_ba_optimize_timed", "okvis_ba_begin", "okvis_ba_iterate", "okvis_ba_finish",
    "okvis_ba_evaluate_cost", "okvis_ba_get_state", "okvis_ba_fetch_results", "okvis_ba_fetch_imu_caches", "okvis_ba_array_size", "okvis_ba_download",
    "okvis_ba_reduced_dim", "okvis_ba_pair_count", "okvis_ba_pairs", "okvis_ba_last_iterate_ms",
    "okvis_ba_profile_iterations", "okvis_ba_profile_launches", "okvis_ba_algorithmic_bytes", "okvis_ba_synchronize",
    "okvis_ba_helper_timeouts", "okvis_ba_launch_route", "okvis_ba_marginalize", "okvis_ba_marginalize_begin", "okvis_ba_marginalize_end",
    "okvis_ba_store_create", "okvis_ba_store_patch", "okvis_ba_store_view", "okvis_ba_store_destroy", "okvis_ba_set_patchable",
    "okvis_ba_patch_window", "okvis_ba_patched_view", "okvis_ba_set_marg_prior_values",
    "okvis_ba_dense_solve", "okvis_ba_reduced_solve", "okvis_ba_shard", "okvis_ba_batch_run", "okvis_ba_gather_records", "okvis_ba_batch_run_gathered",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None


class BackendError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().okvis_ba_error_string(status).decode() if _lib is not None else str(status)
        super().__init__(f"okvis_ba status {status}: {msg} {what}")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -m okvis_amd.build` (hipcc, gfx950). "
                          "okvis_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        getattr(L, s)  # raises AttributeError if the library does not export the header's symbol
    vp = C.c_void_p
    L.okvis_ba_error_string.restype = C.c_char_p
    L.okvis_ba_error_string.argtypes = [C.c_int]
    L.okvis_ba_get_limits.argtypes = [C.POINTER(LimitsC)]
    L.okvis_ba_default_options.argtypes = [C.POINTER(OptionsC)]
    L.okvis_ba_create.argtypes = [C.POINTER(vp), C.c_int]
    L.okvis_ba_destroy.argtypes = [vp]
    L.okvis_ba_upload.argtypes = [vp, C.c_int, C.POINTER(WindowC)]
    L.okvis_ba_set_state.argtypes = [vp, C.c_int, _dp, _dp, _dp]
    L.okvis_ba_store_create.argtypes = [C.POINTER(WindowC), C.POINTER(vp)]
    L.okvis_ba_store_patch.argtypes = [vp, C.POINTER(PatchC)]
    L.okvis_ba_store_view.argtypes = [vp, C.POINTER(WindowC)]
    L.okvis_ba_store_destroy.argtypes = [vp]
    L.okvis_ba_store_destroy.restype = None
    L.okvis_ba_set_patchable.argtypes = [vp, C.c_int]
    L.okvis_ba_patch_window.argtypes = [vp, C.c_int, C.POINTER(PatchC)]
    L.okvis_ba_patched_view.argtypes = [vp, C.c_int, C.POINTER(WindowC)]
    L.okvis_ba_set_marg_prior_values.argtypes = [vp, C.c_int, _dp, _dp]
    L.okvis_ba_check_window.argtypes = [C.POINTER(WindowC), C.POINTER(OptionsC), C.POINTER(C.c_int64)]
    L.okvis_ba_check_window_lists.argtypes = [C.POINTER(WindowC), C.POINTER(OptionsC), C.c_int32, C.c_int32, _ip, C.c_int64,
                                              C.POINTER(C.c_int64)]
    L.okvis_ba_get_state.argtypes = [vp, C.c_int, _dp, _dp, _dp]
    L.okvis_ba_fetch_results.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    L.okvis_ba_fetch_imu_caches.argtypes = [vp, C.c_int, _dp]
    L.okvis_ba_set_options.argtypes = [vp, C.POINTER(OptionsC)]
    L.okvis_ba_optimize.argtypes = [vp, C.c_int, C.POINTER(SummaryC)]
    L.okvis_ba_optimize_timed.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.POINTER(SummaryC)]
    L.okvis_ba_begin.argtypes = [vp]
    L.okvis_ba_iterate.argtypes = [vp, C.c_int]
    L.okvis_ba_finish.argtypes = [vp, C.POINTER(SummaryC)]
    L.okvis_ba_dense_solve.argtypes = [C.c_int, C.c_int32, _dp, _dp, _dp, C.POINTER(C.c_int32)]
    L.okvis_ba_reduced_solve.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, _dp, _dp, _dp, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int32), C.c_int32, _dp, C.c_int64]
    L.okvis_ba_shard.argtypes = [C.c_int32, C.c_int32, C.c_int32, _ip, C.POINTER(C.c_int32)]
    L.okvis_ba_batch_run.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.POINTER(WindowC), C.POINTER(OptionsC), C.c_int,
                                     C.c_void_p, C.POINTER(C.c_int32)]
    L.okvis_ba_marginalize.argtypes = [vp, C.c_int, C.POINTER(MargSpecC), C.POINTER(MargResultC)]
    L.okvis_ba_marginalize_begin.argtypes = [vp, C.c_int, C.POINTER(MargSpecC), C.POINTER(MargResultC)]
    L.okvis_ba_marginalize_end.argtypes = [vp, C.POINTER(MargResultC)]
    L.okvis_ba_evaluate_cost.argtypes = [vp, _dp]
    L.okvis_ba_array_size.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.okvis_ba_download.argtypes = [vp, C.c_int, C.c_int, _dp, C.c_int64]
    L.okvis_ba_reduced_dim.argtypes = [vp, C.c_int, _ip]
    L.okvis_ba_pair_count.argtypes = [vp, C.c_int, _ip]
    L.okvis_ba_helper_timeouts.argtypes = [vp, C.POINTER(C.c_int64)]
    L.okvis_ba_launch_route.argtypes = [vp, _ip]
    L.okvis_ba_pairs.argtypes = [vp, C.c_int, _ip, _ip]
    L.okvis_ba_last_iterate_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.okvis_ba_profile_iterations.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.okvis_ba_profile_launches.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.okvis_ba_algorithmic_bytes.argtypes = [vp] + [C.POINTER(C.c_int64)] * 4
    L.okvis_ba_synchronize.argtypes = [vp]
    _lib = L
    return L


def check(status, what=""):
    if status != 0:
        raise BackendError(status, what)
