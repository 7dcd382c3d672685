"""Build the gfx950 shared library in-tree (okvis_amd/lib/libokvis_amd_ba.so) with hipcc.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libokvis_amd_ba.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
# translation units of the HIP library and the headers each one is rebuilt for
BA_HEADERS = ["ba_math.hpp", "ba_types.hpp", "ba_device.hpp", "ba_linearize.hpp", "ba_linearize2.hpp", "ba_schur.hpp", "ba_schur2.hpp", "ba_solve.hpp", "ba_chain.hpp", "ba_imu.hpp",
              "ba_marg.hpp", "ba_marg_tiles.hpp", "ba_chol_tiles.hpp", "ba_ldl16.hpp", "ba_store.hpp", os.path.join("..", "..", "include", "okvis_amd_ba.h"),
              # the parts of ba_capi.hip (one translation unit)
              "capi_solver.inc", "capi_index_build.inc", "capi_launch.inc", "capi_standalone.inc", "capi_marginalize.inc"]
UNITS = {
    "ba_capi.hip": BA_HEADERS,                                     # the bundle-adjustment path (include/okvis_amd_ba.h)
    "store_capi.hip": ["ba_store.hpp", os.path.join("..", "..", "include", "okvis_amd_ba.h")],   # host-side window container
    "dist_capi.hip": [os.path.join("..", "..", "include", "okvis_amd_ba.h")],   # record all-gather over RCCL (dlopen)
    "fe_capi.hip": ["fe_kernels.hpp", "ba_math.hpp", os.path.join("..", "..", "include", "okvis_amd_frontend.h"),
                    os.path.join("..", "..", "include", "okvis_amd_ba.h")],   # frontend pieces (include/okvis_amd_frontend.h)
}
SOURCES = list(UNITS)
HOST_SOURCES = [os.path.join("host", f) for f in ("estimator.hpp", "estimator.cpp", "replay.hpp", "replay.cpp", "replay_main.cpp", "okvis_config.hpp", "okvis_config.cpp",
                                                   "estimator_capi.cpp", "okvis_estimator_adapter.hpp")]
HEADERS = sorted({h for hs in UNITS.values() for h in hs} | set(HOST_SOURCES))


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps)


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def stale() -> bool:
    return (any(_newer(_obj(s), [s] + UNITS[s]) for s in SOURCES) or _newer(LIB, []) or
            any(os.path.getmtime(_obj(s)) > os.path.getmtime(LIB) for s in SOURCES) or _newer(HOST_LIB, HOST_SOURCES))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    relink = force or not os.path.exists(LIB)
    for s in SOURCES:
        if force or _newer(_obj(s), [s] + UNITS[s]):
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, s), "-o", _obj(s)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink or any(os.path.getmtime(_obj(s)) > os.path.getmtime(LIB) for s in SOURCES):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in SOURCES], "-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_host(verbose)
    return LIB


HOST_LIB = os.path.join(HERE, "lib", "libokvis_amd_estimator.so")
REPLAY_EXE = os.path.join(HERE, "lib", "okvis_amd_replay")


def build_host(verbose: bool = False) -> str:
    """C++ host layer (okvis_amd::Estimator + flat C wrapper), linked against the HIP library."""
    host = os.path.join(CSRC, "host")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", os.path.join(host, "estimator.cpp"),
           os.path.join(host, "estimator_capi.cpp"), os.path.join(host, "replay.cpp"), os.path.join(host, "okvis_config.cpp"), "-o", HOST_LIB, "-L" + os.path.dirname(LIB), "-lokvis_amd_ba",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # okvis_amd_replay <dataset folder> [trajectory.csv]: the backend-side counterpart of okvis_app_synchronous
    exe = ["g++", "-std=c++17", "-O2", "-Wall", os.path.join(host, "replay_main.cpp"), "-o", REPLAY_EXE,
           "-L" + os.path.dirname(LIB), "-lokvis_amd_estimator", "-lokvis_amd_ba", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(exe), flush=True)
    subprocess.check_call(exe)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
