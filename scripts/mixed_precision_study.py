#!/usr/bin/env python3
"""BASELINE configs[4]: fp32 Jacobian/Hessian build with fp64 reduced-camera solve (mixed-precision tolerance
study) on the BASELINE configs[1] window.  Prints one JSON line: final-cost deviation of the mixed path from the
fp64 path after the same number of iterations, state deviation, per-kernel time of both."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis_amd import solver, synthetic  # noqa: E402
from okvis_amd.window import default_options  # noqa: E402


def run(ws, fp32, iters, gn=False):
    opt = default_options()
    opt.fp32_linearize = 1 if fp32 else 0
    if gn:
        opt.gauss_newton = 1
        opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
    b = solver.WindowBatch(ws, options=opt)
    s = b.optimize(iters)
    st = [b.get_state(i) for i in range(len(ws))]
    b.close()
    return s, st


def throughput(ws, fp32, steps=100):
    opt = default_options()
    opt.fp32_linearize = 1 if fp32 else 0
    opt.gauss_newton = 1
    opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
    b = solver.WindowBatch(ws, options=opt)
    b.begin(); b.iterate(20); b.iterate(steps); b.synchronize()
    b.iterate(steps)
    ms = b.last_iterate_ms()
    prof = b.profile_iterations(20)
    b.finish(); b.close()
    return len(ws) * steps / (ms * 1e-3), {k: v * 1e3 / 20 for k, v in prof.items()}


def main():
    seeds = [20240923 + i for i in range(8)]
    ws = [synthetic.config_A(seed=s) for s in seeds]
    out = {"config": "BASELINE configs[4] on configs[1] windows (10 KF / 2 cam / 400 landmarks / 8000 obs), 8 seeds; DOGLEG (the default strategy), round-4 kernels (piece path of the linearise launch)"}
    for iters in (10, 30):
        s64, st64 = run(ws, False, iters)
        s32, st32 = run(ws, True, iters)
        dev = [abs(a["final_cost"] - b["final_cost"]) / b["final_cost"] for a, b in zip(s32, s64)]
        dpos = max(np.abs(a[0][:, :3] - b[0][:, :3]).max() for a, b in zip(st32, st64))
        dlm = max(np.abs(a[2][:, :3] - b[2][:, :3]).max() for a, b in zip(st32, st64))
        out[f"dogleg_{iters}_iterations"] = {
            "final_cost_rel_dev_max": max(dev), "final_cost_rel_dev_median": float(np.median(dev)),
            "iterations_fp64": [x["iterations"] for x in s64], "iterations_mixed": [x["iterations"] for x in s32],
            "termination_fp64": [x["termination"] for x in s64], "termination_mixed": [x["termination"] for x in s32],
            "max_position_dev_m": dpos, "max_landmark_dev_m": dlm}
    big = [synthetic.config_A(seed=20240923 + i) for i in range(64)]
    t64, k64 = throughput(big, False)
    t32, k32 = throughput(big, True)
    out["throughput_64_windows"] = {"fp64_it_per_s": t64, "mixed_it_per_s": t32, "per_kernel_us_fp64": k64,
                                    "per_kernel_us_mixed": k32}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
