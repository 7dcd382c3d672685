#!/usr/bin/env python3
"""Timing of okvis_ba_marginalize next to the CPU oracle on the same sub-window.
Prints one JSON line.  Two shapes: what the stock pipeline hands over per frame (6 frames, 150 landmarks,
~1.4 k observations, one pose + two speed/bias blocks eliminated) and a BASELINE-configs[1]-sized window.

Measurement script, not product code: the oracle is called in the same role as in bench.py's `cpu_baseline` leg (a
reported CPU column and the H-parity figure), never part of the measured GPU path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis_amd import solver, synthetic  # noqa: E402
from okvis_amd.window import default_options  # noqa: E402
from tests import oracle_lib  # noqa: E402  (test infrastructure: CPU baseline only)


def run(name, w, poses, sbs, reps=20, with_oracle=True):
    pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8)
    pm[poses] = 1; sm[sbs] = 1
    b = solver.WindowBatch([w], options=default_options())
    g = b.marginalize(0, pm, sm)          # warm-up (module load, allocations)
    b.synchronize()
    per_call = []
    for _ in range(reps):
        t0 = time.perf_counter()
        g = b.marginalize(0, pm, sm)
        per_call.append(time.perf_counter() - t0)
    t_gpu = float(np.median(per_call))      # (the first calls after the warm-up still pay clock ramp-up)
    b.close()
    if not with_oracle:   # (the dense restatement of a configs[2]-sized window would take minutes): self-consistency instead
        jtj = g["J"].T @ g["J"]
        return {"case": name, "observations": int(w.n_obs), "landmarks": int(w.n_lm), "reduced_dim": int(w.reduced_dim()),
                "prior_dim": int(g["dim"]), "rank": int(g["rank"]), "jacobi_sweeps": list(g["sweeps"]), "gpu_ms_per_call": t_gpu * 1e3,
                "gpu_ms_min_max": [min(per_call) * 1e3, max(per_call) * 1e3], "cpu_oracle_ms_per_call": None,
                "JtJ_vs_H_rel_diff": float(np.abs(jtj - g["H"]).max() / np.abs(g["H"]).max()),
                "H_asymmetry": float(np.abs(g["H"] - g["H"].T).max() / np.abs(g["H"]).max())}
    o = oracle_lib.OracleWindow(w)
    t0 = time.perf_counter()
    n_cpu = 3
    for _ in range(n_cpu):
        r = oracle_lib.OracleWindow(w).marginalize(pm, sm)
    t_cpu = (time.perf_counter() - t0) / n_cpu
    err = np.abs(g["H"] - r["H"]).max() / np.abs(r["H"]).max()
    return {"case": name, "observations": int(w.n_obs), "landmarks": int(w.n_lm), "reduced_dim": int(o.D),
            "prior_dim": int(g["dim"]), "jacobi_sweeps": list(g["sweeps"]), "gpu_ms_per_call": t_gpu * 1e3, "gpu_ms_min_max": [min(per_call) * 1e3, max(per_call) * 1e3], "cpu_oracle_ms_per_call": t_cpu * 1e3,
            "H_rel_diff": float(err)}


def main():
    out = [run("pipeline-like", synthetic.small_window(seed=9, K=6, L=150, visibility=0.8), [0], [0, 1]),
           run("configs[1]-sized", synthetic.config_A(), [0, 1], [0, 1, 2, 3, 4], reps=10),
           run("20 frames, D = 300 (HBM workspace)", synthetic.make_window(20, 200, 1.0, seed=33, frame_dt=0.1), [0, 1], [0, 1], reps=5)]
    if "--config-c" in sys.argv:
        out.append(run("configs[2]-sized, D = 750 (HBM workspace)", synthetic.config_C(), [0, 1], [0, 1], reps=3, with_oracle=False))
    print(json.dumps({"marginalize": out, "note": "gpu = whole okvis_ba_marginalize call (linearise + landmark Schur + "
                      "dense elimination + the two decompositions (Cholesky / pivoted Cholesky / Jacobi, see jacobi_sweeps) + download), window already uploaded; "
                      "cpu = oracle restatement on the full dense matrix, 1 thread, not Eigen"}))


if __name__ == "__main__":
    main()
