#!/usr/bin/env python3
"""The reference's own okvis_ceres/test/TestEstimator.cpp scenario over several std::srand seeds, on either side:
  oracle/_ref/reference_test_estimator      — compiled against the drop-in (GPU backend)
  oracle/_ref/reference_test_estimator_ref  — compiled against the reference's own okvis::Estimator (CPU, ceres stand-in)
Prints one JSON object: per seed and extrinsics case the three quantities the test asserts on (TestEstimator.cpp:229-236:
speed/bias error norm < 0.04, rotation error < 1e-2, translation error < 0.1 m).
usage: test_estimator_seeds.py <binary> <seed> [<seed> ...]   (seeds run as parallel processes)"""
import json
import os
import re
import subprocess
import sys

import numpy as np


def parse(text):
    cases = []
    blocks = text.split("== LAST OPTIMIZATION ==")[1:]
    for b in blocks:
        m = re.search(r"estimated T_WS:\s*\n((?:[^\n]+\n){4})correct T_WS:\s*\n((?:[^\n]+\n){4})\s*([-+0-9.eE]+)", b)
        if not m:
            continue
        est = np.array([[float(x) for x in ln.split()] for ln in m.group(1).strip().splitlines()])
        cor = np.array([[float(x) for x in ln.split()] for ln in m.group(2).strip().splitlines()])
        dR = cor[:3, :3] @ est[:3, :3].T
        # 2 |vec(q_err)| as the test computes it = 2 sin(angle / 2); from the skew part of the printed matrices (six digits each:
        # the trace would lose a small angle), sin(angle) = |vee(dR - dR^T)| / 2
        sin_a = 0.5 * np.linalg.norm([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
        ang = np.arcsin(min(1.0, sin_a))
        cases.append({"translation_error_m": float(np.linalg.norm(cor[:3, 3] - est[:3, 3])), "rotation_error": float(2.0 * np.sin(ang / 2.0)),
                      "speed_bias_error": float(m.group(3))})
    return cases


def main():
    exe, seeds = sys.argv[1], [int(s) for s in sys.argv[2:]]
    par = int(os.environ.get("SEEDS_PARALLEL", "8"))
    out = {"binary": os.path.basename(exe), "bounds": {"translation_error_m": 0.1, "rotation_error": 1e-2, "speed_bias_error": 0.04}, "seeds": {}}
    for i in range(0, len(seeds), par):
        procs = {s: subprocess.Popen([exe], env=dict(os.environ, OKVIS_TEST_SEED=str(s)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for s in seeds[i:i + par]}
        for s, p in procs.items():
            text = p.communicate()[0]
            if os.environ.get("SEEDS_RAW_DIR"):   # keep the raw output next to the summary
                os.makedirs(os.environ["SEEDS_RAW_DIR"], exist_ok=True)
                open(os.path.join(os.environ["SEEDS_RAW_DIR"], f"{os.path.basename(exe)}_seed{s}.txt"), "w").write(text)
            out["seeds"][str(s)] = {"exit_code": p.returncode, "cases": parse(text)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
