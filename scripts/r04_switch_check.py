#!/usr/bin/env python3
"""A batch of windows (CHECK_WINDOWS, default 48) through optimize() in both trust-region modes; the final states and summaries go to
an .npz so that two runs with a switch of the solver set / not set (an environment variable read once per process, e.g.
OKVIS_BA_NO_PRE) can be compared bit for bit:  r04_merge_check.py run out.npz | cmp a.npz b.npz"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(out):
    from okvis_amd import solver, synthetic
    from okvis_amd.window import default_options
    res = {}
    n = int(os.environ.get("CHECK_WINDOWS", "48"))
    wins = ([synthetic.make_window(10, 400, 1.0, 9_100_000 + i) for i in range(max(0, n - 8))] +
            [synthetic.make_window(6 + i % 3, 150 + 20 * i, 0.6, 9_200_000 + i) for i in range(8)])[-n:]
    if os.environ.get("CHECK_K"):   # windows of CHECK_K frames (CHECK_K - 1 IMU terms each)
        wins = [synthetic.make_window(int(os.environ["CHECK_K"]), 120 + 10 * i, 0.7, 9_300_000 + i) for i in range(n)]
    for mode in ("dogleg", "gn"):
        o = default_options()
        if mode == "gn":
            o.gauss_newton = 1
        b = solver.WindowBatch(wins, device=0, options=o)
        if os.environ.get("CHECK_SET_STATE"):   # the biases change between the upload and the first evaluation
            for w in range(n):
                sb = b.get_state(w)[1].copy()
                sb[:, 3:6] += 2.0e-3 * (1 + w % 3)
                sb[:, 6:9] -= 1.0e-2
                b.set_state(w, sb=sb)
        sm = b.optimize(12)
        st = [b.get_state(w) for w in range(n)]
        res[mode + "_cost"] = np.array([x["final_cost"] for x in sm])
        res[mode + "_iter"] = np.array([x["iterations"] for x in sm])
        res[mode + "_succ"] = np.array([x["successful_steps"] for x in sm])
        res[mode + "_pose"] = np.concatenate([x[0].reshape(-1) for x in st])
        res[mode + "_sb"] = np.concatenate([x[1].reshape(-1) for x in st])
        res[mode + "_lm"] = np.concatenate([x[2].reshape(-1) for x in st])
        if os.environ.get("CHECK_SET_STATE"):
            res[mode + "_redo"] = np.concatenate([np.asarray(b.array("IMU_REDO_COUNT", w)).reshape(-1) for w in range(n)])
        res[mode + "_timeouts"] = np.array([b.helper_timeouts() if hasattr(b, "helper_timeouts") else -1])
        b.close()
    np.savez(out, **res)
    print("saved", out, {k: (v.shape, float(np.sum(v))) for k, v in res.items() if "cost" in k or "timeouts" in k})


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = [k for k in A.files if not np.array_equal(A[k], B[k])]
    print("identical" if not bad else f"DIFFERENT: {bad}")
    for k in bad:
        print(k, np.abs(A[k] - B[k]).max())
    print("timeouts", A["dogleg_timeouts"], A["gn_timeouts"], B["dogleg_timeouts"], B["gn_timeouts"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[2]) if sys.argv[1] == "run" else cmp(sys.argv[2], sys.argv[3]))
