"""diagnostics (not a test): spread of the 20-iteration far-start DOGLEG cost (seed 41) under 1-ulp perturbations of the landmark
start values, each run against the long double referee of the SAME perturbed input: fp64 oracle and GPU (landmarks per linearise
group 16 = default for one window, 64, 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, default_options
from tests import oracle_lib

def opts():
    o = default_options(STRATEGY_DOGLEG)
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    return o

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 41
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
rng = np.random.default_rng(0)
rows = {"oracle fp64": [], "GPU 16": [], "GPU 64": [], "GPU 8": []}
for t in range(12):
    lm2 = w.lm.copy()
    if t:
        lm2[:, :3] = np.nextafter(w.lm[:, :3], w.lm[:, :3] + rng.choice([-1.0, 1.0], size=w.lm[:, :3].shape))
    r = oracle_lib.OracleWindow(w, extended=True); r.set_state(lm=lm2); ref = r.optimize(n_it, opts())
    o = oracle_lib.OracleWindow(w); o.set_state(lm=lm2)
    rows["oracle fp64"].append((o.optimize(n_it, opts())["final_cost"] - ref["final_cost"]) / ref["final_cost"])
    for cap in (16, 64, 8):
        og = opts()
        og.tuning.group_lm = cap
        b = solver.WindowBatch([w], options=og)
        b.set_state(0, lm=lm2)
        s = b.optimize(n_it)[0]
        b.close()
        rows["GPU %d" % cap].append((s["final_cost"] - ref["final_cost"]) / ref["final_cost"] if s["successful_steps"] == ref["successful_steps"] else float("nan"))
for k, v in rows.items():
    print("%-12s median |d| %.1e  max %.1e   %s" % (k, np.nanmedian(np.abs(v)), np.nanmax(np.abs(v)), " ".join("%+.1e" % x for x in v)))
