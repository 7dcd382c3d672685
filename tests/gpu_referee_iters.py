"""diagnostics (not a test): per-iteration distance of the fp64 oracle and the GPU to the long double referee on a perturbed far-start
DOGLEG case (perturbation index as in gpu_referee_spread.py)."""
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
pert = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
rng = np.random.default_rng(0)
lm2 = w.lm.copy()
for t in range(1, pert + 1):
    lm2 = w.lm.copy()
    lm2[:, :3] = np.nextafter(w.lm[:, :3], w.lm[:, :3] + rng.choice([-1.0, 1.0], size=w.lm[:, :3].shape))
for n in range(1, 21):
    r = oracle_lib.OracleWindow(w, extended=True); r.set_state(lm=lm2); ref = r.optimize(n, opts())
    o = oracle_lib.OracleWindow(w); o.set_state(lm=lm2); so = o.optimize(n, opts())
    b = solver.WindowBatch([w], options=opts()); b.set_state(0, lm=lm2); sg = b.optimize(n)[0]
    xr, xo, xg = r.get_state(), o.get_state(), b.get_state()
    b.close()
    print("%2d it (%2d acc) cost %.9f radius %.6g | oracle: cost %.1e radius %.1e pose %.1e lm %.1e | GPU: cost %.1e radius %.1e pose %.1e lm %.1e%s" % (
        n, ref["successful_steps"], ref["final_cost"], ref["final_radius"],
        abs(so["final_cost"] - ref["final_cost"]) / ref["final_cost"], abs(so["final_radius"] - ref["final_radius"]) / ref["final_radius"],
        np.abs(xo[0] - xr[0]).max(), np.abs(xo[2] - xr[2]).max(),
        abs(sg["final_cost"] - ref["final_cost"]) / ref["final_cost"], abs(sg["final_radius"] - ref["final_radius"]) / ref["final_radius"],
        np.abs(xg[0] - xr[0]).max(), np.abs(xg[2] - xr[2]).max(),
        "" if sg["successful_steps"] == ref["successful_steps"] else " (steps differ)"))
