"""diagnostics (not a test): where does the far-start DOGLEG case of test_dogleg_rejected_steps (seed 41) pick up its distance to the
oracle?  The referee is the oracle built in long double (oracle/liboracle_ld.so).  From states along the referee's trajectory, the
fp64 oracle and the GPU each linearise and solve the first DOGLEG system (mu = 1e-8); every intermediate array is compared with the
referee's, error relative to the array's largest entry.  Then the 20-iteration run itself: per-iteration cost of the three sides.

    python tools/gpu_referee_diag.py [--cpu] [seed]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import synthetic
from okvis_amd.window import STRATEGY_DOGLEG, default_options
from tests import oracle_lib

cpu_only = "--cpu" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seed = int(args[0]) if args else 41
NAMES = ("OBS_RESIDUAL", "IMU_RESIDUAL", "LM_V", "LM_B", "PAIR_W", "HPP", "GRADIENT", "REDUCED_S", "REDUCED_RHS", "STEP")


def opts(**kw):
    o = default_options(STRATEGY_DOGLEG)
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def err(a, ref):
    if a.shape != ref.shape or ref.size == 0:
        return float("nan")
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300)


w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
if not cpu_only:
    from okvis_amd import solver

print("seed %d: arrays of the first DOGLEG system from states along the referee's trajectory; error vs the long double build, relative "
      "to the largest entry" % seed)
for k in [int(x) for x in os.environ.get("REFEREE_K", "0,4,9,14,19").split(",")]:
    ld = oracle_lib.OracleWindow(w, extended=True)
    if k:
        ld.optimize(k, opts())
    X = ld.get_state()
    ref = oracle_lib.OracleWindow(w, extended=True)
    ref.set_state(*X); ref.linearize(); ref.solve(1e4, opts())
    o = oracle_lib.OracleWindow(w)
    o.set_state(*X); o.linearize(); o.solve(1e4, opts())
    row_o = {n: err(o.array(n), ref.array(n)) for n in NAMES}
    row_g = {}
    if not cpu_only:
        b = solver.WindowBatch([w], options=opts(debug_arrays=1, use_graph=0))
        b.set_state(0, *X)
        b.begin(); b.iterate(1)
        for n in NAMES:
            try:
                row_g[n] = err(b.array(n), ref.array(n))
            except Exception:   # an array this launch path does not keep
                row_g[n] = float("nan")
        b.close()
    print("after %2d iterations:" % k)
    for n in NAMES:
        print("   %-14s oracle fp64 %.1e%s" % (n, row_o[n], ("    GPU %.1e" % row_g[n]) if row_g else ""))

print("per-iteration cost, relative distance to the referee (fp64 oracle%s)" % ("" if cpu_only else " / GPU"))
for n in (1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20):
    cr = oracle_lib.OracleWindow(w, extended=True).optimize(n, opts())
    co = oracle_lib.OracleWindow(w).optimize(n, opts())
    line = "   %2d iterations (%2d accepted): cost %.9f   oracle fp64 %.1e" % (n, cr["successful_steps"], cr["final_cost"],
                                                                             abs(co["final_cost"] - cr["final_cost"]) / cr["final_cost"])
    if not cpu_only:
        b = solver.WindowBatch([w], options=opts())
        cg = b.optimize(n)[0]
        b.close()
        line += "    GPU %.1e%s" % (abs(cg["final_cost"] - cr["final_cost"]) / cr["final_cost"],
                                    "" if cg["successful_steps"] == cr["successful_steps"] else " (steps differ)")
    print(line)
