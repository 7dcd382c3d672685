"""diagnostics (not a test): the far-start DOGLEG case iteration by iteration — cost, radius and the IMU terms' reference biases of
the GPU and the fp64 oracle against the long double referee.    python tools/gpu_referee_iters.py [seed]"""
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
w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
for n in range(1, 21):
    r = oracle_lib.OracleWindow(w, extended=True); ref = r.optimize(n, opts())
    o = oracle_lib.OracleWindow(w); so = o.optimize(n, opts())
    b = solver.WindowBatch([w], options=opts()); sg = b.optimize(n)[0]
    xr, xg = r.get_state(), b.get_state()
    bg, br, bo = b.array("IMU_SB_REF"), r.array("IMU_SB_REF"), o.array("IMU_SB_REF")
    b.close()
    print("%2d it (%2d acc) cost %.9f radius %.6g | GPU: cost %.1e radius %.1e pose %.1e sb %.1e lm %.1e imu ref %.1e | oracle: cost %.1e imu ref %.1e%s" % (
        n, ref["successful_steps"], ref["final_cost"], ref["final_radius"],
        abs(sg["final_cost"] - ref["final_cost"]) / ref["final_cost"], abs(sg["final_radius"] - ref["final_radius"]) / ref["final_radius"],
        np.abs(xg[0] - xr[0]).max(), np.abs(xg[1] - xr[1]).max(), np.abs(xg[2] - xr[2]).max(), np.abs(bg - br).max(),
        abs(so["final_cost"] - ref["final_cost"]) / ref["final_cost"], np.abs(bo - br).max(),
        "" if sg["successful_steps"] == ref["successful_steps"] else " (steps differ)"))
