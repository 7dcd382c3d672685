"""diagnostics (not a test): the far-start DOGLEG cases of test_dogleg_rejected_steps against the oracle, after 20 iterations
(not converged) and at convergence, for several landmark-per-group limits of the index build (okvis_ba_tuning::group_lm)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, default_options
from tests import oracle_lib
for seed in (41, 42, 43, 44):
    w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
    for n in (20, 40, 80):
        o = default_options(STRATEGY_DOGLEG); o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
        sr = oracle_lib.OracleWindow(w).optimize(n, o)
        row = []
        for cap in (64, 32, 24, 16, 8):
            o.tuning.group_lm = cap
            b = solver.WindowBatch([w], options=o)
            sg = b.optimize(n)[0]
            b.close()
            same = (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"])
            row.append("%d: %.1e%s" % (cap, abs(sg["final_cost"] - sr["final_cost"]) / sr["final_cost"], "" if same else " (steps differ)"))
        print("seed %d, %2d iterations (%d accepted), cost %.6f:  %s" % (seed, n, sr["successful_steps"], sr["final_cost"], "   ".join(row)))
