"""diagnostics (not a test): windows above the LDS solver's size (tiled solver, D = 300) against the long double referee.  The tiled
solver is not compensated (ba_chol_tiles.hpp says why; an experiment with it changed the last digits only), so the two GPU columns —
with and without OKVIS_BA_TUNE_NO_LDL_COMP — agree by construction."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, STRATEGY_LM, default_options
from tests import oracle_lib

def opts(strategy, **kw):
    o = default_options(strategy)
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    for k, v in kw.items(): setattr(o, k, v)
    return o

cases = [("K=20 L=200 seed 33", synthetic.make_window(20, 200, 1.0, seed=33, frame_dt=0.1)),
         ("K=20 L=200 seed 34 far", synthetic.make_window(20, 200, 1.0, seed=34, frame_dt=0.1, pose_noise=(0.3, np.deg2rad(4.0)), landmark_noise=0.5))]
for name, w in cases:
    for sname, o in (("dogleg r=40", opts(STRATEGY_DOGLEG, initial_radius=40.0)), ("dogleg", opts(STRATEGY_DOGLEG)), ("lm", opts(STRATEGY_LM))):
        t0 = time.time()
        r = oracle_lib.OracleWindow(w, extended=True).optimize(8, o)
        a = oracle_lib.OracleWindow(w).optimize(8, o)
        row = "%-24s %-12s D %d  cost %.6f (%d acc)  oracle %.1e" % (name, sname, w.reduced_dim(), r["final_cost"], r["successful_steps"], abs(a["final_cost"] - r["final_cost"]) / r["final_cost"])
        for off in (0, 1):
            o.tuning.flags = 0x8 if off else 0   # OKVIS_BA_TUNE_NO_LDL_COMP
            b = solver.WindowBatch([w], options=o)
            g = b.optimize(8)[0]
            b.close()
            row += "   GPU%s %.1e%s" % (" (no comp)" if off else "", abs(g["final_cost"] - r["final_cost"]) / r["final_cost"], "" if g["successful_steps"] == r["successful_steps"] else " (steps differ)")
        o.tuning.flags = 0
        print(row, " [%.0f s]" % (time.time() - t0), flush=True)
