"""diagnostics (not a test): radius-limited DOGLEG runs, iteration by iteration, GPU and fp64 oracle against the long double referee"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, default_options
from tests import oracle_lib
case = int(sys.argv[1]) if len(sys.argv) > 1 else 2
radius = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
w = synthetic.small_window(**G.SMALL[case])
def opts():
    o = default_options(STRATEGY_DOGLEG); o.initial_radius = radius
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    return o
print("small[%d] radius %g" % (case, radius))
for n in range(1, 9):
    b = solver.WindowBatch([w], options=opts()); g = b.optimize(n)[0]; xg = b.get_state(); b.close()
    a = oracle_lib.OracleWindow(w); sa = a.optimize(n, opts()); xa = a.get_state()
    r = oracle_lib.OracleWindow(w, extended=True); sr = r.optimize(n, opts()); xr = r.get_state()
    c = sr["final_cost"]
    print("%d it (%d acc) cost %.6f radius %.6g | GPU: cost %.1e radius %.1e pose %.1e lm %.1e | oracle: cost %.1e radius %.1e pose %.1e lm %.1e" % (
        n, sr["successful_steps"], c, sr["final_radius"],
        abs(g["final_cost"] - c) / c, abs(g["final_radius"] - sr["final_radius"]) / sr["final_radius"], np.abs(xg[0] - xr[0]).max(), np.abs(xg[2] - xr[2]).max(),
        abs(sa["final_cost"] - c) / c, abs(sa["final_radius"] - sr["final_radius"]) / sr["final_radius"], np.abs(xa[0] - xr[0]).max(), np.abs(xa[2] - xr[2]).max()))
