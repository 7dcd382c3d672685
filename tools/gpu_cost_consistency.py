"""diagnostics (not a test): is the cost a radius-limited DOGLEG run reports the cost of the state it returns?  The GPU's summary
against (a) a fresh evaluation by the GPU at the returned state, (b) the fp64 oracle's and the long double referee's cost AT THAT STATE"""
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
for n in range(1, 7):
    b = solver.WindowBatch([w], options=opts()); g = b.optimize(n)[0]; xg = b.get_state()
    fresh = b.evaluate_cost()[0] if hasattr(b, "evaluate_cost") else float("nan")
    b.close()
    a = oracle_lib.OracleWindow(w); a.set_state(*xg); ca = a.cost()
    r = oracle_lib.OracleWindow(w, extended=True); r.set_state(*xg); cr = r.cost()
    print("%d it: summary %.9f  fresh GPU evaluation %s  oracle at that state %.9f  referee at that state %.9f | summary vs referee-at-state %.1e, fresh vs referee-at-state %.1e" % (
        n, g["final_cost"], fresh, ca, cr, abs(g["final_cost"] - cr) / cr, abs((fresh if isinstance(fresh, float) else float(fresh)) - cr) / cr))
print("---- the same with the oracle's preintegration kept at the START bias (as inside a run)")
for n in range(1, 4):
    b = solver.WindowBatch([w], options=opts()); g = b.optimize(n)[0]; xg = b.get_state()
    ref_g = b.array("IMU_SB_REF") if False else None
    b.close()
    r = oracle_lib.OracleWindow(w, extended=True); r.linearize(); r.set_state(*xg); cr = r.cost()
    r2 = oracle_lib.OracleWindow(w, extended=True); r2.set_state(*xg); cr2 = r2.cost()
    rr = oracle_lib.OracleWindow(w, extended=True); sr = rr.optimize(n, opts())
    print("%d it: GPU summary %.9f | referee at GPU's state, preintegration at the start bias %.9f, at the state's bias %.9f | referee's own run %.9f; IMU reference biases after the referee's run differ from the start bias: %s" % (
        n, g["final_cost"], cr, cr2, sr["final_cost"], np.abs(rr.array("IMU_SB_REF").reshape(-1, 9) - w.sb[:-1]).max()))
