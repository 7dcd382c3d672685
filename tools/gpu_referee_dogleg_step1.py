"""diagnostics (not a test): the first (radius-limited) DOGLEG step of small[2] at radius 30, component by component against the referee"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, default_options
from tests import oracle_lib
np.set_printoptions(linewidth=220, precision=3)
case = int(sys.argv[1]) if len(sys.argv) > 1 else 2
radius = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
w = synthetic.small_window(**G.SMALL[case])
def opts(**kw):
    o = default_options(STRATEGY_DOGLEG); o.initial_radius = radius
    o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0
    for k, v in kw.items(): setattr(o, k, v)
    return o
x0 = (w.pose.copy(), w.sb.copy(), w.lm.copy())
b = solver.WindowBatch([w], options=opts(debug_arrays=1, use_graph=0)); g = b.optimize(1)[0]; xg = b.get_state()
gstep, ggrad = b.array("STEP"), b.array("GRADIENT"); b.close()
r = oracle_lib.OracleWindow(w, extended=True); sr = r.optimize(1, opts()); xr = r.get_state()
a = oracle_lib.OracleWindow(w); sa = a.optimize(1, opts()); xa = a.get_state()
print("accepted", g["successful_steps"], sr["successful_steps"], "cost", g["final_cost"], sr["final_cost"])
for name, i in (("pose", 0), ("sb", 1), ("lm", 2)):
    dg = xg[i] - x0[i][:xg[i].shape[0]] if x0[i].shape == xg[i].shape else None
    print(name, "GPU-referee max %.2e  oracle-referee max %.2e   |step| max %.2e" % (np.abs(xg[i] - xr[i]).max(), np.abs(xa[i] - xr[i]).max(), np.abs(xr[i] - x0[i]).max()))
print("sb GPU - referee per frame:"); print(xg[1] - xr[1])
print("sb step (referee):"); print(xr[1] - x0[1])
ratio = (xg[1] - x0[1]) / np.where(np.abs(xr[1] - x0[1]) > 0, xr[1] - x0[1], 1.0)
print("sb step ratio GPU / referee - 1:"); print(ratio - 1.0)
ref0 = oracle_lib.OracleWindow(w, extended=True); ref0.linearize(); ref0.solve(1e4, opts())
rs, rg = ref0.array("STEP"), ref0.array("GRADIENT")
D = rs.size
print("dGN (stored Gauss-Newton point) GPU vs referee, relative to max: %.2e ; per entry abs diff / |entry| (sb part):" % (np.abs(gstep - rs).max() / np.abs(rs).max()))
Dp = D - 9 * w.n_sb
print(((gstep - rs) / np.where(rs != 0, rs, 1.0))[Dp:].reshape(-1, 9))
print("gradient GPU vs referee rel to max %.2e" % (np.abs(ggrad - rg).max() / np.abs(rg).max()))
