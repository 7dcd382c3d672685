"""diagnostics (not a test): the cases of tests/test_gpu_dogleg.py and test_gpu_structure_paths.py that run at north_star's 1e-6 —
how far the GPU and the fp64 oracle are from each other and from the long double referee (cost, relative; poses, absolute)"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G
from okvis_amd import solver, synthetic
from okvis_amd.window import STRATEGY_DOGLEG, STRATEGY_LM, default_options
from tests import oracle_lib

def opts(strategy=STRATEGY_DOGLEG, **kw):
    o = default_options(strategy)
    for k, v in kw.items(): setattr(o, k, v)
    return o

def run(name, w, n, o):
    b = solver.WindowBatch([w], options=o); g = b.optimize(n)[0]; xg = b.get_state(); b.close()
    a = oracle_lib.OracleWindow(w); sa = a.optimize(n, o); xa = a.get_state()
    r = oracle_lib.OracleWindow(w, extended=True); sr = r.optimize(n, o); xr = r.get_state()
    c = sr["final_cost"]
    same = (g["iterations"], g["successful_steps"], g["termination"]) == (sa["iterations"], sa["successful_steps"], sa["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"])
    print("%-44s cost GPU-oracle %.1e | vs referee GPU %.1e oracle %.1e | pose GPU-oracle %.1e GPU-ref %.1e lm GPU-oracle %.1e%s" % (
        name, abs(g["final_cost"] - sa["final_cost"]) / c, abs(g["final_cost"] - c) / c, abs(sa["final_cost"] - c) / c,
        np.abs(xg[0] - xa[0]).max(), np.abs(xg[0] - xr[0]).max(), np.abs(xg[2] - xa[2]).max(), "" if same else "  BOOKKEEPING DIFFERS"), flush=True)

for case in range(len(G.SMALL)):
    w = synthetic.small_window(**G.SMALL[case])
    for radius in (1e4, 30.0, 1.0):
        run("dogleg small[%d] radius %g" % (case, radius), w, 10, opts(initial_radius=radius))
run("dogleg no jacobi scaling small[1] r=100", synthetic.small_window(**G.SMALL[1]), 8, opts(jacobi_scaling=0, initial_radius=100.0))
wA = synthetic.config_A()
run("dogleg config A", wA, 10, opts())
run("dogleg config A r=50", wA, 6, opts(initial_radius=50.0))
wL = synthetic.make_window(20, 200, 1.0, seed=33, frame_dt=0.1)
run("dogleg D=300", wL, 6, opts())
run("dogleg D=300 r=40", wL, 6, opts(initial_radius=40.0))
for seed in (41, 42, 43, 44):
    w = synthetic.small_window(seed=seed, K=5, L=60, pose_noise=(0.4, np.deg2rad(6.0)), landmark_noise=0.8)
    run("LM r=1e8 far start seed %d, 25 it" % seed, w, 25, opts(STRATEGY_LM, initial_radius=1e8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
    run("dogleg far start seed %d, 20 it" % seed, w, 20, opts(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
