"""diagnostics (not a test): test_dogleg_batch_with_different_slot_counts's windows against the fp64 oracle and the long double referee"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
import make_golden as G
import oracle_lib
def opts():
    o = default_options(); o.initial_radius = 30.0
    return o
ws = [synthetic.small_window(**kw) for kw in G.SMALL] + [synthetic.small_window(seed=s, K=4, L=50) for s in (21, 22)]
b = solver.WindowBatch(ws, options=opts())
sg = b.optimize(7)
for i, w in enumerate(ws):
    a = oracle_lib.OracleWindow(w).optimize(7, opts())
    r = oracle_lib.OracleWindow(w, extended=True).optimize(7, opts())
    g = sg[i]
    print("window %d: cost gpu-oracle %.1e gpu-referee %.1e oracle-referee %.1e | radius gpu-oracle %.1e gpu-referee %.1e oracle-referee %.1e | steps %s %s %s" % (
        i, abs(g["final_cost"] - a["final_cost"]) / a["final_cost"], abs(g["final_cost"] - r["final_cost"]) / r["final_cost"], abs(a["final_cost"] - r["final_cost"]) / r["final_cost"],
        abs(g["final_radius"] - a["final_radius"]) / a["final_radius"], abs(g["final_radius"] - r["final_radius"]) / r["final_radius"], abs(a["final_radius"] - r["final_radius"]) / r["final_radius"],
        (g["iterations"], g["successful_steps"]), (a["iterations"], a["successful_steps"]), (r["iterations"], r["successful_steps"])))
b.close()
