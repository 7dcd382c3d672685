"""diagnostics (not a test): okvis_ba_marginalize and the fp64 oracle's MarginalizationError against the oracle in long double:
H, b0, J^T J, J^T e0 relative to the largest entry.    python tools/gpu_referee_marg.py [--cpu]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import synthetic
from okvis_amd.window import default_options
from tests import oracle_lib
cpu_only = "--cpu" in sys.argv
if not cpu_only:
    from okvis_amd import solver

def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.shape == b.shape else float("nan")

def flags(w, poses=(), sbs=()):
    pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8)
    pm[list(poses)] = 1; sm[list(sbs)] = 1
    return pm, sm

def row(x, r):
    return "H %.1e  b0 %.1e  J^T J %.1e  J^T e0 %.1e  rank %d/%d" % (rel(x["H"], r["H"]), rel(x["b0"], r["b0"]), rel(x["J"].T @ x["J"], r["J"].T @ r["J"]),
                                                                      rel(x["J"].T @ x["e0"], r["J"].T @ r["e0"]), x["rank"], r["rank"])

cases = [("seed 41 K=5 L=40, pose 0 + sb 0,1", synthetic.small_window(seed=41, K=5, L=40), ([0], [0, 1])),
         ("seed 42 K=4 L=30, landmarks only", synthetic.small_window(seed=42, K=4, L=30), ([], [])),
         ("config A, poses 0,1 + sb 0..4", synthetic.config_A(), ([0, 1], [0, 1, 2, 3, 4])),
         ("D = 300, poses 0,1 + sb 0,1", synthetic.make_window(20, 30, 1.0, 2, frame_dt=0.1), ([0, 1], [0, 1]))]
for name, w, (p, s) in cases:
    pm, sm = flags(w, p, s)
    r = oracle_lib.OracleWindow(w, extended=True).marginalize(pm, sm)
    o = oracle_lib.OracleWindow(w).marginalize(pm, sm)
    print(name)
    print("   oracle fp64:", row(o, r))
    if not cpu_only:
        b = solver.WindowBatch([w], options=default_options())
        g = b.marginalize(0, pm, sm)
        b.close()
        print("   GPU        :", row(g, r))
    # second stage: the prior of the first stage rides into a second marginalisation where the structure allows it
