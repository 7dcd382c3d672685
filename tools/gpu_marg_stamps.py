"""Diagnostic: clock stamps inside marg_dense_kernel (debug_arrays) for a gauge-deficient and a full-rank sub-window."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
names = {1: "previous prior + index lists", 2: "eliminated block: scaling + Cholesky/inverse", 3: "M, b0, H (Schur complement)",
         4: "kept block: scaling + copy", 5: "plain Cholesky attempt", 6: "copy for pivoted Cholesky", 7: "pivoted Cholesky + bounds", 8: "J, e0"}
for label, noprior in (("gauge-deficient kept block", True), ("full-rank kept block", False)):
    w = synthetic.small_window(seed=51, K=5, L=40)
    if noprior:
        w.pprior_pose = np.zeros(0, np.int32); w.pprior_meas = np.zeros((0, 7)); w.pprior_sqrtinfo = np.zeros((0, 36))
    opt = default_options(); opt.debug_arrays = 1; opt.use_graph = 0
    b = solver.WindowBatch([w], options=opt)
    pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8); pm[0] = 1; sm[0] = 1
    b.marginalize(0, pm, sm); g = b.marginalize(0, pm, sm)
    p = b.array("PROF")
    print(label, "dim", g["dim"], "rank", g["rank"], "sweeps", list(g["sweeps"]))
    prev = 0
    for k in range(1, 9):
        if p[k] > p[prev]:
            print(f"  {names[k]:44s} {(p[k]-p[prev])/2100:8.2f} us"); prev = k
    print("  total", (p[prev] - p[0]) / 2100, "us")
    if p[12] > p[6]:
        print("  inside the pivoted path: bracket %.2f, elimination %.2f, gather %.2f, X11 + L21 X11 %.2f, bounds %.2f us" % tuple((p[k] - p[q]) / 2100 for k, q in ((9, 6), (10, 9), (11, 10), (12, 11), (7, 12))))
    b.close()
