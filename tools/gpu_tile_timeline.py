"""diagnostics (not a test): timeline of the tiled Cholesky's tasks for BASELINE configs[2] (wall_clock64, 100 MHz)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
opt.use_graph = 0; opt.debug_arrays = 2
w = synthetic.config_C()
b = solver.WindowBatch([w], options=opt)
b.begin(); b.iterate(4); b.synchronize()
p = b.array("PROF")[64:].reshape(-1, 4) / 100.0   # us
nT = 16
t0 = p[0, 1]
idx = lambda i, j: sum(nT - c for c in range(j)) + (i - j)
print("diag tile j: start | deps met (last update applied) | factor+inverse done | published     [us from kernel start]")
prev = 0
for j in range(nT):
    d = p[idx(j, j)] - t0
    line = f"  j={j:2d}  {d[0]:8.1f} {d[1]:8.1f} {d[2]:8.1f} {d[3]:8.1f}   factor {d[2]-d[1]:6.1f}  publish {d[3]-d[2]:5.1f}  since previous diagonal {d[3]-prev:6.1f}"
    prev = d[3]
    print(line)
print("factorisation span", p[:136, 3].max() - t0)
print("x_j published at:", " ".join(f"{v - t0:.1f}" for v in p[136:152, 3]))
