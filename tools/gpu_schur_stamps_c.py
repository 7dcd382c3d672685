"""Diagnostic: clock64() stamps of Schur workgroup 0 on BASELINE configs[2] (50 keyframes, several 96-row tiles per dimension)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0; opt.use_graph = 0; opt.debug_arrays = 2
b = solver.WindowBatch([synthetic.config_C(seed=20240923)], options=opt)
b.begin(); b.iterate(6); b.synchronize()
p = b.array("PROF")
sn = {17: "prologue + decision", 18: "operands requested, V^-1, first zero + barrier", 19: "batch 0: fill", 20: "batch 0: products",
      21: "remaining batches", 22: "accumulators out", 23: "diagonal blocks / rows"}
for k in range(17, 24):
    print(f"  {sn[k]:48s} {(p[k]-p[k-1])/2100:8.2f} us")
print("  total", (p[23]-p[16])/2100, "us")
