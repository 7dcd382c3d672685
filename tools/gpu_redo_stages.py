"""Diagnostic: cycles per stage of the IMU re-preintegration (factor 0 of one configs[1] window; debug_arrays)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.use_graph = 0; opt.debug_arrays = 1
b = solver.WindowBatch([synthetic.config_A()], options=opt)
b.begin(); b.synchronize()
p = b.array("PROF")
names = ["0 loop control", "1 per-step quantities", "2 Delta_q chain + cross recursion", "3 rotations, integrals", "4 prefix sums",
         "5 per-step blocks of F", "6 dv/db_g prefix + dp_term", "8 covariance recursion", "7 totals + carries", "(after last chunk)", "9 inverse + Cholesky + store"]
tot = 0
for k, n in enumerate(names):
    tot += p[50 + k]; print(f"  {n:36s} {p[50+k]:10.0f} cyc {p[50+k]/2100:8.2f} us")
print("  total", tot / 2100, "us")
b.finish(); b.close()
