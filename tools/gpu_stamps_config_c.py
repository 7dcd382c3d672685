"""diagnostics (not a test): phase stamps of solve_kernel<true> for BASELINE configs[2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
opt.use_graph = 0; opt.debug_arrays = 2
b = solver.WindowBatch([synthetic.config_C()], options=opt)
b.begin(); b.iterate(6); b.synchronize()
p = b.array("PROF")
names = {1: "decision + schur-partial sums", 2: "-", 5: "imu + priors + marg assembly", 6: "damping", 9: "end"}
prev = 0
for k in (1, 2, 5, 6):
    print(f"  {names[k]:32s} {(p[k] - p[prev]) / 2100:8.2f} us"); prev = k
print(f"  of the assembly: IMU factors {(p[58] - p[2]) / 2100:.2f} us, priors and the rest {(p[5] - p[58]) / 2100:.2f} us")
