"""Diagnostic (not a test): clock64() phase stamps of linearise group 0 of window 0, alone and under load.
usage: python tools/gpu_lin_stamps.py [n_windows] [reserved0]   (reserved0 bit 3 = staged kernel, bit 2 = no fused launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0; opt.use_graph = 0; opt.debug_arrays = 2
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt.reserved0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = solver.WindowBatch([synthetic.config_A(seed=20240923 + i) for i in range(NW)], options=opt)
b.begin(); b.iterate(12); b.synchronize()
p = b.array("PROF")
ln = {41: "entry: records + index lists requested", 42: "phase A (back-substitution) + park", 43: "phase B residual/Jacobian/products",
      44: "merge + piece records | stage + barrier", 45: "C(a) per landmark", 46: "C(b) per pair", 47: "C(c) per block", 48: "barrier", 49: "-",
      50: "fused reduction", 51: "scalars"}
print(f"linearise phases of group 0 (thread 0), {NW} windows, reserved0 {opt.reserved0}:")
for k in range(41, 52):
    d = p[k] - p[k - 1]; print(f"  {ln[k]:44s} {d:10.0f} cyc {d/2100:8.2f} us")
print("  total", (p[51] - p[40]) / 2100, "us")
