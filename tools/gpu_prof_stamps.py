import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0; opt.use_graph = 0; opt.debug_arrays = 2
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt.reserved0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # reserved0: 1 = level-scheduled speed/bias elimination forced on, 2 = forced off, 0 = auto   # stamps are taken by window 0; NW > 1 shows them under load
b = solver.WindowBatch([synthetic.config_A(seed=20240923 + i) for i in range(NW)], options=opt)
b.begin(); b.iterate(12); b.synchronize()
p = b.array("PROF")
names = {0:"start",1:"decision+spart",2:"-",5:"imu+priors+marg",6:"damping",7:"cholesky",8:"backsub",9:"end"}
print("solve: wave 0 decision done at", (p[3]-p[0])/2100, "us; wave 1 shadow assembly done at", (p[4]-p[0])/2100, "us")
print("solve phases (cycles, us@2.1GHz):")
prev = 0
for k in (1, 2, 5, 6, 7, 8, 9):
    d = p[k]-p[prev]; prev = k; print(f"  {names[k]:16s} {d:10.0f} cyc  {d/2100:8.2f} us")
print("  total", p[9]-p[0], (p[9]-p[0])/2100)
sn = {17: "decision", 18: "batch 0: zero tables + V^-1", 19: "batch 0: fill Y / W tables", 20: "batch 0: block products",
      21: "remaining batches", 22: "U_pp / cross lists", 23: "slice reduction + write"}
print("schur phases (workgroup 0, thread 0):")
for k in range(17, 24):
    d = p[k]-p[k-1]; print(f"  {sn[k]:30s} {d:10.0f} cyc {d/2100:8.2f} us")
print("  total", (p[23]-p[16])/2100, "us")
print("sb levels (us): level 0 factor", (p[53]-p[52])/2100, " Y", (p[54]-p[53])/2100, " update", (p[55]-p[54])/2100, " all levels", (p[56]-p[52])/2100, " dense chol", (p[7]-p[56])/2100, " dense backsub", (p[57]-p[7])/2100, " sb recovery", (p[8]-p[57])/2100)
print("kb=0: panel", p[11]-p[10], "trailing(thread0 work)", p[26]-p[11], "trailing+barrier", p[12]-p[11])
print("kb=12: panel", p[14]-p[13], "trailing+barrier", p[15]-p[14])

print("kb=12 look-ahead (cycles from phase start): entries ready", p[30]-p[14], " broadcast", p[31]-p[14], " chol6", p[34]-p[14], " trinv6", p[35]-p[14], " factor done", p[32]-p[14], " rhs wave done", p[33]-p[14], " barrier released", p[15]-p[14])

ln = {41: "wins/ctrl/group loads", 42: "phase A (back-substitution)", 43: "phase B loads + residual/Jacobian", 44: "stage write + barrier",
      45: "C(a) per landmark", 46: "C(b) per pair", 47: "C(c) per block (thread 0)", 48: "barrier", 49: "scalars"}
print("linearise phases of group 0 (thread 0):")
for k in range(41, 50):
    d = p[k] - p[k - 1]; print(f"  {ln[k]:36s} {d:10.0f} cyc {d/2100:8.2f} us")
print("  total", (p[49] - p[40]) / 2100, "us")
print("one IMU workgroup (factor 0, no re-preintegration):", (p[63] - p[62]) / 2100, "us")
print("IMU workgroup (us): entry + bias check", (p[36]-p[62])/2100, " cache -> LDS", (p[37]-p[36])/2100, " F + error (one work-item)", (p[39]-p[37])/2100, " J = sqrtInfo F", (p[59]-p[39])/2100, " H, g", (p[63]-p[59])/2100)
