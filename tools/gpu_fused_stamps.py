"""diagnostics (not a test): where one window-iteration spends its time with and without the fused linearise + reduce launch"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for r0 in (0, 4):
    opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0
    opt.use_graph = 0; opt.debug_arrays = 2; opt.reserved0 = r0
    b = solver.WindowBatch([synthetic.config_A(seed=20240923 + i) for i in range(NW)], options=opt)
    b.begin(); b.iterate(12); b.synchronize()
    p = b.array("PROF")
    T = 2100.0
    print("reserved0 =", r0, "(4 = separate Schur launch)")
    print("  solve: decision done %.2f us, shadow assembly done %.2f, barrier %.2f, imu+priors %.2f, damping %.2f, ldl %.2f, backsub %.2f, end %.2f" % (
        (p[3]-p[0])/T, (p[4]-p[0])/T, (p[1]-p[0])/T, (p[5]-p[0])/T, (p[6]-p[0])/T, (p[7]-p[0])/T, (p[8]-p[0])/T, (p[9]-p[0])/T))
    print("  linearise group 0: loads %.2f A %.2f B %.2f stage %.2f Ca %.2f Cb %.2f Cc %.2f barrier %.2f scalars %.2f | reduce %.2f | total %.2f us" % (
        tuple((p[k]-p[k-1])/T for k in range(41, 49)) + ((p[51]-p[50])/T, (p[50]-p[49])/T, (p[51]-p[40])/T)))
    if True:
        print("  reduction (schur wg 0 / fused group 0): decision %.2f, tables+V^-1 (fast: V^-1) %.2f, fill (zero) %.2f, products (fill) %.2f, more batches (mfma) %.2f, lists (store) %.2f, write %.2f | total %.2f us" % (
            ((p[17]-p[16])/T if r0 else 0.0, (p[18]-(p[17] if r0 else p[49]))/T) + tuple((p[k]-p[k-1])/T for k in range(19, 24)) + ((p[23]-(p[16] if r0 else p[49]))/T,)))
    opt.use_graph = 1; opt.debug_arrays = 0
    b2 = solver.WindowBatch([synthetic.config_A(seed=20240923 + i) for i in range(NW)], options=opt)
    b2.begin(); b2.iterate(40); b2.synchronize()
    best = 1e9
    for _ in range(5):
        b2.iterate(100); best = min(best, b2.last_iterate_ms() * 10)
    print("  %.1f us per iteration (graph)" % best)
    b.close(); b2.close()
