"""Diagnostics (not a test): phase stamps of the solve kernel of window 0 (clock64 ticks and us at 2.38 GHz).  python tools/gpu_solve_stamps.py [n_windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0; opt.use_graph = 0; opt.debug_arrays = 2
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt.tuning.solve_mode = {"dense": 1, "chain": 2}.get(sys.argv[2] if len(sys.argv) > 2 else "", 0)   # okvis_ba_tuning::solve_mode (default: auto)
b = solver.WindowBatch([synthetic.config_A(seed=20240923 + i) for i in range(NW)], options=opt)
b.begin(); b.iterate(12); b.synchronize()
p = b.array("PROF")
T = 2380.0
print("head (us after the first instruction): control words + window record %.2f, (stamp 0 %.2f), prologue requests issued %.2f, control record + trial partials %.2f, "
      "wave 0 decision done %.2f, Schur sums in LDS %.2f" % tuple((p[k] - p[43]) / T for k in (40, 0, 41, 42, 3, 4)))
names = [(1, "decision + Schur partial sums"), (58, "IMU records"), (5, "priors + marginalisation prior"), (6, "convergence + damping + rhs column"),
         (7, "LDL^T + back-substitution"), (9, "scalars, trial states, ctrl")]
prev = 0
for k, n in names:
    print("  %-36s %8.0f ticks %7.2f us" % (n, p[k] - p[prev], (p[k] - p[prev]) / T)); prev = k
print("  total %.2f us from stamp 0, %.2f us from the first instruction" % ((p[9] - p[0]) / T, (p[9] - p[43]) / T))
print("  tail: step vector %.2f, trial states %.2f, sums + barrier %.2f, ctrl %.2f us" % ((p[31] - p[7]) / T, (p[32] - p[31]) / T, (p[33] - p[32]) / T, (p[9] - p[33]) / T))
ids = [43, 40, 0, 41, 42, 3, 44, 4, 1, 2, 45, 46, 58, 5, 6, 10, 7, 8, 31, 32, 33, 9]
base = min(p[k] for k in ids if p[k] > 0)
print("  raw stamps (us after the earliest): " + "  ".join("%d:%.2f" % (k, (p[k] - base) / T) for k in ids if p[k] > 0))
import numpy as np
q = np.asarray(p[64:64 + 128]).view(np.int64)
print("  LDL^T solver: load %.2f, factor %.2f, back-substitution %.2f us" % ((q[1] - q[0]) / T, (q[2] - q[1]) / T, (q[3] - q[2]) / T))
if q[16] > 0 and q[20] > q[16]:   # chain solver (ba_chain.hpp): its own phases around the pose system's LDL^T
    print("  chain solver: sweeps %.2f, pose update %.2f, pose LDL^T %.2f, speed/bias back-substitution %.2f us  (route: %s)" %
          ((q[17] - q[16]) / T, (q[18] - q[17]) / T, (q[19] - q[18]) / T, (q[20] - q[19]) / T, b.launch_route()))
    print("    sweeps (us after their start): chain wave left %.2f right %.2f, column wave left %.2f right %.2f;  pose update: matrix-core waves %.2f, G %.2f;  "
          "back-substitution: u %.2f" % (tuple((q[i] - q[16]) / T for i in (21, 22, 23, 24)) + tuple((q[i] - q[17]) / T for i in (25, 26)) + ((q[27] - q[19]) / T,)))
    print("    pose update, waves through after (us): " + " ".join("%.2f" % ((q[40 + w] - q[17]) / T) for w in range(16)))
    print("    u of the back-substitution, waves through after (us): " + " ".join("%.2f" % ((q[56 + w] - q[19]) / T) for w in range(16)))
    print("    chain wave left, per step (ticks): " + " ".join("%d" % (q[33 + i] - q[32 + i]) for i in range(8) if q[33 + i] > 0 and q[32 + i] > 0))
if q[32] > 0 and q[33] > q[32]:
    nbk = int(np.count_nonzero(q[32:48]))
    print("  wave 0's steps (cycles; hand-overs requested again .. both there in brackets): " + " ".join("%d (%d)" % (q[33 + k] - q[32 + k], q[64 + k] - q[48 + k]) for k in range(nbk - 1)))
print("  waves at the barrier of the head (us after stamp 0): " + " ".join("%d:%.2f" % (w, (p[170 + w] - p[0]) / T) for w in range(16)))
