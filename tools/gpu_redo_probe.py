"""Diagnostic (not a test): how often does ImuError's re-preintegration trigger fire in the bench loop?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0
ws = [synthetic.config_A(seed=20240923 + i) for i in range(4)]
b = solver.WindowBatch(ws, options=opt)
b.begin()
prev = np.zeros((4, 9)); prev_sb = None
for blk in range(12):
    b.iterate(10); b.synchronize()
    c = np.array([b.array("IMU_REDO_COUNT", w) for w in range(4)])
    sb = np.array([b.array("SB", w).reshape(-1, 9) for w in range(4)])
    dbg = 0 if prev_sb is None else np.abs(sb[:, :, 3:6] - prev_sb[:, :, 3:6]).max()
    dba = 0 if prev_sb is None else np.abs(sb[:, :, 6:9] - prev_sb[:, :, 6:9]).max()
    print(f"it {10*(blk+1):4d} redos/window/10it {(c-prev).sum(1).astype(int)}  max|d b_g| {dbg:.2e} max|d b_a| {dba:.2e}")
    prev, prev_sb = c, sb
