import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0
ws = [synthetic.config_A(seed=20240923+i) for i in range(16)]
b = solver.WindowBatch(ws, options=opt)
b.begin(); b.iterate(30); b.synchronize()
c0 = np.array([b.array("IMU_REDO_COUNT", w).sum() for w in range(16)])
b.iterate(100); b.synchronize()
c1 = np.array([b.array("IMU_REDO_COUNT", w).sum() for w in range(16)])
print("redo counts after 30:", c0.astype(int)); print("redos during next 100:", (c1-c0).astype(int))
s = b.finish(); print([round(x['final_cost'],1) for x in s])
