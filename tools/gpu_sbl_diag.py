"""diagnostics (not a test): level-scheduled speed/bias elimination on / off — parity and time per iteration"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
on, nwin = int(sys.argv[1]), int(sys.argv[2])   # on = 1: level-scheduled elimination forced on (options.reserved0 bit 0), 0: forced off (bit 1)
ws = [synthetic.config_A(seed=20240923 + i) for i in range(nwin)]
opt = default_options(); opt.reserved0 = 1 if on else 2
opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
b = solver.WindowBatch(ws, options=opt)
b.begin(); b.iterate(50); b.synchronize(); b.iterate(50); ms = b.last_iterate_ms()
pl = {k: float(np.median(v)) * 1e3 for k, v in b.profile_launches(40).items()}
s = b.finish()
print("sb levels   " if on else "dense order ", "windows", nwin, "us/iter %.1f" % (ms / 50 * 1e3), "cost0 %.12f" % s[0]["final_cost"], "launch medians us", {k: round(v, 1) for k, v in pl.items()}, flush=True)
