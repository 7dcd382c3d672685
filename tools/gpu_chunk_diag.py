"""diagnostics (not a test): landmarks per Schur workgroup (options.schur_lm_per_block) vs time per iteration"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ws = [synthetic.config_A(seed=20240923 + i) for i in range(nwin)]
for lm in (0, 16, 24, 32, 48, 64):
    opt = default_options(); opt.schur_lm_per_block = lm
    opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
    b = solver.WindowBatch(ws, options=opt)
    b.begin(); b.iterate(50); b.synchronize(); b.iterate(100); ms = b.last_iterate_ms()
    pl = {k: round(float(np.median(v)) * 1e3, 1) for k, v in b.profile_launches(40).items()}
    s = b.finish(); b.close()
    print("lm per block", lm, "windows", nwin, "us/iter %.1f" % (ms / 100 * 1e3), pl, "cost %.9f" % s[0]["final_cost"], flush=True)
