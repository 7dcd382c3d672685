"""diagnostics (not a test): per-slot trust-region record of the HIP dogleg path next to the oracle's trace"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G
import oracle_lib as O
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
case, radius, n = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
w = synthetic.small_window(**G.SMALL[case])
opt = default_options(0); opt.initial_radius = radius
b = solver.WindowBatch([w], options=opt)
b.begin()
names = "radius mu cost cA beta dl model A C E gd_p ddd_p rho last_model iter succ kind expl pend acc done max_iter inval chol".split()
for k in range(n + 4):
    b.iterate(1)
    c = b.array("CTRL")
    print("gpu slot", k, " ".join(f"{a}={v:.17g}" for a, v in zip(names, c)))
print(b.finish()[0])
os.environ["ORC_TRACE"] = "1"
o = O.OracleWindow(w)
print(o.optimize(n, opt))
