#!/usr/bin/env python3
"""BASELINE configs[2] (50 keyframes / 2000 landmarks / 200 000 observations, D = 750): per-iteration time on the
GPU (Gauss-Newton mode, graph replay) with the per-kernel split, next to the CPU oracle.  One JSON line.

Measurement script, not product code: the oracle (tests/oracle_lib.py -> oracle/) is called here in the same role
as in bench.py's `cpu_baseline` leg — a reported CPU column, never part of the measured GPU path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
from tests import oracle_lib

w = synthetic.config_C()
opt = default_options(); opt.gauss_newton = 1
opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
b = solver.WindowBatch([w], options=opt)
b.begin(); b.iterate(5); b.iterate(20); b.synchronize()
b.iterate(20)
ms = b.last_iterate_ms() / 20
prof = {k: v * 1e3 / 10 for k, v in b.profile_iterations(10).items()}
s = b.finish()[0]
b.close()
ow = oracle_lib.OracleWindow(w)
tc = ow.time_iterations(3, opt) / 3
print(json.dumps({"config": "BASELINE configs[2]: 50 KF / 2 cam / 2000 landmarks / %d obs, D = %d" % (w.n_obs, w.reduced_dim()),
                  "gpu_ms_per_iteration": ms, "gpu_iterations_per_s": 1e3 / ms, "per_kernel_us": prof,
                  "cpu_oracle_ms_per_iteration": tc * 1e3, "final_cost": s["final_cost"]}))
