import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import test_gpu_random_sweep as T, oracle_lib
from okvis_amd import solver
from okvis_amd.window import default_options
bad = 0
for seed in range(32, 232):
    w, opt, n = T._case(seed)
    o = default_options()
    for k, v in opt.items(): setattr(o, k, v)
    try:
        b = solver.WindowBatch([w], options=o); sg = b.optimize(n)[0]; b.close()
    except Exception as e:
        print(seed, "EXC", e); bad += 1; continue
    sr = oracle_lib.OracleWindow(w).optimize(n, o)
    rel = abs(sg["final_cost"] - sr["final_cost"]) / max(sr["final_cost"], 1e-12)
    same = (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"])
    if rel > 1e-9 or not same:
        bad += 1; print(seed, rel, same, sg["iterations"], sr["iterations"], sg["termination"], sr["termination"])
print("done, mismatches:", bad)
