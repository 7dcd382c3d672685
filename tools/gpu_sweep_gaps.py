"""diagnostics (not a test): per-seed deviations GPU vs oracle of the random sweep (tests/test_gpu_random_sweep.py), and of both
against the oracle built in long double (the referee, tests/test_oracle_referee.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver
from okvis_amd.window import default_options
from tests import oracle_lib
from tests.test_gpu_random_sweep import _case
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 32):
    w, opt, n = _case(seed)
    o = default_options()
    for k, v in opt.items():
        setattr(o, k, v)
    b = solver.WindowBatch([w], options=o)
    sg = b.optimize(n)[0]
    ow = oracle_lib.OracleWindow(w)
    sr = ow.optimize(n, o)
    pg, sbg, lg = b.get_state(); pr, sbr, lr = ow.get_state()
    ld = oracle_lib.OracleWindow(w, extended=True)
    sl = ld.optimize(n, o)
    cl = max(sl["final_cost"], 1e-12)
    print(seed, "cost %.1e" % (abs(sg["final_cost"] - sr["final_cost"]) / max(sr["final_cost"], 1e-12)),
          "pose %.1e sb %.1e lm %.1e" % (np.abs(pg - pr).max(), np.abs(sbg - sbr).max() if sbg.size else 0, np.abs(lg - lr).max()),
          "| vs referee: GPU %.1e oracle %.1e" % (abs(sg["final_cost"] - sl["final_cost"]) / cl, abs(sr["final_cost"] - sl["final_cost"]) / cl),
          "book", (sg["iterations"], sg["successful_steps"], sg["termination"]) == (sr["iterations"], sr["successful_steps"], sr["termination"]))
    b.close()
