"""diagnostics (not a test): where a DOGLEG optimize(10) of 64 fresh windows spends its wall time (begin / iterate / finish)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opt = default_options()
if len(sys.argv) > 2:
    opt.tuning.flags = int(sys.argv[2], 0)      # okvis_ba_tuning::flags, e.g. 0x100 = OKVIS_BA_TUNE_FORK_SMALL
res = []
b = None
for rep in range(6):
    ws = [synthetic.config_A(seed=777 + 64 * rep + i) for i in range(NW)]
    t0 = time.perf_counter()
    if b is None:
        b = solver.WindowBatch(ws, options=opt)
    else:
        b.upload(ws)      # the same solver: allocations and launch graphs are kept
    b.synchronize()
    t1 = time.perf_counter()
    b.begin(); b.synchronize()
    t2 = time.perf_counter()
    b.iterate(10); b.synchronize()
    t3 = time.perf_counter()
    s = b.finish()
    t4 = time.perf_counter()
    sl = b.array("SLOTS", 0) if hasattr(b, "array") else None
    res.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, np.mean([x["iterations"] for x in s])))
for r in res:
    print("upload %.2f ms  begin %.3f ms  iterate(10) %.3f ms  finish %.3f ms  mean iterations %.1f" % (r[0] * 1e3, r[1] * 1e3, r[2] * 1e3, r[3] * 1e3, r[4]))
