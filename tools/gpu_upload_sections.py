"""diagnostics (not a test): host time of okvis_ba_upload by section for a replay-sized window (OKVIS_BA_DEBUG=build,upload)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKVIS_BA_DEBUG"] = "build,upload"
from okvis_amd import solver, synthetic
from okvis_amd.window import WindowC, default_options
w = synthetic.make_window(8, 430, 0.5, seed=20240924)
rng = np.random.default_rng(0)
Dm = 45
w.marg_J = np.triu(rng.standard_normal((Dm, Dm))) + 3 * np.eye(Dm); w.marg_e0 = rng.standard_normal(Dm)
w.marg_block_type = np.array([0, 1, 0, 1, 0, 1], np.int32); w.marg_block_idx = np.array([0, 0, 1, 1, 2, 2], np.int32)
w.marg_block_off = np.array([0, 6, 15, 21, 30, 36], np.int32)
w.marg_lin = np.zeros((6, 9)); w.marg_lin[[0, 2, 4], :7] = w.pose[:3]; w.marg_lin[[1, 3, 5]] = w.sb[:3]
b = solver.WindowBatch([w], options=default_options())
wc, keep = w.as_c()
arr = (WindowC * 1)(wc)
ts = []
for i in range(300):
    b.synchronize()
    t0 = time.perf_counter()
    assert b._L.okvis_ba_upload(b._h, 1, arr) == 0
    ts.append(time.perf_counter() - t0)
print("upload median %.4f ms, p10 %.4f" % (np.median(ts) * 1e3, np.percentile(ts, 10) * 1e3))
b.close()
