"""diagnostics (not a test): a D = 300 marginalisation, GPU against the oracle, field by field"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKVIS_BA_DEBUG"] = "marg"
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
from tests import oracle_lib
oracle_lib.lib()
w = synthetic.make_window(20, 30, 1.0, 2, frame_dt=0.1)
pm = np.zeros(w.n_pose, np.uint8); sm = np.zeros(w.n_sb, np.uint8); pm[[0, 1]] = 1; sm[[0, 1]] = 1
b = solver.WindowBatch([w], options=default_options())
g = b.marginalize(0, pm, sm)
r = oracle_lib.OracleWindow(w).marginalize(pm, sm, None)
print("dim", g["dim"], r["dim"], "rank", g["rank"], r["rank"], "sweeps", g["sweeps"])
for k in ("H", "b0", "J", "e0"):
    print(k, "nan", int(np.isnan(g[k]).sum()), "max", float(np.nanmax(np.abs(g[k]))), "ref max", float(np.abs(r[k]).max()))
print("H rel", np.abs(g["H"] - r["H"]).max() / np.abs(r["H"]).max())
print("JtJ rel", np.abs(g["J"].T @ g["J"] - r["H"]).max() / np.abs(r["H"]).max())
ev = np.linalg.eigvalsh(0.5 * (r["H"] + r["H"].T) / np.sqrt(np.outer(np.diag(r["H"]), np.diag(r["H"]))))
print("scaled eigenvalues: min %.3e max %.3e  eps n max %.3e" % (ev.min(), ev.max(), 2.2e-16 * len(ev) * ev.max()))
