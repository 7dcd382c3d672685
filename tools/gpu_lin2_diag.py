"""Diagnostic: first linearisation of one window on the piece path against the staged kernel, array by array."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options

def run(res0, w):
    opt = default_options(); opt.debug_arrays = 1; opt.use_graph = 0; opt.reserved0 = res0
    b = solver.WindowBatch([w], options=opt)
    b.begin()
    out = {"cost": b.finish()[0]["initial_cost"]}
    out.update({n: b.array(n).copy() for n in ("OBS_RESIDUAL", "LM_V", "LM_B", "LM_HQ", "PAIR_W")})
    out["GRADIENT"] = np.zeros(1)
    b.close()
    return out

w = synthetic.config_A() if len(sys.argv) < 2 else synthetic.small_window(seed=3, K=4, L=60)
new, old = run(4, w), run(12, w)
print("cost", new["cost"], old["cost"])
for n in ("OBS_RESIDUAL", "LM_V", "LM_B", "LM_HQ", "PAIR_W", "GRADIENT"):
    a, b_ = new[n], old[n]
    d = np.abs(a - b_)
    sc = np.abs(b_).max() + 1e-300
    bad = np.nonzero(d > 1e-9 * sc)[0]
    print(f"{n:14s} n {a.size:6d} max rel diff {d.max() / sc:.3e}  mismatching entries {bad.size}  first {bad[:12]}")
    if n in ("LM_V", "LM_HQ", "PAIR_W") and bad.size:
        per = {"LM_V": 6, "LM_HQ": 6, "PAIR_W": 18}[n]
        ids = np.unique(bad // per)
        print("   items", ids[:60], "count", ids.size)
        i0 = ids[0]
        print("   new", a[per * i0:per * i0 + per], "\n   old", b_[per * i0:per * i0 + per])
    if False:
        lms = np.unique(bad // 6)
        print("   landmarks", lms[:40])
        l = lms[0]
        print("   new", a[6 * l:6 * l + 6], "\n   old", b_[6 * l:6 * l + 6])
obs_lm = np.asarray(w.obs_lm)
print("obs per landmark (first 8):", np.bincount(obs_lm)[:8], " poses fixed:", np.asarray(w.pose_fixed)[:12])
