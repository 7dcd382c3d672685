"""diagnostics (not a test): does the order of the landmarks in a window matter?  The same replay-sized window with its landmarks
in random order, and sorted by their number of observations (the order a real run has: OKVIS landmark ids grow with time)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic, _lib
from okvis_amd.window import default_options

def reorder(w, perm):
    """landmark new index n = old landmark perm[n]"""
    import copy
    v = copy.deepcopy(w)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    v.lm = np.asarray(w.lm)[perm].copy()
    v.obs_lm = inv[np.asarray(w.obs_lm)].astype(np.int32)
    v.sort_observations()
    return v

rng = np.random.default_rng(5)
# visibility varies per landmark: old landmarks seen by all frames, new ones by the last frame or two
w0 = synthetic.make_window(8, 600, 1.0, seed=20240924)
keep = np.ones(w0.obs_lm.size, bool)
first = rng.integers(0, 8, w0.n_lm)                 # the first frame that sees the landmark
keep &= np.asarray(w0.obs_pose) >= first[np.asarray(w0.obs_lm)]
for n in ("obs_lm", "obs_pose", "obs_ext", "obs_cam", "obs_sqrtw", "obs_uv"):
    setattr(w0, n, np.asarray(getattr(w0, n))[keep].copy())
cnt = np.bincount(np.asarray(w0.obs_lm), minlength=w0.n_lm)
orders = {"random": rng.permutation(w0.n_lm), "by age (most observations first)": np.argsort(-cnt, kind="stable"),
          "interleaved old / new": None}
o = np.argsort(-cnt, kind="stable"); il = np.empty_like(o); il[0::2] = o[:(len(o) + 1) // 2]; il[1::2] = o[::-1][:len(o) // 2]
orders["interleaved old / new"] = il
L = _lib.lib()
for name, perm in orders.items():
    w = reorder(w0, perm)
    for dog in (0, 1):
        opt = default_options()
        if not dog:
            opt.gauss_newton = 1; opt.function_tolerance = 0; opt.gradient_tolerance = 0; opt.parameter_tolerance = 0
        b = solver.WindowBatch([w], options=opt)
        if dog:
            import time
            ts = []
            for rep in range(30):
                b.upload([w]); b.synchronize(); t0 = time.perf_counter(); b.optimize(10); ts.append(time.perf_counter() - t0)
            print("%-34s DOGLEG optimize(10) %.3f ms" % (name, np.median(ts[5:]) * 1e3))
        else:
            b.begin(); b.iterate(50); b.synchronize(); b.iterate(200); ms = b.last_iterate_ms()
            pl = b.profile_launches(40)
            print("%-34s %d observations  GN %.1f us / iteration   launches (median us): %s" % (name, w.n_obs, ms / 200 * 1e3, {k: round(float(np.median(v)), 1) for k, v in pl.items()}))
        b.close()
