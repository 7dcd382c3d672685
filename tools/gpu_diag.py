"""Stage-by-stage GPU-vs-oracle diagnostic (run on the GPU box: python tools/gpu_diag.py).

Not a pytest module: prints the relative difference of every intermediate array so that one gpurun call
localises a bug to a kernel.  The pytest parity tests are in tests/test_gpu_parity.py.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from okvis_amd import solver, synthetic  # noqa: E402
from okvis_amd.window import default_options  # noqa: E402
from tests import oracle_lib  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.shape != b.shape:
        return f"SHAPE {a.shape} vs {b.shape}"
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return f"{d / s:.3e} (max|ref| {s:.3e})"


def stage_compare(w, label, n_iter=0):
    print(f"==== {label}: K={w.meta.get('K')} L={w.n_lm} obs={w.n_obs} imu={w.n_imu} D={w.reduced_dim()} ====", flush=True)
    opt = default_options()
    opt.debug_arrays = 1
    opt.use_graph = 0
    b = solver.WindowBatch([w], options=opt)
    o = oracle_lib.OracleWindow(w)
    c_ref = o.linearize()
    b.begin()
    s = b.finish()
    print("cost0 gpu", s[0]["final_cost"], "ref", c_ref, "rel", abs(s[0]["final_cost"] - c_ref) / c_ref)
    for name in ("OBS_RESIDUAL", "LM_V", "LM_B", "LM_HQ", "PAIR_W", "IMU_RESIDUAL", "LM_QUALITY", "GRADIENT"):
        try:
            print(f"  {name:14s}", rel(b.array(name), o.array(name)))
        except Exception as e:  # noqa: BLE001
            print(f"  {name:14s} ERROR {e}")
    pa, pb = b.pairs()
    oa, ob = o.pairs()
    print("  pairs equal", np.array_equal(pa, oa) and np.array_equal(pb, ob))
    # one solve
    b.begin()
    b.iterate(1)
    assert o.solve(opt.initial_radius, opt) == 0
    for name in ("REDUCED_S", "REDUCED_RHS", "DAMPING", "STEP"):
        try:
            print(f"  {name:14s}", rel(b.array(name), o.array("STEP" if name == "STEP" else name) if name != "DAMPING"
                                        else np.clip(np.diag(o.array("HPP").reshape(o.D, o.D)), 1e-12, 1e64)))
        except Exception as e:  # noqa: BLE001
            print(f"  {name:14s} ERROR {e}")
    s = b.finish()
    print("  after 1 it:", s[0])
    so = oracle_lib.OracleWindow(w).optimize(1)
    print("  oracle 1 it:", so)
    for n in (2, 5, 10, 30):
        bb = solver.WindowBatch([w], options=opt)
        sg = bb.optimize(n)[0]
        oo = oracle_lib.OracleWindow(w)
        sr = oo.optimize(n)
        pg, sbg, lg = bb.get_state()
        pr, sbr, lr = oo.get_state()
        print(f"  n={n:3d} cost gpu {sg['final_cost']:.12e} ref {sr['final_cost']:.12e} rel {abs(sg['final_cost']-sr['final_cost'])/sr['final_cost']:.2e}"
              f" it {sg['iterations']}/{sr['iterations']} succ {sg['successful_steps']}/{sr['successful_steps']} term {sg['termination']}/{sr['termination']}"
              f" cholfail {sg.get('reserved')} dpose {np.abs(pg-pr).max():.2e} dsb {np.abs(sbg-sbr).max():.2e} dlm {np.abs(lg-lr).max():.2e}", flush=True)
        bb.close()
    b.close()


def main():
    print("limits", solver.limits())
    t = time.time()
    stage_compare(synthetic.small_window(seed=1, K=4, L=40, with_imu=True), "small fixed-ext")
    stage_compare(synthetic.small_window(seed=2, K=4, L=40, estimate_extrinsics="shared"), "small shared-ext")
    stage_compare(synthetic.small_window(seed=3, K=3, L=30, estimate_extrinsics="perframe"), "small perframe-ext")
    stage_compare(synthetic.config_A(), "config A")
    # graph path + batch
    ws = [synthetic.config_A(seed=20240923 + i) for i in range(4)]
    opt = default_options()
    bb = solver.WindowBatch(ws, options=opt)
    t0 = time.time()
    sg = bb.optimize(10)
    print("batch graph optimize(10) wall", time.time() - t0)
    for i, w in enumerate(ws):
        sr = oracle_lib.OracleWindow(w).optimize(10)
        print(f"  win{i} gpu {sg[i]['final_cost']:.10e} ref {sr['final_cost']:.10e} rel {abs(sg[i]['final_cost']-sr['final_cost'])/sr['final_cost']:.2e}")
    # timing
    opt2 = default_options()
    opt2.function_tolerance = 0
    opt2.gradient_tolerance = 0
    opt2.parameter_tolerance = 0
    for nb in (1, 8, 64):
        ws = [synthetic.config_A(seed=20240923 + i) for i in range(nb)]
        bb = solver.WindowBatch(ws, options=opt2)
        bb.begin()
        bb.iterate(20)
        bb.synchronize()
        bb.iterate(20)
        ms = bb.last_iterate_ms()
        print(f"batch {nb}: 20 iterations {ms:.3f} ms -> {nb*20/ms*1e3:.0f} window-iterations/s", flush=True)
        print("   per-kernel ms over 10 it:", bb.profile_iterations(10))
        bb.finish()
        bb.close()
    print("total", time.time() - t)


if __name__ == "__main__":
    main()
