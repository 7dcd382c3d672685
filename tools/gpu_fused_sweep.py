"""diagnostics (not a test): fused linearise + reduce launch against the separate Schur launch over batch sizes, us per iteration"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
for nwin in [int(x) for x in sys.argv[1:]] or (1, 2, 4, 8, 12, 16, 24, 32, 64):
    ws = [synthetic.config_A(seed=20240923 + i) for i in range(nwin)]
    out = []
    for r0 in (0, 4):
        opt = default_options(); opt.reserved0 = r0
        opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
        b = solver.WindowBatch(ws, options=opt)
        best = 1e9
        b.begin(); b.iterate(40); b.synchronize()
        for _ in range(5):
            b.iterate(100); best = min(best, b.last_iterate_ms() / 100 * 1e3)
        b.finish(); b.close()
        out.append(best)
    print("%3d windows: fused %.1f us, separate %.1f us" % (nwin, out[0], out[1]), flush=True)
