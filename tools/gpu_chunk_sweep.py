"""diagnostics (not a test): landmarks per Schur chunk x sub-batch streams at small batch sizes, us per iteration"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
for nwin in (1, 2, 4, 8, 16):
    ws = [synthetic.config_A(seed=20240923 + i) for i in range(nwin)]
    res = []
    for streams, lm in itertools.product((1, 2), (16, 24, 32, 40, 48, 64)):
        if streams > nwin:
            continue
        opt = default_options(); opt.n_streams = streams; opt.schur_lm_per_block = lm
        opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
        b = solver.WindowBatch(ws, options=opt)
        best = 1e9
        b.begin(); b.iterate(40); b.synchronize()
        for _ in range(5):
            b.iterate(100); best = min(best, b.last_iterate_ms() / 100 * 1e3)
        b.finish(); b.close()
        res.append((best, streams, lm))
    res.sort()
    print(nwin, "windows:", ["%.1f us (streams %d, %d lm)" % r for r in res], flush=True)
