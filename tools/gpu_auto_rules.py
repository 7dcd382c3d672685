"""diagnostics (not a test): which launch settings win at which batch size (streams x level elimination x Schur chunk)"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
for nwin in (4, 8, 12, 16, 24, 32, 48):
    ws = [synthetic.config_A(seed=20240923 + i) for i in range(nwin)]
    res = []
    for streams, sbl, lm in itertools.product((1, 2, 3), (1, 2), (32, 48)):
        if streams > nwin:
            continue
        opt = default_options(); opt.n_streams = streams; opt.reserved0 = sbl; opt.schur_lm_per_block = lm
        opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
        b = solver.WindowBatch(ws, options=opt)
        b.begin(); b.iterate(40); b.synchronize(); b.iterate(120); ms = b.last_iterate_ms()
        b.finish(); b.close()
        res.append((ms / 120 * 1e3, streams, "levels" if sbl == 1 else "dense", lm))
    res.sort()
    print(nwin, "windows: best", ["%.1f us (streams %d, %s, %d lm)" % r for r in res[:3]], " worst %.1f" % res[-1][0], flush=True)
