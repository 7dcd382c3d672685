import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
w = synthetic.small_window(seed=9, K=8, L=350, visibility=0.8)
for use_graph in (1, 0):
    opt = default_options(); opt.use_graph = use_graph
    opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
    b = solver.WindowBatch([w], options=opt)
    b.begin(); b.synchronize()
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); b.iterate(10); b.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("use_graph", use_graph, "iterate(10) wall ms:", [round(t, 3) for t in ts], "D", w.reduced_dim(), "obs", w.n_obs)
    b.finish(); b.close()
