import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
r0 = int(sys.argv[1])
opt = default_options(); opt.reserved0 = r0; opt.use_graph = 0
opt.gauss_newton = 1; opt.function_tolerance = opt.gradient_tolerance = opt.parameter_tolerance = 0.0
b = solver.WindowBatch([synthetic.config_A(seed=20240923)], options=opt)
b.begin(); b.iterate(300); b.synchronize(); b.finish(); b.close()
