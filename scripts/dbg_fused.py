import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
for r0 in (0, 4):
    opt = default_options(); opt.reserved0 = r0; opt.use_graph = 0
    b = solver.WindowBatch([synthetic.config_A(seed=1)], options=opt)
    print(r0, b.optimize(3)[0])
    b.close()
