import sys; sys.path.insert(0,'/root/repo')
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.use_graph = 0
b = solver.WindowBatch([synthetic.config_A()], options=opt)
b.begin(); b.synchronize()
p = b.array("PROF")
names = ["stage 0 control", "stage 1 per-step", "stage 2 dq/cross serial", "stages 3-7", "suffix passes", "combine", "P reduce (dense rows)", "inverse + chol", "-", "first-search"]
for k, n in enumerate(names): print(f"{n:28s} {p[40+k]:10.0f} cyc {p[40+k]/2100:8.1f} us")
print("total", p[40:50].sum()/2100)
