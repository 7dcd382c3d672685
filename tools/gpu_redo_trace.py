"""Diagnostic: run under rocprofv3 --kernel-trace; every begin() after a fresh upload re-preintegrates all factors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options
opt = default_options(); opt.use_graph = 0
w = synthetic.config_A()
for rep in range(6):
    b = solver.WindowBatch([w], options=opt)
    b.begin(); b.iterate(3); b.finish(); b.close()
