"""Diagnostics (not a test): device ticks of the two LDS-resident solvers of the reduced system over window shapes
(okvis_ba_reduced_solve): where the chain solver pays.  python tools/gpu_chain_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.chain_emulation import chain_structured_system
from tests.test_gpu_chain_solve import reduced_solve
from okvis_amd.window import SOLVE_CHAIN, SOLVE_DENSE
print("poses sb  D   dense  chain  (device ticks per solve, mean of 20)")
for n_pose, n_sb in [(10, 10), (10, 8), (10, 6), (10, 5), (10, 4), (10, 3), (10, 2), (8, 3), (6, 3), (6, 6), (4, 4), (12, 10), (5, 5), (3, 3)]:
    rng = np.random.default_rng(7)
    H, g, Dp = chain_structured_system(rng, n_pose, n_sb, pose_prior=1e10)
    _, td, _ = reduced_solve(H, g, Dp, SOLVE_DENSE, repeats=20)
    _, tc, _ = reduced_solve(H, g, Dp, SOLVE_CHAIN, repeats=20)
    print(f"{n_pose:5d} {n_sb:2d} {H.shape[0]:3d} {td:6d} {tc:6d}  {tc / td:.2f}", flush=True)
