"""GPU: the library built with -DLDL_SIGNAL_FENCE (scripts/build_variant.sh fenced -DLDL_SIGNAL_FENCE — every hand-over of the
LDLᵀ and chain solvers behind a workgroup release fence instead of the in-order-LDS argument of ba_ldl16.hpp::ldl_signal) gives
the SAME BITS as the shipped one: steps, costs and iteration counts of single windows in both solve modes and of a 22-window batch.
Skipped when the variant has not been built (okvis_amd/lib_variants/fenced/, git-ignored; __graft_entry__.build() builds it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FENCED = os.path.join(ROOT, "okvis_amd", "lib_variants", "fenced")

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from okvis_amd import solver, synthetic
from okvis_amd.window import default_options, SOLVE_DENSE, SOLVE_CHAIN
out = {}
for mode in (SOLVE_DENSE, SOLVE_CHAIN):
    for seed in (1, 2):
        o = default_options()
        o.tuning.solve_mode = mode
        b = solver.WindowBatch([synthetic.config_A(seed=seed)], options=o)
        s = b.optimize(6)[0]
        out["w%%d_m%%d" %% (seed, mode)] = [s["iterations"], s["successful_steps"], float(s["final_cost"]).hex(),
                                           [float(x).hex() for x in b.array("STEP")[:12]]]
        b.close()
o = default_options()
o.gauss_newton = 1
b = solver.WindowBatch([synthetic.config_A(seed=100 + i) for i in range(22)], options=o)
ss = b.optimize(5)
out["batch"] = [float(s["final_cost"]).hex() for s in ss]
b.close()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(lib_dir):
    env = dict(os.environ)
    if lib_dir:
        env["OKVIS_AMD_LIB_DIR"] = lib_dir
    else:
        env.pop("OKVIS_AMD_LIB_DIR", None)
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.skipif(not os.path.exists(os.path.join(FENCED, "libokvis_amd_ba.so")), reason="the fenced variant has not been built")
def test_fenced_hand_overs_give_the_same_bits():
    a, b = _run(None), _run(FENCED)
    assert a.keys() == b.keys() and len(a["batch"]) == 22
    for k in a:
        assert a[k] == b[k], k
