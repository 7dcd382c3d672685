"""diagnostics (not a test): per-frame deviation of okvis_amd::Estimator from the reference's okvis::Estimator"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import ref_lib as R, estimator_scenarios as S
from okvis_amd import estimator as E
kw = dict(n_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 6, iters=int(sys.argv[2]) if len(sys.argv) > 2 else 6, seed=13,
          marginalize=(len(sys.argv) > 3 and sys.argv[3] == "1"))
tr_r, _ = S.sliding_window(R.RefEstimator, R.RefFrame, **kw)
tr_g, _ = S.sliding_window(lambda: E.Estimator(0), E.Frame, **kw)
for a, b in zip(tr_r, tr_g):
    dp = max(np.abs(a["poses"][f] - b["poses"][f]).max() for f in a["poses"])
    ds = max(np.abs(a["sbs"][f] - b["sbs"][f]).max() for f in a["sbs"])
    print(a["frame"], "ref", a["summary"]["initial_cost"], a["summary"]["final_cost"], a["summary"]["iterations"],
          "gpu", b["summary"]["initial_cost"], b["summary"]["final_cost"], b["summary"]["iterations"], "dpose %.2e dsb %.2e" % (dp, ds))
