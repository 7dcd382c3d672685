"""diagnostics (not a test): per-frame iteration counts / costs of the replay on the reference Estimator and on the backend"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ref_lib as R, replay_scenario as RS
from okvis_amd import recording, estimator as E
d = tempfile.mkdtemp()
recording.write_synthetic_recording(d, duration_s=6.0, n_points=280, seed=5)
rec = RS.read(d)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
a = RS.replay(rec, R.RefEstimator, R.RefFrame, max_frames=n)
b = RS.replay(rec, lambda: E.Estimator(0), E.Frame, max_frames=n)
for x, y in zip(a, b):
    sx, sy = x["summary"], y["summary"]
    print(x["frame"], "it %d/%d ok %d/%d term %d/%d  c0 %.10g/%.10g  c %.10g/%.10g" % (
        sx["iterations"], sy["iterations"], sx["successful_steps"], sy["successful_steps"], sx["termination"], sy["termination"],
        sx["initial_cost"], sy["initial_cost"], sx["final_cost"], sy["final_cost"]))
