"""diagnostics (not a test): longer / different replays - stability of the whole backend loop"""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import recording
for kw in (dict(duration_s=20.0, seed=11), dict(duration_s=10.0, seed=12, frame_rate_hz=20.0, n_points=1500, max_keypoints=300),
           dict(duration_s=10.0, seed=13, detect_prob=0.5, pixel_noise=1.5, n_points=700),
           dict(duration_s=10.0, seed=14, keyframe_every=2), dict(duration_s=10.0, seed=15, keyframe_every=9)):
    d = tempfile.mkdtemp()
    info = recording.write_synthetic_recording(d, **kw)
    r = recording.run_replay(d)
    print(kw, "->", {k: (round(float(v), 4) if not isinstance(v, (bool, int)) else v) for k, v in r.items()}, flush=True)
    shutil.rmtree(d)
