"""diagnostics (not a test): per-frame timing split of the C++ replay on a synthetic recording, with the window patched between
frames (default) and flattened + uploaded every frame (--no-patch).  OKVIS_BA_DEBUG=build in the environment adds the mean host
time of the solver's sections (index build, container edit, enqueue) to stderr."""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import recording
d = tempfile.mkdtemp()
# OKVIS_AMD_REPLAY_AGE_IDS=1: landmark ids grow with the time their tracks start (a real run's ids) instead of following the point cloud
recording.write_synthetic_recording(d, duration_s=8.0, ids_by_first_sighting=bool(os.environ.get("OKVIS_AMD_REPLAY_AGE_IDS")))
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "okvis_amd", "lib", "okvis_amd_replay")
for extra in ([], ["--no-patch"], [], ["--no-patch"]):
    p = subprocess.run([exe, d] + extra, capture_output=True, text=True)
    print("\n".join(p.stdout.splitlines()[-10:]))
    print("\n".join(l for l in p.stderr.splitlines() if "build_window" in l).replace("  ", "\n    "))
