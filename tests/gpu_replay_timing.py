"""diagnostics (not a test): per-frame timing split of the C++ replay on a synthetic recording"""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from okvis_amd import recording
d = tempfile.mkdtemp()
recording.write_synthetic_recording(d, duration_s=8.0)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "okvis_amd", "lib", "okvis_amd_replay")
p = subprocess.run([exe, d], capture_output=True, text=True)
print(p.stdout[-500:])
print("\n".join(p.stderr.splitlines()[10:40]))
