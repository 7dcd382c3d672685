#!/bin/bash
# round 4, after the index build rewrite (one pass over the observations, pieces laid out with the groups): the whole GPU suite with
# every estimator window checked against a fresh flatten, the default bench line (frame_host record), the replay's per-frame split on
# both window routes with the host sections of the solver.  -> gpurun_out/r04_host/
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_host
mkdir -p $OUT
export TMPDIR=/tmp
OKVIS_AMD_CHECK_PATCH=1 timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -8
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 300 $OUT/bench_default.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_host/bench_default.json"))
fh = d.get("frame_host") or {}
print({k: fh.get(k) for k in ("upload_ms", "optimize10_ms", "patch_newest_frame_ms", "marginalize_ms")})
print(fh.get("estimator_replay"))
print("single", d.get("single_window"))
PY
timeout 200 python tools/gpu_replay_timing.py > $OUT/replay_timing.txt 2>&1
grep -E "medians|route" $OUT/replay_timing.txt
OKVIS_BA_DEBUG_BUILD=1 timeout 200 python tools/gpu_replay_timing.py > $OUT/replay_sections.txt 2>&1
grep -E "route|patch:|index build|staging|observations|groups|lists" $OUT/replay_sections.txt | head -24
