#!/bin/bash
# quick pass: the parity tests closest to the kernels + a few bench lines + stamps
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_quick
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_dogleg.py tests/test_gpu_structure_paths.py tests/test_gpu_marginalization.py tests/test_gpu_random_sweep.py -m gpu -q -x > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | tail -8
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
run() { name=$1; shift; ( "$@" ) > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {}).get("launch_us", {})
    print("%-22s %9.0f it/s  %.4f ms/step  launches %s  single %s" % ("$name", d["value"], d["ms_per_step"], {k: round(v["median"], 1) for k, v in r.items()}, d.get("single_window", {}).get("iterations_per_s")))
except Exception as e:
    print("$name failed", e)
PY
}
for a in "$@"; do eval "$a"; done
