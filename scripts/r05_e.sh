#!/bin/bash
# round 5: quick check — selected GPU tests, solve stamps, quick bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_q
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=240 -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
for n in 1 22; do timeout 120 python tools/gpu_solve_stamps.py $n > $O/stamps_new_$n.txt 2>&1; done
grep -E "total|LDL" $O/stamps_new_1.txt $O/stamps_new_22.txt
timeout 120 python tools/gpu_prof_stamps.py > $O/prof_stamps.txt 2>&1; tail -12 $O/prof_stamps.txt
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
timeout 300 $B > $O/bench_new.json 2> $O/bench_new.err
python - <<PY
import json
d = json.loads(open("$O/bench_new.json").read().strip().splitlines()[-1])
r = d.get("roofline", {}).get("launch_us", {})
print("%9.0f it/s  %.4f ms/step  launches %s  single %s" % (d["value"], d["ms_per_step"], {k: round(x["median"], 1) for k, x in r.items()}, d.get("single_window", {}).get("iterations_per_s")))
PY
echo done
