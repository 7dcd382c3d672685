#!/bin/bash
# round 5, re-entry pass: whole GPU suite on HEAD, phase stamps and quick bench lines against the round-4 library (same lease)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=240 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
for v in r4 new; do
  d=$R/okvis_amd/lib_variants/$v; [ $v = new ] && d=$R/okvis_amd/lib
  for n in 1 8 22; do
    OKVIS_AMD_LIB_DIR=$d timeout 120 python tools/gpu_solve_stamps.py $n > $O/stamps_${v}_$n.txt 2>&1
  done
  echo "== $v, 1 window"; cat $O/stamps_${v}_1.txt
done
echo "== new, 22 windows"; cat $O/stamps_new_22.txt
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
for v in r4 new; do
  d=$R/okvis_amd/lib_variants/$v; [ $v = new ] && d=$R/okvis_amd/lib
  OKVIS_AMD_LIB_DIR=$d timeout 300 $B > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {}).get("launch_us", {})
    print("%-8s %9.0f it/s  %.4f ms/step  launches %s  single %s" % ("$v", d["value"], d["ms_per_step"], {k: round(x["median"], 1) for k, x in r.items()}, d.get("single_window", {}).get("iterations_per_s")))
except Exception as e:
    print("$v failed", e)
PY
done
echo done
