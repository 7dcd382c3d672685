#!/bin/bash
# round 5, third GPU pass: priors formed by wave 0 in front of the barrier, one trip for 9 chunks, roles spread over waves;
# masked elimination and MFMA issue micro-benchmarks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 30 tests/micro/bin/mfma_lat > $O/mfma_lat.txt 2>&1; cat $O/mfma_lat.txt
for b in ldl16_z ldl16_z_e16; do echo "== $b"; timeout 60 tests/micro/bin/$b > $O/$b.txt 2>&1; grep -E "elimination|D= *-?150|step" $O/$b.txt | head -26; done
timeout 900 python -m pytest tests -m gpu -q --timeout=240 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
for n in 1 8 22; do timeout 120 python tools/gpu_solve_stamps.py $n > $O/stamps_new_$n.txt 2>&1; done
echo "== new, 1 window"; cat $O/stamps_new_1.txt; echo "== new, 22 windows"; cat $O/stamps_new_22.txt
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
for v in new; do
  d=$R/okvis_amd/lib
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
