#!/bin/bash
# round 5, second GPU pass: speed/bias-first solver ordering with zero-block skipping, IMU records in LDS order; whole GPU suite,
# stamps, quick bench lines against the round-4 library, LDL^T micro-benchmark (dense + structured), kernel trace by launch shape.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 60 tests/micro/bin/ldl16_z > $O/ldl16_z.txt 2>&1; grep -E "D= *-?1[0-9][0-9]|D= *90" $O/ldl16_z.txt | head -12
timeout 60 tests/micro/bin/ldl16_z_trace 1 x > $O/ldl16_z_trace.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout=240 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
for v in r4 new; do
  d=$R/okvis_amd/lib_variants/$v; [ $v = new ] && d=$R/okvis_amd/lib
  for n in 1 8 22; do
    OKVIS_AMD_LIB_DIR=$d timeout 120 python tools/gpu_solve_stamps.py $n > $O/stamps_${v}_$n.txt 2>&1
  done
done
echo "== new, 1 window"; cat $O/stamps_new_1.txt; echo "== new, 22 windows"; cat $O/stamps_new_22.txt
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
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- python $R/bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_graph.json 2> $O/bench_graph.err
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
f=$(find $O/graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_graph.csv
rm -rf $O/graph
grep -E "solve|schur|linearize2|small" $O/kernel_by_shape_graph.csv | head -20
echo done
