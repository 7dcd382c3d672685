#!/bin/bash
# round 3: default bench line (PMC passes + CPU baselines + DOGLEG / configs[2] records), rocprofv3 kernel traces of the same
# command (eager and graph) with the per-launch-shape split, SQ counters of the solve kernel; summaries -> gpurun_out/prof_r03/
# (copied into profiles/r03_* afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --total-windows 64 --no-pmc --no-cpu-baseline --no-extras > $O/bench_total64.json 2> $O/bench_total64.err
for n in 1 2 4 8 16 32 64 128 256; do
  python bench.py --windows $n --no-pmc --no-cpu-baseline --no-extras --profile-steps 0 --repeats 15 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'windows': $n, 'iterations_per_s': d['value'], 'us_per_step': d['ms_per_step']*1e3}))"
done > $O/batch_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_$mode.json 2> $O/bench_$mode.err
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  t=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$mode.csv
  rm -rf $O/$mode
done
# SQ counters (own pass, kernel trace only) of an eager 64-window run
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o p -- \
  python $R/bench.py --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 2 --steps 20 --warmup 4 --profile-steps 0 > /dev/null 2> $O/pmc.err
c=$(find $O/pmc -name "*counter_collection.csv" | head -1)
[ -n "$c" ] && python - "$c" > $O/pmc_sq_64windows.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ba::"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-22s %16.0f per launch" % (c, v / max(1, n[(k, c)])))
PY
rm -rf $O/pmc
head -c 400 $O/bench_default.json; echo; head -12 $O/kernel_stats_graph.csv; cat $O/kernel_by_shape_graph.csv; cat $O/batch_sweep.jsonl
