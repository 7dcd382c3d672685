#!/bin/bash
# round 2: default bench line (PMC passes + CPU baselines), rocprofv3 kernel traces of the same command (eager and graph),
# per-launch-shape split; summaries -> gpurun_out/prof_r02/ (copied into profiles/r02_* afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --total-windows 64 --no-pmc --no-cpu-baseline > $O/bench_total64.json 2> $O/bench_total64.err
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline --repeats 5 > $O/bench_$mode.json 2> $O/bench_$mode.err
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  t=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$mode.csv
  rm -rf $O/$mode
done
head -c 600 $O/bench_default.json; echo; cat $O/kernel_stats_eager.csv | head -12; cat $O/kernel_by_shape_eager.csv
