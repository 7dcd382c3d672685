#!/bin/bash
# round 4: rocprofv3 kernel traces (eager = all windows in one launch, graph = the timed loop's sub-batch launches) with the
# per-launch-shape split; summaries -> gpurun_out/prof_r04/ (copied into profiles/r04_* afterwards).  Extra env goes in front.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04${TAG:+_$TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline --no-extras --repeats 5 $BENCH_ARGS > $O/bench_$mode.json 2> $O/bench_$mode.err
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  t=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$mode.csv
  rm -rf $O/$mode
done
head -14 $O/kernel_stats_graph.csv | cut -c1-160; cat $O/kernel_by_shape_graph.csv; echo; cat $O/kernel_by_shape_eager.csv
