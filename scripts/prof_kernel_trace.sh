#!/bin/bash
# rocprofv3 kernel-trace of the default bench workload (eager and graph launch); summaries -> gpurun_out/prof_r01/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_r01
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline > $R/gpurun_out/prof_r01/bench_$mode.json 2> $R/gpurun_out/prof_r01/bench_$mode.err
done
find $R/gpurun_out/prof_r01 -name "*kernel_stats.csv" -exec sh -c 'echo {}; cat {}' \;
