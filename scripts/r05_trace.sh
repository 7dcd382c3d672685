#!/bin/bash
# round 5: rocprofv3 kernel trace of the bench command (graph replay), split by launch shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_t
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- python $R/bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_graph.json 2> $O/bench_graph.err
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
rm -rf $O/graph
grep -E "solve|schur|linearize2|small" $O/kernel_by_shape_graph.csv
