#!/bin/bash
# round 4, after the estimator's patch route, the tiled marginalisation tail and the helper time-out fallback: the whole GPU suite
# (every estimator window checked against a fresh flatten), the default bench line, the kernel trace of the same command, the
# marginalisation sizes with their kernel trace, the replay timing of both window routes.  -> gpurun_out/prof_r04_final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04_final
mkdir -p $O
cd $R
export TMPDIR=/tmp
OKVIS_AMD_CHECK_PATCH=1 timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
head -c 400 $O/bench_default.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- \
  python $R/bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_graph.json 2> $O/bench_graph.err
f=$(find $O/graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_graph.csv
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
rm -rf $O/graph
timeout 200 python $R/scripts/bench_marginalize.py --config-c > $O/bench_marginalize.json 2> $O/bench_marginalize.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/marg -o p -- python $R/scripts/bench_marginalize.py --config-c > $O/bench_marginalize_under_rocprof.json 2>> $O/bench_marginalize.err
f=$(find $O/marg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_marginalize.csv
rm -rf $O/marg
cd $R
timeout 200 python tools/gpu_replay_timing.py > $O/replay_timing.txt 2>&1
timeout 120 python scripts/bench_config_c.py > $O/bench_config_C.json 2> $O/bench_config_C.err
grep -E "medians|route" $O/replay_timing.txt; head -12 $O/kernel_stats_marginalize.csv | cut -c1-160; tail -c 300 $O/bench_config_C.json
