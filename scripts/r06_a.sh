#!/bin/bash
# round 6, first call: the whole GPU suite (incl. tests/test_gpu_batch64.py: the headline configuration against the oracle), the bench line
# with the driver's arguments, and the rocprofv3 kernel statistics of the same bench command.  -> gpurun_out/r06_a/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
export TMPDIR=/tmp
OKVIS_AMD_DEBUG=check_patch timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
head -c 400 $O/bench_driver_args.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- \
  python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 1.0 > $O/bench_graph.json 2> $O/bench_graph.err
f=$(find $O/graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_graph.csv
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
rm -rf $O/graph
head -8 $O/kernel_stats_graph.csv
echo done
