#!/bin/bash
# round 6, final tree: the whole GPU suite (estimator windows checked against a fresh flatten), the bench line with the driver's
# arguments, solve stamps, chain shapes, rocprofv3 kernel statistics of the bench command -> gpurun_out/r06_final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
export TMPDIR=/tmp
OKVIS_AMD_DEBUG=check_patch timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
head -c 300 $O/bench_driver_args.json; echo
for n in 1 22 64; do timeout 120 python tools/gpu_solve_stamps.py $n > $O/solve_stamps_$n.txt 2>&1; done
timeout 120 python tools/gpu_solve_stamps.py 1 dense > $O/solve_stamps_1_dense.txt 2>&1
timeout 200 python tools/gpu_chain_shapes.py > $O/chain_shapes.txt 2>&1
timeout 300 python tools/gpu_dogleg_phases.py 64 > $O/dogleg_phases_64.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- \
  python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 1.0 > $O/bench_graph.json 2> $O/bench_graph.err
f=$(find $O/graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_graph.csv
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
[ -n "$t" ] && python $R/scripts/stream_gaps.py $t > $O/stream_gaps.txt 2>&1
rm -rf $O/graph
head -6 $O/kernel_stats_graph.csv
cat $O/stream_gaps.txt | head -30
echo done
