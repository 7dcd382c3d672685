#!/bin/bash
# round 6: the whole GPU suite (twice: the second run tells a flaky test from a broken one), the bench line with the driver's arguments,
# rocprofv3 kernel statistics of the bench command, the solve kernel's stamps.  -> gpurun_out/r06_e/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
export TMPDIR=/tmp
OKVIS_AMD_DEBUG=check_patch timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_2.log 2>&1
tail -3 $O/pytest_gpu_2.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
head -c 300 $O/bench_driver_args.json; echo
for n in 1 22 64; do timeout 120 python tools/gpu_solve_stamps.py $n > $O/solve_stamps_$n.txt 2>&1; done
timeout 200 python tools/gpu_chain_shapes.py > $O/chain_shapes.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o p -- \
  python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 1.0 > $O/bench_graph.json 2> $O/bench_graph.err
f=$(find $O/graph -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_graph.csv
t=$(find $O/graph -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_graph.csv
rm -rf $O/graph
head -6 $O/kernel_stats_graph.csv
echo done
