#!/bin/bash
# round 6: the new tests again + where a DOGLEG optimize(10) of 64 windows spends its time (phases, kernel statistics by shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dogleg.py tests/test_gpu_matcher_binding.py tests/test_gpu_rccl_stub.py tests/test_gpu_replay.py -m gpu -q > $O/pytest_new.log 2>&1
tail -25 $O/pytest_new.log
timeout 300 python tools/gpu_dogleg_phases.py 64 > $O/dogleg_phases_64.txt 2>&1
cat $O/dogleg_phases_64.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dl -o p -- python $R/tools/gpu_dogleg_phases.py 64 > $O/dogleg_prof.txt 2>&1
f=$(find $O/dl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_dogleg.csv
t=$(find $O/dl -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_dogleg.csv
[ -n "$t" ] && python $R/scripts/stream_gaps.py $t > $O/stream_gaps_dogleg.txt 2>&1
[ -n "$t" ] && head -400 $t > $O/kernel_trace_head.csv
rm -rf $O/dl
head -12 $O/kernel_stats_dogleg.csv
cat $O/kernel_by_shape_dogleg.csv | head -30
echo done
