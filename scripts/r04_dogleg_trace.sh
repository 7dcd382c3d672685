#!/bin/bash
# kernel timeline of one 64-window optimize(10) in the reference's DOGLEG mode -> gpurun_out/r04_dogleg/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_dogleg${TAG:+_$TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/scripts/r04_dogleg_trace.py run > $O/plain.txt 2>&1
cat $O/plain.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python $R/scripts/r04_dogleg_trace.py run > $O/traced.txt 2>&1
cat $O/traced.txt | tail -3
t=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python $R/scripts/r04_dogleg_trace.py digest $t > $O/timeline.txt
rm -rf $O/tr
tail -25 $O/timeline.txt
