#!/bin/bash
# round 3, second part: configs[2] kernel trace + MFMA counters of the tiled Cholesky, and the kernel averages of ONE window with
# the fused linearise + reduce launch and with the separate Schur launch.  Summaries -> gpurun_out/prof_r03/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_trace -o p -- python $R/scripts/bench_config_c.py > $O/bench_config_C.json 2>/dev/null
f=$(find $O/c_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_config_C.csv
rm -rf $O/c_trace
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/c_pmc -o p -- python $R/scripts/bench_config_c.py > /dev/null 2>&1
c=$(find $O/c_pmc -name "*counter_collection.csv" | head -1)
[ -n "$c" ] && timeout 60 python - "$c" > $O/pmc_config_C_mfma.csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,mean")
for k, v in sorted(d.items()):
    print("%s,%s,%d,%.1f" % (k[0].replace(",", ";"), k[1], len(v), sum(v) / len(v)))
PY
rm -rf $O/c_pmc
for m in 0 4; do
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one$m -o x -- python $R/scripts/prof_one_window.py $m > /dev/null 2>&1
  f=$(find $O/one$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_one_window_$([ $m = 0 ] && echo fused || echo separate).csv
  rm -rf $O/one$m
done
ls -la $O
