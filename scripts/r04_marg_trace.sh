#!/bin/bash
# marginalisation sizes (pipeline-like ... configs[2]'s D = 750) with the kernel trace of the same command -> gpurun_out/r04_marg/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_marg
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python $R/scripts/bench_marginalize.py --config-c > $O/bench_marginalize.json 2> $O/bench_marginalize.err
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_marginalize.csv
t=$(find $O/tr -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python - "$t" > $O/kernel_trace_marginalize_last_call.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last okvis_ba_marginalize call of the run is the D = 750 one: from its last marg_dense_kernel launch on
last = max(i for i, r in enumerate(rows) if "marg_dense_kernel" in r["Kernel_Name"])
first = last
while first > 0 and "linearize" not in rows[first]["Kernel_Name"] and "small_kernel" not in rows[first]["Kernel_Name"]:
    first -= 1
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%9.1f us  +%8.1f us  %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Grid_Size", "")))
PY
rm -rf $O/tr
tail -c 900 $O/bench_marginalize.json; echo; cat $O/kernel_trace_marginalize_last_call.txt
