#!/bin/bash
# kernel trace of the C++ replay (80 frames): per kernel the calls, median and total time -> gpurun_out/r04_replay_trace/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_replay_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python - <<PY
import sys
sys.path.insert(0, "$R")
from okvis_amd import recording
recording.write_synthetic_recording("/tmp/rec_r04", duration_s=8.0)
PY
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- $R/okvis_amd/lib/okvis_amd_replay /tmp/rec_r04 > $O/replay.txt 2>&1
tail -4 $O/replay.txt
t=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY' > $O/kernels.txt
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    wg = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * (int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]))
    rows[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} {'calls':>6s} {'median us':>10s} {'total ms':>9s}")
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{name[:70]:70s} {len(v):6d} {v[len(v) // 2]:10.1f} {sum(v) / 1e3:9.2f}")
PY
rm -rf $O/tr
cat $O/kernels.txt
