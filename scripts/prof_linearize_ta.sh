#!/bin/bash
# vector-memory path counters (TA / TCP) of the linearise kernel on a saturated batch; summary -> gpurun_out/prof_lin_ta/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-256}
OUT=$R/gpurun_out/prof_lin_ta
mkdir -p $OUT
CMD="python $R/bench.py --windows $W --streams 1 --steps 20 --warmup 30 --no-graph --no-cpu-baseline --no-pmc --profile-steps 0"
# at most two counters of one block per pass (more: "exceeds the capabilities of the hardware" and the tool hangs) -> timeout
timeout 100 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>$OUT/p1.err
timeout 100 rocprofv3 --pmc TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/p2 -o p -- $CMD > /dev/null 2>$OUT/p2.err
python - <<PY > $OUT/summary.txt
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("p1", "p2"):
    for f in glob.glob(R + "/gpurun_out/prof_lin_ta/%s/**/*counter_collection.csv" % p, recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "linearize" not in r["Kernel_Name"] and "schur" not in r["Kernel_Name"]: continue
            d[(r["Kernel_Name"][:32], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(d.items()):
            v = v[len(v) // 2:]
            print(p, k[0], k[1], len(v), sum(v) / len(v))
PY
cat $OUT/summary.txt; tail -2 $OUT/p1.err
