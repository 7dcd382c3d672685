#!/bin/bash
# SQ counters of the linearise kernel on a saturated batch (256 windows, eager, one stream).
# Separate --pmc passes (8 SQ counters per pass); summary -> gpurun_out/prof_lin/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-256}
OUT=$R/gpurun_out/prof_lin
mkdir -p $OUT
CMD="python $R/bench.py --windows $W --streams 1 --steps 20 --warmup 30 --no-graph --no-cpu-baseline --no-pmc --profile-steps 0"
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>$OUT/p1.err
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/p2 -o p -- $CMD > /dev/null 2>$OUT/p2.err
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -o p -- $CMD > /dev/null 2>$OUT/p3.err
python - <<PY > $OUT/summary.txt
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("p1", "p2", "p3"):
    for f in glob.glob(R + "/gpurun_out/prof_lin/%s/**/*counter_collection.csv" % p, recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            d[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(d.items()):
            v = v[len(v) // 2:]          # steady state: second half of the launches
            print(p, k[0], k[1], len(v), sum(v) / len(v))
PY
cat $OUT/summary.txt
tail -n 3 $OUT/p1.err; tail -n 3 $OUT/p3.err
