#!/bin/bash
# config C (BASELINE configs[2]): kernel trace + MFMA / VALU counters of the tiled Cholesky.  Summaries -> gpurun_out/prof_c/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c/trace -o p -- python $R/scripts/bench_config_c.py > $R/gpurun_out/prof_c/bench.json 2>/dev/null
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $R/gpurun_out/prof_c/pmc -o p -- python $R/scripts/bench_config_c.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for f in glob.glob(R + "/gpurun_out/prof_c/pmc/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(d.items()):
        print(k[0], k[1], len(v), sum(v) / len(v))
PY
cat $R/gpurun_out/prof_c/trace/p_kernel_stats.csv | cut -c1-140
