#!/bin/bash
# round 6: sub-batch streams x hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES), 64 windows -> gpurun_out/r06_streams/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_streams
mkdir -p $O
cd $R
for q in default 8 16; do
  for n in 2 3 4 5 6 8; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 200 python bench.py --steps 20 --warmup 5 --streams $n --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 > $O/q${q}_s$n.json 2> $O/q${q}_s$n.err
    python - <<PY
import json
try:
    d = json.load(open("$O/q${q}_s$n.json"))
    print("queues $q streams $n: %.0f it/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("queues $q streams $n: failed", e)
PY
  done
done
echo done
