#!/bin/bash
# round 6 experiment: sub-batch streams of mixed priorities (the runtime's hardware-queue pools are per priority)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_streams2
mkdir -p $O
cd $R
for q in default 8; do
  for m in 1 2 3; do
    for n in 3 4 5 6 8; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      OKVIS_BA_EXP_PRIO=$m timeout 200 python bench.py --steps 20 --warmup 5 --streams $n --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 > $O/q${q}_m${m}_s$n.json 2> $O/q${q}_m${m}_s$n.err
      python - <<PY
import json
try:
    d = json.load(open("$O/q${q}_m${m}_s$n.json"))
    print("queues $q prio-mode $m streams $n: %.0f it/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("queues $q prio-mode $m streams $n: failed", e)
PY
    done
  done
done
echo done
