#!/bin/bash
# round 6 experiment: eager launches (no graphs) on 3-6 sub-batch streams with 4 / 8 / 16 hardware queues
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_eager
mkdir -p $O
cd $R
for q in default 8 16; do
  for n in 3 4 5 6; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 200 python bench.py --steps 20 --warmup 5 --streams $n --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 > $O/q${q}_s$n.json 2> $O/q${q}_s$n.err
    python -c "
import json
d = json.load(open('$O/q${q}_s$n.json')); print('eager, queues $q streams $n: %.0f it/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
echo done
