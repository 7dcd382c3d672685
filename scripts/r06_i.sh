#!/bin/bash
# round 6: the ADVICE tests (fenced variant, tiled solver against the referee), the early windows against the referee (printed)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_oracle_referee.py tests/test_gpu_dogleg.py -m gpu -q > $O/pytest_a.log 2>&1; tail -8 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_estimator_vs_reference.py -m gpu -q -s -k early > $O/pytest_early.log 2>&1; grep -v "^$" $O/pytest_early.log | tail -30
for w in 96 128 192 256 512; do for n in 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --windows $w --streams $n --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 > $O/b_${w}_$n.json 2> $O/b_${w}_$n.err
  python -c "
import json; d = json.load(open('$O/b_${w}_$n.json')); print('$w windows, $n streams: %.0f it/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
done; done
echo done
