#!/bin/bash
# round 6: DOGLEG runs with the factor launch on a side stream per sub-batch (two sub-batches + two branches)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_fork2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_separate_launch.py tests/test_gpu_dogleg.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for f in 0 0x200; do
  timeout 300 python tools/gpu_dogleg_phases.py 64 $f > $O/dogleg_phases_$f.txt 2>&1; echo "dogleg, flags $f"; tail -4 $O/dogleg_phases_$f.txt
done
for f in 0 0x400; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 1.0 --tune flags=$f > $O/bench_$f.json 2> $O/bench_$f.err
  python -c "
import json; d = json.load(open('$O/bench_$f.json')); print('GN bench, flags $f: %.0f it/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
done
echo done
