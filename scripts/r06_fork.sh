#!/bin/bash
# round 6 experiment: the IMU / prior factor launch on a side branch of the graph (OKVIS_BA_TUNE_FORK_SMALL)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_fork
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_separate_launch.py -m gpu -q -x 2>&1 | tail -3
for f in 0 0x100; do
  timeout 300 python tools/gpu_dogleg_phases.py 64 $f > $O/dogleg_phases_$f.txt 2>&1; tail -4 $O/dogleg_phases_$f.txt
  timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 --tune flags=$f > $O/bench_$f.json 2> $O/bench_$f.err
  python -c "
import json; d = json.load(open('$O/bench_$f.json')); print('flags $f: %.0f it/s %.4f ms/step  oracle %s' % (d['value'], d['ms_per_step'], d.get('max_rel_cost_dev_vs_oracle')))"
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 256 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 --tune flags=$f > $O/bench256_$f.json 2> $O/bench256_$f.err
  python -c "
import json; d = json.load(open('$O/bench256_$f.json')); print('flags $f, 256 windows: %.0f it/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
done
echo done
