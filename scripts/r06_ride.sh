#!/bin/bash
# round 6: the evaluation of the IMU / prior factors riding in the decision-free Schur launch (schur_ride_kernel + small_prepare_kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_ride
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_separate_launch.py tests/test_gpu_dogleg.py tests/test_gpu_switches.py tests/test_gpu_chain_solve.py -m gpu -q -x > $O/pytest_ride.log 2>&1; tail -5 $O/pytest_ride.log
for f in 0 0x100; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 1.0 --tune flags=$f > $O/bench_$f.json 2> $O/bench_$f.err
  python -c "
import json; d = json.load(open('$O/bench_$f.json')); print('flags $f: %.0f it/s %.4f ms/step  oracle %s  rides %s' % (d['value'], d['ms_per_step'], d.get('max_rel_cost_dev_vs_oracle'), d['oracle_check']['route'].get('small_rides') if d.get('oracle_check') else None))"
  timeout 300 python bench.py --steps 20 --warmup 5 --windows 256 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 --tune flags=$f > $O/bench256_$f.json 2> $O/bench256_$f.err
  python -c "
import json; d = json.load(open('$O/bench256_$f.json')); print('flags $f, 256 windows: %.0f it/s %.4f ms/step' % (d['value'], d['ms_per_step']))"
  timeout 300 python tools/gpu_dogleg_phases.py 64 $f > $O/dogleg_phases_$f.txt 2>&1; tail -3 $O/dogleg_phases_$f.txt
done
echo done
