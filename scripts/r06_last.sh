#!/bin/bash
# round 6, last check of the committed tree: smoke(), the whole GPU suite, the bench line with the driver's arguments
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_last
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python -c "
import json; d = json.load(open('$O/bench_driver_args.json')); print('%.0f it/s %.4f ms/step single %.0f dogleg %.0f configC %.3f oracle %.1e' % (d['value'], d['ms_per_step'], d['single_window']['iterations_per_s'], d['dogleg']['iterations_per_s'], d['config_C']['ms_per_iteration'], d['max_rel_cost_dev_vs_oracle']))"
echo done
