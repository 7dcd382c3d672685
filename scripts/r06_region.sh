#!/bin/bash
# round 6: fixed cost of one timed region (graph launches, stagger, synchronisation) — steps per region and stagger_us
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_region
mkdir -p $O
cd $R
run() { name=$1; shift
  timeout 300 python bench.py --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 "$@" > $O/$name.json 2> $O/$name.err
  python -c "
import json; d = json.load(open('$O/$name.json')); print('%-24s %.0f it/s %.4f ms/step (hip events %.4f)' % ('$name', d['value'], d['ms_per_step'], d['hip_event_ms_per_step']))"
}
run k20 --steps 20
run k50 --steps 50
run k200 --steps 200
run k20_stagger_none --steps 20 --tune stagger_us=-1
run k20_stagger5 --steps 20 --tune stagger_us=5
run k20_stagger10 --steps 20 --tune stagger_us=10
run k20_stagger30 --steps 20 --tune stagger_us=30
run k200_stagger_none --steps 200 --tune stagger_us=-1
run k200_stagger10 --steps 200 --tune stagger_us=10
timeout 300 python -m pytest tests/test_gpu_estimator_vs_reference.py -m gpu -q -k early 2>&1 | tail -3
echo done
