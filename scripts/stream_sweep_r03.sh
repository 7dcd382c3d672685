#!/bin/bash
# diagnostics: sub-batch stream count at 64 windows and two more batch sizes (one JSON line each -> value, ms per step)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for n in 2 3 4; do timeout 60 python bench.py --windows 64 --streams $n --no-pmc --no-cpu-baseline --no-extras --profile-steps 0 --repeats 15 2>/dev/null | p "streams=$n"; done
for w in 48 96; do timeout 60 python bench.py --windows $w --no-pmc --no-cpu-baseline --no-extras --profile-steps 0 --repeats 15 2>/dev/null | p "windows=$w"; done
