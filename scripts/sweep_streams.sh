#!/bin/bash
# batch throughput vs number of sub-batch streams (okvis_ba_options.n_streams) at 64 windows
cd "${GRAFT_REPO_ROOT:-.}"
for q in 4 8; do
for s in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --streams $s --no-pmc --no-cpu-baseline --repeats 5 --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('hwq', $q, 'streams', $s, round(d['value']), d['ms_per_step'])"
done
done
