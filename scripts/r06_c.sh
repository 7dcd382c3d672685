#!/bin/bash
# round 6: quick loop on the chain solver — its tests and the solve kernel's phase stamps in both modes.  -> gpurun_out/r06_c/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain_solve.py -m gpu -q -x -s > $O/pytest_chain.log 2>&1
tail -8 $O/pytest_chain.log
for n in 1 64; do for m in dense chain; do timeout 120 python tools/gpu_solve_stamps.py $n $m > $O/solve_stamps_${n}_$m.txt 2>&1; done; done
grep -h "chain solver\|LDL^T solver\|LDL^T + back\|total" $O/solve_stamps_*.txt | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-extras --repeats 20 --min-timed-s 0.5 --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 64 windows', d['value'], d['ms_per_step'], d['single_window'])"
echo done
