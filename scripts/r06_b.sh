#!/bin/bash
# round 6: the chain solver (ba_chain.hpp) — unit tests through okvis_ba_reduced_solve, windows through both solvers against the oracle,
# then the whole suite with the chain solver as the default, and the bench line.  -> gpurun_out/r06_b/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain_solve.py -m gpu -q -x -s > $O/pytest_chain.log 2>&1
tail -25 $O/pytest_chain.log
if grep -q "failed" $O/pytest_chain.log; then echo "chain tests failed: stopping"; exit 0; fi
OKVIS_AMD_DEBUG=check_patch timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
head -c 300 $O/bench_driver_args.json; echo
for n in 1 22 64; do for m in dense chain; do timeout 120 python tools/gpu_solve_stamps.py $n $m > $O/solve_stamps_${n}_$m.txt 2>&1; done; done
echo done
