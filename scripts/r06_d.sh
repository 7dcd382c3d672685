#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
for n in 1; do for m in chain; do timeout 120 python tools/gpu_solve_stamps.py $n $m > $O/solve_stamps_${n}_$m.txt 2>&1; done; done
grep -h -A3 "chain solver" $O/solve_stamps_1_chain.txt | cut -c1-300
