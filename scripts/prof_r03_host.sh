#!/bin/bash
# round 3, third part: what surrounds the iterations of a frame - okvis_ba_marginalize on four sub-window sizes (LDS path and HBM
# workspace), the per-frame split of the C++ replay, the phase stamps of the dense marginalisation kernel.  -> gpurun_out/prof_r03/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd $R
timeout 150 python scripts/bench_marginalize.py --config-c > $O/bench_marginalize.json 2> $O/bench_marginalize.err
timeout 60 python tools/gpu_replay_timing.py > $O/replay_timing.txt 2>&1
timeout 40 python tools/gpu_marg_stamps.py > $O/marg_stamps.txt 2>&1
tail -c 600 $O/bench_marginalize.json; tail -4 $O/replay_timing.txt
