#!/bin/bash
# Multi-rank evidence that fits a 1-GPU lease:
#  (1) one rank with the process group forced on: RCCL init + all_reduce(MAX) + all_gather execute on the GPU
#  (2) two processes sharing GPU 0 (RCCL refuses two ranks on one device -> gloo for the collective): the sharded
#      driver with world_size 2 running the real HIP path, BASELINE configs[3] (64 windows in total)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
OKVIS_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --no-pmc --no-cpu-baseline --repeats 5 \
  > gpurun_out/r02_bench_force_dist_rccl_1rank.json 2> gpurun_out/r02_bench_force_dist.err
echo "force-dist rc=$?"; tail -c 600 gpurun_out/r02_bench_force_dist_rccl_1rank.json; tail -3 gpurun_out/r02_bench_force_dist.err
OKVIS_SHARE_GPU=1 OKVIS_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --total-windows 64 --no-pmc --no-cpu-baseline --repeats 5 \
  > gpurun_out/r02_bench_2proc_1gpu_gloo.json 2> gpurun_out/r02_bench_2proc.err
echo "2-proc rc=$?"; tail -c 900 gpurun_out/r02_bench_2proc_1gpu_gloo.json; tail -3 gpurun_out/r02_bench_2proc.err
