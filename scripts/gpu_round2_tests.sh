cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dogleg.py -x -q 2>&1 | tail -30 > gpurun_out/dogleg1.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/gpu_all1.log
tail -5 gpurun_out/dogleg1.log; tail -5 gpurun_out/gpu_all1.log
