#!/bin/bash
# round 6: the tests added after r06_e (timed take-back, the executed matcher binding, the stub RCCL ranks, replay with the
# reference's configuration file) -> gpurun_out/r06_f/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dogleg.py tests/test_gpu_matcher_binding.py tests/test_gpu_rccl_stub.py tests/test_gpu_replay.py -m gpu -q -x > $O/pytest_new.log 2>&1
tail -25 $O/pytest_new.log
echo done
