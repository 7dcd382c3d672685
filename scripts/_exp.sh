cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_cache; mkdir -p $O
export TMPDIR=/tmp
OKVIS_AMD_CHECK_PATCH=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -6
timeout 200 python tests/gpu_replay_timing.py > $O/replay_timing.txt 2>&1
grep -E "medians|route" $O/replay_timing.txt
