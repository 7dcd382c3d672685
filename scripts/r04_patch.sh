#!/bin/bash
# the estimator's patch route on the GPU: the estimator / replay / adapter / patch tests with every window checked against a fresh
# flatten (OKVIS_AMD_CHECK_PATCH), the helper time-out test, and the per-frame timing of the replay with and without patches
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_patch
mkdir -p $OUT
export TMPDIR=/tmp
OKVIS_AMD_CHECK_PATCH=1 timeout 1500 python -m pytest tests/test_gpu_patch.py tests/test_gpu_estimator.py tests/test_gpu_replay.py tests/test_gpu_adapter.py tests/test_gpu_estimator_vs_reference.py -m gpu -q > $OUT/pytest_estimator.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest_estimator.log | tail -8
timeout 600 python -m pytest tests/test_gpu_structure_paths.py -m gpu -q -x -k "helper" > $OUT/pytest_helpers.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest_helpers.log | tail -4
timeout 200 python tools/gpu_replay_timing.py > $OUT/replay_timing.txt 2>&1
grep -E "medians|route" $OUT/replay_timing.txt
OKVIS_BA_DEBUG_BUILD=1 timeout 200 python tools/gpu_replay_timing.py > $OUT/replay_sections.txt 2>&1
grep -E "build_window|route" $OUT/replay_sections.txt | head -4
