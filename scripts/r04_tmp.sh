cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04_patch
OKVIS_BA_DEBUG_BUILD=1 timeout 200 python tests/gpu_replay_timing.py > gpurun_out/r04_patch/replay_sections.txt 2>&1
