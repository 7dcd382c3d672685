#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tests/micro/dpp_shift.hip -o /tmp/dpp_shift && /tmp/dpp_shift 2>&1 | tee gpurun_out/r04_diag/dpp.txt
python tools/gpu_lin2_diag.py 2>&1 | tee gpurun_out/r04_diag/diag_A.txt
python tools/gpu_lin2_diag.py small 2>&1 | tee gpurun_out/r04_diag/diag_small.txt
