#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dogleg.py tests/test_gpu_structure_paths.py tests/test_gpu_estimator.py tests/test_gpu_estimator_vs_reference.py -x -q 2>&1 | tail -15
python bench.py --no-pmc --no-cpu-baseline --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('batch', round(d['value']), d['ms_per_step'], 'single', d['single_window'], d['roofline']['per_kernel_us'])"
