#!/bin/bash
# round 5: compensated elimination (variant comp_all) against the default build — micro-benchmark, spread against the long double
# referee, the tests that are sensitive to the reduced solve
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
export TMPDIR=/tmp
for b in ldl16 ldl16_comp; do echo "== $b"; timeout 60 tests/micro/bin/$b 2>&1 | head -6; done
for v in default comp_all; do
  if [ $v = default ]; then unset OKVIS_AMD_LIB_DIR; else export OKVIS_AMD_LIB_DIR=okvis_amd/lib_variants/$v; fi
  echo "== $v"
  timeout 300 python tools/gpu_referee_spread.py > $O/spread41_$v.txt 2>&1; cat $O/spread41_$v.txt
  timeout 600 python -m pytest tests/test_gpu_random_sweep.py tests/test_gpu_dogleg.py tests/test_gpu_parity.py tests/test_gpu_dense_solve.py -m gpu -q --timeout=240 > $O/pytest_$v.log 2>&1
  grep -E "passed|failed" $O/pytest_$v.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_$v.log | head
  timeout 100 python tools/gpu_solve_stamps.py 1 2>&1 | grep -E "total|LDL"
done
