#!/bin/bash
# round 5, first GPU pass: the whole GPU suite on the new solve kernel, phase stamps / quick bench lines of the library variants
# (round-4 library, new head with the old LDL^T chain, kernel-argument preload), the LDL^T micro-benchmark variants.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=240 > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
if ! grep -q " passed" $O/pytest_gpu.log || grep -q "failed" $O/pytest_gpu.log; then
  echo "--- suite on the old-LDL variant (bisect)"
  OKVIS_AMD_LIB_DIR=$R/okvis_amd/lib_variants/oldldl timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dogleg.py -m gpu -q -x > $O/pytest_oldldl.log 2>&1
  tail -4 $O/pytest_oldldl.log
fi
for v in r4 new oldldl kpre; do
  d=$R/okvis_amd/lib_variants/$v; [ $v = new ] && d=$R/okvis_amd/lib
  for n in 1 8; do
    OKVIS_AMD_LIB_DIR=$d timeout 120 python tools/gpu_solve_stamps.py $n > $O/stamps_${v}_$n.txt 2>&1
  done
  echo "== $v (1 window)"; cat $O/stamps_${v}_1.txt
done
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
for v in r4 new oldldl kpre; do
  d=$R/okvis_amd/lib_variants/$v; [ $v = new ] && d=$R/okvis_amd/lib
  OKVIS_AMD_LIB_DIR=$d timeout 300 $B > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {}).get("launch_us", {})
    print("%-8s %9.0f it/s  %.4f ms/step  launches %s  single %s" % ("$v", d["value"], d["ms_per_step"], {k: round(x["median"], 1) for k, x in r.items()}, d.get("single_window", {}).get("iterations_per_s")))
except Exception as e:
    print("$v failed", e)
PY
done
for b in ldl16_p0r0 ldl16_p1r0 ldl16_p1r1; do
  echo "== $b"; timeout 60 tests/micro/bin/$b > $O/$b.txt 2>&1; grep -E "elimination|D=150|D=174|D= 90|step" $O/$b.txt | head -16
done
timeout 60 tests/micro/bin/ldl16_trace 1 x > $O/ldl16_trace.txt 2>&1
echo done
