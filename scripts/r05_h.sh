#!/bin/bash
# round 5: one Newton step in the reciprocal of the pivots of the blocks that are not compensated (variant nr1) against the default:
# distances to the long double referee, the GPU suite, the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
export TMPDIR=/tmp
export OKVIS_AMD_LIB_DIR=okvis_amd/lib_variants/nr1
timeout 200 python tools/gpu_referee_spread.py > $O/spread41_nr1.txt 2>&1; cat $O/spread41_nr1.txt
timeout 200 python tools/gpu_sweep_gaps.py > $O/sweep_gaps_nr1.txt 2>&1
python - <<PY
import re
g=[]; o=[]
for l in open("$O/sweep_gaps_nr1.txt"):
    m=re.search(r"GPU ([0-9.e+-]+) oracle ([0-9.e+-]+)", l)
    if m: g.append(float(m.group(1))); o.append(float(m.group(2)))
import statistics as st
print("sweep vs referee: GPU median %.1e max %.1e | oracle median %.1e max %.1e" % (st.median(g), max(g), st.median(o), max(o)))
PY
timeout 400 python -m pytest tests -m gpu -q -x --timeout=240 2>&1 | grep -E "passed|failed|^FAILED" | head -3
for v in nr1 default nr1 default; do
  if [ $v = default ]; then unset OKVIS_AMD_LIB_DIR; else export OKVIS_AMD_LIB_DIR=okvis_amd/lib_variants/nr1; fi
  python bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}).get('launch_us',{}); print('$v: %.0f it/s single %s solve %s' % (d['value'], d.get('single_window',{}).get('iterations_per_s'), r.get('solve',{}).get('median')))"
done
