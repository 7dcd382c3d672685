#!/bin/bash
# round 5: A/B of the compensated elimination of the prior blocks (OKVIS_BA_NO_LDL_COMP=1 switches it off): kernel trace by shape + bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_g
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in comp nocomp comp nocomp; do
  if [ $v = nocomp ]; then export OKVIS_BA_NO_LDL_COMP=1; else unset OKVIS_BA_NO_LDL_COMP; fi
  python $R/bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}).get('launch_us',{}); print('$v: %.0f it/s single %s launches %s' % (d['value'], d.get('single_window',{}).get('iterations_per_s'), {k: round(x['median'],1) for k,x in r.items()}))"
done
for v in comp nocomp; do
  if [ $v = nocomp ]; then export OKVIS_BA_NO_LDL_COMP=1; else unset OKVIS_BA_NO_LDL_COMP; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python $R/bench.py --no-pmc --no-cpu-baseline --no-extras --repeats 5 > /dev/null 2>&1
  t=$(find $O/$v -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$v.csv
  rm -rf $O/$v
  echo "== $v"; grep -E "solve" $O/kernel_by_shape_$v.csv | head -6
done
cd $R
for v in comp nocomp; do
  if [ $v = nocomp ]; then export OKVIS_BA_NO_LDL_COMP=1; else unset OKVIS_BA_NO_LDL_COMP; fi
  for n in 1 64; do timeout 100 python tools/gpu_solve_stamps.py $n 2>&1 | grep -E "total|LDL\^T solver" | sed "s/^/$v $n: /"; done
done
