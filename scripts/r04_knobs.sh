#!/bin/bash
# A/B of launch-shape knobs on the 64-window bench (fixed-radius loop) and on the DOGLEG optimize(10) call, all in one lease:
# fused mode beyond 48 windows, IMU / prior factors inside the linearise launch.  -> gpurun_out/r04_knobs/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_knobs
mkdir -p $O
cd $R
run() {   # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-extras --no-pmc --no-cpu-baseline --repeats 12 > $O/bench_$name.json 2> $O/bench_$name.err
  env "$@" python scripts/r04_dogleg_trace.py run > $O/dogleg_$name.txt 2>&1
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
d = json.load(open(f"gpurun_out/r04_knobs/bench_{n}.json"))
dl = [l for l in open(f"gpurun_out/r04_knobs/dogleg_{n}.txt") if l.startswith("batch")]
print(f"{n:28s} {d['value'] / 1e3:7.1f} k it/s  {d['ms_per_step'] * 1e3:6.1f} us/step   single {d['single_window']['ms_per_iteration'] * 1e3:5.1f} us   dogleg {dl[-1].split('optimize(10)')[1].split(',')[0].strip()}")
PY
}
run default OKVIS_DUMMY=1
run fused64 OKVIS_BA_FUSED_MAX_WINDOWS=64
run fused64_unsplit OKVIS_BA_FUSED_MAX_WINDOWS=64 OKVIS_BA_SPLIT_SMALL_MIN=100000
run unsplit OKVIS_BA_SPLIT_SMALL_MIN=100000
