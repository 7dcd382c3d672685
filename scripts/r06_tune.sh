#!/bin/bash
# round 6 experiment: launch-shape knobs of okvis_ba_options::tuning at 64 and 256 windows (bench.py --tune)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_tune
mkdir -p $O
cd $R
run() {  # name windows args...
  name=$1; w=$2; shift 2
  timeout 300 python bench.py --steps 20 --warmup 5 --windows $w --no-pmc --no-cpu-baseline --no-extras --repeats 5 --min-timed-s 0.5 "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json"))
    print("%-40s %4d windows: %.0f it/s  %.4f ms/step  bytes_impl/window-iteration %s" % ("$name", $w, d["value"], d["ms_per_step"], d["roofline"]["step"].get("bytes_impl_per_window_iteration")))
except Exception as e:
    print("$name: failed", e)
PY
}
run base64 64
run fused64 64 --tune fused_max_windows=64
run fused64_s2 64 --tune fused_max_windows=64 --streams 2
run glm16_64 64 --tune group_lm=16
run glm64_64 64 --tune group_lm=64
run split_never64 64 --tune split_small_min=1000
run occ3_64 64 --tune lin2_occupancy=3
run dense64 64 --tune solve_mode=1
run base256 256
run fused256 256 --tune fused_max_windows=256
run glm64_256 256 --tune group_lm=64
run s2_256 256 --streams 2
run s1_256 256 --streams 1
run base128 128
run fused128 128 --tune fused_max_windows=128
echo done
