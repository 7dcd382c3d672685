#!/bin/bash
# Round 4: parity of the piece path (full GPU suite), then A/B of the linearise variants.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
run() { name=$1; shift; ( "$@" ) > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {}).get("launch_us", {})
    print("%-22s %9.0f it/s  %.4f ms/step  launches %s  single %s" % ("$name", d["value"], d["ms_per_step"], {k: round(v["median"], 1) for k, v in r.items()}, d.get("single_window", {}).get("iterations_per_s")))
except Exception as e:
    print("$name failed", e)
PY
}
run old64         env OKVIS_BA_NO_LIN2=1 $B
run lin2_gpw1     env OKVIS_BA_LIN2_GPW=1 $B
run lin2_gpw1_occ4 env OKVIS_BA_LIN2_GPW=1 OKVIS_BA_LIN2_OCC=4 $B
run lin2_gpw2     env OKVIS_BA_LIN2_GPW=2 $B
run lin2_gpw3     env OKVIS_BA_LIN2_GPW=3 $B
run lin2_auto     $B
run lin2_fork     env OKVIS_BA_LIN2_GPW=1 OKVIS_BA_SMALL_FORK=1 $B
run lin2_comb     env OKVIS_BA_SPLIT_SMALL_MIN=100000 $B
run lin2_fused64  env OKVIS_BA_FUSED_MAX_WINDOWS=64 OKVIS_BA_SPLIT_SMALL_MIN=100000 $B
run old256        env OKVIS_BA_NO_LIN2=1 $B --windows 256
run lin2_256_gpw1 env OKVIS_BA_LIN2_GPW=1 $B --windows 256
run lin2_256_gpw4 env OKVIS_BA_LIN2_GPW=4 $B --windows 256
run lin2_256_auto $B --windows 256
run old1          env OKVIS_BA_NO_LIN2=1 $B --windows 1
run lin2_1        $B --windows 1
run lin2_8        $B --windows 8
run old8          env OKVIS_BA_NO_LIN2=1 $B --windows 8
for nw in 1 64 256; do
  OKVIS_BA_LIN2_GPW=1 python tests/gpu_lin_stamps.py $nw 4 > $OUT/stamps_lin2_$nw.txt 2>&1
done
python tests/gpu_lin_stamps.py 1 0 > $OUT/stamps_lin2_fused_1.txt 2>&1
cat $OUT/stamps_lin2_1.txt $OUT/stamps_lin2_64.txt $OUT/stamps_lin2_256.txt $OUT/stamps_lin2_fused_1.txt
