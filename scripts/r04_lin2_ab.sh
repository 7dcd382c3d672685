#!/bin/bash
# Round 4, first GPU pass: parity of the piece path (full GPU suite), then A/B of the linearise variants.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --no-pmc --no-extras --no-cpu-baseline --repeats 12"
run() { name=$1; shift; echo "== $name"; ( "$@" ) > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$name", d["value"], d["ms_per_step"], {k: r.get(k) for k in ("launch_us",)} if r else "")
except Exception as e:
    print("$name failed", e)
PY
}
run old64      env OKVIS_BA_NO_LIN2=1 $B
run lin2_occ3  env OKVIS_BA_LIN2_OCC=3 $B
run lin2_occ4  env OKVIS_BA_LIN2_OCC=4 $B
run lin2_comb  env OKVIS_BA_SPLIT_SMALL_MIN=100000 $B
run lin2_fused64 env OKVIS_BA_FUSED_MAX_WINDOWS=64 $B
run lin2_fused64_occ env OKVIS_BA_FUSED_MAX_WINDOWS=64 OKVIS_BA_SPLIT_SMALL_MIN=100000 $B
run old256     env OKVIS_BA_NO_LIN2=1 $B --windows 256
run lin2_256_occ3 env OKVIS_BA_LIN2_OCC=3 $B --windows 256
run lin2_256_occ4 env OKVIS_BA_LIN2_OCC=4 $B --windows 256
run old1       env OKVIS_BA_NO_LIN2=1 $B --windows 1
run lin2_1     $B --windows 1
run lin2_8     $B --windows 8
run old8       env OKVIS_BA_NO_LIN2=1 $B --windows 8
for nw in 1 64 256; do
  python tests/gpu_lin_stamps.py $nw 4 > $OUT/stamps_lin2_$nw.txt 2>&1
  python tests/gpu_lin_stamps.py $nw 12 > $OUT/stamps_old_$nw.txt 2>&1
done
OKVIS_BA_LIN2_OCC=4 python tests/gpu_lin_stamps.py 256 4 > $OUT/stamps_lin2_256_occ4.txt 2>&1
python tests/gpu_lin_stamps.py 1 0 > $OUT/stamps_lin2_fused_1.txt 2>&1
cat $OUT/stamps_lin2_1.txt $OUT/stamps_old_1.txt $OUT/stamps_lin2_256.txt $OUT/stamps_old_256.txt
