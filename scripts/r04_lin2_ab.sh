#!/bin/bash
# Round 4: parity (full GPU suite), then A/B of the kernel variants.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r04_ab
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc " $OUT/pytest.log | tail -12
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
run vschur_64     env OKVIS_BA_NO_SCHUR2=1 $B
run mfma_64       $B
run mfma_256      $B --windows 256
run vschur_256    env OKVIS_BA_NO_SCHUR2=1 $B --windows 256
run mfma_128      $B --windows 128
run mfma_32_sep   env OKVIS_BA_FUSED_MAX_WINDOWS=0 $B --windows 32
run mfma_32       $B --windows 32
run mfma_8_sep    env OKVIS_BA_FUSED_MAX_WINDOWS=0 $B --windows 8
run mfma_8        $B --windows 8
run mfma_1_sep    env OKVIS_BA_FUSED_MAX_WINDOWS=0 $B --windows 1
run mfma_1        $B --windows 1
python scripts/bench_config_c.py > $OUT/config_c.json 2> $OUT/config_c.err; tail -3 $OUT/config_c.json | cut -c1-600
OKVIS_BA_NO_SCHUR2=1 python scripts/bench_config_c.py > $OUT/config_c_vschur.json 2> $OUT/config_c_vschur.err; tail -3 $OUT/config_c_vschur.json | cut -c1-600
