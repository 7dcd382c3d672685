#!/bin/bash
# round 4: default bench line (PMC passes + CPU baselines + DOGLEG / configs[2] / fp32 records), rocprofv3 kernel traces of the
# same command (eager and graph) with the per-launch-shape split, SQ + MFMA counters of an eager 64-window run, configs[2]
# trace + MFMA counters, batch sweep, marginalisation sizes, replay timing, mixed-precision study.  -> gpurun_out/prof_r04/
# (copied into profiles/r04_* afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --total-windows 64 --no-pmc --no-cpu-baseline --no-extras > $O/bench_total64.json 2> $O/bench_total64.err
for n in 1 2 4 8 16 32 64 128 256; do
  python bench.py --windows $n --no-pmc --no-cpu-baseline --no-extras --profile-steps 0 --repeats 15 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'windows': $n, 'iterations_per_s': d['value'], 'us_per_step': d['ms_per_step']*1e3}))"
done > $O/batch_sweep.jsonl
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_$mode.json 2> $O/bench_$mode.err
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  t=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$mode.csv
  rm -rf $O/$mode
done
summarise() {   # counter_collection.csv -> per kernel mean per launch
python - "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ba::"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-30s %16.0f per launch" % (c, v / max(1, n[(k, c)])))
PY
}
# SQ counters (own pass, kernel trace only) of an eager 64-window run
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o p -- \
  python $R/bench.py --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 2 --steps 20 --warmup 4 --profile-steps 0 > /dev/null 2> $O/pmc.err
c=$(find $O/pmc -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_sq_64windows.txt
rm -rf $O/pmc
# matrix-core counters of the same run (the Schur kernel of configs[1] windows is schur_mfma_kernel<3>)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc2 -o p -- \
  python $R/bench.py --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 2 --steps 20 --warmup 4 --profile-steps 0 > /dev/null 2> $O/pmc2.err
c=$(find $O/pmc2 -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_mfma_64windows.txt
rm -rf $O/pmc2
# configs[2]
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_trace -o p -- python $R/scripts/bench_config_c.py > $O/bench_config_C.json 2>/dev/null
f=$(find $O/c_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_config_C.csv
rm -rf $O/c_trace
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/c_pmc -o p -- python $R/scripts/bench_config_c.py > /dev/null 2>&1
c=$(find $O/c_pmc -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_config_C_mfma.txt
rm -rf $O/c_pmc
OKVIS_BA_SCHUR2_LARGE=1 timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/c_pmc -o p -- python $R/scripts/bench_config_c.py > $O/bench_config_C_schur_mfma.json 2>/dev/null
c=$(find $O/c_pmc -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_config_C_schur_mfma.txt
rm -rf $O/c_pmc
cd $R
timeout 150 python scripts/bench_marginalize.py --config-c > $O/bench_marginalize.json 2> $O/bench_marginalize.err
timeout 60 python tools/gpu_replay_timing.py > $O/replay_timing.txt 2>&1
timeout 200 python scripts/mixed_precision_study.py > $O/mixed_precision.json 2> $O/mixed_precision.err
for nw in 1 64; do OKVIS_BA_FUSED_MAX_WINDOWS=0 python tools/gpu_lin_stamps.py $nw 4 > $O/lin_stamps_$nw.txt 2>&1; python tools/gpu_prof_stamps.py $nw 4 > $O/stamps_$nw.txt 2>&1; done
python tools/gpu_solve_stamps.py > $O/solve_stamps.txt 2>&1
head -c 300 $O/bench_default.json; echo; head -8 $O/kernel_stats_graph.csv | cut -c1-150; cat $O/batch_sweep.jsonl; cat $O/pmc_sq_64windows.txt | head -40
